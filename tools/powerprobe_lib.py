"""rocm-smi power / clock sampler around a kernel loop (shared by tools/powerprobe.py and tools/dvfsprobe.py)."""
import re
import subprocess
import threading
import time

import torch


def sample(out, stop):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            p = re.search(r"Power \(W\): ([\d.]+)", txt)
            s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
            out.append((float(p.group(1)) if p else -1, int(s.group(1)) if s else -1))
        except Exception as e:
            out.append((-1, -1))
        time.sleep(0.3)


def run(name, fn, flops, secs=5.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out, stop = [], threading.Event()
    th = threading.Thread(target=sample, args=(out, stop)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop.set(); th.join()
    out = [o for o in out[2:] if o[0] > 0]
    pw = sum(o[0] for o in out) / max(len(out), 1); ck = sum(o[1] for o in out) / max(len(out), 1)
    print(f"{name:34s}: {dt/n*1e6:8.1f} us  {flops/(dt/n)/1e12:7.1f} TF/s  power {pw:6.0f} W  sclk {ck:5.0f} MHz  ({len(out)} samples)", flush=True)


