"""[r06] where a wave of the round-6 forward kernel spends its cycles: per-block sums of shader-clock intervals of wave 0 (ATTN_STAMP
build: DALLE_HIP_LIB=tools/_build/libdalle_hip_stamp.so).  Intervals: top (DMA issue + rescale check), phase A (S MFMAs + exp), phase B
(PV MFMAs + row max), barrier (vmcnt(0) + s_barrier), prologue, epilogue, idle steps, other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import numpy as np, torch, dalle_hip as dh
from kbench import rb
B, H, S = 32, 4, 1280
d = H * 128
qkv = rb(B * S, 3 * d, scale=0.3)
o = torch.empty(B * S, d, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
dh.set_option("attn_fwd", 1)
for _ in range(5):
    dh.attention_fwd(qkv, o, lse, B, H, S)
buf = torch.zeros(512 * 12, dtype=torch.int64, device="cuda")
dh.set_debug_buffer(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dh.attention_fwd(qkv, o, lse, B, H, S); e1.record(); torch.cuda.synchronize()
dh.set_debug_buffer(None)
a = buf.cpu().numpy().reshape(512, 12).astype(np.float64)
names = ["top(dma+check)", "phase A", "phase B", "barrier", "prologue", "epilogue", "#steps", "idle steps", "other", "#items", "kernel"]
print(f"kernel {e0.elapsed_time(e1)*1e3:.1f} us (stamped build)")
steps, items = a[:, 6].mean(), a[:, 9].mean()
print(f"per block (wave 0): {steps:.1f} active steps, {items:.1f} items, life {a[:,10].mean():.0f} ticks")
for i in (0, 1, 2, 3):
    print(f"  {names[i]:16s} {a[:, i].mean():10.0f} ticks total = {a[:, i].sum() / a[:, 6].sum():8.1f} per step")
for i in (4, 5, 7, 8):
    print(f"  {names[i]:16s} {a[:, i].mean():10.0f} ticks total = {a[:, i].sum() / a[:, 9].sum():8.1f} per item")
print("  sum of parts / life:", a[:, [0, 1, 2, 3, 4, 5, 7, 8]].sum() / a[:, 10].sum())
