"""Builds libdalle_hip.so (hand-written HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so it travels with the
repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libdalle_hip.so")
SOURCES = ["elementwise.hip", "gemm.hip", "attention.hip", "vae.hip", "comm.hip", "host.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/dalle_hip.h"]:
        p = os.path.join(CSRC, name)
        if os.path.isfile(p):
            h.update(name.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [_hipcc()] + FLAGS + ["-c", sp, "-o", obj]
        if verbose:
            print("[dalle_hip.build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed for {src}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print("[dalle_hip.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    open(stamp, "w").write(dig)
    return LIB


def build_variant(tag, defines, verbose=False):
    """an experiment build of the same sources with extra -D switches: tools/_build/libdalle_hip_<tag>.so (travels with the
    gpurun snapshot, never loaded by the product; select with DALLE_HIP_LIB for same-call A/B runs, tools/ab.sh)"""
    out_dir = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools", "_build")
    objdir = os.path.join(out_dir, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(out_dir, f"libdalle_hip_{tag}.so")
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [_hipcc()] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed for {src} ({tag})")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    if verbose:
        print("[dalle_hip.build] variant", lib)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # python build.py --variant <tag> DEFINE [DEFINE ...]
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build(force="--force" in sys.argv))
