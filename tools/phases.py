"""Per-block phase timeline of the 256x128 NT GEMM (shader-clock stamps written by the kernel when a debug buffer is set):
prologue (first tile load), main loop, epilogue; co-residency and gaps per (XCD, CU) slot."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import numpy as np, torch, dalle_hip as dh
from kbench import rb
M, N, K = 40960, int(sys.argv[1]) if len(sys.argv) > 1 else 50816, int(sys.argv[2]) if len(sys.argv) > 2 else 512
A, Bt, bias = rb(M, K), rb(N, K, scale=0.05), rb(N)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
dh.set_option("nt4", 2)
for _ in range(2):
    dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 1, bias=bias)
nblk = (M // 256) * ((N + 127) // 128)
dbg = torch.zeros(nblk, 5, dtype=torch.int64, device="cuda")
dh.set_debug_buffer(dbg)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 1, bias=bias)
e1.record()
torch.cuda.synchronize()
dh.set_debug_buffer(None)
d = dbg.cpu().numpy().astype(np.int64)
t0, t1, t2, t3, hw = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
print(f"kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} blocks; span of stamps {(t3.max()-t0.min())/1e3:.1f} k ticks")
tick_per_us = (t3.max() - t0.min()) / (e0.elapsed_time(e1) * 1e3)
print(f"ticks/us ~ {tick_per_us:.1f}")
for name, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("epilogue", t3 - t2), ("block life", t3 - t0)):
    print(f"{name:10s}: mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f} ticks  ({v.mean()/tick_per_us:.2f} us)")
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; use cu/sh/se + xcc via block id % 8
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
xcd = np.arange(nblk) % 8
slot = xcd * 1024 + cu
order = np.argsort(t0)
gaps, conc = [], []
by = {}
for i in order:
    by.setdefault(slot[i], []).append(i)
nslots = len(by)
life = 0
for s_, lst in by.items():
    ends = sorted(t3[lst]); starts = sorted(t0[lst])
    life += (t3[lst] - t0[lst]).sum()
print(f"distinct (xcd,cu) seen: {nslots}; mean co-resident blocks per CU = total block life / (CUs x span) = {life/(nslots*(t3.max()-t0.min())):.2f}")


def tn_phases(I, J):
    X, dY = rb(M, I), rb(M, J)
    dW = torch.empty(I, J, dtype=torch.float32, device="cuda")
    w = torch.empty(dh.gemm_tn_workspace_bytes(M, I, J) + 1024, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        dh.gemm_tn(X, I, dY, J, dW, M, I, J, w)
    dbg = torch.zeros(8192, 8, dtype=torch.int64, device="cuda")
    dh.set_debug_buffer(dbg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dh.gemm_tn(X, I, dY, J, dW, M, I, J, w)
    e1.record()
    torch.cuda.synchronize()
    dh.set_debug_buffer(None)
    d = dbg.cpu().numpy().astype(np.int64)
    d = d[d[:, 3] != 0]
    t0, t1, t2, t3, nt = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    ghz = np.median((t3 - t0) / np.maximum(d[:, 6] - d[:, 5], 1) * 0.1)
    print(f"cycle counter runs at {ghz:.3f} ticks/ns (against the 100 MHz realtime counter)")
    print(f"TN {I}x{J}: kernel+reduce {e0.elapsed_time(e1)*1e3:.1f} us, {len(d)} blocks, {int(np.median(nt))} steps/block")
    for name, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("per step", (t2 - t1) / np.maximum(nt, 1)), ("epilogue", t3 - t2), ("block life", t3 - t0)):
        print(f"  {name:10s}: mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f} ticks")


if len(sys.argv) > 3 and sys.argv[3] == "tn":
    tn_phases(2048, 512)
    tn_phases(512, 1536)
    tn_phases(512, 512)
