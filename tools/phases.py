"""Per-block phase timeline of the 256x128 NT GEMM (shader-clock stamps written by the kernel when a debug buffer is set):
prologue (first tile load), main loop, epilogue; co-residency and gaps per (XCD, CU) slot."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import numpy as np, torch, dalle_hip as dh
from kbench import rb
M, N, K = 40960, int(sys.argv[1]) if len(sys.argv) > 1 else 50816, int(sys.argv[2]) if len(sys.argv) > 2 else 512
MODE = sys.argv[3] if len(sys.argv) > 3 else "bias"     # bias | softmax | tn
A, Bt, bias = rb(M, K), rb(N, K, scale=0.05), rb(N)
C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
parts = torch.empty(dh.gemm_nt_softmax_partials(N), M, dtype=torch.float32, device="cuda")
dh.set_option("nt4", 2)
dh.set_option("nt8p", 0)


def launch():
    if MODE == "softmax":
        dh.gemm_nt_softmax(A, K, Bt, K, bias, None, C, N, parts, M, N, K)
    else:
        dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, 1, bias=bias)


def nt_phases(lds):
    """lds = dynamic LDS requested per block: 49152 -> 3 blocks / CU, 65536 -> 2, 98304 -> 1"""
    dh.set_option("nt4_lds", lds)
    for _ in range(3):
        launch()
    nblk = (M // 256) * ((N + 127) // 128)
    dbg = torch.zeros(nblk, 6, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 1e3
    dh.set_debug_buffer(dbg)
    e0.record(); launch(); e1.record()
    torch.cuda.synchronize()
    dh.set_debug_buffer(None)
    d = dbg.cpu().numpy().astype(np.int64)
    t0, t1, t2, t3, hw, t4 = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 5]
    us = e0.elapsed_time(e1) * 1e3
    print(f"--- {MODE} N={N} K={K} lds/block {lds}: kernel {plain_us:.1f} us plain, {us:.1f} us stamped, {nblk} blocks; {2*M*N*K/plain_us/1e6:.0f} TF/s")
    tick_per_us = (t4.max() - t0.min()) / us
    print(f"ticks/us ~ {tick_per_us:.1f}")
    for name, v in (("prologue", t1 - t0), ("main loop", t2 - t1), ("per k-step(32)", (t2 - t1) / (K / 32)), ("epilogue issue", t3 - t2), ("store drain", t4 - t3), ("block life", t4 - t0)):
        print(f"{name:15s}: mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f} ticks  ({v.mean()/tick_per_us:.2f} us)")
    # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; use cu/sh/se + xcc via block id % 8
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
    slot = (np.arange(nblk) % 8) * 1024 + cu
    nslots = len(set(slot.tolist()))
    life = (t4 - t0).sum()
    mfma_ticks = (K / 32) * 32 * 16.0   # per wave: K/32 k-steps x 32 MFMAs 16x16x32 x 16 cycles
    print(f"distinct (xcd,cu) seen: {nslots}; mean co-resident blocks per CU = {life/(nslots*(t4.max()-t0.min())):.2f}; "
          f"own MFMA issue / block life = {mfma_ticks/(t4-t0).mean():.3f}")


def nt8p_phases():
    dh.set_option("nt4", 0); dh.set_option("nt8p", 2)
    for _ in range(3):
        launch()
    dbg = torch.zeros(256, 6, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 1e3
    dh.set_debug_buffer(dbg)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    dh.set_debug_buffer(None)
    dh.set_option("nt4", 2); dh.set_option("nt8p", 0)   # (this tool's other sections force the 256x128 kernel)
    d = dbg.cpu().numpy().astype(np.float64)
    d = d[d[:, 4] > 0]
    life, main, epi, first, n = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    print(f"--- persistent 256x256 {MODE} N={N} K={K}: kernel {plain_us:.1f} us plain ({2*M*N*K/plain_us/1e6:.0f} TF/s), {e0.elapsed_time(e1)*1e3:.1f} us stamped; "
          f"{len(d)} blocks, {n.mean():.1f} tiles each; ticks/us {life.mean()/(e0.elapsed_time(e1)*1e3):.0f}")
    print(f"per tile: life {np.mean(life/n):.0f}  main loop {np.mean(main/n):.0f} (per k-step(64) {np.mean(main/n)/(K/64):.0f}; first k-step incl. store drain {np.mean(first/n):.0f})  "
          f"epilogue issue {np.mean(epi/n):.0f}; own MFMA issue per wave {K/64*64*16:.0f}")


if MODE == "softmax8p" or MODE == "bias8p":
    MODE = MODE[:-2]
    nt8p_phases()
    sys.exit(0)
if MODE != "tn":
    for lds in (49152, 65536, 98304):
        nt_phases(lds)
    dh.set_option("nt4_lds", 49152)


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


if MODE == "tn":
    tn_phases(2048, 512)
    tn_phases(512, 1536)
    tn_phases(512, 512)
