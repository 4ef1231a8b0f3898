"""Kernel micro-benchmarks on the GPU box (torch.cuda.Event timing on the launch stream, random data).
Prints one line per kernel/shape: time, achieved TFLOP/s or GB/s.  Used to steer optimisation; the
judged numbers come from bench.py + rocprofv3 summaries under profiles/.
usage: python tools/kbench.py [nt] [tn] [attn] [head] [misc]   (default: all)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
import dalle_hip as dh

DEV = "cuda"


def timeit(fn, iters=int(os.environ.get('KB_ITERS', '20')), warm_ms=float(os.environ.get('KB_WARM_MS', '60'))):
    """warm-up by TIME, not by count: from idle the clocks take ~20 ms of work to settle (attention_bwd measured 334 -> 302 ->
    293 us over three consecutive rounds of 23 calls in one process), which a fixed 3 calls does not cover"""
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


_FLUSH = {}


def timeit_cold(fn, warm=None, flush_mb=int(os.environ.get("KB_FLUSH_MB", "768")), iters=int(os.environ.get('KB_ITERS', '20'))):
    """KB_COLD=1: every timed call starts with cold caches -- a KB_FLUSH_MB buffer (three times the 256-MB Infinity Cache) is
    rewritten before it, and `warm()` (optional) then re-touches what WOULD be warm inside the step (e.g. the operand the previous
    kernel has just written).  Times each call with its own event pair; the flush is outside the timed region.  For kernels whose
    in-step time differs from the hot-loop time (the residual-adding N = 512 products: 77 us in the step, 56 us in a hot loop)."""
    if "buf" not in _FLUSH or _FLUSH["buf"].numel() != flush_mb << 18:
        _FLUSH["buf"] = torch.empty(flush_mb << 18, dtype=torch.float32, device=DEV)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    total = 0.0
    for i in range(iters):
        _FLUSH["buf"].fill_(float(i))
        if warm is not None:
            warm()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total / iters * 1e-3


def rb(*shape, scale=1.0):
    if os.environ.get("KB_ZERO") == "1":      # zero-filled operands: the package draws less power and clocks higher (DVFS check)
        return torch.zeros(*shape, device=DEV, dtype=torch.bfloat16)
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16)


def ws(n):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=DEV)


def bench_gemm_nt(M, N, K, flags=0, tag="", variants=(("auto", 1),)):
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    bias, res = rb(N), rb(M, N)
    rs = torch.rand(M, device=DEV)
    for name, nt4 in variants:
        dh.set_option("nt4", nt4 if nt4 < 8 else 0)
        dh.set_option("nt8", 2 if nt4 == 8 else 0)
        dh.set_option("nt8p", 2 if nt4 == 9 else 0)            # 9: persistent 256x256 kernel
        dh.set_option("ntr", 2 if nt4 == 12 else 0)            # 12: full-row 160x512 tiles (N = 512 only)
        run = lambda: dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, flags, bias=bias, residual=res, relu_src=res, rowscale=rs)   # noqa: E731
        if os.environ.get("KB_COLD") == "1":
            # cold caches, A re-touched (the kernel before has just written it), weights re-touched (hot in every step): what is
            # cold is what the step leaves cold -- the residual / mask source and the output
            t = timeit_cold(run, warm=lambda: (A.add_(0), Bt.add_(0)))
            name = name + "/cold"
        else:
            t = timeit(run)
        print(f"gemm_nt{tag} M={M} N={N} K={K} flags={flags} {name:8s}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    dh.set_option("nt4", 1)
    dh.set_option("nt8", 1)
    dh.set_option("nt8p", 1)
    dh.set_option("ntr", 1)


def bench_splitk(M, N, K, ns, nt8):
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    rs = torch.rand(M, device=DEV)
    w = ws(dh.gemm_nt_splitk_workspace_bytes(M, N, ns))
    dh.set_option("nt8", nt8)
    t = timeit(lambda: dh.gemm_nt_splitk(A, K, Bt, K, C, M, N, K, ns, w, rowscale=rs), iters=10)
    print(f"gemm_nt_splitk M={M} N={N} K={K} nsplit={ns} nt8={nt8}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    dh.set_option("nt8", 1)


def bench_gemm_tn(M, I, J, weighted=False):
    for tn8 in (0, 2, 0, 2):    # 128x128 | 256x256 tiles (repeated: drift check)
        dh.set_option("tn8", tn8)
        _bench_gemm_tn(M, I, J, weighted, tn8)
    dh.set_option("tn8", 0)


def _bench_gemm_tn(M, I, J, weighted, tn8):
    X, dY = rb(M, I), rb(M, J)
    dW = torch.empty(I, J, dtype=torch.float32, device=DEV)
    db = torch.empty(J, dtype=torch.float32, device=DEV)
    bw = rb(M) if weighted else None
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    t = timeit(lambda: dh.gemm_tn(X, I, dY, J, dW, M, I, J, w, dbias=db, bias_weights=bw))
    print(f"gemm_tn M={M} I={I} J={J} weighted_bias={int(weighted)} tn8={tn8}: {t*1e6:9.1f} us  {2*M*I*J/t/1e12:8.1f} TF/s", flush=True)


def bench_attention(B, H, S):
    d = H * 128
    qkv = rb(B * S, 3 * d, scale=0.3)
    o = torch.empty(B * S, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=DEV)
    d_o = rb(B * S, d)
    delta = torch.empty(3, B, H, S, dtype=torch.float32, device=DEV)
    dqkv = torch.empty(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
    fl = 4.0 * B * H * S * S * 128   # dense-equivalent forward flops
    t = timeit(lambda: dh.attention_fwd(qkv, o, lse, B, H, S))
    print(f"attention_fwd B={B} H={H} S={S}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s dense-equivalent ({fl/2/t/1e12:.1f} executed)", flush=True)
    t = timeit(lambda: dh.attention_bwd(qkv, o, d_o, lse, delta, dqkv, B, H, S))
    print(f"attention_bwd B={B} H={H} S={S}: {t*1e6:9.1f} us  {2.5*fl/t/1e12:7.1f} TF/s dense-equivalent ({2.5*fl/2/t/1e12:.1f} executed)", flush=True)


def bench_head(M, K, V):
    Vp = (V + 127) // 128 * 128
    X, Wt, W = rb(M, K), rb(Vp, K, scale=0.02), rb(K, Vp, scale=0.02)
    bias = torch.zeros(Vp, device=DEV)
    bias[V:] = -30000.0
    bias = bias.to(torch.bfloat16)
    labels = torch.randint(0, V, (M,), device=DEV, dtype=torch.int32)
    zl = torch.empty(M, dtype=torch.float32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    nparts = dh.gemm_nt_softmax_partials(Vp)
    part = torch.empty(nparts, M, dtype=torch.float32, device=DEV)
    E = torch.empty(M, Vp, dtype=torch.bfloat16, device=DEV)
    loss = torch.empty(M, dtype=torch.float32, device=DEV)
    rsc = torch.empty(M, dtype=torch.float32, device=DEV)
    rsb = torch.empty(M, dtype=torch.bfloat16, device=DEV)
    Xs = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    dW = torch.empty(K, Vp, dtype=torch.float32, device=DEV)
    db = torch.empty(Vp, dtype=torch.float32, device=DEV)
    dX = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    w = ws(dh.gemm_tn_workspace_bytes(M, K, Vp))
    fl = 2.0 * M * K * V
    rows = [
        ("label_logit", lambda: dh.label_logit(X, K, Wt, K, bias, labels, zl, flag, M, K, V), None),
        ("gemm_nt (logits, bias)", lambda: dh.gemm_nt(X, K, Wt, K, E, Vp, M, Vp, K, dh.GEMM_BIAS, bias=bias), fl),
        ("cross_entropy (old path)", lambda: dh.cross_entropy(E, Vp, labels, loss, None, M, V, 1.0 / M), None),
        ("gemm_nt_softmax (label shift)", lambda: dh.gemm_nt_softmax(X, K, Wt, K, bias, zl, E, Vp, part, M, Vp, K), fl),
        ("gemm_nt_softmax (no shift)", lambda: dh.gemm_nt_softmax(X, K, Wt, K, bias, None, E, Vp, part, M, Vp, K), fl),
        ("softmax_finish", lambda: dh.softmax_finish(part, nparts, zl, None, labels, X, K, Wt, K, bias, E, Vp, Vp, loss, rsc, rsb, Xs, flag, M, K, V, 1.0 / M), None),
        ("head wgrad (weighted bias)", lambda: dh.gemm_tn(Xs, K, E, Vp, dW, M, K, Vp, w, dbias=db, bias_weights=rsb), fl),
        ("head dgrad (rowscale)", lambda: dh.gemm_nt(E, Vp, W, Vp, dX, K, M, K, Vp, dh.GEMM_ROWSCALE, rowscale=rsc), fl),
    ]
    for name, fn, f in rows:
        t = timeit(fn, iters=10)
        print(f"head M={M} K={K} V={V} {name:28s}: {t*1e6:9.1f} us" + (f"  {f/t/1e12:8.1f} TF/s" if f else ""), flush=True)


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"nt", "tn", "attn", "head"}
    for kv in os.environ.get("KB_OPTIONS", "").split(","):   # e.g. KB_OPTIONS=nt4_stages=2,tn8=0
        if "=" in kv:
            dh.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    M, d = 40960, 512
    if "nt" in what:
        both = (("nt2", 0), ("nt4", 2))
        bench_gemm_nt(M, 3 * d, d, 0, " qkv", both)
        bench_gemm_nt(M, d, d, 5, " attn-out", both)
        bench_gemm_nt(M, 4 * d, d, 3, " ffn1", both)
        bench_gemm_nt(M, d, 4 * d, 5, " ffn2", both)
        bench_gemm_nt(M, 4 * d, d, 8, " ffn2-dgrad", both)
        bench_gemm_nt(M, d, 4 * d, 0, " ffn1-dgrad", both)
        bench_gemm_nt(M, d, 3 * d, 0, " qkv-dgrad", both)
    if "big" in what:
        three = (("nt2", 0), ("nt4", 2), ("nt8", 8))
        bench_gemm_nt(M, 2048, 2048, 0, " big", three)
        bench_gemm_nt(M, 4096, 4096, 0, " big", three)
        bench_gemm_nt(M, 50816, 512, 1, " logits", three)
        bench_gemm_nt(M, 2048, 512, 3, " ffn1", three)
        bench_gemm_nt(M, 512, 2048, 5, " ffn2", three)
        bench_gemm_nt(M, 512, 50816, 32, " head-dgrad", (("nt2", 0), ("nt8", 8)))
        for ns, nt8 in ((2, 0), (4, 0), (4, 2), (8, 2)):
            bench_splitk(M, 512, 50816, ns, nt8)
    if "k512" in what:     # the K = 512 products on every NT tile + the fused softmax head on the 256x128 / 256x256 tiles
        three = (("nt2", 0), ("nt4", 2), ("nt8", 8), ("nt8p", 9))
        bench_gemm_nt(M, 3 * d, d, 0, " qkv", three)
        bench_gemm_nt(M, 4 * d, d, 3, " ffn1", three)
        bench_gemm_nt(M, 4 * d, d, 8, " ffn2-dgrad", three)
        Vp = 50816
        X, Wt = rb(M, d), rb(Vp, d, scale=0.02)
        bias = torch.zeros(Vp, device=DEV, dtype=torch.bfloat16)
        part = torch.empty(dh.gemm_nt_softmax_partials(Vp), M, dtype=torch.float32, device=DEV)
        E = torch.empty(M, Vp, dtype=torch.bfloat16, device=DEV)
        for name, nt4, nt8, nt8p in (("nt4", 2, 0, 0), ("nt8", 0, 2, 0), ("nt8p", 0, 0, 2), ("nt4", 2, 0, 0), ("nt8", 0, 2, 0), ("nt8p", 0, 0, 2)):
            dh.set_option("nt4", nt4); dh.set_option("nt8", nt8); dh.set_option("nt8p", nt8p)
            t = timeit(lambda: dh.gemm_nt_softmax(X, d, Wt, d, bias, None, E, Vp, part, M, Vp, d), iters=10)
            print(f"head gemm_nt_softmax (no shift) {name}: {t*1e6:9.1f} us  {2.0*M*d*50771/t/1e12:8.1f} TF/s", flush=True)
        dh.set_option("nt4", 1); dh.set_option("nt8", 1); dh.set_option("nt8p", 1)
    if "pmchead" in what:    # the fused-softmax vocabulary GEMM on the 256x128 and the persistent 256x256 kernels (tools/pmc_kernel.sh)
        Vp = 50816
        X, Wt = rb(M, d), rb(Vp, d, scale=0.02)
        bias = torch.zeros(Vp, device=DEV, dtype=torch.bfloat16)
        part = torch.empty(dh.gemm_nt_softmax_partials(Vp), M, dtype=torch.float32, device=DEV)
        E = torch.empty(M, Vp, dtype=torch.bfloat16, device=DEV)
        for name, nt4, nt8p in (("nt4", 2, 0), ("nt8p", 0, 2)):
            dh.set_option("nt4", nt4); dh.set_option("nt8p", nt8p)
            t = timeit(lambda: dh.gemm_nt_softmax(X, d, Wt, d, bias, None, E, Vp, part, M, Vp, d), iters=3)
            print(f"head gemm_nt_softmax (no shift) {name}: {t*1e6:9.1f} us", flush=True)
        dh.set_option("nt4", 1); dh.set_option("nt8p", 1)
    if "n512" in what:     # the products with N = 512 outputs: 128x128 tiles vs full-row 160x512 tiles
        two = (("nt2", 0), ("ntr", 12), ("nt2", 0), ("ntr", 12))
        bench_gemm_nt(M, d, d, 5, " attn-out", two)
        bench_gemm_nt(M, d, 4 * d, 5, " ffn2", two)
        bench_gemm_nt(M, d, 4 * d, 0, " ffn1-dgrad", two)
        bench_gemm_nt(M, d, 3 * d, 0, " qkv-dgrad", two)
        bench_gemm_nt(M, d, d, 0, " attn-out-dgrad", two)
    if "ln512" in what:    # product + bias + residual -> LayerNorm: two kernels vs the fused full-row form
        for K, tag in ((d, "attn-out"), (4 * d, "ffn2")):
            A, Bt = rb(M, K), rb(d, K, scale=0.05)
            bias, res, gam, bet = rb(d), rb(M, d), rb(d), rb(d)
            C, Y = torch.empty(M, d, dtype=torch.bfloat16, device=DEV), torch.empty(M, d, dtype=torch.bfloat16, device=DEV)
            mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
            for rep in range(2):
                t1 = timeit(lambda: dh.gemm_nt(A, K, Bt, K, C, d, M, d, K, 5, bias=bias, residual=res))
                t2 = timeit(lambda: dh.layernorm_fwd(C, gam, bet, Y, mean, rstd, M, d))
                t3 = timeit(lambda: dh.gemm_nt_ln(A, K, Bt, K, C, d, M, d, K, gam, bet, Y, d, mean, rstd, bias=bias, residual=res))
                print(f"ln512 {tag} K={K}: gemm_nt {t1*1e6:7.1f} us + layernorm_fwd {t2*1e6:6.1f} us = {(t1+t2)*1e6:7.1f} us   fused gemm_nt_ln {t3*1e6:7.1f} us", flush=True)
    if "tn" in what:
        for I, J in ((4 * d, d), (d, 4 * d), (d, d), (d, 3 * d), (d, 50816)):
            bench_gemm_tn(M, I, J, weighted=(J == 50816))
    if "attn" in what:
        bench_attention(32, 4, 1280)
    if "head" in what:
        bench_head(M, d, 50771)
