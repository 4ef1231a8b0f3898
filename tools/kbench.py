"""Kernel micro-benchmarks on the GPU box (torch.cuda.Event timing on the launch stream, random data).
Prints one line per kernel/shape: time, achieved TFLOP/s or GB/s.  Used to steer optimisation; the
judged numbers come from bench.py + rocprofv3 summaries under profiles/."""
import os
import sys
import math

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
import dalle_hip as dh

DEV = "cuda"


def timeit(fn, iters=int(os.environ.get('KB_ITERS', '20')), warm=int(os.environ.get('KB_WARM', '3'))):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rb(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(torch.bfloat16)


def ws(n):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=DEV)


def bench_gemm_nt(M, N, K, flags=0, tag=""):
    A, Bt = rb(M, K), rb(N, K, scale=0.05)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    bias, res = rb(N), rb(M, N)
    for name, opts in (("nt2", dict(nt2=1, glds=1, nt3=0, nt4=0)), ("nt4", dict(nt2=1, glds=1, nt3=0, nt4=2)), ("nt2 again", dict(nt2=1, glds=1, nt3=0, nt4=0))):
        for k, v in opts.items():
            dh.set_option(k, v)
        t = timeit(lambda: dh.gemm_nt(A, K, Bt, K, C, N, M, N, K, flags, bias=bias, residual=res, relu_src=res))
        print(f"gemm_nt{tag} M={M} N={N} K={K} flags={flags} {name:8s}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    dh.set_option("nt2", 1)
    dh.set_option("nt3", 0)
    dh.set_option("nt4", 1)
    dh.set_option("glds", 1)


def bench_gemm_tn(M, I, J):
    X, dY = rb(M, I), rb(M, J)
    dW = torch.empty(I, J, dtype=torch.float32, device=DEV)
    db = torch.empty(J, dtype=torch.float32, device=DEV)
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    for v in (0, 1, 0, 1):
        dh.set_option("tn_streamk", v)
        t = timeit(lambda: dh.gemm_tn(X, I, dY, J, dW, M, I, J, w, dbias=db))
        print(f"gemm_tn M={M} I={I} J={J} streamk={v}: {t*1e6:9.1f} us  {2*M*I*J/t/1e12:8.1f} TF/s", flush=True)
    dh.set_option("tn_streamk", 1)


def bench_attention(B, H, S):
    d = H * 128
    qkv = rb(B * S, 3 * d, scale=0.3)
    T = [torch.empty(B, H, 128, S, dtype=torch.bfloat16, device=DEV) for _ in range(4)]
    def tr():
        for i in range(3):
            dh.transpose_strided(qkv.data_ptr() + i * d * 2, T[i], B, H, S, 128, S * 3 * d, 128, 3 * d)
    tr()
    o = torch.empty(B * S, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=DEV)
    t = timeit(lambda: dh.attention_fwd(qkv, T[2], o, lse, B, H, S))
    fl = 4.0 * B * H * S * S * 128  # dense count (QK^T + PV), as SURVEY §8(d) counts it
    print(f"attn_fwd B={B} H={H} S={S}: {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TF/s dense-equivalent ({fl/2/t/1e12:.1f} causal)", flush=True)
    d_o = rb(B * S, d)
    dh.transpose_strided(d_o.data_ptr(), T[3], B, H, S, 128, S * d, 128, d)
    delta = torch.empty(3, B, H, S, dtype=torch.float32, device=DEV)
    dqkv = torch.empty(B * S, 3 * d, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: dh.attention_bwd(qkv, T[0], T[1], o, d_o, T[3], lse, delta, dqkv, B, H, S))
    print(f"attn_bwd B={B} H={H} S={S}: {t*1e6:9.1f} us  {2*fl/t/1e12:8.1f} TF/s dense-equivalent", flush=True)
    t = timeit(tr)
    print(f"3 head transposes: {t*1e6:9.1f} us  {3*2*B*S*d*2/t/1e9:8.1f} GB/s", flush=True)


def bench_misc(M, d, V, ld):
    x, g, b = rb(M, d), rb(d), rb(d)
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=DEV)
    rstd = torch.empty(M, dtype=torch.float32, device=DEV)
    t = timeit(lambda: dh.layernorm_fwd(x, g, b, y, mean, rstd, M, d))
    print(f"ln_fwd M={M} d={d}: {t*1e6:9.1f} us  {2*M*d*2/t/1e9:8.1f} GB/s", flush=True)
    dx = torch.empty_like(x)
    dg = torch.empty(d, dtype=torch.float32, device=DEV)
    db = torch.empty(d, dtype=torch.float32, device=DEV)
    w = ws(dh.layernorm_bwd_workspace_bytes(M, d))
    t = timeit(lambda: dh.layernorm_bwd(y, x, g, mean, rstd, x, dx, dg, db, w, M, d))
    print(f"ln_bwd M={M} d={d}: {t*1e6:9.1f} us  {4*M*d*2/t/1e9:8.1f} GB/s", flush=True)
    h = rb(M, 4 * d)
    out = torch.empty(4 * d, dtype=torch.float32, device=DEV)
    w2 = ws(dh.colsum_workspace_bytes(M, 4 * d))
    t = timeit(lambda: dh.colsum(h, 4 * d, out, M, 4 * d, w2))
    print(f"colsum M={M} N={4*d}: {t*1e6:9.1f} us  {M*4*d*2/t/1e9:8.1f} GB/s", flush=True)
    z = rb(M, ld, scale=2.0)
    labels = torch.randint(0, V, (M,), device=DEV, dtype=torch.int32)
    lr = torch.empty(M, dtype=torch.float32, device=DEV)
    t = timeit(lambda: dh.cross_entropy(z, ld, labels, lr, None, M, V, 1e-4), iters=5)
    print(f"cross_entropy M={M} V={V}: {t*1e6:9.1f} us  {2*M*ld*2/t/1e9:8.1f} GB/s (read+write)", flush=True)
    n = 71_601_747 // 4 * 4
    p, gg, m, v = (torch.randn(n, device=DEV) * 0.01 for _ in range(4))
    v.abs_()
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    nrm = torch.ones(1, dtype=torch.float32, device=DEV)
    t = timeit(lambda: dh.adam_step(p, gg, m, v, pb, n, nrm, 1.0, 1e-3, 0.9, 0.999, 1e-6, 0.0, 1.0), iters=5)
    print(f"adam n={n}: {t*1e6:9.1f} us  {n*30/t/1e9:8.1f} GB/s", flush=True)
    w3 = ws(dh.sumsq_workspace_bytes(n))
    t = timeit(lambda: dh.sumsq(gg, n, nrm, w3), iters=5)
    print(f"sumsq n={n}: {t*1e6:9.1f} us  {n*4/t/1e9:8.1f} GB/s", flush=True)
    tok = torch.randint(0, 50771, (M,), device=DEV, dtype=torch.int32)
    tok[::3] = 50257
    dwte = torch.zeros(50771, d, dtype=torch.float32, device=DEV)
    dwpe = torch.zeros(1280, d, dtype=torch.float32, device=DEV)
    t = timeit(lambda: dh.embed_bwd(tok, x, dwte, dwpe, M // 1280, 1280, d, 50771), iters=5)
    print(f"embed_bwd M={M}: {t*1e6:9.1f} us", flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    M = 32 * 1280
    which = sys.argv[1:] or ["gemm", "tn", "attn", "misc"]
    if "pmc" in which:   # short list for counter collection
        dh.set_option("nt3", 0)
        for (mm, nn, kk_, fl) in ((8192, 8192, 8192, 0), (M, 50816, 512, 1), (M, 1536, 512, 0), (M, 512, 2048, 5)):
            A, Bt = rb(mm, kk_), rb(nn, kk_, scale=0.05)
            C = torch.empty(mm, nn, dtype=torch.bfloat16, device=DEV)
            bias, res = rb(nn), rb(mm, nn)
            for _ in range(3):
                dh.gemm_nt(A, kk_, Bt, kk_, C, nn, mm, nn, kk_, fl, bias=bias, residual=res)
            torch.cuda.synchronize()
        bench_gemm_tn(M, 2048, 512)
    if "gemm" in which:
        bench_gemm_nt(M, 1536, 512, tag="[qkv]")
        bench_gemm_nt(M, 512, 512, 5, tag="[outproj]")
        bench_gemm_nt(M, 2048, 512, 3, tag="[ffn1]")
        bench_gemm_nt(M, 512, 2048, 5, tag="[ffn2]")
        bench_gemm_nt(M, 50816, 512, 1, tag="[logits]")
        bench_gemm_nt(M, 512, 50816, 0, tag="[dlogits]")
        bench_gemm_nt(8192, 8192, 8192, 0, tag="[square]")
    if "tn" in which:
        bench_gemm_tn(M, 512, 1536)
        bench_gemm_tn(M, 512, 512)
        bench_gemm_tn(M, 2048, 512)
        bench_gemm_tn(M, 512, 50816)
    if "attn" in which:
        bench_attention(32, 4, 1280)
    if "misc" in which:
        bench_misc(M, 512, 50771, 50816)
