"""[r06] the gang stream-K weight gradient on 128 x 256 tiles (option tn_wide) against the 128 x 128 kernel: results on ragged shapes,
then timings alternated in one process.  (The logs profiles/r06_tn_wide.log were taken with the experiment's whole-launch forms as well:
w1 = 128 x 256 tiles, w2 = 256 x 128 tiles, w3 = the gang stream-K that became the product path.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb, ws
DEV = "cuda"

def run(M, I, J, ldx, ldy, weighted, wide, bias=True):
    g = torch.Generator().manual_seed(M + I + J)
    X = (torch.randn(M, ldx, generator=g)).to(torch.bfloat16).to(DEV)
    Y = (torch.randn(M, ldy, generator=g)).to(torch.bfloat16).to(DEV)
    bw = (torch.rand(M, generator=g) + 0.5).to(torch.bfloat16).to(DEV) if weighted else None
    dW = torch.full((I, J), float("nan"), device=DEV); db = torch.full((J,), float("nan"), device=DEV) if bias else None
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    dh.set_option("tn_wide", wide)
    dh.gemm_tn(X, ldx, Y, ldy, dW, M, I, J, w, dbias=db, bias_weights=bw)
    dh.set_option("tn_wide", 1)
    torch.cuda.synchronize()
    return X, Y, bw, dW, db

bad = 0
SKIP = os.environ.get("SKIP_CHECK") == "1"
for (M, I, J, ldx, ldy, weighted) in [] if SKIP else [(4096, 512, 1024, 512, 1024, False), (1000, 264, 520, 272, 528, True), (97, 128, 256, 128, 256, False),
                                      (5000, 8, 24, 8, 24, True), (33, 520, 136, 520, 136, False), (2048, 512, 2816, 512, 2816, True),
                                      (200, 512, 33288, 512, 33288, True), (33, 512, 40000, 512, 40000, False), (1000, 256, 66000, 256, 66000, True)]:
    X, Y, bw, d0, b0 = run(M, I, J, ldx, ldy, weighted, 0)
    ref = X[:, :I].float().t() @ Y[:, :J].float()
    rb_ = ((bw.float()[:, None] if weighted else 1.0) * Y[:, :J].float()).sum(0)
    for wide in (1,):
        _, _, _, d1, b1 = run(M, I, J, ldx, ldy, weighted, wide)
        e = float((d1 - ref).abs().max()) / (1e-6 + float(ref.abs().max()))
        e0 = float((d0 - ref).abs().max()) / (1e-6 + float(ref.abs().max()))
        eb = float((b1 - rb_).abs().max()) / (1e-6 + float(rb_.abs().max()))
        ok = e < 2e-3 and eb < 2e-3 and not torch.isnan(d1).any() and not torch.isnan(b1).any()
        bad += not ok
        if wide == 1:   # deterministic: a second run gives the same bits
            _, _, _, d2, b2 = run(M, I, J, ldx, ldy, weighted, wide)
            ok = ok and torch.equal(d1, d2) and torch.equal(b1, b2)
        print(f"M={M} I={I} J={J} weighted={int(weighted)} wide={wide}: err {e:.2e} (128x128: {e0:.2e}) bias {eb:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("failures", bad)

for (M, I, J, weighted) in [(40960, 512, 65536, True), (40960, 512, 49152, True), (40960, 512, 50816, True)]:
    X, Y = rb(M, I), rb(M, J)
    dW = torch.empty(I, J, device=DEV); db = torch.empty(J, device=DEV)
    bw = rb(M) if weighted else None
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    out = []
    for rep in range(2):
        for wide in (0, 1):
            dh.set_option("tn_wide", wide)
            t = timeit(lambda: dh.gemm_tn(X, I, Y, J, dW, M, I, J, w, dbias=db, bias_weights=bw))
            out.append(f"w{wide} {t*1e6:7.1f} us {2*M*I*J/t/1e12:6.0f} TF/s")
    dh.set_option("tn_wide", 1)
    print(f"gemm_tn M={M} I={I} J={J}: " + " | ".join(out), flush=True)
