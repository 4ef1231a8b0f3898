"""[r06] randomised shapes for the weight-gradient paths on 128 x 256 tiles: dmi_gemm_tn on the gang stream-K (ragged I / J / M, leading
dimensions larger than the widths, weighted and plain bias sums, reserved CUs) against fp32 and against the 128 x 128 kernel; dmi_gemm_tn_group
with random problem sets whose plan is the wide one against the single launches (one-off stress run, not part of the suite)."""
import os, sys, random, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "dalle-mtf_amd")]
import torch
import dalle_hip as dh
DEV = "cuda"
rng = random.Random(int(os.environ.get("STRESS_SEED", "11")))
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16).to(DEV)
def ws(n): return torch.empty(max(int(n), 256) + 256, dtype=torch.uint8, device=DEV)
bad = 0; n_sk = 0
for it in range(40):
    reserve = rng.choice([0, 0, 16, 40])
    wti = rng.choice([1, 2, 3, 4, 5, 8])
    I = 128 * (wti - 1) + 8 * rng.randint(1, 16)
    gangs = (2 * ((256 - reserve) & ~7) // wti) & ~7
    wtj = gangs + rng.randint(0, 40)
    J = 256 * (wtj - 1) + 8 * rng.randint(1, 32)
    M = rng.randint(1, 700)
    ldx, ldy = I + 8 * rng.randint(0, 3), J + 8 * rng.randint(0, 3)
    weighted = rng.random() < 0.5 and M % 2 == 0    # (bias_weights needs an even M: found by this script, now an argument check)
    X, Y = rnd(M, ldx, seed=it), rnd(M, ldy, seed=it + 100)
    bw = (torch.rand(M, generator=torch.Generator().manual_seed(it)) + 0.5).to(torch.bfloat16).to(DEV) if weighted else None
    w = ws(dh.gemm_tn_workspace_bytes(M, I, J))
    dh.set_option("reserve_cus", reserve)
    outs = []
    for wide in (1, 0):
        dh.set_option("tn_wide", wide); dh.set_option("tn_tail", wide)
        dW = torch.full((I, J), float("nan"), device=DEV); db = torch.full((J,), float("nan"), device=DEV)
        dh.gemm_tn(X, ldx, Y, ldy, dW, M, I, J, w, dbias=db, bias_weights=bw)
        outs.append((dW, db))
    dh.set_option("tn_wide", 1); dh.set_option("tn_tail", 1); dh.set_option("reserve_cus", 0)
    (d1, b1), (d0, b0) = outs
    ref = X[:, :I].float().t() @ Y[:, :J].float()
    rb = ((bw.float()[:, None] if weighted else 1.0) * Y[:, :J].float()).sum(0)
    mag = X[:, :I].float().abs().t() @ Y[:, :J].float().abs() + 1e-6
    e = float(((d1 - ref).abs() / mag).max()); e01 = float(((d1 - d0).abs() / mag).max())
    eb = float((b1 - rb).abs().max() / (rb.abs().max() + 1e-6))
    ok = (not torch.isnan(d1).any()) and (not torch.isnan(b1).any()) and e < 1e-5 and e01 < 3e-7 and eb < 1e-5
    n_sk += int(not torch.equal(d1, d0))
    bad += not ok
    if not ok: print("gemm_tn mismatch", M, I, J, ldx, ldy, reserve, e, e01, eb, flush=True)
print("gemm_tn gang stream-K: failures", bad, "| shapes whose result differs from the 128x128 kernel in some cut stripe:", n_sk, "of 40", flush=True)
bad2 = 0; n_wide = 0
tries = 0
while n_wide < 25 and tries < 4000:
    tries += 1
    n = rng.randint(2, 4)
    shapes = [(128 * rng.randint(1, 16) - 8 * rng.randint(0, 5), 256 * rng.randint(1, 8) - 8 * rng.randint(0, 9)) for _ in range(n)]
    M = rng.randint(700, 9000)
    if dh.gemm_tn_group_plan(shapes, M) == 0: continue
    n_wide += 1
    probs = []
    for k, (I, J) in enumerate(shapes):
        q = dict(X=rnd(M, I, seed=tries + k), ldx=I, dY=rnd(M, J, seed=tries + k + 50), ldy=J, dW=torch.full((I, J), float("nan"), device=DEV), I=I, J=J,
                 ws=ws(dh.gemm_tn_workspace_bytes(M, I, J)))
        if rng.random() < 0.6: q["dbias"] = torch.full((J,), float("nan"), device=DEV)
        probs.append(q)
    dh.gemm_tn_group(probs, M)
    for q in probs:
        ref = q["X"].float().t() @ q["dY"].float()
        mag = q["X"].float().abs().t() @ q["dY"].float().abs() + 1e-6
        e = float(((q["dW"] - ref).abs() / mag).max())
        eb = 0.0
        if "dbias" in q:
            rb = q["dY"].float().sum(0); eb = float((q["dbias"] - rb).abs().max() / (q["dY"].float().abs().sum(0).max() + 1e-6))
        if not (e < 1e-5 and eb < 1e-5):
            bad2 += 1; print("group mismatch", M, shapes, e, eb, flush=True)
print("gemm_tn_group on the wide tile: problem sets", n_wide, "(of", tries, "drawn) failures", bad2, flush=True)
torch.cuda.synchronize()
