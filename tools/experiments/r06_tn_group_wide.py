"""[r06] the two FFN weight gradients of a block: two launches on 128 x 128 tiles (production until now) vs one grouped launch on
128 x 128 tiles vs one grouped launch on 128 x 256 tiles; slab reduces included (one batched launch in every arm)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
from kbench import timeit, rb, ws
DEV = "cuda"
M, d = 40960, 512
h, dx, xn2, dhh = rb(M, 4 * d), rb(M, d), rb(M, d), rb(M, 4 * d)
dW2 = torch.empty(4 * d, d, device=DEV); db2 = torch.empty(d, device=DEV)
dW1 = torch.empty(d, 4 * d, device=DEV); db1 = torch.empty(4 * d, device=DEV)
w2, w1 = ws(dh.gemm_tn_workspace_bytes(M, 4 * d, d)), ws(dh.gemm_tn_workspace_bytes(M, d, 4 * d))
deferred = dh.DeferredReduces()

def two():
    dh.gemm_tn(h, 4 * d, dx, d, dW2, M, 4 * d, d, w2, dbias=db2, deferred=deferred)
    dh.gemm_tn(xn2, d, dhh, 4 * d, dW1, M, d, 4 * d, w1, dbias=db1, deferred=deferred)
    deferred.run()

def group():
    dh.gemm_tn_group([dict(X=h, ldx=4 * d, dY=dx, ldy=d, dW=dW2, I=4 * d, J=d, ws=w2, dbias=db2),
                      dict(X=xn2, ldx=d, dY=dhh, ldy=4 * d, dW=dW1, I=d, J=4 * d, ws=w1, dbias=db1)], M, deferred=deferred)
    deferred.run()

two(); torch.cuda.synchronize()
r2, r1, rb2, rb1 = dW2.clone(), dW1.clone(), db2.clone(), db1.clone()
for wide in (0, 1):
    dh.set_option("tn_wide", wide)
    dW2.fill_(float("nan")); dW1.fill_(float("nan")); db1.fill_(float("nan")); db2.fill_(float("nan"))
    group(); torch.cuda.synchronize()
    e = max(float((dW2 - r2).abs().max() / r2.abs().max()), float((dW1 - r1).abs().max() / r1.abs().max()),
            float((db2 - rb2).abs().max() / rb2.abs().max()), float((db1 - rb1).abs().max() / rb1.abs().max()))
    print(f"group wide={wide}: max rel diff to the two launches {e:.2e}", flush=True)
for rep in range(3):
    out = []
    dh.set_option("tn_wide", 0); out.append(f"two launches {timeit(two)*1e6:7.1f} us")
    out.append(f"group 128x128 {timeit(group)*1e6:7.1f} us")
    dh.set_option("tn_wide", 1); out.append(f"group 128x256 {timeit(group)*1e6:7.1f} us")
    print(" | ".join(out), flush=True)

# all four gradients of a block: FFN pair (wide group) + attention pair (128 x 128 group) vs ONE wide group of four
o, dxb, xn1, dqkv = rb(M, d), rb(M, d), rb(M, d), rb(M, 3 * d)
dWo = torch.empty(d, d, device=DEV); dbo = torch.empty(d, device=DEV); dWq = torch.empty(d, 3 * d, device=DEV)
wo, wq = ws(dh.gemm_tn_workspace_bytes(M, d, d)), ws(dh.gemm_tn_workspace_bytes(M, d, 3 * d))
P_ffn = [dict(X=h, ldx=4 * d, dY=dx, ldy=d, dW=dW2, I=4 * d, J=d, ws=w2, dbias=db2),
         dict(X=xn2, ldx=d, dY=dhh, ldy=4 * d, dW=dW1, I=d, J=4 * d, ws=w1, dbias=db1)]
P_att = [dict(X=o, ldx=d, dY=dxb, ldy=d, dW=dWo, I=d, J=d, ws=wo, dbias=dbo),
         dict(X=xn1, ldx=d, dY=dqkv, ldy=3 * d, dW=dWq, I=d, J=3 * d, ws=wq)]

def pairs():
    dh.gemm_tn_group(P_ffn, M, deferred=deferred)
    dh.gemm_tn_group(P_att, M, deferred=deferred)
    deferred.run()

def four():
    dh.gemm_tn_group(P_ffn + P_att, M, deferred=deferred)
    deferred.run()

dh.set_option("tn_wide", 1)
pairs(); torch.cuda.synchronize()
ref = [t.clone() for t in (dW2, dW1, dWo, dWq, db2, db1, dbo)]
four(); torch.cuda.synchronize()
e = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip((dW2, dW1, dWo, dWq, db2, db1, dbo), ref))
print(f"group of four vs the two pairs: max rel diff {e:.2e}", flush=True)
for rep in range(3):
    print(f"two pairs {timeit(pairs)*1e6:7.1f} us | group of four {timeit(four)*1e6:7.1f} us", flush=True)
