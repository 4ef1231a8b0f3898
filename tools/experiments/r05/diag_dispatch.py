"""where do the production dispatch and the all-128x128 dispatch first differ at B = 32?  (forward activations, layer by layer)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dalle-mtf_amd"), os.path.join(ROOT, "tests")]
import torch
import dalle_hip as dh
from oracle import dalle_oracle as do
from test_headline_parity_gpu import _headline_engine

B = int(os.environ.get("DIAG_B", "32"))
tokens = torch.from_numpy(do.assemble_tokens(do.synthetic_captions(B, 256, 50258, seed=1), do.synthetic_image_tokens(B, 1024, 512, seed=2), 50258)).cuda()
names = ("ntr", "nt8p", "nt8", "nt4")
saved = {n: dh.get_option(n) for n in names}
snaps = []
for mode in os.environ.get("DIAG_MODES", "prod,plain,prod").split(","):
    for n in names:
        dh.set_option(n, saved[n] if mode == "prod" else (3 if n == "nt8p" else 0))
    eng = _headline_engine(B)
    eng.forward(tokens, need_grad=True)
    torch.cuda.synchronize()
    snap = {}
    for l in range(eng.L):
        for nm in ("xn1", "qkv", "o", "x1", "xn2", "h"):
            snap[f"{l}/{nm}"] = getattr(eng, nm)[l].clone()
        snap[f"{l}/X"] = eng.X[l].clone()
    snap["X_L"] = eng.X[eng.L].clone(); snap["xnf"] = eng.xnf.clone(); snap["zl"] = eng.zl.clone()
    snap["rowsum_part"] = eng.rowsum_part.clone(); snap["loss_rows"] = eng.loss_rows.clone(); snap["rowscale"] = eng.rowscale.clone()
    eng.backward(); torch.cuda.synchronize()
    snap["g"] = eng.g.clone()
    snaps.append((mode, snap))
    del eng; torch.cuda.empty_cache()
for i in range(1, len(snaps)):
    print(f"--- {snaps[0][0]}[0] vs {snaps[i][0]}[{i}]")
    for k in snaps[0][1]:
        a, b = snaps[0][1][k], snaps[i][1][k]
        nd = int((a != b).sum())
        if nd:
            print(f"{k:16s} differing {nd:10d} of {a.numel():12d}  max abs {float((a.float() - b.float()).abs().max()):.3e}")
