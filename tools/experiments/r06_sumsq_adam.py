"""[r06] gradient-norm sweep direction vs the Adam kernel that follows it: the pair timed at the dalle_example parameter count, with ~300 MB of
unrelated traffic in front (as the backward's last kernels leave the Infinity Cache), options alternated in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch, dalle_hip as dh
n = 71_601_747 // 4 * 4
p, g, m, v = (torch.randn(n, device="cuda") * 0.01 for _ in range(4))
v.abs_()
pb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
ws = torch.empty(dh.sumsq_workspace_bytes(n) + 256, dtype=torch.uint8, device="cuda")
gn = torch.zeros(1, device="cuda")
junk = torch.empty(80 << 20, dtype=torch.float32, device="cuda")


def pair():
    junk.mul_(1.0001)                       # 640 MB of traffic: what precedes the clip in the step is not the gradient buffer's head
    g[-26_000_000:].mul_(1.0)               # ... and the embedding gradients are the last thing written
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    dh.sumsq(g, n, gn, ws)
    e1.record()
    dh.adam_step(p, g, m, v, pb, n, gn, 1.0, 1e-4, 0.9, 0.999, 1e-6, 0.0)
    e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3


for _ in range(5):
    pair()
for rep in range(4):
    out = []
    for rev in (0, 1):
        dh.set_option("sumsq_rev", rev)
        ts = [pair() for _ in range(10)]
        out.append(f"rev={rev}: sumsq {sum(t[0] for t in ts)/10:6.1f} us  adam {sum(t[1] for t in ts)/10:6.1f} us  pair {sum(t[0]+t[1] for t in ts)/10:6.1f} us")
    print("   ".join(out), flush=True)
