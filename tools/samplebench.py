"""Generation speed at the dalle_example shape: KV-cached incremental decode vs one full forward per token.
usage: python tools/samplebench.py [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
from src.dalle_mtf.engine import DalleEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
eng = DalleEngine(512, 6, 4, 50258, 512, 256, 1024, batch_size=B, hparams=dict(lr=1e-3, train_steps=10))
eng.init_params(seed=1)
text = torch.randint(0, 50257, (B, 256), dtype=torch.int32, device="cuda")
eng.sample_image_tokens(text[:, :], temperature=1.0, top_k=32, seed=0)      # warm-up (allocations, clocks, graph capture)
out = {}
eng.sample_image_tokens(text[:, :], temperature=1.0, top_k=32, seed=0, fused_sampling=False)
for name, kw in (("kv-cached, host-launched", dict(decode_graph=False)), ("kv-cached, HIP graph", dict(fused_sampling=False)),
                 ("kv-cached, HIP graph, greedy", dict(fused_sampling=False, temperature=0.0)),
                 ("kv-cached, HIP graph incl. the draw", dict()), ("kv-cached, HIP graph incl. the draw, greedy", dict(temperature=0.0))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    args = dict(temperature=1.0, top_k=32, seed=1)
    args.update(kw)
    out[name] = eng.sample_image_tokens(text, **args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {B} x 1024 image tokens in {dt:.3f} s = {B * 1024 / dt:.0f} tokens/s ({dt / 1024 * 1e3:.3f} ms per position)")
print("graph == host-launched tokens:", bool(torch.equal(out["kv-cached, host-launched"], out["kv-cached, HIP graph"])),
      " fused greedy == torch greedy:", bool(torch.equal(out["kv-cached, HIP graph, greedy"], out["kv-cached, HIP graph incl. the draw, greedy"])))
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in range(256, 256 + 200):
    eng.decode_step(text[:, 0].contiguous(), p)
torch.cuda.synchronize()
print(f"decode_step alone (graph replay + 3 host-side scalar copies): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per position")
toks2 = torch.full((B, eng.S), 50258, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(32):
    eng.forward(toks2, need_grad=False)
torch.cuda.synchronize()
df = (time.perf_counter() - t0) / 32
print(f"full forward per position: {df * 1e3:.3f} ms -> {1024 * df:.3f} s per {B} x 1024 tokens ({dt and 1024 * df / dt:.1f}x the cached path)")
