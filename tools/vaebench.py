"""Discrete-VAE train-step timing (SURVEY.md §8(d) C1 vae_example, C4 vae_coco at 16 images / GPU): forward with
Gumbel noise + backward + TF-Adam, synthetic images.  Prints images/s, image tokens/s and algorithmic TFLOP/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dalle-mtf_amd"))
import torch
from src.vae_tf import DiscreteVAE


def conv_flops(vae):
    """2*Hout*Wout*Cout*K*K*Cin per conv (transpose conv: input-sized), fwd, per image."""
    fl = 0
    for c in vae.convs:
        # reference channel counts (pad channels of the MI355X layout are not credited); conv: output-sized,
        # transpose conv: input-sized (H x W here is the input grid of every spec)
        hw = c.Ho * c.Wo if c.kind != "up" else c.H * c.W
        fl += 2 * hw * c.cout_ref * c.kk * c.cin_ref
    g = vae.grid
    fl += 2 * 2 * g * g * vae.n_hid * vae.num_tokens
    return fl


for name, B, steps in (("vae_example", 32, 20), ("vae_coco", 16, 5)):
    p = json.load(open(os.path.join(ROOT, "configs", name + ".json")))
    vae = DiscreteVAE(num_tokens=p["num_tokens"], dimensions=p["dataset"]["image_size"], convblocks=p["convblocks"],
                      dim=p.get("dim") or 512, hidden_dim=p.get("hidden_dim") or 64, input_channels=p["n_channels"],
                      use_bf16=bool(p.get("use_bf16")), batch_size=B, mode="train")
    vae.init_params()
    img = (torch.randint(0, 256, (B, vae.H, vae.W, 3), device="cuda").float() - 127.5) / 127.5

    def step():
        vae.forward(img, return_recon_loss=True, hard_gumbel=bool(p.get("train_gumbel_hard", True)), temperature=1.0)
        vae.backward()
        vae.optimizer_step(p["lr"])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    try:
        fl = 3 * conv_flops(vae) * B
    except Exception as e:  # the per-conv attribute names are an implementation detail
        fl = float("nan")
    print(json.dumps({"config": name, "per_gpu_batch": B, "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1),
                      "image_tokens_per_s": round(B * vae.grid ** 2 / dt), "train_tflops": round(fl / dt / 1e12, 2),
                      "mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    del vae
    torch.cuda.empty_cache()
