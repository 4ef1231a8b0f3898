"""Generates tests/golden/*.npz from the CPU oracle at fixed seeds (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself cannot be run here (TensorFlow 2.4 /
mesh-tensorflow 0.1.18 are not installable), so these vectors pin the ORACLE restatement: any drift of
oracle/*.py or of the HIP path against them is caught by tests/test_golden.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dalle_oracle as do  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

DALLE_SMALL = dict(n_embd=128, text_vocab_size=120, image_vocab_size=24, text_seq_len=24, image_seq_len=40,
                   n_layers=2, n_heads=1)
HP = dict(lr=1e-3, train_steps=1000, warmup_steps=2, gradient_clipping=1.0)
VAE_SMALL = dict(num_tokens=64, dimensions=16, convblocks=[[2, 16], [2, 64]])


def dalle_case():
    cfg = do.DalleConfig(**DALLE_SMALL)
    P = do.init_params(cfg, seed=77, perturb=0.05)
    text = do.synthetic_captions(2, cfg.text_seq_len, cfg.text_vocab_size, seed=5)
    img = do.synthetic_image_tokens(2, cfg.image_seq_len, cfg.image_vocab_size, seed=6)
    tokens = do.assemble_tokens(text, img, cfg.text_vocab_size)
    from collections import OrderedDict
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P.items())
    taps = {}
    loss, loss_batch, logits = do.forward(Pt, tokens, cfg, return_logits=True, taps=taps)
    loss_f, grads = do.loss_and_grads(P, tokens, cfg)
    out = dict(tokens=tokens, labels=do.shift_labels(tokens, cfg.eos_token_id), loss=np.float32(loss_f),
               loss_batch=loss_batch.detach().numpy(), logits=logits.detach().numpy(),
               hidden_final=taps["layer_1"].detach().numpy(), embed=taps["embed"].detach().numpy())
    for k in ("embedding/wte", "positional_embedding/wpe", "layer_0/attn/q", "layer_0/attn/o", "layer_1/mlp/mlp_linear_1/kernel",
              "layer_1/mlp/mlp_linear_2/bias", "layer_0/norm_1/g", "to_logits/linear_out/kernel", "to_logits/linear_out/bias"):
        out["grad:" + k] = grads[k]
    # one optimizer step (step index 1 so the warm-up lr is non-zero)
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v = {k: np.zeros_like(v) for k, v in P.items()}
    P2 = {k: a.copy() for k, a in P.items()}
    _, gnorm, lr = do.train_step(P2, m, v, tokens, cfg, 1, HP)
    out["gnorm"], out["lr"] = np.float32(gnorm), np.float32(lr)
    out["after:layer_0/attn/q"] = P2["layer_0/attn/q"]
    out["after:to_logits/linear_out/bias"] = P2["to_logits/linear_out/bias"]
    np.savez_compressed(os.path.join(HERE, "dalle_small.npz"), **out)
    print("dalle_small.npz: loss", loss_f, "gnorm", gnorm, "lr", lr)


def vae_case():
    cfg = vo.VaeConfig(**VAE_SMALL)
    P = vo.init_params(cfg, seed=11, bias_perturb=0.02)
    img = vo.synthetic_images(2, 16, seed=3)
    u = vo.synthetic_uniforms((2, cfg.grid, cfg.grid, cfg.num_tokens), seed=4)
    from collections import OrderedDict
    Pt = OrderedDict((k, torch.tensor(v)) for k, v in P.items())
    logits = vo.forward(Pt, torch.tensor(img), cfg, return_logits=True).numpy()
    loss_h, grads_h, out_h = vo.loss_and_grads(P, img, u, cfg, hard=True, temp=1.0)
    loss_s, grads_s, out_s = vo.loss_and_grads(P, img, u, cfg, hard=False, temp=0.7)
    res = dict(img=img, u=u, logits=logits, tokens=np.argmax(logits, -1).reshape(2, -1).astype(np.int32),
               loss_hard=np.float32(loss_h), recon_hard=out_h, loss_soft=np.float32(loss_s), recon_soft=out_s)
    for k in grads_h:
        res["grad_hard:" + k] = grads_h[k]
        res["grad_soft:" + k] = grads_s[k]
    np.savez_compressed(os.path.join(HERE, "vae_small.npz"), **res)
    print("vae_small.npz: loss_hard", loss_h, "loss_soft", loss_s)


if __name__ == "__main__":
    torch.set_num_threads(1)  # deterministic summation order on the host
    dalle_case()
    vae_case()
