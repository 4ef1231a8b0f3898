"""CPU oracle for the discrete-VAE path.

TEST INFRASTRUCTURE ONLY (see oracle/dalle_oracle.py header).  PARITY: CALL GRAPH PINNED, THIRD-PARTY PRIMITIVES
UNPINNED: restated from the reference's call sites (src/vae_tf/models.py, src/vae_tf/layers.py, src/model_fns_tf.py) plus the
published semantics of tf.layers.conv2d / conv2d_transpose / tf.train.AdamOptimizer
(SURVEY.md Appendix A.8); TensorFlow 2.4.0 is not installable here.  Round 4: checked against the reference's own
src/vae_tf files executed over a TF shim (oracle/refshim, tests/test_reference_callsite.py).

Tensors are NHWC at the API (as the reference, vae_tf/models.py:171 comment) and kernels are in the
TF layouts of Appendix B ([kh,kw,Cin,Cout]; transpose conv [kh,kw,Cout,Cin]).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


class VaeConfig:
    """DiscreteVAE.__init__ src/vae_tf/models.py:47-79."""

    def __init__(self, num_tokens: int, dimensions: int, convblocks: Sequence[Sequence[int]],
                 input_channels: int = 3, stack_factor: int = 1, use_bf16: bool = False,
                 recompute_grad: bool = False):
        self.num_tokens = num_tokens
        self.H = self.W = dimensions
        self.convblocks = [tuple(b) for b in convblocks]
        self.num_ch = input_channels
        self.stack_factor = stack_factor
        assert math.log2(stack_factor).is_integer()      # models.py:78
        self.use_bf16 = use_bf16
        self.recompute_grad = recompute_grad
        self.n_hid = self.convblocks[-1][1]
        self.grid = (self.H // stack_factor) // (2 ** len(self.convblocks))


def param_specs(cfg: VaeConfig) -> "OrderedDict[str, tuple]":
    """Appendix B VAE names.  encoder vae_tf/models.py:88-103; codebook :111-113;
    decoder :131-155 (final conv named 'conv2d' by tf.layers default)."""
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    cin = cfg.num_ch * cfg.stack_factor ** 2
    for b, (stack, ch) in enumerate(cfg.convblocks):
        for i in range(stack):
            p = f"encoder/block_{b}/layer_{i}/"
            if i == 0:
                sp[p + "conv_downsample/kernel"] = (4, 4, cin, ch)
                sp[p + "conv_downsample/bias"] = (ch,)
            else:
                sp[p + "conv_in/kernel"] = (3, 3, ch, ch)
                sp[p + "conv_in/bias"] = (ch,)
                sp[p + "conv_out/kernel"] = (3, 3, ch, ch)
                sp[p + "conv_out/bias"] = (ch,)
        cin = ch
    sp["codebook/codebook"] = (cfg.n_hid, cfg.num_tokens)
    cin = cfg.n_hid
    for b, (stack, ch) in enumerate(reversed(cfg.convblocks)):
        for i in range(stack):
            p = f"decoder/block_{b}/layer_{i}/"
            if i == 0:
                sp[p + "conv_upsample/kernel"] = (4, 4, ch, cin)   # [kh,kw,Cout,Cin]
                sp[p + "conv_upsample/bias"] = (ch,)
            else:
                sp[p + "conv_in/kernel"] = (3, 3, ch, ch)
                sp[p + "conv_in/bias"] = (ch,)
                sp[p + "conv_out/kernel"] = (3, 3, ch, ch)
                sp[p + "conv_out/bias"] = (ch,)
        cin = ch
    cout = cfg.num_ch * cfg.stack_factor ** 2
    sp["decoder/conv2d/kernel"] = (1, 1, cin, cout)
    sp["decoder/conv2d/bias"] = (cout,)
    return sp


def init_params(cfg: VaeConfig, seed: int = 4321, bias_perturb: float = 0.0):
    """glorot-uniform kernels / zero biases (tf.layers defaults, Appendix A.8); codebook glorot-uniform
    (tf.get_variable with no initializer, vae_tf/models.py:113)."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_specs(cfg).items():
        if len(shape) == 1:
            a = np.zeros(shape, np.float32)
            if bias_perturb > 0:
                a = (rng.standard_normal(shape) * bias_perturb).astype(np.float32)
        else:
            if len(shape) == 4:
                rf = shape[0] * shape[1]
                fan_in, fan_out = shape[2] * rf, shape[3] * rf
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        out[name] = a
    return out


def n_params(cfg: VaeConfig) -> int:
    return int(sum(int(np.prod(s)) for s in param_specs(cfg).values()))


# ------------------------------------------------------------------ ops


def _rb(x, bf16):
    """use_bf16: round trip through bf16 (the gradient flowing back through the rounded tensor is rounded to bf16 as well:
    autograd's backward of the up-cast converts it to the bf16 source dtype -- see oracle/dalle_oracle.py _RoundBF16Grad)"""
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _rbw(x, bf16):
    """cast of a WEIGHT to the activation dtype; use_bf16 = "fp32w" rounds the weight in the forward but keeps its gradient in
    fp32 (an implementation that accumulates weight gradients in fp32)"""
    if not bf16:
        return x
    if bf16 == "fp32w":
        return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()
    return x.to(torch.bfloat16).to(torch.float32)


def _force(force, name, computed):
    """teacher forcing for backward-parity tests (see oracle/dalle_oracle.py _force): forward value = force[name] (the
    activation the implementation under test stored), gradient = the oracle's"""
    if force is None or name not in force:
        return computed
    f = torch.as_tensor(force[name], dtype=torch.float32).reshape(computed.shape)
    return f + (computed - computed.detach())


def conv2d_same(x_nhwc, kernel, bias, stride):
    """tf.layers.conv2d(padding='SAME') NHWC, kernel [kh,kw,Cin,Cout] (Appendix A.8).
    SAME: out = ceil(H/s); pad_total = max((out-1)*s + k - H, 0); before = total//2."""
    kh, kw, cin, cout = kernel.shape
    x = x_nhwc.permute(0, 3, 1, 2)
    H, W = x.shape[2], x.shape[3]
    oh, ow = -(-H // stride), -(-W // stride)
    ph = max((oh - 1) * stride + kh - H, 0)
    pw = max((ow - 1) * stride + kw - W, 0)
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    w = kernel.permute(3, 2, 0, 1)
    y = F.conv2d(x, w, bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose_same(x_nhwc, kernel, bias, stride=2):
    """tf.layers.conv2d_transpose(k=4, s=2, 'SAME'): out = s*H; kernel [kh,kw,Cout,Cin]; equals the
    gradient w.r.t. the input of the SAME stride-2 conv with the same kernel (Appendix A.8)."""
    kh, kw, cout, cin = kernel.shape
    assert (kh, kw, stride) == (4, 4, 2)
    x = x_nhwc.permute(0, 3, 1, 2)
    w = kernel.permute(3, 2, 0, 1)                 # [Cin, Cout, kh, kw]
    y = F.conv_transpose2d(x, w, bias, stride=2, padding=1)
    return y.permute(0, 2, 3, 1)


def space_to_depth(x, r):
    B, H, W, C = x.shape
    x = x.view(B, H // r, r, W // r, r, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H // r, W // r, r * r * C)


def depth_to_space(x, r):
    B, H, W, C = x.shape
    c = C // (r * r)
    x = x.view(B, H, W, r, r, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H * r, W * r, c)


def encoder(P, img, cfg: VaeConfig, force=None):
    """v1: vae_tf/models.py:81-120.  Per block: conv 4x4 s2 SAME (no activation), then (stack-1) x
    `x + conv3x3(relu(conv3x3(x)))`; then fp32 x @ codebook.
    force: optional {variable scope of a conv: its output} teacher forcing (conv_in: after the ReLU; conv_out: after the residual add)"""
    bf = cfg.use_bf16
    x = _rb(img, bf)
    if cfg.stack_factor > 1:
        x = space_to_depth(x, cfg.stack_factor)
    for b, (stack, ch) in enumerate(cfg.convblocks):
        for i in range(stack):
            p = f"encoder/block_{b}/layer_{i}/"
            if i == 0:
                x = _rb(conv2d_same(x, _rbw(P[p + "conv_downsample/kernel"], bf), _rbw(P[p + "conv_downsample/bias"], bf), 2), bf)
                x = _force(force, p + "conv_downsample", x)
            else:
                o = _rb(conv2d_same(x, _rbw(P[p + "conv_in/kernel"], bf), _rbw(P[p + "conv_in/bias"], bf), 1), bf)
                o = _force(force, p + "conv_in", torch.relu(o))
                o = _rb(conv2d_same(o, _rbw(P[p + "conv_out/kernel"], bf), _rbw(P[p + "conv_out/bias"], bf), 1), bf)
                x = _force(force, p + "conv_out", _rb(x + o, bf))
    return x @ P["codebook/codebook"]              # fp32 matmul, models.py:115-118


def gumbel_softmax(logits, u, temperature=1.0, hard=True):
    """v2: vae_tf/layers.py:4-21 with INJECTED uniforms u ~ U[1e-9, 1) (TF's Philox stream cannot be
    reproduced): g = -log(-log u); y = softmax((logits + g)/T); hard: one_hot(argmax y) with
    straight-through gradient (stop_gradient(hard - y) + y)."""
    g = -torch.log(-torch.log(u))
    y = torch.softmax((logits + g) / temperature, dim=-1)
    if hard:
        idx = torch.argmax(y, dim=-1)
        y_hard = F.one_hot(idx, y.shape[-1]).to(y.dtype)
        y = (y_hard - y).detach() + y
    return y


def decoder(P, x, cfg: VaeConfig, force=None):
    """v3: vae_tf/models.py:123-163.  x @ codebook^T (tied); per reversed block: conv-transpose 4x4 s2
    (no activation) then residual stacks; final 1x1 conv; fp32; depth_to_space."""
    bf = cfg.use_bf16
    x = x @ P["codebook/codebook"].t()
    x = _force(force, "dec_in", _rb(x, bf))
    for b, (stack, ch) in enumerate(reversed(cfg.convblocks)):
        for i in range(stack):
            p = f"decoder/block_{b}/layer_{i}/"
            if i == 0:
                x = _rb(conv2d_transpose_same(x, _rbw(P[p + "conv_upsample/kernel"], bf), _rbw(P[p + "conv_upsample/bias"], bf)), bf)
                x = _force(force, p + "conv_upsample", x)
            else:
                o = _rb(conv2d_same(x, _rbw(P[p + "conv_in/kernel"], bf), _rbw(P[p + "conv_in/bias"], bf), 1), bf)
                o = _force(force, p + "conv_in", torch.relu(o))
                o = _rb(conv2d_same(o, _rbw(P[p + "conv_out/kernel"], bf), _rbw(P[p + "conv_out/bias"], bf), 1), bf)
                x = _force(force, p + "conv_out", _rb(x + o, bf))
    x = _rb(conv2d_same(x, _rbw(P["decoder/conv2d/kernel"], bf), _rbw(P["decoder/conv2d/bias"], bf), 1), bf)
    x = _force(force, "decoder/conv2d", x)
    if cfg.stack_factor > 1:
        x = depth_to_space(x, cfg.stack_factor)
    return x


def mse_loss(pred, target):
    """v4: vae_tf/layers.py:24-25."""
    return torch.mean((pred - target) ** 2)


def forward(P, img, cfg: VaeConfig, u=None, return_recon_loss=False, return_logits=False,
            hard_gumbel=True, temperature=1.0, force=None):
    """DiscreteVAE.forward vae_tf/models.py:165-184."""
    logits = encoder(P, img, cfg, force)
    if return_logits:
        return logits
    y = gumbel_softmax(logits, u, temperature, hard_gumbel)
    out = decoder(P, y, cfg, force)
    if not return_recon_loss:
        return out
    return mse_loss(img, out), out


def temperature(step: int, params: dict) -> float:
    """v6: src/model_fns_tf.py:40-45."""
    if params.get("temp_anneal_steps", None):
        frac = min(np.float32(step) / np.float32(params["temp_anneal_steps"]), np.float32(1.0))
        return float(np.float32(params["temp_start"]) - frac * np.float32(params["temp_start"] - params["temp"]))
    return float(params.get("temp", 1.0))


def loss_and_grads(params_np, img_np, u_np, cfg: VaeConfig, hard=True, temp=1.0, force=None):
    P = OrderedDict((n, torch.tensor(a, requires_grad=True)) for n, a in params_np.items())
    loss, out = forward(P, torch.tensor(img_np), cfg, torch.tensor(u_np), return_recon_loss=True,
                        hard_gumbel=hard, temperature=temp, force=force)
    loss.backward()
    grads = OrderedDict((n, p.grad.detach().numpy().copy()) for n, p in P.items())
    return float(loss.detach()), grads, out.detach().numpy()


def tf_adam_step(params, grads, m, v, step_t: int, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (src/model_fns_tf.py:58-60; Appendix A.8): t = step+1;
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; p -= lr_t * m/(sqrt(v)+eps)."""
    f32 = np.float32
    t = step_t
    lr_t = f32(lr) * f32(math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    for n in params:
        g = grads[n]
        m[n] = f32(beta1) * m[n] + f32(1 - beta1) * g
        v[n] = f32(beta2) * v[n] + f32(1 - beta2) * g * g
        params[n] = (params[n] - lr_t * m[n] / (np.sqrt(v[n]) + f32(eps))).astype(np.float32)
    return params, m, v


def synthetic_images(B: int, size: int, channels: int = 3, seed: int = 0) -> np.ndarray:
    """uint8 uniform -> (x - 127.5)/127.5  (src/input_fns.py:20)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, size=(B, size, size, channels), dtype=np.uint8)
    return ((x.astype(np.float32) - 127.5) / 127.5).astype(np.float32)


def synthetic_uniforms(shape, seed: int = 7) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.uniform(1e-9, 1.0, size=shape).astype(np.float32)
