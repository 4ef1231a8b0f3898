"""CPU oracle for the DALL-E transformer train step.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`dalle-mtf_amd/`) may import this
file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do.

PARITY: CALL GRAPH PINNED, THIRD-PARTY PRIMITIVES UNPINNED.  The reference (EleutherAI/DALLE-mtf) is pure
Python on top of `mesh_tensorflow==0.1.18` / `tensorflow==2.4.0` (requirements.txt:1-2); neither is vendored
under /root/reference nor installable here, and the reference ships no tests, golden vectors
or fixtures.  This file restates the arithmetic from the reference's own call
sites plus the published semantics of the third-party ops.  Round 4: the restatement is
checked against the reference's OWN files executed unmodified over shims of those libraries (oracle/refshim,
tests/golden/make_ref_callsite_golden.py, tests/test_reference_callsite.py: variables, logits, loss, every
gradient, schedule, clip, Adam step) -- which pins the call graph; the primitives underneath (restated in the shims
exactly as here) stay unpinned.  They are double-checked against independent PyTorch built-ins in
tests/test_oracle.py (F.layer_norm, F.scaled_dot_product_attention(scale=1), F.cross_entropy) and by analytic
known-answer tests.

Every function cites the reference file:line (paths relative to /root/reference) it follows.
Plain fp32 PyTorch-CPU; gradients come from torch autograd over this restatement.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# integer paths (bit-exact)
# --------------------------------------------------------------------------------------


def image_tokens_from_logits(vae_logits: np.ndarray) -> np.ndarray:
    """a1: src/model_fns.py:76-77 -- tokens = argmax(vae_logits, -1); reshape (B, g*g) row-major over
    (h, w); cast int32.  tf.math.argmax returns the smallest index among ties (Appendix A.8), which
    is also numpy's rule."""
    B = vae_logits.shape[0]
    tok = np.argmax(vae_logits, axis=-1)
    return tok.reshape(B, -1).astype(np.int32)


def assemble_tokens(text: np.ndarray, img_tokens: np.ndarray, text_vocab_size: int) -> np.ndarray:
    """a2: src/model_fns.py:118-119 -- concat(text[B,T], img_tokens + text_vocab_size) on axis 1."""
    text = np.asarray(text, dtype=np.int32)
    img = np.asarray(img_tokens, dtype=np.int32) + np.int32(text_vocab_size)
    return np.concatenate([text, img], axis=1).astype(np.int32)


def shift_labels(tokens: np.ndarray, eos_token_id: int) -> np.ndarray:
    """a11: src/dalle_mtf/models.py:407-410 (+ pad op src/dalle_mtf/ops.py:56-68) --
    labels = pad(tokens, [0, 1], value=eos)[:, 1:]  i.e. labels[t] = tokens[t+1], labels[S-1] = eos."""
    tokens = np.asarray(tokens, dtype=np.int32)
    out = np.empty_like(tokens)
    out[:, :-1] = tokens[:, 1:]
    out[:, -1] = np.int32(eos_token_id)
    return out


def truncate_or_pad_label(label: np.ndarray, text_seq_len: int, padding_id: int) -> np.ndarray:
    """src/input_fns.py:32-38 -- right-pad with padding_id then keep the first text_seq_len ids."""
    label = np.asarray(label, dtype=np.int32).reshape(-1)
    out = np.full((text_seq_len,), padding_id, dtype=np.int32)
    n = min(label.shape[0], text_seq_len)
    out[:n] = label[:n]
    return out


# --------------------------------------------------------------------------------------
# parameters (SURVEY.md Appendix B names / shapes / initialisers)
# --------------------------------------------------------------------------------------


class DalleConfig:
    """Hyper-parameters as the reference's DALLE.__init__ takes them (src/dalle_mtf/models.py:143-184)."""

    def __init__(self, n_embd, text_vocab_size, image_vocab_size, text_seq_len, image_seq_len,
                 n_layers, n_heads, eos_token_id=None):
        self.n_embd = n_embd
        self.text_vocab_size = text_vocab_size
        self.image_vocab_size = image_vocab_size
        self.text_seq_len = text_seq_len
        self.image_seq_len = image_seq_len
        self.total_seq_dim = text_seq_len + image_seq_len          # models.py:153
        self.n_layers = n_layers
        self.n_heads = n_heads
        self.total_tokens = text_vocab_size + image_vocab_size + 1  # models.py:157 (extra for EOS)
        self.eos_token_id = self.total_tokens - 1 if eos_token_id is None else eos_token_id  # :158
        assert n_embd % n_heads == 0                                # models.py:232
        self.kv_dim = n_embd // n_heads                             # models.py:167


def param_specs(cfg: DalleConfig) -> "OrderedDict[str, tuple]":
    """name -> (shape, kind, stddev); kinds: normal / ones / zeros.  Appendix B; sources:
    wte models.py:189-195 (N(0,.02)); wpe :204-211 (N(0,.01)); norm g/b :377-385;
    attn q/k/v/o: mtf attention_params_simple (models.py:235-241; Appendix A.1/A.2 -- bias-free,
    fold_scaling_into_initializer: q ~ N(0,(d*k)^-1/2), k,v ~ N(0,d^-1/2), o ~ N(0,(H*k)^-1/2));
    o_b :306-310; mlp_linear_1 N(0,.02), mlp_linear_2 N(0,.02/sqrt(L)) :320-321,363-371;
    to_logits :391-395."""
    d, H, k, L, V, S = cfg.n_embd, cfg.n_heads, cfg.kv_dim, cfg.n_layers, cfg.total_tokens, cfg.total_seq_dim
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    sp["embedding/wte"] = ((V, d), "normal", 0.02)
    sp["positional_embedding/wpe"] = ((S, d), "normal", 0.01)
    for i in range(L):
        p = f"layer_{i}/"
        sp[p + "norm_1/g"] = ((d,), "ones", 0.0)
        sp[p + "norm_1/b"] = ((d,), "zeros", 0.0)
        sp[p + "attn/q"] = ((d, H * k), "normal", (d * k) ** -0.5)
        sp[p + "attn/k"] = ((d, H * k), "normal", d ** -0.5)
        sp[p + "attn/v"] = ((d, H * k), "normal", d ** -0.5)
        sp[p + "attn/o"] = ((H * k, d), "normal", (H * k) ** -0.5)
        sp[p + "attn/compute_output_bias/o_b"] = ((d,), "zeros", 0.0)
        sp[p + "norm_2/g"] = ((d,), "ones", 0.0)
        sp[p + "norm_2/b"] = ((d,), "zeros", 0.0)
        sp[p + "mlp/mlp_linear_1/kernel"] = ((d, 4 * d), "normal", 0.02)
        sp[p + "mlp/mlp_linear_1/bias"] = ((4 * d,), "zeros", 0.0)
        sp[p + "mlp/mlp_linear_2/kernel"] = ((4 * d, d), "normal", 0.02 / math.sqrt(L))
        sp[p + "mlp/mlp_linear_2/bias"] = ((d,), "zeros", 0.0)
    sp["to_logits/layer_norm/g"] = ((d,), "ones", 0.0)
    sp["to_logits/layer_norm/b"] = ((d,), "zeros", 0.0)
    sp["to_logits/linear_out/kernel"] = ((d, V), "normal", 0.02)
    sp["to_logits/linear_out/bias"] = ((V,), "zeros", 0.0)
    return sp


def init_params(cfg: DalleConfig, seed: int = 1234, perturb: float = 0.0) -> "OrderedDict[str, np.ndarray]":
    """Initial weights.  TF's Philox streams cannot be reproduced (SURVEY §7 'RNG'), so parity is on
    identical exported weights: both sides load THESE arrays.  `perturb` adds N(0, perturb) noise to
    the ones/zeros-initialised tensors so tests exercise gains/biases non-trivially."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, kind, std) in param_specs(cfg).items():
        if kind == "normal":
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif kind == "ones":
            a = np.ones(shape, dtype=np.float32)
        else:
            a = np.zeros(shape, dtype=np.float32)
        if perturb > 0.0 and kind != "normal":
            a = a + rng.standard_normal(shape, dtype=np.float32) * np.float32(perturb)
        out[name] = a.astype(np.float32)
    return out


def n_params(cfg: DalleConfig) -> int:
    return int(sum(int(np.prod(s)) for (s, _, _) in param_specs(cfg).values()))


# --------------------------------------------------------------------------------------
# forward (fp32; optional bf16 emulation of the activation dtype)
# --------------------------------------------------------------------------------------


class _RoundBF16Grad(torch.autograd.Function):
    """y = bf16(x) in the forward and dx = bf16(dy) in the backward, written out explicitly.  Under the reference's `bf_16`
    policy a tensor of the activation dtype has a gradient of the activation dtype too (mtf computes every op's gradient in
    the op's own dtype, src/dalle_mtf/ops.py:76-82 VariableDType(activation=bf16)).  NOTE (round 4): the plain round trip
    `x.to(bfloat16).to(float32)` does exactly the same -- autograd's backward of the up-cast converts the incoming gradient
    to the bf16 source dtype -- so `bf16=True` has always rounded backward tensors as well (tests/test_oracle.py pins the two
    forms against each other); round 3's reading that the oracle kept backward activations in fp32 was wrong."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _rb(x: torch.Tensor, bf16) -> torch.Tensor:
    """Round-trip through bf16 when emulating the reference's bf16 activation dtype
    (src/dalle_mtf/ops.py:76-82: master bf16 / slice fp32 / activation bf16).  The gradient that flows back through the
    rounded tensor is rounded to bf16 as well (see _RoundBF16Grad).
    bf16 = False: fp32 throughout;  True (or "grad", the explicit spelling): forward tensor and its gradient rounded."""
    if not bf16:
        return x
    if bf16 is True:
        return x.to(torch.bfloat16).to(torch.float32)
    return _RoundBF16Grad.apply(x)


def _rbw(x: torch.Tensor, bf16) -> torch.Tensor:
    """activation-dtype cast of a WEIGHT (mtf casts the fp32 slice to bf16 for compute; the weight gradient is the bf16 result
    of a bf16 einsum, cast back to fp32 at src/optimizers.py:44).  bf16 = "fp32w": the weight is rounded in the forward but
    its gradient stays the fp32 value of that einsum -- what an implementation that accumulates weight gradients in fp32
    produces (bf16 rounding of a gradient tensor costs 1.6e-3 relative L2)."""
    if not bf16:
        return x
    if isinstance(bf16, str) and "fp32w" in bf16:
        return x + (x.to(torch.bfloat16).to(torch.float32) - x).detach()
    return _rb(x, bf16)


class _RoundGradOnly(torch.autograd.Function):
    """identity in the forward, bf16 rounding of the gradient in the backward.  Mode "...+ds" places it on the attention
    logits: the reference computes the logits einsum and its gradient in fp32 (float32_logits, Appendix A.3), a matrix-core
    implementation feeds dS = P (dP - delta) to the dQ / dK products as a bf16 operand.  Used only to ATTRIBUTE the remaining
    difference of the q / k weight gradients under teacher forcing (tests/test_headline_parity_gpu.py)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def _force(force, name: str, computed: torch.Tensor) -> torch.Tensor:
    """Teacher forcing for backward-parity tests: when `force` holds a tensor for site `name`, the FORWARD value of the site
    becomes that tensor (e.g. the activation the implementation under test stored for its backward) while the derivative
    structure stays the oracle's (straight-through: value = forced, gradient = gradient of `computed`).  Two faithful bf16
    implementations diverge in the forward by rounding-boundary and ReLU-mask flips that different fp32 summation orders
    cause (a flipped mask fraction f costs sqrt(f) in relative L2 upstream); forcing removes that divergence so that what
    remains is the backward arithmetic alone."""
    if force is None or name not in force:
        return computed
    f = torch.as_tensor(force[name], dtype=torch.float32).reshape(computed.shape)
    return f + (computed - computed.detach())


def layer_norm(x, g, b, eps=1e-5):
    """a6: src/dalle_mtf/models.py:373-389 + norm src/dalle_mtf/layers.py:30-33 --
    x -= mean(x); s = mean(x^2); x * rsqrt(s + eps) * g + b  (biased variance, eps 1e-5)."""
    u = x.mean(dim=-1, keepdim=True)
    xc = x - u
    s = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(s + eps) * g + b


def attn_mask(S: int) -> torch.Tensor:
    """a5: src/dalle_mtf/models.py:221-227 -- mask[i, j] = (i < j) * -1e10 (additive)."""
    i = torch.arange(S).view(S, 1)
    j = torch.arange(S).view(1, S)
    return (i < j).to(torch.float32) * -1e10


class _FlashCore(torch.autograd.Function):
    """softmax(logits) @ v with the backward written the way a flash-attention kernel computes it (attribution mode "+fa"):
    forward  P = softmax(logits) (fp32), a = bf16(bf16(P) @ v);
    backward dv = bf16(P)^T da;  dP = da v^T kept in fp32;  delta = sum_d da * a taken from the ROUNDED output a (the kernel
    reads the stored bf16 O);  dlogits = P * (dP - delta).
    The reference's autograd subtracts sum_j P_j dP_j instead -- the same number in exact arithmetic, but delta inherits the
    bf16 rounding of a, and its error enters every dlogit of a row with one sign (the row no longer sums to zero)."""

    @staticmethod
    def forward(ctx, logits, v):
        w = torch.exp(logits - torch.logsumexp(logits, dim=-1, keepdim=True))
        wr = w.to(torch.bfloat16).to(torch.float32)
        a = (wr @ v).to(torch.bfloat16).to(torch.float32)
        ctx.save_for_backward(w, wr, v, a)
        return a

    @staticmethod
    def backward(ctx, da):
        w, wr, v, a = ctx.saved_tensors
        da = da.to(torch.bfloat16).to(torch.float32)
        dv = wr.transpose(-1, -2) @ da
        dp = da @ v.transpose(-1, -2)
        delta = (da * a).sum(dim=-1, keepdim=True)
        return w * (dp - delta), dv


def attention(x, wq, wk, wv, wo, o_b, n_heads, mask, bf16=False, force=None, site=""):
    """a7: src/dalle_mtf/models.py:229-315.  q,k,v = x@Wq, x@Wk, x@Wv (bias-free, [d, H*k] heads-major,
    Appendix A.1); logits = q.k^T UNSCALED in fp32 (Appendix A.2/A.3) + mask; softmax over keys
    (exp(x - logsumexp(x))); @v; @Wo + o_b."""
    B, S, d = x.shape
    k = d // n_heads
    q = _force(force, site + "q", _rb(x @ wq, bf16)).view(B, S, n_heads, k).transpose(1, 2)   # [B,H,S,k]
    kk = _force(force, site + "k", _rb(x @ wk, bf16)).view(B, S, n_heads, k).transpose(1, 2)
    v = _force(force, site + "v", _rb(x @ wv, bf16)).view(B, S, n_heads, k).transpose(1, 2)
    logits = q @ kk.transpose(-1, -2)                              # fp32, no 1/sqrt(k)
    if isinstance(bf16, str) and "ds" in bf16:
        logits = _RoundGradOnly.apply(logits)
    logits = logits + mask
    if isinstance(bf16, str) and "fa" in bf16:
        a = _FlashCore.apply(logits, v)
        a = _force(force, site + "a", a.transpose(1, 2).reshape(B, S, d))
        return _rb(a @ wo + o_b, bf16)
    w = torch.exp(logits - torch.logsumexp(logits, dim=-1, keepdim=True))
    if isinstance(bf16, str) and "dp32" in bf16:
        # attribution mode: P rounded to bf16 in the forward, its gradient dP = dO V^T NOT rounded (a flash-style kernel keeps
        # dP in the fp32 accumulators; the reference's bf16 einsum rounds it).  dS = P (dP - sum_j P dP) cancels heavily when
        # dP varies little over the keys, so that rounding can dominate the q / k gradients of a layer.
        w = w + (w.to(torch.bfloat16).to(torch.float32) - w).detach()
    else:
        w = _rb(w, bf16)                                           # "cast to v dtype" (A.3)
    a = _rb(w @ v, bf16)                                           # [B,H,S,k]
    a = _force(force, site + "a", a.transpose(1, 2).reshape(B, S, d))
    return _rb(a @ wo + o_b, bf16)


def mlp(x, w1, b1, w2, b2, bf16=False, force=None, site=""):
    """a8: src/dalle_mtf/models.py:317-324 + linear :361-371 -- relu(x@W1+b1)@W2+b2."""
    h = _force(force, site + "h", _rb(torch.relu(x @ w1 + b1), bf16))
    return _rb(h @ w2 + b2, bf16)


def forward_hidden(P: Dict[str, torch.Tensor], tokens: torch.Tensor, cfg: DalleConfig, bf16=False,
                   taps: Optional[dict] = None, force: Optional[dict] = None) -> torch.Tensor:
    """a3,a4,a9: embedding models.py:186-201, positional :203-219, transformer :337-346, block :326-335.
    force: optional {site: tensor} teacher forcing (see _force); sites: "embed", "layer_{i}/xn1|q|k|v|a|x1|xn2|h|out"."""
    B, S = tokens.shape
    W = (lambda n: _rbw(P[n], bf16))                                # activation-dtype cast of the weights
    x = W("embedding/wte")[tokens.long()]                          # mtf.gather (A.6)
    x = _force(force, "embed", _rb(x + W("positional_embedding/wpe")[:S], bf16))
    mask = attn_mask(S)
    if taps is not None:
        taps["embed"] = x
    for i in range(cfg.n_layers):
        p = f"layer_{i}/"
        h = _force(force, p + "xn1", _rb(layer_norm(x, W(p + "norm_1/g"), W(p + "norm_1/b")), bf16))
        a = attention(h, W(p + "attn/q"), W(p + "attn/k"), W(p + "attn/v"), W(p + "attn/o"),
                      W(p + "attn/compute_output_bias/o_b"), cfg.n_heads, mask, bf16, force, p)
        x = _force(force, p + "x1", _rb(x + a, bf16))
        h = _force(force, p + "xn2", _rb(layer_norm(x, W(p + "norm_2/g"), W(p + "norm_2/b")), bf16))
        m = mlp(h, W(p + "mlp/mlp_linear_1/kernel"), W(p + "mlp/mlp_linear_1/bias"),
                W(p + "mlp/mlp_linear_2/kernel"), W(p + "mlp/mlp_linear_2/bias"), bf16, force, p)
        x = _force(force, p + "out", _rb(x + m, bf16))
        if taps is not None:
            taps[f"layer_{i}"] = x
    return x


def to_logits(P, x, bf16=False, force=None):
    """a10: src/dalle_mtf/models.py:391-395 -- LN(x) @ Wout + bout, then cast to fp32."""
    W = (lambda n: _rbw(P[n], bf16))
    h = _force(force, "xnf", _rb(layer_norm(x, W("to_logits/layer_norm/g"), W("to_logits/layer_norm/b")), bf16))
    return _rb(h @ W("to_logits/linear_out/kernel") + W("to_logits/linear_out/bias"), bf16)


def loss_fn(logits: torch.Tensor, labels: torch.Tensor, num_microbatches: int = 1):
    """a12: src/dalle_mtf/models.py:348-359 + mtf softmax_cross_entropy_with_logits (Appendix A.5):
    loss_batch = logsumexp(logits) - logits[label] (z_loss 0); loss = mean over ALL B*S positions;
    / num_microbatches."""
    lse = torch.logsumexp(logits, dim=-1)
    picked = torch.gather(logits, -1, labels.long().unsqueeze(-1)).squeeze(-1)
    loss_batch = lse - picked
    loss = loss_batch.mean() / num_microbatches
    return loss, loss_batch


def forward(P: Dict[str, torch.Tensor], tokens: np.ndarray, cfg: DalleConfig, bf16=False,
            return_logits=False, taps: Optional[dict] = None, force: Optional[dict] = None):
    """DALLE.forward src/dalle_mtf/models.py:397-416."""
    tok = torch.as_tensor(np.asarray(tokens), dtype=torch.int64)
    x = forward_hidden(P, tok, cfg, bf16, taps, force)
    logits = to_logits(P, x, bf16, force)
    labels = torch.as_tensor(shift_labels(np.asarray(tokens), cfg.eos_token_id), dtype=torch.int64)
    loss, loss_batch = loss_fn(logits, labels)
    if return_logits:
        return loss, loss_batch, logits
    return loss, loss_batch


def loss_and_grads(params_np: Dict[str, np.ndarray], tokens: np.ndarray, cfg: DalleConfig, bf16=False, force=None):
    """mtf.gradients([loss], trainable_variables) src/optimizers.py:34; cast to fp32 :44."""
    P = OrderedDict((n, torch.tensor(a, dtype=torch.float32, requires_grad=True)) for n, a in params_np.items())
    loss, _ = forward(P, tokens, cfg, bf16, force=force)
    loss.backward()
    grads = OrderedDict((n, (p.grad.detach().numpy().copy() if p.grad is not None
                             else np.zeros(tuple(p.shape), np.float32))) for n, p in P.items())
    return float(loss.detach()), grads


# --------------------------------------------------------------------------------------
# optimizer (a13)
# --------------------------------------------------------------------------------------


def learning_rate(step: int, lr: float, train_steps: int, warmup_steps: int = 3000,
                  lr_decay: str = "cosine", lr_decay_end: Optional[int] = None) -> float:
    """src/optimizers.py:46-76.  cosine: tf.train.cosine_decay(lr, step, end, alpha=0.1)
    = lr*((1-a)*0.5*(1+cos(pi*min(step,T)/T)) + a) (Appendix A.8); linear: polynomial_decay power 1 to
    0.1*lr; warm-up: lr * step/warmup while step < warmup (int compare, float32 math)."""
    end = train_steps if lr_decay_end is None else lr_decay_end
    f32 = np.float32
    if lr_decay == "linear":
        s = min(step, end)
        v = f32((lr - lr * 0.1) * (1.0 - s / end) + lr * 0.1)
    elif lr_decay == "cosine":
        s = min(step, end)
        cosine = 0.5 * (1.0 + math.cos(math.pi * s / end))
        v = f32(lr * ((1.0 - 0.1) * cosine + 0.1))
    else:
        v = f32(lr)
    if warmup_steps > 0:
        if step < warmup_steps:
            v = f32(v * f32(f32(step) / f32(warmup_steps)))
    return float(v)


def clip_by_global_norm(grads: Dict[str, np.ndarray], clip_norm: float = 1.0):
    """src/optimizers.py:11-16 -- g * (c / max(||g||_2, c))."""
    sq = np.float64(0.0)
    for g in grads.values():
        sq += np.sum(np.square(g.astype(np.float64)))
    gn = np.float32(np.sqrt(sq))
    mult = np.float32(clip_norm) / np.maximum(gn, np.float32(clip_norm))
    return OrderedDict((n, (g * mult).astype(np.float32)) for n, g in grads.items()), float(gn)


def use_weight_decay(name: str, weight_decay: float) -> bool:
    """src/optimizers.py:82-89 exclude_from_weight_decay=["norm","bias"] (re.search), formula :181-188."""
    if not weight_decay:
        return False
    return ("norm" not in name) and ("bias" not in name)


def adam_step(params, grads, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0):
    """mtf.optimize.AdamWeightDecayOptimizer.apply_grad, mirrored in src/optimizers.py:154-177:
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; update = m/(sqrt(v)+eps) [+ wd*p]; p -= lr*update.
    NO bias correction; eps 1e-6 (src/optimizers.py:87)."""
    f32 = np.float32
    for n in params:
        g = grads[n].astype(np.float32)
        m[n] = f32(beta1) * m[n] + f32(1.0 - beta1) * g
        v[n] = f32(beta2) * v[n] + f32(1.0 - beta2) * (g * g)
        upd = m[n] / (np.sqrt(v[n]) + f32(eps))
        if use_weight_decay(n, weight_decay):
            upd = upd + params[n] * f32(weight_decay)
        params[n] = (params[n] - f32(lr) * upd).astype(np.float32)
    return params, m, v


def train_step(params, m, v, tokens, cfg: DalleConfig, step: int, hp: dict, bf16=False):
    """One full reference train step: fwd/bwd (src/model_fns.py:168-181) -> clip -> Adam
    (src/optimizers.py:102-103).  hp keys: lr, train_steps, warmup_steps, lr_decay, gradient_clipping,
    weight_decay, beta_1, beta_2, epsilon (defaults src/optimizers.py:24-28,84-87)."""
    loss, grads = loss_and_grads(params, tokens, cfg, bf16)
    clip = hp.get("gradient_clipping", 1.0)
    gnorm = None
    if clip is not None:
        grads, gnorm = clip_by_global_norm(grads, clip)
    lr = learning_rate(step, hp["lr"], hp["train_steps"], hp.get("warmup_steps", 3000),
                       hp.get("lr_decay", "cosine"), hp.get("lr_decay_end"))
    adam_step(params, grads, m, v, lr, hp.get("beta_1", 0.9), hp.get("beta_2", 0.999),
              hp.get("epsilon", 1e-6), hp.get("weight_decay", 0.0))
    return loss, gnorm, lr


# --------------------------------------------------------------------------------------
# synthetic inputs (SURVEY §8(d) 'Synthetic inputs')
# --------------------------------------------------------------------------------------


def synthetic_captions(B: int, text_seq_len: int, text_vocab_size: int, seed: int = 1) -> np.ndarray:
    """int32 uniform [0, padding_id) for a random length in [1, T], right-padded with
    padding_id = text_vocab_size - 1 (train_dalle.py:49: the GPT-2 '<|padding|>' id 50257)."""
    rng = np.random.default_rng(seed)
    pad = text_vocab_size - 1
    out = np.full((B, text_seq_len), pad, dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(1, text_seq_len + 1))
        out[b, :n] = rng.integers(0, pad, size=n, dtype=np.int32)
    return out


def synthetic_image_tokens(B: int, image_seq_len: int, image_vocab_size: int, seed: int = 2) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, image_vocab_size, size=(B, image_seq_len), dtype=np.int32)
