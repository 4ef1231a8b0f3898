"""gumbel_softmax / mse_loss surface of src/vae_tf/layers.py, bound to the HIP kernels (csrc/vae.hip)."""
import torch

import dalle_hip as dh


def gumbel_softmax(logits: torch.Tensor, axis=-1, temperature=1.0, hard=True, noise=None):
    """reference layers.py:4-21 on a [..., T] fp32 device tensor; returns (sample, soft_sample) as bf16.
    `noise`: uniforms in [1e-9, 1) (drawn with torch.rand when omitted)."""
    assert axis in (-1, logits.dim() - 1)
    T = logits.shape[-1]
    M = logits.numel() // T
    lg = logits.contiguous().view(M, T).float()
    u = torch.empty_like(lg).uniform_(1e-9, 1.0) if noise is None else noise.to(lg.device).reshape(M, T).float().contiguous()
    y = torch.empty(M, T, dtype=torch.bfloat16, device=lg.device)
    ys = torch.empty_like(y)
    dh.gumbel_softmax_fwd(lg, u, y, ys, None, M, T, temperature, hard)
    return y.view(logits.shape), ys.view(logits.shape)


def mse_loss(prediction_bf16_padded, target_fp32, cin):
    """reference layers.py:24-25: mean((prediction - target)^2); prediction is the [N, Cp] padded bf16 activation."""
    N, Cp = prediction_bf16_padded.shape
    loss = torch.zeros(1, dtype=torch.float32, device=prediction_bf16_padded.device)
    ws = torch.empty(dh.mse_workspace_bytes(), dtype=torch.uint8, device=loss.device)
    dh.mse_loss(target_fp32.contiguous().view(N, cin), prediction_bf16_padded, None, loss, N, cin, Cp, 1.0, ws)
    return loss[0]
