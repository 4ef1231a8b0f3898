"""DALLE -- same constructor / forward surface as the reference class (src/dalle_mtf/models.py:141-416),
re-hosted on the MI355X engine (hand-written HIP kernels behind libdalle_hip's C ABI)."""
from collections import defaultdict

import torch

from .engine import DalleEngine
from .ops import get_variable_dtype


class DALLE:
    def __init__(self, n_embd, text_vocab_size=12800, image_vocab_size=512, text_seq_len=256, image_seq_len=1024,
                 n_layers=6, n_heads=8, batch_size=32, bf_16=True, attn_mask=None, mode="train",
                 is_incremental_inference=False, context=None, loss_fn=None, params=None, eos_token_id=None,
                 activation_fn=None, device="cuda", process_group=None, world_size=1, global_batch_size=None, comm=None):
        self.n_embd = n_embd
        self.text_vocab_size = text_vocab_size
        self.image_vocab_size = image_vocab_size
        self.text_seq_len = text_seq_len
        self.image_seq_len = image_seq_len
        self.total_seq_dim = text_seq_len + image_seq_len
        self.n_layers = n_layers
        self.n_heads = n_heads
        self.total_tokens = text_vocab_size + image_vocab_size + 1  # extra for EOS (reference :157)
        self.eos_token_id = self.total_tokens - 1 if eos_token_id is None else eos_token_id
        self.bf_16 = bf_16
        self.variable_dtype = get_variable_dtype(bf_16)
        if not bf_16:
            # the reference's shipped configs/dalle_example.json has "bf_16": false = fp32 activations; the MI355X kernels keep fp32
            # masters / optimizer state but ALWAYS compute activations in bf16 with fp32 accumulation (stated
            # deviation): say so instead of silently running narrower arithmetic than the reference would
            import warnings
            warnings.warn("DALLE(bf_16=False): fp32 master weights and optimizer state are kept, but the MI355X kernels compute "
                          "activations in bf16 with fp32 accumulation -- narrower than the reference's fp32 activations for this "
                          "setting (loss agrees to ~1e-5 relative, gradients to a few percent per tensor; DESIGN.md §2)", stacklevel=2)
        self.mode = mode
        self.batch_size = batch_size
        if attn_mask is not None:
            raise NotImplementedError("custom attn_mask: the kernels implement the reference's causal mask (models.py:221-227)")
        if is_incremental_inference or context is not None:
            raise NotImplementedError("incremental inference is unfinished upstream (predict raises NotImplementedError, model_fns.py:135)")
        if loss_fn is not None or activation_fn is not None:
            raise NotImplementedError("custom loss_fn / activation_fn: kernels implement softmax-CE and ReLU (reference defaults)")
        params = {} if params is None else params
        self.params = defaultdict(lambda: None, params)
        for k in ("embed_dropout", "attention_dropout", "residual_dropout"):
            if self.params.get(k):
                raise NotImplementedError(f"{k} > 0 is not supported (all shipped configs use 0)")
        if (self.params.get("scale_type") or "scale_by_depth") != "scale_by_depth":
            raise NotImplementedError("scale_type other than scale_by_depth")
        self.engine = DalleEngine(n_embd, n_layers, n_heads, text_vocab_size, image_vocab_size, text_seq_len,
                                  image_seq_len, batch_size, global_batch_size=global_batch_size,
                                  eos_token_id=eos_token_id, hparams=dict(self.params), device=device,
                                  process_group=process_group, world_size=world_size, comm=comm)
        self.dimensions = {"embed_dim": n_embd, "final_vocab_dim": self.total_tokens, "total_seq_dim": self.total_seq_dim,
                           "heads_dim": n_heads, "kv_dim": n_embd // n_heads, "batch_dim": batch_size}

    def variables(self):
        """name -> shape under the reference's checkpoint names."""
        d, V, S = self.n_embd, self.total_tokens, self.total_seq_dim
        out = {"embedding/wte": (V, d), "positional_embedding/wpe": (S, d)}
        for i in range(self.n_layers):
            p = f"layer_{i}/"
            out.update({p + "norm_1/g": (d,), p + "norm_1/b": (d,), p + "attn/q": (d, d), p + "attn/k": (d, d),
                        p + "attn/v": (d, d), p + "attn/o": (d, d), p + "attn/compute_output_bias/o_b": (d,),
                        p + "norm_2/g": (d,), p + "norm_2/b": (d,), p + "mlp/mlp_linear_1/kernel": (d, 4 * d),
                        p + "mlp/mlp_linear_1/bias": (4 * d,), p + "mlp/mlp_linear_2/kernel": (4 * d, d),
                        p + "mlp/mlp_linear_2/bias": (d,)})
        out.update({"to_logits/layer_norm/g": (d,), "to_logits/layer_norm/b": (d,),
                    "to_logits/linear_out/kernel": (d, V), "to_logits/linear_out/bias": (V,)})
        return out

    def sample(self, text_tokens, vae=None, temperature=1.0, top_k=0, seed=0):
        """text ids [B, text_seq_len] -> image-token ids [B, image_seq_len] (and the decoded images when a DiscreteVAE is
        given): the generation path the reference leaves unfinished (model_fns.py:135-136)."""
        toks = self.engine.sample_image_tokens(text_tokens, temperature=temperature, top_k=top_k, seed=seed)
        return (toks, vae.decode_tokens(toks)) if vae is not None else toks

    def forward(self, features, return_loss=True, return_logits=False):
        """features["tokens"]: int32 [B, S] device tensor.  Returns (loss, loss_batch[, logits]) like the
        reference (models.py:397-416); with return_loss=False returns the fp32 logits only."""
        tokens = features["tokens"] if isinstance(features, dict) else features
        tokens = tokens.to(device=self.engine.dev, dtype=torch.int32)
        need_grad = self.mode == "train" and return_loss and not return_logits
        loss = self.engine.forward(tokens, need_grad=need_grad)
        if not return_loss:
            return self.engine.logits()
        loss_batch = self.engine.loss_rows.view(self.batch_size, self.total_seq_dim)
        if return_logits:
            return loss[0], loss_batch, self.engine.logits()
        return loss[0], loss_batch
