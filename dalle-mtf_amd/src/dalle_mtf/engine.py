"""DalleEngine -- sequences the HIP kernels of libdalle_hip into the DALL-E train step.

Replaces, for the hot path, what the reference obtains from Mesh-TensorFlow's graph + lowering
(src/model_fns.py:80-94,168-202 -> mtf.Graph / mtf.Lowering) and its optimizer
(src/optimizers.py:19-104): forward (src/dalle_mtf/models.py:397-416), a hand-written backward,
global-norm clip + Adam without bias correction, and the data-parallel gradient all-reduce that mtf
inserts implicitly for `layout: batch_dim:data` -- here RCCL via
torch.distributed, bucketed and overlapped with backward on the collective's own stream.

PyTorch is plumbing only (device memory, streams, torch.distributed); all arithmetic runs in the
hand-written kernels behind the C ABI.  There is no CPU fallback.

Memory plan (HBM, per GPU): parameters live in ONE flat fp32 buffer (+ flat grads, Adam m, v) laid out in
reverse-usage order [to_logits | layer_{L-1} .. layer_0 | wpe | wte] so finished gradients always form
a contiguous prefix (= all-reduce buckets).  bf16 compute copies: `pb` (same offsets, natural [in,out]
layout, written by the Adam kernel) and `pbt` ([out,in] copies of the GEMM weights for the forward).
Activations needed by backward are kept in bf16 per layer (no recompute; 288 GB HBM).
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

import dalle_hip as dh
from ..dp import GradReducer

HEAD_DIM = 128
ALIGN = 128  # elements


def _round_up(x, m):
    return (x + m - 1) // m * m


class ParamLayout:
    """Flat layout of the trainable variables.  Internal tensors fuse q|k|v into one [d, 3d] matrix and
    pad the vocabulary axis of the output projection to a multiple of 128; `export`/`load` translate
    to/from the reference's variable names and shapes."""

    def __init__(self, n_embd, n_layers, n_heads, total_tokens, total_seq):
        d, L, V, S = n_embd, n_layers, total_tokens, total_seq
        self.d, self.L, self.V, self.S = d, L, V, S
        self.Vp = _round_up(V, 128)
        ent: List[Tuple[str, tuple]] = []
        ent += [("to_logits/linear_out/kernel", (d, self.Vp)), ("to_logits/linear_out/bias", (self.Vp,)),
                ("to_logits/layer_norm/g", (d,)), ("to_logits/layer_norm/b", (d,))]
        for i in reversed(range(L)):
            p = f"layer_{i}/"
            ent += [(p + "mlp/mlp_linear_2/kernel", (4 * d, d)), (p + "mlp/mlp_linear_2/bias", (d,)),
                    (p + "mlp/mlp_linear_1/kernel", (d, 4 * d)), (p + "mlp/mlp_linear_1/bias", (4 * d,)),
                    (p + "norm_2/g", (d,)), (p + "norm_2/b", (d,)),
                    (p + "attn/o", (d, d)), (p + "attn/compute_output_bias/o_b", (d,)),
                    (p + "attn/qkv", (d, 3 * d)),
                    (p + "norm_1/g", (d,)), (p + "norm_1/b", (d,))]
        ent += [("positional_embedding/wpe", (S, d)), ("embedding/wte", (V, d))]
        self.entries = ent
        self.offset: Dict[str, int] = {}
        self.shape: Dict[str, tuple] = {}
        off = 0
        for name, shp in ent:
            self.offset[name] = off
            self.shape[name] = shp
            off += _round_up(int(np.prod(shp)), ALIGN)
        self.total = off
        # transposed ([out, in]) bf16 copies consumed by the forward GEMMs
        self.t_offset: Dict[str, int] = {}
        toff = 0
        for name, shp in ent:
            if len(shp) == 2 and ("kernel" in name or "attn/" in name):
                self.t_offset[name] = toff
                toff += _round_up(int(np.prod(shp)), ALIGN)
        self.t_total = toff
        # bucket boundaries (prefix ends) in element offsets: after head, after each layer, end
        self.bucket_ends: List[int] = []
        self.bucket_ends.append(self.offset[f"layer_{L-1}/mlp/mlp_linear_2/kernel"] if L > 0 else self.offset["positional_embedding/wpe"])
        for i in reversed(range(L)):
            nxt = f"layer_{i-1}/mlp/mlp_linear_2/kernel" if i > 0 else "positional_embedding/wpe"
            self.bucket_ends.append(self.offset[nxt])
        self.bucket_ends.append(self.total)
        # offsets at which backward has finished a prefix of the flat gradient buffer, in completion order: the head's
        # kernel + bias (right after its weight-gradient GEMM, before the input gradient), each layer (the head LayerNorm's
        # gain / bias ride with layer L-1), finally the embeddings.  The exchange pieces follow these cuts (src/dp.py).
        self.ready_points: List[int] = [self.offset["to_logits/layer_norm/g"]] + self.bucket_ends[1:]

    def numel(self, name):
        return int(np.prod(self.shape[name]))


class DalleEngine:
    def __init__(self, n_embd, n_layers, n_heads, text_vocab_size, image_vocab_size, text_seq_len, image_seq_len,
                 batch_size, global_batch_size=None, eos_token_id=None, hparams: Optional[dict] = None,
                 device="cuda", process_group=None, world_size=1, comm=None):
        if not torch.cuda.is_available():
            raise dh.DalleHipError("DalleEngine needs a HIP device (MI355X); there is no CPU fallback")
        dh.lib()
        assert n_embd % n_heads == 0, "n_state must be divisible by n_heads"
        if n_embd // n_heads != HEAD_DIM:
            raise dh.DalleHipError(f"attention kernels are built for head dim {HEAD_DIM} (n_embd/n_heads = {n_embd // n_heads}); "
                                   "the reference README recommends exactly this ratio")
        self.d, self.L, self.H = n_embd, n_layers, n_heads
        self.text_vocab_size, self.image_vocab_size = text_vocab_size, image_vocab_size
        self.T, self.S = text_seq_len, text_seq_len + image_seq_len
        assert self.S % 8 == 0, "total sequence length must be a multiple of 8"
        self.V = text_vocab_size + image_vocab_size + 1
        self.eos = self.V - 1 if eos_token_id is None else eos_token_id
        self.B = batch_size
        self.B_global = global_batch_size or batch_size * world_size
        self.M = self.B * self.S
        self.dev = torch.device(device)
        self.pg, self.world = process_group, world_size
        self.hp = dict(hparams or {})
        self.lay = ParamLayout(n_embd, n_layers, n_heads, self.V, self.S)
        self.Vp = self.lay.Vp
        n = self.lay.total
        f32 = dict(dtype=torch.float32, device=self.dev)
        b16 = dict(dtype=torch.bfloat16, device=self.dev)
        self.p = torch.zeros(n, **f32)
        self.g = torch.zeros(n, **f32)
        self.m = torch.zeros(n, **f32)
        self.v = torch.zeros(n, **f32)
        self.pb = torch.zeros(n, **b16)
        self.pbt = torch.zeros(self.lay.t_total, **b16)
        self.global_step = 0
        # The exchange's RCCL channels run beside the BACKWARD: there the persistent kernels leave CUs for them (a block of a
        # one-block-per-CU kernel whose CU an intruder holds starts when the others have finished: the launch takes twice as long --
        # measured with the token sort as the intruder).  16 CUs cost 3.7 % of a single-GPU step when applied to
        # the whole step, so the option is set for the backward only; 0 = off.  Unmeasured on a multi-GPU node, hence OFF by
        # default (it also hands the full-row products back to the 128x128 kernel and turns the fused LayerNorm forms off):
        # hparams["dp_reserve_cus"], DALLE_DP_RESERVE_CUS or bench.py --reserve-cus select it for the first multi-GPU A/B.
        self.dp_reserve_cus = int(self.hp.get("dp_reserve_cus", os.environ.get("DALLE_DP_RESERVE_CUS", "0"))) if world_size > 1 else 0
        self._alloc_activations()
        # gradient exchange: RCCL behind the C ABI when `comm` (dp.init_comm) is given, torch.distributed otherwise
        self.reducer = GradReducer(self.g, world_size, comm=comm, pg=process_group)

    # ------------------------------------------------------------------ parameter access
    def view(self, buf, name):
        o = self.lay.offset[name]
        return buf[o:o + self.lay.numel(name)].view(self.lay.shape[name])

    def tview(self, name):
        o = self.lay.t_offset[name]
        r, c = self.lay.shape[name]
        return self.pbt[o:o + r * c].view(c, r)

    def load_reference_params(self, P: Dict[str, np.ndarray]):
        """Load weights given under the reference's variable names/shapes."""
        d, V = self.d, self.V
        with torch.no_grad():
            for name, shp in self.lay.entries:
                dst = self.view(self.p, name)
                if name.endswith("attn/qkv"):
                    base = name[:-3]
                    cat = np.concatenate([P[base + "q"], P[base + "k"], P[base + "v"]], axis=1)
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(cat)))
                elif name == "to_logits/linear_out/kernel":
                    dst.zero_()
                    dst[:, :V].copy_(torch.from_numpy(np.ascontiguousarray(P[name])))
                elif name == "to_logits/linear_out/bias":
                    dst.fill_(-30000.0)   # pad logits can never win the softmax (and are masked in the CE kernel)
                    dst[:V].copy_(torch.from_numpy(np.ascontiguousarray(P[name])))
                else:
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(P[name])).view(shp))
        self.refresh_compute_copies(cast=True)

    def export_reference(self, buf=None) -> "OrderedDict[str, np.ndarray]":
        """Inverse of load_reference_params for any flat buffer (params, grads, m, v)."""
        buf = self.p if buf is None else buf
        out: "OrderedDict[str, np.ndarray]" = OrderedDict()
        d, V = self.d, self.V
        for name, shp in self.lay.entries:
            a = self.view(buf, name).detach().float().cpu().numpy()
            if name.endswith("attn/qkv"):
                base = name[:-3]
                out[base + "q"], out[base + "k"], out[base + "v"] = (np.ascontiguousarray(a[:, i * d:(i + 1) * d]) for i in range(3))
            elif name == "to_logits/linear_out/kernel":
                out[name] = np.ascontiguousarray(a[:, :V])
            elif name == "to_logits/linear_out/bias":
                out[name] = np.ascontiguousarray(a[:V])
            else:
                out[name] = a
        return out

    def init_params(self, seed=1234):
        """Reference initialisers drawn with torch's generator on the host."""
        g = torch.Generator().manual_seed(seed)
        d, L, H = self.d, self.L, self.H
        k = d // H
        P = OrderedDict()

        def nrm(shape, std):
            return (torch.randn(*shape, generator=g) * std).numpy()
        P["embedding/wte"] = nrm((self.V, d), 0.02)
        P["positional_embedding/wpe"] = nrm((self.S, d), 0.01)
        for i in range(L):
            p = f"layer_{i}/"
            P[p + "norm_1/g"], P[p + "norm_1/b"] = np.ones(d, np.float32), np.zeros(d, np.float32)
            P[p + "attn/q"] = nrm((d, d), (d * k) ** -0.5)
            P[p + "attn/k"] = nrm((d, d), d ** -0.5)
            P[p + "attn/v"] = nrm((d, d), d ** -0.5)
            P[p + "attn/o"] = nrm((d, d), (H * k) ** -0.5)
            P[p + "attn/compute_output_bias/o_b"] = np.zeros(d, np.float32)
            P[p + "norm_2/g"], P[p + "norm_2/b"] = np.ones(d, np.float32), np.zeros(d, np.float32)
            P[p + "mlp/mlp_linear_1/kernel"], P[p + "mlp/mlp_linear_1/bias"] = nrm((d, 4 * d), 0.02), np.zeros(4 * d, np.float32)
            P[p + "mlp/mlp_linear_2/kernel"], P[p + "mlp/mlp_linear_2/bias"] = nrm((4 * d, d), 0.02 / math.sqrt(L)), np.zeros(d, np.float32)
        P["to_logits/layer_norm/g"], P["to_logits/layer_norm/b"] = np.ones(d, np.float32), np.zeros(d, np.float32)
        P["to_logits/linear_out/kernel"], P["to_logits/linear_out/bias"] = nrm((d, self.V), 0.02), np.zeros(self.V, np.float32)
        self.load_reference_params(P)

    def refresh_compute_copies(self, cast=False):
        """bf16 natural copy (if not already written by the Adam kernel) + [out,in] copies for the fwd GEMMs."""
        if cast:
            dh.cast_f32_bf16(self.p, self.pb, self.lay.total)
        if getattr(self, "_t_table", None) is None:   # one launch for all [in,out] -> [out,in] weight copies
            rows, tile = [], 0
            for name in self.lay.t_offset:
                r, c = self.lay.shape[name]
                rows.append([self.lay.offset[name], self.lay.t_offset[name], r, c, tile])
                tile += ((r + 63) // 64) * ((c + 63) // 64)
            self._t_table = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._t_tiles = tile
        dh.transpose_batch(self.pb, self.pbt, self._t_table, self._t_table.shape[0], self._t_tiles)

    # ------------------------------------------------------------------ buffers
    def _alloc_activations(self):
        M, d, L, B, H, S, Vp = self.M, self.d, self.L, self.B, self.H, self.S, self.Vp
        b16 = dict(dtype=torch.bfloat16, device=self.dev)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.tokens = torch.zeros(B, S, dtype=torch.int32, device=self.dev)
        self.labels = torch.zeros(B, S, dtype=torch.int32, device=self.dev)
        self.X = [torch.empty(M, d, **b16) for _ in range(L + 1)]       # residual stream entering layer l
        # hparams["recompute_grad"] (the reference wraps every block in mtf.recompute_grad, src/dalle_mtf/models.py:342-343):
        # only the residual stream X[l] is kept per layer; the block's inner activations live in ONE shared set of buffers
        # and backward() re-runs the block's forward (bit-identical kernels) before differentiating it: 0.63 GB -> 0.1 GB per
        # layer at B=32, S=1280, for one extra block forward (~+30 % of the step's flops).
        self.recompute = bool(self.hp.get("recompute_grad", False))
        nl = 1 if self.recompute else L

        def per_layer(make):
            bufs = [make() for _ in range(nl)]
            return [bufs[l % nl] for l in range(L)]
        self.xn1 = per_layer(lambda: torch.empty(M, d, **b16))
        self.qkv = per_layer(lambda: torch.empty(M, 3 * d, **b16))
        self.o = per_layer(lambda: torch.empty(M, d, **b16))
        self.lse = per_layer(lambda: torch.empty(B, H, S, **f32))
        self.x1 = per_layer(lambda: torch.empty(M, d, **b16))
        self.xn2 = per_layer(lambda: torch.empty(M, d, **b16))
        self.h = per_layer(lambda: torch.empty(M, 4 * d, **b16))
        # [r05] the ReLU mask of the FFN as bits: FFN-1's epilogue emits them, the FFN-2 input gradient reads M * 4d / 8 bytes instead
        # of the whole h (168 MB per layer at dalle_example) -- where the library runs both products on the kernel that has the bit
        # forms (dmi_relu_bits_auto); bit-identical to the relu_src form (tested)
        self.use_relu_bits = bool(self.hp.get("relu_bits", True)) and dh.relu_bits_auto(M, 4 * d, d)
        self.hbits = per_layer(lambda: torch.empty(dh.relu_bits_bytes(M, 4 * d), dtype=torch.uint8, device=self.dev)) if self.use_relu_bits else None
        self.stats = per_layer(lambda: [torch.empty(M, **f32) for _ in range(4)])  # mean1, rstd1, mean2, rstd2
        self.xnf = torch.empty(M, d, **b16)
        self.statf = [torch.empty(M, **f32) for _ in range(2)]
        self.z = torch.empty(M, Vp, **b16)      # eval: logits; train: E = exp(logit), patched into unnormalised dlogits
        self.loss_rows = torch.empty(M, **f32)
        self.loss = torch.zeros(1, **f32)
        self.gnorm_sq = torch.zeros(1, **f32)
        # fused softmax head (training path, include/dalle_hip.h K7/K8 (b))
        self.nparts = dh.gemm_nt_softmax_partials(Vp)
        self.zl = torch.empty(M, **f32)                      # label logit (loss = logsumexp - label logit)
        self.rowsum_part = torch.empty(self.nparts, M, **f32)
        self.rowscale = torch.empty(M, **f32)                # dz_scale / sum_v exp(.)
        self.rowscale_bf = torch.empty(_round_up(M, 128) + 128, **b16)
        self.xs = torch.empty(M, d, **b16)                   # rowscale * LN_f(x): left operand of the head weight gradient
        self.head_flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # token-id order for the embedding scatter-add: sorted on a side stream while the forward runs
        self.tok_sorted = torch.empty(M, dtype=torch.int32, device=self.dev)
        self.tok_perm = torch.empty(M, dtype=torch.int32, device=self.dev)
        self.sort_ws = torch.empty(dh.sort_tokens_workspace_bytes(M), dtype=torch.uint8, device=self.dev)
        self.embed_ws = torch.empty(dh.embed_bwd_workspace_bytes(B, S, d), dtype=torch.uint8, device=self.dev)
        self.sort_stream = torch.cuda.Stream(device=self.dev)
        self._sort_done = None
        # (launching the sort at the start of the BACKWARD instead -- side stream under the head's weight gradient, or on the main stream -- measured
        # the same within noise: 14.49 / 14.57 / 14.57 ms over three alternations, profiles/r06_ab_sort_at.log)
        # scratch
        self.dx = [torch.empty(M, d, **b16) for _ in range(2)]
        self.dxn = torch.empty(M, d, **b16)
        self.dh = torch.empty(M, 4 * d, **b16)
        self.dqkv = torch.empty(M, 3 * d, **b16)
        self.d_o = torch.empty(M, d, **b16)
        self.delta = torch.empty(3, B, H, S, **f32)   # delta | (lse, delta) pairs for the dK/dV kernel's DMA
        wsz = max(dh.gemm_tn_workspace_bytes(M, d, Vp), dh.gemm_tn_workspace_bytes(M, 4 * d, d),
                  dh.gemm_tn_workspace_bytes(M, d, 4 * d), dh.gemm_tn_workspace_bytes(M, d, 3 * d),
                  dh.gemm_tn_workspace_bytes(M, d, d), dh.layernorm_bwd_workspace_bytes(M, d),
                  dh.sumsq_workspace_bytes(self.lay.total), dh.gemm_nt_splitk_workspace_bytes(M // 2 + 256, d, 8))
        self.ws = torch.empty(int(wsz) + 1024, dtype=torch.uint8, device=self.dev)
        # the four weight gradients of a block keep their split-m slabs in separate workspaces and their seven slab reduces
        # run as ONE launch at the end of the block's backward (dmi_reduce_slabs_batch)
        self.ws_blk = [torch.empty(int(dh.gemm_tn_workspace_bytes(M, i_, j_)) + 256, dtype=torch.uint8, device=self.dev)
                       for i_, j_ in ((4 * d, d), (d, 4 * d), (d, d), (d, 3 * d))]
        self.deferred = dh.DeferredReduces()
        # tuning switch: hparams win, the environment variable gives the default (A/B runs: tools/ab_env.sh)
        # LayerNorm fused into the products that feed it (dmi_gemm_nt_ln): the full-row tiles exist for n_embd = 512
        # Measured (profiles/r04q_kbench_ln512.log, r04q_ab_fuse_ln.log): out-projection + norm_2 46.5 us fused vs 33.4 + 16.1 us,
        # FFN-2 + norm 98.9 vs 77.9 + 16.4 us, step 15.98 vs 15.95 ms -- with ONE tile per CU nothing overlaps the fused epilogue,
        # and its two 160-KB outputs per tile leave at the CU's ~14 B / clk store-issue rate, which costs what the HBM-bound
        # standalone kernel costs.  (Round 4: off by default.)
        # [r05] ON by default: with the residual rows fetched as 16-byte pieces (through the row swap) instead of 8-byte pieces in the
        # accumulator layout the fused form wins: 14.92 -> 14.83 ms/step same-call (profiles/r05g_ab_fuse_ln.log)
        # [r06] both fused LayerNorm forms are gated on the library's own predicate (dmi_gemm_nt_ln_auto: N = 512, operand sizes inside the
        # 32-bit buffer offsets of the full-row kernel for the widest product that uses it, no CUs reserved for a concurrent exchange) at
        # buffer-allocation time; the two-kernel path is the fallback (advisor finding, round 5: the fused calls have no fallback of their own)
        ln_ok = dh.gemm_nt_ln_auto(M, d, 4 * d) and self.dp_reserve_cus == 0
        self.fuse_ln = bool(self.hp.get("fuse_ln", os.environ.get("DALLE_FUSE_LN", "1") != "0")) and ln_ok
        # FFN-2 -> next norm_1: not under recompute_grad, whose re-run of a block starts from the stored residual stream with a
        # standalone norm_1 (its statistics sum in another order; the re-run must reproduce the forward bit for bit)
        self.fuse_ln1 = self.fuse_ln and not self.recompute
        self.hp.setdefault("dgrad_tail_split", os.environ.get("DALLE_DGRAD_TAIL", "1") != "0")
        self.hp.setdefault("defer_reduces", os.environ.get("DALLE_DEFER_REDUCES", "1") != "0")
        self.hp.setdefault("wgrad_pair", os.environ.get("DALLE_WGRAD_PAIR", "1") != "0")
        # [r06] all FOUR weight gradients of a block in one launch, on 128 x 256 tiles (96 tiles x 5 row splits; the library picks the wide
        # tile when the union fills the block slots within every problem's slab count): 335 -> 266 us per block in isolation
        # (profiles/r06_tn_group_wide.log); needs the pair launch and the deferred reduces
        self.hp.setdefault("wgrad_group4", os.environ.get("DALLE_WGRAD_GROUP4", "1") != "0")
        # (only where the library's plan for the four is the wide one: at n_embd = 1024 / 2048 the union is 384+ tiles, which whole
        # 128 x 128 tiles of ONE launch would run in ragged residencies -- those shapes keep the separate launches)
        self.wgrad_group4 = bool(self.hp["wgrad_group4"] and self.hp["wgrad_pair"] and self.hp["defer_reduces"]
                                 and dh.gemm_tn_group_plan([(4 * d, d), (d, 4 * d), (d, d), (d, 3 * d)], M) > 0)
        self.hp.setdefault("lnbwd_chain", os.environ.get("DALLE_LNBWD_CHAIN", "1") != "0")
        # [r05] LayerNorm backward fused into the two input-gradient products that feed a LayerNorm (dmi_gemm_nt_lnbwd: n_embd = 512,
        # full-row tiles): dxn is never written, 12 of the 13 ln_bwd launches of a dalle_example step disappear
        self.fuse_lnbwd = bool(self.hp.get("fuse_lnbwd", os.environ.get("DALLE_FUSE_LNBWD", "1") != "0")) and ln_ok
        if self.fuse_lnbwd:
            # one partial buffer per LayerNorm: the 2L gain / bias reduces run as ONE batched launch at the end of the backward
            # (per block under data parallelism, where the exchange takes a block's gradients as soon as it is done)
            self.lnb_batch = bool(self.hp.get("lnbwd_batch_finish", os.environ.get("DALLE_LNBWD_BATCH", "1") != "0"))
            npart = dh.gemm_nt_lnbwd_parts(M) * 2 * d
            self.lnb_part = [torch.empty(npart, dtype=torch.float32, device=self.dev) for _ in range(2 * L if self.lnb_batch else 1)]
            self.ln_ws_final = torch.empty(int(dh.layernorm_bwd_workspace_bytes(M, d)) + 256, dtype=torch.uint8, device=self.dev)

    # ------------------------------------------------------------------ forward
    def _w(self, name):
        return self.view(self.pb, name)

    def forward(self, tokens: torch.Tensor, need_grad=True) -> torch.Tensor:
        """tokens int32 [B,S] on device.  Returns the device scalar loss = mean over ALL B*S positions of
        -log softmax(logits)[label] (src/dalle_mtf/models.py:348-359), labels = shift(tokens) (:407-410).
        need_grad=True (training): the softmax is fused into the vocabulary projection -- self.z then holds the
        unnormalised dlogits E, self.rowscale their per-row factor (already scaled by 1/(global B*S*microbatches)).
        need_grad=False (evaluation): self.z holds the bf16 logits (see logits()); the loss is the plain mean
        (the reference forces num_microbatches = 1 outside training, src/model_fns.py:150-154)."""
        M, d, L, B, H, S, Vp = self.M, self.d, self.L, self.B, self.H, self.S, self.Vp
        assert tokens.shape == (B, S) and tokens.dtype == torch.int32
        self.tokens.copy_(tokens)
        self._sort_done = None
        if need_grad:   # token-id order for the embedding backward: twelve 3-5 us launches on a side stream (see dmi_sort_tokens)
            self._launch_sort()
        dh.shift_labels(self.tokens, self.labels, B, S, self.eos)
        dh.embed_fwd(self.tokens, self._w("embedding/wte"), self._w("positional_embedding/wpe"), self.X[0], S, d, self.V)
        for l in range(L):
            self._block_forward(l)
        if not (self.fuse_ln1 and L > 0):      # (fused: written by the last block's FFN-2)
            dh.layernorm_fwd(self.X[L], self._w("to_logits/layer_norm/g"), self._w("to_logits/layer_norm/b"), self.xnf,
                             self.statf[0], self.statf[1], M, d)
        Wt, bias = self.tview("to_logits/linear_out/kernel"), self._w("to_logits/linear_out/bias")
        nmb = (self.hp.get("num_microbatches", 1) or 1) if need_grad else 1
        if need_grad:
            dh.label_logit(self.xnf, d, Wt, d, bias, self.labels, self.zl, self.head_flag, M, d, self.V)
        hook = getattr(self, "event_hook", None)  # bench.py: HIP events around the largest single launch
        if hook is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if need_grad:
            dh.gemm_nt_softmax(self.xnf, d, Wt, d, bias, None, self.z, Vp, self.rowsum_part, M, Vp, d)   # no exponent shift
        else:
            dh.gemm_nt(self.xnf, d, Wt, d, self.z, Vp, M, Vp, d, dh.GEMM_BIAS, bias=bias)
        if hook is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            hook().append((e0, e1))
        if need_grad:
            dh.softmax_finish(self.rowsum_part, self.nparts, self.zl, None, self.labels, self.xnf, d, Wt, d, bias, self.z, Vp, Vp,
                              self.loss_rows, self.rowscale, self.rowscale_bf, self.xs, self.head_flag, M, d, self.V,
                              1.0 / (self.B_global * S * nmb))
        else:
            dh.cross_entropy(self.z, Vp, self.labels, self.loss_rows, None, M, self.V, 0.0)
        dh.sum_f32(self.loss_rows, M, 1.0 / (M * nmb), self.loss)
        return self.loss

    def _block_forward(self, l):
        """one transformer block (src/dalle_mtf/models.py:326-335): X[l] -> X[l+1]; also what backward() re-runs under
        recompute_grad.
        fuse_ln (n_embd = 512): the two products that end in the residual stream run on full-row tiles and emit the LayerNorm
        that follows them in the same pass (dmi_gemm_nt_ln) -- out-projection + residual -> norm_2, FFN-2 + residual -> the
        NEXT block's norm_1 (or to_logits' norm): 12 of the 13 standalone LayerNorm launches of a forward pass disappear."""
        M, d, B, H, S, L = self.M, self.d, self.B, self.H, self.S, self.L
        p = f"layer_{l}/"
        x = self.X[l]
        st = self.stats[l]
        rerun = getattr(self, "_in_backward", False)
        if not (self.fuse_ln1 and l > 0):   # (fused: written by block l-1's FFN-2)
            dh.layernorm_fwd(x, self._w(p + "norm_1/g"), self._w(p + "norm_1/b"), self.xn1[l], st[0], st[1], M, d)
        dh.gemm_nt(self.xn1[l], d, self.tview(p + "attn/qkv"), d, self.qkv[l], 3 * d, M, 3 * d, d)
        dh.attention_fwd(self.qkv[l], self.o[l], self.lse[l], B, H, S)   # no transposed copies: hardware transpose reads
        if self.fuse_ln:
            dh.gemm_nt_ln(self.o[l], d, self.tview(p + "attn/o"), d, self.x1[l], d, M, d, d,
                          self._w(p + "norm_2/g"), self._w(p + "norm_2/b"), self.xn2[l], d, st[2], st[3],
                          bias=self._w(p + "attn/compute_output_bias/o_b"), residual=x)
        else:
            dh.gemm_nt(self.o[l], d, self.tview(p + "attn/o"), d, self.x1[l], d, M, d, d, dh.GEMM_BIAS | dh.GEMM_RESIDUAL,
                       bias=self._w(p + "attn/compute_output_bias/o_b"), residual=x)
            dh.layernorm_fwd(self.x1[l], self._w(p + "norm_2/g"), self._w(p + "norm_2/b"), self.xn2[l], st[2], st[3], M, d)
        if self.use_relu_bits:
            dh.gemm_nt_relu_bits(self.xn2[l], d, self.tview(p + "mlp/mlp_linear_1/kernel"), d, self.h[l], 4 * d, M, 4 * d, d,
                                 self._w(p + "mlp/mlp_linear_1/bias"), self.hbits[l])
        else:
            dh.gemm_nt(self.xn2[l], d, self.tview(p + "mlp/mlp_linear_1/kernel"), d, self.h[l], 4 * d, M, 4 * d, d,
                       dh.GEMM_BIAS | dh.GEMM_RELU, bias=self._w(p + "mlp/mlp_linear_1/bias"))
        if rerun:   # the re-run stops here: X[l+1] is already stored
            return
        W2, b2 = self.tview(p + "mlp/mlp_linear_2/kernel"), self._w(p + "mlp/mlp_linear_2/bias")
        if self.fuse_ln1:
            if l + 1 < L:
                q = f"layer_{l + 1}/norm_1/"
                y, mean, rstd = self.xn1[l + 1], self.stats[l + 1][0], self.stats[l + 1][1]
            else:
                q = "to_logits/layer_norm/"
                y, mean, rstd = self.xnf, self.statf[0], self.statf[1]
            dh.gemm_nt_ln(self.h[l], 4 * d, W2, 4 * d, self.X[l + 1], d, M, d, 4 * d, self._w(q + "g"), self._w(q + "b"), y, d, mean, rstd,
                          bias=b2, residual=self.x1[l])
        else:
            dh.gemm_nt(self.h[l], 4 * d, W2, 4 * d, self.X[l + 1], d, M, d, 4 * d, dh.GEMM_BIAS | dh.GEMM_RESIDUAL, bias=b2, residual=self.x1[l])

    def logits(self) -> torch.Tensor:
        """fp32 logits [B,S,V] of the last forward(need_grad=False) ("go to full precision", models.py:395)."""
        # (after a training forward self.z holds unnormalised dlogits, not logits)
        return self.z.view(self.B, self.S, self.Vp)[:, :, :self.V].float()

    # ------------------------------------------------------------------ sampling
    def sample_image_tokens(self, text: torch.Tensor, temperature: float = 1.0, top_k: int = 0, seed: int = 0,
                            kv_cache: bool = True, decode_graph: bool = True, fused_sampling: bool = True) -> torch.Tensor:
        """Autoregressive image-token sampling: text int32 [B, T] -> image-token ids [B, P] in [0, image_vocab_size).
        The reference scaffolds this (is_incremental_inference, models.py:246-254,281-285) but its predict path raises
        NotImplementedError (model_fns.py:135-136).  Logits are restricted to the image vocabulary; temperature / top-k /
        greedy (temperature 0).

        kv_cache=True (default): ONE full forward over the text prefix fills the per-layer key/value cache (the [B*S, 3d]
        projection buffers of the forward pass), then every further token is one incremental step over B rows --
        decode_step(): QKV GEMM writing row `pos` of the cache in place, dmi_attention_decode (one query against keys
        0..pos), out-projection, MLP -- ~90 launches of a few microseconds instead of a 1280-position forward, replayed as
        one HIP graph (decode_graph; see decode_step).
        fused_sampling (with kv_cache and decode_graph): the draw itself is a kernel at the end of the replayed graph
        (dmi_sample_tokens: temperature / top-k / Gumbel-max categorical draw, noise a pure function of (seed, position, row))
        that writes the chosen token where the next step's embedding reads it -- P graph replays back to back, no host round
        trip per position.  fused_sampling=False launches the same draw kernel from the host after each decode step (same
        tokens for the same seed: the noise is a pure function of (seed, position, row, index)).
        kv_cache=False: the plain form, one full evaluation forward per generated position (the causal mask makes the
        not-yet-generated tail irrelevant); kept as the cross-check the cached path is tested against."""
        B, T, S, P = self.B, self.T, self.S, self.S - self.T
        assert text.shape == (B, T)
        lo, hi = self.text_vocab_size, self.text_vocab_size + self.image_vocab_size
        toks = torch.full((B, S), lo, dtype=torch.int32, device=self.dev)
        toks[:, :T] = text.to(device=self.dev, dtype=torch.int32)
        if kv_cache and self.recompute:
            kv_cache = False     # recompute_grad keeps ONE shared set of projection buffers: there is no per-layer cache to decode from
        if kv_cache and decode_graph and fused_sampling and self.image_vocab_size <= 8192:
            self.forward(toks, need_grad=False)          # prefill: k, v of the text positions are in the cache
            D = self._decode_state()
            D["tok"].copy_(toks[:, T - 1])
            D["pos_i"][0:1].fill_(T - 1)
            inv_t = np.array([1.0 / temperature if temperature > 0 else 0.0], dtype=np.float32).view(np.uint32)[0]
            prm = np.array([inv_t, int(top_k), seed & 0xffffffff, (seed >> 32) & 0xffffffff], dtype=np.uint32).view(np.int32)
            D["params"].copy_(torch.from_numpy(prm))
            for _ in range(P):                           # position T-1+i predicts image token i; the graph advances the position itself
                self._run_decode(sample=True, graph=True)
            return D["out"].clone()

        nv = hi - lo
        if nv > 8192:
            raise dh.DalleHipError(f"sample_image_tokens: the draw kernel (dmi_sample_tokens) handles image vocabularies up to 8192 (got {nv})")
        bias = self._w("to_logits/linear_out/bias")[lo:hi]
        nxt = torch.empty(B, dtype=torch.int32, device=self.dev)

        def pick(z, ldz, zbias, position):
            """the draw itself is dmi_sample_tokens on every path (temperature / top-k / Gumbel-max with counter-based noise
            hash(seed, position, row, index), first maximum when temperature <= 0): host-launched here, the last node of the
            replayed graph on the fused path -- the same (seed, position) gives the same draw on both."""
            dh.sample_tokens(z, ldz, zbias, B, nv, temperature=temperature, top_k=top_k, seed=seed, pos=position,
                             token_offset=lo, next_tok=nxt)
            return nxt

        for pos in range(P):
            if not kv_cache:
                self.forward(toks, need_grad=False)
                # the position before predicts token T + pos; the evaluation head already carries the bias (GEMM epilogue)
                z = self.z.view(B, S, self.Vp)[:, T + pos - 1, lo:hi]
                tok = pick(z, S * self.Vp, None, T + pos - 1)
            else:
                if pos == 0:
                    self.forward(toks, need_grad=False)    # prefill: leaves k, v of the text positions in the cache
                # (position T - 1 is decoded again rather than read from the prefill's logits: every cached path then takes
                # every token through the same arithmetic)
                self.decode_step(toks[:, T + pos - 1].contiguous(), T + pos - 1, graph=decode_graph)
                tok = pick(self._dec["z"], nv, bias, T + pos - 1)      # bf16 head output + bias, as the fused path draws
            toks[:, T + pos] = tok
        return (toks[:, T:] - lo).contiguous()

    def _decode_state(self):
        if getattr(self, "_dec", None) is None:
            B, d = self.B, self.d
            b16 = dict(dtype=torch.bfloat16, device=self.dev)
            f32 = dict(dtype=torch.float32, device=self.dev)
            i32 = dict(dtype=torch.int32, device=self.dev)
            self._dec = dict(x=[torch.empty(B, d, **b16) for _ in range(2)], xn=torch.empty(B, d, **b16), o=torch.empty(B, d, **b16),
                             h=torch.empty(B, 4 * d, **b16), st=[torch.empty(B, **f32) for _ in range(2)],
                             z=torch.empty(B, self.image_vocab_size, **b16), fresh=torch.empty(B, 3 * d, **b16),
                             tok=torch.empty(B, **i32), pos_i=torch.zeros(2, **i32),    # [position, scratch counter of the sampler]
                             logits=torch.empty(B, self.image_vocab_size, **f32),
                             params=torch.zeros(4, **i32), out=torch.zeros(B, self.S - self.T, **i32),
                             graphs={}, warm=set())
        return self._dec

    def decode_step(self, tokens_at_pos: torch.Tensor, pos: int, graph: bool = True) -> torch.Tensor:
        """Incremental inference (reference hooks src/dalle_mtf/models.py:246-254,281-285): the hidden state of sequence
        position `pos` alone, given the tokens int32 [B] at that position and the key/value cache of positions < pos left by
        forward() / earlier decode steps in self.qkv[l].  Returns fp32 logits over the IMAGE vocabulary [B, image_vocab_size]
        (what predicts the token at pos + 1; the buffer is reused by the next call).

        graph=True: the ~90 launches of a step are a few microseconds of GPU work each, so the step is launch-bound when
        driven from the host; it is captured ONCE as a HIP graph and replayed for every position.  Nothing position-dependent
        is a by-value kernel argument: the position lives in device memory (pos_dev of dmi_embed_fwd -- the positional-embedding
        row -- and of dmi_attention_decode), and the QKV GEMM writes a fixed staging buffer that the attention kernel
        moves into cache row pos.  graph=False runs the same launches eagerly (the cross-check)."""
        B, S = self.B, self.S
        assert tokens_at_pos.shape == (B,) and tokens_at_pos.dtype == torch.int32 and 0 <= pos < S
        D = self._decode_state()
        D["tok"].copy_(tokens_at_pos)
        D["pos_i"][0:1].fill_(pos)
        self._run_decode(sample=False, graph=graph)
        return D["logits"]

    def _run_decode(self, sample: bool, graph: bool):
        D = self._dec
        if not graph:
            self._decode_body(sample)
        elif sample not in D["graphs"] and sample not in D["warm"]:
            self._decode_body(sample)      # first step eager: lazily created views / copies come into being outside the capture
            D["warm"].add(sample)
        else:
            if sample not in D["graphs"]:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other host threads (input producer) stay free to call HIP
                    self._decode_body(sample)
                D["graphs"][sample] = g    # (capture records, it does not execute: the replay below is the step)
            D["graphs"][sample].replay()

    def _decode_body(self, sample: bool = False):
        """the launches of one decode step; reads D[tok] and the position D[pos_i][0] from device memory.  sample=False: writes
        D[logits].  sample=True: draws the next token (settings in D[params]) into D[tok] and column pos - (T - 1) of
        D[out], then advances the position (inside the sampling kernel).
        B <= 32: LayerNorm rides in the prologue of the product that consumes it (dmi_ln_gemm_nt) -- 5 dependent launches per
        block instead of 7; a dependent launch costs ~7 us on this part, more than any of these kernels' work."""
        B, d, L, H, S = self.B, self.d, self.L, self.H, self.S
        D = self._dec
        x, x1, xn, o, h, st, z, fresh = D["x"][0], D["x"][1], D["xn"], D["o"], D["h"], D["st"], D["z"], D["fresh"]
        fuse_ln = B <= 32 and d <= 2048 and self.image_vocab_size % 16 == 0 and self.hp.get("decode_fuse_ln", True)

        def ln_dense(inp, ln, W, out, N, flags=0, bias=None):       # out = LN(inp) . W^T (+ bias)(ReLU)
            g, b = self._w(ln + "/g"), self._w(ln + "/b")
            if fuse_ln:
                dh.ln_gemm_nt(inp, d, g, b, W, d, out, N, B, N, d, flags, bias=bias)
            else:
                dh.layernorm_fwd(inp, g, b, xn, st[0], st[1], B, d)
                dh.gemm_nt(xn, d, W, d, out, N, B, N, d, flags, bias=bias)

        dh.embed_fwd(D["tok"], self._w("embedding/wte"), self._w("positional_embedding/wpe"), x, S, d, self.V,
                     pos_dev=D["pos_i"])                                  # every row takes wpe[pos]
        for l in range(L):
            p = f"layer_{l}/"
            cache = self.qkv[l]                                        # [B*S, 3d]; row b*S + pos <- q | k | v of this step
            ln_dense(x, p + "norm_1", self.tview(p + "attn/qkv"), fresh, 3 * d)
            dh.attention_decode(cache, o, B, H, S, 0, fresh=fresh, pos_dev=D["pos_i"])
            dh.gemm_nt(o, d, self.tview(p + "attn/o"), d, x1, d, B, d, d, dh.GEMM_BIAS | dh.GEMM_RESIDUAL,
                       bias=self._w(p + "attn/compute_output_bias/o_b"), residual=x)
            ln_dense(x1, p + "norm_2", self.tview(p + "mlp/mlp_linear_1/kernel"), h, 4 * d, dh.GEMM_BIAS | dh.GEMM_RELU,
                     bias=self._w(p + "mlp/mlp_linear_1/bias"))
            dh.gemm_nt(h, 4 * d, self.tview(p + "mlp/mlp_linear_2/kernel"), 4 * d, x, d, B, d, 4 * d,
                       dh.GEMM_BIAS | dh.GEMM_RESIDUAL, bias=self._w(p + "mlp/mlp_linear_2/bias"), residual=x1)
        lo, nv = self.text_vocab_size, self.image_vocab_size
        Wt = self.tview("to_logits/linear_out/kernel")                 # [Vp, d]: rows lo .. lo + nv are the image vocabulary
        ln_dense(x, "to_logits/layer_norm", Wt[lo:lo + nv], z, nv)
        bias = self._w("to_logits/linear_out/bias")[lo:lo + nv]
        if sample:
            dh.sample_tokens(z, nv, bias, B, nv, params_dev=D["params"], pos_dev=D["pos_i"], advance=True, token_offset=lo,
                             next_tok=D["tok"], out=D["out"], out_col0=self.T - 1)
        else:
            dh.logits_f32(z, nv, bias, D["logits"], B, nv)       # "go to full precision for the logits" (models.py:394-395)

    # ------------------------------------------------------------------ backward
    def _gv(self, name):
        return self.view(self.g, name)

    def _wgrad(self, X, ldx, dY, ldy, dW, M, I, J, dbias=None, bias_weights=None, slot=None):
        """dW = X^T dY (+ fused bias gradient).  slot 0..3: one of the block's four gradients -- its slab reduces are deferred
        to the block's single reduce launch."""
        if slot is None or not self.hp["defer_reduces"]:
            dh.gemm_tn(X, ldx, dY, ldy, dW, M, I, J, self.ws, dbias=dbias, bias_weights=bias_weights)
        else:
            dh.gemm_tn(X, ldx, dY, ldy, dW, M, I, J, self.ws_blk[slot], dbias=dbias, bias_weights=bias_weights,
                       deferred=self.deferred)

    def _launch_sort(self):
        main = torch.cuda.current_stream()
        self.sort_stream.wait_stream(main)
        with torch.cuda.stream(self.sort_stream):
            dh.sort_tokens(self.tokens, self.tok_sorted, self.tok_perm, self.M, self.V, self.sort_ws)
            self._sort_done = torch.cuda.Event()
            self._sort_done.record(self.sort_stream)

    def backward(self, allreduce=True):
        """Gradients of the last forward(need_grad=True) into the flat fp32 buffer.  With world_size > 1 every finished
        prefix of the buffer is handed to the exchange (src/dp.py: SUM all-reduce in <= 64 MB pieces on the side stream) --
        the explicit form of mtf's implicit all-reduce over the `data` mesh axis (src/model_fns.py:81-82,189)."""
        reserve = self.dp_reserve_cus if allreduce else 0
        if not reserve:
            return self._backward(allreduce)
        dh.set_option("reserve_cus", reserve)     # (read at launch time: applies to the launches enqueued inside the try)
        try:
            return self._backward(allreduce)
        finally:                                  # the option is process-global: never leave it set for later forwards / other engines
            dh.set_option("reserve_cus", 0)

    def _backward(self, allreduce):
        M, d, L, B, H, S, Vp = self.M, self.d, self.L, self.B, self.H, self.S, self.Vp
        ws = self.ws
        E = self.z   # unnormalised dlogits: dlogits[m, :] = rowscale[m] * E[m, :]
        rp = self.lay.ready_points
        done = [0]

        def ready(upto):   # g[done, upto) is final on this stream: hand it to the exchange
            if allreduce:
                self.reducer.ready(done[0], upto)
            done[0] = upto

        if self._sort_done is None:
            self._launch_sort()
        # head: dW = (rowscale * xnf)^T E, dbias = rowscale^T E, dxn = rowscale * (E W^T)
        self._wgrad(self.xs, d, E, Vp, self._gv("to_logits/linear_out/kernel"), M, d, Vp,
                    dbias=self._gv("to_logits/linear_out/bias"), bias_weights=self.rowscale_bf)
        ready(rp[0])
        # K = vocabulary: main-loop-bound -> 256x256 tiles, one 8-wave block per CU (the library picks that kernel for long-K
        # launches that fill whole residencies of the 256 CUs).  The rows of the whole residencies run unsplit; the rows of the
        # ragged last residency run with K split so that they also fill the chip (fp32 slabs, deterministic reduce).
        # Measured per step (profiles/r02c_*): 1.46 + 0.47 ms vs 1.81 + 0.46 ms with 128x128 tiles.
        Wk = self._w("to_logits/linear_out/kernel")
        tn8 = (d + 255) // 256
        whole_rows = min(M, (((M + 255) // 256) * tn8 // 256) * 256 // tn8 * 256) if self.hp["dgrad_tail_split"] else M
        if whole_rows in (0, M) or Vp < 8192:
            dh.gemm_nt(E, Vp, Wk, Vp, self.dxn, d, M, d, Vp, dh.GEMM_ROWSCALE, rowscale=self.rowscale)
        else:
            tail_rows = M - whole_rows
            tail_tiles = ((tail_rows + 255) // 256) * tn8
            ns = next((k for k in (2, 4, 8) if (tail_tiles * k) % 256 == 0 and Vp // k >= 4096), 2)
            dh.gemm_nt(E, Vp, Wk, Vp, self.dxn, d, whole_rows, d, Vp, dh.GEMM_ROWSCALE, rowscale=self.rowscale)
            dh.gemm_nt_splitk(E[whole_rows:], Vp, Wk, Vp, self.dxn[whole_rows:], tail_rows, d, Vp, ns, self.ws,
                              rowscale=self.rowscale[whole_rows:])
        dxa, dxb = self.dx
        pend = []

        def ln_bwd(idx, dy, x, g, mean, rstd, dres, dx, dg, db):
            if self.fuse_lnbwd and self.lnb_batch and idx == 2 * L:    # the head's LayerNorm joins the batched finish of the fused ones
                dh.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, None, None, self.ln_ws_final, M, d)
                pend.append((self.ln_ws_final, dg, db, M))
            else:
                dh.layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, ws, M, d)

        def flush_ln():
            while pend:
                dh.layernorm_bwd_finish_batch(pend[:16], d)
                del pend[:16]

        def lnbwd(idx, A, K, Wn, x, g, mean, rstd, dres, dx, dg, db, B2=None, C2=None):
            """product + LayerNorm backward in one pass (dmi_gemm_nt_lnbwd); the gain / bias partials are summed right away or,
            batched, with the other LayerNorms' at the next flush_ln().  B2 / C2: the product that consumes dx, chained in the
            same launch (C2 = dx . B2^T)."""
            kw = dict(B2=B2, ldb2=d, C2=C2) if B2 is not None else {}
            if self.lnb_batch:
                part = self.lnb_part[idx]
                dh.gemm_nt_lnbwd(A, K, Wn, K, M, d, K, x, g, mean, rstd, dres, dx, part, **kw)
                # (finish_batch derives the number of partial rows from a row count: 32 rows per partial row)
                pend.append((part, dg, db, 32 * dh.gemm_nt_lnbwd_parts(M)))
            else:
                dh.gemm_nt_lnbwd(A, K, Wn, K, M, d, K, x, g, mean, rstd, dres, dx, self.lnb_part[0], dg=dg, db=db, **kw)

        ln_bwd(2 * L, self.dxn, self.X[L], self._w("to_logits/layer_norm/g"), self.statf[0], self.statf[1], None, dxa,
               self._gv("to_logits/layer_norm/g"), self._gv("to_logits/layer_norm/b"))
        for bi, l in enumerate(reversed(range(L))):
            p = f"layer_{l}/"
            st = self.stats[l]
            if self.recompute:
                self._in_backward = True
                self._block_forward(l)
                self._in_backward = False
            # FFN
            pair = self.hp["wgrad_pair"] and self.hp["defer_reduces"]
            group4 = pair and self.wgrad_group4    # (dxa, self.dh, dxb stay untouched until the group's launch behind the attention backward)
            if not group4:
                self._wgrad(self.h[l], 4 * d, dxa, d, self._gv(p + "mlp/mlp_linear_2/kernel"), M, 4 * d, d,
                            dbias=self._gv(p + "mlp/mlp_linear_2/bias"), slot=0)
            if self.use_relu_bits:
                dh.gemm_nt_mask_bits(dxa, d, self._w(p + "mlp/mlp_linear_2/kernel"), d, self.dh, 4 * d, M, 4 * d, d, self.hbits[l])
            else:
                dh.gemm_nt(dxa, d, self._w(p + "mlp/mlp_linear_2/kernel"), d, self.dh, 4 * d, M, 4 * d, d, dh.GEMM_RELU_MASK,
                           relu_src=self.h[l])
            if not group4:
                self._wgrad(self.xn2[l], d, self.dh, 4 * d, self._gv(p + "mlp/mlp_linear_1/kernel"), M, d, 4 * d,
                            dbias=self._gv(p + "mlp/mlp_linear_1/bias"), slot=1)
            chain = self.fuse_lnbwd and self.hp["lnbwd_chain"]     # ... and the out-projection's input gradient d_o = dxb . Wo^T in the same launch
            if self.fuse_lnbwd:
                lnbwd(2 * l + 1, self.dh, 4 * d, self._w(p + "mlp/mlp_linear_1/kernel"), self.x1[l], self._w(p + "norm_2/g"), st[2], st[3],
                      dxa, dxb, self._gv(p + "norm_2/g"), self._gv(p + "norm_2/b"),
                      B2=self._w(p + "attn/o") if chain else None, C2=self.d_o if chain else None)
            else:
                dh.gemm_nt(self.dh, 4 * d, self._w(p + "mlp/mlp_linear_1/kernel"), 4 * d, self.dxn, d, M, d, 4 * d)
                ln_bwd(2 * l + 1, self.dxn, self.x1[l], self._w(p + "norm_2/g"), st[2], st[3], dxa, dxb,
                       self._gv(p + "norm_2/g"), self._gv(p + "norm_2/b"))
            # attention
            if not pair:
                self._wgrad(self.o[l], d, dxb, d, self._gv(p + "attn/o"), M, d, d,
                            dbias=self._gv(p + "attn/compute_output_bias/o_b"), slot=2)
            if not chain:
                dh.gemm_nt(dxb, d, self._w(p + "attn/o"), d, self.d_o, d, M, d, d)
            dh.attention_bwd(self.qkv[l], self.o[l], self.d_o, self.lse[l], self.delta, self.dqkv, B, H, S)
            if pair:   # [r05] the out-projection and QKV kernels' gradients in ONE launch: 16 + 48 tiles fill the chip together
                probs = [dict(X=self.o[l], ldx=d, dY=dxb, ldy=d, dW=self._gv(p + "attn/o"), I=d, J=d, ws=self.ws_blk[2],
                              dbias=self._gv(p + "attn/compute_output_bias/o_b")),
                         dict(X=self.xn1[l], ldx=d, dY=self.dqkv, ldy=3 * d, dW=self._gv(p + "attn/qkv"), I=d, J=3 * d, ws=self.ws_blk[3])]
                if group4:   # [r06] ... and the two FFN gradients with them
                    probs = [dict(X=self.h[l], ldx=4 * d, dY=dxa, ldy=d, dW=self._gv(p + "mlp/mlp_linear_2/kernel"), I=4 * d, J=d,
                                  ws=self.ws_blk[0], dbias=self._gv(p + "mlp/mlp_linear_2/bias")),
                             dict(X=self.xn2[l], ldx=d, dY=self.dh, ldy=4 * d, dW=self._gv(p + "mlp/mlp_linear_1/kernel"), I=d, J=4 * d,
                                  ws=self.ws_blk[1], dbias=self._gv(p + "mlp/mlp_linear_1/bias"))] + probs
                dh.gemm_tn_group(probs, M, deferred=self.deferred)
            else:
                self._wgrad(self.xn1[l], d, self.dqkv, 3 * d, self._gv(p + "attn/qkv"), M, d, 3 * d, slot=3)
            if self.fuse_lnbwd:
                lnbwd(2 * l, self.dqkv, 3 * d, self._w(p + "attn/qkv"), self.X[l], self._w(p + "norm_1/g"), st[0], st[1],
                      dxb, dxa, self._gv(p + "norm_1/g"), self._gv(p + "norm_1/b"))
            else:
                dh.gemm_nt(self.dqkv, 3 * d, self._w(p + "attn/qkv"), 3 * d, self.dxn, d, M, d, 3 * d)
                ln_bwd(2 * l, self.dxn, self.X[l], self._w(p + "norm_1/g"), st[0], st[1], dxb, dxa,
                       self._gv(p + "norm_1/g"), self._gv(p + "norm_1/b"))
            self.deferred.run()        # the block's seven slab reduces in one launch
            if allreduce and self.world > 1:
                flush_ln()             # the exchange takes this block's gradients now
            ready(rp[1 + bi])
        # embeddings: positions visited in token-id order (sorted on the side stream during the forward)
        if self._sort_done is not None:
            torch.cuda.current_stream().wait_event(self._sort_done)
        dh.embed_bwd(self.tok_sorted, self.tok_perm, dxa, self._gv("embedding/wte"), self._gv("positional_embedding/wpe"),
                     B, S, d, self.V, self.embed_ws)
        flush_ln()
        ready(rp[L + 1])

    def wait_grads(self):
        self.reducer.finish()

    # ------------------------------------------------------------------ optimizer
    def learning_rate(self, step=None) -> float:
        """src/optimizers.py:46-76 (cosine/linear decay to 0.1*lr, linear warm-up)."""
        hp = self.hp
        step = self.global_step if step is None else step
        lr0 = hp["lr"]
        end = hp.get("lr_decay_end") or hp["train_steps"]
        decay = hp.get("lr_decay", "cosine") or "cosine"
        warm = hp.get("warmup_steps", 3000)
        warm = 3000 if warm is None else warm
        f32 = np.float32
        s = min(step, end)
        if decay == "linear":
            v = f32((lr0 - lr0 * 0.1) * (1.0 - s / end) + lr0 * 0.1)
        elif decay == "cosine":
            v = f32(lr0 * ((1.0 - 0.1) * 0.5 * (1.0 + math.cos(math.pi * s / end)) + 0.1))
        else:
            v = f32(lr0)
        if warm > 0 and step < warm:
            v = f32(v * f32(f32(step) / f32(warm)))
        return float(v)

    def optimizer_step(self):
        """clip_by_global_norm (src/optimizers.py:11-16) + AdamWeightDecayOptimizer without bias correction
        (src/optimizers.py:82-89,154-177) on the all-reduced gradients; refreshes the bf16 compute copies."""
        self.wait_grads()
        hp = self.hp
        n = self.lay.total
        clip = hp.get("gradient_clipping", 1.0)
        lr = self.learning_rate()
        b1 = hp.get("beta_1") or 0.9
        b2 = hp.get("beta_2") or 0.999
        eps = hp.get("epsilon") or 1e-6
        wd = hp.get("weight_decay") or 0.0
        gn = None
        cl = 0.0
        if clip is not None:
            dh.sumsq(self.g, n, self.gnorm_sq, self.ws)
            gn, cl = self.gnorm_sq, float(clip)
        if not wd:
            dh.adam_step(self.p, self.g, self.m, self.v, self.pb, n, gn, cl, lr, b1, b2, eps, 0.0)
        else:
            for name, _ in self.lay.entries:
                o, k = self.lay.offset[name], _round_up(self.lay.numel(name), ALIGN)
                use = ("norm" not in name) and ("bias" not in name) and not name.endswith("o_b")
                dh.adam_step(self.p[o:o + k], self.g[o:o + k], self.m[o:o + k], self.v[o:o + k], self.pb[o:o + k], k, gn, cl,
                             lr, b1, b2, eps, wd if use else 0.0)
        self.refresh_compute_copies(cast=False)
        self.global_step += 1
        return lr

    def train_step(self, tokens: torch.Tensor) -> torch.Tensor:
        """One optimizer step.  With hparams["num_microbatches"] = n > 1, `tokens` holds n micro-batches of B rows
        ([n*B, S]): gradients are accumulated locally and reduced once, and the loss is the sum of the micro-batch
        means / n (mtf.serialize_training_step as used at src/model_fns.py:156-166; src/dalle_mtf/models.py:356)."""
        nmb = self.hp.get("num_microbatches", 1) or 1
        if nmb == 1:
            loss = self.forward(tokens, need_grad=True)
            self.backward()
            self.optimizer_step()
            return loss
        assert tokens.shape == (nmb * self.B, self.S), f"expected {nmb} micro-batches of {self.B} rows"
        if getattr(self, "gacc", None) is None:
            self.gacc = torch.empty_like(self.g)
        if getattr(self, "loss_acc", None) is None:
            self.loss_acc = torch.zeros_like(self.loss)
        self.loss_acc.zero_()
        for i in range(nmb):
            loss = self.forward(tokens[i * self.B:(i + 1) * self.B], need_grad=True)
            self.loss_acc += loss
            self.backward(allreduce=False)
            if i == 0:
                self.gacc.copy_(self.g)
            else:
                dh.add_f32(self.gacc, self.g, self.lay.total)
        self.g.copy_(self.gacc)
        self.reducer.ready(0, self.lay.total)
        self.optimizer_step()
        return self.loss_acc

    def grad_norm(self) -> float:
        return float(torch.sqrt(self.gnorm_sq).item())

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        return {"p": self.p.detach().cpu(), "m": self.m.detach().cpu(), "v": self.v.detach().cpu(),
                "global_step": self.global_step}

    def load_state_dict(self, sd):
        self.p.copy_(sd["p"]); self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.global_step = int(sd["global_step"])
        self.refresh_compute_copies(cast=True)
