"""DiscreteVAE -- same constructor / forward surface as the reference (src/vae_tf/models.py:46-184), re-hosted on
libdalle_hip: every convolution is a tap list feeding the MFMA GEMMs -- gathered implicitly inside the GEMM
(dmi_conv_gemm_nt / dmi_conv_wgrad_tn) for 64-channel-aligned layers, through a materialised im2col (dmi_im2col +
dmi_gemm_nt / dmi_gemm_tn) otherwise; Gumbel-softmax and MSE are dedicated kernels (csrc/vae.hip).  bf16 activations/weights, fp32 accumulation, fp32
master weights and Adam state; the codebook logits are produced in fp32 (reference: fp32 matmul, models.py:113-118).

Layer structure (reference lines): encoder 81-120: per block a 4x4 s2 SAME conv (no activation) then (stack-1) x
`x + conv3x3(relu(conv3x3(x)))`; fp32 `x @ codebook`.  decoder 123-163: `y @ codebook^T` (tied), per reversed block a
4x4 s2 conv-transpose (no activation) then the residual stacks, final 1x1 conv.  forward 165-184.

Internal tensor conventions: activations NHWC as [B*H*W, C] bf16; the 3-channel image is padded to 8 channels and the
3-channel reconstruction to 64 so that every GEMM keeps K % 64 == 0 / N % 8 == 0; the corresponding kernel rows /
columns are zero and stay zero (their gradients are exactly 0).  `export_reference` / `load_reference_params`
translate to the TF variable names and shapes of SURVEY.md Appendix B.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

import dalle_hip as dh
from ..dp import GradReducer

IMG_CP = 8     # padded image channels
OUT_CP = 64    # padded reconstruction channels
ALIGN = 128
WS_POOL = 7    # weight gradients in flight before their slab reduces are flushed (2 reduce items each, 16 per batched launch)


def _ru(x, m):
    return (x + m - 1) // m * m


TAPS4 = [(ky - 1, kx - 1) for ky in range(4) for kx in range(4)]       # 4x4 s2 SAME: pad 1 before (Appendix A.8)
TAPS3 = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]       # 3x3 s1 SAME
TAPS3_REV = [(1 - ky, 1 - kx) for ky in range(3) for kx in range(3)]   # input gradient of the 3x3 conv


def _parity(p):
    """output-parity class of the stride-2 adjoint: [(k index, dy, dx)] for the 2x2 contributing taps."""
    def ax(par):
        return [(1, 0), (3, -1)] if par == 0 else [(0, 1), (2, 0)]   # (k, source offset)
    py, px = p >> 1, p & 1
    return [(ky * 4 + kx, dy, dx) for (ky, dy) in ax(py) for (kx, dx) in ax(px)]


class _Conv:
    def __init__(self, name, kind, cin, cout, H, W, cin_ref=None, cout_ref=None):
        self.name, self.kind, self.cin, self.cout, self.H, self.W = name, kind, cin, cout, H, W
        self.cin_ref = cin if cin_ref is None else cin_ref     # channels of the reference variable
        self.cout_ref = cout if cout_ref is None else cout_ref
        self.kk = {"down": 16, "up": 16, "res": 9, "final": 1}[kind]
        if kind == "down":
            self.Ho, self.Wo = H // 2, W // 2
        elif kind == "up":
            self.Ho, self.Wo = 2 * H, 2 * W
        else:
            self.Ho, self.Wo = H, W
        # kernel stored [kk][A][Bn]: conv: A=cin, Bn=cout ; conv-transpose (TF [kh,kw,Cout,Cin]): A=cout(z), Bn=cin(y)
        self.A, self.Bn = (cout, cin) if kind == "up" else (cin, cout)


class DiscreteVAE:
    def __init__(self, num_tokens, dimensions, convblocks, dim=512, hidden_dim=64, input_channels=3, recompute_grad=False,
                 use_bf16=False, stack_factor=1, batch_size=32, mode="train", device="cuda", process_group=None, world_size=1,
                 comm=None):
        if not torch.cuda.is_available():
            raise dh.DalleHipError("DiscreteVAE needs a HIP device (MI355X); there is no CPU fallback")
        dh.lib()
        self.num_tokens = num_tokens
        self.dim, self.hdim = dim, hidden_dim
        self.num_ch = input_channels
        self.H = self.W = dimensions
        self.convblocks = [tuple(b) for b in convblocks]
        self.recompute_grad = recompute_grad     # memory-only option: residual branches are re-run in backward()
        self.bf16 = use_bf16                     # kernels always compute in bf16 with fp32 accumulation
        assert math.log2(stack_factor).is_integer()
        # tf.space_to_depth in front of the encoder / tf.depth_to_space behind the decoder (reference :85-86,158-161): the
        # network runs on [B, H/s, W/s, C*s*s]; two index kernels do the re-layout, everything else is unchanged
        self.stack_factor = stack_factor
        assert dimensions % stack_factor == 0
        self.Hs = dimensions // stack_factor                 # spatial size the convolutions see
        self.c_st = input_channels * stack_factor ** 2       # channels after space_to_depth
        self.img_cp = _ru(self.c_st, 8)                      # ... padded for the 16-byte vector paths
        assert self.c_st <= OUT_CP, "stack_factor too large for the padded reconstruction width"
        self.fp32_tokens = False   # return_logits path on the exact-fp32 kernels (set by dalle_model_fn's load_vae_model)
        for _, ch in self.convblocks:
            assert ch % 16 == 0, "channel counts must be multiples of 16"
        assert num_tokens % 64 == 0, "num_tokens must be a multiple of 64"
        assert self.convblocks[-1][1] % 64 == 0, "the last block's channel count (codebook width) must be a multiple of 64"
        self.B, self.mode = batch_size, mode
        self.dev = torch.device(device)
        self.pg, self.world = process_group, world_size
        self.n_hid = self.convblocks[-1][1]
        self.grid = self.Hs // (2 ** len(self.convblocks))
        self.global_step = 0
        self._graphs, self._graph_warm, self._dyn_on, self._img_static = {}, False, False, None
        self._build_graph()
        self._alloc()
        self._dyn = torch.zeros(2, dtype=torch.float32, device=self.dev)   # [lr_t, temperature] read by graph-replayed kernels
        # gradient exchange (mean over replicas = SUM here x grad_scale 1/world in the Adam kernel), overlapped with backward
        self.reducer = GradReducer(self.g, world_size, comm=comm, pg=process_group)

    # ------------------------------------------------------------------ structure
    def _build_graph(self):
        convs: List[_Conv] = []
        H, cin, cin_ref = self.Hs, self.img_cp, self.c_st
        for b, (stack, ch) in enumerate(self.convblocks):
            for i in range(stack):
                p = f"encoder/block_{b}/layer_{i}/"
                if i == 0:
                    convs.append(_Conv(p + "conv_downsample", "down", cin, ch, H, H, cin_ref=cin_ref))
                    H //= 2
                else:
                    convs.append(_Conv(p + "conv_in", "res", ch, ch, H, H))
                    convs.append(_Conv(p + "conv_out", "res", ch, ch, H, H))
            cin = cin_ref = ch
        self.n_enc = len(convs)
        for b, (stack, ch) in enumerate(reversed(self.convblocks)):
            for i in range(stack):
                p = f"decoder/block_{b}/layer_{i}/"
                if i == 0:
                    convs.append(_Conv(p + "conv_upsample", "up", cin, ch, H, H))
                    H *= 2
                else:
                    convs.append(_Conv(p + "conv_in", "res", ch, ch, H, H))
                    convs.append(_Conv(p + "conv_out", "res", ch, ch, H, H))
            cin = ch
        convs.append(_Conv("decoder/conv2d", "final", cin, OUT_CP, H, H, cout_ref=self.c_st))
        self.convs = convs
        # flat parameter layout
        self.offset: Dict[str, int] = {}
        self.shape: Dict[str, tuple] = {}
        off = 0

        def add(name, shp):
            nonlocal off
            self.offset[name], self.shape[name] = off, shp
            off += _ru(int(np.prod(shp)), ALIGN)
        for c in convs[:self.n_enc]:
            add(c.name + "/kernel", (c.kk, c.A, c.Bn))
            add(c.name + "/bias", (c.cout,))
        add("codebook/codebook", (self.n_hid, self.num_tokens))
        for c in convs[self.n_enc:]:
            add(c.name + "/kernel", (c.kk, c.A, c.Bn))
            add(c.name + "/bias", (c.cout,))
        self.total = off

    def _alloc(self):
        B, dev = self.B, self.dev
        b16 = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        n = self.total
        self.p, self.g = torch.zeros(n, **f32), torch.zeros(n, **f32)
        self.m, self.v = torch.zeros(n, **f32), torch.zeros(n, **f32)
        self.pb = torch.zeros(n, **b16)
        self.gc2 = torch.zeros(self.n_hid * self.num_tokens, **f32)  # second contribution to the tied codebook gradient
        # derived weight copies: views of ONE flat buffer, so that the per-step refresh is two batched launches
        # (dmi_transpose_bf16_batch + dmi_weight_gather_batch) instead of one launch per copy
        self.wf, self.wd, self.wp = {}, {}, {}
        plan = []                                   # (dict, key, index or None, rows, cols)
        max_col = 0
        for c in self.convs:
            K = c.kk * c.cin
            Kp = _ru(K, 64)
            M_out = B * c.Ho * c.Wo
            if c.kind in ("down", "res", "final"):
                plan.append(("wf", c.name, None, c.cout, Kp))                             # [co][(k,ci)] fwd
                max_col = max(max_col, M_out * Kp)
            if c.kind == "res":
                plan.append(("wd", c.name, None, c.cin, _ru(9 * c.cout, 64)))             # [ci][(k,co)] dgrad
                max_col = max(max_col, M_out * _ru(9 * c.cout, 64))
            if c.kind == "down":
                for q in range(4):
                    plan.append(("wp", c.name, q, c.cin, _ru(4 * c.cout, 64)))            # dgrad parity
                max_col = max(max_col, M_out * _ru(4 * c.cout, 64))
            if c.kind == "up":
                for q in range(4):
                    plan.append(("wp", c.name, q, c.cout, _ru(4 * c.cin, 64)))            # fwd parity [cz][(a,b,cy)]
                plan.append(("wd", c.name, None, c.cin, _ru(16 * c.cout, 64)))            # [cy][(k,cz)] for dy
                max_col = max(max_col, B * c.H * c.W * _ru(4 * c.cin, 64), B * c.H * c.W * _ru(16 * c.cout, 64))
        plan.append(("cb", "codebook_t", None, self.num_tokens, self.n_hid))
        off = 0
        self._wc_off = {}
        for kind, name, q, rows, cols in plan:
            self._wc_off[(kind, name, q)] = off
            off += _ru(rows * cols, ALIGN)
        self.wcopies = torch.zeros(off, **b16)
        for kind, name, q, rows, cols in plan:
            o = self._wc_off[(kind, name, q)]
            v = self.wcopies[o:o + rows * cols].view(rows, cols)
            if kind == "wp":
                self.wp.setdefault(name, [None] * 4)[q] = v
            elif kind == "cb":
                self.codebook_t = v
            else:
                getattr(self, kind)[name] = v
        self._refresh_tables = None
        self.col = torch.empty(max_col, **b16)
        # im2col matrices of the forward convolutions kept for their weight gradients (288 GB HBM: ~8 GB for vae_coco at 16
        # images) -- the backward pass then gathers only dy; allocated on first use in train mode
        self.col_keep = {}
        self._col_valid = set()
        self.keep_cols = (self.mode == "train")
        # activations
        self.x_img = torch.empty(B * self.Hs * self.Hs, self.img_cp, **b16)
        self.img_st = torch.empty(B * self.Hs * self.Hs, self.c_st, **f32) if self.stack_factor > 1 else None
        self.act_in = [None] * len(self.convs)     # input of each conv (kept for the weight gradients)
        self.act_out = []
        # recompute_grad (reference vae_tf/models.py:8-43,102,150): the residual branch is re-run in the backward pass instead of
        # keeping its inner activation -- here: every conv_in output shares ONE buffer, refilled by backward() before use
        shared = None
        if self.recompute_grad:
            shared = torch.empty(max([B * c.Ho * c.Wo * c.cout for c in self.convs if c.name.endswith("conv_in")] + [8]), **b16)
        for c in self.convs:
            if shared is not None and c.name.endswith("conv_in"):
                self.act_out.append(shared[:B * c.Ho * c.Wo * c.cout].view(B * c.Ho * c.Wo, c.cout))
            else:
                self.act_out.append(torch.empty(B * c.Ho * c.Wo, c.cout, **b16))
        Mg = B * self.grid * self.grid
        self.Mg = Mg
        self.logits = torch.empty(Mg, self.num_tokens, **f32)
        self.u = torch.empty(Mg, self.num_tokens, **f32)
        self.y = torch.empty(Mg, self.num_tokens, **b16)
        self.y_soft = torch.empty(Mg, self.num_tokens, **b16)
        self.index = torch.empty(Mg, dtype=torch.int32, device=dev)
        self.xdec = torch.empty(Mg, self.n_hid, **b16)
        self.loss = torch.zeros(1, **f32)
        # scratch for parity outputs and gradients
        max_act = max(int(a.numel()) for a in self.act_out + [self.x_img])
        self.par = torch.empty(max_act, **b16)
        self.ga = torch.empty(max_act, **b16)
        self.gb = torch.empty(max_act, **b16)
        self.gc = torch.empty(max_act, **b16)
        self.dy = torch.empty(Mg, self.num_tokens, **b16)
        self.dlogits = torch.empty(Mg, self.num_tokens, **b16)
        wsz = 1 << 20
        for c in self.convs:
            M_out = B * c.Ho * c.Wo
            M_in = B * c.H * c.W
            wsz = max(wsz, dh.gemm_tn_workspace_bytes(max(M_out, M_in), c.kk * max(c.cin, c.cout), max(c.cin, c.cout)),
                      dh.colsum_workspace_bytes(max(M_out, M_in), max(c.cout, c.cin)))
        wsz = max(wsz, dh.gemm_tn_workspace_bytes(Mg, self.n_hid, self.num_tokens), dh.mse_workspace_bytes())
        self.ws = torch.empty(int(wsz) + 1024, dtype=torch.uint8, device=dev)
        # backward(): the split-m slab reduces of the weight gradients are DEFERRED and run a few layers at a time in one launch
        # (the small configurations spent a fifth of their step in ~60 reduce launches of a few microseconds); every pending
        # weight gradient owns one workspace of this pool until the flush
        self.ws_cs = torch.empty(int(max(dh.colsum_workspace_bytes(max(B * c.Ho * c.Wo, B * c.H * c.W), max(c.cout, c.cin))
                                         for c in self.convs)) + 1024, dtype=torch.uint8, device=dev)   # column sums reduce at once
        self.ws_pool = [torch.empty_like(self.ws) for _ in range(WS_POOL)]     # (self.ws stays free for the immediate calls)
        self._ws_next = 0
        self.deferred = dh.DeferredReduces()

    # ------------------------------------------------------------------ parameters
    def view(self, buf, name):
        o = self.offset[name]
        shp = self.shape[name]
        return buf[o:o + int(np.prod(shp))].view(shp)

    def init_params(self, seed=4321):
        """glorot-uniform kernels / codebook, zero biases (tf.layers defaults)."""
        g = torch.Generator().manual_seed(seed)
        P = OrderedDict()
        for name, shp in self.reference_variables().items():
            if len(shp) == 1:
                P[name] = np.zeros(shp, np.float32)
            else:
                if len(shp) == 4:
                    rf = shp[0] * shp[1]
                    fi, fo = shp[2] * rf, shp[3] * rf
                else:
                    fi, fo = shp
                lim = math.sqrt(6.0 / (fi + fo))
                P[name] = ((torch.rand(*shp, generator=g) * 2 - 1) * lim).numpy()
        self.load_reference_params(P)

    def reference_variables(self) -> "OrderedDict[str, tuple]":
        out: "OrderedDict[str, tuple]" = OrderedDict()
        for i, c in enumerate(self.convs):
            if i == self.n_enc:
                out["codebook/codebook"] = (self.n_hid, self.num_tokens)
            k = {16: 4, 9: 3, 1: 1}[c.kk]
            if c.kind == "up":
                out[c.name + "/kernel"] = (k, k, c.cout_ref, c.cin_ref)
            else:
                out[c.name + "/kernel"] = (k, k, c.cin_ref, c.cout_ref)
            out[c.name + "/bias"] = (c.cout_ref,)
        return out

    def load_reference_params(self, P: Dict[str, np.ndarray]):
        with torch.no_grad():
            self.p.zero_()
            for c in self.convs:
                w = torch.from_numpy(np.ascontiguousarray(P[c.name + "/kernel"])).float()
                k = w.shape[0]
                w = w.reshape(k * k, w.shape[2], w.shape[3])
                dst = self.view(self.p, c.name + "/kernel")
                dst[:, :w.shape[1], :w.shape[2]] = w
                bsrc = torch.from_numpy(np.ascontiguousarray(P[c.name + "/bias"])).float()
                self.view(self.p, c.name + "/bias")[:bsrc.shape[0]] = bsrc
            self.view(self.p, "codebook/codebook").copy_(torch.from_numpy(np.ascontiguousarray(P["codebook/codebook"])))
        self.refresh_compute_copies(cast=True)

    def export_reference(self, buf=None) -> "OrderedDict[str, np.ndarray]":
        buf = self.p if buf is None else buf
        out: "OrderedDict[str, np.ndarray]" = OrderedDict()
        ref = self.reference_variables()
        for c in self.convs:
            shp = ref[c.name + "/kernel"]
            a = self.view(buf, c.name + "/kernel").detach().float().cpu().numpy()
            out[c.name + "/kernel"] = np.ascontiguousarray(a[:, :shp[2], :shp[3]]).reshape(shp)
            out[c.name + "/bias"] = np.ascontiguousarray(self.view(buf, c.name + "/bias").detach().float().cpu().numpy()[:shp[3] if c.kind != "up" else shp[2]])
        out["codebook/codebook"] = self.view(buf, "codebook/codebook").detach().float().cpu().numpy()
        return out

    def refresh_compute_copies(self, cast=False):
        """bf16 master copy (cast) + every derived weight layout of the step, in two batched launches: the [in,out] -> [out,in]
        transposes (forward kernels, transposed-conv dy kernels, codebook) and the tap gathers (dgrad / output-parity kernels).
        A transpose whose row count needs zero padding to the 64-multiple K pitch (none in the shipped configurations) keeps its
        own launch."""
        if cast:
            dh.cast_f32_bf16(self.p, self.pb, self.total)
        if self._refresh_tables is None:
            tr, tile, ga, blk, padded = [], 0, [], 0, []
            for c in self.convs:
                src = self.offset[c.name + "/kernel"]
                K = c.kk * c.cin
                if c.kind in ("down", "res", "final"):
                    if _ru(K, 64) == K:
                        tr.append([src, self._wc_off[("wf", c.name, None)], K, c.cout, tile])
                        tile += ((K + 63) // 64) * ((c.cout + 63) // 64)
                    else:
                        padded.append((c.name, "wf", K, _ru(K, 64), c.cout))
                if c.kind == "res":
                    ga.append((src, self._wc_off[("wd", c.name, None)], c.cin, c.cout, list(range(9)), _ru(9 * c.cout, 64)))
                if c.kind == "down":
                    for q in range(4):
                        ga.append((src, self._wc_off[("wp", c.name, q)], c.cin, c.cout, [t[0] for t in _parity(q)], _ru(4 * c.cout, 64)))
                if c.kind == "up":
                    for q in range(4):
                        ga.append((src, self._wc_off[("wp", c.name, q)], c.cout, c.cin, [t[0] for t in _parity(q)], _ru(4 * c.cin, 64)))
                    Kz = 16 * c.cout
                    if _ru(Kz, 64) == Kz:
                        tr.append([src, self._wc_off[("wd", c.name, None)], Kz, c.cin, tile])
                        tile += ((Kz + 63) // 64) * ((c.cin + 63) // 64)
                    else:
                        padded.append((c.name, "wd", Kz, _ru(Kz, 64), c.cin))
            tr.append([self.offset["codebook/codebook"], self._wc_off[("cb", "codebook_t", None)], self.n_hid, self.num_tokens, tile])
            tile += ((self.n_hid + 63) // 64) * ((self.num_tokens + 63) // 64)
            rows = []
            for src, dst, A, Bn, idx, ldo in ga:
                rows.append([src, dst, A, Bn, len(idx), ldo, blk] + idx + [0] * (16 - len(idx)))
                blk += (A * ldo + 2047) // 2048
            self._refresh_tables = dict(tr=torch.tensor(tr, dtype=torch.int64, device=self.dev), tiles=tile,
                                        ga=torch.tensor(rows, dtype=torch.int64, device=self.dev) if rows else None, blocks=blk,
                                        padded=padded)
        T = self._refresh_tables
        dh.transpose_batch(self.pb, self.wcopies, T["tr"], T["tr"].shape[0], T["tiles"])
        if T["ga"] is not None:
            dh.weight_gather_batch(self.pb, self.wcopies, T["ga"], T["ga"].shape[0], T["blocks"])
        for name, kind, R, Rp, C in T["padded"]:
            dh.transpose_padded(self.view(self.pb, name + "/kernel"), getattr(self, kind)[name], R, Rp, C)

    # ------------------------------------------------------------------ conv building blocks
    def _w(self, name):
        return self.view(self.pb, name)

    @staticmethod
    def _implicit_ok(c: _Conv):
        """layers whose im2col can stay implicit in all three GEMMs: 64-channel-aligned input, power-of-two output grid."""
        p2 = lambda v: v > 0 and (v & (v - 1)) == 0
        return c.cin % 64 == 0 and p2(c.Ho) and p2(c.Wo)

    def _conv_fwd(self, c: _Conv, x, out, flags=0, residual=None):
        B = self.B
        bias = self._w(c.name + "/bias")
        if c.kind == "final" and c.cin % 64 == 0:
            dh.gemm_nt(x, c.cin, self.wf[c.name], c.cin, out, c.cout, B * c.H * c.W, c.cout, c.cin, flags | dh.GEMM_BIAS, bias=bias)
        elif c.kind in ("down", "res", "final"):
            K = c.kk * c.cin
            Kp = _ru(K, 64)
            taps, s = {"down": (TAPS4, 2), "res": (TAPS3, 1), "final": ([(0, 0)], 1)}[c.kind]
            if c.kind != "final" and self._implicit_ok(c):
                # implicit im2col in forward, input gradient and weight gradient: no column matrix for this layer at all
                hook = getattr(self, "event_hook", None)   # bench.py: HIP events around one named convolution launch
                timed = hook is not None and hook[0] == c.name
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                dh.conv_gemm_nt(x, B, c.H, c.W, c.cin, c.Ho, c.Wo, s, taps, self.wf[c.name], Kp, out, c.cout, c.cout,
                                flags | dh.GEMM_BIAS, bias=bias, residual=residual)
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    hook[1].append((e0, e1))
                return
            col = self.col
            if self.keep_cols and c.kind != "final":
                col = self.col_keep.get(c.name)
                if col is None:
                    col = self.col_keep[c.name] = torch.empty(B * c.Ho * c.Wo * Kp, dtype=torch.bfloat16, device=self.dev)
                self._col_valid.add(c.name)
            dh.im2col(x, col, B, c.H, c.W, c.cin, c.Ho, c.Wo, s, taps, Kp)
            dh.gemm_nt(col, Kp, self.wf[c.name], Kp, out, c.cout, B * c.Ho * c.Wo, c.cout, Kp, flags | dh.GEMM_BIAS,
                       bias=bias, residual=residual)
        else:  # up: 4 output-parity GEMMs + interleave
            Kp = _ru(4 * c.cin, 64)
            Mi = B * c.H * c.W
            for p in range(4):
                taps = [(t[1], t[2]) for t in _parity(p)]
                if c.cin % 64 == 0:
                    dh.conv_gemm_nt(x, B, c.H, c.W, c.cin, c.H, c.W, 1, taps, self.wp[c.name][p], Kp, self.par[p * Mi * c.cout:],
                                    c.cout, c.cout, dh.GEMM_BIAS, bias=bias)
                    continue
                dh.im2col(x, self.col, B, c.H, c.W, c.cin, c.H, c.W, 1, taps, Kp)
                dh.gemm_nt(self.col, Kp, self.wp[c.name][p], Kp, self.par[p * Mi * c.cout:], c.cout, Mi, c.cout, Kp, dh.GEMM_BIAS, bias=bias)
            dh.pixel_interleave(self.par, out, B, c.H, c.W, c.cout)

    # ------------------------------------------------------------------ forward
    def forward(self, features, return_recon_loss=False, return_logits=False, hard_gumbel=True, temperature=1.0, noise=None,
                need_grad=None):
        """features: images NHWC fp32 in [-1,1] (or {"inputs": ...}).  Mirrors vae_tf/models.py:165-184.
        `noise`: optional uniforms [B,g,g,num_tokens] in [1e-9,1) (injected for parity; drawn with torch.rand otherwise)."""
        img = features["inputs"] if isinstance(features, dict) else features
        img = img.to(device=self.dev, dtype=torch.float32).contiguous()
        B = self.B
        assert img.shape == (B, self.H, self.W, self.num_ch), f"expected {(B, self.H, self.W, self.num_ch)}, got {tuple(img.shape)}"
        self.img = img
        if return_logits and self.fp32_tokens:
            return self.encode_logits_fp32(img)
        self._col_valid.clear()
        Np = B * self.Hs * self.Hs
        if self.stack_factor > 1:
            dh.space_to_depth_f32(img, self.img_st, B, self.Hs, self.Hs, self.num_ch, self.stack_factor, self.c_st)
            img = self.img_st          # the loss is permutation-invariant: computed in the stacked layout
        self.img_net = img
        dh.pad_channels(img, self.x_img, Np, self.c_st, self.img_cp)
        x = self.x_img
        convs = self.convs
        i = 0
        while i < self.n_enc:
            c = convs[i]
            self.act_in[i] = x
            if c.kind == "down":
                self._conv_fwd(c, x, self.act_out[i])
                x = self.act_out[i]
                i += 1
            else:  # residual pair: conv_in (bias+relu), conv_out (bias + residual x)
                self._conv_fwd(c, x, self.act_out[i], flags=dh.GEMM_RELU)
                self.act_in[i + 1] = self.act_out[i]
                self._conv_fwd(convs[i + 1], self.act_out[i], self.act_out[i + 1], flags=dh.GEMM_RESIDUAL, residual=x)
                x = self.act_out[i + 1]
                i += 2
        self.x_enc = x
        dh.gemm_nt(x, self.n_hid, self.codebook_t, self.n_hid, self.logits, self.num_tokens, self.Mg, self.num_tokens, self.n_hid,
                   dh.GEMM_OUT_F32)
        if return_logits:
            return self.logits.view(B, self.grid, self.grid, self.num_tokens)
        if noise is None:
            self.u.uniform_(1e-9, 1.0)
        elif not (isinstance(noise, str) and noise == "preset"):   # "preset": self.u was filled by the caller (graph replay)
            self.u.copy_(noise.to(self.dev).reshape(self.Mg, self.num_tokens))
        self.temperature = float(temperature)
        dh.gumbel_softmax_fwd(self.logits, self.u, self.y, self.y_soft, self.index, self.Mg, self.num_tokens, self.temperature,
                              hard_gumbel, temperature_dev=self._temp_dev())
        x = self._decoder_from_y()
        if not return_recon_loss:
            return self.reconstruction()
        need_grad = (self.mode == "train") if need_grad is None else need_grad
        # d(out) of mean((img-out)^2); 1/world folds the CrossShardOptimizer mean (src/model_fns_tf.py:61) into the gradient
        dh.mse_loss(self.img_net, x, self.ga if need_grad else None, self.loss, B * self.Hs * self.Hs, self.c_st, OUT_CP, 1.0, self.ws)
        return self.loss[0], self.reconstruction()

    def _decoder_from_y(self):
        """decoder (vae_tf/models.py:123-163) on self.y [Mg, num_tokens] (soft / hard one-hot): y @ codebook^T, then per reversed
        block a 4x4 s2 conv-transpose and the residual stacks, final 1x1 conv -> self.out_pad"""
        convs = self.convs
        dh.gemm_nt(self.y, self.num_tokens, self._w("codebook/codebook"), self.num_tokens, self.xdec, self.n_hid, self.Mg, self.n_hid,
                   self.num_tokens)
        x = self.xdec
        i = self.n_enc
        while i < len(convs):
            c = convs[i]
            self.act_in[i] = x
            if c.kind in ("up", "final"):
                self._conv_fwd(c, x, self.act_out[i])
                x = self.act_out[i]
                i += 1
            else:
                self._conv_fwd(c, x, self.act_out[i], flags=dh.GEMM_RELU)
                self.act_in[i + 1] = self.act_out[i]
                self._conv_fwd(convs[i + 1], self.act_out[i], self.act_out[i + 1], flags=dh.GEMM_RESIDUAL, residual=x)
                x = self.act_out[i + 1]
                i += 2
        self.out_pad = x
        return x

    def decode_tokens(self, tokens):
        """image-token ids [B, g*g] (what DALL-E emits) -> images [B,H,W,C] fp32: one-hot rows through the decoder.
        The sampling half the reference leaves unfinished (predict raises upstream, src/model_fns.py:135-136)."""
        tok = tokens.to(device=self.dev, dtype=torch.int64).reshape(self.Mg)
        assert int(tok.min()) >= 0 and int(tok.max()) < self.num_tokens
        self.y.zero_()
        self.y.scatter_(1, tok.view(-1, 1), 1.0)       # index plumbing: the one-hot the hard Gumbel path would produce
        self._decoder_from_y()
        return self.reconstruction()

    def reconstruction(self):
        out = torch.empty(self.B, self.H, self.W, self.num_ch, dtype=torch.float32, device=self.dev)
        if self.stack_factor == 1:
            dh.unpad_channels(self.out_pad, out, self.B * self.H * self.W, self.num_ch, OUT_CP)
            return out
        st = torch.empty(self.B * self.Hs * self.Hs, self.c_st, dtype=torch.float32, device=self.dev)
        dh.unpad_channels(self.out_pad, st, self.B * self.Hs * self.Hs, self.c_st, OUT_CP)
        dh.depth_to_space_f32(st, out, self.B, self.Hs, self.Hs, self.num_ch, self.stack_factor, self.c_st)
        return out

    # ------------------------------------------------------------------ fp32 tokenising encoder
    def encode_logits_fp32(self, img):
        """Encoder + `x @ codebook` entirely in fp32 from the fp32 master weights (dmi_conv2d_f32): the path the reference
        takes when the VAE tokenises images for DALL-E (src/model_fns.py:43-51 builds it without use_bf16; encoder
        src/vae_tf/models.py:81-120).  Forward only.  Returns logits [B, g, g, num_tokens] fp32."""
        B, dev = self.B, self.dev
        if getattr(self, "_f32", None) is None:
            mx = max(B * c.Ho * c.Wo * c.cout for c in self.convs[:self.n_enc])
            self._f32 = dict(x=torch.empty(B * self.Hs * self.Hs, self.img_cp, dtype=torch.float32, device=dev),
                             a=[torch.empty(mx, dtype=torch.float32, device=dev) for _ in range(3)])
        f = self._f32
        dh.space_to_depth_f32(img, f["x"], B, self.Hs, self.Hs, self.num_ch, self.stack_factor, self.img_cp)
        x, free = f["x"], list(f["a"])

        def conv(c, xin, out, relu=False, residual=None):
            taps, st = (TAPS4, 2) if c.kind == "down" else (TAPS3, 1)
            dh.conv2d_f32(xin, B, c.H, c.W, c.cin, c.Ho, c.Wo, st, taps, self.view(self.p, c.name + "/kernel"),
                          self.view(self.p, c.name + "/bias"), residual, out, c.cout, relu=relu)
        i = 0
        while i < self.n_enc:
            c = self.convs[i]
            if c.kind == "down":
                out = free.pop(0)
                conv(c, x, out)
                if x is not f["x"]:
                    free.append(x)
                x = out
                i += 1
            else:
                h, out = free.pop(0), free.pop(0)
                conv(c, x, h, relu=True)
                conv(self.convs[i + 1], h, out, residual=x)
                free.extend([h, x])
                x = out
                i += 2
        dh.conv2d_f32(x, self.Mg, 1, 1, self.n_hid, 1, 1, 1, [(0, 0)], self.view(self.p, "codebook/codebook"), None, None,
                      self.logits, self.num_tokens)
        return self.logits.view(B, self.grid, self.grid, self.num_tokens)

    # ------------------------------------------------------------------ backward
    def _gv(self, name):
        return self.view(self.g, name)

    def _defer_ws(self, defer):
        """(workspace, deferred list or None) of the next weight gradient: immediate reduce on the shared workspace, or a pool
        slot whose reduces run at the next _flush_reduces()"""
        if not defer:
            return self.ws, None
        if self._ws_next >= WS_POOL:
            self._flush_reduces()
        w = self.ws_pool[self._ws_next]
        self._ws_next += 1
        return w, self.deferred

    def _flush_reduces(self):
        self.deferred.run()
        self._ws_next = 0

    def _wgrad(self, c: _Conv, x_in, dy, defer=False):
        """dW[(k,ci)][co] = col(x)^T . dy (+ bias gradient fused) -- lands directly in the TF kernel layout."""
        B = self.B
        ws, dfr = self._defer_ws(defer)
        if c.kind == "final":
            dh.gemm_tn(x_in, c.cin, dy, c.cout, self._gv(c.name + "/kernel"), B * c.H * c.W, c.cin, c.cout, ws,
                       dbias=self._gv(c.name + "/bias"), deferred=dfr)
            return
        K = c.kk * c.cin
        Kp = _ru(K, 64)
        taps, s = (TAPS4, 2) if c.kind == "down" else (TAPS3, 1)
        if self._implicit_ok(c):
            dh.conv_wgrad_tn(x_in, B, c.H, c.W, c.cin, c.Ho, c.Wo, s, taps, dy, c.cout, c.cout, self._gv(c.name + "/kernel"),
                             ws, dbias=self._gv(c.name + "/bias"), deferred=dfr)
            return
        col = self.col
        if c.name in self._col_valid:        # the forward pass of this step left col(x_in) in its kept buffer
            col = self.col_keep[c.name]
        else:
            dh.im2col(x_in, col, B, c.H, c.W, c.cin, c.Ho, c.Wo, s, taps, Kp)
        dh.gemm_tn(col, Kp, dy, c.cout, self._gv(c.name + "/kernel"), B * c.Ho * c.Wo, K, c.cout, ws,
                   dbias=self._gv(c.name + "/bias"), deferred=dfr)

    def _dgrad3(self, c: _Conv, dy, out, flags=0, residual=None, relu_src=None):
        Kp = _ru(9 * c.cout, 64)
        if c.cout % 64 == 0:   # implicit im2col: the 9x column matrix of dy never exists in HBM (bit-identical results)
            dh.conv_gemm_nt(dy, self.B, c.H, c.W, c.cout, c.H, c.W, 1, TAPS3_REV, self.wd[c.name], Kp, out, c.cin, c.cin, flags,
                            residual=residual, relu_src=relu_src)
            return
        dh.im2col(dy, self.col, self.B, c.H, c.W, c.cout, c.H, c.W, 1, TAPS3_REV, Kp)
        dh.gemm_nt(self.col, Kp, self.wd[c.name], Kp, out, c.cin, self.B * c.H * c.W, c.cin, Kp, flags, residual=residual, relu_src=relu_src)

    def _up_backward(self, c: _Conv, x_in, dz, out, defer=False):
        """backward of the transposed conv z = conv2d_transpose(y) (vae_tf/models.py:133-137): a stride-2 4x4 SAME conv of dz.
        dW[kh,kw,Cout,Cin] and dbias into g, dy [B*H*W, cin] into `out`."""
        B = self.B
        Mi = B * c.H * c.W
        Kz = 16 * c.cout
        Kzp = _ru(Kz, 64)
        p2 = lambda v: v > 0 and (v & (v - 1)) == 0
        ws, dfr = self._defer_ws(defer)
        if c.cout % 64 == 0 and p2(c.H) and p2(c.W):
            # both GEMMs gather dz implicitly
            dh.colsum(dz, c.cout, self._gv(c.name + "/bias"), B * c.Ho * c.Wo, c.cout, self.ws_cs)
            dh.conv_wgrad_tn(dz, B, c.Ho, c.Wo, c.cout, c.H, c.W, 2, TAPS4, x_in, c.cin, c.cin,
                             self._gv(c.name + "/kernel"), ws, deferred=dfr)
            dh.conv_gemm_nt(dz, B, c.Ho, c.Wo, c.cout, c.H, c.W, 2, TAPS4, self.wd[c.name], Kzp, out, c.cin, c.cin)
        else:
            dh.im2col(dz, self.col, B, c.Ho, c.Wo, c.cout, c.H, c.W, 2, TAPS4, Kzp)     # stride-2 conv view of dz
            dh.colsum(dz, c.cout, self._gv(c.name + "/bias"), B * c.Ho * c.Wo, c.cout, self.ws_cs)
            dh.gemm_tn(self.col, Kzp, x_in, c.cin, self._gv(c.name + "/kernel"), Mi, Kz, c.cin, ws, deferred=dfr)
            dh.gemm_nt(self.col, Kzp, self.wd[c.name], Kzp, out, c.cin, Mi, c.cin, Kzp)

    def _dgrad_down(self, c: _Conv, dy, out):
        """input gradient of the 4x4 stride-2 SAME conv (vae_tf/models.py:90-92) = its transposed conv: 4 output-parity GEMMs
        over the 2x2 contributing taps + a pixel interleave; dy [B*Ho*Wo, cout] -> out [B*H*W, cin]."""
        B = self.B
        Mo = B * c.Ho * c.Wo
        Kp = _ru(4 * c.cout, 64)
        for p in range(4):
            taps = [(t[1], t[2]) for t in _parity(p)]
            if c.cout % 64 == 0:
                dh.conv_gemm_nt(dy, B, c.Ho, c.Wo, c.cout, c.Ho, c.Wo, 1, taps, self.wp[c.name][p], Kp,
                                self.par[p * Mo * c.cin:], c.cin, c.cin)
                continue
            dh.im2col(dy, self.col, B, c.Ho, c.Wo, c.cout, c.Ho, c.Wo, 1, taps, Kp)
            dh.gemm_nt(self.col, Kp, self.wp[c.name][p], Kp, self.par[p * Mo * c.cin:], c.cin, Mo, c.cin, Kp)
        dh.pixel_interleave(self.par, out, B, c.Ho, c.Wo, c.cin)

    def backward(self):
        """Gradients of the last forward(return_recon_loss=True) into the flat fp32 buffer `g`."""
        B, convs = self.B, self.convs
        d = self.ga                       # gradient wrt the padded reconstruction [M0, 64]
        spare = [self.gb, self.gc]
        i = len(convs) - 1
        mark = [self.total]               # g[mark, total) has been handed to the exchange

        def ready_down_to(off):           # the layout follows the forward order, backward finishes it from the end
            if off < mark[0]:
                if self.world > 1:
                    self._flush_reduces()  # the exchange takes g[off, mark): its pending slab reduces must have landed
                self.reducer.ready(off, mark[0])
                mark[0] = off
        while i >= 0:
            c = convs[i]
            lowest = convs[i - 1] if c.kind == "res" else c     # a residual pair is processed as one unit
            if c.kind == "final":
                M = B * c.H * c.W
                self._wgrad(c, self.act_in[i], d, defer=True)
                nd = spare[0]
                dh.gemm_nt(d, c.cout, self._w(c.name + "/kernel"), c.cout, nd, c.cin, M, c.cin, c.cout)   # K = 64 padded channels
                spare[0], d = d, nd
                i -= 1
            elif c.kind == "res":           # c = conv_out of a residual pair (i-1 = conv_in)
                cin_conv = convs[i - 1]
                x_in, r = self.act_in[i - 1], self.act_in[i]
                if self.recompute_grad:      # re-run the branch's first conv into the shared buffer (bit-identical to the forward)
                    self._conv_fwd(cin_conv, x_in, r, flags=dh.GEMM_RELU)
                self._wgrad(c, r, d, defer=True)
                da = spare[0]
                self._dgrad3(c, d, da, flags=dh.GEMM_RELU_MASK, relu_src=r)
                self._wgrad(cin_conv, x_in, da, defer=True)
                nd = spare[1]
                self._dgrad3(cin_conv, da, nd, flags=dh.GEMM_RESIDUAL, residual=d)
                spare[1], d = d, nd
                i -= 2
            elif c.kind == "up":
                nd = spare[0]
                self._up_backward(c, self.act_in[i], d, nd, defer=True)
                spare[0], d = d, nd
                i -= 1
            else:  # down
                self._wgrad(c, self.act_in[i], d, defer=True)
                if i > 0:
                    nd = spare[0]
                    self._dgrad_down(c, d, nd)
                    spare[0], d = d, nd
                i -= 1
            ready_down_to(self.offset[lowest.name + "/kernel"])
            if i == self.n_enc - 1:
                # ---- tied codebook + gumbel (between decoder and encoder); d = gradient wrt xdec [Mg, n_hid]
                T, nh, Mg = self.num_tokens, self.n_hid, self.Mg
                dh.gemm_tn(d, nh, self.y, T, self.gc2, Mg, nh, T, self.ws)                               # dC from x_dec = y C^T
                dh.gemm_nt(d, nh, self.codebook_t, nh, self.dy, T, Mg, T, nh)                            # dy = dxdec . C
                dh.gumbel_softmax_bwd(self.dy, self.y_soft, self.dlogits, Mg, T, self.temperature, temperature_dev=self._temp_dev())
                dh.gemm_tn(self.x_enc, nh, self.dlogits, T, self._gv("codebook/codebook"), Mg, nh, T, self.ws)  # dC from logits = x C
                dh.add_f32(self._gv("codebook/codebook"), self.gc2, nh * T)
                nd = spare[0]
                dh.gemm_nt(self.dlogits, T, self._w("codebook/codebook"), T, nd, nh, Mg, nh, T)          # dx_enc = dlogits . C^T
                spare[0], d = d, nd
                ready_down_to(self.offset["codebook/codebook"])
        ready_down_to(0)
        self._flush_reduces()

    # ------------------------------------------------------------------ optimizer
    def optimizer_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """tf.train.AdamOptimizer (src/model_fns_tf.py:58-60; Appendix A.8): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
        p -= lr_t*m/(sqrt(v)+eps).  grad_scale 1/world = CrossShardOptimizer's mean over replicas (:61)."""
        self.reducer.finish()
        lr_t = self.lr_t(lr, beta1, beta2)
        dh.adam_step(self.p, self.g, self.m, self.v, self.pb, self.total, None, 0.0, lr_t, beta1, beta2, eps, 0.0, 1.0 / self.world,
                     lr_dev=self._dyn[0:1] if self._dyn_on else None)
        self.refresh_compute_copies(cast=False)
        self.global_step += 1

    def lr_t(self, lr, beta1=0.9, beta2=0.999):
        t = self.global_step + 1
        return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

    # ------------------------------------------------------------------ whole step as one HIP graph
    def _temp_dev(self):
        return self._dyn[1:2] if self._dyn_on else None

    def train_step(self, features, lr, hard_gumbel=True, temperature=1.0, noise=None, graph=None):
        """forward + backward + optimizer_step.  The small configurations are launch-bound (vae_example: ~170 launches of a
        few microseconds each; 3.2 ms per step from Python against ~1 ms of kernel time, profiles/r02_bench_vae_example.json),
        so the step is captured ONCE as a HIP graph and replayed: per step the host then issues the image copy, the Gumbel
        uniforms (drawn eagerly), two scalar fills and one graph launch.  The learning rate lr_t
        (tf.train.AdamOptimizer's bias-corrected rate, src/model_fns_tf.py:58) and the annealed temperature (:40-45) live in
        a 2-float device buffer the kernels read (dmi_adam_step lr_dev / dmi_gumbel_softmax_* temperature_dev), so one graph
        serves the whole schedule.  Eager when: data-parallel (the exchange runs on its own stream and communicator), a
        bench event hook is set, DALLE_VAE_GRAPH=0, or graph=False.  Same kernels, same order: results are bit-identical to
        the eager step (tested).  Returns the device scalar loss."""
        want = (os.environ.get("DALLE_VAE_GRAPH", "1") != "0") if graph is None else bool(graph)
        img = features["inputs"] if isinstance(features, dict) else features
        if not want or self.world > 1 or getattr(self, "event_hook", None) is not None:
            self.forward(img, return_recon_loss=True, hard_gumbel=hard_gumbel, temperature=temperature, noise=noise, need_grad=True)
            self.backward()
            self.optimizer_step(lr)
            return self.loss[0]
        if self._img_static is None:
            self._img_static = torch.empty(self.B, self.H, self.W, self.num_ch, dtype=torch.float32, device=self.dev)
        self._img_static.copy_(img.to(dtype=torch.float32), non_blocking=True)
        if noise is None:
            self.u.uniform_(1e-9, 1.0)
        else:
            self.u.copy_(noise.to(self.dev).reshape(self.Mg, self.num_tokens))
        self._dyn[0:1].fill_(self.lr_t(lr))          # by-value kernel arguments: no host buffer whose lifetime could race a replay
        self._dyn[1:2].fill_(float(temperature))
        self.temperature = float(temperature)
        key = bool(hard_gumbel)
        g = self._graphs.get(key)
        if g is None:
            self._dyn_on = True
            try:
                if not self._graph_warm:
                    # first step eager (on the kernels' device-scalar path): lazily allocated buffers come into being
                    # outside the capture
                    self._step_body(key)
                    self._graph_warm = True
                    return self.loss[0]
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                step0 = self.global_step
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other host threads (input producer) stay free to call HIP
                    self._step_body(key)
                self.global_step = step0          # capture records, it does not execute: the replay below is the step
                self._graphs[key] = g
            finally:
                self._dyn_on = False
        g.replay()
        self.global_step += 1
        return self.loss[0]

    def _step_body(self, hard_gumbel):
        self.forward(self._img_static, return_recon_loss=True, hard_gumbel=hard_gumbel, temperature=1.0, noise="preset", need_grad=True)
        self.backward()
        self.optimizer_step(1.0)      # lr / temperature arguments are ignored on the device-scalar path

    def state_dict(self):
        # reference-named variables as plain tensors in a plain dict: loadable with torch.load(weights_only=True)
        return {"vae_variables": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.export_reference().items()},
                "m": self.m.detach().cpu(), "v": self.v.detach().cpu(), "global_step": self.global_step}

    def load_state_dict(self, sd):
        self.load_reference_params({k: (v.numpy() if torch.is_tensor(v) else np.asarray(v))
                                    for k, v in sd["vae_variables"].items()})
        if "m" in sd:
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"])
        self.global_step = int(sd.get("global_step", 0))
