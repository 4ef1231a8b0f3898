"""ctypes binding of libdalle_hip.so (include/dalle_hip.h).

Thin plumbing only: torch tensors supply device memory and the current HIP stream; every
function below forwards raw pointers + sizes to the C ABI and raises on a non-zero status.
There is NO CPU fallback: if the library is missing or a GPU is absent the call fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DALLE_HIP_LIB") or os.path.join(_HERE, "libdalle_hip.so")  # override: A/B of two builds
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "dalle_hip.h")

GEMM_BIAS, GEMM_RELU, GEMM_RESIDUAL, GEMM_RELU_MASK, GEMM_OUT_F32, GEMM_ROWSCALE = 1, 2, 4, 8, 16, 32

_lib = None


class DalleHipError(RuntimeError):
    pass


def declared_symbols():
    """Every dmi_* function the public header declares."""
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(dmi_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DalleHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the product path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        for kv in os.environ.get("DALLE_HIP_OPTIONS", "").split(","):   # A/B hook: e.g. DALLE_HIP_OPTIONS=tn8=0,attn_xcd=0
            if "=" in kv:
                if _lib.dmi_set_option(kv.split("=")[0].strip().encode(), int(kv.split("=")[1])) != 0:
                    raise DalleHipError(f"DALLE_HIP_OPTIONS: unknown option {kv!r}")
    return _lib


def _declare(L):
    P, I, L64, F = c_void_p, c_int, c_int64, c_float
    sig = {
        "dmi_last_error_string": (c_char_p, []),
        "dmi_version": (I, []),
        "dmi_get_option": (I, [c_char_p]),
        "dmi_set_option": (I, [c_char_p, I]),
        "dmi_set_debug_buffer": (I, [P]),
        "dmi_embed_fwd": (I, [P, P, P, P, L64, I, I, I, P, P]),
        "dmi_sort_tokens_workspace_bytes": (L64, [L64]),
        "dmi_sort_tokens": (I, [P, P, P, L64, I, P, P]),
        "dmi_embed_bwd_workspace_bytes": (L64, [I, I, I]),
        "dmi_embed_bwd": (I, [P, P, P, P, P, I, I, I, I, P, P]),
        "dmi_layernorm_fwd": (I, [P, P, P, P, P, P, L64, I, F, P]),
        "dmi_layernorm_bwd_workspace_bytes": (L64, [L64, I]),
        "dmi_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, L64, I, P]),
        "dmi_layernorm_bwd_finish_batch": (I, [P, P, P, P, I, I, P]),
        "dmi_gemm_nt": (I, [P, I, P, I, P, I, I, I, I, I, P, P, P, P, P]),
        "dmi_gemm_nt_splitk_workspace_bytes": (L64, [I, I, I]),
        "dmi_gemm_nt_splitk": (I, [P, I, P, I, P, I, I, I, I, P, P, P]),
        "dmi_gemm_tn_workspace_bytes": (L64, [I, I, I]),
        "dmi_gemm_tn": (I, [P, I, P, I, P, P, P, I, I, I, P, P, P, P]),
        "dmi_gemm_tn_group": (I, [P, I, I, P, P, P]),
        "dmi_gemm_tn_group_plan": (I, [P, P, I, I]),
        "dmi_reduce_slabs_batch": (I, [P, I, P]),
        "dmi_colsum_workspace_bytes": (L64, [L64, I]),
        "dmi_colsum": (I, [P, I, P, L64, I, P, P]),
        "dmi_transpose_bf16": (I, [P, P, I, I, I, P]),
        "dmi_attention_fwd": (I, [P, P, P, I, I, I, P]),
        "dmi_attention_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
        "dmi_attention_decode": (I, [P, P, P, I, I, I, I, P, P]),
        "dmi_label_logit": (I, [P, I, P, I, P, P, P, P, L64, I, I, P]),
        "dmi_gemm_nt_softmax_partials": (L64, [I]),
        "dmi_gemm_nt_softmax": (I, [P, I, P, I, P, P, P, I, P, I, I, I, P]),
        "dmi_softmax_finish": (I, [P, I, P, P, P, P, I, P, I, P, P, I, I, P, P, P, P, P, L64, I, I, F, P]),
        "dmi_shift_labels": (I, [P, P, I, I, I, P]),
        "dmi_cross_entropy": (I, [P, I, P, P, P, L64, I, F, P]),
        "dmi_sum_f32": (I, [P, L64, F, P, P]),
        "dmi_assemble_tokens": (I, [P, P, P, I, I, I, I, I, P]),
        "dmi_sample_tokens": (I, [P, I, P, I, I, F, I, ctypes.c_uint64, P, I, P, I, I, P, P, I, I, P]),
        "dmi_ln_gemm_nt": (I, [P, I, P, P, F, P, I, P, I, I, I, I, I, P, P]),
        "dmi_logits_f32": (I, [P, I, P, P, I, I, P]),
        "dmi_gemm_nt_ln": (I, [P, I, P, I, P, I, I, I, I, P, P, P, P, F, P, I, P, P, P]),
        "dmi_gemm_nt_lnbwd_parts": (I, [I]),
        "dmi_gemm_nt_lnbwd": (I, [P, I, P, I, I, I, I, P, P, P, P, P, P, P, P, I, P, P]),
        "dmi_layernorm_bwd_finish_parts": (I, [P, I, P, P, I, P]),
        "dmi_relu_bits_bytes": (L64, [I, I]),
        "dmi_relu_bits_auto": (I, [I, I, I]),
        "dmi_gemm_nt_ln_auto": (I, [I, I, I]),
        "dmi_gemm_nt_relu_bits": (I, [P, I, P, I, P, I, I, I, I, P, P, P]),
        "dmi_gemm_nt_mask_bits": (I, [P, I, P, I, P, I, I, I, I, P, P]),
        "dmi_sumsq_workspace_bytes": (L64, [L64]),
        "dmi_sumsq": (I, [P, L64, P, P, P]),
        "dmi_adam_step": (I, [P, P, P, P, P, L64, P, F, F, F, F, F, F, F, P, P]),
        "dmi_cast_f32_bf16": (I, [P, P, L64, P]),
        "dmi_transpose_bf16_padded": (I, [P, P, I, I, I, P]),
        "dmi_transpose_bf16_batch": (I, [P, P, P, I, L64, P]),
        "dmi_im2col": (I, [P, P, I, I, I, I, I, I, I, I, P, P, I, P]),
        "dmi_conv_gemm_nt": (I, [P, I, I, I, I, I, I, I, I, P, P, P, I, P, I, I, I, P, P, P, P]),
        "dmi_conv_wgrad_tn_workspace_bytes": (L64, [I, I, I]),
        "dmi_conv_wgrad_tn": (I, [P, I, I, I, I, I, I, I, I, P, P, P, I, I, P, P, P, P, P, P]),
        "dmi_weight_gather": (I, [P, P, I, I, I, P, I, P]),
        "dmi_weight_gather_batch": (I, [P, P, P, I, L64, P]),
        "dmi_pixel_interleave": (I, [P, P, I, I, I, I, P]),
        "dmi_pad_channels": (I, [P, P, L64, I, I, P]),
        "dmi_unpad_channels": (I, [P, P, L64, I, I, P]),
        "dmi_gumbel_softmax_fwd": (I, [P, P, P, P, P, L64, I, F, I, P, P]),
        "dmi_gumbel_softmax_bwd": (I, [P, P, P, L64, I, F, P, P]),
        "dmi_mse_workspace_bytes": (L64, []),
        "dmi_mse_loss": (I, [P, P, P, P, L64, I, I, F, P, P]),
        "dmi_add_f32": (I, [P, P, L64, P]),
        "dmi_comm_load": (I, [c_char_p]),
        "dmi_comm_unique_id_bytes": (I, []),
        "dmi_comm_unique_id": (I, [P]),
        "dmi_comm_init": (I, [P, I, I, P]),
        "dmi_comm_destroy": (I, [P]),
        "dmi_allreduce_bucket": (I, [P, P, L64, P]),
        "dmi_comm_broadcast_f32": (I, [P, P, L64, I, P]),
        "dmi_conv2d_f32": (I, [P, I, I, I, I, I, I, I, I, P, P, P, P, P, P, I, I, P]),
        "dmi_space_to_depth_f32": (I, [P, P, I, I, I, I, I, I, P]),
        "dmi_depth_to_space_f32": (I, [P, P, I, I, I, I, I, I, P]),
        "dmi_crc32c": (ctypes.c_uint32, [c_char_p, ctypes.c_size_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    # optional (VAE) entry points are declared by dalle_hip.vae when present


def _check(rc, what):
    if rc != 0:
        raise DalleHipError(f"{what}: status {rc}: {lib().dmi_last_error_string().decode()}")


def _p(t):
    """device pointer of a tensor, or a raw device address (int) for sub-buffer views (row chunks)."""
    if t is None:
        return 0
    return t if isinstance(t, int) else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not isinstance(t, int) and not t.is_cuda:
            raise DalleHipError("dalle_hip ops need CUDA(HIP) tensors; there is no CPU fallback")


def set_option(name: str, value: int):
    _check(lib().dmi_set_option(name.encode(), int(value)), "set_option")


def get_option(name: str) -> int:
    return lib().dmi_get_option(name.encode())


# ------------------------------------------------------------------ wrappers (torch tensors in/out)

def embed_fwd(tokens, wte, wpe, x, S, d, vocab, pos_dev=None):
    _dev(tokens, wte, wpe, x)
    _check(lib().dmi_embed_fwd(_p(tokens), _p(wte), _p(wpe), _p(x), tokens.numel(), S, d, vocab, _p(pos_dev), _stream()), "embed_fwd")


def sort_tokens_workspace_bytes(n):
    return int(lib().dmi_sort_tokens_workspace_bytes(n))


def sort_tokens(tokens, sorted_tokens, perm, n, vocab, ws):
    """stable sort of the token ids: sorted_tokens ascending, perm = source positions (int32 device tensors)."""
    _dev(tokens, sorted_tokens, perm, ws)
    _check(lib().dmi_sort_tokens(_p(tokens), _p(sorted_tokens), _p(perm), n, vocab, _p(ws), _stream()), "sort_tokens")


def embed_bwd_workspace_bytes(B, S, d):
    return int(lib().dmi_embed_bwd_workspace_bytes(B, S, d))


def embed_bwd(sorted_tokens, perm, dx, dwte, dwpe, B, S, d, vocab, ws):
    _dev(sorted_tokens, perm, dx, dwte, dwpe, ws)
    _check(lib().dmi_embed_bwd(_p(sorted_tokens), _p(perm), _p(dx), _p(dwte), _p(dwpe), B, S, d, vocab, _p(ws), _stream()),
           "embed_bwd")


def layernorm_fwd(x, g, b, y, mean, rstd, rows, d, eps=1e-5):
    _dev(x, g, b, y, mean, rstd)
    _check(lib().dmi_layernorm_fwd(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), rows, d, eps, _stream()), "layernorm_fwd")


def layernorm_bwd_workspace_bytes(rows, d):
    return lib().dmi_layernorm_bwd_workspace_bytes(rows, d)


def layernorm_bwd(dy, x, g, mean, rstd, dres, dx, dg, db, ws, rows, d):
    """dg = db = None: deferred -- the partials stay in `ws` until layernorm_bwd_finish_batch"""
    _dev(dy, x, g, mean, rstd, dx, ws)
    _check(lib().dmi_layernorm_bwd(_p(dy), _p(x), _p(g), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dg), _p(db),
                                   _p(ws), rows, d, _stream()), "layernorm_bwd")


def layernorm_bwd_finish_batch(items, d):
    """items: [(workspace, dg, db, rows)] of deferred layernorm_bwd calls (at most 16, one width)"""
    import ctypes
    n = len(items)
    PA, LA = ctypes.c_void_p * n, ctypes.c_int64 * n
    _check(lib().dmi_layernorm_bwd_finish_batch(PA(*[_p(i[0]) for i in items]), PA(*[_p(i[1]) for i in items]), PA(*[_p(i[2]) for i in items]),
                                                LA(*[int(i[3]) for i in items]), n, d, _stream()), "layernorm_bwd_finish_batch")


def gemm_nt(A, lda, Bt, ldb, C, ldc, M, N, K, flags=0, bias=None, residual=None, relu_src=None, rowscale=None):
    _dev(A, Bt, C)
    _check(lib().dmi_gemm_nt(_p(A), lda, _p(Bt), ldb, _p(C), ldc, M, N, K, flags, _p(bias), _p(residual),
                             _p(relu_src), _p(rowscale), _stream()), "gemm_nt")


def gemm_nt_ln(A, lda, Bt, ldb, C, ldc, M, N, K, gamma, beta, Y, ldy, mean, rstd, bias=None, residual=None, eps=1e-5):
    """C = bf16(A . Bt^T + bias + residual) and, in the same pass, Y = LayerNorm(C) * gamma + beta with the row statistics
    (N = 512: full-row tiles).  Raises DalleHipError (unsupported) for other widths: the caller keeps gemm_nt + layernorm_fwd."""
    _dev(A, Bt, C, gamma, beta, Y, mean, rstd)
    assert mean.dtype == torch.float32 and rstd.dtype == torch.float32 and mean.numel() >= M and rstd.numel() >= M
    _check(lib().dmi_gemm_nt_ln(_p(A), lda, _p(Bt), ldb, _p(C), ldc, M, N, K, _p(bias), _p(residual), _p(gamma), _p(beta), float(eps),
                                _p(Y), ldy, _p(mean), _p(rstd), _stream()), "gemm_nt_ln")


def gemm_nt_lnbwd_parts(M):
    return int(lib().dmi_gemm_nt_lnbwd_parts(M))


def gemm_nt_lnbwd(A, lda, Bt, ldb, M, N, K, x, gamma, mean, rstd, dres, dx, part, dg=None, db=None, B2=None, ldb2=0, C2=None):
    """dx = LayerNorm-backward(x; gamma, mean, rstd)(A . Bt^T) + dres in one pass (N = 512: full-row tiles); part: fp32
    [gemm_nt_lnbwd_parts(M), 2 N] partial gain | bias gradients.  dg / db given: the partials are summed into them right away.
    B2 / C2: the product that consumes dx chained in the same launch, C2 = dx . B2^T."""
    _dev(A, Bt, x, gamma, mean, rstd, dres, dx, part, dg, db, B2, C2)
    P = gemm_nt_lnbwd_parts(M)
    assert part.dtype == torch.float32 and part.numel() >= P * 2 * N
    _check(lib().dmi_gemm_nt_lnbwd(_p(A), lda, _p(Bt), ldb, M, N, K, _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(part),
                                   _p(B2), ldb2, _p(C2), _stream()), "gemm_nt_lnbwd")
    if dg is not None:
        _check(lib().dmi_layernorm_bwd_finish_parts(_p(part), P, _p(dg), _p(db), N, _stream()), "layernorm_bwd_finish_parts")


def relu_bits_bytes(M, N):
    return int(lib().dmi_relu_bits_bytes(M, N))


def gemm_nt_ln_auto(M, N, K):
    """True where gemm_nt_ln / gemm_nt_lnbwd accept the shape and the library would pick the full-row kernel itself."""
    return bool(lib().dmi_gemm_nt_ln_auto(M, N, K))


def relu_bits_auto(M, N, K):
    """True where the library's dispatch runs [M, N, K] on the kernel that has the bit forms of the ReLU mask"""
    return bool(lib().dmi_relu_bits_auto(M, N, K))


def gemm_nt_relu_bits(A, lda, Bt, ldb, C, ldc, M, N, K, bias, bits):
    """C = bf16(relu(A . Bt^T + bias)); bits (uint8, relu_bits_bytes(M, N)) = one bit per output, C > 0"""
    _dev(A, Bt, C, bias, bits)
    assert bits.dtype == torch.uint8 and bits.numel() >= relu_bits_bytes(M, N)
    _check(lib().dmi_gemm_nt_relu_bits(_p(A), lda, _p(Bt), ldb, _p(C), ldc, M, N, K, _p(bias), _p(bits), _stream()), "gemm_nt_relu_bits")


def gemm_nt_mask_bits(A, lda, Bt, ldb, C, ldc, M, N, K, bits):
    """C = bf16(A . Bt^T) where the bit written by gemm_nt_relu_bits is set, 0 elsewhere (= gemm_nt with GEMM_RELU_MASK, relu_src = h)"""
    _dev(A, Bt, C, bits)
    assert bits.dtype == torch.uint8 and bits.numel() >= relu_bits_bytes(M, N)
    _check(lib().dmi_gemm_nt_mask_bits(_p(A), lda, _p(Bt), ldb, _p(C), ldc, M, N, K, _p(bits), _stream()), "gemm_nt_mask_bits")


def ln_gemm_nt(X, ldx, gamma, beta, Bt, ldb, C, ldc, M, N, K, flags=0, bias=None, eps=1e-5):
    """C = LayerNorm(X) . Bt^T (+ bias)(ReLU) for the decode step (M <= 32)"""
    _dev(X, gamma, beta, Bt, C)
    _check(lib().dmi_ln_gemm_nt(_p(X), ldx, _p(gamma), _p(beta), eps, _p(Bt), ldb, _p(C), ldc, M, N, K, flags, _p(bias), _stream()),
           "ln_gemm_nt")


def attention_decode(qkv, o, B, H, S, pos, fresh=None, pos_dev=None):
    """one query position against the K/V cache held in the [B*S, 3d] projection buffer.  fresh=None: row pos already
    written; fresh = [B, 3d] staging buffer: the kernel moves it into row pos.  pos_dev (int32 [1] on the device) overrides
    pos -- the graph-replayable form."""
    _dev(qkv, o)
    if fresh is not None:
        _dev(fresh)
    if pos_dev is not None:
        _dev(pos_dev)
        assert pos_dev.dtype == torch.int32
    _check(lib().dmi_attention_decode(_p(qkv), _p(fresh), _p(o), B, H, S, int(pos), _p(pos_dev), _stream()), "attention_decode")


def gemm_nt_splitk_workspace_bytes(M, N, nsplit):
    return lib().dmi_gemm_nt_splitk_workspace_bytes(M, N, nsplit)


def gemm_nt_splitk(A, lda, Bt, ldb, C, M, N, K, nsplit, ws, rowscale=None):
    _dev(A, Bt, C, ws)
    _check(lib().dmi_gemm_nt_splitk(_p(A), lda, _p(Bt), ldb, _p(C), M, N, K, nsplit, _p(rowscale), _p(ws), _stream()),
           "gemm_nt_splitk")


def gemm_tn_workspace_bytes(M, I, J):
    return lib().dmi_gemm_tn_workspace_bytes(M, I, J)


class ReduceItem(ctypes.Structure):
    """dmi_reduce_item (include/dalle_hip.h)."""
    _fields_ = [("slabs", c_void_p), ("out", c_void_p), ("nsplit", c_int), ("n4", c_int64)]


class DeferredReduces:
    """collects the slab reduces that dmi_gemm_tn calls leave behind; run() launches them as one kernel"""

    def __init__(self, capacity=16):
        self.items = (ReduceItem * capacity)()
        self.n, self.capacity = 0, capacity

    def run(self):
        if self.n:
            _check(lib().dmi_reduce_slabs_batch(ctypes.byref(self.items), self.n, _stream()), "reduce_slabs_batch")
        self.n = 0


def gemm_tn(X, ldx, dY, ldy, dW, M, I, J, ws, dbias=None, bias_weights=None, deferred: "DeferredReduces" = None):
    """deferred: the final slab reduces are appended to it instead of being launched (ws must then be exclusive to this call
    until deferred.run())."""
    _dev(X, dY, dW, ws, dbias, bias_weights)
    if deferred is None:
        _check(lib().dmi_gemm_tn(_p(X), ldx, _p(dY), ldy, _p(dW), _p(dbias), _p(bias_weights), M, I, J, _p(ws), None, None,
                                 _stream()), "gemm_tn")
        return
    assert deferred.n + 2 <= deferred.capacity
    cnt = c_int(0)
    slot = ctypes.byref(deferred.items, deferred.n * ctypes.sizeof(ReduceItem))
    _check(lib().dmi_gemm_tn(_p(X), ldx, _p(dY), ldy, _p(dW), _p(dbias), _p(bias_weights), M, I, J, _p(ws), slot,
                             ctypes.byref(cnt), _stream()), "gemm_tn")
    deferred.n += cnt.value


class TnProblem(ctypes.Structure):
    """dmi_tn_problem (include/dalle_hip.h)."""
    _fields_ = [("X", c_void_p), ("ldx", c_int), ("dY", c_void_p), ("ldy", c_int), ("dW", c_void_p), ("dbias", c_void_p),
                ("bias_weights", c_void_p), ("I", c_int), ("J", c_int), ("workspace", c_void_p)]


def gemm_tn_group_plan(shapes, M):
    """row splits of the grouped launch of weight gradients with shapes [(I, J), ...] when it runs on 128 x 256 tiles, 0 otherwise"""
    n = len(shapes)
    Is, Js = (c_int * n)(*[s[0] for s in shapes]), (c_int * n)(*[s[1] for s in shapes])
    return int(lib().dmi_gemm_tn_group_plan(Is, Js, n, M))


def gemm_tn_group(problems, M, deferred: "DeferredReduces" = None):
    """several weight gradients over the same M rows in one launch.  problems: dicts with X, ldx, dY, ldy, dW, I, J, ws and
    optionally dbias, bias_weights (as gemm_tn); deferred as in gemm_tn."""
    n = len(problems)
    arr = (TnProblem * n)()
    for k, q in enumerate(problems):
        _dev(q["X"], q["dY"], q["dW"], q["ws"], q.get("dbias"), q.get("bias_weights"))
        assert q["ws"].numel() * q["ws"].element_size() >= gemm_tn_workspace_bytes(M, q["I"], q["J"])
        arr[k] = TnProblem(_p(q["X"]), q["ldx"], _p(q["dY"]), q["ldy"], _p(q["dW"]), _p(q.get("dbias")), _p(q.get("bias_weights")),
                           q["I"], q["J"], _p(q["ws"]))
    if deferred is None:
        _check(lib().dmi_gemm_tn_group(ctypes.byref(arr), n, M, None, None, _stream()), "gemm_tn_group")
        return
    assert deferred.n + 2 * n <= deferred.capacity
    cnt = c_int(0)
    slot = ctypes.byref(deferred.items, deferred.n * ctypes.sizeof(ReduceItem))
    _check(lib().dmi_gemm_tn_group(ctypes.byref(arr), n, M, slot, ctypes.byref(cnt), _stream()), "gemm_tn_group")
    deferred.n += cnt.value


def colsum_workspace_bytes(M, N):
    return lib().dmi_colsum_workspace_bytes(M, N)


def colsum(Y, ldy, out, M, N, ws):
    _dev(Y, out, ws)
    _check(lib().dmi_colsum(_p(Y), ldy, _p(out), M, N, _p(ws), _stream()), "colsum")


def transpose(inp, out, batch, R, C):
    _dev(inp, out)
    _check(lib().dmi_transpose_bf16(_p(inp), _p(out), batch, R, C, _stream()), "transpose")


def set_debug_buffer(t):
    """tools only: u64 device tensor [blocks, 5] receiving per-block phase timestamps of the 256x128 NT kernel (None: off)."""
    fn = lib().dmi_set_debug_buffer
    fn.restype, fn.argtypes = c_int, [c_void_p]
    fn(_p(t))


def transpose_batch(in_base, out_base, table, n, total_tiles):
    """table: int64 device tensor [n,5] = {in_off, out_off, R, C, first_tile}."""
    _dev(in_base, out_base, table)
    _check(lib().dmi_transpose_bf16_batch(_p(in_base), _p(out_base), _p(table), n, total_tiles, _stream()), "transpose_batch")


def attention_fwd(qkv, o, lse, B, H, S):
    _dev(qkv, o, lse)
    _check(lib().dmi_attention_fwd(_p(qkv), _p(o), _p(lse), B, H, S, _stream()), "attention_fwd")


def attention_bwd(qkv, o, d_o, lse, scratch, dqkv, B, H, S):
    _dev(qkv, o, d_o, lse, scratch, dqkv)
    _check(lib().dmi_attention_bwd(_p(qkv), _p(o), _p(d_o), _p(lse), _p(scratch), _p(dqkv), B, H, S, _stream()), "attention_bwd")


def shift_labels(tokens, labels, B, S, eos):
    _dev(tokens, labels)
    _check(lib().dmi_shift_labels(_p(tokens), _p(labels), B, S, eos, _stream()), "shift_labels")


def cross_entropy(z, ldz, labels, loss_rows, lse, M, V, dz_scale):
    _dev(z, labels, loss_rows)
    _check(lib().dmi_cross_entropy(_p(z), ldz, _p(labels), _p(loss_rows), _p(lse), M, V, dz_scale, _stream()), "cross_entropy")


def label_logit(X, ldx, Wt, ldw, bias, labels, zl, flag, M, K, V):
    _dev(X, Wt, bias, labels, zl, flag)
    _check(lib().dmi_label_logit(_p(X), ldx, _p(Wt), ldw, _p(bias), _p(labels), _p(zl), _p(flag), M, K, V, _stream()), "label_logit")


def gemm_nt_softmax_partials(N):
    return int(lib().dmi_gemm_nt_softmax_partials(N))


def gemm_nt_softmax(X, ldx, Wt, ldw, bias, rowshift, E, lde, rowsum_part, M, N, K):
    _dev(X, Wt, bias, E, rowsum_part)
    _check(lib().dmi_gemm_nt_softmax(_p(X), ldx, _p(Wt), ldw, _p(bias), _p(rowshift), _p(E), lde, _p(rowsum_part), M, N, K,
                                     _stream()), "gemm_nt_softmax")


def softmax_finish(rowsum_part, nparts, label_logit, rowshift, labels, X, ldx, Wt, ldw, bias, E, lde, N, loss_rows, rowscale,
                   rowscale_bf16, Xs, flag, M, K, V, dz_scale):
    _dev(rowsum_part, label_logit, rowshift, labels, X, Wt, bias, E, loss_rows, flag)
    _check(lib().dmi_softmax_finish(_p(rowsum_part), nparts, _p(label_logit), _p(rowshift), _p(labels), _p(X), ldx, _p(Wt), ldw,
                                    _p(bias), _p(E), lde, N,
                                    _p(loss_rows), _p(rowscale), _p(rowscale_bf16), _p(Xs), _p(flag), M, K, V, float(dz_scale),
                                    _stream()), "softmax_finish")


def sum_f32(x, n, scale, out):
    _dev(x, out)
    _check(lib().dmi_sum_f32(_p(x), n, scale, _p(out), _stream()), "sum_f32")


def assemble_tokens(text, vae_logits, tokens_out, B, T, P, C, text_vocab):
    _dev(text, vae_logits, tokens_out)
    _check(lib().dmi_assemble_tokens(_p(text), _p(vae_logits), _p(tokens_out), B, T, P, C, text_vocab, _stream()), "assemble_tokens")


def sample_tokens(z, ldz, bias, B, nv, temperature=1.0, top_k=0, seed=0, pos=0, token_offset=0, next_tok=None, out=None,
                  out_col0=0, params_dev=None, pos_dev=None, advance=False):
    """next image token per row of head logits z bf16 [B, ldz] (+ bias bf16 [nv]): temperature / top-k / greedy; the draw is
    a pure function of (seed, position, row).  params_dev (uint32 [4]) / pos_dev (int32) override the by-value settings.
    pos_dev is int32 [1] (the position) -- or, with advance=True, int32 [2]: [position, arrival counter]; the kernel's last
    block to finish increments [0] and resets [1], which must be zero on entry (include/dalle_hip.h)."""
    _dev(z)
    assert z.dtype == torch.bfloat16 and (bias is None or bias.dtype == torch.bfloat16)
    if pos_dev is not None:
        assert pos_dev.dtype == torch.int32 and pos_dev.numel() >= (2 if advance else 1), \
            "sample_tokens: pos_dev must be int32 [2] ([position, zeroed counter]) when advance=True, int32 [1] otherwise"
    for t in (bias, next_tok, out, params_dev, pos_dev):
        if t is not None:
            _dev(t)
    out_ld = int(out.shape[1]) if out is not None else 0
    _check(lib().dmi_sample_tokens(_p(z), ldz, _p(bias), B, nv, float(temperature), int(top_k), int(seed) & (2 ** 64 - 1),
                                   _p(params_dev), int(pos), _p(pos_dev), int(bool(advance)), int(token_offset), _p(next_tok), _p(out), out_ld,
                                   int(out_col0), _stream()), "sample_tokens")


def logits_f32(z, ldz, bias, out, B, nv):
    """out fp32 [B, nv] = float(z bf16 [B, ldz][:, :nv]) + float(bias bf16 [nv] or None)"""
    _dev(z, out)
    assert z.dtype == torch.bfloat16 and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= B * nv
    if bias is not None:
        _dev(bias)
        assert bias.dtype == torch.bfloat16
    _check(lib().dmi_logits_f32(_p(z), int(ldz), _p(bias), _p(out), int(B), int(nv), _stream()), "logits_f32")


def sumsq_workspace_bytes(n):
    return lib().dmi_sumsq_workspace_bytes(n)


def sumsq(g, n, out, ws):
    _dev(g, out, ws)
    _check(lib().dmi_sumsq(_p(g), n, _p(out), _p(ws), _stream()), "sumsq")


def adam_step(p, g, m, v, p_bf16, n, gnorm_sq, clip, lr, beta1, beta2, eps, wd, grad_scale=1.0, lr_dev=None):
    """lr_dev: optional 1-element fp32 DEVICE tensor the kernel reads the learning rate from (HIP-graph replay)"""
    _dev(p, g, m, v)
    _check(lib().dmi_adam_step(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(gnorm_sq), clip, lr, beta1, beta2, eps,
                               wd, grad_scale, _p(lr_dev), _stream()), "adam_step")


def cast_f32_bf16(inp, out, n):
    _dev(inp, out)
    _check(lib().dmi_cast_f32_bf16(_p(inp), _p(out), n, _stream()), "cast_f32_bf16")


# ------------------------------------------------------------------ VAE wrappers

def transpose_padded(inp, out, R_valid, R_pitch, C):
    _dev(inp, out)
    _check(lib().dmi_transpose_bf16_padded(_p(inp), _p(out), R_valid, R_pitch, C, _stream()), "transpose_padded")


def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def conv_gemm_nt(x, B, H, W, C, Ho, Wo, stride, taps, Wt, ldw, out, ldc, N, flags=0, bias=None, residual=None, relu_src=None):
    """implicit-im2col convolution GEMM (include/dalle_hip.h: dmi_conv_gemm_nt); taps = [(dy, dx), ...]; C % 64 == 0."""
    _dev(x, Wt, out)
    dy, dx = _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps])
    _check(lib().dmi_conv_gemm_nt(_p(x), B, H, W, C, Ho, Wo, stride, len(taps), ctypes.cast(dy, c_void_p),
                                  ctypes.cast(dx, c_void_p), _p(Wt), ldw, _p(out), ldc, N, flags,
                                  _p(bias), _p(residual), _p(relu_src), _stream()), "conv_gemm_nt")


def conv_wgrad_tn_workspace_bytes(M, K, N):
    return int(lib().dmi_conv_wgrad_tn_workspace_bytes(M, K, N))


def conv_wgrad_tn(x, B, H, W, C, Ho, Wo, stride, taps, dY, ldy, N, dW, ws, dbias=None, deferred: "DeferredReduces" = None):
    """implicit-im2col weight gradient (include/dalle_hip.h: dmi_conv_wgrad_tn); Ho, Wo powers of two, C % 64 == 0.
    deferred: as gemm_tn (ws must then be exclusive to this call until deferred.run())."""
    _dev(x, dY, dW, ws)
    dy, dx = _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps])
    if deferred is None:
        _check(lib().dmi_conv_wgrad_tn(_p(x), B, H, W, C, Ho, Wo, stride, len(taps), ctypes.cast(dy, c_void_p),
                                       ctypes.cast(dx, c_void_p), _p(dY), ldy, N, _p(dW), _p(dbias), _p(ws), None, None, _stream()),
               "conv_wgrad_tn")
        return
    assert deferred.n + 2 <= deferred.capacity
    cnt = c_int(0)
    slot = ctypes.byref(deferred.items, deferred.n * ctypes.sizeof(ReduceItem))
    _check(lib().dmi_conv_wgrad_tn(_p(x), B, H, W, C, Ho, Wo, stride, len(taps), ctypes.cast(dy, c_void_p),
                                   ctypes.cast(dx, c_void_p), _p(dY), ldy, N, _p(dW), _p(dbias), _p(ws), slot, ctypes.byref(cnt),
                                   _stream()), "conv_wgrad_tn")
    deferred.n += cnt.value


def im2col(x, out, B, H, W, C, Ho, Wo, stride, taps, ldo):
    """taps: list of (dy, dx) host tuples."""
    _dev(x, out)
    dy, dx = _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps])
    _check(lib().dmi_im2col(_p(x), _p(out), B, H, W, C, Ho, Wo, stride, len(taps), ctypes.cast(dy, c_void_p),
                            ctypes.cast(dx, c_void_p), ldo, _stream()), "im2col")


def weight_gather(inp, out, A, Bn, idx, ldo):
    _dev(inp, out)
    ia = _iarr(idx)
    _check(lib().dmi_weight_gather(_p(inp), _p(out), A, Bn, len(idx), ctypes.cast(ia, c_void_p), ldo, _stream()), "weight_gather")


def pixel_interleave(in4, out, B, Ht, Wt, C):
    _dev(in4, out)
    _check(lib().dmi_pixel_interleave(_p(in4), _p(out), B, Ht, Wt, C, _stream()), "pixel_interleave")


def pad_channels(inp, out, N, Cin, Cp):
    _dev(inp, out)
    _check(lib().dmi_pad_channels(_p(inp), _p(out), N, Cin, Cp, _stream()), "pad_channels")


def unpad_channels(inp, out, N, Cin, Cp):
    _dev(inp, out)
    _check(lib().dmi_unpad_channels(_p(inp), _p(out), N, Cin, Cp, _stream()), "unpad_channels")


def weight_gather_batch(in_base, out_base, table, n, total_blocks):
    _dev(in_base, out_base, table)
    _check(lib().dmi_weight_gather_batch(_p(in_base), _p(out_base), _p(table), n, total_blocks, _stream()), "weight_gather_batch")


def gumbel_softmax_fwd(logits, u, y, y_soft, index, M, T, temperature, hard, temperature_dev=None):
    _dev(logits, u, y, y_soft, index)
    _check(lib().dmi_gumbel_softmax_fwd(_p(logits), _p(u), _p(y), _p(y_soft), _p(index), M, T, float(temperature), int(bool(hard)),
                                        _p(temperature_dev), _stream()), "gumbel_softmax_fwd")


def gumbel_softmax_bwd(dy, y_soft, dlogits, M, T, temperature, temperature_dev=None):
    _dev(dy, y_soft, dlogits)
    _check(lib().dmi_gumbel_softmax_bwd(_p(dy), _p(y_soft), _p(dlogits), M, T, float(temperature), _p(temperature_dev), _stream()),
           "gumbel_softmax_bwd")


def mse_workspace_bytes():
    return lib().dmi_mse_workspace_bytes()


def mse_loss(img, outp, dout, loss, N, Cin, Cp, grad_scale, ws):
    _dev(img, outp, dout, loss, ws)
    _check(lib().dmi_mse_loss(_p(img), _p(outp), _p(dout), _p(loss), N, Cin, Cp, float(grad_scale), _p(ws), _stream()), "mse_loss")


def add_f32(dst, src, n):
    _dev(dst, src)
    _check(lib().dmi_add_f32(_p(dst), _p(src), n, _stream()), "add_f32")


def conv2d_f32(x, B, H, W, C, Ho, Wo, stride, taps, Wk, bias, residual, out, N, relu=False):
    """fp32 convolution on the exact-fp32 matrix cores (tokenising encoder); taps = [(dy, dx), ...]."""
    _dev(x, Wk, bias, residual, out)
    dy, dx = _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps])
    _check(lib().dmi_conv2d_f32(_p(x), B, H, W, C, Ho, Wo, stride, len(taps), ctypes.cast(dy, c_void_p), ctypes.cast(dx, c_void_p),
                                _p(Wk), _p(bias), _p(residual), _p(out), N, int(bool(relu)), _stream()), "conv2d_f32")


def space_to_depth_f32(img, stacked, B, Hs, Ws, C, s, Cp):
    _dev(img, stacked)
    _check(lib().dmi_space_to_depth_f32(_p(img), _p(stacked), B, Hs, Ws, C, s, Cp, _stream()), "space_to_depth_f32")


def depth_to_space_f32(stacked, img, B, Hs, Ws, C, s, Cp):
    _dev(stacked, img)
    _check(lib().dmi_depth_to_space_f32(_p(stacked), _p(img), B, Hs, Ws, C, s, Cp, _stream()), "depth_to_space_f32")


# ------------------------------------------------------------------ data-parallel exchange (RCCL behind the C ABI)

def comm_load():
    """bind the librccl this process already maps (PyTorch-ROCm ships its own copy) or the system one."""
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    path = os.environ.get("DALLE_RCCL_LIB") or (cand if os.path.exists(cand) else "")
    _check(lib().dmi_comm_load(path.encode()), "comm_load")


def comm_unique_id() -> bytes:
    comm_load()
    buf = ctypes.create_string_buffer(lib().dmi_comm_unique_id_bytes())
    _check(lib().dmi_comm_unique_id(ctypes.cast(buf, c_void_p)), "comm_unique_id")
    return buf.raw


def comm_init(nranks: int, rank: int, unique_id: bytes) -> int:
    """blocking collective; returns the opaque communicator handle (int)."""
    comm_load()
    out = c_void_p()
    buf = ctypes.create_string_buffer(unique_id, len(unique_id))
    _check(lib().dmi_comm_init(ctypes.byref(out), nranks, rank, ctypes.cast(buf, c_void_p)), "comm_init")
    return out.value


def comm_destroy(comm):
    if comm:
        _check(lib().dmi_comm_destroy(c_void_p(comm)), "comm_destroy")


def allreduce_bucket(comm, g, n, stream=None):
    _dev(g)
    _check(lib().dmi_allreduce_bucket(c_void_p(comm), _p(g), n, _stream() if stream is None else stream), "allreduce_bucket")


def comm_broadcast_f32(comm, buf, n, root=0, stream=None):
    _dev(buf)
    _check(lib().dmi_comm_broadcast_f32(c_void_p(comm), _p(buf), n, root, _stream() if stream is None else stream), "comm_broadcast")
