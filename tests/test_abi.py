"""The drop-in boundary on CPU (no compute): libdalle_hip.so loads, exports every entry point include/dalle_hip.h declares,
reports argument errors through its status code + thread-local message (never an exception or a crash), and its option hooks
answer for every documented name.  The parity tests proper (tests/*_gpu.py) call through the same ctypes bindings."""
import ctypes
import os
import re

import pytest

import dalle_hip as dh

HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "dalle_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 60 and names == sorted(dh.declared_symbols())
    L = ctypes.CDLL(dh.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert dh.lib().dmi_version() >= 100


def test_every_entry_point_has_a_python_binding_signature():
    L = dh.lib()
    unbound = [n for n in _declared() if getattr(L, n).argtypes is None and n not in ("dmi_version",)]
    # argtypes stays None only for zero-argument entry points
    zero_arg = {"dmi_version", "dmi_last_error_string", "dmi_comm_unique_id_bytes"}
    assert set(unbound) <= zero_arg, sorted(set(unbound) - zero_arg)


def test_argument_errors_come_back_as_status_and_message():
    L = dh.lib()
    rc = L.dmi_gemm_nt(None, 0, None, 0, None, 0, 128, 128, 64, 0, None, None, None, None, None)
    assert rc == -1                                   # DMI_ERR_INVALID
    msg = L.dmi_last_error_string().decode()
    assert "gemm_nt" in msg and "null" in msg.lower()
    rc = L.dmi_attention_fwd(None, None, None, 1, 1, 128, None)
    assert rc == -1 and "attention_fwd" in L.dmi_last_error_string().decode()
    buf = ctypes.create_string_buffer(64)
    rc = L.dmi_attention_fwd(buf, buf, buf, 1, 1, 12, None)    # S must be a multiple of 8
    assert rc == -1 and "multiple of 8" in L.dmi_last_error_string().decode()
    with pytest.raises(dh.DalleHipError):
        dh._check(rc, "attention_fwd")
    # [r06] weighted bias sums need an even M (the weights travel as dwords; found by tools/experiments/r06_stress_tn.py): refused before
    # any launch -- the pointers below are never dereferenced
    fake = ctypes.c_void_p(0x10000)
    rc = L.dmi_gemm_tn(fake, 128, fake, 128, fake, fake, fake, 595, 128, 128, fake, None, None, None)
    assert rc == -1 and "even M" in L.dmi_last_error_string().decode()


def test_option_hooks():
    for name in ("nt4", "nt8", "tn8", "tn8_max_tiles", "tn_tail", "tn_wide", "attn_xcd"):
        v = dh.get_option(name)
        assert v >= 0, name
        dh.set_option(name, v)                      # round trip
        assert dh.get_option(name) == v
    assert dh.get_option("no_such_option") == -1
    with pytest.raises(dh.DalleHipError):
        dh.set_option("no_such_option", 1)


def test_workspace_queries_are_pure_host_functions():
    assert dh.lib().dmi_gemm_tn_workspace_bytes(40960, 512, 2048) > 0
    assert dh.lib().dmi_sort_tokens_workspace_bytes(40960) > 0
    assert dh.lib().dmi_gemm_nt_softmax_partials(50816) == 794
    assert dh.lib().dmi_layernorm_bwd_workspace_bytes(40960, 512) > 0


def test_grouped_weight_gradient_plan():
    """[r06] dmi_gemm_tn_group_plan (host arithmetic only): the row-split count when the grouped launch runs on 128 x 256 tiles, else 0.
    The engine groups all four gradients of a block where this is non-zero (n_embd = 512: 96 tiles x 5 splits) and keeps the separate
    launches elsewhere (n_embd = 1024 / 2048: the union is 384+ tiles, one unsplit launch would run ragged residencies)."""
    four = lambda d: [(4 * d, d), (d, 4 * d), (d, d), (d, 3 * d)]   # noqa: E731
    assert dh.gemm_tn_group_plan(four(512), 40960) == 5
    assert dh.gemm_tn_group_plan(four(512), 2560) == 5
    assert dh.gemm_tn_group_plan(four(512)[:2], 40960) == 8        # the FFN pair: the single launches' split count
    assert dh.gemm_tn_group_plan(four(512)[2:], 40960) == 0        # out-projection + QKV: 16 splits > QKV's 10 slabs
    assert dh.gemm_tn_group_plan(four(1024), 40960) == 0 and dh.gemm_tn_group_plan(four(2048), 8192) == 0
    saved = dh.get_option("tn_wide")
    dh.set_option("tn_wide", 0)
    try:
        assert dh.gemm_tn_group_plan(four(512), 40960) == 0
    finally:
        dh.set_option("tn_wide", saved)
