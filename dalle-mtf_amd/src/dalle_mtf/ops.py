"""Helpers mirroring src/dalle_mtf/ops.py of the reference."""
from collections import namedtuple

import torch

VariableDType = namedtuple("VariableDType", ["master_dtype", "slice_dtype", "activation_dtype"])


def exists(x):
    return x is not None


def get_variable_dtype(bf_16=True):
    """reference ops.py:76-82: checkpoints in master dtype, training copy (Adam) in slice dtype (fp32),
    compute in activation dtype.  The MI355X kernels compute in bf16 with fp32 accumulation; with
    bf_16=False the master stays fp32 but activations are still bf16 (stated deviation)."""
    if bf_16:
        return VariableDType(torch.bfloat16, torch.float32, torch.bfloat16)
    return VariableDType(torch.float32, torch.float32, torch.bfloat16)


def pad(x: torch.Tensor, paddings, dim_name=None, pad_value=0, name=None):
    """reference ops.py:56-68 (CustomPadOperation): constant pad along the LAST axis of an integer tensor.
    The product path uses the device kernel dmi_shift_labels; this host helper exists for API parity."""
    before, after = paddings
    return torch.nn.functional.pad(x, (before, after), value=pad_value)
