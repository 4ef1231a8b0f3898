"""refshim -- runs the reference's OWN model / optimizer Python on CPU without TensorFlow or mesh-tensorflow.

TEST INFRASTRUCTURE ONLY: imported by tests/ and by tests/golden/make_ref_callsite_golden.py, never by the product
(`dalle-mtf_amd/`), `bench.py`'s timed region or anything that runs on the GPU box -- /root/reference does not exist there.

Why: the oracle (oracle/dalle_oracle.py) is a restatement, and "parity unpinned" because tensorflow==2.4.0 /
mesh_tensorflow==0.1.18 (requirements.txt:1-2) can be neither vendored nor installed and the reference ships no tests or golden
vectors.  The reference's files are nevertheless plain Python over ~60 library calls.  This package provides those calls
(tfshim.py, mtfshim.py: eager PyTorch-CPU restatements of the published semantics) under the module names
`tensorflow.compat.v1` / `mesh_tensorflow`, imports the reference's files FROM WHERE THEY LIE (nothing is copied, nothing is written
there: byte-code caching is off while they are imported) and executes
them: `DALLE.__init__` / `DALLE.forward` (src/dalle_mtf/models.py:141-416) with its `layers.norm`, its `ops.pad` (an
mtf.Operation subclass, lowered through its own `lower()`), `get_optimizer` / `clip_by_global_norm` (src/optimizers.py:11-104).

What the comparison `oracle == reference-over-shim` pins: the reference's call graph -- which operations, in which order, with
which arguments, dimension names, variable names, shapes and initialiser constants (label shift through pad + gather, the
-1e10 mask from range / less / cast, pre-LN blocks, where the biases enter, loss normalisation, /num_microbatches, the clip
multiplier, the warm-up / cosine schedule, Adam without bias correction).  What it does not pin: the third-party primitives
themselves -- they are restated here exactly as Appendix A restates them for the oracle.  DESIGN.md §2 says "call-graph pinned,
primitives unpinned" accordingly.

The golden vectors this produces are committed (tests/golden/ref_callsite_*.npz) so that the comparison runs wherever the tests
run; where /root/reference exists the tests also regenerate them and require bit-equality with the committed files."""
import contextlib
import importlib
import os
import sys
import types

from . import mtfshim, tfshim

DEFAULT_ROOT = os.environ.get("DALLE_REFERENCE_ROOT", "/root/reference")
_ALIAS = "_dalle_mtf_reference"          # the reference's `src` directory, imported as a package under this name
_SHIMMED = ("tensorflow", "tensorflow.compat", "tensorflow.compat.v1", "tensorflow.compat.v2", "tensorflow.python",
            "tensorflow.python.tpu", "tensorflow.python.tpu.tpu_estimator", "mesh_tensorflow",
            "mesh_tensorflow.ops", "mesh_tensorflow.transformer", "mesh_tensorflow.transformer.attention",
            "mesh_tensorflow.layers", "mesh_tensorflow.optimize", "mesh_tensorflow.utils")


def _module(name, namespace):
    m = types.ModuleType(name)
    m.__dict__.update({k: v for k, v in vars(namespace).items() if not k.startswith("__")})
    return m


def available(root=DEFAULT_ROOT):
    return os.path.isfile(os.path.join(root, "src", "dalle_mtf", "models.py"))


@contextlib.contextmanager
def installed(root=DEFAULT_ROOT):
    """sys.modules carries the shims and the aliased reference package inside the block and is restored afterwards"""
    names = list(_SHIMMED) + [k for k in sys.modules if k == _ALIAS or k.startswith(_ALIAS + ".")]
    saved = {k: sys.modules.get(k) for k in names}
    tf1 = _module("tensorflow.compat.v1", tfshim)
    tf_root = types.ModuleType("tensorflow")
    compat = types.ModuleType("tensorflow.compat")
    compat.v1, compat.v2 = tf1, tf1
    tf_root.compat = compat
    mtf = _module("mesh_tensorflow", mtfshim)
    ops = _module("mesh_tensorflow.ops", mtfshim)
    tr = types.ModuleType("mesh_tensorflow.transformer")
    att = _module("mesh_tensorflow.transformer.attention", mtfshim.transformer.attention)
    tr.attention = att
    tr.utils = mtfshim.transformer.utils
    mtf.transformer = tr
    mtf.ops = ops
    mods = {"tensorflow": tf_root, "tensorflow.compat": compat, "tensorflow.compat.v1": tf1, "tensorflow.compat.v2": tf1,
            "mesh_tensorflow": mtf, "mesh_tensorflow.ops": ops, "mesh_tensorflow.transformer": tr,
            "mesh_tensorflow.transformer.attention": att, "mesh_tensorflow.layers": _module("mesh_tensorflow.layers", mtfshim.layers),
            "mesh_tensorflow.optimize": _module("mesh_tensorflow.optimize", mtfshim.optimize),
            "mesh_tensorflow.utils": _module("mesh_tensorflow.utils", mtfshim.utils)}
    tpu_est = types.ModuleType("tensorflow.python.tpu.tpu_estimator")
    tpu_est.TPUEstimatorSpec = tfshim.TPUEstimatorSpec
    py, py_tpu = types.ModuleType("tensorflow.python"), types.ModuleType("tensorflow.python.tpu")
    py.tpu, py_tpu.tpu_estimator, tf_root.python = py_tpu, tpu_est, py
    mods.update({"tensorflow.python": py, "tensorflow.python.tpu": py_tpu, "tensorflow.python.tpu.tpu_estimator": tpu_est})
    pkg = types.ModuleType(_ALIAS)
    pkg.__path__ = [os.path.join(root, "src")]
    mods[_ALIAS] = pkg
    sys.modules.update(mods)
    tfshim.reset()
    write_bytecode = sys.dont_write_bytecode
    sys.dont_write_bytecode = True        # importing must not leave __pycache__ directories in the (read-only) reference tree
    try:
        yield
    finally:
        sys.dont_write_bytecode = write_bytecode
        for k in list(sys.modules):
            if k == _ALIAS or k.startswith(_ALIAS + "."):
                del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def reference_module(name):
    """inside `installed()`: the reference module src/<name> (e.g. "dalle_mtf.models", "optimizers")"""
    return importlib.import_module(_ALIAS + "." + name)
