"""mtfshim -- the part of `mesh_tensorflow` 0.1.18 that the reference's DALL-E model and optimizer files call, eager over
PyTorch-CPU, so that the reference's OWN Python (src/dalle_mtf/models.py, layers.py, ops.py, src/optimizers.py) executes
unmodified and its results can be compared with the oracle's restatement.

TEST INFRASTRUCTURE ONLY (see oracle/refshim/__init__.py).  mesh_tensorflow is an un-vendored third-party dependency
(requirements.txt:2) that cannot be installed here; each function below restates the published behaviour of the upstream
function of the same name (mesh_tensorflow/ops.py, layers.py, transformer/attention.py, optimize.py; summarised in SURVEY.md
Appendix A.1-A.7) -- names-as-dimensions broadcasting, einsum over named dimensions, one-hot gather, the attention parameter
layout, softmax as exp(x - logsumexp(x)) with a max shift, reduce_mean as sum * (1 / n).  What this buys: the ORDER and the
ARGUMENTS of the operations -- the model as the reference's authors wrote it -- come from the reference's files, not from a
restatement; what it cannot buy: a check of these restated primitives against the real library.

Tensors hold torch values eagerly (float32 for floating tensors -- a bfloat16 tensor holds bf16-representable float32 values --,
int64 for integers); gradients are torch autograd's.  A subclass of `Operation` defined OUTSIDE this module (the reference's
CustomPadOperation, ScalarSummaryOperation) is evaluated lazily through its own `lower()` with a one-device lowering stub."""
import collections
import contextlib
import math
import string
import types

import torch

from . import tfshim as tf


# ---- dimensions and shapes (mesh_tensorflow/ops.py: Dimension, Shape, convert_to_*) -----------------------------------------
Dimension = collections.namedtuple("Dimension", ["name", "size"])


def convert_to_dimension(d):
    if d is None:
        return None
    if isinstance(d, Dimension):
        return d
    name, size = d
    return Dimension(name, size)


class Shape:
    def __init__(self, dims):
        self._dims = [convert_to_dimension(d) for d in dims]
        names = [d.name for d in self._dims]
        if len(set(names)) != len(names):
            raise ValueError("Shape must not have repeated dimensions %s" % (self._dims,))

    dims = property(lambda self: list(self._dims))
    ndims = property(lambda self: len(self._dims))
    dimension_names = property(lambda self: [d.name for d in self._dims])
    to_integer_list = property(lambda self: [d.size for d in self._dims])

    @property
    def size(self):
        n = 1
        for d in self._dims:
            n *= d.size
        return n

    def __repr__(self):
        return "Shape[%s]" % ", ".join("%s=%d" % (d.name, d.size) for d in self._dims)

    def __eq__(self, other):
        return isinstance(other, Shape) and self._dims == other._dims

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(tuple(self._dims))

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __getitem__(self, key):
        r = self._dims[key]
        return Shape(r) if isinstance(key, slice) else r

    def __add__(self, other):
        if isinstance(other, Shape):
            other = other.dims
        if isinstance(other, Dimension):
            other = [other]
        return Shape(self.dims + list(other))

    def __sub__(self, other):
        if other is None:
            return self
        if isinstance(other, Shape):
            other = other.dims
        if isinstance(other, Dimension):
            other = [other]
        for d in other:
            if d not in self._dims:
                raise ValueError("Subtracting a dimension from a shape requires that the shape contain it: %s - %s" % (self, d))
        return Shape([d for d in self._dims if d not in other])

    def get_dim_by_name(self, name):
        for d in self._dims:
            if d.name == name:
                return d
        raise ValueError("Dimension %s not found in %s" % (name, self))

    def rename_dimension(self, old_name, new_name):
        if old_name not in self.dimension_names:
            raise ValueError("Shape %s does not have dimension named %s" % (self, old_name))
        return Shape([Dimension(new_name, d.size) if d.name == old_name else d for d in self._dims])


def convert_to_shape(x):
    if x is None:
        return None
    if isinstance(x, Shape):
        return x
    if isinstance(x, Dimension):
        return Shape([x])
    if isinstance(x, str):       # "data:16,model:2" (mesh shapes, src/model_fns.py:81)
        return Shape([Dimension(p.split(":")[0].strip(), int(p.split(":")[1])) for p in x.replace(";", ",").split(",") if p.strip()])
    return Shape(list(x))


def convert_to_layout_rules(x):
    """ "batch_dim:data" -> [("batch_dim", "data")]  (src/model_fns.py:82)"""
    if isinstance(x, str):
        return [tuple(q.strip() for q in p.split(":")) for p in x.replace(";", ",").split(",") if p.strip()]
    return list(x)


class VariableDType:
    """master (checkpoint) / slice (per-device training copy) / activation (compute) dtypes (src/dalle_mtf/ops.py:76-82)."""

    def __init__(self, master_dtype=tf.float32, slice_dtype=None, activation_dtype=None):
        self.master_dtype = master_dtype
        self.slice_dtype = slice_dtype or master_dtype
        self.activation_dtype = activation_dtype or master_dtype


# ---- graph, mesh, operations, tensors ---------------------------------------------------------------------------------------
class Graph:
    def __init__(self):
        self.operations = []
        self.name_to_variable = collections.OrderedDict()

    @property
    def trainable_variables(self):
        return [v for v in self.name_to_variable.values() if v.trainable]

    @property
    def all_variables(self):
        return list(self.name_to_variable.values())


class Mesh:
    def __init__(self, graph, name, variable_placer=None):
        self.graph, self.name = graph, name


def _round_to(x, dtype):
    """hold `x` at the precision of tf dtype `dtype` (floating tensors live as float32)"""
    if dtype is tf.bfloat16 or dtype is tf.float16:
        return x.to(dtype.torch).to(torch.float32)
    return x


class Operation:
    """Base class the reference subclasses (CustomPadOperation src/dalle_mtf/ops.py:6-53, ScalarSummaryOperation
    src/utils/utils.py:197-213): inputs, outputs, mesh, name; subclasses implement lower()."""

    def __init__(self, inputs, mesh=None, name=None):
        if mesh is None:
            if not inputs:
                raise ValueError("mesh must be specified if no inputs")
            mesh = inputs[0].mesh
        self._inputs = list(inputs)
        self._outputs = []
        self._mesh = mesh
        self._name = name or type(self).__name__
        mesh.graph.operations.append(self)

    inputs = property(lambda self: self._inputs)
    outputs = property(lambda self: self._outputs)
    mesh = property(lambda self: self._mesh)
    graph = property(lambda self: self._mesh.graph)
    name = property(lambda self: self._name)

    def _initialize_splittable_and_unsplittable_dims(self, default_splittability, exception_dims_iterable=None):
        return frozenset(), frozenset()

    def _initialize_all_dimensions_as_splittable(self):
        return frozenset(), frozenset()

    def lower(self, lowering):
        raise NotImplementedError("lower() of %s" % type(self).__name__)


class _LaidOut:
    """one-device stand-in for a LaidOutTensor"""

    def __init__(self, t):
        self.t = t

    tensor_list = property(lambda self: [self.t])

    def to_laid_out_tensor(self):
        return self


class _MeshImpl:
    def tensor_dimension_to_mesh_axis(self, dim):
        return None     # nothing is split: one device

    def slicewise(self, fn, *args):
        return _LaidOut(fn(*[a.t if isinstance(a, _LaidOut) else a for a in args]))


class _Tensors:
    def __getitem__(self, t):
        return _LaidOut(t.value)


class _Lowering:
    """what an Operation.lower() of the reference needs from mtf.Lowering on a single device"""
    tensors = _Tensors()

    def mesh_impl(self, op):
        return _MeshImpl()

    def set_tensor_lowering(self, tensor, laid_out):
        tensor._value = laid_out.t if isinstance(laid_out, _LaidOut) else laid_out


class Tensor:
    """mtf.Tensor(operation, shape, dtype): value filled in eagerly by this module's functions, or on first use through the
    owning operation's lower() (operations defined by the reference)."""

    def __init__(self, operation, shape, dtype, name=None, index=0):
        self._operation = operation
        self._shape = convert_to_shape(shape)
        self._dtype = dtype
        self._name = name or (operation.name + ":" + str(index))
        self._value = None

    operation = property(lambda self: self._operation)
    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)
    mesh = property(lambda self: self._operation.mesh)
    graph = property(lambda self: self._operation.graph)
    name = property(lambda self: self._name)
    size = property(lambda self: self._shape.size)

    @property
    def value(self):
        if self._value is None:
            self._operation.lower(_Lowering())
            if self._value is None:
                raise RuntimeError("%s.lower() did not produce %s" % (type(self._operation).__name__, self._name))
            if tuple(self._value.shape) != tuple(self._shape.to_integer_list):
                raise RuntimeError("%s: lowered value has shape %s, declared %s" % (self._name, tuple(self._value.shape), self._shape))
        return self._value

    def __repr__(self):
        return "Tensor[%s, %s, %s]" % (self._name, self._shape, self._dtype)

    def __add__(self, other):
        return add(self, other)

    def __radd__(self, other):
        return add(self, other)

    def __sub__(self, other):
        return sub(self, other)

    def __rsub__(self, other):
        return sub(other, self)

    def __mul__(self, other):
        return multiply(self, other)

    def __rmul__(self, other):
        return multiply(self, other)

    def __neg__(self):
        return negative(self)

    def __truediv__(self, other):
        return divide(self, other)

    def __rtruediv__(self, other):
        return divide(other, self)


class _Eager(Operation):
    pass


def _make(value, shape, dtype, like=None, mesh=None, name=None):
    """an output tensor of one of this module's own operations"""
    shape = convert_to_shape(shape)
    op = _Eager([like] if like is not None else [], mesh=mesh, name=name or "op")
    t = Tensor(op, shape, dtype)
    assert tuple(value.shape) == tuple(shape.to_integer_list), (name, tuple(value.shape), shape)
    if dtype.is_floating:
        value = _round_to(value.to(torch.float32), dtype)
    elif dtype.is_integer:
        value = value.to(torch.int64)
    t._value = value
    op._outputs = [t]
    return t


def _align(t, out_dims):
    """t's value permuted / unsqueezed to the dimension order `out_dims` (size 1 where t lacks a dimension)"""
    dims = t.shape.dims
    for d in dims:
        if d not in out_dims:
            raise ValueError("dimension %s of %s is not in the output shape %s" % (d, t, out_dims))
    present = [d for d in out_dims if d in dims]
    v = t.value.permute([dims.index(d) for d in present]) if present else t.value
    return v.reshape([d.size if d in dims else 1 for d in out_dims])


def _broadcast_shape(s1, s2, given=None):
    """mesh_tensorflow/ops.py _infer_binary_broadcast_shape: the longer shape first, then the other's missing dimensions"""
    if given is not None:
        return convert_to_shape(given)
    if len(s1.dims) < len(s2.dims):
        return _broadcast_shape(s2, s1)
    lis = list(s1.dims)
    for d in s2.dims:
        if d not in lis:
            lis.append(d)
    return Shape(lis)


def _binary(fn, x1, x2, output_shape=None, output_dtype=None, name=None):
    shape = _broadcast_shape(x1.shape, x2.shape, output_shape)
    v = fn(_align(x1, shape.dims), _align(x2, shape.dims))
    v = v.expand([d.size for d in shape.dims]) if tuple(v.shape) != tuple(shape.to_integer_list) else v
    return _make(v, shape, output_dtype or x1.dtype, like=x1, name=name)


def _unary(fn, x, dtype=None, name=None):
    return _make(fn(x.value), x.shape, dtype or x.dtype, like=x, name=name)


def _scalar(x):
    return x.item() if isinstance(x, torch.Tensor) else x


# elementwise (ScalarAddOperation / ScalarMultiplyOperation when the other operand is a Python number)
def add(x1, x2, output_shape=None, name=None):
    if not isinstance(x2, Tensor):
        return _unary(lambda v: v + _scalar(x2), x1, name="scalar_add")
    if not isinstance(x1, Tensor):
        return add(x2, x1)
    return _binary(lambda a, b: a + b, x1, x2, output_shape, name="add")


def negative(x, name=None):
    return _unary(lambda v: -v, x, name="negative")


def sub(x1, x2, output_shape=None, name=None):
    """mtf.sub: x1 + (-x2); a number on the right is a scalar add of its negation"""
    if not isinstance(x2, Tensor):
        return _unary(lambda v: v + (-_scalar(x2)), x1, name="scalar_sub")
    if not isinstance(x1, Tensor):
        return add(negative(x2), x1)
    return add(x1, negative(x2), output_shape=output_shape)


def multiply(x1, x2, output_shape=None, name=None):
    if not isinstance(x2, Tensor):
        return _unary(lambda v: v * _scalar(x2), x1, name="scalar_mul")
    if not isinstance(x1, Tensor):
        return multiply(x2, x1)
    return _binary(lambda a, b: a * b, x1, x2, output_shape, name="mul")     # (upstream: an einsum without reduced dims)


def reciprocal(x, name=None):
    return _unary(lambda v: 1.0 / v, x, name="reciprocal")


def divide(x1, x2, output_shape=None, name=None):
    """mtf.divide: by a number -> scalar multiply by 1 / x2; by a tensor -> multiply(x1, reciprocal(x2))"""
    if not isinstance(x2, Tensor):
        return _unary(lambda v: v * (1.0 / _scalar(x2)), x1, name="scalar_div")
    if not isinstance(x1, Tensor):
        return multiply(reciprocal(x2), x1)
    return multiply(x1, reciprocal(x2), output_shape=output_shape)


def maximum(x1, x2, output_shape=None, name=None):
    return _binary(torch.maximum, x1, x2, output_shape, name="maximum")


def less(x1, x2, output_shape=None, name=None):
    return _binary(lambda a, b: a < b, x1, x2, output_shape, output_dtype=tf.bool, name="less")


def square(x, name=None):
    return _unary(lambda v: v * v, x, name="square")


def sqrt(x, name=None):
    return _unary(torch.sqrt, x, name="sqrt")


def rsqrt(x, name=None):
    return _unary(torch.rsqrt, x, name="rsqrt")


def exp(x, name=None):
    return _unary(torch.exp, x, name="exp")


def log(x, name=None):
    return _unary(torch.log, x, name="log")


def relu(x, name=None):
    return _unary(torch.relu, x, name=name or "relu")


def stop_gradient(x):
    return _unary(lambda v: v.detach(), x, name="stop_gradient")


def add_n(xs):
    if not xs:
        return 0
    r = xs[0]
    for x in xs[1:]:
        r = add(r, x)
    return r


def cast(x, dtype, name=None):
    if dtype is x.dtype:
        return x
    if dtype.is_floating:
        return _make(x.value.to(torch.float32), x.shape, dtype, like=x, name="cast")
    if dtype.is_integer:
        return _make(x.value.to(torch.int64), x.shape, dtype, like=x, name="cast")
    return _make(x.value != 0, x.shape, dtype, like=x, name="cast")


def to_float(x, name=None):
    return cast(x, tf.float32)


def constant(mesh, value, shape=None, dtype=tf.float32):
    shape = convert_to_shape(shape or [])
    v = torch.as_tensor(_scalar(value) if not isinstance(value, (list, tuple)) else value)
    v = v.to(torch.float32 if dtype.is_floating else torch.int64)
    return _make(v.expand(shape.to_integer_list).clone() if v.dim() == 0 else v.reshape(shape.to_integer_list), shape, dtype, mesh=mesh,
                 name="constant")


def import_tf_tensor(mesh, tf_tensor, shape=None, name=None):
    v = torch.as_tensor(tf_tensor)
    shape = convert_to_shape(shape)
    if v.is_floating_point():
        dtype = tf.float32
    elif v.dtype == torch.bool:
        dtype = tf.bool
    else:
        dtype = tf.int64 if v.dtype == torch.int64 else tf.int32
    return _make(v.reshape(shape.to_integer_list), shape, dtype, mesh=mesh, name=name or "import")


def import_fully_replicated(mesh, tf_tensor, shape, name=None):
    return import_tf_tensor(mesh, tf_tensor, shape, name=name)


def range(mesh, dim, dtype, name=None):     # noqa: A001  (the upstream name)
    dim = convert_to_dimension(dim)
    return _make(torch.arange(dim.size), Shape([dim]), dtype, mesh=mesh, name=name or "range")


def broadcast(x, new_shape):
    new_shape = convert_to_shape(new_shape)
    return _make(_align(x, new_shape.dims).expand(new_shape.to_integer_list), new_shape, x.dtype, like=x, name="broadcast")


def rename_dimension(x, old_name, new_name):
    return _make(x.value, x.shape.rename_dimension(old_name, new_name), x.dtype, like=x, name="rename")


def reshape(x, new_shape, name=None):
    new_shape = convert_to_shape(new_shape)
    assert new_shape.size == x.shape.size
    return _make(x.value.reshape(new_shape.to_integer_list), new_shape, x.dtype, like=x, name="reshape")


def transpose(x, new_shape, name=None):
    new_shape = convert_to_shape(new_shape)
    assert set(new_shape.dims) == set(x.shape.dims)
    return _make(_align(x, new_shape.dims), new_shape, x.dtype, like=x, name="transpose")


def replace_dimensions(tensor_or_shape, old_dim_or_dims, new_dim_or_dims):
    """mesh_tensorflow/ops.py replace_dimensions: a run of consecutive dimensions replaced by another run of the same total
    size (row-major reshape): [.., heads*kv, ..] <-> [.., heads, kv, ..] (heads-major)."""
    if isinstance(tensor_or_shape, Tensor):
        return reshape(tensor_or_shape, replace_dimensions(tensor_or_shape.shape, old_dim_or_dims, new_dim_or_dims))
    in_dims = tensor_or_shape.dims
    old = [old_dim_or_dims] if isinstance(old_dim_or_dims, Dimension) else list(old_dim_or_dims)
    new = [new_dim_or_dims] if isinstance(new_dim_or_dims, Dimension) else list(new_dim_or_dims)
    if math.prod(d.size for d in old) != math.prod(d.size for d in new):
        raise ValueError("replace_dimensions: sizes differ %s -> %s" % (old, new))
    pos = in_dims.index(old[0])
    if in_dims[pos:pos + len(old)] != old:
        raise ValueError("replace_dimensions: %s is not a consecutive run of %s" % (old, in_dims))
    return Shape(in_dims[:pos] + new + in_dims[pos + len(old):])


def _reduction_output_shape(x, output_shape, reduced_dim):
    if output_shape is None:
        return Shape([]) if reduced_dim is None else x.shape - reduced_dim
    if reduced_dim is not None and [reduced_dim] != [d for d in x.shape.dims if d not in output_shape.dims]:
        raise ValueError("reduced_dim contradicts output_shape")
    return output_shape


def _reduce(fn, x, output_shape, reduced_dim, name):
    output_shape = _reduction_output_shape(x, convert_to_shape(output_shape), convert_to_dimension(reduced_dim))
    axes = [i for i, d in enumerate(x.shape.dims) if d not in output_shape.dims]
    kept = [d for d in x.shape.dims if d in output_shape.dims]
    v = fn(x.value, axes) if axes else x.value
    r = _make(v, Shape(kept), x.dtype, like=x, name=name)
    return r if kept == output_shape.dims else transpose(r, output_shape)


def reduce_sum(x, disable_positional_args=None, output_shape=None, reduced_dim=None, name=None):
    assert disable_positional_args is None
    return _reduce(lambda v, ax: v.sum(dim=ax), x, output_shape, reduced_dim, "reduce_sum")


def reduce_max(x, disable_positional_args=None, output_shape=None, reduced_dim=None, name=None):
    assert disable_positional_args is None
    return _reduce(lambda v, ax: v.amax(dim=ax), x, output_shape, reduced_dim, "reduce_max")


def reduce_mean(x, disable_positional_args=None, output_shape=None, reduced_dim=None, name=None):
    """mtf.reduce_mean: reduce_sum(x) * (output_shape.size / x.shape.size)  (Appendix A.5)"""
    assert disable_positional_args is None
    output_shape = _reduction_output_shape(x, convert_to_shape(output_shape), convert_to_dimension(reduced_dim))
    if output_shape == x.shape:
        return x
    return reduce_sum(x, output_shape=output_shape) * (output_shape.size / x.shape.size)


def reduce_logsumexp(x, reduced_dim, extra_logit=None, name=None):
    """mesh_tensorflow/ops.py reduce_logsumexp: log(sum(exp(x - max))) + max with the max detached"""
    assert extra_logit is None
    reduced_dim = convert_to_dimension(reduced_dim)
    reduced_shape = x.shape - reduced_dim
    max_logit = reduce_max(stop_gradient(x), output_shape=reduced_shape)
    exp_x = exp(x - max_logit)
    sum_exp_x = reduce_sum(exp_x, output_shape=reduced_shape)
    return log(sum_exp_x) + max_logit


def log_softmax(x, reduced_dim, extra_logit=None, name=None):
    return x - reduce_logsumexp(x, reduced_dim, extra_logit=extra_logit)


def softmax(x, reduced_dim, extra_logit=None, name=None):
    """mtf.softmax = exp(log_softmax)  (Appendix A.3)"""
    return exp(log_softmax(x, reduced_dim, extra_logit=extra_logit))


def einsum(xs, output_shape=None, reduced_dims=None, name=None):
    """mtf.einsum over named dimensions: dimensions absent from the output are summed; default output = the input dimensions in
    order of first appearance minus the reduced ones (default reduced = the dimensions that occur in more than one input)."""
    output_shape = convert_to_shape(output_shape)
    count, input_dims = collections.OrderedDict(), []
    for x in xs:
        for d in x.shape.dims:
            if d not in count:
                input_dims.append(d)
            count[d] = count.get(d, 0) + 1
    if reduced_dims is not None:
        for d in reduced_dims:
            if not isinstance(d, Dimension):
                raise ValueError("reduced_dims must be Dimensions: %r" % (d,))
    if output_shape is None:
        if reduced_dims is None:
            reduced_dims = [d for d, c in count.items() if c > 1]
        output_shape = Shape([d for d in input_dims if d not in reduced_dims])
    elif reduced_dims is not None:
        for d in reduced_dims:
            if d not in count or d in output_shape.dims:
                raise ValueError("einsum: reduced dim %s must be an input dim and not an output dim" % (d,))
        for d in input_dims:
            if d not in reduced_dims and d not in output_shape.dims:
                raise ValueError("einsum: input dim %s is neither reduced nor in the output" % (d,))
    for d in output_shape.dims:
        if d not in count:
            raise ValueError("einsum: output dim %s is in no input" % (d,))
    letters = {d: string.ascii_letters[i] for i, d in enumerate(input_dims)}
    spec = ",".join("".join(letters[d] for d in x.shape.dims) for x in xs) + "->" + "".join(letters[d] for d in output_shape.dims)
    vals = [x.value if x.value.is_floating_point() else x.value.to(torch.float32) if any(y.value.is_floating_point() for y in xs)
            else x.value for x in xs]
    return _make(torch.einsum(spec, *vals), output_shape, xs[0].dtype, like=xs[0], name=name or "einsum")


def one_hot(indices, output_dim, on_value=1.0, off_value=0.0, dtype=tf.float32, name=None):
    output_dim = convert_to_dimension(output_dim)
    v = torch.nn.functional.one_hot(indices.value.to(torch.int64), output_dim.size)
    v = v.to(torch.float32) * (on_value - off_value) + off_value if dtype.is_floating else v
    return _make(v, indices.shape + [output_dim], dtype, like=indices, name="one_hot")


def gather(weights, indices, dim, output_shape=None):
    """mtf.gather: einsum([one_hot(indices, dim, dtype=weights.dtype), weights], reduced_dims=[dim])  (Appendix A.6)"""
    dim = convert_to_dimension(dim)
    if not isinstance(indices, Tensor):
        indices = constant(weights.mesh, indices, dtype=tf.int32)
    if weights.dtype is tf.bool:
        return cast(gather(to_float(weights), indices, dim, output_shape), tf.bool)
    return einsum([one_hot(indices, dim, dtype=weights.dtype), weights], reduced_dims=[dim], output_shape=output_shape)


def dropout(x, is_training=None, keep_prob=None, rate=None, noise_shape=None, name=None):
    if rate is None and keep_prob is not None:
        rate = 1.0 - keep_prob
    if not rate:
        return x
    raise NotImplementedError("refshim: dropout with rate > 0 draws from TF's random streams, which cannot be reproduced")


def recompute_grad(fn, explicit_inputs):
    """mtf.recompute_grad: same forward values, backward recomputed -- the same numbers as calling fn"""
    return fn(*explicit_inputs)


def mtf_slice(x, begin, size, slice_dim_name, name=None):
    i = x.shape.dimension_names.index(slice_dim_name)
    new = Shape([Dimension(d.name, size) if d.name == slice_dim_name else d for d in x.shape.dims])
    return _make(x.value.narrow(i, begin, size), new, x.dtype, like=x, name="slice")


# ---- variables --------------------------------------------------------------------------------------------------------------
class Variable(Operation):
    """mtf.Variable: master value (injected by the harness under the variable's full name) -> output = cast to the activation dtype"""

    def __init__(self, mesh, name, shape, dtype, initializer, trainable):
        scope = tf.get_variable_scope().name
        full_name = scope + "/" + name if scope else name
        super().__init__([], mesh, name=full_name)
        self.shape, self.trainable, self.initializer = shape, trainable, initializer
        self.master_dtype, self.slice_dtype, self.activation_dtype = dtype.master_dtype, dtype.slice_dtype, dtype.activation_dtype
        src = _injected.get(full_name)
        if src is None:
            if not (initializer is not None and initializer.kind == "constant"):
                raise KeyError("refshim: no value injected for variable %r %s (initializer %r)" % (full_name, shape, initializer))
            src = torch.full(shape.to_integer_list, float(initializer.value))
        master = torch.as_tensor(src, dtype=torch.float32).clone().reshape(shape.to_integer_list)
        self.master = _round_to(master, self.master_dtype).requires_grad_(trainable)
        out = Tensor(self, shape, self.activation_dtype, name=full_name)
        out._value = _round_to(self.master, self.activation_dtype)
        self._outputs = [out]
        mesh.graph.name_to_variable[full_name] = self

    value = property(lambda self: self._outputs[0])
    dtype = property(lambda self: self.activation_dtype)


_injected = {}


def inject_variables(values):
    """name -> array: the values mtf.get_variable hands out (constant-initialised variables may be omitted)"""
    _injected.clear()
    _injected.update(values)


def get_variable(mesh, name, shape, dtype=None, master_dtype=None, slice_dtype=None, activation_dtype=None, initializer=None,
                 trainable=True, **kwargs):
    if dtype is None:
        dtype = VariableDType(master_dtype or tf.float32, slice_dtype, activation_dtype)
    elif not isinstance(dtype, VariableDType):
        dtype = VariableDType(dtype, dtype, dtype)
    scope = tf.get_variable_scope().name
    full_name = scope + "/" + name if scope else name
    shape = convert_to_shape(shape)
    if full_name in mesh.graph.name_to_variable:
        var = mesh.graph.name_to_variable[full_name]
        if var.shape != shape:
            raise ValueError("Shape mismatch for variable %s: %s vs %s" % (full_name, var.shape, shape))
        return var.outputs[0]
    return Variable(mesh, name, shape, dtype, initializer, trainable).outputs[0]


class _Assign(Operation):
    def __init__(self, var, new_value):
        super().__init__([], var.mesh, name=var.name + "/assign")
        self.variable, self.new_value = var, new_value


def assign(var, new_val, assign_fn=None, name=None):
    """mtf.assign updates the per-device SLICES (slice dtype, fp32 under both dtype policies of the reference); the master copy
    (bf16 under "bf_16": true) is only refreshed from them when a checkpoint is written"""
    if isinstance(var, Tensor):
        var = var.operation
    return _Assign(var, _round_to(new_val.value.detach().to(torch.float32), var.slice_dtype))


def assign_sub(var, delta, name=None):
    if isinstance(var, Tensor):
        var = var.operation
    return _Assign(var, _round_to((var.master.detach() - delta.value.detach()).to(torch.float32), var.slice_dtype))


def gradients(ys, xs, grad_ys=None):
    """mtf.gradients: d sum(ys) / d xs for tensors xs of the graph (None where there is no path)"""
    assert grad_ys is None
    total = ys[0].value if len(ys) == 1 else sum(y.value.sum() for y in ys)
    gs = torch.autograd.grad(total.sum(), [x.value for x in xs], allow_unused=True, retain_graph=True)
    return [None if g is None else _make(g, x.shape, x.dtype, like=x, name="grad") for g, x in zip(gs, xs)]


# ---- mesh_tensorflow.layers ----------------------------------------------------------------------------------------------------
def _dense(x, new_dims, reduced_dims=None, expert_dims=None, use_bias=True, activation=None, master_dtype=tf.float32,
           slice_dtype=tf.float32, variable_dtype=None, kernel_initializer=None, kernel_weights=None, name=None):
    """mtf.layers.dense (Appendix A.4): kernel [reduced.., new..] and bias [new..] under scope `name`; einsum + bias"""
    if not isinstance(new_dims, list):
        new_dims = [new_dims]
    if variable_dtype is None:
        variable_dtype = VariableDType(master_dtype, slice_dtype, x.dtype)
    expert_dims = expert_dims or []
    if reduced_dims is None:
        reduced_dims = x.shape.dims[-1:]
    w_shape = Shape(expert_dims + reduced_dims + new_dims)
    output_shape = Shape([d for d in x.shape.dims if d not in reduced_dims] + new_dims)
    with tf.variable_scope(name, default_name="dense"):
        if kernel_weights is None:
            kernel_weights = get_variable(x.mesh, "kernel", w_shape, initializer=kernel_initializer, dtype=variable_dtype)
        y = einsum([x, kernel_weights], output_shape)
        if use_bias:
            b = get_variable(x.mesh, "bias", Shape(new_dims), initializer=tf.zeros_initializer(), dtype=variable_dtype)
            y += b
        if activation is not None:
            y = activation(y)
        return y


def _softmax_cross_entropy_with_logits(logits, targets, vocab_dim, z_loss=0.0):
    """mtf.layers.softmax_cross_entropy_with_logits (Appendix A.5): integer targets -> one-hot; -sum(onehot * log_softmax)"""
    if targets.dtype.is_integer:
        if set(targets.shape.dims) != set(logits.shape.dims).difference([vocab_dim]):
            raise ValueError("softmax_cross_entropy_with_logits with hard targets dims in targets=%s should be dims in logits=%s other "
                             "than vocab_dim=%s" % (targets, logits, vocab_dim))
        targets = one_hot(targets, vocab_dim, dtype=logits.dtype)
    elif set(targets.shape.dims) != set(logits.shape.dims):
        raise ValueError("softmax_cross_entropy_with_logits with soft targets: dims differ")
    if vocab_dim not in logits.shape.dims:
        raise ValueError("vocab_dim must be in logits.shape.dims")
    log_z = reduce_logsumexp(logits, vocab_dim)
    log_sm = logits - log_z
    loss = negative(reduce_sum(log_sm * targets, reduced_dim=vocab_dim))
    if z_loss != 0:
        loss += square(log_z) * z_loss
    return loss


layers = types.SimpleNamespace(dense=_dense, us_einsum=einsum, softmax_cross_entropy_with_logits=_softmax_cross_entropy_with_logits)


# ---- mesh_tensorflow.transformer.attention ------------------------------------------------------------------------------------
def _combined_dim(dims):
    return Dimension("_".join(d.name for d in dims), math.prod(d.size for d in dims))


class AttentionParams:
    """mesh_tensorflow/transformer/attention.py AttentionParams with combine_dims=True, shared_kv=False,
    fold_scaling_into_initializer=True (what attention_params_simple builds; Appendix A.1 / A.2): bias-free variables q, k, v of
    shape [input_dim, heads*kv] and o of shape [heads*kv, output_dim]; 1/sqrt(kv) lives in q's INITIALISER only."""

    def __init__(self, mesh, query_input_dim, memory_input_dim, output_dim, key_dim, value_dim, query_heads_dims, memory_heads_dims,
                 variable_dtype, shared_kv=False, no_query=False, combine_dims=True, ensemble_dim=None, keep_query_heads_dims=False,
                 fold_scaling_into_initializer=True):
        assert combine_dims and not shared_kv and not no_query and ensemble_dim is None and not keep_query_heads_dims
        self.query_input_dim, self.memory_input_dim, self.output_dim = query_input_dim, memory_input_dim, output_dim
        self.key_dim, self.value_dim = key_dim, value_dim
        self.fold_scaling_into_initializer = fold_scaling_into_initializer
        self.q_dims = list(query_heads_dims) + [key_dim]
        self.k_dims = list(memory_heads_dims) + [key_dim]
        self.v_dims = list(memory_heads_dims) + [value_dim]
        self.o_dims = list(query_heads_dims) + [value_dim]
        heads = math.prod(d.size for d in query_heads_dims)
        q_std = query_input_dim.size ** -0.5
        if fold_scaling_into_initializer:
            q_std *= key_dim.size ** -0.5
        kv_std = memory_input_dim.size ** -0.5
        o_std = (heads * value_dim.size) ** -0.5
        mk = lambda n, shape, std: get_variable(mesh, n, Shape(shape), initializer=tf.random_normal_initializer(stddev=std),  # noqa: E731
                                                 dtype=variable_dtype)
        self.wq = mk("q", [query_input_dim, _combined_dim(self.q_dims)], q_std)
        self.wk = mk("k", [memory_input_dim, _combined_dim(self.k_dims)], kv_std)
        self.wv = mk("v", [memory_input_dim, _combined_dim(self.v_dims)], kv_std)
        self.wo = mk("o", [_combined_dim(self.o_dims), output_dim], o_std)

    def _project(self, x, w, in_dim, dims):
        ret = einsum([x, w], reduced_dims=[in_dim])
        return replace_dimensions(ret, ret.shape.dims[-1], dims)

    def compute_q(self, query_antecedent):
        ret = self._project(query_antecedent, self.wq, self.query_input_dim, self.q_dims)
        if not self.fold_scaling_into_initializer:
            ret *= self.key_dim.size ** -0.5
        return ret

    def compute_k(self, memory_antecedent):
        return self._project(memory_antecedent, self.wk, self.memory_input_dim, self.k_dims)

    def compute_v(self, memory_antecedent):
        return self._project(memory_antecedent, self.wv, self.memory_input_dim, self.v_dims)

    def compute_output(self, o, output_shape=None):
        o = transpose(o, o.shape - self.o_dims + self.o_dims)
        o = replace_dimensions(o, self.o_dims, self.wo.shape.dims[0])
        return einsum([o, self.wo], output_shape=output_shape, reduced_dims=[self.wo.shape.dims[0]])


def _attention_params_simple(mesh, io_dim, kv_dim, heads_dim, variable_dtype):
    return AttentionParams(mesh, query_input_dim=io_dim, memory_input_dim=io_dim, output_dim=io_dim, key_dim=kv_dim, value_dim=kv_dim,
                           query_heads_dims=[heads_dim], memory_heads_dims=[heads_dim], variable_dtype=variable_dtype)


def _attention(q, k, v, memory_length_dim, key_dim, value_dim, bias=None, dropout_rate=0.0, dropout_broadcast_dims=None,
               extra_logit=None, context=None, float32_logits=True, z_loss_coeff=None):
    """mesh_tensorflow/transformer/attention.py attention (Appendix A.3): fp32 logits = einsum(q, k) over key_dim, UNSCALED;
    + bias; softmax over memory_length_dim; cast to v's dtype; einsum with v."""
    orig_q_shape = q.shape
    if float32_logits:
        k = cast(k, tf.float32)
        q = cast(q, tf.float32)
    logits = einsum([q, k], reduced_dims=[key_dim])
    if bias is not None:
        logits += cast(bias, logits.dtype)
    weights = softmax(logits, memory_length_dim, extra_logit=extra_logit)
    weights = cast(weights, v.dtype)
    weights = dropout(weights, rate=dropout_rate)
    outputs_shape = q.shape - key_dim + value_dim
    outputs = einsum([weights, v], outputs_shape)
    return reshape(outputs, orig_q_shape - [key_dim] + [value_dim])


_attention_module = types.SimpleNamespace(attention_params_simple=_attention_params_simple, AttentionParams=AttentionParams,
                                          attention=_attention)
transformer = types.SimpleNamespace(attention=_attention_module)


# ---- mesh_tensorflow.optimize -----------------------------------------------------------------------------------------------------
class Optimizer:
    def apply_grads(self, grads, variables):
        ops = []
        for grad, var in zip(grads, variables):
            ops.extend(self.apply_grad(grad, var))
        return ops


class AdamWeightDecayOptimizer(Optimizer):
    """mtf.optimize.AdamWeightDecayOptimizer as the reference documents it in its commented copy (src/optimizers.py:107-188,
    Appendix A.7): no bias correction; update = m' / (sqrt(v') + eps) [+ weight_decay * w]; w -= lr * update; state variables
    <var>/adam_m, <var>/adam_v, zero-initialised."""

    def __init__(self, learning_rate, weight_decay_rate=0.0, beta_1=0.9, beta_2=0.999, epsilon=1e-6, exclude_from_weight_decay=None,
                 variable_dtype=None):
        self.learning_rate, self.weight_decay_rate = learning_rate, weight_decay_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.exclude_from_weight_decay = exclude_from_weight_decay

    def apply_grad(self, grad, var):
        import re
        if grad is None:
            return []
        grad = to_float(grad)
        m = get_variable(var.mesh, var.name + "/adam_m", var.shape, initializer=tf.zeros_initializer(), trainable=False)
        v = get_variable(var.mesh, var.name + "/adam_v", var.shape, initializer=tf.zeros_initializer(), trainable=False)
        next_m = self.beta_1 * m + (1.0 - self.beta_1) * grad
        next_v = self.beta_2 * v + (1.0 - self.beta_2) * square(grad)
        update = next_m / (sqrt(next_v) + self.epsilon)
        use_decay = bool(self.weight_decay_rate) and not any(re.search(r, var.name) for r in (self.exclude_from_weight_decay or []))
        if use_decay:
            update += to_float(var.value) * self.weight_decay_rate
        update_with_lr = self.learning_rate * update
        return [assign_sub(var, update_with_lr), assign(m, next_m), assign(v, next_v)]


optimize = types.SimpleNamespace(Optimizer=Optimizer, AdamWeightDecayOptimizer=AdamWeightDecayOptimizer)
utils = types.SimpleNamespace(SCALAR_SUMMARIES_COLLECTION_KEY="mtf_scalar_summaries")


# ---- control-plane names of src/model_fns.py (mesh implementation, lowering, hooks): one device, nothing to lower ------------------
class _PlacementMeshImpl:
    def __init__(self, shape, layout, devices):
        self.shape, self.layout_rules, self.devices = convert_to_shape(shape), convert_to_layout_rules(layout), devices


placement_mesh_impl = types.SimpleNamespace(PlacementMeshImpl=_PlacementMeshImpl)


class Lowering:
    """mtf.Lowering(graph, {mesh: mesh_impl}): here the tensors already hold their values"""

    def __init__(self, graph, mesh_to_impl, autostack=True, log_file=None):
        self.graph, self.mesh_to_impl = graph, mesh_to_impl

    def export_to_tf_tensor(self, x):
        return x.value

    def lowered_operation(self, op):
        return op


class MtfRestoreHook:
    def __init__(self, lowering):
        self.lowering = lowering


class MtfCheckpointSaverListener:
    def __init__(self, lowering):
        self.lowering = lowering


utils.remove_summaries = lambda: None
utils.outside_all_rewrites = contextlib.nullcontext


def _serialize_num_microbatches(batch_dim, sequence_length, mesh_shape, layout_rules, tokens_per_microbatch_per_replica=None):
    """mesh_tensorflow/transformer/utils.py serialize_num_microbatches (Appendix A.7): no token budget -> 1; otherwise the batch
    per replica is cut so that a micro-batch holds at most that many tokens (rounded to a divisor of the per-replica batch)."""
    if not tokens_per_microbatch_per_replica:
        return 1
    mesh_shape, rules = convert_to_shape(mesh_shape), convert_to_layout_rules(layout_rules)
    replicas = 1
    for dim_name, axis in rules:
        if dim_name == batch_dim.name:
            replicas = mesh_shape.get_dim_by_name(axis).size
    batch_per_replica = batch_dim.size // replicas
    num = (batch_per_replica * sequence_length) // tokens_per_microbatch_per_replica
    num = max(1, min(batch_per_replica, num))
    while batch_per_replica % num:
        num -= 1
    return num


transformer.utils = types.SimpleNamespace(serialize_num_microbatches=_serialize_num_microbatches)
