"""tfshim -- the handful of `tensorflow.compat.v1` names the reference's model / optimizer files touch, over PyTorch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/refshim/__init__.py).  TensorFlow 2.4.0 (requirements.txt:1) is not installable here; this
module restates the published behaviour of exactly the calls the reference makes on its train path:

  tf.variable_scope / tf.get_variable_scope / tf.name_scope        variable names = "/".join(scopes) + "/" + name
  tf.random_normal_initializer / constant_initializer / zeros_initializer   recorded, not sampled (weights are injected)
  tf.float32 / bfloat16 / int32 / int64 / bool                      dtype objects with is_integer / is_floating
  tf.constant, tf.cast, tf.pad                                       eager, on torch tensors
  tf.train.get_or_create_global_step, cosine_decay, polynomial_decay
  tf.logging.info / warning                                          no-ops
  tf.layers.conv2d / conv2d_transpose, tf.get_variable, tf.matmul, tf.nn.relu / softmax, tf.random_uniform (injected),
  tf.argmax / one_hot / stop_gradient / reduce_mean / square / space_to_depth / depth_to_space    the discrete VAE's calls (A.8)
  tf.shape / maximum / expand_dims / squeeze / gather / range / reshape, tf.image.crop_and_resize / decode_jpeg (PIL)   src/input_fns.py:4-38

"tf tensors" are plain torch tensors (0-dim for the scalars of src/optimizers.py:19-76)."""
import contextlib
import math as _pymath
import types

import torch


class DType:
    def __init__(self, name, torch_dtype, is_integer=False, is_floating=False, is_bool=False):
        self.name, self.torch, self.is_integer, self.is_floating, self.is_bool = name, torch_dtype, is_integer, is_floating, is_bool

    def __repr__(self):
        return f"tf.{self.name}"


float32 = DType("float32", torch.float32, is_floating=True)
float64 = DType("float64", torch.float64, is_floating=True)
bfloat16 = DType("bfloat16", torch.bfloat16, is_floating=True)
float16 = DType("float16", torch.float16, is_floating=True)
int32 = DType("int32", torch.int32, is_integer=True)
int64 = DType("int64", torch.int64, is_integer=True)
bool = DType("bool", torch.bool, is_bool=True)     # noqa: A001  (the reference spells it tf.bool)

AUTO_REUSE = object()

# ---- variable scopes ---------------------------------------------------------------------------------------------------
_scopes = []


class _Scope:
    @property
    def name(self):
        return "/".join(_scopes)


def get_variable_scope():
    return _Scope()


_default_name_count = {}


@contextlib.contextmanager
def variable_scope(name_or_scope=None, default_name=None, reuse=None, **_kw):
    """tf.variable_scope(name): pushes `name` (which may itself contain '/'); name None -> default_name, made unique per
    enclosing scope the way TF does ("dense", "dense_1", ...)."""
    name = name_or_scope
    if name is None:
        key = ("/".join(_scopes), default_name)
        n = _default_name_count.get(key, 0)
        _default_name_count[key] = n + 1
        name = default_name if n == 0 else f"{default_name}_{n}"
    _scopes.append(str(name))
    try:
        yield _Scope()
    finally:
        _scopes.pop()


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield


def reset():
    del _scopes[:]
    _default_name_count.clear()
    _state["global_step"] = torch.zeros((), dtype=torch.int64)
    _variables.clear()
    _collections.clear()
    del restore_requests[:]
    adam_state.clear()


# ---- initialisers (recorded so that the harness can check them against the oracle's parameter table) -----------------------
class _Init:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return f"{self.kind}({ {k: v for k, v in self.__dict__.items() if k != 'kind'} })"


def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=None):
    return _Init("normal", mean=mean, stddev=stddev)


def constant_initializer(value=0, dtype=None):
    return _Init("constant", value=value)


def zeros_initializer(dtype=None):
    return _Init("constant", value=0)


def ones_initializer(dtype=None):
    return _Init("constant", value=1)


# ---- eager ops ---------------------------------------------------------------------------------------------------------
def constant(value, dtype=None, shape=None, name=None):
    t = torch.as_tensor(value, dtype=dtype.torch if dtype is not None else None)
    if shape is not None:
        t = t.expand(tuple(shape)).clone() if t.dim() == 0 else t.reshape(tuple(shape))
    return t


def pad(tensor, paddings, mode="CONSTANT", name=None, constant_values=0):
    """tf.pad: paddings[i] = [before, after] for axis i."""
    flat = []
    for before, after in reversed(list(paddings)):
        flat += [int(before), int(after)]
    return torch.nn.functional.pad(tensor, flat, mode="constant", value=constant_values)


_state = {"global_step": torch.zeros((), dtype=torch.int64)}


def set_global_step(step):
    _state["global_step"] = torch.as_tensor(int(step), dtype=torch.int64)


def _get_or_create_global_step():
    return _state["global_step"]


def _cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0, name=None):
    """tf.train.cosine_decay (Appendix A.8): lr * ((1 - alpha) * 0.5 * (1 + cos(pi * min(step, T) / T)) + alpha), in lr's dtype."""
    lr = torch.as_tensor(learning_rate)
    step = torch.minimum(torch.as_tensor(global_step).to(lr.dtype), torch.as_tensor(float(decay_steps), dtype=lr.dtype))
    completed = step / torch.as_tensor(float(decay_steps), dtype=lr.dtype)
    cosine_decayed = 0.5 * (1.0 + torch.cos(torch.as_tensor(_pymath.pi, dtype=lr.dtype) * completed))
    return lr * ((1 - alpha) * cosine_decayed + alpha)


def _polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False, name=None):
    """tf.train.polynomial_decay, cycle=False: (lr - end) * (1 - min(step, T) / T) ** power + end."""
    assert not cycle
    lr = torch.as_tensor(learning_rate)
    step = torch.minimum(torch.as_tensor(global_step).to(lr.dtype), torch.as_tensor(float(decay_steps), dtype=lr.dtype))
    p = step / torch.as_tensor(float(decay_steps), dtype=lr.dtype)
    return (lr - end_learning_rate) * torch.pow(1 - p, power) + end_learning_rate


train = types.SimpleNamespace(get_or_create_global_step=_get_or_create_global_step, cosine_decay=_cosine_decay,
                              polynomial_decay=_polynomial_decay)
logging = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None, warn=lambda *a, **k: None)


# ---- the plain-TF calls of the discrete VAE (src/vae_tf/models.py, layers.py; semantics: SURVEY.md Appendix A.8) --------------
# tf tensors are torch tensors in the reference's NHWC layout; dtypes below may be this module's DType objects or torch dtypes.
_variables = {}        # full name -> leaf torch tensor (requires_grad), in creation order
_injected = {}
_uniforms = []         # queue of arrays tf.random_uniform hands out (TF's own streams cannot be reproduced)


def inject_variables(values, uniforms=()):
    _variables.clear()
    _injected.clear()
    _injected.update(values)
    del _uniforms[:]
    _uniforms.extend(uniforms)


def created_variables():
    return dict(_variables)


def _torch_dtype(d):
    return d.torch if hasattr(d, "torch") else d


class _DTypeView:
    """what `tensor.dtype` must look like to the reference's code (`a.dtype.is_floating`, src/vae_tf/models.py:23)"""

    def __init__(self, td):
        self.torch = td

    is_floating = property(lambda self: self.torch.is_floating_point)
    is_integer = property(lambda self: not self.torch.is_floating_point and self.torch is not torch.bool)

    def __getattr__(self, name):          # everything torch's own Python code asks a dtype (is_floating_point, is_complex, ...)
        return getattr(self.torch, name)

    def __eq__(self, other):
        return _torch_dtype(other) == self.torch

    def __hash__(self):
        return hash(self.torch)

    def __repr__(self):
        return repr(self.torch)


class _TFTensor(torch.Tensor):
    """a torch tensor whose .dtype carries TensorFlow's dtype attributes (values of ops on it are _TFTensors again)"""

    @property
    def dtype(self):
        return _DTypeView(torch._C.TensorBase.dtype.__get__(self))


def cast(x, dtype, name=None):
    """tf.cast for this module's DType objects and for torch dtypes (`out.dtype` of a shim tensor)"""
    td = _torch_dtype(dtype)
    x = torch.as_tensor(x)
    if td is torch.bfloat16 and x.is_floating_point():
        return x.to(torch.bfloat16).to(torch.float32)     # bf16 values are held in float32 (see mtfshim)
    return x.to(td)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **_kw):
    """tf.get_variable: full name = variable scope + "/" + name; an existing name is returned again (reuse)"""
    scope = get_variable_scope().name
    full = scope + "/" + name if scope else name
    if full in _variables:
        for w in _watchers:
            w.append(_variables[full])
        return _variables[full]
    if full not in _injected:
        raise KeyError("refshim: no value injected for tf variable %r %s" % (full, shape))
    v = torch.as_tensor(_injected[full], dtype=torch.float32).clone()
    if shape is not None and tuple(v.shape) != tuple(int(s) for s in shape):
        raise ValueError("variable %s: injected shape %s, requested %s" % (full, tuple(v.shape), tuple(shape)))
    _variables[full] = v.requires_grad_(trainable)
    for w in _watchers:
        w.append(_variables[full])
    return _variables[full]


def _same_pad(n, k, s):
    """TF "SAME": out = ceil(n / s); total padding = max((out - 1) s + k - n, 0), the smaller half in front"""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _conv2d(inputs, filters, kernel_size, strides=(1, 1), padding="valid", name=None, use_bias=True, **_kw):
    """tf.layers.conv2d: NHWC, kernel [kh, kw, Cin, Cout] and bias [Cout] under scope `name` (default "conv2d", made unique)"""
    kh, kw = kernel_size
    sh, sw = strides
    with variable_scope(name, default_name="conv2d"):
        w = get_variable("kernel", [kh, kw, int(inputs.shape[-1]), filters])
        b = get_variable("bias", [filters]) if use_bias else None
    x = inputs.permute(0, 3, 1, 2)
    if padding.upper() == "SAME":
        (pt, pb), (pl, pr) = _same_pad(x.shape[2], kh, sh), _same_pad(x.shape[3], kw, sw)
        x = torch.nn.functional.pad(x, (pl, pr, pt, pb))
    y = torch.nn.functional.conv2d(x, w.permute(3, 2, 0, 1), b, stride=(sh, sw))
    return y.permute(0, 2, 3, 1)


def _conv2d_transpose(inputs, filters, kernel_size, strides=(1, 1), padding="valid", name=None, use_bias=True, **_kw):
    """tf.layers.conv2d_transpose, SAME: output = input * stride; kernel [kh, kw, Cout, Cin]; the gradient of the SAME strided
    convolution with respect to its input (which pads (k - s) / 2 on each side for the even sizes the reference uses)"""
    kh, kw = kernel_size
    sh, sw = strides
    assert padding.upper() == "SAME" and (kh - sh) % 2 == 0 and (kw - sw) % 2 == 0
    with variable_scope(name, default_name="conv2d_transpose"):
        w = get_variable("kernel", [kh, kw, filters, int(inputs.shape[-1])])
        b = get_variable("bias", [filters]) if use_bias else None
    y = torch.nn.functional.conv_transpose2d(inputs.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=(sh, sw),
                                             padding=((kh - sh) // 2, (kw - sw) // 2))
    return y.permute(0, 2, 3, 1)


def _unsupported(*_a, **_k):
    raise NotImplementedError("refshim: this tf.layers call is not on the reference's train path")


layers = types.SimpleNamespace(conv2d=_conv2d, conv2d_transpose=_conv2d_transpose, dense=_unsupported, batch_normalization=_unsupported)
nn = types.SimpleNamespace(relu=lambda x, name=None: torch.relu(x),
                           softmax=lambda x, axis=-1, name=None: torch.softmax(x, dim=axis))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a = a.transpose(-1, -2) if transpose_a else a
    b = b.transpose(-1, -2) if transpose_b else b
    return a @ b


def space_to_depth(x, block_size, name=None):
    """NHWC: output channel index = (dy * block + dx) * C + c"""
    B, H, W, C = x.shape
    r = block_size
    return x.reshape(B, H // r, r, W // r, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // r, W // r, r * r * C)


def depth_to_space(x, block_size, name=None):
    B, H, W, C = x.shape
    r = block_size
    return x.reshape(B, H, W, r, r, C // (r * r)).permute(0, 1, 3, 2, 4, 5).reshape(B, H * r, W * r, C // (r * r))


def random_uniform(shape, minval=0, maxval=None, dtype=None, seed=None, name=None):
    u = torch.as_tensor(_uniforms.pop(0), dtype=torch.float32)
    assert tuple(u.shape) == tuple(int(s) for s in shape), (tuple(u.shape), tuple(shape))
    return u


def log(x, name=None):
    return torch.log(x)


def argmax(x, axis=None, name=None, output_type=None):
    return torch.argmax(x, dim=axis)          # first maximum among ties, as tf.math.argmax


def one_hot(indices, depth, on_value=None, off_value=None, axis=None, dtype=None, name=None):
    v = torch.nn.functional.one_hot(indices, int(depth)).to(torch.float32)
    return v if axis in (None, -1, v.dim() - 1) else v.movedim(-1, axis)


def stop_gradient(x, name=None):
    return x.detach()


def reduce_mean(x, axis=None, name=None):
    return x.mean() if axis is None else x.mean(dim=axis)


def square(x, name=None):
    return x * x


_watchers = []      # lists that collect the variables tf.get_variable hands out (tf.custom_gradient's variable watcher)


class _CustomGradientBridge(torch.autograd.Function):
    """value = the function's (stop-gradient) result; backward = the user's grad function, called with the upstream gradient and
    the variables the forward touched, exactly as tf.custom_gradient does"""

    @staticmethod
    def forward(ctx, result, grad_fn, n_args, *inputs):
        ctx.grad_fn, ctx.n_args, ctx.n_vars = grad_fn, n_args, len(inputs) - n_args
        ctx.variables = inputs[n_args:]
        return result.clone()

    @staticmethod
    def backward(ctx, dresult):
        with torch.enable_grad():
            r = ctx.grad_fn(dresult, variables=list(ctx.variables)) if ctx.n_vars else ctx.grad_fn(dresult)
        arg_grads, var_grads = (r if ctx.n_vars else (r, []))
        return (None, None, None) + tuple(arg_grads) + tuple(var_grads)


def custom_gradient(f):
    """tf.custom_gradient: f(*args) -> (result, grad_fn); the result's gradient is whatever grad_fn(dresult[, variables]) returns
    for (args, variables touched by f).  The reference's VAE uses it for its recompute_grad (src/vae_tf/models.py:8-43)."""
    def wrapper(*args, **kwargs):
        # what grad_fn's closure will hold: leaves that require grad, typed so that `a.dtype.is_floating` works
        leaf_args = [a.detach().as_subclass(_TFTensor).requires_grad_(a.is_floating_point()) for a in args]
        watched = []
        _watchers.append(watched)
        try:
            result, grad_fn = f(*leaf_args, **kwargs)
        finally:
            _watchers.pop()
        variables = list(dict.fromkeys(watched))
        return _CustomGradientBridge.apply(result.detach(), grad_fn, len(args), *args, *variables)
    return wrapper


class GradientTape:
    """the subset the reference uses: watch(), gradient(target, sources, output_gradients)"""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, tensors):
        pass      # sources are leaves that already require grad (see custom_gradient)

    def gradient(self, target, sources, output_gradients=None):
        go = output_gradients[0] if isinstance(output_gradients, (list, tuple)) else output_gradients
        gs = torch.autograd.grad(target, list(sources), grad_outputs=go, allow_unused=True)
        return [torch.zeros_like(s) if g is None else g for g, s in zip(gs, sources)]


# ---- the tf calls of the reference's input helpers (src/input_fns.py:4-38) ----------------------------------------------------
def shape(x, name=None):
    return torch.as_tensor(list(x.shape), dtype=torch.int32)


def maximum(a, b, name=None):
    return torch.maximum(torch.as_tensor(a), torch.as_tensor(b))


def minimum(a, b, name=None):
    return torch.minimum(torch.as_tensor(a), torch.as_tensor(b))


def expand_dims(x, axis, name=None):
    return torch.as_tensor(x).unsqueeze(axis)


def squeeze(x, axis=None, name=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


def gather(params, indices, name=None, axis=0):
    return torch.index_select(params, axis, torch.as_tensor(indices).to(torch.int64).reshape(-1))


def range(start, limit=None, delta=1, dtype=None, name=None):     # noqa: A001
    return torch.arange(int(start)) if limit is None else torch.arange(int(start), int(limit), int(delta))


def reshape(x, shape, name=None):      # noqa: F811
    return x.reshape(tuple(int(s) for s in shape))


def _crop_and_resize(image, boxes, box_indices, crop_size, method="bilinear", extrapolation_value=0.0, name=None):
    """tf.image.crop_and_resize (documented behaviour): box [y1, x1, y2, x2] in normalised coordinates; output pixel (i, j) samples
    y = y1 (H-1) + i (y2-y1)(H-1)/(ch-1) (centre of the box when ch == 1), likewise x; bilinear between the four neighbours; a
    sample outside [0, H-1] x [0, W-1] yields extrapolation_value."""
    image = torch.as_tensor(image).to(torch.float32)
    N, H, W, C = image.shape
    ch, cw = int(crop_size[0]), int(crop_size[1])
    out = []
    for box, bi in zip(boxes, box_indices):
        y1, x1, y2, x2 = [torch.as_tensor(float(v), dtype=torch.float32) for v in box]
        img = image[int(bi)]
        f = lambda v: torch.as_tensor(float(v), dtype=torch.float32)       # noqa: E731
        ii, jj = torch.arange(ch, dtype=torch.float32), torch.arange(cw, dtype=torch.float32)
        ys = y1 * f(H - 1) + ii * ((y2 - y1) * f(H - 1) / f(ch - 1)) if ch > 1 else (0.5 * (y1 + y2) * f(H - 1)).reshape(1)
        xs = x1 * f(W - 1) + jj * ((x2 - x1) * f(W - 1) / f(cw - 1)) if cw > 1 else (0.5 * (x1 + x2) * f(W - 1)).reshape(1)
        oky, okx = (ys >= 0) & (ys <= H - 1), (xs >= 0) & (xs <= W - 1)
        cy, cx = ys.clamp(0, H - 1), xs.clamp(0, W - 1)
        top, bot, lef, rig = cy.floor().long(), cy.ceil().long(), cx.floor().long(), cx.ceil().long()
        ly, lx = (cy - top.float())[:, None, None], (cx - lef.float())[None, :, None]
        t = img[top][:, lef] + (img[top][:, rig] - img[top][:, lef]) * lx
        b = img[bot][:, lef] + (img[bot][:, rig] - img[bot][:, lef]) * lx
        v = t + (b - t) * ly
        mask = (oky[:, None, None] & okx[None, :, None])
        out.append(torch.where(mask, v, torch.as_tensor(float(extrapolation_value))))
    return torch.stack(out)


def _decode_jpeg(contents, channels=0, name=None):
    import io
    import numpy as _np
    from PIL import Image
    im = Image.open(io.BytesIO(contents))
    im = im.convert({1: "L", 3: "RGB"}[channels]) if channels else im
    a = _np.array(im)
    return torch.as_tensor(a if a.ndim == 3 else a[:, :, None])


image = types.SimpleNamespace(crop_and_resize=_crop_and_resize, decode_jpeg=_decode_jpeg)


# ---- control-plane names the reference's model_fns touch (src/model_fns.py:55-236, src/model_fns_tf.py:9-114) ------------------
# TPUEstimator, savers, hooks, summaries: recorded or ignored -- they carry no arithmetic.  The two pieces that do are restated:
# tf.train.AdamOptimizer (Appendix A.8) and tf.tpu.CrossShardOptimizer (mean over shards; one shard here).
estimator = types.SimpleNamespace(ModeKeys=types.SimpleNamespace(TRAIN="train", EVAL="eval", PREDICT="infer"))
GraphKeys = types.SimpleNamespace(GLOBAL_VARIABLES="variables", UPDATE_OPS="update_ops", SAVERS="savers")
_collections = {}
restore_requests = []     # (checkpoint path, names asked for) of every tf.train.init_from_checkpoint call


class _VarHandle:
    def __init__(self, name, t):
        self.name, self.tensor = name + ":0", t


def get_collection(key, scope=None):
    if key == GraphKeys.GLOBAL_VARIABLES:
        return [_VarHandle(n, t) for n, t in _variables.items() if scope is None or n.startswith(scope)]
    return list(_collections.get(key, []))


def add_to_collection(key, value):
    _collections.setdefault(key, []).append(value)


def global_variables():
    return get_collection(GraphKeys.GLOBAL_VARIABLES)


@contextlib.contextmanager
def control_dependencies(_ops):
    yield


def group(*ops, **_kw):
    flat = []
    for o in ops:
        flat.extend(o if isinstance(o, (list, tuple)) else [o])
    return flat


class _AssignAdd:
    def __init__(self, new_value):
        self.new_value = new_value


def assign_add(ref, value, **_kw):
    return _AssignAdd(torch.as_tensor(ref) + value)


def concat(values, axis, name=None):
    return torch.cat([torch.as_tensor(v) for v in values], dim=axis)


def to_int32(x, name=None):
    return torch.as_tensor(x).to(torch.int32)


def no_op(name=None):
    return None


def get_default_graph():
    # name scopes mirror the variable scopes here (every tf.variable_scope opens a name scope of the same name)
    return types.SimpleNamespace(get_collection=lambda key: list(_collections.get(key, [])), get_name_scope=lambda: "/".join(_scopes))


class AdamOptimizer:
    """tf.train.AdamOptimizer (Appendix A.8): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v zero-initialised slots;
    p -= lr_t m / (sqrt(v) + eps); eps 1e-8.  minimize() differentiates the loss w.r.t. every variable created so far."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **_kw):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon

    def minimize(self, loss, global_step=None, var_list=None, **_kw):
        names = list(_variables)
        grads = torch.autograd.grad(loss, [_variables[n] for n in names], allow_unused=True, retain_graph=True)
        t = adam_state.get("t", 0) + 1
        lr_t = float(self.lr) * _pymath.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        out = {"t": t, "grads": {}, "updated": {}, "m": {}, "v": {}}
        for n, g in zip(names, grads):
            p = _variables[n].detach()
            g = torch.zeros_like(p) if g is None else g
            m = self.b1 * adam_state.get("m", {}).get(n, torch.zeros_like(p)) + (1 - self.b1) * g
            v = self.b2 * adam_state.get("v", {}).get(n, torch.zeros_like(p)) + (1 - self.b2) * g * g
            out["grads"][n], out["m"][n], out["v"][n] = g, m, v
            out["updated"][n] = p - lr_t * m / (torch.sqrt(v) + self.eps)
        return out


adam_state = {}     # {"t": steps taken, "m": {...}, "v": {...}} carried in by the harness
train.get_global_step = _get_or_create_global_step
train.AdamOptimizer = AdamOptimizer
train.latest_checkpoint = lambda path: None
train.list_variables = lambda path: [(n, tuple(v.shape)) for n, v in _injected.items()]
train.init_from_checkpoint = lambda path, assignment_map: restore_requests.append((path, sorted(assignment_map)))
train.Saver = lambda *a, **k: types.SimpleNamespace(args=a, kwargs=k)
train.CheckpointSaverHook = lambda *a, **k: types.SimpleNamespace(args=a, kwargs=k)
tpu = types.SimpleNamespace(bfloat16_scope=contextlib.nullcontext, CrossShardOptimizer=lambda opt, **_k: opt)
math = types.SimpleNamespace(argmax=argmax, reduce_mean=reduce_mean)     # tf.math


class TPUEstimatorSpec:
    """tensorflow.python.tpu.tpu_estimator.TPUEstimatorSpec: the record a model_fn returns"""

    def __init__(self, mode=None, loss=None, train_op=None, host_call=None, eval_metrics=None, training_hooks=None,
                 evaluation_hooks=None, **kw):
        self.mode, self.loss, self.train_op, self.host_call, self.eval_metrics = mode, loss, train_op, host_call, eval_metrics
        self.training_hooks, self.evaluation_hooks = training_hooks, evaluation_hooks
