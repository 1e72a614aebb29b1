"""TEST INFRASTRUCTURE (not product code): the subset of TensorFlow / TensorFlow-Probability /
Sonnet / ruamel.yaml that the reference's learner sources call, implemented on PyTorch-CPU, so
that the reference's OWN files - /root/reference/embodied/agents/dreamerv2plus/{agent,nets,
tfutils,tfagent,behaviors}.py, imported unmodified from where they lie - can be executed in this
container (TensorFlow cannot be installed here) and their outputs recorded as golden vectors
(tests/golden/make_reference_golden.py -> tests/golden/reference_*.npz).

What this pins and what it does not.  Everything the reference WROTE runs as written: module
wiring, the scan, stop-gradients, the straight-through sample, KL balancing, loss scales, the
lambda-return, normalisers, AutoAdapt, the hand-written Adam with clip / decay, the slow critic,
the order of updates inside Agent.train.  What is substituted is the LIBRARY underneath: each `tf.*` /
`tfd.*` primitive below is a few lines of torch written from the TensorFlow documentation (the same
documented semantics the oracle's header lists, each pinned by a loop-form numpy test in
tests/test_oracle_pins.py / tests/test_oracle_independent.py).  A primitive the sources do not
reach raises NotImplementedError instead of guessing.

Two deliberate deviations, both so that the comparison with the float64 oracle is tight:
  * `tf.float32` IS `torch.float64`: every "float32" tensor of the
    reference is computed in double precision; dtype identity checks (`x.dtype is tf.float32`)
    still hold;
  * randomness is injected: `tf.random.categorical` and `tfd.Normal.sample` draw from a queue of
    uniforms / standard normals the caller provides (`feed`), by the inverse-CDF rule of the oracle
    (oracle/dreamer_ref.sample_onehot) - the reference samples with seed=None, any exact sampler is
    in-distribution - and every draw is recorded (`FEED.draws`).

Only tests/golden/make_reference_golden.py imports this module.
"""

import contextlib
import functools
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

F64 = torch.float64


# ------------------------------------------------------------------------------------ tensors

class Shape(tuple):
  """tf.TensorShape as far as the sources use it: `shape[:-1] + [a, b]`, slices, == tuple."""

  def __add__(self, other):
    return Shape(tuple(self) + tuple(other))

  def __radd__(self, other):
    return Shape(tuple(other) + tuple(self))

  def __getitem__(self, i):
    r = tuple.__getitem__(self, i)
    return Shape(r) if isinstance(i, slice) else r

  def __eq__(self, other):
    try:
      return tuple(self) == tuple(other)
    except TypeError:
      return False

  def __ne__(self, other):
    return not self.__eq__(other)

  __hash__ = tuple.__hash__

  def as_list(self):
    return list(self)


def _convert(x):
  """Results of torch functions become plain tf.Tensors (never Variables)."""
  if isinstance(x, torch.Tensor):
    return x if type(x) is Tensor else x.as_subclass(Tensor)
  if isinstance(x, (tuple, list)) and not isinstance(x, (torch.Size, Shape)):
    return type(x)(_convert(v) for v in x)
  return x


class Tensor(torch.Tensor):
  """A torch tensor that answers to the tf.Tensor surface the sources touch.  The reference
  monkey-patches .mean / .sum / .reshape / .astype ... onto this class (tfutils.py:24-39); the
  functions of this module therefore never call such methods on their arguments - they go
  through `raw()` and the torch.* function forms."""

  __array_ufunc__ = None   # numpy scalars defer to __rmul__ & co. instead of calling .numpy()

  @classmethod
  def __torch_function__(cls, func, types, args=(), kwargs=None):
    with torch._C.DisableTorchFunctionSubclass():
      ret = func(*args, **(kwargs or {}))
    return _convert(ret)

  @property
  def shape(self):
    return Shape(torch.Tensor.size(self))

  def numpy(self):
    return torch.Tensor.numpy(torch.Tensor.detach(raw(self)))

  # tf tensors are immutable: `x += y` rebinds
  def __iadd__(self, other): return self + other
  def __isub__(self, other): return self - other
  def __imul__(self, other): return self * other
  def __itruediv__(self, other): return self / other

  def __repr__(self):
    return f'tf.Tensor(shape={tuple(self.shape)}, dtype={self.dtype})'


def raw(x):
  """Plain torch view of a value (autograd graph kept)."""
  if isinstance(x, torch.Tensor):
    return x.as_subclass(torch.Tensor)
  return x


def wrap(x):
  return x.as_subclass(Tensor) if isinstance(x, torch.Tensor) else x


def _dtype(dtype):
  if dtype is None:
    return None
  if isinstance(dtype, torch.dtype):
    return dtype
  dt = np.dtype(dtype)
  if dt.kind == 'f':
    return F64
  return {'b': torch.bool, 'u': torch.uint8, 'i': torch.int64}[dt.kind]


def tensor_of(value, dtype=None):
  """tf.convert_to_tensor: numpy floats of any width become the one float type."""
  dtype = _dtype(dtype)
  if isinstance(value, torch.Tensor):
    t = raw(value)
  elif isinstance(value, (list, tuple)) and any(isinstance(v, torch.Tensor) for v in value):
    t = torch.stack([raw(tensor_of(v)) for v in value])
  else:
    a = np.asarray(value)
    if a.dtype.kind == 'f':
      t = torch.tensor(a.astype(np.float64))
    elif a.dtype.kind == 'i' or (a.dtype.kind == 'u' and a.dtype != np.uint8):
      t = torch.tensor(a.astype(np.int64))
    elif a.dtype.kind in 'bu':
      t = torch.tensor(a)
    else:
      raise TypeError(f'cannot convert {type(value)} / {a.dtype} to a tensor')
  if dtype is not None and t.dtype != dtype:
    t = t.to(dtype)
  return wrap(t)


SCOPE = ['']          # absolute name-scope stack (sonnet name scopes)
VARIABLES = []        # every tf.Variable ever created, with its initial value


class Variable(Tensor):

  def __new__(cls, initial_value, trainable=None, dtype=None, name=None):
    init = raw(tensor_of(initial_value, dtype)).detach().clone()
    train = (True if trainable is None else bool(trainable)) and init.dtype.is_floating_point
    self = torch.Tensor._make_subclass(cls, init, train)
    self._trainable = True if trainable is None else bool(trainable)
    self._name = f'{SCOPE[-1]}/{name or "Variable"}:0'.lstrip('/')
    self._initial = init.clone()
    VARIABLES.append(self)
    return self

  def __init__(self, initial_value=None, trainable=None, dtype=None, name=None):
    pass

  @property
  def name(self):
    return self._name

  @property
  def trainable(self):
    return self._trainable

  def _set(self, value):
    v = raw(tensor_of(value)).detach().to(self.dtype)
    with torch.no_grad():
      torch.Tensor.copy_(raw(self), v.expand(torch.Tensor.size(self)))
    return self

  def assign(self, value):
    return self._set(value)

  def assign_add(self, value):
    return self._set(raw(self).detach() + raw(tensor_of(value)).detach())

  def assign_sub(self, value):
    return self._set(raw(self).detach() - raw(tensor_of(value)).detach())

  def reset(self):
    return self._set(self._initial)

  def __repr__(self):
    return f'tf.Variable({self._name}, shape={tuple(self.shape)})'

  def __hash__(self):
    return id(self)


class IndexedSlices:
  pass


# ---------------------------------------------------------------------------------- randomness

class Feed:
  """Queue of noise arrays, consumed by the sampling sites in call order; draws are logged."""

  def __init__(self):
    self.items, self.draws = [], []

  def load(self, items):
    assert not self.items, f'{len(self.items)} noise arrays of the previous call were not consumed'
    self.items = list(items)

  def pop(self, kind, shape):
    assert self.items, f'no noise left for a {kind} draw of shape {shape}'
    k, name, arr = self.items.pop(0)
    assert k == kind, (f'noise "{name}" is for a {k} draw, the sources ask for a {kind} draw '
                       f'of shape {shape}')
    arr = np.asarray(arr, np.float64)
    assert int(np.prod(arr.shape)) == int(np.prod(shape)), (name, arr.shape, tuple(shape))
    return name, torch.tensor(arr).reshape(tuple(shape))


FEED = Feed()


def random_categorical(logits, num_samples, seed=None):
  """tf.random.categorical(logits [rows, C], n) -> int64 [rows, n]; inverse CDF on softmax(logits)
  with the injected uniform: idx = #{c < C-1 : cdf_c <= u * cdf_{C-1}} (oracle sample_onehot)."""
  assert int(num_samples) == 1
  lg = raw(logits).detach()
  name, u = FEED.pop('uniform', lg.shape[:-1])
  probs = torch.softmax(lg, -1)
  cdf = torch.cumsum(probs, -1)
  idx = (cdf[..., :-1] <= (u * cdf[..., -1])[..., None]).sum(-1)
  FEED.draws.append((name, idx.numpy().copy()))
  return wrap(idx[:, None])


# ---------------------------------------------------------------------------------------- nest

def _is_leaf(x):
  return not isinstance(x, (dict, list, tuple)) or isinstance(x, (Shape, torch.Size))


def nest_flatten(s):
  if _is_leaf(s):
    return [s]
  if isinstance(s, dict):
    return [v for k in sorted(s) for v in nest_flatten(s[k])]
  return [v for x in s for v in nest_flatten(x)]


def nest_pack(structure, flat):
  flat = list(flat)
  def build(s):
    if _is_leaf(s):
      return flat.pop(0)
    if isinstance(s, dict):
      built = {k: build(s[k]) for k in sorted(s)}
      return {k: built[k] for k in s}
    return type(s)(build(x) for x in s)
  out = build(structure)
  assert not flat
  return out


def nest_map(fn, *structures):
  first = structures[0]
  if _is_leaf(first):
    return fn(*structures)
  if isinstance(first, dict):
    for s in structures[1:]:
      assert set(s) == set(first), (sorted(first), sorted(s))
    return {k: nest_map(fn, *(s[k] for s in structures)) for k in first}
  for s in structures[1:]:
    assert len(s) == len(first)
  return type(first)(nest_map(fn, *xs) for xs in zip(*structures))


def nest_assert_same(a, b):
  if _is_leaf(a) or _is_leaf(b):
    assert _is_leaf(a) and _is_leaf(b), (type(a), type(b))
    return
  assert type(a) is type(b) and len(a) == len(b), (type(a), type(b))
  if isinstance(a, dict):
    assert set(a) == set(b)
    for k in a:
      nest_assert_same(a[k], b[k])
  else:
    for x, y in zip(a, b):
      nest_assert_same(x, y)


# --------------------------------------------------------------------------------- functions

def _axes(axis, ndim):
  if axis is None:
    return None
  if isinstance(axis, (int, np.integer)):
    return [int(axis)]
  return [int(a) for a in axis]


def _reducer(fn):
  # (plain functions, not functools.partial: the sources bind them as METHODS of tf.Tensor)
  def reduce(x, axis=None, keepdims=False):
    x = raw(tensor_of(x))
    dims = _axes(axis, x.dim())
    if dims is None:      # all axes; of a scalar: over its one element (std of a scalar is 0)
      if x.dim() == 0:
        r = fn(x.reshape(1), [0], False)
        return wrap(r)
      dims = list(range(x.dim()))
    if not dims:          # tf: axis=[] reduces nothing (torch: dim=[] reduces everything)
      return wrap(x)
    return wrap(fn(x, dims, keepdims))
  return reduce


def _amax(x, dims, keep):
  return torch.amax(x, dims, keep)


def _amin(x, dims, keep):
  return torch.amin(x, dims, keep)


def _prod(x, dims, keep):
  for d in sorted((d % x.dim() for d in dims), reverse=True):
    x = torch.prod(x, d, keepdim=keep)
  return x


reduce_mean = _reducer(lambda x, d, k: torch.mean(x, d, k))
reduce_sum = _reducer(lambda x, d, k: torch.sum(x, d, k))
reduce_std = _reducer(lambda x, d, k: torch.sqrt(torch.var(x, d, unbiased=False, keepdim=k)))
reduce_variance = _reducer(lambda x, d, k: torch.var(x, d, unbiased=False, keepdim=k))
reduce_max = _reducer(_amax)
reduce_min = _reducer(_amin)
reduce_prod = _reducer(_prod)
reduce_logsumexp = _reducer(lambda x, d, k: torch.logsumexp(x, d, k))
reduce_any = _reducer(lambda x, d, k: _amax(x.to(torch.uint8), d, k).bool())
reduce_all = _reducer(lambda x, d, k: _amin(x.to(torch.uint8), d, k).bool())


def _ints(shape):
  return [int(s) for s in shape]


def reshape(x, shape):
  return wrap(torch.reshape(raw(x), _ints(shape)))


def cast(x, dtype):
  x = tensor_of(x)
  return wrap(raw(x).to(_dtype(dtype)))


def transpose(x, perm=None):
  x = raw(x)
  if perm is None:
    perm = list(reversed(range(x.dim())))
  return wrap(x.permute(_ints(perm)))


def _unary(fn):
  return lambda x, *a, **k: wrap(fn(raw(tensor_of(x))))


def zeros(shape, dtype=F64):
  return wrap(torch.zeros(_ints(shape) if not isinstance(shape, (int, np.integer)) else [int(shape)],
                          dtype=_dtype(dtype)))


def ones(shape, dtype=F64):
  return wrap(torch.ones(_ints(shape) if not isinstance(shape, (int, np.integer)) else [int(shape)],
                         dtype=_dtype(dtype)))


def concat(values, axis):
  return wrap(torch.cat([raw(tensor_of(v)) for v in values], int(axis)))


def stack(values, axis=0):
  return wrap(torch.stack([raw(tensor_of(v)) for v in values], int(axis)))


def split(value, num_or_size_splits, axis=0):
  v = raw(value)
  if isinstance(num_or_size_splits, (int, np.integer)):
    n = int(num_or_size_splits)
    assert v.shape[axis] % n == 0
    return [wrap(t) for t in torch.split(v, v.shape[axis] // n, int(axis))]
  return [wrap(t) for t in torch.split(v, _ints(num_or_size_splits), int(axis))]


def repeat(x, repeats, axis):
  return wrap(torch.repeat_interleave(raw(x), int(repeats), int(axis)))


def clip_by_value(x, lo, hi):
  x = raw(tensor_of(x))
  lo = raw(tensor_of(lo)).to(x.dtype) if isinstance(lo, torch.Tensor) else lo
  hi = raw(tensor_of(hi)).to(x.dtype) if isinstance(hi, torch.Tensor) else hi
  return wrap(torch.clamp(x, lo, hi))


def where(cond, a, b):
  a_, b_ = (raw(tensor_of(v)) if not isinstance(v, float) else v for v in (a, b))
  if isinstance(b_, float):
    b_ = torch.tensor(b_, dtype=a_.dtype)
  if isinstance(a_, float):
    a_ = torch.tensor(a_, dtype=b_.dtype)
  return wrap(torch.where(raw(cond), a_, b_))


def stop_gradient(x):
  return wrap(raw(x).detach()) if isinstance(x, torch.Tensor) else x


def einsum(eq, *xs):
  return wrap(torch.einsum(eq, *[raw(tensor_of(x)) for x in xs]))


def one_hot(indices, depth, dtype=F64):
  return wrap(F.one_hot(raw(indices).long(), int(depth)).to(_dtype(dtype)))


def cumprod(x, axis=0):
  return wrap(torch.cumprod(raw(x), int(axis)))


def global_norm(tensors):
  return wrap(torch.sqrt(sum((raw(t) ** 2).sum() for t in tensors)))


def clip_by_global_norm(tensors, clip_norm, use_norm=None):
  """tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)."""
  norm = raw(use_norm) if use_norm is not None else raw(global_norm(tensors))
  scale = clip_norm * torch.minimum(1.0 / norm, torch.tensor(1.0 / clip_norm, dtype=norm.dtype))
  return [wrap(raw(t) * scale) for t in tensors], wrap(norm)


def check_numerics(x, message):
  if not bool(torch.isfinite(raw(x)).all()):
    raise FloatingPointError(message)
  return x


def scan(fn, elems, initializer, reverse=False):
  n = nest_flatten(elems)[0].shape[0]
  order = range(n - 1, -1, -1) if reverse else range(n)
  last, outs = initializer, [[] for _ in nest_flatten(initializer)]
  for i in order:
    last = fn(last, nest_map(lambda x: x[i], elems))
    for o, l in zip(outs, nest_flatten(last)):
      o.append(l)
  if reverse:
    outs = [list(reversed(o)) for o in outs]
  return nest_pack(initializer, [stack(o, 0) for o in outs])


def conv2d(x, kernel, strides, padding):
  """tf.nn.conv2d, NHWC, filter [kh, kw, in, out], cross-correlation; VALID, or SAME at stride 1
  with an odd filter (floor(k / 2) zeros per side)."""
  x, w = raw(x), raw(kernel)
  s, k = int(strides), w.shape[0]
  if padding == 'VALID':
    pad = 0
  else:
    assert padding == 'SAME' and s == 1 and k % 2 == 1, (padding, s, k)
    pad = k // 2
  y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=s, padding=pad)
  return wrap(y.permute(0, 2, 3, 1))


def conv2d_transpose(x, kernel, output_shape, strides, padding):
  """tf.nn.conv2d_transpose, NHWC, filter [kh, kw, out, in]: the input-gradient of conv2d (no
  filter flip); VALID output = stride * (in - 1) + k."""
  assert padding == 'VALID', padding
  x, w = raw(x), raw(kernel)
  y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), stride=int(strides))
  y = y.permute(0, 2, 3, 1)
  assert tuple(y.shape) == tuple(_ints(output_shape)), (tuple(y.shape), tuple(output_shape))
  return wrap(y)


def avg_pool(x, ksize, strides, padding):
  x = raw(x)
  assert list(ksize) == [2, 2] and list(strides) == [2, 2] and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
  return wrap(F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1))


def moments(x, axes, keepdims=False):
  x = raw(x)
  dims = _axes(axes, x.dim())
  return (wrap(torch.mean(x, dims, keepdims)),
          wrap(torch.var(x, dims, unbiased=False, keepdim=keepdims)))


def batch_normalization(x, mean, variance, offset, scale, variance_epsilon):
  inv = torch.rsqrt(raw(variance) + variance_epsilon) * raw(scale)
  return wrap(raw(x) * inv + (raw(offset) - raw(mean) * inv))


class GradientTape:
  """Autograd is always recording in torch; the tape only marks where gradients are asked for."""

  LOG = []   # one {variable name: gradient} per .gradient() call

  def __init__(self, persistent=False):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    return False

  def gradient(self, target, sources):
    sources = list(sources)
    with torch._C.DisableTorchFunctionSubclass():
      # (the sources are the leaf Variables themselves: every use went through an alias of them)
      grads = torch.autograd.grad(raw(target), sources, allow_unused=True, retain_graph=True)
    grads = [None if g is None else raw(g).detach() for g in grads]
    GradientTape.LOG.append({s.name: (None if g is None else g.numpy().copy())
                             for s, g in zip(sources, grads)})
    return [None if g is None else wrap(g) for g in grads]


def function(fn=None, **kwargs):
  if fn is None:
    return lambda f: f
  return fn


def _not_needed(name):
  def fn(*a, **k):
    raise NotImplementedError(f'{name}: not reached by the learner step; not implemented')
  return fn


# --------------------------------------------------------------------------- distributions

class Distribution:
  pass


class Normal(Distribution):

  def __init__(self, loc, scale):
    self.loc, self.scale = raw(tensor_of(loc)), raw(tensor_of(scale))
    shape = torch.broadcast_shapes(self.loc.shape, self.scale.shape)
    self.batch_shape, self.event_shape = Shape(shape), Shape(())

  def sample(self, sample_shape=(), seed=None):
    assert tuple(sample_shape) == ()
    name, eps = FEED.pop('normal', self.batch_shape)
    FEED.draws.append((name, eps.numpy().copy()))
    return wrap(self.loc + self.scale * eps)      # reparameterised

  def log_prob(self, x):
    x = raw(tensor_of(x))
    return wrap(-0.5 * ((x - self.loc) / self.scale) ** 2 - torch.log(self.scale)
                - 0.5 * math.log(2 * math.pi))

  def entropy(self):
    return wrap((0.5 * math.log(2 * math.pi * math.e) + torch.log(self.scale))
                .expand(tuple(self.batch_shape)))

  def mean(self):
    return wrap(self.loc.expand(tuple(self.batch_shape)))

  mode = mean


class Bernoulli(Distribution):

  def __init__(self, logits=None, probs=None):
    assert probs is None
    self.logits = raw(tensor_of(logits))
    self.batch_shape, self.event_shape = Shape(self.logits.shape), Shape(())

  def log_prob(self, x):
    x = raw(tensor_of(x)).to(self.logits.dtype)
    return wrap(x * F.logsigmoid(self.logits) + (1 - x) * F.logsigmoid(-self.logits))

  def mean(self):
    return wrap(torch.sigmoid(self.logits))

  def mode(self):
    return wrap((self.logits > 0).to(self.logits.dtype))

  def entropy(self):
    p = torch.sigmoid(self.logits)
    return wrap(-(p * F.logsigmoid(self.logits) + (1 - p) * F.logsigmoid(-self.logits)))


class OneHotCategorical(Distribution):

  def __init__(self, logits=None, probs=None, dtype=F64):
    assert (logits is None) != (probs is None)
    self._logits = None if logits is None else raw(tensor_of(logits))
    self._probs = None if probs is None else raw(tensor_of(probs))
    self.dtype = _dtype(dtype)
    ref = self._logits if self._logits is not None else self._probs
    self.batch_shape, self.event_shape = Shape(ref.shape[:-1]), Shape(ref.shape[-1:])

  def logits_parameter(self):
    return wrap(self._logits if self._logits is not None else torch.log(self._probs))

  def probs_parameter(self):
    return wrap(torch.softmax(self._logits, -1) if self._logits is not None else self._probs)

  def _logp(self):
    return torch.log_softmax(raw(self.logits_parameter()), -1)

  def log_prob(self, x):
    return wrap((raw(tensor_of(x)).to(F64) * self._logp()).sum(-1))

  def entropy(self):
    lp = self._logp()
    return wrap(-(torch.exp(lp) * lp).sum(-1))

  def mode(self):
    lg = raw(self.logits_parameter())
    return wrap(F.one_hot(torch.argmax(lg, -1), lg.shape[-1]).to(self.dtype))

  def mean(self):
    return self.probs_parameter()

  def sample(self, sample_shape=(), seed=None):
    raise NotImplementedError('OneHotCategorical.sample: the sources sample through tfutils.OneHotDist')


class Independent(Distribution):

  def __init__(self, distribution, reinterpreted_batch_ndims):
    self.distribution, self.n = distribution, int(reinterpreted_batch_ndims)
    bs = distribution.batch_shape
    self.batch_shape = Shape(bs[:len(bs) - self.n])
    self.event_shape = Shape(tuple(bs[len(bs) - self.n:]) + tuple(distribution.event_shape))

  def _sum(self, x):
    x = raw(x)
    return wrap(x.sum(list(range(x.dim() - self.n, x.dim()))) if self.n else x)

  def log_prob(self, x):
    return self._sum(self.distribution.log_prob(x))

  def entropy(self):
    return self._sum(self.distribution.entropy())

  def sample(self, *a, **k):
    return self.distribution.sample(*a, **k)

  def mode(self):
    return self.distribution.mode()

  def mean(self):
    return self.distribution.mean()


def kl_divergence(a, b):
  if isinstance(a, Independent):
    assert isinstance(b, Independent) and a.n == b.n
    return a._sum(kl_divergence(a.distribution, b.distribution))
  assert isinstance(a, OneHotCategorical) and isinstance(b, OneHotCategorical), (type(a), type(b))
  la, lb = a._logp(), b._logp()
  return wrap((torch.exp(la) * (la - lb)).sum(-1))


# ----------------------------------------------------------------------------------- sonnet

@contextlib.contextmanager
def _scope(path):
  SCOPE.append(path)
  try:
    yield
  finally:
    SCOPE.pop()


def _scoped(fn):
  @functools.wraps(fn)
  def wrapper(self, *args, **kwargs):
    path = self.__dict__.get('_scope_path')
    if path is None:
      return fn(self, *args, **kwargs)
    with _scope(path):
      return fn(self, *args, **kwargs)
  return wrapper


class _ModuleMeta(type):
  """sonnet enters the module's own name scope around every method, so that variables created
  lazily inside __call__ are named <scope at construction>/<module name>/<variable>."""

  def __new__(mcls, name, bases, ns):
    for key, value in list(ns.items()):
      if isinstance(value, types.FunctionType) and (not key.startswith('__') or key in ('__call__', '__init__')):
        ns[key] = _scoped(value)
    return super().__new__(mcls, name, bases, ns)


class SntModule(metaclass=_ModuleMeta):

  def __init__(self, name=None):
    object.__setattr__(self, '_snt_name', name or type(self).__name__)
    object.__setattr__(self, '_scope_path', f'{SCOPE[-1]}/{self._snt_name}'.lstrip('/'))

  @property
  def name(self):
    return self._snt_name

  def _walk(self, seen):
    out = []
    def visit(v):
      if isinstance(v, Variable):
        if id(v) not in seen:
          seen.add(id(v))
          out.append(v)
      elif isinstance(v, SntModule):
        if id(v) not in seen:
          seen.add(id(v))
          if type(v).variables is not SntModule.variables:
            try:                     # (tfutils.Optimizer overrides .variables - and reads an
              own = v.variables      #  attribute it never sets; its slots are reached below)
            except AttributeError:
              own = v._walk(seen)
            for x in own:
              visit(x)
          else:
            out.extend(v._walk(seen))
      elif isinstance(v, dict):
        for k in sorted(v, key=str):
          visit(v[k])
      elif isinstance(v, (list, tuple)) and not isinstance(v, (Shape, torch.Size)):
        for x in v:
          visit(x)
    for key in sorted(vars(self)):
      visit(vars(self)[key])
    return out

  @property
  def variables(self):
    return tuple(self._walk({id(self)}))

  @property
  def trainable_variables(self):
    return tuple(v for v in self.variables if v.trainable)


# ------------------------------------------------------------------------------- installation

def _module(name, **attrs):
  m = types.ModuleType(name)
  for k, v in attrs.items():
    setattr(m, k, v)
  sys.modules[name] = m
  return m


def install():
  """Register tensorflow, tensorflow_probability, sonnet and ruamel.yaml stand-ins in sys.modules."""
  if 'tensorflow' in sys.modules:
    assert getattr(sys.modules['tensorflow'], '__tf_on_torch__', False), 'a real tensorflow is loaded'
    return sys.modules['tensorflow']
  tf = _module('tensorflow', __tf_on_torch__=True)
  tf.float32 = tf.float64 = F64
  tf.float16 = torch.float16   # (a distinct object: tfutils.Optimizer tests COMPUTE_DTYPE == tf.float16)
  tf.int32 = tf.int64 = torch.int64
  tf.uint8, tf.bool = torch.uint8, torch.bool
  tf.Tensor, tf.Variable, tf.IndexedSlices = Tensor, Variable, IndexedSlices
  tf.convert_to_tensor = tensor_of
  tf.zeros, tf.ones = zeros, ones
  tf.zeros_like = lambda x, dtype=None: wrap(torch.zeros_like(raw(tensor_of(x)), dtype=_dtype(dtype)))
  tf.ones_like = lambda x, dtype=None: wrap(torch.ones_like(raw(tensor_of(x)), dtype=_dtype(dtype)))
  tf.concat, tf.stack, tf.split, tf.repeat = concat, stack, split, repeat
  tf.reshape, tf.cast, tf.transpose = reshape, cast, transpose
  tf.clip_by_value, tf.where, tf.stop_gradient, tf.einsum, tf.one_hot = clip_by_value, where, stop_gradient, einsum, one_hot
  tf.identity = lambda x: x
  tf.tanh, tf.sign, tf.abs, tf.sqrt = _unary(torch.tanh), _unary(torch.sign), _unary(torch.abs), _unary(torch.sqrt)
  tf.maximum = lambda a, b: wrap(torch.maximum(raw(tensor_of(a)), raw(tensor_of(b)).to(raw(tensor_of(a)).dtype)))
  tf.range = lambda n: wrap(torch.arange(int(n)))
  tf.gather = _not_needed('tf.gather')
  tf.argsort = _not_needed('tf.argsort')
  tf.norm = _not_needed('tf.norm')
  tf.reduce_sum, tf.reduce_all = reduce_sum, reduce_all
  tf.clip_by_global_norm = clip_by_global_norm
  tf.scan, tf.function, tf.GradientTape = scan, function, GradientTape
  tf.get_current_name_scope = lambda: SCOPE[-1]
  tf.debug_nans = False
  tf.math = _module(
      'tensorflow.math', tanh=tf.tanh, log=_unary(torch.log), exp=_unary(torch.exp),
      sigmoid=_unary(torch.sigmoid), rsqrt=_unary(torch.rsqrt), abs=tf.abs,
      is_finite=_unary(torch.isfinite), cumprod=cumprod, top_k=_not_needed('tf.math.top_k'),
      reduce_mean=reduce_mean, reduce_sum=reduce_sum, reduce_std=reduce_std,
      reduce_variance=reduce_variance, reduce_max=reduce_max, reduce_min=reduce_min,
      reduce_prod=reduce_prod, reduce_logsumexp=reduce_logsumexp, reduce_any=reduce_any,
      reduce_all=reduce_all)
  tf.nn = _module(
      'tensorflow.nn', sigmoid=_unary(torch.sigmoid), softmax=lambda x, axis=-1: wrap(torch.softmax(raw(x), axis)),
      softplus=_unary(F.softplus), elu=_unary(F.elu), relu=_unary(torch.relu), tanh=tf.tanh,
      silu=_unary(F.silu), swish=_unary(F.silu),
      gelu=lambda x, approximate=False: wrap(F.gelu(raw(x), approximate='tanh' if approximate else 'none')),
      moments=moments, batch_normalization=batch_normalization, conv2d=conv2d,
      conv2d_transpose=conv2d_transpose, avg_pool=avg_pool)
  tf.nest = _module('tensorflow.nest', map_structure=nest_map, flatten=nest_flatten,
                    pack_sequence_as=nest_pack, assert_same_structure=nest_assert_same)
  tf.random = _module('tensorflow.random', categorical=random_categorical)
  tf.linalg = _module('tensorflow.linalg', global_norm=global_norm)
  tf.debugging = _module('tensorflow.debugging', check_numerics=check_numerics,
                         enable_check_numerics=lambda: None)
  tf.distribute = _module('tensorflow.distribute', has_strategy=lambda: False,
                          get_replica_context=_not_needed('tf.distribute.get_replica_context'),
                          MirroredStrategy=_not_needed('tf.distribute.MirroredStrategy'))
  exp = types.SimpleNamespace(
      enable_tensor_float_32_execution=lambda flag: None, list_physical_devices=lambda kind: [],
      set_memory_growth=lambda *a: None)
  tf.config = _module('tensorflow.config', run_functions_eagerly=lambda flag: None,
                      set_soft_device_placement=lambda flag: None, experimental=exp)
  tf.optimizers = types.SimpleNamespace(Adam=_not_needed('tf.optimizers.Adam'))
  tf.data = types.SimpleNamespace()

  class PerReplica:
    pass
  _module('tensorflow.python')
  _module('tensorflow.python.distribute')
  _module('tensorflow.python.distribute.values', PerReplica=PerReplica)
  sys.modules['tensorflow.python.distribute'].values = sys.modules['tensorflow.python.distribute.values']

  tfp = _module('tensorflow_probability')
  tfp.distributions = _module(
      'tensorflow_probability.distributions', Normal=Normal, Bernoulli=Bernoulli,
      OneHotCategorical=OneHotCategorical, Independent=Independent, kl_divergence=kl_divergence,
      MultivariateNormalDiag=_not_needed('tfd.MultivariateNormalDiag'),
      Uniform=_not_needed('tfd.Uniform'), TruncatedNormal=_not_needed('tfd.TruncatedNormal'),
      Deterministic=_not_needed('tfd.Deterministic'))

  snt = _module('sonnet', Module=SntModule)
  snt.v2 = _module('sonnet.v2', Module=SntModule)

  import re
  import yaml as pyyaml

  class Loader(pyyaml.SafeLoader):
    pass
  # ... and only true / false are booleans (`transform_rewards: off` is the string 'off')
  Loader.yaml_implicit_resolvers = {
      ch: [(tag, rx) for tag, rx in lst if tag != 'tag:yaml.org,2002:bool']
      for ch, lst in pyyaml.SafeLoader.yaml_implicit_resolvers.items()}
  Loader.add_implicit_resolver('tag:yaml.org,2002:bool',
                               re.compile(r'^(?:true|True|TRUE|false|False|FALSE)$'), list('tTfF'))
  # ruamel.yaml reads YAML 1.2, where `1e-4` is a float (PyYAML's YAML 1.1 wants `1.0e-4`)
  Loader.add_implicit_resolver(
      'tag:yaml.org,2002:float',
      re.compile(r'^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$|^[-+]?\.(inf|Inf|INF)$|^\.(nan|NaN|NAN)$'),
      list('-+0123456789.'))

  class YAML:
    def __init__(self, typ='safe'):
      pass
    def load(self, text):
      return pyyaml.load(text, Loader=Loader)
  ruamel = _module('ruamel')
  ruamel.yaml = _module('ruamel.yaml', YAML=YAML)
  sys.modules.setdefault('gym', types.ModuleType('gym'))
  return tf
