"""LazyMetrics / LazyScalar (daydreamer_amd/agent.py): what a pipelined Agent.train returns.

The reference's train() hands back that call's own metrics as numpy values (tfagent.py:67-70,
127-134) and its run loop only collects them - `metrics[key].append(value)` per train call,
`np.nanmean(values, dtype=np.float64)` at the log interval (run/train.py:77-85).  The pipelined
agent returns the same mapping without waiting for the device: these tests pin the host-side
contract (no fetch while the values are only collected, one fetch per call, numpy / float /
format / arithmetic views of a value, a fetch error surfaces where the value is looked at).
The device side (own-call metrics equal the sequential schedule's, bit for bit) is
tests/test_learner_gpu.py::test_pipelined_steps_equal_sequential.
"""

import collections

import numpy as np
import pytest

from daydreamer_amd.agent import LazyMetrics, LazyScalar


def _lazy(vals, log):
  def fetch():
    log.append(1)
    return dict(vals)
  return LazyMetrics(tuple(vals), fetch)


def test_collecting_values_does_not_fetch():
  log, metrics = [], collections.defaultdict(list)
  calls = [_lazy({'model_loss': np.float32(i + 0.5), 'actor_loss': float('nan') if i == 1 else float(i)}, log)
           for i in range(3)]
  for mets in calls:
    [metrics[key].append(value) for key, value in mets.items()]        # run/train.py:78
    assert len(mets) == 2 and 'model_loss' in mets and list(mets) == ['model_loss', 'actor_loss']
  assert not log and not any(m.resolved for m in calls)
  out = {name: np.nanmean(values, dtype=np.float64) for name, values in metrics.items()}   # :83
  assert log == [1, 1, 1] and all(m.resolved for m in calls)           # one fetch per call
  assert out['model_loss'] == pytest.approx(1.5) and out['actor_loss'] == pytest.approx(1.0)
  assert np.nanmean(metrics['model_loss'], dtype=np.float64) == pytest.approx(1.5) and len(log) == 3


def test_views_of_a_value():
  log = []
  m = _lazy({'a': np.float32(1.5), 'n': 7.0}, log)
  v = m['a']
  assert isinstance(v, LazyScalar) and not log
  assert float(v) == 1.5 and log == [1]
  assert isinstance(m['a'], np.float32)                                # after the fetch: the plain values
  assert np.asarray(v).dtype == np.float32 and np.asarray(v, np.float64).dtype == np.float64
  assert f'{v:.2f}' == '1.50' and v.item() == 1.5 and int(LazyScalar(m, 'n')) == 7
  assert v + 1 == 2.5 and 1 + v == 2.5 and v * 2 == 3.0 and v / 3 == 0.5 and -v == -1.5 and abs(v) == 1.5
  assert v < 2 and v >= 1.5 and np.isfinite(v) and np.array_equal(v, np.float32(1.5))
  assert dict(m) == {'a': np.float32(1.5), 'n': 7.0} and log == [1]
  with pytest.raises(KeyError):
    _lazy({'a': 1.0}, [])['missing']


def test_fetch_error_surfaces_at_the_look():
  def fetch():
    raise FloatingPointError('model_loss is not finite')
  m = LazyMetrics(('model_loss',), fetch)
  v = m['model_loss']                                                  # collecting is free ...
  with pytest.raises(FloatingPointError):
    float(v)                                                           # ... looking is not


def test_value_equality_and_remaining_arithmetic():
  """A LazyScalar compares by VALUE like the numpy scalar it stands for (the default identity
  comparison would make `mets[k] == 0.0` silently False) and is unhashable like one."""
  m = _lazy({'a': np.float32(1.5), 'z': 0.0}, [])
  a, z = m['a'], LazyScalar(m, 'z')
  assert isinstance(a, LazyScalar)
  assert (z == 0.0) and not (z != 0.0) and (a == 1.5) and (a != 2) and (a == LazyScalar(m, 'a'))
  assert a // 1 == 1.0 and 4 // a == 2.0 and a % 1 == 0.5 and 4 % a == 1.0 and a ** 2 == 2.25 and 2 ** z == 1.0
  assert +a == 1.5
  with pytest.raises(TypeError):
    hash(a)


class _FakeStream:
  def wait_stream(self, s): pass
  def wait_event(self, e): pass
  def synchronize(self): pass


def test_pipeline_bookkeeping_survives_a_failed_fetch(monkeypatch):
  """Pipeline.step (agent.py): the handle of the step just enqueued is in place BEFORE the previous
  step's metrics are resolved, so a FloatingPointError of step k - 1 (raised inside train call k)
  neither loses step k's metrics nor leaves a stale handle behind; flush() raises a stored error
  once and leaves the pipeline drained."""
  import daydreamer_amd.agent as A
  p = A.Pipeline.__new__(A.Pipeline)   # the host bookkeeping only: no device, no graphs
  fake = _FakeStream()
  class Plan:
    items = [('graph', 1)]
    def replay_on(self, *a, **k): pass
  class Ev:
    def record(self, s): pass
  p.device, p.s1, p.s2 = 'cpu', fake, fake
  p.pa1 = p.pa2 = p.pb = Plan()
  p.ev_in, p.ev_a, p.ev_b = Ev(), Ev(), [Ev(), Ev()]
  p.pub_a = p.pub_b = [None, None]
  p.k, p.pending, p.handle, p.keys, p.tuned = 0, None, None, ("model_loss",), True
  monkeypatch.setattr(A.torch.cuda, 'current_stream', lambda d=None: fake)
  p._publish = lambda pub, stream, clear=False: None
  bad = {1}
  def make_read(step):
    def fn(par):
      if step in bad:
        raise FloatingPointError(f'model_norm is not finite (step {step})')
      return {'model_loss': np.float32(step)}
    return fn
  handles = []
  for step in range(4):
    p._read = make_read(step)
    # (the lambda inside step() calls self._read at resolve time: bind this step's reader now)
    reader = p._read
    try:
      h = p.step()
      h._fetch = (lambda r=reader, par=p.pending: r(par))
      handles.append(h)
    except FloatingPointError:
      # raised by the resolve of step 1 inside call 2: call 2's handle must be the current one
      assert step == 2 and p.k == 3 and p.pending is not None
      p.handle._fetch = (lambda r=reader, par=p.pending: r(par))
      handles.append(p.handle)
  assert [float(h['model_loss']) for i, h in enumerate(handles) if i != 1] == [0.0, 2.0, 3.0]
  with pytest.raises(FloatingPointError):
    handles[1].resolve()
  # flush: resolves the newest handle, clears the bookkeeping
  assert float(p.flush()['model_loss']) == 3.0 and p.pending is None and p.handle is None
  # a failing last step: flush raises once and still drains
  p._read = make_read(1)
  h = p.step()
  h._fetch = (lambda: make_read(1)(0))
  with pytest.raises(FloatingPointError):
    p.flush()
  assert p.pending is None and p.handle is None and p.flush() is None


def test_two_threads_resolving_one_step_fetch_once():
  """A logger thread looks at a call's metrics while the training thread's next call resolves the
  same step (Pipeline.step -> prev.resolve()): exactly one of them runs the fetch, the other waits
  and gets the same values - or the same stored error, never a TypeError from a fetch that the
  first thread has already taken."""
  import threading, time
  for fail in (False, True):
    log, started = [], threading.Event()
    def fetch():
      log.append(1)
      started.set()
      time.sleep(0.05)       # (the device wait of the real fetch)
      if fail:
        raise FloatingPointError('model_loss is not finite')
      return {'a': np.float32(2.0)}
    m = LazyMetrics(('a',), fetch)
    got = []
    def look():
      try:
        got.append(float(m['a']))
      except Exception as e:
        got.append(e)
    th = threading.Thread(target=look)
    th.start()
    started.wait(2.0)
    try:
      got.append(dict(m.resolve())['a'])
    except Exception as e:
      got.append(e)
    th.join(2.0)
    assert log == [1] and len(got) == 2
    if fail:
      assert all(isinstance(g, FloatingPointError) for g in got) and m.failed
    else:
      assert got == [2.0, 2.0] and m.resolved
