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
