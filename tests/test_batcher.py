"""Host-side behaviour of the prefetch thread (role of embodied.Prefetch, reference
core/prefetch.py:15-67): batches are the zipped generators, errors surface on the consumer
side, close() stops the thread."""

import itertools

import numpy as np
import pytest

from daydreamer_amd import agent as agent_mod


def _gen(length=3):
  def gen():
    for s in itertools.count():
      yield {'x': np.full((length, 2), s, np.float32), 'is_first': np.arange(length) == 0}
  return gen


def test_batches_are_the_zipped_generators():
  ds = agent_mod.Batcher(_gen(), 4, device=None)
  a, b = next(ds), next(ds)
  assert a['x'].shape == (4, 3, 2) and a['is_first'].dtype == bool
  assert (a['x'] == 0).all() and (b['x'] == 1).all()
  ds.close(join=True)
  assert not ds._thread.is_alive()
  with pytest.raises(StopIteration):
    next(ds)


def test_generator_errors_surface_on_the_consumer_side():
  def bad():
    yield {'x': np.zeros(2, np.float32)}
    raise ValueError('replay broke')
  ds = agent_mod.Batcher(lambda: bad(), 2, device=None)
  next(ds)
  with pytest.raises(ValueError, match='replay broke'):
    next(ds)
  ds.close(join=True)


def test_sharded_batches_are_marked():
  ds = agent_mod.Batcher(_gen(), 2, device=None, sharded=True)
  assert isinstance(next(ds), agent_mod.ShardedBatch)
  ds.close(join=True)
