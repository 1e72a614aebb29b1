"""DeviceReplay (HBM-resident embodied.Replay) against the reference's FixedLength:
golden picks committed from the reference (tools/make_replay_golden.py), the
reference class itself when /root/reference is present, file-format interop,
and training straight from device minibatches.  CPU: the gather kernel's
restatement (oracle.ref_ops) stands in for dd_replay_gather."""

import pathlib
import sys
import types

import numpy as np
import pytest
import torch

from daydreamer_amd import replay as replay_mod
from oracle import ref_ops

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path('/root/reference')


def episodes(lengths, seed=0):
  rng = np.random.RandomState(seed)
  for e, n in enumerate(lengths):
    yield {
        'image': rng.randint(0, 255, (n, 8, 8, 3)).astype(np.uint8),
        'vector': rng.randn(n, 5),
        'action': rng.uniform(-1, 1, (n, 3)).astype(np.float32),
        'reward': rng.randn(n).astype(np.float32),
        'tag': (e * 1000 + np.arange(n)).astype(np.int32),
        'is_first': np.arange(n) == 0,
        'is_last': np.arange(n) == n - 1,
        'is_terminal': np.zeros(n, bool),
        'log_extra': np.zeros(n, np.float32),
    }


def make(capacity, chunk, **kw):
  return replay_mod.DeviceReplay(chunk=chunk, capacity=capacity, device='cpu',
                                 ops=ref_ops.RefOps('cpu'), **kw)


def test_golden_picks_from_reference():
  g = np.load(ROOT / 'tests' / 'golden' / 'replay_picks.npz')
  rep = make(int(g['capacity']), int(g['chunk']))
  for traj in episodes(g['lengths']):
    rep.add_traj(traj)
  assert rep.stats == {'replay_steps': int(g['steps']), 'replay_trajs': int(g['live'])}
  assert len(rep) == int(g['steps'])
  it = rep.dataset()
  tags = np.array([next(it)['tag'][0] for _ in range(len(g['tags']))])
  np.testing.assert_array_equal(tags, g['tags'])


def test_wire_format_and_batches_match_dataset():
  rep = make(400, 12)
  for traj in episodes([30, 20, 45, 13]):
    rep.add_traj(traj)
  chunk = next(rep.dataset())
  assert 'log_extra' not in chunk
  assert chunk['vector'].dtype == np.float32 and chunk['tag'].dtype == np.int64
  assert chunk['image'].dtype == np.uint8 and chunk['is_terminal'].dtype == bool
  assert chunk['is_first'][0] and not chunk['is_first'][1:].any()
  # device minibatch == stack of the chunks the host generator yields from the same RNG state
  a, b = make(400, 12), make(400, 12)
  for traj in episodes([30, 20, 45, 13]):
    a.add_traj(traj); b.add_traj(traj)
  it = a.dataset()
  host = [next(it) for _ in range(5)]
  dev = b.sample_batch(5)
  for k in host[0]:
    np.testing.assert_array_equal(np.stack([h[k] for h in host]), dev[k].numpy(), err_msg=k)


def test_add_transitions_and_short_episodes(capsys):
  rep = make(1000, 6)
  n = 0
  for traj in episodes([10, 3, 8]):
    for t in range(len(traj['reward'])):
      rep.add({k: v[t] for k, v in traj.items()}, worker=n % 2)
    n += 1
  assert rep.stats == {'replay_steps': 18, 'replay_trajs': 2}
  assert 'Skipping short trajectory of length 3.' in capsys.readouterr().out


def test_ring_wrap_and_eviction_keep_contents():
  rep = make(100, 5, ring_steps=120)
  trajs = list(episodes([40, 35, 30, 38, 25, 33]))
  keys = [rep.add_traj(t) for t in trajs]
  assert rep.steps <= 100 + 33
  for key, (off, n) in rep.table.items():
    src = trajs[keys.index(key)]
    np.testing.assert_array_equal(rep.rings['tag'][off:off + n].numpy(), src['tag'].astype(np.int64))
    np.testing.assert_array_equal(rep.rings['image'][off:off + n].numpy(), src['image'])
  spans = sorted(rep.table.values())
  for (o1, n1), (o2, _) in zip(spans, spans[1:]):
    assert o1 + n1 <= o2


def test_npz_round_trip(tmp_path):
  rep = make(400, 12, directory=tmp_path)
  for traj in episodes([30, 20, 45]):
    rep.add_traj(traj)
  assert rep.save() == str(tmp_path)
  files = sorted(tmp_path.glob('*.npz'))
  assert len(files) == 3 and all(f.stem.count('-') == 3 for f in files)
  rep2 = make(400, 12, directory=tmp_path)
  rep2.load()
  assert rep2.stats == rep.stats
  assert sorted(n for _, n in rep2.table.values()) == [20, 30, 45]
  for key, (off, n) in rep2.table.items():
    o1, _ = rep.table[key]
    for k in rep.rings:
      np.testing.assert_array_equal(rep.rings[k][o1:o1 + n].numpy(), rep2.rings[k][off:off + n].numpy())


@pytest.mark.skipif(not REF.exists(), reason='reference checkout not present')
def test_against_reference_classes(tmp_path):
  sys.modules.setdefault('gym', types.ModuleType('gym'))
  sys.path.insert(0, str(REF))
  import embodied
  lengths = np.random.RandomState(5).randint(3, 50, 60)
  ref = embodied.replay.FixedLength(embodied.replay.RAMStore(300), chunk=9, prio_starts=0.5, prio_ends=1.0)
  rep = make(300, 9, prio_starts=0.5, prio_ends=1.0)
  for traj in episodes(lengths):
    ref.add_traj(traj); rep.add_traj(traj)
    assert len(ref) == len(rep)
  assert ref.stats == rep.stats
  a, b = ref.dataset(), rep.dataset()
  for _ in range(200):
    x, y = next(a), next(b)
    assert x.keys() == y.keys()
    for k in x:
      assert x[k].dtype == y[k].dtype, k
      np.testing.assert_array_equal(x[k], y[k], err_msg=k)
  # episodes written by us are read by the reference's DiskStore and vice versa
  rep.save(tmp_path)
  disk = embodied.replay.DiskStore(tmp_path)
  assert disk.stats() == {'steps': len(rep), 'trajs': len(rep.table)}
  for key in disk.keys():
    off, n = rep.table[key]
    got = disk[key]
    for k in rep.rings:
      np.testing.assert_array_equal(got[k], rep.rings[k][off:off + n].numpy())


def test_agent_trains_from_device_minibatches():
  """Agent.dataset(replay.dataset) -> device minibatches; the step equals the one on
  the same minibatch handed over as host numpy."""
  from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic
  cfgs = agent_mod.Agent.configs
  cfg = config_mod.Config(cfgs['defaults']).update(cfgs['a1_vision']).update(cfgs['debug'])
  cfg = cfg.update({'batch_size': 3, 'replay_chunk': 5, 'imag_horizon': 3})
  obs, act = synthetic.make_spaces(64, 5, 3)
  def agent():
    return agent_mod.Agent(obs, act, None, cfg, _ops=ref_ops.RefOps('cpu'), _device='cpu', _dtype=torch.float64)
  rep = make(500, 5)
  for e in range(4):
    ep = synthetic.make_batch(obs, act, 1, 20, seed=e, terminals=0.0)
    rep.add_traj({k: v[0] for k, v in ep.items()})
  ag1, ag2 = agent(), agent()
  ds = ag1.dataset(rep.dataset)
  batch = next(ds)
  assert all(isinstance(v, torch.Tensor) for v in batch.values())
  host = {k: v.numpy().copy() for k, v in batch.items()}
  _, _, m1 = ag1.train(batch)
  _, _, m2 = ag2.train(host)
  assert m1.keys() == m2.keys()
  for k in m1:
    assert np.array_equal(m1[k], m2[k], equal_nan=True), k


def test_edge_cases(tmp_path):
  """Empty replay, unbounded capacity, oversize episodes, key mismatch, capacity cut-off on
  load (DiskStore.sync keeps the newest episodes), prioritize() is a no-op."""
  rep = make(None, 4, ring_steps=64)
  assert len(rep) == 0 and rep.stats == {'replay_steps': 0, 'replay_trajs': 0}
  with pytest.raises(RuntimeError):
    rep.sample_batch(2)
  with pytest.raises(ValueError):
    rep.add_traj(next(episodes([80])))
  a, b = list(episodes([10, 12]))
  rep.add_traj(a)
  with pytest.raises(KeyError):
    rep.add_traj({k: v for k, v in b.items() if k != 'vector'})
  rep.add_traj(b)
  assert rep.prioritize(['x'], [1.0]) is None
  assert rep.stats == {'replay_steps': 22, 'replay_trajs': 2}   # no capacity: nothing evicted
  out = rep.sample_batch(3)
  assert out['image'].shape == (3, 4, 8, 8, 3) and out['is_first'][:, 0].all()
  # newest-first selection up to the capacity when loading a directory
  src = make(1000, 4, directory=tmp_path)
  import time
  for traj in episodes([10, 11, 12, 13]):
    src.add_traj(traj)
    src.save()
    time.sleep(1.05)   # file names carry a one-second time stamp
  dst = make(30, 4, directory=tmp_path)
  dst.load()
  assert sorted(n for _, n in dst.table.values()) == [12, 13]
  assert len(dst) == 25


# ---- prioritized replay -------------------------------------------------------------------

def _prio_trace(replay, lengths, n_picks):
  sys.path.insert(0, str(ROOT / 'tools'))
  import importlib
  sys.modules.setdefault('gym', types.ModuleType('gym'))
  spec = importlib.util.spec_from_file_location('mpg', ROOT / 'tools' / 'make_prio_golden.py')
  src = (ROOT / 'tools' / 'make_prio_golden.py').read_text()
  # only the `trace` driver is needed (the module imports the reference at top level)
  ns = {'np': np, 'itertools': __import__('itertools'), 'uuid': __import__('uuid'),
        'episodes': episodes}
  start = src.index('def trace(')
  end = src.index("if __name__ == '__main__':")
  exec(src[start:end], ns)
  return ns['trace'](replay, lengths, n_picks, replay.chunk)


@pytest.mark.parametrize('name,kw', [
    ('power', dict(fraction=0.5, exponent=0.5)),
    ('softmax', dict(fraction=0.3, softmax=True, temp=2.0, constant=0.1))])
def test_prioritized_matches_reference_golden(name, kw):
  """DevicePrioritized against a committed trace of the reference's Prioritized replay
  (tools/make_prio_golden.py): the same chunks, the same priority-draw keys and the same
  draw probabilities, with priorities fed back through prioritize() every four draws."""
  g = np.load(ROOT / 'tests' / 'golden' / 'replay_prio.npz')
  rep = replay_mod.DevicePrioritized(chunk=int(g['chunk']), capacity=100000, device='cpu',
                                     ops=ref_ops.RefOps('cpu'), **kw)
  tags, keys, probs = _prio_trace(rep, g['lengths'], len(g[f'{name}_tags']))
  np.testing.assert_array_equal(tags, g[f'{name}_tags'])
  np.testing.assert_array_equal(keys, g[f'{name}_keys'])
  np.testing.assert_allclose(probs, g[f'{name}_probs'], rtol=1e-12)
  st = rep.stats
  assert st['replay_trajs'] == len(g['lengths']) and 0 < st['randomness'] <= 1
  assert st['update_max'] > 0


def test_prioritized_batches_and_eviction():
  rep = replay_mod.DevicePrioritized(chunk=8, capacity=120, device='cpu', ops=ref_ops.RefOps('cpu'))
  for traj in episodes([30, 40, 50, 35]):
    rep.add_traj(traj)
  assert len(rep.prios) == len(rep.table) < 4            # evicted episodes leave the table too
  batch = rep.sample_batch(5)
  assert batch['key'].shape == (5, 8, 3) and batch['key'].dtype == torch.int64
  assert batch['prob'].shape == (5, 8) and batch['prob'].dtype == torch.float64
  assert bool((batch['key'][:, 0] == batch['key'][:, -1]).all())
  rep.prioritize(batch['key'].numpy(), np.abs(np.random.RandomState(0).randn(5, 8)))
  key, index = rep.decode(batch['key'][0, 0].numpy())
  assert key in rep.table and rep.prios.steps[key][index] > 0
  # priorities for an episode that has been evicted meanwhile are ignored, not an error
  for traj in episodes([60, 60], seed=5):
    rep.add_traj(traj)
  rep.prioritize(batch['key'].numpy(), np.ones((5, 8)))
  chunk = next(rep.dataset())
  assert chunk['key'].shape == (8, 3) and chunk['prob'].shape == (8,) and chunk['is_first'][0]


def test_prioritized_episode_of_exactly_one_chunk():
  """length == chunk: a single start per episode.  The reference's start / end up-weighting
  (prios.py:88-92) turns its probability into 0 / 0; here it stays a valid distribution."""
  rep = replay_mod.DevicePrioritized(chunk=8, capacity=200, device='cpu', ops=ref_ops.RefOps('cpu'),
                                     prio_starts=0.0, prio_ends=1.0)
  for traj in episodes([8, 8, 20]):
    rep.add_traj(traj)
  for key, p in rep.prios.start_probs.items():
    assert np.isfinite(p).all() and abs(p.sum() - 1) < 1e-12, (key, p)
  batch = rep.sample_batch(6)
  assert batch['key'].shape == (6, 8, 3)


def test_reseed_gives_ranks_their_own_draws():
  """Rank-sharded dataset (Agent.dataset with world > 1): replays with identical contents and
  the default seed must not hand every rank the same rows."""
  picks = []
  for rank in range(2):
    rep = make(500, 8)
    for traj in episodes([40, 50, 60]):
      rep.add_traj(traj)
    rep.reseed(rank)
    picks.append(rep.sample_batch(16)['tag'][:, 0].numpy().copy())
  assert not np.array_equal(picks[0], picks[1])
  rep0 = make(500, 8)
  for traj in episodes([40, 50, 60]):
    rep0.add_traj(traj)
  rep0.reseed(0)   # rank 0 keeps the seed-0 stream of the single-process learner
  assert np.array_equal(rep0.sample_batch(16)['tag'][:, 0].numpy(), picks[0])
