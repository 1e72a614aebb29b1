"""Golden picks of the reference's FixedLength sampler (embodied/replay/fixed_length.py)
over a RAMStore with eviction, for tests/test_replay.py.  Needs /root/reference; writes
tests/golden/replay_picks.npz:
  lengths   episode lengths inserted in order
  capacity, chunk
  live      number of episodes alive after all insertions
  steps     stored steps after all insertions
  tags      [n_picks] the `tag` value (episode number * 1000 + step) at each sampled chunk's
            first row -> identifies (episode, start) independent of uuids
"""
import pathlib, sys, types
import numpy as np

sys.modules.setdefault('gym', types.ModuleType('gym'))
sys.path.insert(0, '/root/reference')
import embodied  # noqa: E402


def episodes(lengths, seed=0):
  rng = np.random.RandomState(seed)
  for e, n in enumerate(lengths):
    yield {
        'image': rng.randint(0, 255, (n, 8, 8, 3)).astype(np.uint8),
        'vector': rng.randn(n, 5),                       # float64 -> float32 by convert
        'action': rng.uniform(-1, 1, (n, 3)).astype(np.float32),
        'reward': rng.randn(n).astype(np.float32),
        'tag': (e * 1000 + np.arange(n)).astype(np.int32),  # int32 -> int64 by convert
        'is_first': np.arange(n) == 0,
        'is_last': np.arange(n) == n - 1,
        'is_terminal': np.zeros(n, bool),
        'log_extra': np.zeros(n, np.float32),            # dropped
    }


if __name__ == '__main__':
  lengths = np.random.RandomState(1).randint(4, 60, 40)
  capacity, chunk = 400, 12
  store = embodied.replay.RAMStore(capacity)
  replay = embodied.replay.FixedLength(store, chunk=chunk)
  for traj in episodes(lengths):
    replay.add_traj(traj)
  it = replay.dataset()
  tags = np.array([next(it)['tag'][0] for _ in range(300)])
  out = pathlib.Path(__file__).resolve().parent.parent / 'tests' / 'golden' / 'replay_picks.npz'
  np.savez(out, lengths=lengths, capacity=capacity, chunk=chunk, live=len(store),
           steps=len(replay), tags=tags)
  print('wrote', out, 'live', len(store), 'steps', len(replay), tags[:8])
