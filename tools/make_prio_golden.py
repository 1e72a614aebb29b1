"""Golden trace of the reference's Prioritized replay (embodied/replay/prioritized.py +
prios.py) for tests/test_replay.py::test_prioritized_*: the sequence of sampled chunks
(tag of the first row), their priority-draw keys and probabilities, with priorities fed back
every few draws.  Needs /root/reference; writes tests/golden/replay_prio.npz."""
import itertools, pathlib, sys, types, uuid
import numpy as np

sys.modules.setdefault('gym', types.ModuleType('gym'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import embodied  # noqa: E402
from make_replay_golden import episodes  # noqa: E402


def trace(replay, lengths, n_picks, chunk):
  """Drive a Prioritized-like replay; deterministic episode ids via a patched uuid4."""
  counter = itertools.count(1)
  real = uuid.uuid4
  uuid.uuid4 = lambda: uuid.UUID(int=next(counter) * 0x1000193)
  try:
    for traj in episodes(lengths):
      replay.add_traj(traj)
  finally:
    uuid.uuid4 = real
  it = replay.dataset()
  tags, keys, probs = [], [], []
  pending = []
  for i in range(n_picks):
    ch = next(it)
    tags.append(int(ch['tag'][0])); keys.append(ch['key'][0].copy()); probs.append(float(ch['prob'][0]))
    pending.append((ch['key'], np.abs(ch['reward']).astype(np.float64) + 0.01 * i))
    if len(pending) == 4:   # a train step's worth of priorities goes back
      replay.prioritize(np.stack([k for k, _ in pending]), np.stack([p for _, p in pending]))
      pending = []
  return np.array(tags), np.stack(keys), np.array(probs)


if __name__ == '__main__':
  lengths = np.random.RandomState(2).randint(14, 60, 12)
  chunk = 12
  out = {}
  for name, kw in dict(power=dict(fraction=0.5, exponent=0.5),
                       softmax=dict(fraction=0.3, softmax=True, temp=2.0, constant=0.1)).items():
    store = embodied.replay.RAMStore(100000)
    replay = embodied.replay.Prioritized(store, chunk=chunk, **kw)
    tags, keys, probs = trace(replay, lengths, 120, chunk)
    out.update({f'{name}_tags': tags, f'{name}_keys': keys, f'{name}_probs': probs})
  path = pathlib.Path(__file__).resolve().parent.parent / 'tests' / 'golden' / 'replay_prio.npz'
  np.savez_compressed(path, lengths=lengths, chunk=chunk, **out)
  print('wrote', path, out['power_tags'][:8], out['power_probs'][:4])
