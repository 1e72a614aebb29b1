"""HBM-resident replay with the reference's `embodied.Replay` interface.

`DeviceReplay` keeps the `FixedLength` semantics of the reference
(embodied/replay/fixed_length.py:10-81 over a `RAMStore`, replay/store.py:10-57):
whole episodes go in, uniformly sampled episode + uniformly sampled start give
chunks of `chunk` steps whose `is_first[0]` is forced to True, episodes shorter
than `chunk` / `minlen` are skipped, `log_` keys are dropped, dtypes are
canonicalised as `embodied.convert` does (core/convert.py:4-23), and the oldest
episodes are evicted while more than `capacity` steps are stored.

What differs is where the bytes live.  Every key is one ring `[ring_steps, ...]`
in HBM (uint8 images stay uint8: 12 KiB per 64x64x3 step, so 10^6 steps are
12.3 GB of the 288 GB); an episode is one contiguous row range.  A training
minibatch is assembled on the device by `dd_replay_gather` (one 16-byte-lane
copy kernel per key) from B sampled row offsets, and `Agent.train` consumes the
device tensors directly: the learner never waits for PCIe.  The sampler is the
reference's: the same `np.random.RandomState(0)` call sequence on the host, so
with the same episodes inserted in the same order it returns the same chunks
(tests/test_replay.py checks this against the reference class itself and against
committed golden picks).

`dataset()` is the drop-in generator of host numpy chunks for unmodified callers
(`run/learning.py:25`, `run/train.py:65`); `Agent.dataset(replay.dataset)`
recognises a DeviceReplay and switches to device batches.

Episodes travel between processes in the reference's on-disk format
(`DiskStore._format/_save`, replay/store.py:123-153): `save(directory)` /
`load(directory)` write and read `<time>-<uuid>-len<L>-rew<R>.npz`.
"""

import calendar
import collections
import io
import pathlib
import time as timelib
import uuid

import numpy as np
import torch

# wire dtypes of a replay key by numpy kind (the table of reference core/convert.py:4-9):
# floats -> float32, signed integers -> int64, uint8 and bool unchanged
_WIRE = {'f': np.float32, 'i': np.int64, 'b': np.bool_}


def convert(value):
  """Canonical wire dtype of one replay value (behaviour of `embodied.convert`,
  reference core/convert.py:12-23); anything else (unsigned ints wider than a byte, complex,
  strings, objects) is rejected."""
  arr = value if isinstance(value, np.ndarray) else np.array(value)
  if arr.dtype == np.uint8:
    return arr
  want = _WIRE.get(arr.dtype.kind)
  if want is None:
    raise TypeError(f'Unsupported dtype: {arr.dtype}')
  return arr if arr.dtype == want else arr.astype(want)


class DeviceReplay:

  def __init__(self, chunk=64, capacity=None, length=0, prio_starts=0.0,
               prio_ends=1.0, minlen=0, seed=0, device='cuda:0', ops=None,
               ring_steps=None, directory=None):
    self.chunk = int(chunk)
    self.capacity = capacity and int(capacity)
    self.length = length
    self.minlen = minlen
    self.prio_starts = prio_starts
    self.prio_ends = prio_ends
    self._seed = int(seed)
    self.random = np.random.RandomState(seed=seed)
    self.device = torch.device(device)
    self._ops = ops
    if ring_steps is None:
      # eviction keeps <= capacity steps live (+ the episode being inserted); the slack
      # absorbs the unused tail left when an episode does not fit before the wrap
      ring_steps = (self.capacity + max(self.capacity // 4, 8 * self.chunk)
                    if self.capacity else 1_000_000)
    self.ring_steps = int(ring_steps)
    self.directory = directory and pathlib.Path(directory)
    self.rings = None            # key -> [ring_steps, ...] device tensor
    self.table = collections.OrderedDict()  # episode id -> (offset, length), oldest first
    self.steps = 0
    self.head = 0
    self.ongoing = collections.defaultdict(lambda: collections.defaultdict(list))
    self._saved = set()
    self.stamps = {}             # episode id -> insertion time (epoch seconds, strictly increasing)
    self._seq = 0
    self._out = {}               # batch size -> output buffers

  def reseed(self, rank):
    """Rank-dependent sampling stream (data-parallel learners with a rank-sharded dataset:
    Agent.dataset calls this once, so replays holding the same episodes draw different rows)."""
    base = getattr(self, '_seed', 0)
    self.random = np.random.RandomState(seed=base + 7919 * int(rank))
    prios = getattr(self, 'prios', None)
    if prios is not None:
      prios.random = np.random.RandomState(seed=base + 7919 * int(rank))
    self._rank_seeded = True

  # ------------------------------------------------------------ embodied.Replay

  def __len__(self):
    return self.steps

  @property
  def stats(self):
    return {'replay_steps': self.steps, 'replay_trajs': len(self.table)}

  def add(self, tran, worker=0):
    """One transition of `worker`'s running episode (FixedLength.add, reference
    replay/fixed_length.py:38-44): a transition flagged is_first starts the episode over,
    is_last - or reaching `length` steps when set - hands it to add_traj."""
    episode = self.ongoing[worker]
    if tran['is_first']:
      episode.clear()
    for name, value in tran.items():
      episode[name].append(value)
    steps = len(episode['is_first'])
    if tran['is_last'] or (self.length and steps >= self.length):
      del self.ongoing[worker]
      self.add_traj(episode)

  def add_traj(self, traj, key=None, stamp=None):
    """Store one whole episode {key: [steps, ...]} (FixedLength.add_traj, reference
    replay/fixed_length.py:46-52); returns its id, or None when it is shorter than a chunk
    (or `minlen`) and therefore skipped."""
    length = min(len(v) for v in traj.values())
    if length < max(self.chunk, self.minlen):
      print(f'Skipping short trajectory of length {length}.')
      return None
    traj = {k: convert(v) for k, v in traj.items() if not k.startswith('log_')}
    if length > self.ring_steps:
      raise ValueError(f'episode of {length} steps exceeds the ring ({self.ring_steps})')
    if self.rings is None:
      self.rings = {
          k: torch.empty((self.ring_steps,) + v.shape[1:],
                         dtype=torch.from_numpy(v[:1]).dtype, device=self.device)
          for k, v in traj.items()}
    if set(traj) != set(self.rings):
      raise KeyError(f'episode keys {sorted(traj)} != replay keys {sorted(self.rings)}')
    if self.head + length > self.ring_steps:
      self.head = 0
    lo, hi = self.head, self.head + length
    # the ring is written in order, so whatever overlaps the new range is the oldest data
    for old in [k for k, (o, n) in self.table.items() if o < hi and lo < o + n]:
      self._drop(old)
    for k, v in traj.items():
      src = torch.from_numpy(np.ascontiguousarray(v))
      self.rings[k][lo:hi].copy_(src.reshape(self.rings[k][lo:hi].shape))
    key = key or uuid.uuid4().hex
    self.table[key] = (lo, length)
    # insertion time (DiskStore stamps a file when the episode is inserted, store.py:141-145),
    # kept per episode and made strictly increasing (episodes arriving within one second get
    # consecutive seconds), so that save() names files in insertion order and sorted(glob) -
    # the order DiskStore.sync / load() reads them back in - is that order.  The format stays
    # the reference's parseable '%Y%m%dT%H%M%S'.
    self._seq = max(int(timelib.time()), self._seq + 1) if stamp is None else max(stamp, self._seq)
    self.stamps[key] = self._seq
    self.steps += length
    self.head = hi
    while self.capacity and len(self.table) > 1 and self.steps > self.capacity:
      self._drop(next(iter(self.table)))  # RAMStore._enforce_limit, store.py:51-56
    return key

  def dataset(self):
    """Host-side chunks (numpy) for callers that batch themselves."""
    while True:
      pick = self._pick()
      if pick is None:
        print('Waiting for episodes.')
        timelib.sleep(1)
        continue
      start = pick[1]
      chunk = {k: r[start:start + self.chunk].cpu().numpy() for k, r in self.rings.items()}
      chunk['is_first'] = np.zeros(self.chunk, bool)
      chunk['is_first'][0] = True
      yield chunk

  def prioritize(self, keys, priorities):
    """Uniform replay keeps no priorities (reference FixedLength has none either); the
    prioritised variant is DevicePrioritized below."""

  def save(self, directory=None):
    """Write every not-yet-written episode as `<time>-<id>-len<L>-rew<R>.npz`
    (store.py:123-153) and return the directory (None without one)."""
    directory = pathlib.Path(directory) if directory else self.directory
    if directory is None:
      return None
    directory.mkdir(parents=True, exist_ok=True)
    for key, (off, n) in self.table.items():
      if key in self._saved:
        continue
      traj = {k: r[off:off + n].cpu().numpy() for k, r in self.rings.items()}
      stamp = timelib.strftime('%Y%m%dT%H%M%S', timelib.gmtime(self.stamps[key]))
      reward = str(int(traj['reward'].sum())).replace('-', 'm') if 'reward' in traj else '0'
      with io.BytesIO() as stream:
        np.savez_compressed(stream, **traj)
        (directory / f'{stamp}-{key}-len{n}-rew{reward}.npz').write_bytes(stream.getvalue())
      self._saved.add(key)
    return str(directory)

  def load(self, data=None):
    """Read the newest episodes of a directory up to `capacity` steps, oldest of
    them first (DiskStore.sync, store.py:106-119)."""
    directory = pathlib.Path(data) if data else self.directory
    if directory is None or not directory.exists():
      return
    selected, steps = [], 0
    for filename in reversed(sorted(directory.glob('*.npz'))):
      _, key, length, _ = filename.stem.split('-')
      length = int(length[3:])
      if self.capacity and steps + length > self.capacity:
        break
      selected.append((filename, key))
      steps += length
    for filename, key in reversed(selected):
      if key in self.table:
        continue
      with np.load(filename) as f:
        traj = {k: f[k] for k in f.keys()}
      stamp = calendar.timegm(timelib.strptime(filename.stem.split('-')[0], '%Y%m%dT%H%M%S'))
      if self.add_traj(traj, key=key, stamp=stamp) is not None:
        self._saved.add(key)

  # ------------------------------------------------------------------- native

  @property
  def ops(self):
    if self._ops is None:
      from . import hipops
      self._ops = hipops.HipOps(self.device, ws_bytes=1 << 20)
    return self._ops

  def _drop(self, key):
    _, n = self.table.pop(key)
    self.steps -= n
    self._saved.discard(key)
    self.stamps.pop(key, None)

  def _pick(self):
    """FixedLength._sample, fixed_length.py:64-77: (episode id, first ring row)."""
    keys = tuple(self.table.keys())
    if not keys:
      return None
    key = keys[self.random.randint(0, len(keys))]
    offset, total = self.table[key]
    lower = 0
    upper = total - self.chunk + 1
    if self.prio_starts:
      lower -= int(self.chunk * self.prio_starts)
    if self.prio_ends:
      upper += int(self.chunk * self.prio_ends)
    index = self.random.randint(lower, upper)
    index = int(np.clip(index, 0, total - self.chunk))
    return key, offset + index

  def sample_batch(self, batch):
    """One [batch, chunk, ...] minibatch as device tensors (buffers are reused by the
    next call on the same stream)."""
    picks = []
    while len(picks) < batch:
      pick = self._pick()
      if pick is None:
        raise RuntimeError('DeviceReplay.sample_batch: no episodes stored')
      picks.append(pick[1])
    return self._gather(picks, batch)

  def _gather(self, picks, batch):
    """Assemble the chunks starting at ring rows `picks` on the device (dd_replay_gather)."""
    starts = torch.tensor(picks, dtype=torch.int64).to(self.device, non_blocking=True)
    out = self._out.get(batch)
    if out is None:
      out = {k: torch.empty((batch, self.chunk) + tuple(r.shape[1:]), dtype=r.dtype,
                            device=self.device) for k, r in self.rings.items()}
      out['is_first'] = torch.empty((batch, self.chunk), dtype=torch.bool, device=self.device)
      self._out[batch] = out
    for k, r in self.rings.items():
      if k != 'is_first':
        self.ops.replay_gather(r, starts, out[k])
    self.ops.replay_gather(None, starts, out['is_first'], first_flag=True)
    return out

  def batches(self, batch):
    while True:
      if not self.table:
        print('Waiting for episodes.')
        timelib.sleep(1)
        continue
      yield self.sample_batch(batch)


class PriorityTable:
  """Two-level sampling distribution over (episode, chunk start) with per-step priorities:
  the behaviour of the reference's `Priorities` (embodied/replay/prios.py:8-152), restated.

  Every episode holds one priority per step.  `aggregate(steps)` maps them to one
  non-negative weight per chunk start (Prioritized builds it as a windowed sum,
  prioritized.py:25-31).  Within an episode, starts are drawn from
  `fraction * weights/sum + (1 - fraction) * uniform`, where `uniform` up-weights the first /
  last start by `(len(steps) - n_starts) * prio_starts / prio_ends` (prios.py:126-131);
  episodes are drawn from `fraction * totals/sum + (1 - fraction) * n_starts/sum`
  (prios.py:134-148).  Infinite weights (fresh, never-trained chunks) win outright: only they
  keep probability mass in the weighted part (prios.py:122-124, 139-141).  Draws use an own
  `RandomState(0)` with the reference's call sequence (prios.py:52-64)."""

  def __init__(self, aggregate, fraction, prio_starts, prio_ends):
    self.aggregate, self.fraction = aggregate, fraction
    self.prio_starts, self.prio_ends = prio_starts, prio_ends
    self.random = np.random.RandomState(seed=0)
    self.steps, self.start_probs, self.totals = {}, {}, {}   # per episode id
    self._episode_probs = None
    self.samples = collections.defaultdict(int)
    self.update_min, self.update_max = np.inf, -np.inf

  def __len__(self):
    return len(self.steps)

  def _refresh(self, key):
    weights = self.aggregate(self.steps[key])
    assert (weights >= 0).all(), weights
    self.totals[key] = weights.sum()          # before infinities are replaced
    fresh = np.isposinf(weights)
    if fresh.any():
      weights = fresh.astype(np.float64)
    n = len(weights)
    uniform = np.full(n, 1.0 / n)
    if (self.prio_starts or self.prio_ends) and n > 1:
      # (an episode of exactly one chunk has a single start: the reference's up-weighting,
      # prios.py:88-92, multiplies it by both factors and divides by a zero sum -> NaN
      # probabilities; such an episode keeps its plain uniform start instead)
      extra = len(self.steps[key]) - n
      up = uniform.copy()
      up[0] *= extra * self.prio_starts
      up[-1] *= extra * self.prio_ends
      if up.sum() > 0:
        uniform = up / up.sum()
    mass = weights.sum()
    weighted = uniform if mass == 0 else weights / mass
    self.start_probs[key] = self.fraction * weighted + (1 - self.fraction) * uniform
    self._episode_probs = None

  def add(self, key, prios):
    self.steps[key] = np.asarray(prios, np.float64).copy()
    self._refresh(key)

  def update(self, key, index, prios):
    prios = np.asarray(prios, np.float64)
    self.update_min = min(self.update_min, prios.min())
    self.update_max = max(self.update_max, prios.max())
    if key not in self.steps or not 0 <= index <= len(self.steps[key]):
      raise KeyError(key)
    try:
      self.steps[key][index:index + len(prios)] = prios
    except ValueError:
      raise KeyError(key)
    self._refresh(key)

  def remove(self, key):
    for table in (self.steps, self.start_probs, self.totals, self.samples):
      table.pop(key, None)
    self._episode_probs = None

  def _episodes(self):
    if self._episode_probs is None:
      keys = tuple(self.steps)
      counts = np.array([len(self.start_probs[k]) for k in keys], np.float64)
      totals = np.array([self.totals[k] for k in keys], np.float64)
      fresh = np.isposinf(totals)
      if fresh.any():
        totals = fresh.astype(np.float64)
      mass = totals.sum()
      weighted = np.full(len(keys), 1.0 / len(keys)) if mass == 0 else totals / mass
      self._episode_probs = (keys, self.fraction * weighted + (1 - self.fraction) * counts / counts.sum())
    return self._episode_probs

  def sample(self):
    """(episode id, chunk start, probability of that draw)."""
    keys, probs = self._episodes()
    if len(keys) == 1:
      key, prob = keys[0], 1.0
    else:
      pos = self.random.choice(len(probs), p=probs)
      key, prob = keys[pos], probs[pos]
    inner = self.start_probs[key]
    index = self.random.choice(len(inner), p=inner)
    self.samples[key] += 1
    return key, index, prob * inner[index]

  @property
  def stats(self):
    if len(self) <= 1:
      return {}
    _, probs = self._episodes()
    counts = list(self.samples.values()) or [0]
    return {
        'randomness': float(-(probs @ np.log(probs)) / np.log(len(probs))),
        'seen_frac': len(self.samples) / len(self.steps), 'seen_max': max(counts),
        'sample_frac': sum(counts) / len(self.steps),
        'update_min': self.update_min, 'update_max': self.update_max}


class DevicePrioritized(DeviceReplay):
  """`embodied.replay.Prioritized` (reference replay/prioritized.py:12-135) over the HBM episode
  ring.  Every sampled chunk carries `key` int64[T,3] (episode uuid + chunk start of the
  priority draw, prioritized.py:114-124) and `prob` float64[T]; `Agent.train` hands them back
  with the per-step priorities (`outs['key'], outs['priority']`, agent.py:89-93) and the run
  loop calls `prioritize` (run/learning.py:57-58).  A drawn chunk cools down to priority 0
  (-inf with softmax) until it is re-prioritised (prioritized.py:33-36, 97).

  Faithful to the reference AS WRITTEN: the steps that are returned are chosen by the same
  uniform rule as FixedLength (prioritized.py:100-110 draw the episode and start again from
  `self.random`, independently of the priority draw) - the priority draw only determines the
  `key` / `prob` bookkeeping; and episodes evicted from the ring are dropped from the table
  (the reference leaves them behind, its TODO at prioritized.py:17-18, and ignores late
  priorities for them - as does `prioritize` here)."""

  def __init__(self, chunk=64, capacity=None, prio_starts=0.0, prio_ends=1.0, fraction=0.1,
               softmax=False, temp=1.0, constant=0.0, exponent=0.5, **kw):
    super().__init__(chunk=chunk, capacity=capacity, prio_starts=prio_starts,
                     prio_ends=prio_ends, **kw)
    def aggregate(prios):  # prioritized.py:25-31
      if softmax:
        prios = np.maximum(np.exp(prios / temp) + constant, 0)
      else:
        prios = np.abs(prios) ** exponent
      return np.convolve(prios, np.ones(self.chunk), 'valid')
    self.prios = PriorityTable(aggregate, fraction, prio_starts, prio_ends)
    self.cooldown = np.full(self.chunk, -np.inf if softmax else 0.0, np.float64)
    self.handed_out = set()

  @property
  def stats(self):
    return {**super().stats, **self.prios.stats}

  def add_traj(self, traj, key=None, stamp=None):
    key = super().add_traj(traj, key, stamp)
    if key is not None:
      self.prios.add(key, np.full(self.table[key][1], np.inf, np.float64))
    return key

  def _drop(self, key):
    super()._drop(key)
    self.prios.remove(key)

  @staticmethod
  def encode(key, index):
    return np.frombuffer(uuid.UUID(key).bytes + int(index).to_bytes(8, 'big'), np.int64)

  @staticmethod
  def decode(code):
    raw = np.asarray(code, np.int64).tobytes()
    return uuid.UUID(bytes=raw[:16]).hex, int.from_bytes(raw[16:], 'big')

  def prioritize(self, keys, priorities):
    keys = np.asarray(keys, np.int64)[:, 0]     # replicated along the time axis
    priorities = np.asarray(priorities, np.float64)
    assert priorities.shape == (len(keys), self.chunk), priorities.shape
    for code, prio in zip(keys, priorities):
      assert tuple(code.tolist()) in self.handed_out, code
      key, index = self.decode(code)
      try:
        self.prios.update(key, index, prio)
      except KeyError:
        print('Received priorities for an episode that was already removed.')

  def _pick_prio(self):
    """One chunk: (ring row of its first step, key code int64[3], prob); None when empty."""
    if not self.table:
      return None
    key, index, prob = self.prios.sample()
    self.prios.update(key, index, self.cooldown)
    code = self.encode(key, index)
    self.handed_out.add(tuple(code.tolist()))
    return self._pick()[1], code, prob

  def dataset(self):
    while True:
      pick = self._pick_prio()
      if pick is None:
        print('Waiting for episodes.')
        timelib.sleep(1)
        continue
      start, code, prob = pick
      chunk = {k: r[start:start + self.chunk].cpu().numpy() for k, r in self.rings.items()}
      chunk['is_first'] = np.zeros(self.chunk, bool)
      chunk['is_first'][0] = True
      chunk['key'] = np.repeat(code[None], self.chunk, axis=0)
      chunk['prob'] = np.repeat(np.float64(prob), self.chunk)
      yield chunk

  def sample_batch(self, batch):
    picks = [self._pick_prio() for _ in range(batch)]
    if any(p is None for p in picks):
      raise RuntimeError('DevicePrioritized.sample_batch: no episodes stored')
    out = self._gather([p[0] for p in picks], batch)
    out = dict(out)
    codes = np.stack([p[1] for p in picks])                       # [B, 3]
    out['key'] = torch.from_numpy(np.repeat(codes[:, None], self.chunk, 1).copy()).to(self.device)
    out['prob'] = torch.from_numpy(np.repeat(np.array([p[2] for p in picks], np.float64)[:, None],
                                             self.chunk, 1).copy()).to(self.device)
    return out
