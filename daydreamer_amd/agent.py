"""`Agent`: the reference's `embodied.Agent` surface (reference
embodied/core/base.py:1-30, implemented there by tfagent.py:22-96 +
agent.py:15-139) on top of the MI355X learner.

  Agent(obs_space, act_space, step, config)
  .configs                      named config blocks (agent.py:18-19)
  .dataset(generator_fn)        batches of [B,T,...] numpy (agent.py:108-121)
  .policy(obs, state, mode)     -> ({'action': np}, state)     (agent.py:42-65)
  .train(data, state)           -> (outs, state, metrics)      (agent.py:67-93)
  .report(data)                 -> metrics                     (agent.py:95-106)
  .save() / .load(data)         picklable dict (tfutils.py:116-131)

Data parallelism mirrors tfagent.py:99-116: `train` receives the GLOBAL batch
and every rank (one process per GPU, torch.distributed over RCCL) takes rows
[rank*B/P, (rank+1)*B/P); gradients and the batch statistics feeding the
controllers are summed over ranks.
"""

import atexit
import collections.abc
import os
import queue
import threading
import time
import weakref

import numpy as np
import torch

from . import config as config_mod
from . import graphs
from . import learner as learner_mod
from . import replay as replay_mod
from . import spec as spec_mod


class DistComm:
  """Sum / max all-reduce over the data-parallel group (RCCL on GPUs)."""

  def __init__(self, group=None):
    import torch.distributed as dist
    self.dist = dist
    self.group = group  # None: the default group
    self.rank, self.world = dist.get_rank(), dist.get_world_size()

  def allreduce_sum(self, t):
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

  def allreduce_max(self, t):
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)


class ShardedBatch(dict):
  """A minibatch that already is this rank's rows of the global batch (produced by a
  rank-sharded `Agent.dataset`): `Agent.train` takes it whole."""


class Batcher:
  """`batch_size` replay generators zipped into [B,T,...] minibatches by a prefetch thread
  (role of embodied.Prefetch, reference core/prefetch.py:15-67).  With a GPU `device` the
  thread also stages the minibatch: it stacks into rotating pinned host buffers and issues
  the host-to-device copies on the process's copy stream, so `Agent.train` receives device
  tensors (its own upload becomes a device-to-device copy) and the PCIe transfer of minibatch
  k+1 overlaps the train step of minibatch k.

  Thread discipline: all pinned buffer sets are allocated at once when the first minibatch is
  known, and every HIP runtime call of the thread (allocations, copy issue, event record /
  query) is made under graphs.API_LOCK - the lock graph capture and graph launches hold - so
  the thread is never inside the runtime while the learner thread captures or launches a
  graph.  The thread waits for events by polling outside the lock.  `close()` (also run at
  interpreter exit and when the owning agent is collected) stops the thread; a daemon thread
  left inside a torch call at interpreter shutdown aborts the process."""

  _LIVE = weakref.WeakSet()

  def __init__(self, generator_fn, batch_size, prefetch=2, device=None, sharded=False):
    # `sharded`: batch_size is this rank's share of the global batch; minibatches are
    # marked so that Agent.train does not slice them again (class ShardedBatch)
    self._sharded = sharded
    self._gens = [generator_fn() for _ in range(batch_size)]
    self._queue = queue.Queue(maxsize=prefetch)
    self._error = None
    self._stop = threading.Event()
    self._device = device if (device is not None and torch.device(device).type == 'cuda') else None
    self._sets = prefetch + 2  # pinned buffer sets in rotation
    self._stream = graphs.stream(self._device, 'copy') if self._device is not None else None
    self._thread = threading.Thread(target=self._work, daemon=True)
    Batcher._LIVE.add(self)
    self._thread.start()

  def _put(self, item):
    while not self._stop.is_set():
      try:
        self._queue.put(item, timeout=0.05)
        return True
      except queue.Full:
        continue
    return False

  def _wait(self, ev):
    """Event wait without blocking inside the runtime: query under the lock, sleep outside."""
    while not self._stop.is_set():
      with graphs.API_LOCK:
        if ev.query():
          return
      time.sleep(0.0002)

  def _work(self):
    try:
      if self._device is None:
        while not self._stop.is_set():
          items = [next(g) for g in self._gens]
          if not self._put(self._mark({k: np.stack([it[k] for it in items], 0) for k in items[0]})):
            return
        return
      with graphs.API_LOCK:
        torch.cuda.set_device(self._device)
      stream = self._stream
      pinned, done = None, [None] * self._sets
      n = 0
      while not self._stop.is_set():
        items = [next(g) for g in self._gens]
        i = n % self._sets
        n += 1
        if done[i] is not None:
          self._wait(done[i])  # the copy out of this pinned set has finished
          if self._stop.is_set():   # (_wait returned early: the copy may still read the set)
            return
        if pinned is None:
          with graphs.API_LOCK:
            pinned = [{
                k: torch.empty((len(items),) + np.shape(v), dtype=torch.from_numpy(np.asarray(v)[None]).dtype
                               ).pin_memory() for k, v in items[0].items()} for _ in range(self._sets)]
        host = pinned[i]
        for k, t in host.items():
          np.stack([it[k] for it in items], 0, out=t.numpy())
        with graphs.API_LOCK:
          with torch.cuda.stream(stream):
            dev = {k: t.to(self._device, non_blocking=True) for k, t in host.items()}
            ev = torch.cuda.Event()
            ev.record(stream)
        done[i] = ev
        if not self._put((self._mark(dev), ev)):
          return
    except Exception as e:  # surfaced on the consumer side
      self._error = e
      self._put(None)

  def _mark(self, batch):
    return ShardedBatch(batch) if self._sharded else batch

  def close(self, join=False):
    """Stop the prefetch thread (idempotent).  Non-blocking by default: it may be called from
    a finalizer while the caller holds graphs.API_LOCK, which the thread may be waiting for."""
    self._stop.set()
    t = self._thread
    if join and t is not threading.current_thread() and t.is_alive():
      t.join(timeout=10.0)

  def __iter__(self):
    return self

  def __next__(self):
    # poll: close() from another thread / a finalizer, or a worker that left through a refused
    # _put, must end the iteration instead of leaving the consumer blocked in get() forever
    while True:
      if self._stop.is_set():
        raise StopIteration
      try:
        batch = self._queue.get(timeout=0.05)
        break
      except queue.Empty:
        if not self._thread.is_alive() and self._queue.empty():
          if self._error is not None:
            raise self._error
          raise StopIteration
    if batch is None:
      raise self._error
    if self._device is None:
      return batch
    dev, ev = batch
    with graphs.API_LOCK:
      cur = torch.cuda.current_stream(self._device)
      cur.wait_event(ev)
      for t in dev.values():
        t.record_stream(cur)  # allocated on the copy stream, consumed on this one
    return dev


@atexit.register
def _close_batchers():
  for b in list(Batcher._LIVE):
    b.close(join=True)


class LazyScalar:
  """One metric of a pipelined train call; the device is waited for when it is looked at
  (float(), np.asarray(), format, arithmetic, comparison), not when train() returns."""

  __slots__ = ('_owner', '_key')

  def __init__(self, owner, key):
    self._owner, self._key = owner, key

  def _value(self):
    return self._owner.resolve()[self._key]

  def __array__(self, dtype=None, copy=None):
    a = np.asarray(self._value())
    return a if dtype is None else a.astype(dtype)

  def __float__(self):
    return float(self._value())

  def __int__(self):
    return int(self._value())

  def __bool__(self):
    return bool(self._value())

  def __format__(self, spec):
    return format(self._value(), spec)

  def __repr__(self):
    return repr(self._value())

  def item(self):
    return np.asarray(self._value()).item()

  def __add__(self, o): return self._value() + o
  def __radd__(self, o): return o + self._value()
  def __sub__(self, o): return self._value() - o
  def __rsub__(self, o): return o - self._value()
  def __mul__(self, o): return self._value() * o
  def __rmul__(self, o): return o * self._value()
  def __truediv__(self, o): return self._value() / o
  def __rtruediv__(self, o): return o / self._value()
  def __neg__(self): return -self._value()
  def __abs__(self): return abs(self._value())
  def __floordiv__(self, o): return self._value() // o
  def __rfloordiv__(self, o): return o // self._value()
  def __mod__(self, o): return self._value() % o
  def __rmod__(self, o): return o % self._value()
  def __pow__(self, o): return self._value() ** o
  def __rpow__(self, o): return o ** self._value()
  def __pos__(self): return +self._value()
  def __lt__(self, o): return self._value() < o
  def __le__(self, o): return self._value() <= o
  def __gt__(self, o): return self._value() > o
  def __ge__(self, o): return self._value() >= o
  # value equality like the numpy scalar it stands for (the default would compare identities
  # and `mets[k] == 0.0` would silently be False); unhashable like any object with value __eq__
  def __eq__(self, o): return self._value() == (o._value() if isinstance(o, LazyScalar) else o)
  def __ne__(self, o): return self._value() != (o._value() if isinstance(o, LazyScalar) else o)
  __hash__ = None


class LazyMetrics(collections.abc.Mapping):
  """The metrics dict of ONE pipelined train call - that call's own metrics, as the reference's
  train() returns them (tfagent.py train -> _convert_mets) - read from the device when first
  looked at.  A loop that only collects the values (`metrics[key].append(value)`, run/train.py)
  and aggregates them at its log interval never waits inside train(); `float(mets[k])` right
  after the call waits for that step (and serialises the two streams).  At the latest it is
  fetched when the call after the next one is enqueued (its snapshot slot is reused then).
  Data parallel: the fetch all-reduces the statistics over the ranks (on the read-out
  communicator), so a call's metrics must be looked at EARLY on every rank or on none - a script
  that aggregates on every rank and only prints on rank 0 does that; a look that only rank 0
  takes before the next train call would wait for collectives the other ranks issue later."""

  def __init__(self, keys, fetch):
    self._keys, self._fetch, self._vals, self._error = tuple(keys), fetch, None, None
    # a logger thread may look at the values while the training thread's next call resolves the
    # same step (Pipeline.step): one of them fetches, the other waits and gets the same outcome
    self._lock = threading.Lock()

  def resolve(self):
    if self._vals is None:
      with self._lock:
        if self._vals is not None:
          return self._vals
        if self._error is not None:
          raise self._error
        fetch, self._fetch = self._fetch, None
        try:
          self._vals = fetch()
        except Exception as e:   # (e.g. read_metrics: a loss is not finite) - every later look raises it again
          self._error = e
          raise
    return self._vals

  @property
  def resolved(self):
    return self._vals is not None

  @property
  def failed(self):
    """The fetch raised (and whoever looked has been handed the exception)."""
    return self._error is not None

  def __getitem__(self, key):
    if self._vals is not None:
      return self._vals[key]
    if key not in self._keys:
      raise KeyError(key)
    return LazyScalar(self, key)

  def __iter__(self):
    return iter(self._vals if self._vals is not None else self._keys)

  def __len__(self):
    return len(self._vals if self._vals is not None else self._keys)


class Pipeline:
  """Two-stream software pipeline of the train step (hip.pipeline; the shipped default on one GPU).

  A step is  A1 (world-model forward + backward)  ->  A2 (world-model optimizer,
  hand-over copies)  ->  B (imagination, critic and actor updates).  B(k) only reads
  world-model weights and its own buffers, A1(k+1) only reads weights: they are
  independent, and between them they mix the GPU-filling convolution / head
  contractions with the latency-bound recurrent scans.  So B(k) is replayed on a
  second stream while A1(k+1) runs on the first; A2(k+1), which writes the world-model
  weights, waits for B(k).  The arithmetic and its order inside every phase are those
  of the sequential step (parameters after n steps are bit-identical,
  tests/test_learner_gpu.py).  Every call returns ITS OWN metrics as a `LazyMetrics`: they are
  copied out of the per-step snapshot when first looked at, or when the next call has been
  enqueued, whichever comes first.  A non-finite loss / gradient norm of step k therefore raises
  (FloatingPointError, the check_numerics contract of tfutils.py:207,249) when step k's metrics
  are looked at, at the latest inside train call k + 1; the update that produced it was skipped
  on the device either way (k_adam), exactly as in the sequential schedule.

  Streams.  Both phases run on dedicated library-owned streams (work queued on the default
  stream does not run next to other streams; equal priorities: a high-priority stream for either
  phase was measured at 62 / 99 instead of 43.5 ms per step).  Which two matters: ROCm multiplexes
  HIP streams onto a few hardware queues in creation order, and the steady-state step time falls
  into three classes by the pair the phases (and their graphs' internal branches) land on -
  29.4 / 31.1 / 33.5 ms at configs[1] - and which pairs are in the fast class differs from
  process to process (profiles/r05_pipe_pairs.txt).  So the pair is MEASURED by every pipelined
  learner inside its first pipelined train calls: every ordered
  pair of a pool of four streams runs TRIAL consecutive real train steps, the time between the
  behaviour-phase ends of its last steps (device events: steady state, both phases of consecutive
  steps in flight) is its period, the fastest pair is kept and the
  other pairs' graphs are retired.  12 pairs x 4 steps; parameters stay bit-identical through the
  switches.  `hip.tune_pipeline: false` (or DD_PIPE_TUNE=0) skips it and runs DEFAULT_PAIR (or
  DD_PIPE_PAIR=a,b).  (Rounds 2-4 measured three-step trials from their first tick: the transient
  after a switch, not the period - their choice was noise.)
  """

  DEFAULT_PAIR = (1, 3)
  POOLS = {}  # device -> ([4 phase streams], read-out stream)
  BEST = {}   # device -> the pair the latest measurement of this process selected (a record)
  TRIAL = 4   # train steps per candidate pair: one after the switch, one more, two measured periods

  def __init__(self, learner, device, comm=None, tune=True):
    self.L = learner
    self.device = device
    self.comm = comm  # data-parallel: communicator of the metric read-out
    key = str(torch.device(device))
    if key not in Pipeline.POOLS:
      Pipeline.POOLS[key] = ([graphs.stream(device, f'pipe{i}') for i in range(4)],
                             graphs.stream(device, 'read'))
    self.key = key
    self.pool, self.s3 = Pipeline.POOLS[key]          # s3: metric read-out
    self.cands = [(a, b) for a in range(4) for b in range(4) if a != b]
    self.periods = {}
    # one set of captured graphs per stream pair: a graph executable is only ever
    # launched on one stream (relaunching it on another one crashes the runtime)
    self.plans = {}
    env = os.environ.get('DD_PIPE_PAIR')
    pair = tuple(int(x) for x in env.split(',')) if env else Pipeline.DEFAULT_PAIR
    assert pair in self.cands, pair
    self.pair = None
    self._use_pair(*pair)
    # (every Pipeline measures for itself: the class a pair falls into also depends on how the
    # runtime laid out THIS learner's graph executables - a second agent of one process ran
    # 31.4 ms on the pair the first one had measured at 29.4, tools/interleaved_loop.py)
    self.tuned = not tune or os.environ.get('DD_PIPE_TUNE', '1') == '0' or env is not None
    self.k_tune, self.ticks = 0, []
    self._drained = False   # flush() ran since the previous step: restart the current trial
    self.ev_in = torch.cuda.Event()
    self.stages = [None, None]           # input staging buffers by step parity (Pipeline.stage)
    self.ev_commit = [torch.cuda.Event(), torch.cuda.Event()]
    self.ev_a = torch.cuda.Event()
    self.ev_b = [torch.cuda.Event(), torch.cuda.Event()]
    live = learner.metric_tensors()
    self.pub_a = [{k: torch.empty_like(v) for k, v in live.items()} for _ in range(2)]
    self.pub_b = [{k: torch.empty_like(v) for k, v in live.items()} for _ in range(2)]
    self.k = 0
    self.pending = None  # parity of the step whose metrics have not been fetched yet
    self.handle = None   # its LazyMetrics
    self.keys = ()       # metric names (those of the eager first step)

  def _use_pair(self, a, b):
    """Switch to streams (a, b) of the pool; the caller has drained the pipeline."""
    if (a, b) not in self.plans:
      torch.cuda.synchronize(self.device)
      self.plans[(a, b)] = self.L.capture_pipeline()
    self.pa1, self.pa2, self.pb = self.plans[(a, b)]
    self.s1, self.s2 = self.pool[a], self.pool[b]
    self.pair = (a, b)

  @property
  def n_graphs(self):
    return self.pa1.n_graphs + self.pa2.n_graphs + self.pb.n_graphs

  def _publish(self, pub, stream, clear=False):
    with torch.cuda.stream(stream):
      for k, v in self.L.metric_tensors().items():
        pub[k].copy_(v)
      if clear and getattr(self.L, 'scan_sync', None) is not None:
        # the persistent scans' sticky error word belongs to the step whose snapshot took it:
        # cleared here, in stream order, so that the next step's snapshot holds its own errors only
        self.L.scan_sync[1:2].zero_()

  def stage(self):
    """The input staging buffers of the step about to be enqueued (two sets, by step parity), ready
    to be written on the caller's current stream: that stream waits for the copy that last read
    them (the commit of the step before the previous one - long done)."""
    par = self.k & 1
    if self.stages[par] is None:
      self.stages[par] = self.L.input_stage()
    torch.cuda.current_stream(self.device).wait_event(self.ev_commit[par])
    return self.stages[par]

  def step(self, staged=False):
    """Enqueue one step; returns its metrics as a LazyMetrics.  The previous step's metrics
    are fetched here (after this step's world-model phase has been enqueued), if the caller has
    not looked at them yet.  The caller's current stream holds the uploaded inputs: in the
    learner's input buffers, or (staged) in stage() - then they are copied into the input buffers
    here, on the world-model stream behind the previous step's optimizer: the upload itself did
    not have to wait for the previous world-model phase to finish reading them (it used to, and
    the next world-model phase started ~1 ms after the behaviour phase it runs next to)."""
    cur = torch.cuda.current_stream(self.device)
    if not self.tuned:
      self._tune_step()   # (the stream pair of this step, while the pairs are being measured)
    s1, s2 = self.s1, self.s2
    par = self.k & 1
    s1.wait_stream(cur)                    # inputs / carry reset issued by the caller
    if self.k > 0:
      s1.wait_event(self.ev_a)             # (the previous step may have used other streams)
    if staged:
      with torch.cuda.stream(s1):
        self.L.commit_inputs(self.stages[par])
      self.ev_commit[par].record(s1)
    self.pa1.replay_on(s1)
    self.ev_in.record(s1)
    # A2 = [data-parallel all-reduce of the world-model gradients] + [grad norm, Adam, hand-over].
    # Only the second part must wait for B(k-1), which still reads the world-model weights:
    # the collective (it touches nothing but the model gradient arena) runs next to B(k-1).
    items = self.pa2.items
    last = max(i for i, (kind, _) in enumerate(items) if kind == 'graph')
    self.pa2.replay_on(s1, stop=last)
    if self.pending is not None:
      s1.wait_event(self.ev_b[par ^ 1])
    self.pa2.replay_on(s1, start=last)
    self._publish(self.pub_a[par], s1, clear=True)
    self.ev_a.record(s1)
    s2.wait_event(self.ev_a)
    if self.pending is not None:
      s2.wait_event(self.ev_b[par ^ 1])
    self.pb.replay_on(s2)
    self._publish(self.pub_b[par], s2)
    self.ev_b[par].record(s2)
    if not self.tuned:
      tick = torch.cuda.Event(enable_timing=True)
      tick.record(s2)
      self.ticks.append(tick)
    if not staged:
      cur.wait_event(self.ev_in)           # the next upload must not overtake A1's reads
    # this step's handle is in place BEFORE the previous one is resolved: if that raises (a loss
    # of step k - 1 is not finite) the step just enqueued keeps its metrics and the pipeline its
    # bookkeeping - the next call, flush(), save() ... go on from a consistent state
    prev, self.pending = self.handle, par
    self.k += 1
    self.handle = LazyMetrics(self.keys, lambda: self._read(par))
    if prev is not None and not prev.failed:
      prev.resolve()   # (its snapshot slot is the one the NEXT step publishes into)
    return self.handle

  def _tune_step(self):
    """Stream pair of the step about to be enqueued while the selection is being measured."""
    c, r = divmod(self.k_tune, self.TRIAL)
    if self._drained:
      # the pipeline was drained since the previous step (policy / report / save between two train
      # calls): the period of this trial would include the host's idle gap - start the trial over
      # on the same pair, only back-to-back steps are timed
      self._drained = False
      if r != 0:
        self.k_tune += 1 - r
        self.ticks = []
        return
    self.k_tune += 1
    if r != 0:
      return
    if c > 0:
      # the trial that just ended: two periods between the behaviour-phase ends of its steps
      # 1 .. 3 (step 0 ran next to the previous pair's last behaviour phase)
      t0, t1 = self.ticks[-3], self.ticks[-1]
      t1.synchronize()
      self.periods[self.cands[c - 1]] = t0.elapsed_time(t1) / 2
    self.ticks = []
    if c < len(self.cands):
      a, b = self.cands[c]
    else:
      a, b = min(self.periods, key=self.periods.get)
      if self.comm is not None:
        # data parallel: every rank measured its own periods in lock-step; all of them run the
        # pair rank 0 chose (different pairs per rank = different skew at every collective)
        pick = torch.tensor([a, b], dtype=torch.int64, device=self.device)
        self.comm.dist.broadcast(pick, src=0, group=self.comm.group)
        a, b = int(pick[0]), int(pick[1])
      Pipeline.BEST[self.key] = (a, b)
      self.tuned = True
    if (a, b) != self.pair:
      # (steps on different pairs are ordered by events, ev_a / ev_b; the host only waits here
      # because a first use of a pair captures its graphs)
      self.s1.synchronize()
      self.s2.synchronize()
      self._use_pair(a, b)
    if self.tuned:
      # the losing pairs' graphs (3 plans x 11 pairs) are not needed again: retire them (they are
      # destroyed by a later capture, after a device-wide synchronize - graphs.py)
      for key in [k for k in self.plans if k != (a, b)]:
        for plan in self.plans.pop(key):
          plan.release()

  def tune(self, run_step, force=False):
    """Finish the stream-pair measurement now instead of inside the next train calls: run_step()
    must perform one train step (through step()).  No-op once this pipeline has measured (or
    with the measurement switched off), unless `force`."""
    if force:
      self.k_tune, self.periods, self.ticks, self.tuned = 0, {}, [], False
    budget = 2 * (len(self.cands) * self.TRIAL + 1) + 8
    while not self.tuned:
      # (run_step() that does not come through step() - a minibatch with replay keys, which takes
      # the sequential path - would never finish the measurement)
      budget -= 1
      if budget < 0:
        raise RuntimeError('Pipeline.tune: run_step() does not perform pipelined train steps')
      run_step()

  def _read(self, par):
    """Fetch the metrics of the step with parity `par` (its snapshot slots).  May run on
    whichever thread first looks at a LazyMetrics (a logger thread): the device part holds
    graphs.API_LOCK like every other runtime call of the package."""
    L = self.L
    with graphs.API_LOCK:
      with torch.cuda.stream(self.s3):
        self.s3.wait_event(self.ev_b[par])
        a, b = self.pub_a[par], self.pub_b[par]
        rows = sorted(L.stat_b_slots)
        merged = dict(a)
        for k in L.METRIC_B:
          if k in b:
            merged[k] = b[k]
        for k in ('sums', 'maxs'):
          merged[k] = a[k].clone()
          merged[k][rows] = b[k][rows]
        if self.comm is not None:
          self.comm.allreduce_sum(merged['sums'])
          self.comm.allreduce_max(merged['maxs'])
          merged['bal'] = merged['bal'].clone()
          self.comm.allreduce_sum(merged['bal'])
          for k in L.stat_prereduced:  # identical on every rank already
            merged['sums'][k] /= L.world
        host = {k: v.cpu().numpy() for k, v in merged.items()}
    return L.read_metrics(host)

  def flush(self):
    """Wait for everything in flight; returns the last step's metrics (or None).  An error of
    the last step (non-finite loss) is raised once, here; the pipeline is drained either way."""
    try:
      # (a failure the caller has already been handed - it looked at the metrics - is not raised again)
      mets = self.handle.resolve() if self.pending is not None and not self.handle.failed else None
    finally:
      if self.pending is not None and not self.tuned:
        self._drained = True
      self.pending = self.handle = None
      cur = torch.cuda.current_stream(self.device)
      cur.wait_stream(self.s1)
      cur.wait_stream(self.s2)
      self.s2.synchronize()
    return mets


def pipeline_mode(value, world, graph=True):
  """hip.pipeline -> bool.  'auto' (the shipped default): on for a single process (world size
  1), where the two-stream schedule is 16 % faster and differs from the sequential one only in
  WHEN a call's metrics are read; off under data parallelism, where the sequential schedule
  overlaps its early all-reduce with the encoder backward and the pipeline's three communicators
  have never run on RCCL with more than one rank.  true / false force it (Config.update turns a
  bool into 'True' / 'False' once the key holds a string)."""
  v = str(value).strip().lower()
  assert v in ('auto', 'true', 'false', '1', '0', 'on', 'off'), f'hip.pipeline: {value!r}'
  if not graph:
    return False
  return world == 1 if v == 'auto' else v in ('true', '1', 'on')


class TrainState:
  """Opaque recurrent state handed back to the caller (the carried posterior
  lives in the learner's HBM buffers)."""

  def __init__(self, owner):
    self.owner = owner


class PolicyState:

  def __init__(self, latent, action):
    self.latent, self.action = latent, action


class Agent:

  configs = config_mod.load_configs()
  _warned_host_shard = False

  def __init__(self, obs_space, act_space, step, config, _ops=None,
               _device=None, _dtype=torch.float32):
    self.config = config
    self.cfg = config_mod.to_plain(config)
    self.obs_space = {k: v for k, v in obs_space.items()
                      if not k.startswith('log_')}
    self.act_space = act_space['action']
    # Discrete spaces arrive one-hot encoded (embodied.wrappers.OneHotAction:
    # float32 [n] with .discrete = True) -> 'onehot' actor trained by REINFORCE.
    self.act_discrete = bool(getattr(self.act_space, 'discrete', False))
    self.step = step
    self.act_dim = int(np.prod(self.act_space.shape))
    assert not self.act_discrete or len(self.act_space.shape) == 1, self.act_space
    shapes = {k: tuple(v.shape) for k, v in self.obs_space.items()}
    self.spec = spec_mod.build_spec(self.cfg, shapes, self.act_dim,
                                    self.act_discrete)
    self.rank, self.world, self.comm = 0, 1, None
    try:
      import torch.distributed as dist
      # (DD_FORCE_DIST=1: run the collective code path even with a single rank - a smoke
      # test of the RCCL / graph-cut / multi-communicator plumbing on a one-GPU box)
      if dist.is_available() and dist.is_initialized() and (
          dist.get_world_size() > 1 or os.environ.get('DD_FORCE_DIST') == '1'):
        self.comm = DistComm()
        self.rank, self.world = self.comm.rank, self.comm.world
    except ImportError:
      pass
    if _ops is None:
      # The product path: HIP kernels or nothing.
      from . import hipops
      local = int(os.environ.get('LOCAL_RANK', 0))
      self.device = torch.device(_device or f'cuda:{local}')
      torch.cuda.set_device(self.device)
      self.ops = hipops.HipOps(self.device)
      # arithmetic of every contraction (process-wide library switch): float32 = exact 3-way
      # bf16 split with six products (fp32-level accuracy, default); bfloat16 = the opt-in
      # reduced-precision mode (operands rounded to bf16, fp32 accumulation and storage), the
      # counterpart of the reference's tf.precision: float16 (tfagent.py:161-168)
      prec = str(self.cfg.get('hip', {}).get('precision', 'float32'))
      assert prec in ('float32', 'bfloat16'), prec
      self.ops.set_gemm_mode(1 if prec == 'bfloat16' else int(os.environ.get('DD_GEMM_MODE', 6)))
      # second launch context (own scratch workspace) for the side stream
      self.ops2 = hipops.HipOps(self.device, ws_bytes=1024 << 20)
      self.ops_b = None
      # side context of the behaviour phase (heads of finished time chunks next to the
      # imagination rollout, learner.phase_imagine)
      hipc = self.cfg.get('hip', {})
      self.ops_b2 = (hipops.HipOps(self.device, ws_bytes=1024 << 20)
                     if hipc.get('overlap_heads', True) else None)
    else:
      self.ops = _ops
      self.ops2 = None
      self.ops_b = None
      self.ops_b2 = None
      self.device = torch.device(_device or 'cpu')
    self._dtype = _dtype
    hip = self.cfg.get('hip', {})
    self._use_graph = bool(hip.get('graph', True)) and self.device.type == 'cuda'
    self._noise_seed = int(hip.get('noise_seed', 0))
    # two-stream pipeline of consecutive steps (class Pipeline): 'auto' = on for world size 1
    self._pipeline = pipeline_mode(hip.get('pipeline', 'auto'), self.world, self._use_graph)
    # `auto` also follows the caller's loop.  The pipeline only pays while train calls follow each
    # other (the learner process of run/learning.py): policy / report / save between two train
    # calls must drain it (they read weights the phase in flight writes), and a drained pipeline is
    # the sequential step WITHOUT its in-step overlaps (~1 ms slower at configs[1]).  So after
    # ADAPT train calls in a row that each followed such a call (run/train.py: act, then train) the
    # sequential plan is replayed instead, and after ADAPT train calls in a row without one the
    # pipeline again.  Both schedules produce the same parameters bit for bit, so switching is
    # invisible apart from the metrics' type (host values / LazyMetrics).
    self._adaptive = self._pipeline and str(hip.get('pipeline', 'auto')).strip().lower() == 'auto'
    self._touched, self._streak_touched, self._streak_clean, self._prefer_seq = False, 0, 0, False
    self._seq_plan = None
    # rank-sharded prefetch: every rank assembles and uploads only its own rows
    self._shard_dataset = bool(hip.get('shard_dataset', True))
    self.comm_b = self.comm_m = None
    if self._pipeline:
      from . import hipops
      self.ops_b = hipops.HipOps(self.device, ws_bytes=1024 << 20)
      if self.comm is not None:
        # the two phases' collectives are in flight at the same time on different
        # streams: one communicator each (+ one for the metric read-out); every rank
        # creates them in the same order
        import torch.distributed as dist
        self.comm_b = DistComm(dist.new_group())
        self.comm_m = DistComm(dist.new_group())
    self._pipe = None
    self._last_metrics = None
    self._seed = int(self.cfg.get('seed', 0))
    # Parameter / optimizer arenas are owned by the agent and shared by every
    # learner instance (train, policy, report), so rebuilding for a new batch
    # shape never loses state.
    init = spec_mod.init_params(self.spec, self._seed)
    self.groups = {
        name: learner_mod.ParamGroup(self.spec.group(name), self.device,
                                     trainable=(name != 'critic_target'),
                                     dtype=_dtype)
        for name in ('model', 'actor', 'critic', 'critic_target')}
    for g in self.groups.values():
      g.load(init)
    self.learner = None
    self._plan = None
    self._train_calls = 0
    self._policies = {}
    self._policy_plans = {}   # (rows, sample, noise amount) -> captured Agent.policy launch sequence
    self._pending_load = None

  # ------------------------------------------------------------------ helpers

  def _build_learner(self, batch, length):
    assert batch % self.world == 0, (batch, self.world)  # tfagent.py:113
    self.learner = learner_mod.Learner(
        self.spec, self.ops, self.device, batch // self.world, length,
        rank=self.rank, world=self.world, comm=self.comm,
        noise_seed=self._noise_seed, dtype=self._dtype, groups=self.groups,
        ops2=self.ops2, ops_b=self.ops_b, comm_b=self.comm_b, ops_b2=self.ops_b2,
        dp_overlap=(self.comm is not None and not self._pipeline
                    and os.environ.get('DD_DP_OVERLAP', '1') != '0'))
    if self._pending_load is not None:
      self._apply_load(self._pending_load)
      self._pending_load = None

  def _ensure_params(self):
    """Parameters exist before the first train call (policy / save)."""
    if self.learner is None:
      self._build_learner(self.world * 1, 1)
      self._bootstrap = True

  def _shard(self, data):
    if self.world == 1 or isinstance(data, ShardedBatch):
      return data
    B = len(data['is_first'])
    per = B // self.world
    lo = self.rank * per
    return {k: v[lo:lo + per] for k, v in data.items()}

  # ---------------------------------------------------------------------- API

  def dataset(self, generator_fn):
    """Iterable of [B,T,...] minibatches (reference agent.py:108-121).  A
    `DeviceReplay.dataset` generator is recognised and replaced by minibatches
    gathered in HBM (no host staging); any other generator is zipped on the host.
    Data parallel with `hip.shard_dataset` (default): this rank's B / world rows only.  A
    `DeviceReplay` is reseeded by rank here; a HOST generator is the caller's: every rank must
    draw different rows (seed the sampler by rank, as the reference's per-process replays do) -
    identical generators would hand every rank the same rows (a warning is printed once)."""
    owner = getattr(generator_fn, '__self__', None)
    B = self.cfg['batch_size']
    sharded = self.world > 1 and self._shard_dataset
    if sharded:  # tfagent.py:113: the global batch must divide over the replicas
      assert B % self.world == 0, (B, self.world)
      B //= self.world
    if isinstance(owner, replay_mod.DeviceReplay):
      if sharded and not getattr(owner, '_rank_seeded', False):
        # every rank draws its own B / world rows: replays that hold the same episodes (all
        # ranks loaded one directory) must not draw the same rows with the same default seed
        owner.reseed(self.rank)
      it = owner.batches(B)
      return (ShardedBatch(b) for b in it) if sharded else it
    if sharded and not Agent._warned_host_shard:
      Agent._warned_host_shard = True
      print(f'[daydreamer_amd] rank {self.rank}/{self.world}: rank-sharded dataset over a host generator - '
            'make sure every rank samples with its own seed (DeviceReplay(seed=rank) or a rank-seeded sampler)',
            flush=True)
    batcher = Batcher(generator_fn, B, device=self.device, sharded=sharded)
    weakref.finalize(self, batcher.close)   # the dataset belongs to this agent
    return batcher

  def train(self, data, state=None):
    cls = ShardedBatch if isinstance(data, ShardedBatch) else dict
    data = cls({k: (v if isinstance(v, torch.Tensor) else np.asarray(v))
                for k, v in data.items() if not k.startswith('log_')})
    B, T = data['is_first'].shape[:2]
    if isinstance(data, ShardedBatch):
      B *= self.world   # this rank's rows of the global batch
    L = self.learner
    if L is None or getattr(self, '_bootstrap', False) or (L.Bg, L.T) != (B, T):
      self.flush()
      # controller state (AutoAdapt scales, Normalize moments, slow-critic counter, noise
      # step) lives in the learner instance: carry it over - also out of the bootstrap
      # learner, which is where a load() before the first train() put the checkpoint's
      saved = L.export_state() if L is not None else None
      self._bootstrap = False
      self.learner = None
      self._build_learner(B, T)
      if saved is not None:
        self.learner.import_state(saved)
      L = self.learner
      self._plan, self._pipe, self._seq_plan, self._train_calls = None, None, None, 0
    carry = isinstance(state, TrainState) and state.owner is L
    if self._pipe is not None and not carry:
      # reset_carry reads world-model weights and writes the carried state on this
      # stream: order it after the world-model phase still in flight
      torch.cuda.current_stream(self.device).wait_stream(self._pipe.s1)
    if self._adaptive:
      if self._touched:
        self._streak_touched, self._streak_clean = self._streak_touched + 1, 0
      else:
        self._streak_touched, self._streak_clean = 0, self._streak_clean + 1
      self._touched = False
      if self._streak_touched >= self.ADAPT:
        self._prefer_seq = True
      elif self._streak_clean >= self.ADAPT:
        self._prefer_seq = False
    use_pipe = self._pipeline and not self._prefer_seq and self._train_calls >= 1 and 'key' not in data
    # (pipelined: the minibatch goes into the step's staging buffers - see Pipeline.step)
    staged = (use_pipe and self._pipe is not None and self.device.type == 'cuda' and
              os.environ.get('DD_STAGE_INPUTS', '1') == '1')
    L.upload(self._shard(data), dst=self._pipe.stage() if staged else None)
    if not carry:
      L.reset_carry()
    if use_pipe:
      if self._pipe is None:
        self._pipe = Pipeline(L, self.device, self.comm_m,
                              tune=bool(self.cfg.get('hip', {}).get('tune_pipeline', True)))
        self._pipe.keys = tuple(self._last_metrics)   # (the first call of a learner is eager)
      self._plan = self._pipe
      metrics = self._last_metrics = self._pipe.step(staged=staged)   # this call's metrics, fetched lazily
      self._train_calls += 1
      return {}, TrainState(L), metrics
    self.flush()
    if self._use_graph and self._train_calls >= 1:
      if self._seq_plan is None:
        self._seq_plan = L.capture()
      self._plan = self._seq_plan
      self._plan.replay()
    else:
      L.train_step_device(True)
    self._train_calls += 1
    metrics = self._last_metrics = L.read_metrics()
    outs = {}
    if 'key' in data:  # prioritized replay, agent.py:89-93
      name = self.cfg['priority']
      src = {'reward_loss': L.b['loss_reward'], 'cont_loss': L.b['loss_cont'],
             'kl_loss': L.b['kl']}.get(name)
      if src is None:
        raise NotImplementedError(f'priority: {name}')
      key = self._shard(data)['key']   # this rank's rows, like the priorities below
      outs = {'key': key.cpu().numpy() if isinstance(key, torch.Tensor) else key,
              'priority': src.view(L.B, L.T).cpu().numpy().copy()}
    return outs, TrainState(L), metrics

  train_step = train  # BASELINE.json names the learner step `train_step`
  ADAPT = 4           # consecutive train calls of one kind before `hip.pipeline: auto` switches schedule

  def tune_pipeline(self, data, state=None, force=False):
    """Optional: finish (force: repeat) the pipeline's stream-pair measurement now, with train
    steps on `data` (12 pairs x 4 steps; see Pipeline), instead of inside the next train calls.
    Returns the recurrent state to continue from.  No-op with the sequential schedule."""
    box = [state]
    if not self._pipeline or 'key' in data:
      return state
    self._prefer_seq, self._streak_touched, self._touched = False, 0, False
    for _ in range(2 if self._pipe is None else 0):   # eager step + pipeline creation
      _, box[0], _ = self.train(data, box[0])
    if self._pipe is not None:
      def run():
        _, box[0], _ = self.train(data, box[0])
      self._pipe.tune(run, force=force)
    return box[0]

  def flush(self):
    """Drain the two-stream pipeline (no-op otherwise); returns the (fetched) metrics of the
    last step if one was still in flight."""
    if self._pipe is None:
      return None
    mets = self._pipe.flush()
    if mets is not None:
      self._last_metrics = mets
    return mets

  def policy(self, obs, state=None, mode='train'):
    """reference agent.py:42-65: one obs_step from the carried latent, the actor's sample
    ('train' / 'explore'; expl_behavior None = the task behaviour) or mode ('eval'), then
    tfutils.action_noise with expl_noise / eval_noise."""
    assert mode in ('train', 'eval', 'explore'), mode
    obs = {k: (v if isinstance(v, torch.Tensor) else np.asarray(v))
           for k, v in obs.items() if not k.startswith('log_')}
    n = len(obs['is_first'])
    self._ensure_params()
    self._touched = True
    self.flush()  # (pipelined mode: the behaviour phase in flight still writes the actor)
    P = self._policies.get(n)
    if P is None:
      P = learner_mod.Learner(
          self.spec, self.ops, self.device, n, 1, groups=self.groups,
          noise_seed=self._noise_seed + 1, dtype=self._dtype)
      self._policies[n] = P
    b = P.b
    data = dict(obs)
    data['is_terminal'] = obs.get('is_terminal', np.zeros(n, bool))
    data['reward'] = obs.get('reward', np.zeros(n, np.float32))
    if isinstance(state, PolicyState):
      b['carry'].copy_(state.latent)
      data['action'] = state.action
    else:
      P.reset_carry()
      data['action'] = np.zeros((n, self.act_dim), np.float32)
    P.upload({k: v[:, None] for k, v in data.items()})
    if isinstance(state, PolicyState):
      b['action'].copy_(state.action_dev)
    noise = self.cfg['eval_noise'] if mode == 'eval' else self.cfg['expl_noise']
    # The device work of a policy call is one fixed launch sequence per (batch, sample / mode,
    # noise amount): run eagerly once, then replayed from a HIP graph (~60 launches on a handful
    # of rows are pure launch latency: 0.96 -> 0.4 ms per call at the reference's TEST_CONFIG).
    sample = mode != 'eval'
    entry = self._policy_plans.setdefault((n, sample, float(noise)), dict(calls=0, plan=None))
    if entry['plan'] is not None:
      entry['plan'].replay()
      act = entry['act']
    else:
      act = entry['act'] = P.policy_device(sample=sample, noise=float(noise))
      if self._use_graph and entry['calls'] >= 1:   # (second call: lazily created buffers exist)
        plan = graphs.GraphPlan(self.device)
        P.plan = plan
        try:
          plan.capture(lambda: P.policy_device(sample=sample, noise=float(noise)))
        finally:
          P.plan = graphs.EagerPlan()
        entry['plan'] = plan
      entry['calls'] += 1
    st = PolicyState(b['carry'].clone(), None)
    st.action_dev = act.clone()
    action = act.cpu().numpy().astype(np.float32).reshape((n,) + tuple(self.act_space.shape))
    st.action = action
    return {'action': action}, st

  REPORT_SEQS, REPORT_CTX = 6, 5   # reference agent.py:269-271, behaviors.py:34-36

  def report(self, data):
    """Agent.report (reference agent.py:95-106) = WorldModel.report (agent.py:266-282) +
    Greedy.report under the `task_` prefix (behaviors.py:32-46), nothing is updated:
      * the world-model loss metrics on the batch (WorldModel.loss's metrics dict);
      * `openl_<key>` per image key: [truth | reconstruction of the first 5 steps followed by
        the open-loop prediction from the recorded actions | error] of the first 6
        sequences, tfutils.video_grid layout [T, 3H, 6W, C];
      * `task_imag_<key>`: the decoded imagined rollout of the policy from the 6 states after
        the 5 context steps, [horizon + 1, H, 6W, C].
    `data` may hold numpy arrays or device tensors (Agent.dataset yields the latter).
    The device work of a report - world-model forward, open-loop rollout, imagination, decoding,
    the sigmoid and the grid layout of the videos (dd_video_grid) - is one launch sequence without
    host decisions: the first report of a batch shape runs it eagerly (buffers and statistics
    slots come into being), every later one replays it from a HIP graph; what crosses PCIe is
    the minibatch in and the metric slabs and finished float32 grids out."""
    data = {k: (v if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in data.items() if not k.startswith('log_')}
    B, T = data['is_first'].shape[:2]
    self._touched = True
    self.flush()
    self._ensure_params()
    key = ('report', B, T)
    rep = self._policies.get(key)
    if rep is None:
      rep = self._policies[key] = self._build_report(B, T)
    R = rep['R']
    R.wmkl_scale.copy_(self.learner.wmkl_scale)
    R.upload(data)
    if rep['plan'] is not None:
      rep['plan'].replay()
    else:
      rep['run']()
      if self._use_graph and rep['calls'] >= 1:   # (second call: every lazily created buffer exists)
        plan = graphs.GraphPlan(self.device)
        learners = [R] + list(rep['imag'] or ())
        for L in learners:
          L.plan = plan
        try:
          plan.capture(rep['run'])
        finally:
          for L in learners:
            L.plan = graphs.EagerPlan()
        rep['plan'] = plan
    rep['calls'] += 1
    out = dict(R.read_metrics(wm_only=True))
    for name, grid in rep['grids'].items():
      out[name] = grid.cpu().numpy()
    return out

  def _build_report(self, B, T):
    """The learners, device buffers and launch sequence of Agent.report for a [B, T] batch."""
    R = learner_mod.Learner(
        self.spec, self.ops, self.device, B, T, groups=self.groups,
        noise_seed=self._noise_seed + 2, dtype=self._dtype)
    nseq, ctx = min(self.REPORT_SEQS, B), self.REPORT_CTX
    video = bool(self.spec.dec_convs) and T > ctx
    openl = video and R.H * R.N >= (T - ctx) * B
    H = R.H
    roll = dec = None
    if video:
      # Greedy.report: imagine with the policy from the states after the context steps
      roll = learner_mod.Learner(self.spec, self.ops, self.device, nseq, 1, groups=self.groups,
                                 noise_seed=self._noise_seed + 3, dtype=self._dtype)
      dec = learner_mod.Learner(self.spec, self.ops, self.device, H + 1, nseq, groups=self.groups,
                                noise_seed=self._noise_seed + 3, dtype=self._dtype)
    grids = {}
    hw = self.spec.image_hw if video else 0
    chans = []   # per image key: channel slice of the decoder output / the uint8 input image
    c0 = 0
    for k, shp in (self.spec.dec_cnn_keys.items() if video else ()):
      chans.append((k, c0, c0 + shp[2]))
      c0 += shp[2]
      if openl:
        grids[f'openl_{k}'] = torch.zeros(T, 3 * hw, nseq * hw, shp[2], dtype=torch.float32, device=self.device)
      grids[f'task_imag_{k}'] = torch.zeros(H + 1, hw, nseq * hw, shp[2], dtype=torch.float32, device=self.device)

    def run():
      R.reset_carry()
      R.phase_prep()
      R.phase_prep_b()  # prior-sample noise of the open-loop rollout
      R.phase_wm_fwd(True, training=False)
      if not video:
        return
      post = R.b['post'].view(B, T, R.F)
      if openl:
        z = R.openloop_device(ctx)          # [B, T, h, w, C] pre-sigmoid, batch-major
        for k, a, b in chans:
          self.ops.video_grid(z, R.b['image'], grids[f'openl_{k}'], nseq, T, a, b, T, 1)
      self.ops.copy2d(post[:nseq, ctx - 1], roll.b['traj'][0][:, :R.F])
      roll.phase_prep_b()
      roll.imagine_rollout()
      dec.decoder_fwd(roll.b['traj'].view(-1, R.TW)[:, :R.F])
      zi = dec.dec_act[-1]['z']             # [(H + 1) * nseq, h, w, C], time-major
      for k, a, b in chans:
        self.ops.video_grid(zi, None, grids[f'task_imag_{k}'], nseq, H + 1, a, b, 1, nseq)

    return dict(R=R, imag=(roll, dec) if video else None, run=run, grids=grids, plan=None, calls=0)

  # ------------------------------------------------------------- checkpointing

  def save(self):
    self._ensure_params()
    self._touched = True
    self.flush()
    L = self.learner
    out = {f'params/{k}': v for k, v in L.export_params().items()}
    for gname in ('model', 'actor', 'critic'):
      g = L.groups[gname]
      for p in g.specs:
        off = g.offset[p.name]
        out[f'opt/{gname}/m/{p.name}'] = g.m[off:off + p.size].cpu().numpy().copy().reshape(p.shape)
        out[f'opt/{gname}/v/{p.name}'] = g.v[off:off + p.size].cpu().numpy().copy().reshape(p.shape)
      st = g.opt_state.cpu().numpy()
      out[f'opt/{gname}/step'] = np.asarray(st[0], np.int64)
      out[f'opt/{gname}/grad_scale'] = np.asarray(st[3], np.float32)      # tfutils.py:113-114 (Optimizer.variables)
      out[f'opt/{gname}/good_steps'] = np.asarray(st[4], np.int64)
    out['state/wmkl_scale'] = L.wmkl_scale.cpu().numpy().copy()
    out['state/actent_scale'] = L.actent_scale.cpu().numpy().copy()
    for k, v in L.norm_state.items():
      out[f'state/norm/{k}'] = v.cpu().numpy().copy()
    out['state/slow_updates'] = np.asarray(L.slow_updates, np.int64)
    out['state/noise_step'] = L.step_ctr.cpu().numpy().copy()  # == step_ctr_b when drained
    return out

  def load(self, data):
    self.flush()
    if self.learner is None:
      self._pending_load = dict(data)
      self._ensure_params()
    else:
      self._apply_load(data)

  def _apply_load(self, data):
    L = self.learner
    params = {k[len('params/'):]: v for k, v in data.items()
              if k.startswith('params/')}
    for g in L.groups.values():
      g.load(params)
    for gname in ('model', 'actor', 'critic'):
      g = L.groups[gname]
      for p in g.specs:
        off = g.offset[p.name]
        for slot, buf in (('m', g.m), ('v', g.v)):
          arr = data.get(f'opt/{gname}/{slot}/{p.name}')
          if arr is not None:
            buf[off:off + p.size].copy_(torch.as_tensor(np.asarray(arr)).reshape(-1))
      if f'opt/{gname}/step' in data:
        g.opt_state[0] = float(data[f'opt/{gname}/step'])
      if f'opt/{gname}/grad_scale' in data:
        g.opt_state[3] = float(data[f'opt/{gname}/grad_scale'])
        g.opt_state[4] = float(data[f'opt/{gname}/good_steps'])
    if 'state/wmkl_scale' in data:
      L.wmkl_scale.copy_(torch.as_tensor(data['state/wmkl_scale']))
      L.actent_scale.copy_(torch.as_tensor(data['state/actent_scale']))
      for k in L.norm_state:
        L.norm_state[k].copy_(torch.as_tensor(data[f'state/norm/{k}']))
      L.slow_updates = int(data['state/slow_updates'])
      L.step_ctr.copy_(torch.as_tensor(data['state/noise_step']))
      L.step_ctr_b.copy_(torch.as_tensor(data['state/noise_step']))
