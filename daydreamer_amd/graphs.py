"""HIP-graph execution plan for the learner step, and the process-level launch runtime.

The train step is thousands of small dependent kernel launches (a T-step and an H-step
scan); replaying them from captured HIP graphs removes the host launch cost.  The step is
cut into graph segments at the few points that need host or collective work (data-parallel
all-reduces, the slow-critic copy whose schedule is a host counter); `cut(fn)` marks such a
point.  Role in the reference: the tf.function concrete-function cache of TFAgent.train
(tfagent.py:56-70).

Lifetime rules (the round-2 driver run died with SIGSEGV inside a graph replay):

  * streams: one process-owned HIP stream per (device, role) from the library
    (`dd_stream_create`), wrapped for torch with ExternalStream.  `torch.cuda.Stream()` hands
    out handles of a 32-entry round-robin pool shared with everything else in the process
    (other agents, the prefetch thread, torch.distributed): a capturing stream could BE the
    stream another thread was issuing on, which invalidates or corrupts the capture.
  * graph executables are owned by the library (`dd_graph_capture_end`) and registered here.
    Dropping a plan never destroys anything by itself (an executable must not be freed under
    work that is still queued): the plan's executables are RETIRED, and retired executables are
    destroyed (`dd_graph_destroy`) only at the next capture of the same device - under API_LOCK,
    right after the device-wide synchronize every capture starts with, when nothing of the
    process can be in flight or be launching - and only once at least RECLAIM_THRESHOLD of them
    have piled up (a process that builds one agent never destroys anything; an agent-in-a-loop
    sweep stays bounded).  DD_GRAPH_RECLAIM=<n> sets the threshold, 0 disables reclaiming.
  * every capture and graph launch happens under API_LOCK; the prefetch thread
    (agent.Batcher) takes the same lock around its own runtime calls, so no other host thread
    of this package is inside the HIP runtime while a capture or a launch is.
"""

import ctypes
import os
import threading

import torch

from . import hipops

API_LOCK = threading.RLock()

# Debug check (tests switch it on): count the caching allocator's allocations over a capture.
# The capture is not known to the allocator, so a tensor allocated inside a captured segment
# would be recycled while the graph still writes to it; the learner's captured code allocates
# nothing.  The counter is process-wide: off by default, because another thread of the caller
# (a data loader) may legitimately allocate at the same time.
CHECK_CAPTURE_ALLOCS = False

_STREAMS = {}   # (device index, role) -> torch.cuda.ExternalStream
_EXECS = set()  # every live graph executable of the process: (device index, handle)
_RETIRED = []   # executables of dropped plans, destroyed by the next capture on their device
RECLAIM_THRESHOLD = int(os.environ.get('DD_GRAPH_RECLAIM', 32))
_DESTROYED = [0]
_DBG_SYNC = int(os.environ.get('DD_GRAPH_DBG_SYNC', 0))   # debugging: 1 drain after every graph segment, 2 after every cut function


def stream(device, role):
  """The process-owned stream of `role` on `device` (created on first use, never destroyed)."""
  device = torch.device(device)
  idx = device.index if device.index is not None else torch.cuda.current_device()
  key = (idx, role)
  with API_LOCK:
    s = _STREAMS.get(key)
    if s is None:
      lib = hipops.load_library()
      out = ctypes.c_void_p()
      with torch.cuda.device(idx):
        rc = lib.dd_stream_create(ctypes.byref(out))
      if rc != 0:
        raise RuntimeError(f'dd_stream_create failed ({rc}): {lib.dd_last_error().decode()}')
      s = torch.cuda.ExternalStream(out.value, device=torch.device('cuda', idx))
      _STREAMS[key] = s
    return s


def n_live_graphs():
  """Executables that are instantiated right now (owned by live plans + retired, not yet destroyed)."""
  return len(_EXECS)


def n_destroyed_graphs():
  return _DESTROYED[0]


def _reclaim(lib, idx, force=False):
  """Destroy the retired executables of device `idx`.  Caller holds API_LOCK and has just
  synchronized the device."""
  if not RECLAIM_THRESHOLD and not force:
    return
  mine = [e for e in _RETIRED if e[0] == idx]
  if not force and len(mine) < RECLAIM_THRESHOLD:
    return
  for e in mine:
    _RETIRED.remove(e)
    _EXECS.discard(e)
    with torch.cuda.device(idx):
      rc = lib.dd_graph_destroy(ctypes.c_void_p(e[1]))
    if rc != 0:
      raise RuntimeError(f'dd_graph_destroy failed ({rc}): {lib.dd_last_error().decode()}')
    _DESTROYED[0] += 1


def reclaim(device=None):
  """Synchronize and destroy every retired executable now (tests, long sweeps)."""
  lib = hipops.load_library()
  with API_LOCK:
    for idx in sorted({e[0] for e in _RETIRED}):
      if device is not None and torch.device(device).index not in (None, idx):
        continue
      torch.cuda.synchronize(idx)
      _reclaim(lib, idx, force=True)


class EagerPlan:
  """No capture: run everything as issued (used on CPU tests and as warm-up)."""

  capturing = False

  def cut(self, fn):
    fn()

  def conditional(self, pred, fn):
    if pred():
      fn()


class GraphPlan:

  def __init__(self, device, role='plan'):
    self.device = torch.device(device)
    self.items = []
    self.capturing = False
    self.lib = hipops.load_library()
    # all sequential plans of the process share one stream: they are issued by one thread, in
    # program order
    self.stream = stream(self.device, role)
    self._idx = self.stream.device.index
    self._owned = []

  def release(self):
    """Retire this plan's executables (destroyed by a later capture, see the module docstring);
    the plan must not be replayed afterwards."""
    owned, self._owned = self._owned, []
    self.items = []
    _RETIRED.extend(owned)   # (list.extend is atomic: __del__ may run on any thread)

  def __del__(self):
    try:
      self.release()
    except Exception:  # noqa: BLE001 - interpreter shutdown
      pass

  def _check(self, rc, what):
    if rc != 0:
      raise RuntimeError(f'{what} failed ({rc}): {self.lib.dd_last_error().decode()}')

  def _begin(self):
    self._check(self.lib.dd_graph_capture_begin(self.stream.cuda_stream), 'dd_graph_capture_begin')

  def _end(self):
    exe, nodes = ctypes.c_void_p(), ctypes.c_int()
    self._check(self.lib.dd_graph_capture_end(self.stream.cuda_stream, ctypes.byref(exe),
                                              ctypes.byref(nodes)), 'dd_graph_capture_end')
    # (a segment between two adjacent cut points holds no kernels: exe is NULL, launch a no-op)
    if exe.value:
      _EXECS.add((self._idx, exe.value))
      self._owned.append((self._idx, exe.value))
    self.items.append(('graph', exe.value))

  def capture(self, fn):
    """Record fn() (which issues kernels on the current stream and calls
    cut(...) at host/collective points) without executing it."""
    with API_LOCK:
      torch.cuda.synchronize(self.device)
      _reclaim(self.lib, self._idx)
      check = CHECK_CAPTURE_ALLOCS
      allocs = torch.cuda.memory_stats(self.device).get('allocation.all.allocated', 0) if check else 0
      self.stream.wait_stream(torch.cuda.current_stream(self.device))
      with torch.cuda.stream(self.stream):
        self.capturing = True
        self._begin()
        try:
          fn()
        except BaseException:
          # close the capture without hiding what went wrong (a failed _end inside cut() has already
          # ended it: a second end would raise "illegal state" over the real message)
          self.capturing = False
          try:
            self._end()
          except Exception:  # noqa: BLE001
            pass
          raise
        self.capturing = False
        self._end()
      torch.cuda.current_stream(self.device).wait_stream(self.stream)
      if check:
        after = torch.cuda.memory_stats(self.device).get('allocation.all.allocated', 0)
        if after != allocs:
          raise RuntimeError(f'{after - allocs} device allocations inside a captured segment')

  def cut(self, fn):
    if not self.capturing:
      fn()
      return
    self._end()
    self.items.append(('eager', fn))
    self._begin()

  def conditional(self, pred, fn):
    """A segment whose execution is decided on the host at replay time (pred() is
    evaluated after the preceding cut functions ran): fn() is captured into its own
    graph; it must not contain cut points."""
    if not self.capturing:
      if pred():
        fn()
      return
    self._end()
    self._begin()
    n = len(self.items)
    fn()
    assert len(self.items) == n, 'cut point inside a conditional segment'
    self._end()
    kind, g = self.items.pop()
    self.items.append(('cond', (pred, g)))
    self._begin()

  def _launch(self, exe, stream):
    if exe:
      self._check(self.lib.dd_graph_launch(exe, stream.cuda_stream), 'dd_graph_launch')

  def _run(self, stream, start=0, stop=None):
    for kind, item in self.items[start:stop]:
      if kind == 'graph':
        self._launch(item, stream)
        if _DBG_SYNC & 1:
          stream.synchronize()
      elif kind == 'cond':
        if item[0]():
          self._launch(item[1], stream)
      else:
        item()
        if _DBG_SYNC & 2:
          stream.synchronize()

  def replay(self):
    with API_LOCK:
      cur = torch.cuda.current_stream(self.device)
      self.stream.wait_stream(cur)
      with torch.cuda.stream(self.stream):
        self._run(self.stream)
      cur.wait_stream(self.stream)

  def replay_on(self, stream, start=0, stop=None):
    """Enqueue the plan (or its items [start:stop]) on `stream` and return without joining
    any other stream (the caller orders streams with events: agent.Agent's two-stream
    pipeline)."""
    with API_LOCK:
      with torch.cuda.stream(stream):
        self._run(stream, start, stop)

  @property
  def n_graphs(self):
    return sum(1 for k, _ in self.items if k in ('graph', 'cond'))
