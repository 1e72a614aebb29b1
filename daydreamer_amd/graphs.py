"""HIP-graph execution plan for the learner step.

The train step is thousands of small dependent kernel launches (a T-step and an
H-step scan); replaying them from captured HIP graphs removes the host launch
cost.  The step is cut into graph segments at the few points that need host
or collective work (data-parallel all-reduces, the slow-critic copy whose
schedule is a host counter); `cut(fn)` marks such a point.  PyTorch is used only
for its stream / graph handles.
"""

import warnings

import torch


class EagerPlan:
  """No capture: run everything as issued (used on CPU tests and as warm-up)."""

  capturing = False

  def cut(self, fn):
    fn()

  def conditional(self, pred, fn):
    if pred():
      fn()


class GraphPlan:

  def __init__(self, device):
    self.device = torch.device(device)
    self.items = []
    self.cur = None
    self.capturing = False
    self.stream = torch.cuda.Stream(self.device)

  def _begin(self):
    self.cur = torch.cuda.CUDAGraph()
    # thread_local: the RCCL watchdog thread of torch.distributed may touch the
    # device while this thread captures
    self.cur.capture_begin(capture_error_mode='thread_local')

  def _end(self):
    with warnings.catch_warnings():
      # a segment between two adjacent cut points holds no kernels: fine
      warnings.filterwarnings('ignore', message='The CUDA Graph is empty')
      self.cur.capture_end()
    self.items.append(('graph', self.cur))
    self.cur = None

  def capture(self, fn):
    """Record fn() (which issues kernels on the current stream and calls
    cut(...) at host/collective points) without executing it."""
    torch.cuda.synchronize(self.device)
    self.stream.wait_stream(torch.cuda.current_stream(self.device))
    with torch.cuda.stream(self.stream):
      self.capturing = True
      self._begin()
      try:
        fn()
      finally:
        self._end()
        self.capturing = False
    torch.cuda.current_stream(self.device).wait_stream(self.stream)

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

  def _run(self, start=0, stop=None):
    for kind, item in self.items[start:stop]:
      if kind == 'graph':
        item.replay()
      elif kind == 'cond':
        if item[0]():
          item[1].replay()
      else:
        item()

  def replay(self):
    cur = torch.cuda.current_stream(self.device)
    self.stream.wait_stream(cur)
    with torch.cuda.stream(self.stream):
      self._run()
    cur.wait_stream(self.stream)

  def replay_on(self, stream, start=0, stop=None):
    """Enqueue the plan (or its items [start:stop]) on `stream` and return without joining
    any other stream (the caller orders streams with events: agent.Agent's two-stream
    pipeline)."""
    with torch.cuda.stream(stream):
      self._run(start, stop)

  @property
  def n_graphs(self):
    return sum(1 for k, _ in self.items if k in ('graph', 'cond'))
