"""What the reference's training script does to an `embodied.Agent`, reproduced on the HIP path.

`tests/test_boundary_reference.py` hands our Agent to the reference's own `embodied.run.train`,
but it needs /root/reference (absent on the GPU box) and therefore runs the CPU restatement of the
kernels.  This file restates - in our own words, nothing imported from the reference - exactly the
things that loop does to the agent object (reference run/train.py:18-99, core/timer.py:28-31,
core/checkpoint.py:41-69, core/driver.py:38-77) and runs them on `HipOps`:

  * `Timer.wrap`: `setattr(agent, name, decorated(getattr(agent, name)))` for policy / train /
    report / save - every later call goes through a re-bound attribute of the INSTANCE;
  * a driver that calls `agent.policy(obs, state, mode=...)` with batched observations of one env,
    carries the returned state, resets nothing itself (is_first does), and feeds every transition
    to `replay.add`;
  * `dataset = iter(agent.dataset(replay.dataset))`, a pretrain step, then `train` every few env
    steps interleaved with `policy` calls, `report(batch)` at the log interval;
  * `Checkpoint.load_or_save()` before the loop and `save()` inside it: a pickle of
    `{name: obj.save()}`, later `obj.load(value)` - here into a FRESH agent before its first
    `train` call, which must then continue bit-identically.
"""

import contextlib
import pickle
import time

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


class MiniTimer:
  """The re-binding the reference's Timer.wrap performs (core/timer.py:28-31)."""

  def __init__(self):
    self.durations = {}

  @contextlib.contextmanager
  def scope(self, name):
    start = time.time()
    yield
    self.durations.setdefault(name, []).append(time.time() - start)

  def wrap(self, name, obj, methods):
    for method in methods:
      decorator = self.scope(f'{name}.{method}')   # a ContextDecorator, as in the reference
      setattr(obj, method, decorator(getattr(obj, method)))


class DummyEnv:
  """64x64 camera + vector observations, continuous actions; episodes of `length` steps."""

  def __init__(self, length, adim, seed=0):
    self.length, self.adim = length, adim
    self.rng = np.random.RandomState(seed)
    self.t = 0

  def step(self, action):
    first = self.t == 0
    self.t += 1
    last = self.t >= self.length
    obs = dict(
        image=self.rng.randint(0, 256, (64, 64, 3)).astype(np.uint8),
        vector=self.rng.randn(5).astype(np.float32),
        reward=np.float32(0.0 if first else float(np.tanh(action).sum())),
        is_first=first, is_last=last, is_terminal=last and bool(self.rng.rand() < 0.3))
    if last:
      self.t = 0
    return obs


class MiniCheckpoint:
  """embodied.Checkpoint (core/checkpoint.py): pickle of {name: obj.save()} + obj.load(value)."""

  def __init__(self, path):
    self.path, self.values = path, {}

  def save(self):
    data = {k: v.save() for k, v in self.values.items()}
    data['_timestamp'] = time.time()
    self.path.write_bytes(pickle.dumps(data))

  def load(self):
    data = pickle.loads(self.path.read_bytes())
    for k, v in data.items():
      if not k.startswith('_'):
        self.values[k].load(v)

  def load_or_save(self):
    self.load() if self.path.exists() else self.save()


def _make(tmp_path, step):
  from daydreamer_amd import agent as agent_mod, synthetic
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=4, replay_chunk=8, imag_horizon=3)
  cfg = cfg.update({'logdir': str(tmp_path), 'expl_noise': 0.2})
  obs_space, act_space = synthetic.make_spaces(64, 5, 3)
  return agent_mod.Agent(obs_space, act_space, step, cfg), cfg


def test_run_loop_on_hip_kernels(hip, tmp_path):
  from daydreamer_amd import replay as replay_mod
  step = [0]
  agent, cfg = _make(tmp_path, step)
  assert type(agent.ops).__name__ == 'HipOps' and agent.device.type == 'cuda'
  env = DummyEnv(length=20, adim=3)
  replay = replay_mod.DeviceReplay(chunk=8, capacity=2000, directory=tmp_path / 'episodes')

  timer = MiniTimer()
  timer.wrap('agent', agent, ['policy', 'train', 'report', 'save'])   # run/train.py:18-19
  assert 'train' in vars(agent)            # the instance attribute now shadows the class method

  # ---- prefill with a random policy (run/train.py:56-60)
  rng = np.random.RandomState(1)
  action = np.zeros(3, np.float32)
  for _ in range(60):
    obs = env.step(action)
    action = rng.uniform(-1, 1, 3).astype(np.float32)
    replay.add({**obs, 'action': action})
    step[0] += 1
  assert len(replay) >= 40

  dataset = iter(agent.dataset(replay.dataset))                       # :62
  state = [None]
  _, state[0], mets = agent.train(next(dataset), state[0])            # pretrain, :64-66
  assert helpers.metrics_finite(mets)

  ckpt = MiniCheckpoint(tmp_path / 'checkpoint.pkl')                  # :89-93
  ckpt.values['agent'] = agent
  ckpt.load_or_save()
  assert (tmp_path / 'checkpoint.pkl').exists()

  # ---- the loop: policy every env step, train every 5, report every 20, checkpoint every 30
  pstate, batch, n_train, reports = None, None, 0, []
  action = np.zeros(3, np.float32)
  for it in range(60):
    obs = env.step(action)
    batched = {k: np.asarray(v)[None] for k, v in obs.items()}        # Driver: one env, batch of 1
    mode = 'explore' if it < 20 else 'train'
    out, pstate = agent.policy(batched, pstate, mode=mode)
    action = np.asarray(out['action'][0], np.float32)
    assert action.shape == (3,) and np.isfinite(action).all()
    replay.add({**obs, 'action': action})
    step[0] += 1
    if it % 5 == 0:
      batch = next(dataset)
      outs, state[0], mets = agent.train(batch, state[0])
      n_train += 1
      assert helpers.metrics_finite(mets), it
    if it % 20 == 19:
      rep = agent.report(batch)
      assert any(k.startswith('openl_') for k in rep) and len(rep) > 5, sorted(rep)[:8]
      reports.append(rep)
    if it % 30 == 29:
      ckpt.save()
  assert n_train == 12 and len(reports) == 3
  assert float(agent.learner.groups['model'].opt_state[0]) == 1 + n_train
  for name in ('agent.policy', 'agent.train', 'agent.report', 'agent.save'):
    assert timer.durations.get(name), name

  # ---- resume: a fresh agent loads the pickle BEFORE its first train call and continues
  #      bit-identically (parameters, Adam moments, controller state, RNG step counters)
  ckpt.save()
  agent2, _ = _make(tmp_path, step)
  ckpt2 = MiniCheckpoint(tmp_path / 'checkpoint.pkl')
  ckpt2.values['agent'] = agent2
  ckpt2.load_or_save()
  a, b = agent.save(), agent2.save()
  assert a.keys() == b.keys()
  for k in a:
    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
  host = {k: (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in batch.items()}
  _, _, m1 = agent.train(dict(host), None)
  _, _, m2 = agent2.train(dict(host), None)
  for k in ('model_loss', 'actor_loss', 'extr_critic_loss', 'model_grad_norm'):
    assert float(m1[k]) == float(m2[k]), (k, float(m1[k]), float(m2[k]))
  a, b = agent.save(), agent2.save()
  for k in a:
    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
  # the replay writes the reference's DiskStore episode files
  out = replay.save()
  assert out and list((tmp_path / 'episodes').glob('*.npz'))
