"""Drop-in boundary: the reference's own run loop (embodied.run.train), Driver,
FixedLength replay, Checkpoint and Timer (which re-binds agent methods) drive
our Agent on the dummy continuous env.  Needs /root/reference (absent on the GPU
box -> skipped); kernels are the CPU restatements (host logic test)."""

import pathlib
import sys
import types

import numpy as np
import pytest

REF = pathlib.Path('/root/reference')
pytestmark = pytest.mark.skipif(not REF.exists(), reason='reference checkout not present')


def test_reference_run_loop_drives_agent(tmp_path):
  sys.modules.setdefault('gym', types.ModuleType('gym'))
  sys.path.insert(0, str(REF))
  import embodied
  from daydreamer_amd import agent as agent_mod
  from oracle import ref_ops
  cfgs = agent_mod.Agent.configs
  config = embodied.Config(cfgs['defaults'])
  config = config.update(cfgs['a1']).update(cfgs['debug'])
  config = config.update({
      'logdir': str(tmp_path), 'batch_size': 2, 'replay_chunk': 6, 'imag_horizon': 3,
      'encoder.mlp_keys': 'vector', 'encoder.cnn_keys': 'image',
      'decoder.mlp_keys': 'vector', 'decoder.cnn_keys': 'image',
      'train.steps': 60, 'train.train_fill': 30, 'train.train_every': 10,
      'train.log_every': 20, 'train.eval_every': 40, 'train.pretrain': 1})
  env = embodied.envs.load_env('dummy_continuous', mode='train', logdir=str(tmp_path),
                               **config.env.update({'amount': 1, 'parallel': 'none', 'length': 12}))
  step = embodied.Counter()
  logger = embodied.Logger(step, [])
  agent = agent_mod.Agent(env.obs_space, env.act_space, step, config,
                          _ops=ref_ops.RefOps('cpu'), _device='cpu')
  store = embodied.replay.RAMStore(1000)
  replay = embodied.replay.FixedLength(store, chunk=config.replay_chunk)
  args = embodied.Config(logdir=config.logdir, **config.train)  # reference train.py:35
  embodied.run.train(agent, env, replay, logger, args)
  assert int(step) >= 60
  assert float(agent.learner.groups['model'].opt_state[0]) >= 3
  assert (tmp_path / 'checkpoint.pkl').exists()
  # checkpoint round trip through the reference's Checkpoint (pickle of save())
  saved = agent.save()
  agent2 = agent_mod.Agent(env.obs_space, env.act_space, step, config,
                           _ops=ref_ops.RefOps('cpu'), _device='cpu')
  agent2.load(saved)
  a, b = agent.save(), agent2.save()
  assert a.keys() == b.keys()
  for k in a:
    assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_lazy_metrics_through_the_reference_logger():
  """What the pipelined agent's train() returns (a LazyMetrics of LazyScalars) through the
  reference's own consumers: run/train.py:77-85 collects the values per call and takes
  np.nanmean at the log interval, embodied.Logger.add does np.array(value) and accepts ranks
  0 / 2 / 3 / 4 (core/logger.py:25-33), the JSON-lines output formats floats."""
  import collections
  sys.modules.setdefault('gym', types.ModuleType('gym'))
  sys.path.insert(0, str(REF))
  import embodied
  from daydreamer_amd.agent import LazyMetrics
  fetched = []
  def lazy(i):
    vals = {'model_loss': np.float32(10.0 - i), 'actor_loss': np.float32(float('nan') if i == 1 else i)}
    return LazyMetrics(tuple(vals), lambda: (fetched.append(i), vals)[1])
  step = embodied.Counter()
  seen = []
  logger = embodied.Logger(step, [seen.extend])
  metrics = collections.defaultdict(list)
  for i in range(3):                                   # run/train.py:77-78
    mets = lazy(i)
    [metrics[key].append(value) for key, value in mets.items()]
    step.increment()
  assert not fetched                                   # collecting never waited for the device
  for name, values in metrics.items():                 # run/train.py:83-85
    logger.scalar('train/' + name, np.nanmean(values, dtype=np.float64))
  logger.add(lazy(7), prefix='last')                   # a LazyMetrics handed to the logger as it is
  logger.write()
  got = {name: float(value) for _, name, value in seen}
  assert got['train/model_loss'] == pytest.approx(9.0) and got['train/actor_loss'] == pytest.approx(1.0)
  assert got['last/model_loss'] == pytest.approx(3.0) and got['last/actor_loss'] == pytest.approx(7.0)
  assert sorted(fetched) == [0, 1, 2, 7]
  assert all(np.asarray(value).shape == () for _, _, value in seen)
