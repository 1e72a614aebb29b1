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
