"""Agent API host logic on CPU (kernel restatements, float64) against the oracle:
`Agent.policy` (all modes, action noise, continuous and one-hot), `Agent.report` (loss
metrics, open-loop grid, Greedy's imagined rollout), checkpoint load before the first train.
The same bodies run on the MI355X in tests/test_agent_gpu.py."""

import pytest

from oracle import ref_ops
import agent_cases

TOL = dict(sample=1e-9, action=2e-7, latent=1e-9, video=1e-7, metric=1e-6)  # (float32 outputs)


@pytest.mark.parametrize('discrete', [False, True])
@pytest.mark.parametrize('noise', [0.0, 0.3])
def test_policy_matches_oracle(discrete, noise):
  agent_cases.policy_parity(ref_ops.RefOps('cpu'), discrete, TOL, noise)


@pytest.mark.parametrize('discrete', [False, True])
def test_report_matches_oracle(discrete):
  adopted, draws = agent_cases.report_parity(ref_ops.RefOps('cpu'), discrete, TOL)
  assert adopted == 0 and draws > 0


def test_policy_and_report_match_oracle_resnet():
  """The same two bodies with `cnn: resnet` (residual encoder in policy, decoder in report)."""
  agent_cases.policy_parity(ref_ops.RefOps('cpu'), False, TOL, 0.0, cnn='resnet')
  adopted, draws = agent_cases.report_parity(ref_ops.RefOps('cpu'), False, TOL, cnn='resnet')
  assert adopted == 0 and draws > 0


def test_load_before_first_train_keeps_controller_state():
  agent_cases.load_before_train_keeps_controller_state(ref_ops.RefOps('cpu'))


def test_train_with_prioritized_replay_round_trip():
  """Prioritised minibatches (embodied.replay.Prioritized contract, run/learning.py:55-58):
  the batch carries `key` / `prob`, Agent.train returns outs = {key, priority} with the
  configured per-step loss map (agent.py:89-93), replay.prioritize consumes them."""
  import numpy as np
  import helpers
  from daydreamer_amd import agent as agent_mod, replay as replay_mod, synthetic
  import torch
  cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=3, replay_chunk=6, imag_horizon=2)
  cfg = cfg.update({'priority': 'reward_loss'})
  obs, act = synthetic.make_spaces(64, 5, 3)
  rep = replay_mod.DevicePrioritized(chunk=6, capacity=500, device='cpu', ops=ref_ops.RefOps('cpu'))
  for e in range(4):
    ep = synthetic.make_batch(obs, act, 1, 20, seed=e, smooth_images=True)
    rep.add_traj({**{k: v[0] for k, v in ep.items()}, 'is_last': np.arange(20) == 19})
  ag = agent_mod.Agent(obs, act, None, cfg, _ops=ref_ops.RefOps('cpu'), _device='cpu',
                       _dtype=torch.float64)
  ds = ag.dataset(rep.dataset)
  state = None
  for _ in range(3):
    batch = next(ds)
    assert 'key' in batch and 'prob' in batch
    outs, state, mets = ag.train(batch, state)
    assert outs['key'].shape == (3, 6, 3) and outs['priority'].shape == (3, 6)
    assert (outs['priority'] >= 0).all()
    rep.prioritize(outs['key'], outs['priority'])
  assert rep.prios.update_max > 0
