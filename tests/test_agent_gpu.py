"""Agent API on the MI355X (HIP kernels, float32) against the float64 oracle: `Agent.policy`
in every mode with and without action noise (continuous and one-hot actors), `Agent.report`
from host and from device minibatches (what `Agent.dataset` yields), checkpoint load before
the first train call.  Bodies: tests/agent_cases.py (shared with the CPU host-logic tests)."""

import pytest

import agent_cases

pytestmark = pytest.mark.gpu

# float32 kernels vs float64 oracle through the encoder + one obs_step + actor / decoder
TOL = dict(sample=1e-6, action=2e-4, latent=2e-4, video=2e-3, metric=1e-3)


@pytest.mark.parametrize('discrete', [False, True])
@pytest.mark.parametrize('noise', [0.0, 0.3])
def test_policy_matches_oracle(hip, discrete, noise):
  agent_cases.policy_parity(None, discrete, TOL, noise)


@pytest.mark.parametrize('discrete,device_batch', [(False, False), (False, True), (True, True)])
def test_report_matches_oracle(hip, discrete, device_batch):
  adopted, draws = agent_cases.report_parity(None, discrete, TOL, device_batch=device_batch, calls=3)
  print(f'report: {adopted} of {draws} draws adopted from the device')
  assert adopted <= max(1, draws // 5000)


def test_policy_and_report_match_oracle_resnet(hip):
  """`cnn: resnet`: the residual encoder inside Agent.policy, the residual decoder inside
  Agent.report (open-loop and imagined grids), device minibatch in."""
  agent_cases.policy_parity(None, False, TOL, 0.3, cnn='resnet')
  adopted, draws = agent_cases.report_parity(None, False, TOL, device_batch=True, cnn='resnet', calls=3)
  assert adopted <= max(1, draws // 5000)


def test_load_before_first_train_keeps_controller_state(hip):
  agent_cases.load_before_train_keeps_controller_state(None)
