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


def test_load_before_first_train_keeps_controller_state():
  agent_cases.load_before_train_keeps_controller_state(ref_ops.RefOps('cpu'))
