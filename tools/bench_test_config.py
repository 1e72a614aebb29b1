#!/usr/bin/env python
"""Step times at the reference's own TEST_CONFIG (embodied/agents/dreamerv2plus/tests.py:26-39:
`defaults` with batch 8 x chunk 8, `.*\\.layers` 2, `.*\\.units` 128, `.*\\.cnn_depth` 16, `.*\\.wd$` 0,
dummy_discrete env = 64x64x3 image + 7-vector + scalar step / reward / is_terminal inputs, one-hot
5-way action, horizon 15) - the only configuration for which the reference states timings, as the
upper bounds of its own tests (tests.py:70-71, 88-89, 105-106: train <= 1.3 x 0.02 s, policy <=
1.3 x 0.007 s, report <= 1.3 x 0.01 s, hardware unstated).  Context figures, not BASELINE.json's metric.

  python tools/bench_test_config.py [--cpu]      (--cpu: build and step once on the CPU kernels)
"""
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

from daydreamer_amd import agent as agent_mod, config as config_mod, synthetic  # noqa: E402

TEST_CONFIG = {'replay_chunk': 8, 'batch_size': 8, r'.*\.layers': 2, r'.*\.units': 128,
               r'.*\.cnn_depth': 16, r'.*\.wd$': 0.0}


def spaces():
  S = synthetic.Space
  obs = {'image': S(np.uint8, (64, 64, 3)), 'vector': S(np.float32, (7,)), 'step': S(np.int32, ()),
         'reward': S(np.float32, ()), 'is_first': S(bool, ()), 'is_last': S(bool, ()), 'is_terminal': S(bool, ())}
  act = S(np.float32, (5,), 0, 1)
  act.discrete = True
  return obs, {'action': act, 'reset': S(bool, ())}


def main():
  cpu = '--cpu' in sys.argv
  cfgs = config_mod.load_configs()
  cfg = config_mod.Config(cfgs['defaults']).update(TEST_CONFIG)
  obs, act = spaces()
  kw = {}
  if cpu:
    from oracle import ref_ops
    kw = dict(_ops=ref_ops.RefOps('cpu'), _device='cpu')
  ag = agent_mod.Agent(obs, act, None, cfg, **kw)
  B, T = 8, 8
  batch = synthetic.make_batch(obs, act, B, T, seed=0, terminals=0.05, smooth_images=True)
  batch['step'] = np.tile(np.arange(T, dtype=np.int32), (B, 1))
  sync = (lambda: None) if cpu else torch.cuda.synchronize

  def timed(fn, n, warm):
    for _ in range(warm):
      fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
      fn()
    sync()
    return 1e3 * (time.perf_counter() - t0) / n

  box = [None]
  def train():
    _, box[0], mets = ag.train(batch, box[0])
    float(mets['model_loss'])
  n, warm = (1, 1) if cpu else (50, 5)
  if not cpu:   # the pipelined learner's stream-pair trial steps are start-up work: finish them untimed
    box[0] = ag.tune_pipeline(batch, box[0])
  t_train = timed(train, n, warm)
  mets_box = []
  def train_lazy():
    out = ag.train(batch, box[0])
    box[0] = out[1]
    mets_box.append(out[2])
  t_lazy = None
  if not cpu:
    t_lazy = timed(lambda: (train_lazy(), mets_box.__delitem__(slice(0, -2))), n, warm)
    ag.flush()
  o = {k: v[:, 0] for k, v in batch.items() if k not in ('action', 'reset')}
  pst = [None]
  def policy():
    out, pst[0] = ag.policy(o, pst[0], 'train')
    np.asarray(out['action'])
  t_policy = timed(policy, n, warm)
  def report():
    rep = ag.report(batch)
    float(rep['model_loss_mean'])
  t_report = timed(report, max(1, n // 5), 1)
  H = cfg.imag_horizon
  print(f'TEST_CONFIG (batch {B} x chunk {T}, horizon {H}, one-hot 5-way action, units 128, cnn_depth 16, '
        f'deter 1024, stoch 32x32){" on the CPU kernels" if cpu else " on the MI355X"}:')
  print(f'  agent.train   {t_train:8.3f} ms  = {B * T * H / t_train * 1e3:9.0f} imagined env-steps/s   '
        f'(reference test bound <= 26 ms, >= 36.9 k steps/s; tests.py:70-71; the metrics of every call looked at '
        f'before the next call)')
  if t_lazy is not None:
    print(f'  agent.train   {t_lazy:8.3f} ms  = {B * T * H / t_lazy * 1e3:9.0f} imagined env-steps/s   '
          f'(metrics collected per call, looked at later: run/train.py:77-85)')
  print(f'  agent.policy  {t_policy:8.3f} ms  (batch {B}; reference test bound <= 9.1 ms; tests.py:88-89)')
  print(f'  agent.report  {t_report:8.3f} ms  (reference test bound <= 13 ms; tests.py:105-106)')


if __name__ == '__main__':
  main()
