"""Stress of the HIP-graph lifecycle across agents (round-2 driver crash: SIGSEGV inside
graph replay of a fresh agent after earlier agents' graphs had been dropped).

  loop x N { create Agent, a few train calls (eager, capture, replay), [pipelined variant],
             [minibatches through the Batcher prefetch thread], del, gc, empty_cache }

usage: graph_stress.py [--iters N] [--modes seq,pipe,batcher] [--steps K] [--keep]
Run under `rocgdb -batch -ex run -ex bt --args python tools/graph_stress.py ...` for a native
backtrace.
"""
import argparse
import faulthandler
import gc
import itertools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

import helpers
from daydreamer_amd import agent as agent_mod, synthetic


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=30)
  ap.add_argument('--steps', type=int, default=4)
  ap.add_argument('--modes', default='seq,pipe,batcher')
  ap.add_argument('--keep', action='store_true', help='keep every agent alive (no destruction)')
  ap.add_argument('--no-fh', action='store_true')
  args = ap.parse_args()
  if not args.no_fh:
    faulthandler.enable()
  modes = args.modes.split(',')
  obs, act = synthetic.make_spaces(64, 5, 3)
  kept = []
  t0 = time.time()
  for it in range(args.iters):
    mode = modes[it % len(modes)]
    B, T, H = 4 + 2 * (it % 2), 6 + 2 * (it % 3), 3 + it % 2
    cfg = helpers.make_config(('a1_vision', 'debug'), batch_size=B, replay_chunk=T, imag_horizon=H)
    cfg = cfg.update({'hip.pipeline': mode == 'pipe'})
    ag = agent_mod.Agent(obs, act, None, cfg)
    state = None
    if mode == 'batcher':
      def gen():
        for s in itertools.count():
          ep = synthetic.make_batch(obs, act, 1, T, seed=s % 5, smooth_images=True)
          yield {k: v[0] for k, v in ep.items()}
      ds = iter(ag.dataset(gen))
      for i in range(args.steps):
        _, state, mets = ag.train(next(ds), state)
    else:
      data = synthetic.make_batch(obs, act, B, T, seed=it, smooth_images=True)
      for i in range(args.steps):
        _, state, mets = ag.train(data, state)
      ag.flush()
    assert helpers.metrics_finite(mets), (it, mode)
    print(f'iter {it} {mode} B{B} T{T} H{H} ok  model_loss {float(mets["model_loss"]):.3f} '
          f'{time.time() - t0:.1f}s', flush=True)
    if args.keep:
      kept.append(ag)
    del ag, state
    if mode == 'batcher':
      del ds
    gc.collect()
    if it % 3 == 2:
      torch.cuda.empty_cache()
  torch.cuda.synchronize()
  print('STRESS_OK', args.iters, flush=True)


if __name__ == '__main__':
  main()
