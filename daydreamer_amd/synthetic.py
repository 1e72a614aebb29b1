"""Synthetic replay minibatches and observation spaces for tests and bench.

Wire format follows what `embodied.Replay.dataset()` + batching delivers to
`Agent.train` (reference replay/fixed_length.py:64-81, core/convert.py:4-23):
`{image: u8[B,T,H,W,3], vector: f32[B,T,d], reward: f32[B,T],
is_first/is_last/is_terminal: bool[B,T], action: f32[B,T,A], reset: bool[B,T]}`
with `is_first[:, 0] = True`.
"""

import numpy as np


class Space:
  """Minimal stand-in for embodied.Space (reference core/space.py:4-18):
  only dtype / shape / discrete are consumed by the learner."""

  def __init__(self, dtype, shape=(), low=None, high=None):
    self.dtype = np.dtype(dtype)
    self.shape = tuple(shape) if hasattr(shape, '__len__') else (shape,)
    self.low, self.high = low, high
    self.discrete = (np.issubdtype(self.dtype, np.integer) or
                     self.dtype == bool)

  def __repr__(self):
    return f'Space({self.dtype.name}, {self.shape})'


def make_spaces(image=64, vector=16, action=16):
  obs = {}
  if image:
    obs['image'] = Space(np.uint8, (image, image, 3))
  if vector:
    obs['vector'] = Space(np.float32, (vector,))
  obs.update(
      reward=Space(np.float32), is_first=Space(bool), is_last=Space(bool),
      is_terminal=Space(bool))
  act = {'action': Space(np.float32, (action,), -1, 1), 'reset': Space(bool)}
  return obs, act


def make_batch(obs_space, act_space, batch, length, seed=0, terminals=0.0,
               smooth_images=False):
  """Seeded synthetic minibatch."""
  rng = np.random.default_rng(seed)
  data = {}
  for key, space in obs_space.items():
    shape = (batch, length) + space.shape
    if key.startswith('is_') or key == 'reward':
      continue
    if space.dtype == np.uint8:
      if smooth_images:
        h, w = space.shape[:2]
        yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w),
                             indexing='ij')
        ph = rng.uniform(0, 2 * np.pi, (batch, length, 1, 1, space.shape[2]))
        fr = rng.uniform(1, 4, (batch, length, 1, 1, space.shape[2]))
        img = 0.5 + 0.5 * np.sin(
            2 * np.pi * fr * (yy + xx)[None, None, :, :, None] / 2 + ph)
        data[key] = (img * 255).astype(np.uint8)
      else:
        data[key] = rng.integers(0, 256, shape, dtype=np.uint8)
    else:
      data[key] = rng.standard_normal(shape).astype(np.float32)
  adim = act_space['action'].shape
  data['action'] = rng.uniform(-1, 1, (batch, length) + adim).astype(
      np.float32)
  data['reward'] = rng.standard_normal((batch, length)).astype(np.float32)
  first = np.zeros((batch, length), bool)
  first[:, 0] = True
  term = rng.random((batch, length)) < terminals
  data['is_first'] = first
  data['is_last'] = term.copy()
  data['is_terminal'] = term
  data['reset'] = np.zeros((batch, length), bool)
  return data
