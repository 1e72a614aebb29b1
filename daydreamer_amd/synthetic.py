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


def _flags():
  return dict(reward=Space(np.float32), is_first=Space(bool), is_last=Space(bool),
              is_terminal=Space(bool))


def config_spaces(name):
  """Observation / action spaces of the BASELINE.json workloads, by config block name:
    a1            proprio only (vector 7), 6-dim continuous action          configs[0]
    a1_vision     64x64 RGB + 16-dim proprio, 16-dim continuous action      configs[1]
    xarm          image 64x64x3 + depth 64x64x1 + 20 proprio dims in five keys (reference
                  envs/robot_interface.py:395-410, 7 joints), one-hot 6-way action  configs[2]
    ur5_multicam  two 128x128 RGB cameras (image, image2) + 19 proprio dims (6 joints),
                  one-hot 6-way action                                      configs[3]
    a1_scaled     as a1_vision                                              configs[4]
  """
  def robot(joints):
    return {'cartesian_position': Space(np.float32, (6,)),
            'joint_positions': Space(np.float32, (joints,)),
            'gripper_pos': Space(np.float32, (1,)), 'gripper_side': Space(np.float32, (3,)),
            'grasped_side': Space(np.float32, (3,))}
  def onehot(n):  # embodied.wrappers.OneHotAction: float32 [n] with .discrete = True
    sp = Space(np.float32, (n,), 0, 1)
    sp.discrete = True
    return {'action': sp, 'reset': Space(bool)}
  if name == 'a1':
    return make_spaces(0, 7, 6)
  if name in ('a1_vision', 'a1_scaled'):
    return make_spaces(64, 16, 16)
  if name == 'xarm':
    obs = {'image': Space(np.uint8, (64, 64, 3)), 'depth': Space(np.uint8, (64, 64, 1)),
           **robot(7), **_flags()}
    return obs, onehot(6)
  if name in ('ur5_multicam',):
    obs = {'image': Space(np.uint8, (128, 128, 3)), 'image2': Space(np.uint8, (128, 128, 3)),
           **robot(6), **_flags()}
    return obs, onehot(6)
  if name == 'ur5':
    return {'image': Space(np.uint8, (64, 64, 3)), **robot(6), **_flags()}, onehot(6)
  raise KeyError(name)


def make_batch(obs_space, act_space, batch, length, seed=0, terminals=0.0,
               smooth_images=False):
  """Seeded synthetic minibatch; one-hot actions for a discrete
  action space."""
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
  if getattr(act_space['action'], 'discrete', False):
    idx = np.random.RandomState(seed + 17).randint(0, adim[0], (batch, length))
    data['action'] = np.eye(adim[0], dtype=np.float32)[idx]
  data['reward'] = rng.standard_normal((batch, length)).astype(np.float32)
  first = np.zeros((batch, length), bool)
  first[:, 0] = True
  term = rng.random((batch, length)) < terminals
  data['is_first'] = first
  data['is_last'] = term.copy()
  data['is_terminal'] = term
  data['reset'] = np.zeros((batch, length), bool)
  return data
