"""Static description of the learner's networks: dimensions and the ordered
parameter list (name, shape, optimizer group, initialiser).

Shapes and initialisers follow the reference's lazily-built modules
(reference nets.py): Linear init `U(+-sqrt(3*outscale/mean(fan_in,fan_out)))`
(nets.py:567-569), Conv2D init ignores the kernel area (nets.py:541-543),
transposed Conv2D includes it (nets.py:523-525); biases zero, LayerNorm
scale one / bias zero (nets.py:595-596); `initial_deter` zero (nets.py:58-60).
Linear has a bias only without LayerNorm (nets.py:563); Conv2D always has one
(nets.py:548-553).
"""

import re
from dataclasses import dataclass, field

import numpy as np


@dataclass
class ParamSpec:
  name: str
  shape: tuple
  group: str            # 'model' | 'actor' | 'critic' | 'critic_target'
  init: str             # 'uniform' | 'zeros' | 'ones'
  limit: float = 0.0
  decay: bool = False   # matches the optimizer's wd_pattern ('kernel')

  @property
  def size(self):
    return int(np.prod(self.shape)) if self.shape else 1


@dataclass
class ConvLayer:
  name: str
  k: int
  c_small: int   # channels on the stride-2-downsampled side
  c_big: int     # channels on the full-resolution side
  h_small: int
  h_big: int
  norm: bool


@dataclass
class ModelSpec:
  cfg: dict
  obs_shapes: dict
  act_dim: int
  act_discrete: bool = False
  deter: int = 0
  units: int = 0
  groups: int = 0       # number of categorical latents
  classes: int = 0
  stoch: int = 0        # groups * classes
  feat: int = 0         # deter + stoch
  embed: int = 0
  enc_cnn_keys: list = field(default_factory=list)
  enc_mlp_keys: list = field(default_factory=list)
  enc_mlp_in: int = 0
  dec_cnn_keys: dict = field(default_factory=dict)
  dec_mlp_keys: dict = field(default_factory=dict)
  enc_convs: list = field(default_factory=list)
  dec_convs: list = field(default_factory=list)
  image_hw: int = 0
  image_c: int = 0
  params: list = field(default_factory=list)

  def group(self, name):
    return [p for p in self.params if p.group == name]


def _lin_limit(fan_in, fan_out, outscale=1.0):
  return float(np.sqrt(3.0 * outscale / np.mean([fan_in, fan_out])))


def build_spec(cfg, obs_shapes, act_dim, act_discrete=False):
  """cfg: nested plain dict; obs_shapes: name -> shape tuple; act_discrete:
  one-hot action space ('onehot' actor, reference nets.py:480-491)."""
  s = ModelSpec(cfg=cfg, obs_shapes=dict(obs_shapes), act_dim=act_dim,
                act_discrete=act_discrete)
  r = cfg['rssm']
  assert r['classes'], 'only the discrete latent is implemented'
  assert r['initial'] == 'learned2' and r['gru_layers'] == 1
  assert r['post_layers'] == 1 and r['norm'] == 'layer' and r['act'] == 'elu'
  s.deter, s.units = r['deter'], r['units']
  s.groups, s.classes = r['stoch'], r['classes']
  s.stoch = s.groups * s.classes
  s.feat = s.deter + s.stoch
  P = s.params
  wdp = cfg['model_opt'].get('wd_pattern', 'kernel')

  def add(name, shape, group, init, limit=0.0):
    P.append(ParamSpec(name, tuple(shape), group, init, limit,
                       decay=bool(re.search(wdp, group + '/' + name))))

  def dense_ln(prefix, fan_in, units, group):
    add(f'{prefix}/kernel', (fan_in, units), group, 'uniform',
        _lin_limit(fan_in, units))
    add(f'{prefix}/norm/scale', (units,), group, 'ones')
    add(f'{prefix}/norm/bias', (units,), group, 'zeros')

  def dense_bias(prefix, fan_in, units, group, outscale=1.0):
    add(f'{prefix}/kernel', (fan_in, units), group, 'uniform',
        _lin_limit(fan_in, units, outscale))
    add(f'{prefix}/bias', (units,), group, 'zeros')

  def trunk(prefix, fan_in, layers, units, group):
    for i in range(layers):
      dense_ln(f'{prefix}/dense{i}', fan_in, units, group)
      fan_in = units
    return fan_in

  shapes = {k: tuple(v) for k, v in obs_shapes.items()
            if not k.startswith('log_')}
  # ---- encoder (reference nets.py:186-232, 291-305)
  enc = cfg['encoder']
  assert enc['cnn'] == 'simple' and enc['norm'] == 'layer'
  es = {k: v for k, v in shapes.items() if k not in ('is_first', 'is_last')}
  s.enc_cnn_keys = [k for k, v in es.items()
                    if re.match(enc['cnn_keys'], k) and len(v) == 3]
  s.enc_mlp_keys = [k for k, v in es.items()
                    if re.match(enc['mlp_keys'], k) and len(v) in (0, 1)]
  s.embed = 0
  if s.enc_cnn_keys:
    hw = {es[k][:2] for k in s.enc_cnn_keys}
    assert len(hw) == 1 and es[s.enc_cnn_keys[0]][0] == es[s.enc_cnn_keys[0]][1]
    s.image_hw = es[s.enc_cnn_keys[0]][0]
    s.image_c = sum(es[k][2] for k in s.enc_cnn_keys)
    h, cin, depth = s.image_hw, s.image_c, enc['cnn_depth']
    for i, k in enumerate(enc['cnn_kernels']):
      ho = (h - k) // 2 + 1
      name = f'enc/cnn/conv{i}'
      add(f'{name}/kernel', (k, k, cin, depth), 'model', 'uniform',
          float(np.sqrt(3.0 / np.mean([cin, depth]))))
      add(f'{name}/bias', (depth,), 'model', 'zeros')
      add(f'{name}/norm/scale', (depth,), 'model', 'ones')
      add(f'{name}/norm/bias', (depth,), 'model', 'zeros')
      s.enc_convs.append(ConvLayer(name, k, depth, cin, ho, h, True))
      h, cin, depth = ho, depth, depth * 2
    s.embed += h * h * cin
  if s.enc_mlp_keys:
    s.enc_mlp_in = sum(int(np.prod(es[k])) if es[k] else 1
                       for k in s.enc_mlp_keys)
    s.embed += trunk('enc/mlp', s.enc_mlp_in, enc['mlp_layers'],
                     enc['mlp_units'], 'model')
  assert s.embed > 0, 'encoder has no inputs'
  # ---- RSSM (reference nets.py:11-183)
  add('rssm/initial_deter', (s.deter,), 'model', 'zeros')
  dense_ln('rssm/img_in', s.stoch + act_dim, s.units, 'model')
  dense_ln('rssm/gru_out', s.deter + s.units, 3 * s.deter, 'model')
  fan = s.deter
  for i in range(r['prior_layers']):
    dense_ln(f'rssm/img_out_{i}', fan, s.units, 'model')
    fan = s.units
  dense_bias('rssm/img_stats', s.units, s.stoch, 'model')
  dense_ln('rssm/obs_out', s.deter + s.embed, s.units, 'model')
  dense_bias('rssm/obs_stats', s.units, s.stoch, 'model')
  # ---- decoder (reference nets.py:235-327)
  dec = cfg['decoder']
  assert dec['cnn'] == 'simple' and dec['image_dist'] == 'mse'
  assert list(dec['inputs']) == ['deter', 'stoch']
  ds = {k: v for k, v in shapes.items()
        if k not in ('is_first', 'is_last', 'is_terminal', 'reward')}
  s.dec_cnn_keys = {k: v for k, v in ds.items()
                    if re.match(dec['cnn_keys'], k) and len(v) == 3}
  s.dec_mlp_keys = {k: v for k, v in ds.items()
                    if re.match(dec['mlp_keys'], k) and len(v) == 1}
  if s.dec_cnn_keys:
    kernels = list(dec['cnn_kernels'])
    cimg = sum(v[2] for v in s.dec_cnn_keys.values())
    depth = dec['cnn_depth'] * 2 ** (len(kernels) - 2)
    h, cin = 1, s.feat
    for i, k in enumerate(kernels[:-1]):
      hb = 2 * h + k - 2
      name = f'dec/cnn/conv{i}'
      add(f'{name}/kernel', (k, k, depth, cin), 'model', 'uniform',
          float(np.sqrt(3.0 / (k * k * np.mean([depth, cin])))))
      add(f'{name}/bias', (depth,), 'model', 'zeros')
      add(f'{name}/norm/scale', (depth,), 'model', 'ones')
      add(f'{name}/norm/bias', (depth,), 'model', 'zeros')
      s.dec_convs.append(ConvLayer(name, k, cin, depth, h, hb, True))
      h, cin, depth = hb, depth, depth // 2
    k = kernels[-1]
    hb = 2 * h + k - 2
    add('dec/cnn/out/kernel', (k, k, cimg, cin), 'model', 'uniform',
        float(np.sqrt(3.0 / (k * k * np.mean([cimg, cin])))))
    add('dec/cnn/out/bias', (cimg,), 'model', 'zeros')
    s.dec_convs.append(ConvLayer('dec/cnn/out', k, cin, cimg, h, hb, False))
    hw = list(s.dec_cnn_keys.values())[0][0]
    assert hb == hw, ('decoder output size', hb, hw)
  if s.dec_mlp_keys:
    fan = trunk('dec/mlp', s.feat, dec['mlp_layers'], dec['mlp_units'],
                'model')
    for k, v in s.dec_mlp_keys.items():
      dense_bias(f'dec/mlp/dist_{k}/out', fan, int(np.prod(v)), 'model',
                 outscale=0.1)  # DistLayer default outscale, nets.py:431
  # ---- heads (reference agent.py:152-153, nets.py:394-492)
  for name, key in (('reward', 'reward_head'), ('cont', 'cont_head')):
    c = cfg[key]
    assert c['norm'] == 'layer' and list(c['inputs']) == ['deter', 'stoch']
    fan = trunk(name, s.feat, c['layers'], c['units'], 'model')
    dense_bias(f'{name}/dist_out/out', fan, 1, 'model', c['outscale'])
  c = cfg['actor']
  fan = trunk('actor', s.feat, c['layers'], c['units'], 'actor')
  dense_bias('actor/dist_out/out', fan, act_dim, 'actor', c['outscale'])
  if not act_discrete:  # the std layer exists only for 'normal' (nets.py:453-456)
    dense_bias('actor/dist_out/std', fan, act_dim, 'actor', 1.0)
  c = cfg['critic']
  for group in ('critic', 'critic_target'):
    fan = trunk(group, s.feat, c['layers'], c['units'], group)
    dense_bias(f'{group}/dist_out/out', fan, 1, group, c['outscale'])
  return s


def init_params(spec, seed=0):
  """Random initial weights as a name -> float32 ndarray dict."""
  rng = np.random.RandomState(seed)
  out = {}
  for p in spec.params:
    if p.init == 'uniform':
      out[p.name] = rng.uniform(-p.limit, p.limit, p.shape).astype(np.float32)
    elif p.init == 'ones':
      out[p.name] = np.ones(p.shape, np.float32)
    else:
      out[p.name] = np.zeros(p.shape, np.float32)
  return out
