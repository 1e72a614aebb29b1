"""Static description of the learner's networks: dimensions and the ordered
parameter list (name, shape, optimizer group, initialiser).

Shapes and initialisers follow the reference's lazily-built modules
(reference nets.py): Linear init `U(+-sqrt(3*outscale/mean(fan_in,fan_out)))`
(nets.py:567-569), Conv2D init ignores the kernel area (nets.py:541-543),
transposed Conv2D includes it (nets.py:523-525); biases zero, LayerNorm
scale one / bias zero (nets.py:595-596); `initial_deter` zero (nets.py:58-60).
Linear has a bias only without LayerNorm (nets.py:563); Conv2D always has one
(nets.py:548-553) unless built with bias=False (the 1x1 skip convolutions of the residual
blocks, nets.py:354, 387).

`cnn: resnet` (ImageEncoderResnet / ImageDecoderResnet, nets.py:330-391): stride-1 SAME 3x3
convolutions with pre-activation (LayerNorm over the INPUT channels, then the activation, then
the convolution: nets.py:510-513), residual blocks `skip + 0.1 * x`, 2x2 average pooling between
encoder stages, 2x repetition between decoder stages, a Linear(1024) on the 4x4 encoder output
and a Linear(16 * depth) in front of the decoder.
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
class ResBlock:
  """One residual block (reference nets.py:351-358 / 384-391) on h x h pixels."""
  name: str      # parameter prefix: <name>a, <name>b (3x3, pre-activation), <name>s (1x1 skip)
  h: int
  cin: int
  depth: int

  @property
  def skip(self):
    return self.cin != self.depth


@dataclass
class ResNet:
  prefix: str            # 'enc/cnn' | 'dec/cnn'
  hw: int                # image side
  cimg: int              # image channels
  depth: int             # channels after 'in' (encoder) / in front of 'out' (decoder)
  stages: list = field(default_factory=list)   # per stage: list of ResBlock
  feat_h: int = 4        # side of the flattened end (4 x 4)
  feat_c: int = 0        # channels at the 4 x 4 end
  units: int = 0         # encoder: width of the 'out' Linear; decoder: width of the 'in' Linear


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
  enc_res: ResNet = None   # cnn: resnet
  dec_res: ResNet = None
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

  def conv_same(name, k, cin, cout, bias=True, preact=False):
    # Conv2D, not transposed: the limit ignores the kernel area (nets.py:541-543); with
    # preact the LayerNorm acts on the INPUT (cin channels), nets.py:510-513
    add(f'{name}/kernel', (k, k, cin, cout), 'model', 'uniform',
        float(np.sqrt(3.0 / np.mean([cin, cout]))))
    if bias:
      add(f'{name}/bias', (cout,), 'model', 'zeros')
    if preact:
      add(f'{name}/norm/scale', (cin,), 'model', 'ones')
      add(f'{name}/norm/bias', (cin,), 'model', 'zeros')

  def res_block(name, h, cin, depth):
    blk = ResBlock(name, h, cin, depth)
    if blk.skip:
      conv_same(f'{name}s', 1, cin, depth, bias=False)
    conv_same(f'{name}a', 3, cin, depth, preact=True)
    conv_same(f'{name}b', 3, depth, depth, preact=True)
    return blk

  def res_stage_count(hw):
    n = int(np.log2(hw)) - 2   # nets.py:339, 372
    assert n >= 1 and 2 ** (n + 2) == hw, ('resnet needs a power-of-two image side >= 8', hw)
    return n

  shapes = {k: tuple(v) for k, v in obs_shapes.items()
            if not k.startswith('log_')}
  # ---- encoder (reference nets.py:186-232, 291-305)
  enc = cfg['encoder']
  assert enc['cnn'] in ('simple', 'resnet') and enc['norm'] == 'layer' and enc['act'] == 'elu'
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
    if enc['cnn'] == 'resnet':   # nets.py:330-358
      net = ResNet('enc/cnn', h, cin, depth)
      conv_same('enc/cnn/in', 3, cin, depth)
      c = depth
      for i in range(res_stage_count(h)):
        h //= 2   # avg_pool in front of the stage's blocks
        blocks = []
        for j in range(enc['cnn_blocks']):
          blocks.append(res_block(f'enc/cnn/s{i}b{j}', h, c, depth))
          c = depth
        net.stages.append(blocks)
        depth *= 2
      net.feat_h, net.feat_c, net.units = h, c, 1024
      dense_bias('enc/cnn/out', h * h * c, 1024, 'model')
      s.enc_res = net
      s.embed += 1024
    for i, k in enumerate(enc['cnn_kernels'] if enc['cnn'] == 'simple' else ()):
      ho = (h - k) // 2 + 1
      name = f'enc/cnn/conv{i}'
      add(f'{name}/kernel', (k, k, cin, depth), 'model', 'uniform',
          float(np.sqrt(3.0 / np.mean([cin, depth]))))
      add(f'{name}/bias', (depth,), 'model', 'zeros')
      add(f'{name}/norm/scale', (depth,), 'model', 'ones')
      add(f'{name}/norm/bias', (depth,), 'model', 'zeros')
      s.enc_convs.append(ConvLayer(name, k, depth, cin, ho, h, True))
      h, cin, depth = ho, depth, depth * 2
    if enc['cnn'] == 'simple':
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
  assert dec['cnn'] in ('simple', 'resnet') and dec['image_dist'] == 'mse'
  assert dec['norm'] == 'layer' and dec['act'] == 'elu'
  assert list(dec['inputs']) == ['deter', 'stoch']
  ds = {k: v for k, v in shapes.items()
        if k not in ('is_first', 'is_last', 'is_terminal', 'reward')}
  s.dec_cnn_keys = {k: v for k, v in ds.items()
                    if re.match(dec['cnn_keys'], k) and len(v) == 3}
  s.dec_mlp_keys = {k: v for k, v in ds.items()
                    if re.match(dec['mlp_keys'], k) and len(v) == 1}
  if s.dec_cnn_keys and dec['cnn'] == 'resnet':   # nets.py:361-391
    cimg = sum(v[2] for v in s.dec_cnn_keys.values())
    hw = list(s.dec_cnn_keys.values())[0][0]
    assert all(v[0] == hw and v[1] == hw for v in s.dec_cnn_keys.values())
    n = res_stage_count(hw)
    depth = 2 ** n * dec['cnn_depth']
    net = ResNet('dec/cnn', hw, cimg, 0, feat_h=4, feat_c=depth, units=16 * depth)
    dense_bias('dec/cnn/in', s.feat, 16 * depth, 'model')
    h, c = 4, depth
    for i in range(n):
      blocks = []
      for j in range(dec['cnn_blocks']):
        blocks.append(res_block(f'dec/cnn/s{i}b{j}', h, c, depth))
        c = depth
      net.stages.append(blocks)
      h *= 2    # repeated 2x after the stage's blocks
      depth //= 2
    net.depth = c
    conv_same('dec/cnn/out', 3, c, cimg)
    s.dec_res = net
    # the image layer in the terms of the stride-2 decoder's bookkeeping (its z / dz buffers,
    # the image loss and the bias-gradient fold are shared)
    s.dec_convs.append(ConvLayer('dec/cnn/out', 3, c, cimg, hw, hw, False))
  elif s.dec_cnn_keys:
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
