import numpy as np

from daydreamer_amd import config, spec, synthetic


def test_yaml12_scalars():
  d = config.load_yaml('a: 1e-4\nb: off\nc: 1e6\nd: True\ne: [1, 2]\nf: 3e-3\ng: yes\n')
  assert d['a'] == 1e-4 and isinstance(d['a'], float)
  assert d['b'] == 'off' and d['g'] == 'yes'
  assert d['c'] == 1e6 and d['d'] is True and d['f'] == 3e-3


def test_configs_defaults_and_blocks():
  cfgs = config.load_configs()
  cfg = config.Config(cfgs['defaults'])
  assert cfg.transform_rewards == 'off'
  assert cfg.model_opt.lr == 1e-4 and cfg.model_opt.eps == 1e-6
  assert cfg.rssm.deter == 1024 and cfg['rssm.units'] == 1024
  a1 = cfg.update(cfgs['a1'])
  assert a1.rssm.deter == 256 and a1.actor.minstd == 0.1 and a1.discount == 0.995
  dbg = a1.update(cfgs['debug'])
  assert dbg.actor.units == 64 and dbg.reward_head.layers == 2 and dbg.rssm.units == 64
  assert dbg.encoder.mlp_units == 512  # '.*\\.units' does not match 'mlp_units'
  assert dbg.model_opt.wd == 0.0
  # type-preserving casts and unknown keys
  assert isinstance(cfg.update({'replay_size': 5e5}).replay_size, float)
  assert cfg.update({'batch_size': 16.0}).batch_size == 16
  try:
    cfg.update({'nonexistent': 1})
    assert False
  except KeyError:
    pass


def test_param_counts_match_survey():
  """C2 parameter counts (19.33 M world model, 1.46 M actor,
  1.45 M critic)."""
  cfgs = config.load_configs()
  cfg = config.Config(cfgs['defaults']).update(cfgs['a1_vision'])
  obs, act = synthetic.make_spaces(64, 16, 16)
  sp = spec.build_spec(config.to_plain(cfg), {k: v.shape for k, v in obs.items()}, 16)
  n = lambda g: sum(p.size for p in sp.group(g))
  assert abs(n('model') / 1e6 - 19.33) < 0.02, n('model')
  assert abs(n('actor') / 1e6 - 1.46) < 0.01
  assert abs(n('critic') / 1e6 - 1.45) < 0.01
  assert sp.embed == 2560 and sp.feat == 1280
