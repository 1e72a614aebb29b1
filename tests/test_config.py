import numpy as np
import pytest

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


def test_config_blocks_mirror_the_reference():
  """daydreamer_amd/configs.yaml against the reference's configs.yaml (skipped where the
  checkout is absent): every key both `defaults` share has the reference's value - except the
  three deliberate ones (float32 arithmetic instead of float16 / TF32, our own prefetcher) - and the
  robot blocks (a1, xarm, ur5) are the reference's key for key; the keys we do not carry belong to
  the exploration behaviours (out of scope)."""
  import pathlib
  import pytest
  path = pathlib.Path('/root/reference/embodied/agents/dreamerv2plus/configs.yaml')
  if not path.exists():
    pytest.skip('reference checkout not present')
  ref, ours = config.load_yaml(path.read_text()), config.load_configs()

  def flat(d, prefix=''):
    out = {}
    for k, v in d.items():
      if isinstance(v, dict):
        out.update(flat(v, f'{prefix}{k}.'))
      else:
        out[prefix + k] = list(v) if isinstance(v, (list, tuple)) else v
    return out

  a, b = flat(ref['defaults']), flat(ours['defaults'])
  shared = [k for k in a if k in b]
  assert len(shared) >= 200
  diff = {k: (a[k], b[k]) for k in shared if a[k] != b[k]}
  assert diff == {'tf.precision': ('float16', 'float32'), 'tf.tensorfloat': (True, False),
                  'data_loader': ('tfdata', 'embodied')}, diff
  ref_only = [k for k in a if k not in b]
  assert all(k.startswith(('expl_', 'disag_', 'ctrl_', 'pbe_')) for k in ref_only), ref_only
  assert all(k.startswith('hip.') for k in b if k not in a)
  for block in ('a1', 'xarm', 'ur5'):
    x, y = flat(ref[block]), flat(ours[block])
    drop = lambda d: {k: v for k, v in d.items() if k != 'train.log_keys_video'}
    assert drop(x) == drop(y), block


def test_every_documented_hip_knob_is_settable():
  """Every `hip.<knob>` that INTEGRATION.md / README.md / DESIGN.md name exists in configs.yaml's
  `hip:` block and can be set through Config.update (round-4 review: `hip.fused_imag` was
  documented but `update` raised KeyError), and every key of the block is documented in
  INTEGRATION.md."""
  import pathlib
  import re
  root = pathlib.Path(__file__).resolve().parents[1]
  cfg = config.Config(config.load_configs()['defaults'])
  block = dict(cfg['hip'])
  named = set()
  for doc in ('INTEGRATION.md', 'README.md', 'DESIGN.md'):
    named |= set(re.findall(r'hip\.([a-z_0-9]+)\b', (root / doc).read_text()))
  named -= {'h', 'so', 'hip'}   # (daydreamer_hip.h, libdaydreamer_hip.so)
  assert named, 'no knob found in the documents'
  for knob in sorted(named):
    assert knob in block, f'hip.{knob} is documented but not in configs.yaml'
    old = block[knob]
    new = (not old) if isinstance(old, bool) else old
    assert cfg.update({f'hip.{knob}': new})['hip'][knob] == new
  text = (root / 'INTEGRATION.md').read_text()
  for knob in block:
    assert f'hip.{knob}' in text, f'hip.{knob} is in configs.yaml but not documented in INTEGRATION.md'


def test_pipeline_mode():
  """hip.pipeline: auto (default) = on for one process, off under data parallelism; a bool set
  through Config.update arrives as 'True' / 'False'."""
  from daydreamer_amd.agent import pipeline_mode
  cfg = config.Config(config.load_configs()['defaults'])
  assert cfg['hip']['pipeline'] == 'auto'
  assert pipeline_mode('auto', 1) and not pipeline_mode('auto', 2) and not pipeline_mode('auto', 1, graph=False)
  for v, want in ((True, True), (False, False), ('true', True), ('false', False)):
    got = cfg.update({'hip.pipeline': v})['hip']['pipeline']
    assert pipeline_mode(got, 1) is want and pipeline_mode(got, 4) is want, (v, got)
  with pytest.raises(AssertionError):
    pipeline_mode('sometimes', 1)
