"""Config handling for the learner: a YAML-1.2-flavoured loader for
`configs.yaml` and a small immutable nested `Config`.

The reference reads its YAML with ruamel (YAML 1.2: `1e-4` is a float, `off` is
a string; reference agent.py:18-19) and wraps it in `embodied.Config`
(reference embodied/core/config.py:7-128: nested dict, dotted keys, regex
pattern updates, type-preserving casts).  PyYAML is YAML 1.1, so the loader
below patches the float / bool resolvers.  `Config` here is an independent
minimal implementation with the same observable behaviour for the operations
the learner and the reference's train script use (`cfg.a.b`, `cfg['a.b']`,
`update`, `flat`, dict-splat of sub-configs).
"""

import pathlib
import re

import yaml


class _Loader(yaml.SafeLoader):
  pass


# YAML 1.2 core schema: floats may omit the dot ("1e-4"); only true/false are
# booleans ("off"/"on"/"yes"/"no" stay strings).
_Loader.yaml_implicit_resolvers = {
    k: [(tag, rx) for tag, rx in v
        if tag not in ('tag:yaml.org,2002:float', 'tag:yaml.org,2002:bool')]
    for k, v in yaml.SafeLoader.yaml_implicit_resolvers.items()}
_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    re.compile(r'''^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                   |\.[0-9_]+(?:[eE][-+]?[0-9]+)?
                   |[0-9][0-9_]*[eE][-+]?[0-9]+
                   |\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$''', re.X),
    list('-+0123456789.'))
_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:bool',
    re.compile(r'^(?:true|True|TRUE|false|False|FALSE)$'), list('tTfF'))


def load_yaml(path_or_text):
  text = str(path_or_text)
  if '\n' not in text and pathlib.Path(text).exists():
    text = pathlib.Path(text).read_text()
  return yaml.load(text, Loader=_Loader)


def load_configs():
  """The agent's named config blocks (mirrors `Agent.configs`,
  reference agent.py:18-19)."""
  return load_yaml(pathlib.Path(__file__).parent / 'configs.yaml')


_IS_PATTERN = re.compile(r'.*[^A-Za-z0-9_.-].*')


def _flatten(mapping, prefix=''):
  out = {}
  for key, value in mapping.items():
    if isinstance(value, dict):
      sep = '\\.' if (_IS_PATTERN.match(key)) else '.'
      out.update(_flatten(value, f'{prefix}{key}{sep}'))
    else:
      out[f'{prefix}{key}'] = value
  return out


def _nest(flat):
  out = {}
  for key, value in flat.items():
    node = out
    parts = key.split('.')
    for part in parts[:-1]:
      node = node.setdefault(part, {})
    node[parts[-1]] = value
  return out


class Config(dict):
  """Immutable nested config with attribute access and pattern updates."""

  def __init__(self, *args, **kwargs):
    flat = _flatten(dict(*args, **kwargs))
    flat = {k: (tuple(v) if isinstance(v, list) else v)
            for k, v in flat.items()}
    object.__setattr__(self, '_flat', flat)
    super().__init__(_nest(flat))

  @property
  def flat(self):
    return dict(self._flat)

  def __getattr__(self, name):
    if name.startswith('_'):
      raise AttributeError(name)
    try:
      return self[name]
    except KeyError:
      raise AttributeError(name)

  def __getitem__(self, name):
    node = dict(self.items())
    for part in name.split('.'):
      if not isinstance(node, dict) or part not in node:
        raise KeyError(name)
      node = node[part]
    return Config(node) if isinstance(node, dict) else node

  def __contains__(self, name):
    try:
      self[name]
      return True
    except KeyError:
      return False

  def __setattr__(self, key, value):
    raise AttributeError('Config is immutable; use update().')

  __setitem__ = __setattr__

  def __reduce__(self):
    return (type(self), (_nest(self._flat),))

  def update(self, *args, **kwargs):
    result = dict(self._flat)
    for key, new in _flatten(dict(*args, **kwargs)).items():
      if _IS_PATTERN.match(key):
        rx = re.compile(key)
        keys = [k for k in result if rx.match(k)]
      else:
        keys = [key]
      if not keys or any(k not in result for k in keys):
        raise KeyError(f'Unknown key or pattern {key}.')
      for k in keys:
        old = result[k]
        if isinstance(new, list):
          new = tuple(new)
        try:
          if isinstance(old, bool):
            cast = bool(new) if not isinstance(new, str) else (
                new.lower() in ('true', '1'))
          elif isinstance(old, int) and isinstance(new, float):
            if float(int(new)) != new:
              raise ValueError(new)
            cast = int(new)
          elif isinstance(old, tuple):
            cast = tuple(new)
          else:
            cast = type(old)(new)
        except (ValueError, TypeError):
          raise TypeError(
              f"Cannot convert '{new}' to {type(old).__name__} for '{k}'.")
        result[k] = cast
    return type(self)(_nest(result))


def to_plain(config):
  """Nested plain dict (tuples -> lists) from a Config, an embodied.Config or a
  dict; the learner consumes this."""
  def conv(x):
    if isinstance(x, dict):
      return {k: conv(v) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
      return [conv(v) for v in x]
    return x
  return conv(dict(config))
