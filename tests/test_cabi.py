"""The C-ABI shared library loads and exports every symbol that
include/daydreamer_hip.h declares (no compute calls: no GPU needed)."""

import ctypes
import pathlib
import re
import subprocess

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_library_exports_header_symbols():
  lib_path = ROOT / 'daydreamer_amd' / 'libdaydreamer_hip.so'
  if not lib_path.exists():
    subprocess.run(['make', '-j8', '-C', str(ROOT / 'daydreamer_amd' / 'csrc')], check=True)
  header = (ROOT / 'include' / 'daydreamer_hip.h').read_text()
  declared = set(re.findall(r'^(?:int|const char\*)\s+(dd_\w+)\s*\(', header, re.M))
  assert len(declared) >= 40, declared
  lib = ctypes.CDLL(str(lib_path))
  missing = [n for n in sorted(declared) if not hasattr(lib, n)]
  assert not missing, missing
  from daydreamer_amd import hipops
  assert set(hipops.EXPORTS) == declared, set(hipops.EXPORTS) ^ declared
  lib.dd_version.restype = ctypes.c_int
  header_version = int(re.search(r'#define DD_ABI_VERSION (\d+)', header).group(1))
  assert lib.dd_version() == header_version == hipops.ABI_VERSION


def test_product_fails_loudly_without_gpu():
  import pytest
  import torch
  from daydreamer_amd import hipops
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  with pytest.raises(RuntimeError):
    hipops.HipOps('cuda:0')


def test_product_never_imports_oracle():
  for path in (ROOT / 'daydreamer_amd').glob('*.py'):
    text = path.read_text()
    assert 'import oracle' not in text and 'from oracle' not in text, path


def test_persistent_scan_shape_queries():
  """dd_observe_scan_supported / dd_observe_scan_bwd_supported are host logic (no launch): the
  learner falls back to the per-layer launch sequence wherever they say no."""
  lib = ctypes.CDLL(str(ROOT / 'daydreamer_amd' / 'libdaydreamer_hip.so'))
  fwd, bwd = lib.dd_observe_scan_supported, lib.dd_observe_scan_bwd_supported
  # (B, deter, units, groups, classes, action dims)
  assert fwd(50, 256, 256, 32, 32, 16) == 1 and bwd(50, 256, 256, 32, 32) == 1      # configs[1]
  assert fwd(64, 256, 256, 32, 32, 16) == 1 and fwd(1, 256, 256, 32, 32, 6) == 1
  assert fwd(65, 256, 256, 32, 32, 16) == 0 and bwd(65, 256, 256, 32, 32) == 0      # > 4 row blocks of 16
  assert fwd(0, 256, 256, 32, 32, 16) == 0
  assert fwd(25, 512, 512, 32, 32, 6) == 1 and bwd(25, 512, 512, 32, 32) == 1       # xarm / ur5: weight planes streamed
  assert fwd(32, 4096, 256, 64, 64, 16) == 0 and bwd(32, 4096, 256, 64, 64) == 0    # a1_scaled: launch sequence
  assert fwd(16, 128, 128, 8, 32, 6) == 0


def test_fused_imagination_shape_queries():
  """dd_imagine_rollout_supported is host logic (no launch): the learner keeps the per-layer
  launch sequence wherever it says no.  (deter, units, groups, classes, action dims, actor
  units, actor layers, prior layers, discrete)"""
  lib = ctypes.CDLL(str(ROOT / 'daydreamer_amd' / 'libdaydreamer_hip.so'))
  q = lib.dd_imagine_rollout_supported
  assert q(256, 256, 32, 32, 16, 512, 4, 3, 0) == 1      # configs[1]
  assert q(256, 256, 32, 32, 6, 512, 4, 3, 0) == 1       # configs[0]
  assert q(256, 256, 32, 32, 16, 512, 4, 3, 1) == 0      # one-hot actions at 256: launch sequence
  assert q(512, 512, 32, 32, 6, 512, 4, 3, 1) == 1       # xarm / ur5: forward-only one-hot rollout (imag_oh.hip)
  assert q(512, 512, 32, 32, 6, 512, 4, 3, 0) == 0       # continuous actions at 512: launch sequence
  assert q(512, 512, 32, 32, 9, 512, 4, 3, 1) == 0       # (the serial action draw covers <= 8 classes)
  assert q(4096, 256, 64, 64, 16, 512, 4, 3, 0) == 0     # a1_scaled
  assert q(256, 256, 32, 32, 16, 512, 2, 3, 0) == 0      # debug block: 2 actor layers
  assert q(256, 256, 32, 32, 16, 256, 4, 3, 0) == 0


def test_exact_column_registry_follows_views():
  """hipops.mark_exact / _exact_cols (host logic of dd_gemm_f32_x): a contraction operand that is a
  view into a registered buffer - row slices, column slices, the flattened [M, W] view of the
  [H+1, N, W] trajectory - gets the exact column range translated into ITS columns; another
  leading dimension, another tensor or a collected owner gets none."""
  import torch
  from daydreamer_amd import hipops
  keep = list(hipops._EXACT)
  hipops._EXACT[:] = []
  try:
    traj = torch.zeros(4, 10, 48)            # [H+1, N, W]: deter 8 | stoch 32 | action 8
    hipops.mark_exact(traj, 8, 40)
    def q(t):
      return hipops._exact_cols(t.data_ptr(), t.stride(0), t.shape[1])
    flat = traj.view(40, 48)
    assert q(flat) == (8, 40)
    assert q(flat[:, :40]) == (8, 40)         # feat = [deter | stoch]
    assert q(flat[10:30, :40]) == (8, 40)     # a row range
    assert q(flat[:, 8:40]) == (0, 32)        # the stoch columns alone
    assert q(flat[:, 16:48]) == (0, 24)       # starts inside the range, runs past it
    assert q(flat[:, 40:]) == (0, 0)          # the action columns
    other = torch.zeros(40, 48)
    assert q(other) == (0, 0)
    assert q(traj.view(80, 24)) == (0, 0)     # another leading dimension: not a row of the buffer
    del traj, flat
    import gc
    gc.collect()
    assert q(other) == (0, 0) and not hipops._EXACT      # the dead entry is dropped on lookup
  finally:
    hipops._EXACT[:] = keep
