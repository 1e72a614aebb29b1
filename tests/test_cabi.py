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
  assert lib.dd_version() >= 1


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
