"""Driver entry points: build() compiles every HIP extension for gfx950 and
the oracle's C helpers; smoke() runs one tiny learner step on cuda:0 and checks
it against the CPU oracle."""

import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))


def build():
  subprocess.run(['make', '-j8', '-C', str(ROOT / 'daydreamer_amd' / 'csrc')],
                 check=True)
  import daydreamer_amd  # noqa: F401
  from daydreamer_amd import hipops
  import ctypes
  lib = ctypes.CDLL(str(hipops._LIB_PATH))
  for name in hipops.EXPORTS:
    getattr(lib, name)


def smoke():
  from daydreamer_amd import selfcheck
  selfcheck.smoke()


if __name__ == '__main__':
  build()
  print('build ok')
