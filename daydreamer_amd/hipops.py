"""ctypes binding of libdaydreamer_hip.so (include/daydreamer_hip.h).

`HipOps` is the only compute backend of the learner: every method launches a
hand-written gfx950 kernel on torch's current HIP stream.  PyTorch is used for
device memory and streams only.  There is no CPU or eager fallback: if the
shared library is missing or no GPU is visible, construction raises.

Tensors are passed as torch tensors; 2-D arguments may be column slices of
wider buffers (stride(1) == 1, stride(0) = leading dimension).
"""

import ctypes
import pathlib

import torch

_LIB_PATH = pathlib.Path(__file__).parent / 'libdaydreamer_hip.so'
_lib = None

c_f = ctypes.c_float
c_d = ctypes.c_double
c_i = ctypes.c_int
c_l = ctypes.c_long
c_p = ctypes.c_void_p
c_z = ctypes.c_size_t
c_u = ctypes.c_uint
c_ull = ctypes.c_ulonglong

_SIGS = {
    'dd_gemm_set_mode': [c_i],
    'dd_gemm_set_ws': [c_i, c_i, c_i],
    'dd_gemm_f32': [c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_i, c_i, c_f, c_f, c_p, c_p, c_z, c_p, c_p],
    'dd_gemm_f32_x': [c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_i, c_i, c_f, c_f, c_p, c_p, c_z, c_p, c_i, c_i, c_i, c_i, c_p],
    'dd_splitk_finish': [c_p, c_i, c_p, c_l, c_i, c_i, c_f, c_p, c_p],
    'dd_conv2d_s2_down': [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_z, c_p],
    'dd_conv2d_s2_up': [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_z, c_p],
    'dd_conv2d_s2_wgrad': [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_z, c_p],
    'dd_conv2d_s2_down_ln': [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_z, c_p],
    'dd_conv2d_s2_down_lnbwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_z, c_p],
    'dd_conv2d_s2_wgrad_ln': [c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_z, c_p],
    'dd_conv2d_same': [c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_z, c_p],
    'dd_conv2d_same_bwd_data': [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_z, c_p],
    'dd_conv2d_same_wgrad': [c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_p, c_z, c_p],
    'dd_pool2': [c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_p],
    'dd_repeat2': [c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_f, c_p],
    'dd_ln_act_fwd': [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_i, c_f, c_p, c_p],
    'dd_ln_act_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_z, c_p, c_i, c_f, c_p],
    'dd_ln_act_fwd_head': [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p],
    'dd_ln_act_bwd_head': [c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_z, c_p],
    'dd_ln_bwd_parts': [c_i, c_i],
    'dd_ln_param_grad': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_z, c_p],
    'dd_col_sum': [c_p, c_l, c_p, c_f, c_l, c_i, c_p, c_z, c_p],
    'dd_gru_cell_fwd': [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_p, c_i, c_f, c_p],
    'dd_gru_cell_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    'dd_observe_scan_supported': [c_i] * 6,
    'dd_scan_wprep': [c_p, c_l, c_i, c_i, c_i, c_p, c_p],
    'dd_observe_scan_fwd': [c_i] * 8 + [c_f] + [c_p] * 32,
    'dd_scan_wprep_rows': [c_p, c_l, c_i, c_i, c_p, c_p],
    'dd_observe_scan_bwd_supported': [c_i] * 5,
    'dd_observe_scan_bwd': [c_i] * 7 + [c_f] + [c_p] * 30,
    'dd_stats_sample_fwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_i, c_p, c_i, c_f, c_p, c_p],
    'dd_onehot_sample_host': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_i],
    'dd_stats_sample_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_p],
    'dd_cat_kl_fwd': [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    'dd_cat_kl_bwd': [c_p, c_l, c_p, c_l, c_p, c_f, c_f, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    'dd_image_loss': [c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_i, c_f, c_p],
    'dd_video_grid': [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_l, c_l, c_p],
    'dd_mse_loss': [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    'dd_scalar_loss': [c_p, c_p, c_p, c_p, c_l, c_f, c_i, c_p],
    'dd_normal_head_fwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_f, c_f, c_p],
    'dd_normal_head_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_p],
    'dd_action_noise': [c_p, c_l, c_p, c_l, c_i, c_i, c_f, c_i, c_p],
    'dd_actent_stats': [c_p, c_l, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_p, c_z, c_p],
    'dd_imag_returns_fwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_f, c_f, c_i, c_p],
    'dd_imag_returns_bwd': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_l, c_f, c_f, c_p],
    'dd_critic_loss': [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_p],
    'dd_actor_seed': [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_p],
    'dd_sub': [c_p, c_p, c_p, c_l, c_p],
    'dd_symexp': [c_p, c_p, c_l, c_p],
    'dd_onehot_entropy': [c_p, c_l, c_p, c_i, c_i, c_f, c_p],
    'dd_onehot_policy_grad': [c_p, c_l, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p],
    'dd_philox': [c_p, c_l, c_l, c_i, c_l, c_l, c_ull, c_p, c_u, c_i, c_p],
    'dd_counter_add': [c_p, c_ull, c_p],
    'dd_stamp': [c_p, c_p],
    'dd_reduce_stats_multi': [c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    'dd_reduce_stats': [c_p, c_l, c_l, c_p, c_p, c_p],
    'dd_autoadapt_update': [c_p, c_p, c_i, c_d, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_p],
    'dd_normalize_update': [c_p, c_p, c_d, c_p, c_d, c_d, c_i, c_i, c_p, c_p],
    'dd_scalar_mul': [c_p, c_p, c_p, c_f, c_i, c_p],
    'dd_grad_norm': [c_p, c_l, c_p, c_p, c_z, c_i, c_p],
    'dd_adam_step': [c_p, c_p, c_p, c_p, c_l, c_l, c_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p],
    'dd_fill': [c_p, c_l, c_f, c_p],
    'dd_reset_mask2': [c_p, c_l, c_p, c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_l, c_i, c_p, c_l, c_l, c_p],
    'dd_reset_mask_bwd2': [c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_p, c_l, c_i, c_p, c_l, c_l, c_p],
    'dd_axpy': [c_p, c_f, c_p, c_p, c_l, c_i, c_p],
    'dd_balance_stats': [c_p, c_p, c_p, c_l, c_f, c_i, c_p, c_p, c_z, c_p],
    'dd_replay_gather': [c_p, c_l, c_p, c_i, c_i, c_p, c_i, c_p],
    'dd_copy2d': [c_p, c_l, c_p, c_l, c_l, c_i, c_p],
    'dd_reset_mask': [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_l, c_i, c_p],
    'dd_reset_mask_bwd': [c_p, c_l, c_p, c_l, c_p, c_l, c_l, c_i, c_p],
    'dd_batch_prep': [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_l, c_i, c_p],
    'dd_tanh_fwd': [c_p, c_p, c_i, c_p],
    'dd_tanh_bwd': [c_p, c_p, c_p, c_i, c_f, c_p],
    'dd_imag_wprep': [c_p, c_l, c_i, c_i, c_i, c_p, c_p],
    'dd_imagine_rollout_supported': [c_i] * 9,
    'dd_imag_set_rows': [c_i],
    'dd_imag_wprep_t': [c_p, c_l, c_i, c_i, c_p, c_p],
    'dd_imagine_rollout_bwd': [c_i] * 7 + [c_f] + [ctypes.POINTER(c_p), c_i, c_p],
    'dd_imagine_rollout_fwd': [c_i] * 10 + [c_f] * 3 + [ctypes.POINTER(c_p), c_i, c_p],
    'dd_imagine_rollout_oh_fwd': [c_i] * 11 + [c_f] * 2 + [ctypes.POINTER(c_p), c_i, c_p],
    'dd_stream_create': [ctypes.POINTER(c_p)],
    'dd_stream_destroy': [c_p],
    'dd_graph_capture_begin': [c_p],
    'dd_graph_capture_end': [c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_i)],
    'dd_graph_launch': [c_p, c_p],
    'dd_graph_destroy': [c_p],
    'dd_install_crash_handler': [],
}

EXPORTS = sorted(list(_SIGS) + ['dd_version', 'dd_last_error'])
ABI_VERSION = 11   # include/daydreamer_hip.h DD_ABI_VERSION


def load_library():
  """Load the shared library and declare prototypes.  Raises if missing."""
  global _lib
  if _lib is not None:
    return _lib
  if not _LIB_PATH.exists():
    raise RuntimeError(
        f'{_LIB_PATH} not found: build it with __graft_entry__.build() '
        f'(make -C daydreamer_amd/csrc).  There is no fallback path.')
  lib = ctypes.CDLL(str(_LIB_PATH))
  for name, args in _SIGS.items():
    fn = getattr(lib, name)
    fn.argtypes = args
    fn.restype = c_i
  lib.dd_version.restype = c_i
  lib.dd_last_error.restype = ctypes.c_char_p
  if lib.dd_version() != ABI_VERSION:   # a stale build: signatures below may not match
    raise RuntimeError(f'{_LIB_PATH} has ABI version {lib.dd_version()}, this binding needs '
                       f'{ABI_VERSION}: rebuild it (make -C daydreamer_amd/csrc)')
  _lib = lib
  return lib


def _ptr(t):
  return 0 if t is None else t.data_ptr()


def _mat(t):
  """(ptr, ld) of a row-major 2-D view."""
  assert t.dim() == 2, t.shape
  assert t.shape[1] == 1 or t.stride(1) == 1, (t.shape, t.stride())
  ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
  return t.data_ptr(), ld


def _vec(t):
  """(ptr, stride) of a 1-D view."""
  assert t.dim() == 1, t.shape
  return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else 1)


def onehot_sample_host(x, u, G, C, unimix, mode=0):
  """Host twin of `HipOps.stats_fwd`'s draw (dd_onehot_sample_host): x [rows, G*C] float32 and
  u [rows, G] float32 CPU tensors -> (index int32 [rows, G], stoch [rows, G*C], logit).  Runs on
  the host cores from the kernel's own source, so indices equal the device's bit for bit."""
  lib = load_library()
  x = x.detach().to('cpu', torch.float32).contiguous()
  rows = x.shape[0]
  assert x.shape == (rows, G * C)
  if u is not None:
    u = u.detach().to('cpu', torch.float32).contiguous()
    assert u.shape == (rows, G)
  idx = torch.zeros(rows, G, dtype=torch.int32)
  stoch, logit = torch.zeros(rows, G * C), torch.zeros(rows, G * C)
  rc = lib.dd_onehot_sample_host(
      x.data_ptr(), G * C, _ptr(u), G, logit.data_ptr(), G * C, stoch.data_ptr(), G * C,
      idx.data_ptr(), G, rows, G, C, unimix, mode)
  if rc != 0:
    raise RuntimeError(f'dd_onehot_sample_host failed ({rc}): {lib.dd_last_error().decode()}')
  return idx, stoch, logit


# int32 words of the fused observe scans' `sync2` buffer (csrc/scan.hip: the launches memset 512
# row-block counters at word 576)
SCAN_SYNC_WORDS = 576 + 512


# Buffers with columns that hold values exact in ONE bfloat16 plane (the one-hot `stoch` columns of
# the feature matrices): (weakref to the owning tensor, row length in floats, lo, hi).  A
# contraction whose operand is a view into such a buffer with the buffer's row length as its
# leading dimension goes through dd_gemm_f32_x, which leaves out the plane products that are
# exactly zero (bit-identical result).  The owner being alive guarantees that the address range is
# still that buffer; dead entries are dropped on lookup.
_EXACT = []


def mark_exact(t, lo, hi):
  """Declare that columns [lo, hi) of the last axis of the contiguous tensor `t` only ever hold
  values exact in bfloat16 (one-hot classes) whenever a contraction reads it."""
  import weakref
  assert t.is_contiguous() and 0 <= lo < hi <= t.shape[-1]
  _EXACT.append((weakref.ref(t), t.shape[-1], lo, hi))


def _exact_cols(ptr, ld, ncols):
  """Exact column range of the 2-D view (ptr, leading dimension ld, ncols columns), or (0, 0)."""
  dead = False
  out = (0, 0)
  for ref, row, lo, hi in _EXACT:
    t = ref()
    if t is None:
      dead = True
      continue
    base = t.data_ptr()
    if ld == row and base <= ptr < base + 4 * t.numel():
      col0 = ((ptr - base) // 4) % row
      if col0 + ncols <= row:
        a, b = max(lo - col0, 0), min(hi - col0, ncols)
        if b > a:
          out = (a, b)
      break
  if dead:
    _EXACT[:] = [e for e in _EXACT if e[0]() is not None]
  return out


class Slabs:
  """Deferred split-K partial sums sitting in a HipOps workspace (dd_gemm_f32 `deferred`).
  `beta` / `bias` are what the CONSUMER must apply to C together with the sum.  They are the
  caller's values except on HipOps.gemm's K-peel path (ragged contraction axis): there the
  remainder call has already folded beta * C (and the bulk call carries the bias), so the handle
  says beta = 1.0 whatever the caller passed."""

  def __init__(self, n, M, N, beta, bias):
    self.n, self.M, self.N, self.beta, self.bias = n, M, N, beta, bias


class HipOps:

  name = 'hip'
  SCAN_SYNC_WORDS = SCAN_SYNC_WORDS

  def __init__(self, device='cuda:0', ws_bytes=2048 << 20):
    if not torch.cuda.is_available():
      raise RuntimeError('HipOps needs a visible MI355X (no CPU fallback).')
    self.lib = load_library()
    self.device = torch.device(device)
    self.ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=self.device)
    self.ws_bytes = ws_bytes
    self.trace = None  # list of (label, flops, start_event, end_event) when profiling

  def mark_exact(self, t, lo, hi):
    """Columns [lo, hi) of `t`'s rows hold one-hot classes whenever a contraction reads them
    (module-level registry: every launch context of the process sees it)."""
    mark_exact(t, lo, hi)

  def set_gemm_mode(self, mode):
    """Arithmetic of all contractions of the process (dd_gemm_set_mode): 6 exact-split fp32
    (default), 0 native fp32 MFMA, 1 bf16 inputs (reduced precision, opt-in)."""
    prev = self.lib.dd_gemm_set_mode(int(mode))
    if prev < 0:
      raise RuntimeError(f'dd_gemm_set_mode({mode}): {self.lib.dd_last_error().decode()}')
    return prev

  def _traced(self, label, flops, fn):
    """Bracket one contraction launch with HIP events on the launch stream
    (bench.py's live roofline measurement)."""
    if self.trace is None:
      return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(self.device))
    rc = fn()
    e1.record(torch.cuda.current_stream(self.device))
    self.trace.append((label, flops, e0, e1))
    return rc

  @property
  def stream(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  def _check(self, rc, name):
    if rc != 0:
      raise RuntimeError(
          f'{name} failed ({rc}): {self.lib.dd_last_error().decode()}')

  # ---- contractions ---------------------------------------------------------

  def gemm(self, A, B, C, ta=False, tb=False, alpha=1.0, beta=0.0, bias=None, defer=False):
    """defer=True: if the GEMM splits K, skip its reduce pass and return a `Slabs`
    handle for the consumer (ln_act_fwd / gru_fwd / stats_fwd / ln_act_bwd `pre=`), which
    must be the next operation on this launch context; otherwise (and always with
    defer=False) C is complete and None is returned."""
    M, N = C.shape
    K = A.shape[0] if ta else A.shape[1]
    assert (A.shape[1] if ta else A.shape[0]) == M, (A.shape, C.shape, ta)
    assert (B.shape == (N, K)) if tb else (B.shape == (K, N)), (B.shape, K, N, tb)
    a, lda = _mat(A)
    b, ldb = _mat(B)
    c, ldc = _mat(C)
    # A ragged float4 axis (a multiple-of-four rule of the branch-free 16-byte loaders that only
    # the LAST 1-3 elements break: K = stoch + action = 1030 of the one-hot-action configs) would put
    # the whole contraction on the bounds-checked scalar loaders.  Peel the remainder off instead:
    # the 1-3 leftover k (or rows) as their own small call, the multiple-of-four bulk on the fast path.
    if a % 16 == 0 and b % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0:
      if ta and M % 4 and M >= 256:               # rows of C past the last multiple of four
        M4 = M & ~3
        self.gemm(A[:, M4:], B, C[M4:], ta, tb, alpha, beta, bias)
        self.gemm(A[:, :M4], B, C[:M4], ta, tb, alpha, beta, bias)
        return None
      if (not ta or tb) and K % 4 and K >= 64 and (not ta or M % 4 == 0) and (tb or N % 4 == 0):
        K4 = K & ~3
        self.gemm(A[K4:] if ta else A[:, K4:], B[:, K4:] if tb else B[K4:], C, ta, tb, alpha, beta)
        return self.gemm(A[:K4] if ta else A[:, :K4], B[:, :K4] if tb else B[:K4], C, ta, tb,
                         alpha, 1.0, bias, defer)
    # in deferred mode bias / beta are applied by the consumer together with the sum
    flag = ctypes.c_int(0) if (defer and alpha == 1.0) else None
    xa = _exact_cols(a, lda, A.shape[1]) if _EXACT else (0, 0)
    xb = _exact_cols(b, ldb, B.shape[1]) if _EXACT and not tb else (0, 0)
    if xa[1] or xb[1]:
      self._check(self._traced(f'gemm {M}x{N}x{K} ta{int(ta)} tb{int(tb)} B{4 * (M * K + K * N + M * N)}', 2.0 * M * N * K, lambda: self.lib.dd_gemm_f32_x(
          a, b, c, M, N, K, lda, ldb, ldc, int(ta), int(tb), alpha, beta,
          _ptr(bias), self.ws.data_ptr(), self.ws_bytes,
          ctypes.byref(flag) if flag is not None else None, xa[0], xa[1], xb[0], xb[1],
          self.stream)), 'dd_gemm_f32_x')
    else:
      self._check(self._traced(f'gemm {M}x{N}x{K} ta{int(ta)} tb{int(tb)} B{4 * (M * K + K * N + M * N)}', 2.0 * M * N * K, lambda: self.lib.dd_gemm_f32(
          a, b, c, M, N, K, lda, ldb, ldc, int(ta), int(tb), alpha, beta,
          _ptr(bias), self.ws.data_ptr(), self.ws_bytes,
          ctypes.byref(flag) if flag is not None else None, self.stream)), 'dd_gemm_f32')
    if flag is not None and flag.value > 0:
      return Slabs(flag.value, M, N, beta, bias)
    return None

  def _pre(self, pre, rows, cols):
    """(slabs, n_slabs, beta_pre, bias_pre) arguments of a consumer."""
    if pre is None:
      return 0, 0, 0.0, 0
    assert (pre.M, pre.N) == (rows, cols), ((pre.M, pre.N), (rows, cols))
    return self.ws.data_ptr(), pre.n, pre.beta, _ptr(pre.bias)

  def conv_down(self, big, w, bias, small, k, in_scale=1.0):
    n, hb, wb, cb = big.shape
    n2, hs, ws, cs = small.shape
    assert n == n2 and big.is_contiguous() and small.is_contiguous()
    assert tuple(w.shape) == (k, k, cb, cs) and w.is_contiguous()
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    self._check(self._traced(f'conv_down n{n} {hb}x{cb}->{hs}x{cs} k{k} B{big.numel() * big.element_size() + 4 * (small.numel() + w.numel())}', fl, lambda: self.lib.dd_conv2d_s2_down(
        big.data_ptr(), int(big.dtype == torch.uint8), w.data_ptr(), _ptr(bias),
        small.data_ptr(), n, hb, wb, cb, hs, ws, cs, k, in_scale,
        self.ws.data_ptr(), self.ws_bytes, self.stream)), 'dd_conv2d_s2_down')

  def conv_up(self, small, w, bias, big, k):
    n, hs, ws, cs = small.shape
    n2, hb, wb, cb = big.shape
    assert n == n2 and big.is_contiguous() and small.is_contiguous()
    assert tuple(w.shape) == (k, k, cb, cs) and w.is_contiguous()
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    self._check(self._traced(f'conv_up n{n} {hs}x{cs}->{hb}x{cb} k{k} B{4 * (big.numel() + small.numel() + w.numel())}', fl, lambda: self.lib.dd_conv2d_s2_up(
        small.data_ptr(), w.data_ptr(), _ptr(bias), big.data_ptr(),
        n, hs, ws, cs, hb, wb, cb, k, self.ws.data_ptr(), self.ws_bytes,
        self.stream)), 'dd_conv2d_s2_up')

  def conv_wgrad(self, big, small, dw, k, in_scale=1.0, beta=0.0):
    n, hb, wb, cb = big.shape
    n2, hs, ws, cs = small.shape
    assert n == n2 and big.is_contiguous() and small.is_contiguous()
    assert tuple(dw.shape) == (k, k, cb, cs) and dw.is_contiguous()
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    self._check(self._traced(f'conv_wgrad n{n} {hb}x{cb},{hs}x{cs} k{k} B{big.numel() * big.element_size() + 4 * (small.numel() + dw.numel())}', fl, lambda: self.lib.dd_conv2d_s2_wgrad(
        big.data_ptr(), int(big.dtype == torch.uint8), small.data_ptr(),
        dw.data_ptr(), n, hb, wb, cb, hs, ws, cs, k, in_scale, beta,
        self.ws.data_ptr(), self.ws_bytes, self.stream)), 'dd_conv2d_s2_wgrad')

  def conv_down_ln(self, big, w, bias, gamma, beta_ln, z, out, stats, k, in_scale=1.0):
    """conv_down + ln_act_fwd (LayerNorm + ELU) of an image-side layer as one pass where the
    geometry is covered (dd_conv2d_s2_down_ln), else the two launches: writes z (pre-norm), out
    and stats [pixels, 2] either way."""
    n, hb, wb, cb = big.shape
    n2, hs, ws, cs = z.shape
    assert n == n2 and big.is_contiguous() and z.is_contiguous() and out.is_contiguous() and stats.is_contiguous()
    assert tuple(w.shape) == (k, k, cb, cs) and w.is_contiguous() and tuple(out.shape) == tuple(z.shape)
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    rc = self._traced(f'conv_down n{n} {hb}x{cb}->{hs}x{cs} k{k} B{big.numel() * big.element_size() + 4 * (2 * z.numel() + w.numel())}', fl, lambda: self.lib.dd_conv2d_s2_down_ln(
        big.data_ptr(), int(big.dtype == torch.uint8), w.data_ptr(), _ptr(bias), gamma.data_ptr(), beta_ln.data_ptr(),
        z.data_ptr(), out.data_ptr(), stats.data_ptr(), n, hb, wb, cb, hs, ws, cs, k, in_scale,
        self.ws.data_ptr(), self.ws_bytes, self.stream))
    if rc == 1:   # geometry not covered
      if self.trace:
        self.trace.pop()
      self.conv_down(big, w, bias, z, k, in_scale)
      return self.ln_act_fwd(z.view(-1, cs), gamma, beta_ln, out.view(-1, cs), stats, True)
    self._check(rc, 'dd_conv2d_s2_down_ln')

  def conv_down_lnbwd(self, big, w, z, stats, gamma, beta_ln, dout, dz, dgamma, dbeta, dbias, k):
    """Data gradient of an image-side transposed convolution (big = gradient of the image, float)
    followed by the LayerNorm + ELU backward of the layer in front of it, as one pass where the
    geometry is covered (dd_conv2d_s2_down_lnbwd: `dout` stays untouched), else conv_down into
    `dout` + ln_act_bwd.  Writes dz, dgamma, dbeta, dbias either way."""
    n, hb, wb, cb = big.shape
    n2, hs, ws, cs = dz.shape
    assert n == n2 and big.is_contiguous() and dz.is_contiguous() and z.is_contiguous() and stats.is_contiguous()
    assert tuple(w.shape) == (k, k, cb, cs) and w.is_contiguous() and big.dtype == torch.float32
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    rc = self._traced(f'conv_down n{n} {hb}x{cb}->{hs}x{cs} k{k} B{4 * (big.numel() + 2 * dz.numel() + w.numel())}', fl, lambda: self.lib.dd_conv2d_s2_down_lnbwd(
        big.data_ptr(), w.data_ptr(), z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta_ln.data_ptr(),
        dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dbias.data_ptr(), 0, n, hb, wb, cb, hs, ws, cs, k,
        self.ws.data_ptr(), self.ws_bytes, self.stream))
    if rc == 1:   # geometry not covered
      if self.trace:
        self.trace.pop()
      self.conv_down(big, w, None, dout, k)
      return self.ln_act_bwd(dout.view(-1, cs), z.view(-1, cs), None, stats, gamma, dz.view(-1, cs), dgamma, dbeta,
                             False, True, dbias, beta=beta_ln)
    self._check(rc, 'dd_conv2d_s2_down_lnbwd')

  def conv_wgrad_ln(self, big, dout, z, stats, gamma, beta_ln, dz, dw, dgamma, dbeta, dbias, k, in_scale=1.0):
    """Filter gradient of an image-side Conv2D + LayerNorm + ELU layer from the gradient at the
    layer OUTPUT: ln_act_bwd (activation recomputed from z) + conv_wgrad as one pass where the
    geometry is covered (dd_conv2d_s2_wgrad_ln: dz never travels through HBM and `dz` stays
    untouched), else the two launches (which write `dz`)."""
    n, hb, wb, cb = big.shape
    n2, hs, ws, cs = dout.shape
    assert n == n2 and big.is_contiguous() and dout.is_contiguous() and z.is_contiguous() and stats.is_contiguous()
    assert tuple(dw.shape) == (k, k, cb, cs) and dw.is_contiguous() and tuple(z.shape) == tuple(dout.shape)
    fl = 2.0 * n * hs * ws * k * k * cb * cs
    rc = self._traced(f'conv_wgrad n{n} {hb}x{cb},{hs}x{cs} k{k} B{big.numel() * big.element_size() + 4 * (2 * dout.numel() + dw.numel())}', fl, lambda: self.lib.dd_conv2d_s2_wgrad_ln(
        big.data_ptr(), int(big.dtype == torch.uint8), in_scale, dout.data_ptr(), z.data_ptr(), stats.data_ptr(),
        gamma.data_ptr(), beta_ln.data_ptr(), dw.data_ptr(), 0.0, dgamma.data_ptr(), dbeta.data_ptr(),
        dbias.data_ptr(), 0, n, hb, wb, cb, hs, ws, cs, k, self.ws.data_ptr(), self.ws_bytes, self.stream))
    if rc == 1:   # geometry not covered: nothing was launched
      if self.trace:
        self.trace.pop()
      self.ln_act_bwd(dout.view(-1, cs), z.view(-1, cs), None, stats, gamma, dz.view(-1, cs), dgamma, dbeta,
                      False, True, dbias, beta=beta_ln)
      return self.conv_wgrad(big, dz, dw, k, in_scale)
    self._check(rc, 'dd_conv2d_s2_wgrad_ln')

  # stride-1 SAME convolutions + 2x2 pooling / repetition (residual encoder / decoder)

  def conv_same(self, x, w, bias, y, k, in_scale=1.0, alpha=1.0, beta=0.0):
    n, h, wd, cin = x.shape
    cout = y.shape[3]
    assert tuple(y.shape) == (n, h, wd, cout) and x.is_contiguous() and y.is_contiguous()
    assert tuple(w.shape) == (k, k, cin, cout) and w.is_contiguous()
    fl = 2.0 * n * h * wd * k * k * cin * cout
    self._check(self._traced(f'conv_same n{n} {h}x{cin}->{cout} k{k} B{x.numel() * x.element_size() + 4 * (w.numel() + y.numel())}', fl, lambda: self.lib.dd_conv2d_same(
        x.data_ptr(), int(x.dtype == torch.uint8), w.data_ptr(),
        bias.data_ptr() if bias is not None else None, y.data_ptr(), n, h, wd, cin, cout, k,
        in_scale, alpha, beta, self.ws.data_ptr(), self.ws_bytes, self.stream)), 'dd_conv2d_same')

  def conv_same_bwd(self, dy, w, dx, k, alpha=1.0, beta=0.0):
    n, h, wd, cout = dy.shape
    cin = dx.shape[3]
    assert tuple(dx.shape) == (n, h, wd, cin) and dx.is_contiguous() and dy.is_contiguous()
    assert tuple(w.shape) == (k, k, cin, cout) and w.is_contiguous()
    fl = 2.0 * n * h * wd * k * k * cin * cout
    self._check(self._traced(f'conv_same_bwd n{n} {h}x{cout}->{cin} k{k} B{4 * (dy.numel() + w.numel() + dx.numel())}', fl, lambda: self.lib.dd_conv2d_same_bwd_data(
        dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, wd, cin, cout, k, alpha, beta,
        self.ws.data_ptr(), self.ws_bytes, self.stream)), 'dd_conv2d_same_bwd_data')

  def conv_same_wgrad(self, x, dy, dw, k, in_scale=1.0, alpha=1.0, beta=0.0):
    n, h, wd, cin = x.shape
    cout = dy.shape[3]
    assert tuple(dy.shape) == (n, h, wd, cout) and x.is_contiguous() and dy.is_contiguous()
    assert tuple(dw.shape) == (k, k, cin, cout) and dw.is_contiguous()
    fl = 2.0 * n * h * wd * k * k * cin * cout
    self._check(self._traced(f'conv_same_wgrad n{n} {h}x{cin},{cout} k{k} B{x.numel() * x.element_size() + 4 * (dy.numel() + dw.numel())}', fl, lambda: self.lib.dd_conv2d_same_wgrad(
        x.data_ptr(), int(x.dtype == torch.uint8), dy.data_ptr(), dw.data_ptr(), n, h, wd, cin,
        cout, k, in_scale, alpha, beta, self.ws.data_ptr(), self.ws_bytes, self.stream)),
        'dd_conv2d_same_wgrad')

  def pool2(self, x, y, scale=0.25):
    n, ho, wo, c = y.shape
    assert tuple(x.shape) == (n, 2 * ho, 2 * wo, c) and x.is_contiguous() and y.is_contiguous()
    self._check(self.lib.dd_pool2(x.data_ptr(), y.data_ptr(), n, ho, wo, c, scale, self.stream),
                'dd_pool2')

  def repeat2(self, x, y, scale=1.0, beta=0.0):
    n, hi, wi, c = x.shape
    assert tuple(y.shape) == (n, 2 * hi, 2 * wi, c) and x.is_contiguous() and y.is_contiguous()
    self._check(self.lib.dd_repeat2(x.data_ptr(), y.data_ptr(), n, hi, wi, c, scale, beta,
                                    self.stream), 'dd_repeat2')

  # ---- LayerNorm / GRU -------------------------------------------------------

  def ln_act_fwd(self, z, gamma, beta, out, stats, act=True, pre=None, head=None):
    """head = (w [C, 1], b [1], y [rows, 1]): the one-unit output layer behind this layer, folded in
    (dd_ln_act_fwd_head: y = out @ w + b)."""
    rows, C = z.shape
    zp, ldz = _mat(z)
    op, ldo = _mat(out)
    if head is not None:
      w, b, y = head
      assert w.is_contiguous() and w.numel() == C and y.is_contiguous() and y.numel() == rows
      self._check(self.lib.dd_ln_act_fwd_head(
          zp, ldz, gamma.data_ptr(), beta.data_ptr(), op, ldo, *_mat(stats),
          rows, C, int(act), *self._pre(pre, rows, C), w.data_ptr(), _ptr(b), y.data_ptr(), self.stream),
          'dd_ln_act_fwd_head')
      return
    self._check(self.lib.dd_ln_act_fwd(
        zp, ldz, gamma.data_ptr(), beta.data_ptr(), op, ldo, *_mat(stats),
        rows, C, int(act), *self._pre(pre, rows, C), self.stream), 'dd_ln_act_fwd')

  def ln_act_bwd_head(self, head_dy, head_w, z, out, stats, gamma, dz, dgamma=None, dbeta=None,
                      accumulate=False, act=True, dbias_pre=None, beta=None):
    """ln_act_bwd of a layer whose output gradient is head_dy [rows, 1] x head_w [C, 1] (the
    one-unit output layer behind it): formed in registers, no `dout` tensor."""
    rows, C = z.shape
    assert head_dy.is_contiguous() and head_dy.numel() == rows and head_w.is_contiguous() and head_w.numel() == C
    zp, ldz = _mat(z)
    op, ldo = _mat(out) if (beta is None or not act) else (0, 0)
    dzp, lddz = _mat(dz)
    self._check(self.lib.dd_ln_act_bwd_head(
        head_dy.data_ptr(), head_w.data_ptr(), zp, ldz, op, ldo, *_mat(stats), gamma.data_ptr(), _ptr(beta),
        dzp, lddz, _ptr(dgamma), _ptr(dbeta), _ptr(dbias_pre), int(accumulate), rows, C, int(act),
        self.ws.data_ptr(), self.ws_bytes, self.stream), 'dd_ln_act_bwd_head')

  def ln_act_bwd(self, dout, z, out, stats, gamma, dz, dgamma=None,
                 dbeta=None, accumulate=False, act=True, dbias_pre=None, pre=None, beta=None):
    """beta (the LayerNorm offset): the activation's derivative is recomputed from z instead of
    read from `out` - bit-identical, one tensor less through HBM."""
    rows, C = z.shape
    dp, ldd = _mat(dout)
    zp, ldz = _mat(z)
    op, ldo = _mat(out) if (beta is None or not act) else (0, 0)
    dzp, lddz = _mat(dz)
    self._check(self.lib.dd_ln_act_bwd(
        dp, ldd, zp, ldz, op, ldo, *_mat(stats), gamma.data_ptr(), _ptr(beta), dzp,
        lddz, _ptr(dgamma), _ptr(dbeta), _ptr(dbias_pre), int(accumulate),
        rows, C, int(act),
        self.ws.data_ptr(), self.ws_bytes, *self._pre(pre, rows, C)[:3], self.stream),
        'dd_ln_act_bwd')

  def ln_param_grad(self, dout, z, out, stats, dgamma, dbeta,
                    accumulate=False, act=True):
    rows, C = z.shape
    dp, ldd = _mat(dout)
    zp, ldz = _mat(z)
    op, ldo = _mat(out) if out is not None else (0, 0)
    self._check(self.lib.dd_ln_param_grad(
        dp, ldd, zp, ldz, op, ldo, *_mat(stats), dgamma.data_ptr(),
        dbeta.data_ptr(), int(accumulate), rows, C, int(act),
        self.ws.data_ptr(), self.ws_bytes, self.stream), 'dd_ln_param_grad')

  def col_sum(self, x, out, beta=0.0):
    rows, C = x.shape
    xp, ldx = _mat(x)
    self._check(self.lib.dd_col_sum(
        xp, ldx, out.data_ptr(), beta, rows, C, self.ws.data_ptr(),
        self.ws_bytes, self.stream), 'dd_col_sum')

  def gru_fwd(self, z3, gamma, beta, h, hn, stats, pre=None):
    rows, D = h.shape
    zp, ldz = _mat(z3)
    hp, ldh = _mat(h)
    np_, ldn = _mat(hn)
    self._check(self.lib.dd_gru_cell_fwd(
        zp, ldz, gamma.data_ptr(), beta.data_ptr(), hp, ldh, np_, ldn,
        *_mat(stats), rows, D, *self._pre(pre, rows, 3 * D)[:3], self.stream),
        'dd_gru_cell_fwd')

  def gru_bwd(self, dhn, z3, stats, gamma, beta, h, dz3, dh, dy3, zero=None):
    rows, D = h.shape
    a, lda = _mat(dhn)
    zp, ldz = _mat(z3)
    hp, ldh = _mat(h)
    dzp, lddz = _mat(dz3)
    dhp, lddh = _mat(dh)
    dyp, lddy = _mat(dy3)
    zxp, ldzx = _mat(zero) if zero is not None else (0, 0)
    U = zero.shape[1] if zero is not None else 0
    self._check(self.lib.dd_gru_cell_bwd(
        a, lda, zp, ldz, *_mat(stats), gamma.data_ptr(), beta.data_ptr(),
        hp, ldh, dzp, lddz, dhp, lddh, dyp, lddy, zxp, ldzx, U, rows, D,
        self.stream),
        'dd_gru_cell_bwd')

  # ---- fused observe scan ----------------------------------------------------------

  def observe_scan_supported(self, B, D, U, G, C, A):
    return bool(self.lib.dd_observe_scan_supported(B, D, U, G, C, A))

  def scan_wprep(self, W, planes, Kp):
    """Weight cache of the fused scan: W [K, N] fp32 -> three bf16 planes of N x Kp values in
    fragment-major order (include/daydreamer_hip.h; N % 16 == 0, Kp % 128 == 0)."""
    K, N = W.shape
    assert W.stride(1) == 1 and planes.dtype == torch.int16 and planes.numel() == 3 * N * Kp
    self._check(self.lib.dd_scan_wprep(W.data_ptr(), W.stride(0), K, N, Kp, planes.data_ptr(),
                                       self.stream), 'dd_scan_wprep')

  def observe_scan_fwd(self, B, T, D, U, G, C, A, use_carry, unimix, first, carry, init_deter,
                       init_stoch, u_post, wts, vecs, bufs, w_in, idx_ws, sync2):
    """wts: 4 plane caches; vecs: g1, b1, gg, bg, g3, b3, bias4; bufs: xin, z1, st1, gin, z3, gst,
    post, zo, xo, st3, xq, post_logit (all contiguous, rows b*T + t)."""
    for t in bufs:
      assert t.is_contiguous()
    # barrier words: counter, error word, debug stamps at [0, 576), 512 row-block counters behind
    # them - the launch clears the latter (ABI 6: the buffer grew from 2 words)
    assert sync2.dtype == torch.int32 and sync2.numel() >= SCAN_SYNC_WORDS, sync2.shape
    self._check(self.lib.dd_observe_scan_fwd(
        B, T, D, U, G, C, A, int(use_carry), unimix, first.data_ptr(), _ptr(carry),
        init_deter.data_ptr(), init_stoch.data_ptr(), u_post.data_ptr(),
        *[w.data_ptr() for w in wts], *[v.data_ptr() for v in vecs],
        *[t.data_ptr() for t in bufs], w_in.data_ptr(), idx_ws.data_ptr(), sync2.data_ptr(),
        self.stream), 'dd_observe_scan_fwd')

  def observe_scan_bwd_supported(self, B, D, U, G, C):
    return bool(self.lib.dd_observe_scan_bwd_supported(B, D, U, G, C))

  def scan_wprep_rows(self, W, planes):
    """Weight cache of the fused reverse scan: W [N, K] fp32 -> three bf16 planes, fragment-major,
    operand column n = row n of W (N % 16 == 0, K % 128 == 0)."""
    N, K = W.shape
    assert W.stride(1) == 1 and planes.dtype == torch.int16 and planes.numel() == 3 * N * K
    self._check(self.lib.dd_scan_wprep_rows(W.data_ptr(), W.stride(0), N, K, planes.data_ptr(),
                                            self.stream), 'dd_scan_wprep_rows')

  def observe_scan_bwd(self, B, T, D, U, G, C, flags, unimix, first, acts, dlogit, wts, vecs, grads,
                       sync2):
    """acts: xq, zo, xo, st3, z3, gst, gin, z1, st1; wts: 4 row-plane caches (obs_stats,
    obs_out_h, gru, img_in_s); vecs: g3, gg, bg, g1; grads: dfeat, dxq, dxo, dzo, dz3, dy3, dgin,
    dz1, dxs (all contiguous, rows b*T + t)."""
    for t in list(acts) + list(grads) + [dlogit]:
      assert t.is_contiguous()
    assert sync2.dtype == torch.int32 and sync2.numel() >= SCAN_SYNC_WORDS, sync2.shape
    self._check(self.lib.dd_observe_scan_bwd(
        B, T, D, U, G, C, int(flags), unimix, first.data_ptr(), *[t.data_ptr() for t in acts],
        dlogit.data_ptr(), *[w.data_ptr() for w in wts], *[v.data_ptr() for v in vecs],
        *[t.data_ptr() for t in grads], sync2.data_ptr(), self.stream), 'dd_observe_scan_bwd')

  # ---- fused imagination rollout ------------------------------------------------------

  def imagine_rollout_supported(self, D, U, G, C, A, actor_units, actor_layers, prior_layers, discrete):
    return bool(self.lib.dd_imagine_rollout_supported(D, U, G, C, A, actor_units, actor_layers,
                                                      prior_layers, int(bool(discrete))))

  def stamp(self, buf, slot):
    """buf[9 * slot] counts, buf[9 * slot + 1 + i % 8] (int64 device tensor) = the device wall clock
    (100 MHz) when the stream got here the i-th time."""
    self._check(self.lib.dd_stamp(buf.data_ptr() + 8 * 9 * slot, self.stream), 'dd_stamp')

  def imag_set_rows(self, rows):
    """Rows of the imagination batch per workgroup of the fused forward rollout (process-wide):
    32 (default) or 16.  Returns the previous value."""
    prev = self.lib.dd_imag_set_rows(int(rows))
    if prev < 0:
      raise RuntimeError(f'dd_imag_set_rows({rows}): {self.lib.dd_last_error().decode()}')
    return prev

  def imag_wprep(self, W, planes, col0=0):
    """Weight cache of the fused rollout: W [K, n] fp32 -> columns col0.. of fragment-major bf16
    planes [Npad/16, K/32, 3, 64, 8] (zero-initialised by the caller where padded)."""
    K, n = W.shape
    assert W.stride(1) == 1 and planes.dtype == torch.int16
    self._check(self.lib.dd_imag_wprep(W.data_ptr(), W.stride(0), K, n, col0, planes.data_ptr(),
                                       self.stream), 'dd_imag_wprep')

  def imag_wprep_t(self, W, planes):
    """Transposed weight cache of the fused reverse rollout: W [n, K] fp32 -> fragment-major bf16
    planes of the operand B[k][col] = W[col][k]  ([ceil(n/16), K/32, 3, 64, 8])."""
    n, K = W.shape
    assert W.stride(1) == 1 and planes.dtype == torch.int16
    self._check(self.lib.dd_imag_wprep_t(W.data_ptr(), W.stride(0), K, n, planes.data_ptr(),
                                         self.stream), 'dd_imag_wprep_t')

  def imagine_rollout_bwd(self, N, H, D, U, G, C, A, unimix, tensors):
    """tensors: the 29 (+ optional time-stamp buffer) device tensors of dd_imagine_rollout_bwd."""
    n = len(tensors)
    assert n in (29, 30)
    arr = (c_p * n)(*[t.data_ptr() for t in tensors])
    S = G * C
    img = (S + A) * U + (D + U) * 3 * D + U * D + 2 * U * U + U * S
    flops = 2.0 * N * H * img
    nbytes = 4 * img + 4 * N * H * ((D + S) + S + 3 * D + 8 * U + D + S + A)
    self._check(self._traced(f'imagine_bwd N{N} H{H} B{nbytes}', flops, lambda: self.lib.dd_imagine_rollout_bwd(
        N, H, D, U, G, C, A, unimix, arr, n, self.stream)), 'dd_imagine_rollout_bwd')

  def imagine_rollout_fwd(self, N, H, D, U, G, C, A, actor_units, unimix, lo, hi, tensors, t0=0, t1=None):
    """tensors: the 65 device tensors of dd_imagine_rollout_fwd, in header order."""
    t1 = H + 1 if t1 is None else t1
    n = len(tensors)
    assert n in (65, 66)   # (66: + time-stamp buffer, tools/imag_time.py)
    for t in tensors:
      assert t.is_contiguous() or t.dim() == 2   # (kernel row slices: contiguous rows)
    arr = (c_p * n)(*[t.data_ptr() for t in tensors])
    # algorithmic work (dense contractions, each once): per row, H + 1 policy
    # evaluations and H img_steps; bytes: weights once, every activation buffer written once
    S, F, AU = G * C, D + G * C, actor_units
    actor = F * AU + 3 * AU * AU + AU * 2 * A
    img = (S + A) * U + (D + U) * 3 * D + U * D + 2 * U * U + U * S
    na, ni = t1 - t0, min(t1, H) - t0    # policy evaluations, img_steps of this launch
    flops = 2.0 * N * (na * actor + ni * img)
    nbytes = 4 * (actor + img) + 4 * N * (na * (F + A + 8 * AU + 2 * A) + ni * (2 * U + 3 * D + 6 * U + S))
    self._check(self._traced(f'imagine N{N} H{H} B{nbytes}', flops, lambda: self.lib.dd_imagine_rollout_fwd(
        N, H, t0, t1, D, U, G, C, A, actor_units, unimix, lo, hi, arr, n, self.stream)),
        'dd_imagine_rollout_fwd')

  def imagine_rollout_oh_fwd(self, N, H, D, U, G, C, A, actor_units, row_width, unimix, actor_unimix,
                             tensors, t0=0, t1=None):
    """tensors: the 64 device tensors of dd_imagine_rollout_oh_fwd, in header order."""
    t1 = H + 1 if t1 is None else t1
    n = len(tensors)
    assert n == 64
    for t in tensors:
      assert t.is_contiguous() or t.dim() == 2
    arr = (c_p * n)(*[t.data_ptr() for t in tensors])
    S, F, AU = G * C, D + G * C, actor_units
    actor = F * AU + 3 * AU * AU + AU * A
    img = (S + A) * U + (D + U) * 3 * D + U * D + 2 * U * U + U * S
    na, ni = t1 - t0, min(t1, H) - t0    # policy evaluations, img_steps of this launch
    flops = 2.0 * N * (na * actor + ni * img)
    nbytes = 4 * (actor + img) + 4 * N * (na * (F + A + 8 * AU + 2 * A) + ni * (2 * U + 3 * D + 6 * U + S))
    self._check(self._traced(f'imagine N{N} H{H} B{nbytes}', flops, lambda: self.lib.dd_imagine_rollout_oh_fwd(
        N, H, t0, t1, D, U, G, C, A, actor_units, row_width, unimix, actor_unimix, arr, n, self.stream)),
        'dd_imagine_rollout_oh_fwd')

  # ---- categorical latent -----------------------------------------------------

  def stats_fwd(self, x, u, logit, stoch, G, C, unimix, mode=0, pre=None):
    rows = x.shape[0]
    xp, ldx = _mat(x)
    up, ldu = _mat(u) if u is not None else (0, 0)
    lp, ldl = _mat(logit)
    sp, lds = _mat(stoch)
    self._check(self.lib.dd_stats_sample_fwd(
        xp, ldx, up, ldu, lp, ldl, sp, lds, rows, G, C, unimix, mode,
        *self._pre(pre, rows, G * C), self.stream), 'dd_stats_sample_fwd')

  def stats_bwd(self, x, dlogit, dstoch, dx, G, C, unimix):
    rows = x.shape[0]
    xp, ldx = _mat(x)
    lp, ldl = _mat(dlogit) if dlogit is not None else (0, 0)
    sp, lds = _mat(dstoch) if dstoch is not None else (0, 0)
    dp, ldd = _mat(dx)
    self._check(self.lib.dd_stats_sample_bwd(
        xp, ldx, lp, ldl, sp, lds, dp, ldd, rows, G, C, unimix, self.stream),
        'dd_stats_sample_bwd')

  def kl_fwd(self, post, prior, kl, ent_post, ent_prior, G, C):
    rows = post.shape[0]
    pp, ldp = _mat(post)
    qp, ldq = _mat(prior)
    self._check(self.lib.dd_cat_kl_fwd(
        pp, ldp, qp, ldq, kl.data_ptr(), ent_post.data_ptr(),
        ent_prior.data_ptr(), rows, G, C, self.stream), 'dd_cat_kl_fwd')

  def kl_bwd(self, post, prior, coef_dev, coef_host, balance, dpost, dprior,
             G, C):
    rows = post.shape[0]
    pp, ldp = _mat(post)
    qp, ldq = _mat(prior)
    dp, lddp = _mat(dpost)
    dq, lddq = _mat(dprior)
    self._check(self.lib.dd_cat_kl_bwd(
        pp, ldp, qp, ldq, _ptr(coef_dev), coef_host, balance, dp, lddp, dq,
        lddq, rows, G, C, self.stream), 'dd_cat_kl_bwd')

  # ---- losses / imagination scalars -------------------------------------------

  def image_loss(self, z, img, loss, dz, coef, c0=0, c1=None):
    rows = z.shape[0]
    P = z[0].numel()
    ctot = z.shape[-1]
    c1 = ctot if c1 is None else c1
    assert z.is_contiguous() and img.is_contiguous() and img.dtype == torch.uint8
    self._check(self.lib.dd_image_loss(
        z.data_ptr(), img.data_ptr(), loss.data_ptr(), dz.data_ptr(), rows, P,
        ctot, c0, c1, coef, self.stream), 'dd_image_loss')

  def video_grid(self, z, img, out, nb, nt, c0, c1, zsb, zst):
    """Report videos (dd_video_grid): z [images, H, W, ctot] pre-sigmoid, image (b, t) = number
    b * zsb + t * zst; img: uint8 truth in the same layout or None; out [nt, (3 | 1) * H, nb * W,
    c1 - c0] float32."""
    H, W, ctot = z.shape[-3:]
    secs = 3 if img is not None else 1
    assert z.is_contiguous() and out.is_contiguous() and out.dtype == torch.float32
    assert tuple(out.shape) == (nt, secs * H, nb * W, c1 - c0), (out.shape, nt, secs, H, nb, W, c0, c1)
    assert img is None or (img.is_contiguous() and img.dtype == torch.uint8 and img.shape[-3:] == z.shape[-3:])
    last = (nb - 1) * zsb + (nt - 1) * zst
    assert last < z.numel() // (H * W * ctot) and (img is None or last < img.numel() // (H * W * ctot))
    self._check(self.lib.dd_video_grid(z.data_ptr(), _ptr(img), out.data_ptr(), nb, nt, H, W, ctot,
                                       c0, c1, zsb, zst, self.stream), 'dd_video_grid')

  def mse_loss(self, pred, tgt, loss, dpred, coef):
    rows, D = pred.shape
    pp, ldp = _mat(pred)
    tp, ldt = _mat(tgt)
    dp, ldd = _mat(dpred)
    self._check(self.lib.dd_mse_loss(
        pp, ldp, tp, ldt, loss.data_ptr(), dp, ldd, rows, D, coef,
        self.stream), 'dd_mse_loss')

  def scalar_loss(self, pred, tgt, loss, dpred, coef, kind):
    n = pred.numel()
    assert pred.is_contiguous() and tgt.is_contiguous()
    self._check(self.lib.dd_scalar_loss(
        pred.data_ptr(), tgt.data_ptr(), loss.data_ptr(), dpred.data_ptr(), n,
        coef, kind, self.stream), 'dd_scalar_loss')

  def normal_head_fwd(self, om, os, eps, act, lo, hi):
    rows, A = om.shape
    a, lda = _mat(om)
    b, ldb = _mat(os)
    e, lde = _mat(eps) if eps is not None else (0, 0)
    c, ldc = _mat(act)
    self._check(self.lib.dd_normal_head_fwd(
        a, lda, b, ldb, e, lde, c, ldc, rows, A, lo, hi, self.stream),
        'dd_normal_head_fwd')

  def normal_head_bwd(self, om, os, eps, dact, w, scale, dom, dos, ent_row,
                      rows_ent, lo, hi, ent_coef, ent_lo, ent_div):
    rows, A = om.shape
    a, lda = _mat(om)
    b, ldb = _mat(os)
    e, lde = _mat(eps)
    d, ldd = _mat(dact) if dact is not None else (0, 0)
    m, ldm = _mat(dom)
    s, lds = _mat(dos)
    self._check(self.lib.dd_normal_head_bwd(
        a, lda, b, ldb, e, lde, d, ldd, _ptr(w), _ptr(scale), m, ldm, s, lds,
        _ptr(ent_row), rows, rows_ent, A, lo, hi, ent_coef, ent_lo, ent_div,
        self.stream), 'dd_normal_head_bwd')

  def action_noise(self, act, noise, amount, discrete):
    rows, A = act.shape
    a, lda = _mat(act)
    n, ldn = _mat(noise)
    self._check(self.lib.dd_action_noise(a, lda, n, ldn, rows, A, amount, int(discrete),
                                         self.stream), 'dd_action_noise')

  def actent_stats(self, os, rows, lo, hi, ent_lo, ent_div, out):
    b, ldb = _mat(os)
    self._check(self.lib.dd_actent_stats(
        b, ldb, rows, os.shape[1], lo, hi, ent_lo, ent_div, out.data_ptr(),
        self.ws.data_ptr(), self.ws_bytes, self.stream), 'dd_actent_stats')

  def imag_returns_fwd(self, rew_raw, val_raw, cont_raw, first_cont, reward,
                       value, cont, weight, ret, H, N, gamma, lam, impl='gve'):
    self._check(self.lib.dd_imag_returns_fwd(
        rew_raw.data_ptr(), val_raw.data_ptr(), cont_raw.data_ptr(),
        first_cont.data_ptr(), reward.data_ptr(), value.data_ptr(),
        _ptr(cont), _ptr(weight), ret.data_ptr(), H, N, gamma, lam,
        {'gve': 0, 'gae': 1}[impl], self.stream), 'dd_imag_returns_fwd')

  def imag_returns_bwd(self, dret, dbase, rew_raw, val_raw, cont_raw, value,
                       ret, d_rew_raw, d_val_raw, d_cont_raw, H, N, gamma, lam):
    self._check(self.lib.dd_imag_returns_bwd(
        dret.data_ptr(), _ptr(dbase), rew_raw.data_ptr(), val_raw.data_ptr(),
        cont_raw.data_ptr(), value.data_ptr(), ret.data_ptr(),
        d_rew_raw.data_ptr(), d_val_raw.data_ptr(), d_cont_raw.data_ptr(), H,
        N, gamma, lam, self.stream), 'dd_imag_returns_bwd')

  def critic_loss(self, out, ret, w, loss, dout, coef):
    self._check(self.lib.dd_critic_loss(
        out.data_ptr(), ret.data_ptr(), w.data_ptr(), loss.data_ptr(),
        dout.data_ptr(), out.numel(), coef, self.stream), 'dd_critic_loss')

  def actor_seed(self, ret, base, w, ent_row, sc, loss, dret, dbase, coef):
    self._check(self.lib.dd_actor_seed(
        ret.data_ptr(), base.data_ptr(), w.data_ptr(), _ptr(ent_row),
        sc.data_ptr(), loss.data_ptr(), dret.data_ptr(), dbase.data_ptr(),
        ret.numel(), coef, self.stream), 'dd_actor_seed')

  def onehot_entropy(self, logit, ent_out, ent_div):
    rows, A = logit.shape
    lp, ldl = _mat(logit)
    self._check(self.lib.dd_onehot_entropy(
        lp, ldl, ent_out.data_ptr(), rows, A, ent_div, self.stream),
        'dd_onehot_entropy')

  def onehot_policy_grad(self, logit, action, ret, base, w, sc, scale, dlogit,
                         loss_pg, loss_ent, rows_grad, coef, ent_div):
    rows, A = logit.shape
    lp, ldl = _mat(logit)
    ap, lda = _mat(action)
    dp, ldd = _mat(dlogit)
    self._check(self.lib.dd_onehot_policy_grad(
        lp, ldl, ap, lda, ret.data_ptr(), base.data_ptr(), w.data_ptr(),
        sc.data_ptr(), scale.data_ptr(), dp, ldd, loss_pg.data_ptr(),
        loss_ent.data_ptr(), rows, rows_grad, A, coef, ent_div, self.stream),
        'dd_onehot_policy_grad')

  def sub(self, a, b, o):
    self._check(self.lib.dd_sub(
        a.data_ptr(), b.data_ptr(), o.data_ptr(), o.numel(), self.stream),
        'dd_sub')

  def symexp(self, x, o):
    assert x.is_contiguous() and o.is_contiguous() and x.numel() == o.numel()
    self._check(self.lib.dd_symexp(x.data_ptr(), o.data_ptr(), o.numel(), self.stream), 'dd_symexp')

  # ---- learner state ---------------------------------------------------------------

  def philox(self, out, outer, inner, cols, inner_global, inner_offset, seed,
             step_dev, site, kind):
    assert out.is_contiguous() and out.numel() == outer * inner * cols
    self._check(self.lib.dd_philox(
        out.data_ptr(), outer, inner, cols, inner_global, inner_offset, seed,
        step_dev.data_ptr(), site, kind, self.stream), 'dd_philox')

  def counter_add(self, counter, v=1):
    self._check(self.lib.dd_counter_add(counter.data_ptr(), v, self.stream),
                'dd_counter_add')

  def reduce_stats(self, x, sums, maxs):
    p, stride = _vec(x)
    self._check(self.lib.dd_reduce_stats(
        p, x.shape[0], stride, sums.data_ptr(), maxs.data_ptr(), self.stream),
        'dd_reduce_stats')

  def reduce_stats_multi(self, items):
    """reduce_stats of several (x, sums, maxs) triples in launches of up to 16 vectors."""
    for i0 in range(0, len(items), 16):
      part = items[i0:i0 + 16]
      k = len(part)
      vec = [_vec(x) for x, _, _ in part]
      xs = (ctypes.c_void_p * k)(*[v[0] for v in vec])
      ns = (ctypes.c_long * k)(*[x.shape[0] for x, _, _ in part])
      st = (ctypes.c_long * k)(*[v[1] for v in vec])
      sm = (ctypes.c_void_p * k)(*[s_.data_ptr() for _, s_, _ in part])
      mx = (ctypes.c_void_p * k)(*[m.data_ptr() for _, _, m in part])
      self._check(self.lib.dd_reduce_stats_multi(k, xs, ns, st, sm, mx, self.stream), 'dd_reduce_stats_multi')

  def autoadapt_update(self, scale, sums, count, target, thres, vel, lo, hi,
                       inverse, impl='mult'):
    self._check(self.lib.dd_autoadapt_update(
        scale.data_ptr(), sums.data_ptr(), scale.numel(), count, target, thres,
        vel, lo, hi, int(inverse), {'mult': 0, 'prop': 1}[impl], self.stream),
        'dd_autoadapt_update')

  def normalize_update(self, state, sums, count, in_scale_dev, decay, maxv,
                       impl, do_update, out):
    self._check(self.lib.dd_normalize_update(
        state.data_ptr(), sums.data_ptr(), count, _ptr(in_scale_dev), decay,
        maxv, impl, int(do_update), out.data_ptr(), self.stream),
        'dd_normalize_update')

  def scalar_mul(self, dst, a, b, c):
    self._check(self.lib.dd_scalar_mul(
        dst.data_ptr(), a.data_ptr(), _ptr(b), c, dst.numel(), self.stream),
        'dd_scalar_mul')

  def axpy(self, x, alpha, alpha_dev, y, accumulate=True):
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    self._check(self.lib.dd_axpy(x.data_ptr(), alpha, _ptr(alpha_dev), y.data_ptr(),
                                 x.numel(), int(accumulate), self.stream), 'dd_axpy')

  def balance_stats(self, out, target, loss, thres, kind, out7):
    assert out.is_contiguous() and target.is_contiguous() and loss.is_contiguous()
    self._check(self.lib.dd_balance_stats(
        out.data_ptr(), target.data_ptr(), loss.data_ptr(), out.numel(), thres, kind,
        out7.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self.stream), 'dd_balance_stats')

  def grad_norm(self, g, opt_state, mixed=False):
    assert opt_state.numel() >= 5
    self._check(self.lib.dd_grad_norm(
        g.data_ptr(), g.numel(), opt_state.data_ptr(), self.ws.data_ptr(),
        self.ws_bytes, int(bool(mixed)), self.stream), 'dd_grad_norm')

  def adam_step(self, p, g, m, v, n_decay, opt_state, lr, wd, eps, b1, b2,
                clip, warmup=0):
    self._check(self.lib.dd_adam_step(
        p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
        n_decay, opt_state.data_ptr(), lr, wd, eps, b1, b2, clip, float(warmup), self.stream),
        'dd_adam_step')

  def fill(self, t, v=0.0):
    assert t.is_contiguous()
    self._check(self.lib.dd_fill(t.data_ptr(), t.numel(), v, self.stream),
                'dd_fill')

  def copy2d(self, src, dst):
    rows, cols = src.shape
    s, lds = _mat(src)
    d, ldd = _mat(dst)
    self._check(self.lib.dd_copy2d(s, lds, d, ldd, rows, cols, self.stream),
                'dd_copy2d')

  def replay_gather(self, ring, starts, out, first_flag=False):
    """out[b, t] = ring[starts[b] + t] over whole rows (any dtype); ring
    [rows, ...], starts int64 [B] on the device, out [B, T, ...]."""
    assert (first_flag or ring.is_contiguous()) and out.is_contiguous() and starts.dtype == torch.int64
    B, T = out.shape[:2]
    row_bytes = out[0, 0].numel() * out.element_size()
    assert first_flag or (ring.dtype == out.dtype and ring[0].numel() * ring.element_size() == row_bytes)
    self._check(self.lib.dd_replay_gather(
        _ptr(ring), row_bytes, starts.data_ptr(), B, T, out.data_ptr(),
        int(first_flag), self.stream), 'dd_replay_gather')

  def reset_mask(self, prev, first, init, out):
    rows, cols = out.shape
    p, ldp = _mat(prev) if prev is not None else (0, 0)
    f, fs = _vec(first)
    o, ldo = _mat(out)
    self._check(self.lib.dd_reset_mask(
        p, ldp, f, fs, _ptr(init), o, ldo, rows, cols, self.stream),
        'dd_reset_mask')

  def reset_mask2(self, prev_a, init_a, out_a, prev_b, init_b, out_b, first):
    rows = out_a.shape[0]
    pa, ldpa = _mat(prev_a) if prev_a is not None else (0, 0)
    pb, ldpb = _mat(prev_b) if prev_b is not None else (0, 0)
    f, fs = _vec(first)
    self._check(self.lib.dd_reset_mask2(
        pa, ldpa, _ptr(init_a), *_mat(out_a), out_a.shape[1],
        pb, ldpb, _ptr(init_b), *_mat(out_b), out_b.shape[1], f, fs, rows,
        self.stream), 'dd_reset_mask2')

  def reset_mask_bwd2(self, dout_a, dprev_a, dout_b, dprev_b, first):
    rows = dout_a.shape[0]
    f, fs = _vec(first)
    self._check(self.lib.dd_reset_mask_bwd2(
        *_mat(dout_a), *_mat(dprev_a), dout_a.shape[1],
        *_mat(dout_b), *_mat(dprev_b), dout_b.shape[1], f, fs, rows,
        self.stream), 'dd_reset_mask_bwd2')

  def reset_mask_bwd(self, dout, first, dprev):
    rows, cols = dout.shape
    d, ldo = _mat(dout)
    f, fs = _vec(first)
    p, ldp = _mat(dprev)
    self._check(self.lib.dd_reset_mask_bwd(
        d, ldo, f, fs, p, ldp, rows, cols, self.stream), 'dd_reset_mask_bwd')

  def batch_prep(self, is_first, is_terminal, action, first_f, cont_f,
                 act_masked):
    n = is_first.numel()
    A = action.shape[-1]
    m, ldm = _mat(act_masked)
    self._check(self.lib.dd_batch_prep(
        is_first.data_ptr(), is_terminal.data_ptr(), action.data_ptr(),
        first_f.data_ptr(), cont_f.data_ptr(), m, ldm, n, A, self.stream),
        'dd_batch_prep')

  def tanh_fwd(self, x, y):
    self._check(self.lib.dd_tanh_fwd(x.data_ptr(), y.data_ptr(), x.numel(),
                                     self.stream), 'dd_tanh_fwd')

  def tanh_bwd(self, x, dy, dx, beta=0.0):
    self._check(self.lib.dd_tanh_bwd(
        x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), beta,
        self.stream), 'dd_tanh_bwd')
