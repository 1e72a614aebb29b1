"""Per-kernel parity: every C-ABI entry point (through daydreamer_amd.hipops)
against its CPU restatement oracle/ref_ops.py on the same seeded inputs.
Tolerances: fp32 contractions 2e-4 relative to the output scale (different
summation order), elementwise 1e-5; integer / one-hot outputs bit-exact."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
  g = torch.Generator().manual_seed(seed)
  return torch.randn(*shape, generator=g) * scale


def close(a, b, rtol=2e-4, atol=None, what=''):
  a = a.detach().cpu().double()
  b = b.detach().cpu().double()
  scale = float(b.abs().max()) + 1e-30
  atol = rtol * scale if atol is None else atol
  err = float((a - b).abs().max())
  assert err <= atol, f'{what}: max err {err:.3e} > {atol:.3e} (scale {scale:.3e})'


def both(hip, ref, fn, tensors, outs):
  """Run fn(ops, *tensors) on both backends; tensors listed in `outs` (indices)
  are compared afterwards."""
  cpu = [t.clone() if torch.is_tensor(t) else t for t in tensors]
  gpu = [t.cuda() if torch.is_tensor(t) else t for t in tensors]
  fn(ref, *cpu)
  fn(hip, *gpu)
  torch.cuda.synchronize()
  return [(gpu[i], cpu[i]) for i in outs]


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (50, 256, 1040, 0, 0), (2500, 512, 1280, 0, 0), (130, 70, 33, 0, 0),
    (200, 300, 1000, 0, 1), (1280, 512, 4000, 1, 0), (7, 64, 50, 1, 0),
    (96, 96, 96, 1, 1), (1, 256, 96, 1, 0), (300, 16, 512, 0, 0),
    (64, 6, 30, 0, 1), (333, 129, 1030, 0, 0)])
def test_gemm(hip, ref, M, N, K, ta, tb):
  A = rnd(*((K, M) if ta else (M, K)), seed=1)
  B = rnd(*((N, K) if tb else (K, N)), seed=2)
  C = rnd(M, N, seed=3)
  bias = rnd(N, seed=4)
  for beta, bs in ((0.0, None), (1.0, bias)):
    def fn(ops, A, B, C, bias):
      ops.gemm(A, B, C, bool(ta), bool(tb), 0.5, beta, bias if bs is not None else None)
    (g, c), = both(hip, ref, fn, [A, B, C, bias], [2])
    close(g, c, what=f'gemm {M}x{N}x{K} ta{ta} tb{tb} beta{beta}')


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (300, 512, 1030, 0, 0), (200, 300, 1030, 0, 1), (1030, 512, 300, 1, 0), (1030, 64, 1031, 1, 1),
    (514, 256, 66, 1, 1), (1250, 512, 1030, 0, 0)])
def test_gemm_peeled_remainder(hip, ref, M, N, K, ta, tb):
  """Operands that are column slices of 16-byte aligned rows with a ragged float4 axis (the
  [stoch | action] columns of the padded trajectory rows, K = 1030): HipOps.gemm peels the last
  1-3 k (or output rows) off into their own small call and runs the bulk on the fast loaders."""
  def padded(r, c, seed):
    return rnd(r, (c + 3) // 4 * 4 + 8, seed=seed)
  A = padded(*((K, M) if ta else (M, K)), seed=1)
  B = padded(*((N, K) if tb else (K, N)), seed=2)
  C, bias = rnd(M, N, seed=3), rnd(N, seed=4)
  ac, bc = (M if ta else K), (K if tb else N)
  for beta, bs in ((0.0, None), (1.0, bias)):
    def fn(ops, A, B, C, bias):
      if ops is hip:
        ops.trace = []
      ops.gemm(A[:, :ac], B[:, :bc], C, bool(ta), bool(tb), 0.5, beta, bias if bs is not None else None)
      if ops is hip:
        assert len(ops.trace) >= 2, 'the remainder was not peeled off'
        ops.trace = None
    (g, c), = both(hip, ref, fn, [A, B, C, bias], [2])
    close(g, c, what=f'gemm {M}x{N}x{K} ta{ta} tb{tb} beta{beta}')


def test_role_separated_loop_is_bit_identical(hip):
  """dd_gemm_set_ws: the role-separated form of the 128x128 loop (four MFMA waves + four staging
  waves, k_mfma_gemm_ws) stages the same LDS image and issues the same products in the same order
  as the product loop: plain GEMMs (all four operand layouts, beta / bias, split-K, K tail), the
  strided convolution, its filter gradient and the banded transposed convolution are bit-identical
  with the variant on and off."""
  lib = hip.lib
  def run_all():
    outs = []
    for (M, N, K, ta, tb) in [(300, 256, 1040, 0, 0), (2500, 512, 1280, 0, 1), (1280, 512, 4000, 1, 0),
                              (257, 130, 2052, 1, 1), (4096, 256, 512, 0, 0)]:
      A = rnd(*((K, M) if ta else (M, K)), seed=1).cuda()
      B = rnd(*((N, K) if tb else (K, N)), seed=2).cuda()
      C = rnd(M, N, seed=3).cuda()
      hip.gemm(A, B, C, bool(ta), bool(tb), 0.5, 1.0, rnd(N, seed=4).cuda())
      outs.append(C)
    n, hb, Cb, hs, Cs, k = 6, 30, 64, 13, 128, 6
    big = rnd(n, hb, hb, Cb, seed=5).cuda()
    w = rnd(k, k, Cb, Cs, seed=6, scale=0.05).cuda()
    small = torch.empty(n, hs, hs, Cs, device='cuda')
    hip.conv_down(big, w, None, small, k)
    back = torch.empty_like(big)
    hip.conv_up(small, w, None, back, k)
    dw = torch.empty_like(w)
    hip.conv_wgrad(big, small, dw, k)
    torch.cuda.synchronize()
    return outs + [small, back, dw]
  prev = lib.dd_gemm_set_ws(0, 256, 256)
  if lib.dd_gemm_set_ws(1, 256, 256) < 0:
    pytest.skip('the default build does not instantiate k_mfma_gemm_ws (make WS=1 does)')
  lib.dd_gemm_set_ws(0, 256, 256)
  try:
    base = run_all()
    lib.dd_gemm_set_ws(1, 256, 256)
    var = run_all()
  finally:
    lib.dd_gemm_set_ws(prev, 1024, 1024)
  for i, (a, b) in enumerate(zip(base, var)):
    assert torch.isfinite(a).all() and torch.equal(a, b), i


@pytest.mark.parametrize('M,N,K,ta,tb,which', [
    (2500, 512, 1280, 0, 0, 'a'), (4000, 512, 1280, 0, 0, 'a'), (50, 512, 1280, 0, 0, 'a'),
    (300, 6400, 1280, 0, 0, 'a'), (700, 512, 1280, 0, 1, 'a'), (2500, 512, 1296, 0, 0, 'a'),
    (1280, 512, 2500, 1, 0, 'a'), (1280, 512, 20000, 1, 0, 'a'), (1040, 256, 2500, 1, 0, 'a'),
    (640, 1280, 2500, 1, 0, 'b'), (300, 1280, 512, 0, 0, 'b')])
def test_gemm_exact_planes_is_bit_identical(hip, M, N, K, ta, tb, which):
  """dd_gemm_f32_x: an operand whose feature columns [256, 1280) are one-hot classes (the `stoch`
  part of [deter | stoch], nets.py:88-97) - the three plane products with its zero middle / low
  planes are left out.  Equality with dd_gemm_f32, not a tolerance: k-range form (A not
  transposed; 128x128, 128x64 and 64x64 tiles, K tail, beta / bias), row-tile form (weight
  gradient A^T dY incl. split-K), and the B-side form (decoder filter gradient dY^T feat)."""
  from daydreamer_amd import hipops
  import numpy as np
  D = 256
  # the stored matrix whose columns are features [deter | stoch | (action)]
  shape = ((K, M) if ta else (M, K)) if which == 'a' else (K, N)
  feat = rnd(*shape, seed=1)
  rows = shape[0]
  ncol = feat.shape[1]
  hi = min(1280, D + (ncol - D) // 32 * 32)
  g = np.random.default_rng(5)
  oh = torch.zeros(rows, (hi - D) // 32, 32)
  idx = torch.from_numpy(g.integers(0, 32, size=(rows, (hi - D) // 32)))
  oh.scatter_(2, idx[..., None], 1.0)
  feat[:, D:hi] = oh.view(rows, -1)
  feat = feat.cuda()
  if which == 'a':
    A, B = feat, rnd(*((N, K) if tb else (K, N)), seed=2).cuda()
  else:
    A, B = rnd(*((K, M) if ta else (M, K)), seed=2).cuda(), feat
  bias = rnd(N, seed=4).cuda()
  outs = []
  for marked in (False, True):
    hipops._EXACT[:] = []
    if marked:
      hipops.mark_exact(feat, D, hi)
    res = []
    for beta, bs in ((0.0, None), (1.0, bias)):
      C = rnd(M, N, seed=3).cuda()
      hip.trace = []
      hip.gemm(A, B, C, bool(ta), bool(tb), 1.0, beta, bs)
      hip.trace = None
      res.append(C)
    outs.append(res)
  hipops._EXACT[:] = []
  opA, opB = (A.T if ta else A), (B.T if tb else B)
  want = opA.double() @ opB.double()
  close(outs[1][0], want.cpu(), what='exact-plane gemm vs float64')
  for a, b in zip(*outs):
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_gemm_bf16_input_mode(hip, ref):
  """dd_gemm_set_mode(1): the opt-in reduced-precision arithmetic (operands rounded to bf16,
  one product, fp32 accumulation).  Its error is that of bf16 inputs - relative 2^-9 per
  operand, averaging down with K - far above the default mode's, far below 1e-2 of the
  output scale; results must equal a float64 product of the bf16-ROUNDED operands to fp32
  accumulation accuracy (i.e. the only loss is the documented input rounding)."""
  prev = hip.set_gemm_mode(1)
  try:
    for (M, N, K, ta, tb) in [(2500, 512, 1280, 0, 0), (200, 300, 1000, 0, 1), (1280, 512, 4000, 1, 0)]:
      A = rnd(*((K, M) if ta else (M, K)), seed=1)
      B = rnd(*((N, K) if tb else (K, N)), seed=2)
      C = torch.zeros(M, N).cuda()
      hip.gemm(A.cuda(), B.cuda(), C, bool(ta), bool(tb))
      torch.cuda.synchronize()
      opA, opB = (A.T if ta else A), (B.T if tb else B)
      exact = opA.double() @ opB.double()
      rounded = opA.bfloat16().double() @ opB.bfloat16().double()
      scale = float(exact.abs().max())
      e_mode = float((C.cpu().double() - exact).abs().max()) / scale
      e_acc = float((C.cpu().double() - rounded).abs().max()) / scale
      print(f'bf16-input gemm {M}x{N}x{K}: err vs exact {e_mode:.2e}, vs rounded operands {e_acc:.2e}')
      assert e_acc < 2e-6 and 1e-5 < e_mode < 1e-2, (e_mode, e_acc)
  finally:
    hip.set_gemm_mode(prev)
  # back in the default mode: fp32-level accuracy again
  A, B, C = rnd(300, 700, seed=1), rnd(700, 200, seed=2), torch.zeros(300, 200).cuda()
  hip.gemm(A.cuda(), B.cuda(), C)
  close(C, A.double() @ B.double(), rtol=5e-6, what='default mode restored')


def test_gemm_special_values(hip):
  """Edge values through the exact 3-way bf16 split (gemm_core.h split3): subnormal operands
  (the bf16 matrix pipe may flush them: the result must still be within 1e-36 absolute of the
  float64 product - nothing of normal magnitude is lost), huge-but-finite operands (no spurious
  overflow from the split), and +-inf / NaN operands: the affected outputs must be NON-FINITE
  (inf - inf inside the split turns an inf into NaN; the learner's numerics check treats both
  alike, tfutils.py:207,249), everything else stays exact."""
  M, N, K = 70, 66, 100
  A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
  A[3, 5], A[10, 7], B[9, 4] = 1e-40, -3e-39, 2e-41          # subnormals
  A[20, 11], B[11, 30] = 3e18, 1e19                           # product 3e37 < FLT_MAX
  C = torch.zeros(M, N).cuda()
  hip.gemm(A.cuda(), B.cuda(), C)
  want = A.double() @ B.double()
  got = C.cpu().double()
  assert torch.isfinite(got).all()
  err = (got - want).abs()
  assert float((err / (A.double().abs() @ B.double().abs() + 1e-30)).max()) < 2e-6
  for bad in (float('inf'), float('-inf'), float('nan')):
    A2 = A.clone()
    A2[33, 17] = bad
    hip.gemm(A2.cuda(), B.cuda(), C)
    got = C.cpu()
    assert not torch.isfinite(got[33]).any(), bad            # the whole output row is poisoned
    rest = torch.cat([got[:33], got[34:]])
    assert torch.isfinite(rest).all()
    close(rest, torch.cat([want[:33], want[34:]]), rtol=5e-6, what='rows untouched by the special value')


def test_gemm_views(hip, ref):
  """Column slices of wider buffers as operands and output (ld != cols)."""
  Abuf, Bbuf, Cbuf = rnd(100, 300, seed=1), rnd(80, 64, seed=2), rnd(100, 200, seed=3)
  def fn(ops, Abuf, Bbuf, Cbuf):
    ops.gemm(Abuf[:, 16:96], Bbuf[:, 8:40], Cbuf[:, 100:132], False, False, 1.0, 1.0)
  (g, c), = both(hip, ref, fn, [Abuf, Bbuf, Cbuf], [2])
  close(g, c, what='gemm views')
  # unaligned leading dimension / odd offsets -> scalar loader path
  Abuf, Bbuf, Cbuf = rnd(37, 103, seed=4), rnd(50, 31, seed=5), rnd(37, 31, seed=6)
  def fn2(ops, Abuf, Bbuf, Cbuf):
    ops.gemm(Abuf[:, 3:53], Bbuf, Cbuf, False, False, 1.0, 0.0)
  (g, c), = both(hip, ref, fn2, [Abuf, Bbuf, Cbuf], [2])
  close(g, c, what='gemm unaligned')


CONVS = [  # n, hb, Cb, hs, Cs, k, u8
    (3, 64, 3, 31, 16, 4, True), (3, 31, 16, 14, 32, 4, False),
    (2, 14, 32, 6, 64, 4, False), (5, 6, 64, 2, 128, 4, False),
    (4, 5, 64, 1, 320, 5, False), (3, 13, 32, 5, 64, 5, False),
    (2, 30, 16, 13, 32, 6, False), (2, 64, 3, 30, 16, 6, False),
    (2, 128, 6, 64, 32, 2, False), (1, 64, 3, 31, 16, 4, False), (2, 20, 5, 8, 12, 5, False),
    # the model's own layer geometry (banded transposed conv: all four parities for even k with an
    # output row that only receives the bias at 31 = 2*13 + 4 + 1, one contraction per parity for odd k)
    (3, 13, 128, 5, 256, 5, False), (2, 30, 64, 13, 128, 6, False), (2, 31, 64, 14, 128, 4, False),
    (3, 14, 128, 6, 256, 4, False), (130, 6, 64, 2, 64, 4, False), (2, 9, 64, 3, 32, 5, False),
    # thin image side with 64 feature channels (GEMM + col2im form): the decoder's RGB layer, an
    # encoder-geometry k = 4 layer, an odd kernel on an odd size, one- and two-channel images
    (3, 64, 3, 30, 64, 6, False), (2, 64, 3, 31, 64, 4, False), (2, 33, 3, 15, 64, 5, False),
    (1, 20, 2, 8, 64, 6, False), (40, 64, 1, 30, 64, 6, False),
    # the image-side filter-gradient kernel (conv_image.hip k_conv_image_wgrad: 3 channels, 64 features,
    # k 4 / 6): uint8 and float images, one image (two work items), an odd image count, a 32-wide image,
    # more images than workgroups
    (3, 64, 3, 31, 64, 4, True), (5, 64, 3, 30, 64, 6, True), (1, 64, 3, 31, 64, 4, True),
    (7, 32, 3, 14, 64, 6, False), (2, 32, 3, 15, 64, 4, False), (300, 64, 3, 31, 64, 4, True),
    (270, 16, 3, 6, 64, 6, False), (3, 64, 4, 31, 64, 4, True), (3, 64, 4, 30, 64, 6, False), (40, 64, 4, 31, 64, 4, False)]


@pytest.mark.parametrize('n,hb,Cb,hs,Cs,k,u8', CONVS)
def test_conv_down(hip, ref, n, hb, Cb, hs, Cs, k, u8):
  if u8:
    big = torch.randint(0, 256, (n, hb, hb, Cb), dtype=torch.uint8,
                        generator=torch.Generator().manual_seed(1))
  else:
    big = rnd(n, hb, hb, Cb, seed=1)
  w, bias, small = rnd(k, k, Cb, Cs, seed=2, scale=0.1), rnd(Cs, seed=3), torch.zeros(n, hs, hs, Cs)
  def fn(ops, big, w, bias, small):
    ops.conv_down(big, w, bias, small, k, 1.0 / 255.0 if u8 else 1.0)
  (g, c), = both(hip, ref, fn, [big, w, bias, small], [3])
  close(g, c, what='conv_down')


@pytest.mark.parametrize('n,hb,Cb,hs,Cs,k,u8', CONVS)
def test_conv_up(hip, ref, n, hb, Cb, hs, Cs, k, u8):
  small, w, bias = rnd(n, hs, hs, Cs, seed=1), rnd(k, k, Cb, Cs, seed=2, scale=0.1), rnd(Cb, seed=3)
  big = torch.full((n, hb, hb, Cb), 7.0)
  def fn(ops, small, w, bias, big):
    ops.conv_up(small, w, bias, big, k)
  (g, c), = both(hip, ref, fn, [small, w, bias, big], [3])
  close(g, c, what='conv_up')


@pytest.mark.parametrize('n,hb,Cb,hs,Cs,k,u8', CONVS)
def test_conv_wgrad(hip, ref, n, hb, Cb, hs, Cs, k, u8):
  if u8:
    big = torch.randint(0, 256, (n, hb, hb, Cb), dtype=torch.uint8,
                        generator=torch.Generator().manual_seed(1))
  else:
    big = rnd(n, hb, hb, Cb, seed=1)
  small, dw = rnd(n, hs, hs, Cs, seed=2), rnd(k, k, Cb, Cs, seed=3)
  for beta in (0.0, 1.0):
    def fn(ops, big, small, dw):
      ops.conv_wgrad(big, small, dw, k, 1.0 / 255.0 if u8 else 1.0, beta)
    (g, c), = both(hip, ref, fn, [big, small, dw], [2])
    close(g, c, what=f'conv_wgrad beta{beta}')


@pytest.mark.parametrize('n,hb,Cb,hs,k,u8', [(3, 64, 3, 31, 4, True), (5, 64, 3, 30, 6, False), (130, 32, 4, 15, 4, True),
                                             (2, 64, 1, 31, 4, False), (2, 128, 6, 63, 4, True)])
def test_conv_down_with_layernorm(hip, ref, n, hb, Cb, hs, k, u8):
  """dd_conv2d_s2_down_ln: image-side convolution + LayerNorm + ELU in one pass (the epilogue of
  k_conv_image_down) against conv_down + ln_act_fwd of the CPU restatement: pre-norm rows,
  activations, statistics.  The last geometry (6 channels) takes the wrapper's two-launch path."""
  Cs = 64
  if u8:
    big = torch.randint(0, 256, (n, hb, hb, Cb), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
  else:
    big = rnd(n, hb, hb, Cb, seed=1)
  w, bias = rnd(k, k, Cb, Cs, seed=2, scale=0.1), rnd(Cs, seed=3)
  gamma, beta = 1.0 + 0.1 * rnd(Cs, seed=4), 0.1 * rnd(Cs, seed=5)
  scale = 1.0 / 255.0 if u8 else 1.0
  z_c, out_c, st_c = torch.zeros(n, hs, hs, Cs), torch.zeros(n, hs, hs, Cs), torch.zeros(n * hs * hs, 2)
  ref.conv_down(big, w, bias, z_c, k, scale)
  ref.ln_act_fwd(z_c.view(-1, Cs), gamma, beta, out_c.view(-1, Cs), st_c, True)
  z_g, out_g, st_g = (torch.full(t.shape, 7.0).cuda() for t in (z_c, out_c, st_c))
  hip.conv_down_ln(big.cuda(), w.cuda(), bias.cuda(), gamma.cuda(), beta.cuda(), z_g, out_g, st_g, k, scale)
  torch.cuda.synchronize()
  close(z_g, z_c, what='pre-norm rows')
  close(out_g, out_c, rtol=2e-4, what='activations')
  close(st_g[:, 0], st_c[:, 0], rtol=2e-4, what='mean')
  close(st_g[:, 1], st_c[:, 1], rtol=2e-4, what='rstd')


@pytest.mark.parametrize('n,hb,Cb,hs,k', [(3, 64, 3, 30, 6), (130, 32, 4, 15, 4), (2, 64, 1, 31, 4), (600, 16, 3, 6, 6),
                                          (2, 128, 6, 63, 4)])
def test_conv_down_with_layernorm_backward(hip, ref, n, hb, Cb, hs, k):
  """dd_conv2d_s2_down_lnbwd: the data gradient of an image-side transposed convolution with the
  LayerNorm + ELU backward of the layer in front of it in the epilogue, against conv_down +
  ln_act_bwd of the CPU restatement: dz, dgamma, dbeta, dbias.  The last geometry (6 channels) takes
  the wrapper's two-launch path."""
  Cs = 64
  big, w = rnd(n, hb, hb, Cb, seed=1), rnd(k, k, Cb, Cs, seed=2, scale=0.1)
  z = rnd(n, hs, hs, Cs, seed=3)
  gamma, beta = 1.0 + 0.1 * rnd(Cs, seed=4), 0.1 * rnd(Cs, seed=5)
  zz = z.view(-1, Cs).double()
  stats = torch.stack([zz.mean(1), (zz.var(1, unbiased=False) + 1e-3).rsqrt()], 1).float()
  out = torch.nn.functional.elu((z.view(-1, Cs) - stats[:, :1]) * stats[:, 1:2] * gamma + beta)
  dout_c, dz_c = torch.zeros(n, hs, hs, Cs), torch.zeros(n, hs, hs, Cs)
  dg_c, db_c, dbias_c = torch.zeros(Cs), torch.zeros(Cs), torch.zeros(Cs)
  ref.conv_down(big, w, None, dout_c, k)
  ref.ln_act_bwd(dout_c.view(-1, Cs), z.view(-1, Cs), out, stats, gamma, dz_c.view(-1, Cs), dg_c, db_c,
                 False, True, dbias_c, beta=beta)
  dout_g, dz_g = torch.full((n, hs, hs, Cs), 7.0).cuda(), torch.full((n, hs, hs, Cs), 7.0).cuda()
  dg_g, db_g, dbias_g = (torch.full((Cs,), 7.0).cuda() for _ in range(3))
  hip.conv_down_lnbwd(big.cuda(), w.cuda(), z.cuda(), stats.cuda(), gamma.cuda(), beta.cuda(), dout_g, dz_g,
                      dg_g, db_g, dbias_g, k)
  torch.cuda.synchronize()
  close(dz_g, dz_c, what='dz')
  close(dg_g, dg_c, what='dgamma')
  close(db_g, db_c, what='dbeta')
  close(dbias_g, dbias_c, what='dbias')


@pytest.mark.parametrize('n,hb,hs,k,u8', [(3, 64, 31, 4, True), (5, 64, 30, 6, False), (130, 32, 15, 4, True),
                                          (2, 64, 31, 4, False), (2, 20, 8, 6, True)])
def test_conv_wgrad_with_layernorm_backward(hip, ref, n, hb, hs, k, u8):
  """dd_conv2d_s2_wgrad_ln: the filter gradient of an image-side Conv2D + LayerNorm + ELU layer
  from the gradient at the layer output, with the LayerNorm backward applied while the rows are
  staged (k_conv_image_wgrad LNB) - against ln_act_bwd + conv_wgrad of the CPU restatement: filter
  gradient, LayerNorm scale / offset gradients and the bias gradient.  The last geometry (20-pixel
  image) is not covered by the kernel: the wrapper's two-launch path."""
  Cb, Cs = 3, 64
  if u8:
    big = torch.randint(0, 256, (n, hb, hb, Cb), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
  else:
    big = rnd(n, hb, hb, Cb, seed=1)
  dout, z = rnd(n, hs, hs, Cs, seed=2), rnd(n, hs, hs, Cs, seed=3)
  gamma, beta = 1.0 + 0.1 * rnd(Cs, seed=4), 0.1 * rnd(Cs, seed=5)
  zz = z.view(-1, Cs).double()
  mean, var = zz.mean(1), zz.var(1, unbiased=False)
  stats = torch.stack([mean, (var + 1e-3).rsqrt()], 1).float()
  scale = 1.0 / 255.0 if u8 else 1.0
  # restatement: the two launches
  dz_c, dw_c = torch.zeros(n, hs, hs, Cs), torch.zeros(k, k, Cb, Cs)
  dg_c, db_c, dbias_c = torch.zeros(Cs), torch.zeros(Cs), torch.zeros(Cs)
  out = torch.nn.functional.elu((z.view(-1, Cs) - stats[:, :1]) * stats[:, 1:2] * gamma + beta)
  ref.ln_act_bwd(dout.view(-1, Cs).clone(), z.view(-1, Cs), out, stats, gamma, dz_c.view(-1, Cs), dg_c, db_c,
                 False, True, dbias_c, beta=beta)
  ref.conv_wgrad(big, dz_c, dw_c, k, scale)
  # fused
  dev = lambda t: t.cuda()
  dz_g, dw_g = torch.full((n, hs, hs, Cs), 7.0).cuda(), torch.full((k, k, Cb, Cs), 7.0).cuda()
  dg_g, db_g, dbias_g = (torch.full((Cs,), 7.0).cuda() for _ in range(3))
  hip.conv_wgrad_ln(dev(big), dev(dout), dev(z), dev(stats), dev(gamma), dev(beta), dz_g, dw_g, dg_g, db_g, dbias_g,
                    k, scale)
  torch.cuda.synchronize()
  close(dw_g, dw_c, what='fused filter gradient')
  close(dg_g, dg_c, what='fused dgamma')
  close(db_g, db_c, what='fused dbeta')
  close(dbias_g, dbias_c, what='fused dbias')


# stride-1 SAME convolutions of the residual encoder / decoder: (n, h, Cin, Cout, k, u8)
SAME = [
    (3, 8, 16, 32, 3, False), (2, 16, 32, 32, 3, False), (5, 4, 64, 128, 3, False),
    (2, 32, 3, 16, 3, True), (2, 32, 16, 3, 3, False), (3, 7, 5, 6, 3, False),
    (2, 9, 8, 12, 1, False), (1, 64, 64, 64, 3, False), (40, 4, 256, 256, 3, False),
    (2, 6, 4, 8, 5, False)]


@pytest.mark.parametrize('n,h,Cin,Cout,k,u8', SAME)
def test_conv_same(hip, ref, n, h, Cin, Cout, k, u8):
  if u8:
    x = torch.randint(0, 256, (n, h, h, Cin), dtype=torch.uint8,
                      generator=torch.Generator().manual_seed(1))
  else:
    x = rnd(n, h, h, Cin, seed=1)
  w, bias, y = rnd(k, k, Cin, Cout, seed=2, scale=0.1), rnd(Cout, seed=3), rnd(n, h, h, Cout, seed=4)
  sc = 1.0 / 255.0 if u8 else 1.0
  for alpha, beta, bs in ((1.0, 0.0, True), (0.1, 1.0, True), (1.0, 0.0, False)):
    def fn(ops, x, w, bias, y):
      ops.conv_same(x, w, bias if bs else None, y, k, sc, alpha, beta)
    (g, c), = both(hip, ref, fn, [x, w, bias, y], [3])
    close(g, c, what=f'conv_same alpha{alpha} beta{beta} bias{bs}')


@pytest.mark.parametrize('n,h,Cin,Cout,k,u8', SAME)
def test_conv_same_bwd(hip, ref, n, h, Cin, Cout, k, u8):
  dy, w, dx = rnd(n, h, h, Cout, seed=1), rnd(k, k, Cin, Cout, seed=2, scale=0.1), rnd(n, h, h, Cin, seed=3)
  for alpha, beta in ((1.0, 0.0), (0.1, 1.0)):
    def fn(ops, dy, w, dx):
      ops.conv_same_bwd(dy, w, dx, k, alpha, beta)
    (g, c), = both(hip, ref, fn, [dy, w, dx], [2])
    close(g, c, what=f'conv_same_bwd alpha{alpha} beta{beta}')


@pytest.mark.parametrize('n,h,Cin,Cout,k,u8', SAME)
def test_conv_same_wgrad(hip, ref, n, h, Cin, Cout, k, u8):
  if u8:
    x = torch.randint(0, 256, (n, h, h, Cin), dtype=torch.uint8,
                      generator=torch.Generator().manual_seed(1))
  else:
    x = rnd(n, h, h, Cin, seed=1)
  dy, dw = rnd(n, h, h, Cout, seed=2), rnd(k, k, Cin, Cout, seed=3)
  sc = 1.0 / 255.0 if u8 else 1.0
  for alpha, beta in ((1.0, 0.0), (0.1, 1.0)):
    def fn(ops, x, dy, dw):
      ops.conv_same_wgrad(x, dy, dw, k, sc, alpha, beta)
    (g, c), = both(hip, ref, fn, [x, dy, dw], [2])
    close(g, c, what=f'conv_same_wgrad alpha{alpha} beta{beta}')


def test_conv_same_adjoint_large(hip):
  """Size-independent property at a full-size residual layer (2500 images would be 32x32x64 ->
  64; here 256 images): <conv(x), y> == <x, conv_bwd(y)> and <conv(x), y> == <w, wgrad(x, y)>."""
  n, h, Cin, Cout, k = 256, 32, 64, 64, 3
  x = rnd(n, h, h, Cin, seed=1).cuda()
  y = rnd(n, h, h, Cout, seed=2).cuda()
  w = rnd(k, k, Cin, Cout, seed=3, scale=0.1).cuda()
  cx, cty, dw = torch.empty_like(y), torch.empty_like(x), torch.empty_like(w)
  hip.conv_same(x, w, None, cx, k)
  hip.conv_same_bwd(y, w, cty, k)
  hip.conv_same_wgrad(x, y, dw, k)
  torch.cuda.synchronize()
  a = float((cx.double() * y.double()).sum())
  b = float((x.double() * cty.double()).sum())
  c = float((w.double() * dw.double()).sum())
  assert abs(a - b) <= 1e-5 * abs(a) and abs(a - c) <= 1e-5 * abs(a), (a, b, c)


@pytest.mark.parametrize('n,h,C', [(3, 4, 16), (2, 16, 64), (5, 2, 3), (1, 32, 128), (7, 8, 6)])
def test_pool2_repeat2(hip, ref, n, h, C):
  x, y = rnd(n, 2 * h, 2 * h, C, seed=1), rnd(n, h, h, C, seed=2)
  for scale in (0.25, 1.0):
    def fn(ops, x, y):
      ops.pool2(x, y, scale)
    (g, c), = both(hip, ref, fn, [x, y], [1])
    close(g, c, rtol=1e-6, what=f'pool2 scale{scale}')
  for scale, beta in ((1.0, 0.0), (0.25, 0.0), (0.25, 1.0)):
    def fn(ops, y, x):
      ops.repeat2(y, x, scale, beta)
    (g, c), = both(hip, ref, fn, [y, x], [1])
    close(g, c, rtol=1e-6, what=f'repeat2 scale{scale} beta{beta}')


def test_conv_adjoint_large(hip):
  """Size-independent property at full C2 layer size: <down(x), y> == <x, up(y)>
  (the transposed conv is the exact adjoint of the conv)."""
  n, hb, Cb, hs, Cs, k = 64, 31, 64, 14, 128, 4
  x = rnd(n, hb, hb, Cb, seed=1).cuda()
  y = rnd(n, hs, hs, Cs, seed=2).cuda()
  w = rnd(k, k, Cb, Cs, seed=3, scale=0.1).cuda()
  dx, dy = torch.empty_like(y), torch.empty_like(x)
  hip.conv_down(x, w, None, dx, k)
  hip.conv_up(y, w, None, dy, k)
  a = float((dx.double() * y.double()).sum())
  b = float((x.double() * dy.double()).sum())
  assert abs(a - b) <= 1e-4 * max(abs(a), abs(b), 1.0), (a, b)


@pytest.mark.parametrize('rows,C,act', [(50, 256, 1), (2500, 512, 1), (1000, 64, 1),
                                        (77, 768, 0), (33, 40, 1), (9, 1500, 1), (300, 1024, 0),
                                        (5000, 128, 1), (3, 64, 0), (130, 42, 1)])
def test_ln_act(hip, ref, rows, C, act):
  z, gamma, beta = rnd(rows, C, seed=1, scale=2.0), 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
  out, stats, dout = torch.zeros(rows, C), torch.zeros(rows, 2), rnd(rows, C, seed=4)
  dz, dg, db, dbp = torch.zeros(rows, C), rnd(C, seed=5), rnd(C, seed=6), rnd(C, seed=7)
  def fn(ops, z, gamma, beta, out, stats, dout, dz, dg, db, dbp):
    ops.ln_act_fwd(z, gamma, beta, out, stats, bool(act))
    ops.ln_act_bwd(dout, z, out, stats, gamma, dz, dg, db, True, bool(act), dbp)
  res = both(hip, ref, fn, [z, gamma, beta, out, stats, dout, dz, dg, db, dbp], [3, 4, 6, 7, 8, 9])
  for (g, c), nm in zip(res, ['out', 'stats', 'dz', 'dgamma', 'dbeta', 'dbias_pre']):
    close(g, c, rtol=1e-4, what=f'ln {nm}')
  dg2, db2 = torch.zeros(C), torch.zeros(C)
  def fn2(ops, z, gamma, beta, out, stats, dout, dg2, db2):
    ops.ln_act_fwd(z, gamma, beta, out, stats, bool(act))
    ops.ln_param_grad(dout, z, out, stats, dg2, db2, False, bool(act))
  res = both(hip, ref, fn2, [z, gamma, beta, out, stats, dout, dg2, db2], [6, 7])
  for (g, c), nm in zip(res, ['dgamma', 'dbeta']):
    close(g, c, rtol=1e-4, what=f'ln_param_grad {nm}')
  # the activation recomputed from z (beta given: `out` is not read) == read from `out`, bit for bit
  zg, gg, bg, dog = z.cuda(), gamma.cuda(), beta.cuda(), dout.cuda()
  og, sg = torch.zeros(rows, C, device='cuda'), torch.zeros(rows, 2, device='cuda')
  hip.ln_act_fwd(zg, gg, bg, og, sg, bool(act))
  got = []
  for kw in (dict(), dict(beta=bg)):
    dzg, dgg, dbg, dpg = (torch.zeros(rows, C, device='cuda'), torch.zeros(C, device='cuda'),
                          torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'))
    hip.ln_act_bwd(dog.clone(), zg, og if not kw else torch.full_like(og, float('nan')), sg, gg, dzg,
                   dgg, dbg, False, bool(act), dpg, **kw)
    dz0 = torch.zeros(rows, C, device='cuda')
    hip.ln_act_bwd(dog.clone(), zg, og if not kw else torch.full_like(og, float('nan')), sg, gg, dz0,
                   None, None, False, bool(act), **kw)
    got.append((dzg, dgg, dbg, dpg, dz0))
  torch.cuda.synchronize()
  for a, b in zip(*got):
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_ln_gru_strided_rows(hip, ref):
  """Scan-step views of batch-major [B,T,C] buffers: every operand, including
  the (mean, rstd) stats, is row-strided."""
  B, T, C, D = 6, 5, 256, 64
  z, out, stats = rnd(B, T, C, seed=1), torch.zeros(B, T, C), torch.zeros(B, T, 2)
  gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
  dout, dz = rnd(B, T, C, seed=4), torch.zeros(B, T, C)
  z3, h, hn, gst = rnd(B, T, 3 * D, seed=5), rnd(B, T, D, seed=6), torch.zeros(B, T, D), torch.zeros(B, T, 2)
  g3, b3 = 1 + 0.1 * rnd(3 * D, seed=7), 0.1 * rnd(3 * D, seed=8)
  dhn, dz3, dh, dy3 = rnd(B, T, D, seed=9), torch.zeros(B, T, 3 * D), torch.zeros(B, T, D), torch.zeros(B, T, 3 * D)
  def fn(ops, z, out, stats, gamma, beta, dout, dz, z3, h, hn, gst, g3, b3, dhn, dz3, dh, dy3):
    for t in range(T):
      ops.ln_act_fwd(z[:, t], gamma, beta, out[:, t], stats[:, t], True)
      ops.gru_fwd(z3[:, t], g3, b3, h[:, t], hn[:, t], gst[:, t])
    for t in range(T):
      ops.ln_act_bwd(dout[:, t], z[:, t], out[:, t], stats[:, t], gamma, dz[:, t])
      ops.gru_bwd(dhn[:, t], z3[:, t], gst[:, t], g3, b3, h[:, t], dz3[:, t], dh[:, t], dy3[:, t])
  res = both(hip, ref, fn, [z, out, stats, gamma, beta, dout, dz, z3, h, hn, gst, g3, b3, dhn, dz3, dh, dy3],
             [1, 2, 6, 9, 10, 14, 15, 16])
  for (g, c), nm in zip(res, ['out', 'stats', 'dz', 'hn', 'gstats', 'dz3', 'dh', 'dy3']):
    close(g, c, rtol=1e-4, what=f'strided {nm}')


def test_ln_act_views(hip, ref):
  buf, obuf = rnd(40, 600, seed=1), torch.zeros(40, 700)
  gamma, beta, stats = 1 + 0.1 * rnd(256, seed=2), 0.1 * rnd(256, seed=3), torch.zeros(40, 2)
  def fn(ops, buf, gamma, beta, obuf, stats):
    ops.ln_act_fwd(buf[:, 100:356], gamma, beta, obuf[:, 256:512], stats, True)
  (g, c), = both(hip, ref, fn, [buf, gamma, beta, obuf, stats], [3])
  close(g, c, rtol=1e-5, what='ln views')


@pytest.mark.parametrize('rows,D', [(50, 256), (2500, 256), (17, 64), (5, 1024), (9, 96), (33, 512), (37, 4096), (7, 2048)])
def test_gru(hip, ref, rows, D):
  z3, gamma, beta = rnd(rows, 3 * D, seed=1, scale=2.0), 1 + 0.1 * rnd(3 * D, seed=2), 0.1 * rnd(3 * D, seed=3)
  h, hn, stats, dhn = rnd(rows, D, seed=4), torch.zeros(rows, D), torch.zeros(rows, 2), rnd(rows, D, seed=5)
  dz3, dh, dy3 = torch.zeros(rows, 3 * D), torch.zeros(rows, D), torch.zeros(rows, 3 * D)
  def fn(ops, z3, gamma, beta, h, hn, stats, dhn, dz3, dh, dy3):
    ops.gru_fwd(z3, gamma, beta, h, hn, stats)
    ops.gru_bwd(dhn, z3, stats, gamma, beta, h, dz3, dh, dy3)
  res = both(hip, ref, fn, [z3, gamma, beta, h, hn, stats, dhn, dz3, dh, dy3], [4, 7, 8, 9])
  for (g, c), nm in zip(res, ['hn', 'dz3', 'dh', 'dy3']):
    close(g, c, rtol=1e-4, what=f'gru {nm}')


@pytest.mark.parametrize('rows,G,C,unimix', [(50, 32, 32, 0.01), (2500, 32, 32, 0.01),
                                             (96, 8, 8, 0.01), (10, 64, 64, 0.01), (40, 5, 12, 0.0)])
def test_stats_sample(hip, ref, rows, G, C, unimix):
  x = rnd(rows, G * C, seed=1, scale=2.0)
  u = torch.rand(rows, G, generator=torch.Generator().manual_seed(2))
  logit, wide = torch.zeros(rows, G * C), torch.zeros(rows, G * C + 40)
  dlogit, dstoch, dx = rnd(rows, G * C, seed=3), rnd(rows, G * C, seed=4), torch.zeros(rows, G * C)
  def fn(ops, x, u, logit, wide, dlogit, dstoch, dx):
    ops.stats_fwd(x, u, logit, wide[:, 24:24 + G * C], G, C, unimix, 0)
    ops.stats_bwd(x, dlogit, dstoch, dx, G, C, unimix)
  res = both(hip, ref, fn, [x, u, logit, wide, dlogit, dstoch, dx], [2, 3, 6])
  close(*res[0], rtol=1e-5, what='logit')
  g, c = res[1]
  # discrete draws: BIT-EXACT, zero tolerated rows - against the numpy restatement (ref) and
  # against the host twin compiled from the kernel's own source (dd_onehot_sample_host)
  from daydreamer_amd import hipops
  assert torch.equal(g.cpu(), c), f'{int((g.cpu() != c).any(-1).sum())} rows differ from the restatement'
  idx_h, st_h, _ = hipops.onehot_sample_host(x, u, G, C, unimix, 0)
  assert torch.equal(g.cpu()[:, 24:24 + G * C], st_h), 'device draw != host twin'
  assert torch.equal(g.cpu()[:, 24:24 + G * C].reshape(rows, G, C).argmax(-1).int(), idx_h)
  assert float(g.sum()) == rows * G
  close(*res[2], rtol=1e-4, what='dx')
  # argmax mode
  def fn2(ops, x, logit, wide):
    ops.stats_fwd(x, None, logit, wide[:, 24:24 + G * C], G, C, unimix, 1)
  res = both(hip, ref, fn2, [x, logit, wide], [2])
  assert torch.equal(res[0][0].cpu(), res[0][1])


def test_sampler_edges_bit_exact(hip):
  """Adversarial uniforms: u placed exactly on / one ulp around every CDF edge (where any
  difference in summation order or a fused multiply-add would flip the draw): the device,
  its host twin and the numpy restatement must still agree on every index."""
  from daydreamer_amd import hipops
  from oracle import ref_ops
  rows, G, C, um = 64, 32, 32, 0.01
  x = rnd(rows, G * C, seed=5, scale=2.0)
  _, pm = ref_ops.sample_twin_np(x.numpy(), np.zeros((rows, G), np.float32), G, C, um)
  cdf = np.cumsum(pm.astype(np.float64), -1)
  us = []
  for k in range(3):  # an edge per (row, group), nudged by -1 / 0 / +1 ulp
    e = np.take_along_axis(cdf, np.random.RandomState(k).randint(0, C - 1, (rows, G, 1)), -1)[..., 0]
    u0 = (e / cdf[..., -1]).astype(np.float32)
    us.append(np.nextafter(u0, np.float32(k - 1), dtype=np.float32) if k != 1 else u0)
  for u in us:
    u = torch.from_numpy(np.clip(u, 0, np.float32(1) - np.float32(2 ** -24)))
    lg, st = torch.zeros(rows, G * C).cuda(), torch.zeros(rows, G * C).cuda()
    hip.stats_fwd(x.cuda(), u.cuda(), lg, st, G, C, um, 0)
    idx_h, st_h, _ = hipops.onehot_sample_host(x, u, G, C, um, 0)
    idx_n, _ = ref_ops.sample_twin_np(x.numpy(), u.numpy(), G, C, um)
    assert torch.equal(st.cpu(), st_h)
    assert np.array_equal(idx_h.numpy(), idx_n)


def test_sampler_distribution(hip):
  """Property at full size: empirical class frequencies match the unimixed
  softmax probabilities."""
  rows, G, C = 20000, 32, 32
  x = rnd(1, G * C, seed=1).repeat(rows, 1).cuda()
  u = torch.rand(rows, G, generator=torch.Generator().manual_seed(3)).cuda()
  logit, st = torch.zeros(rows, G * C).cuda(), torch.zeros(rows, G * C).cuda()
  hip.stats_fwd(x, u, logit, st, G, C, 0.01, 0)
  freq = st.mean(0).cpu().double()
  p = torch.exp(logit[0].cpu().double())
  assert float((freq - p).abs().max()) < 5 * (0.25 / rows) ** 0.5


@pytest.mark.parametrize('rows,G,C', [(50, 32, 32), (2500, 32, 32), (96, 8, 8), (10, 64, 64)])
def test_kl(hip, ref, rows, G, C):
  a = torch.log_softmax(rnd(rows, G, C, seed=1), -1).reshape(rows, G * C).contiguous()
  b = torch.log_softmax(rnd(rows, G, C, seed=2), -1).reshape(rows, G * C).contiguous()
  kl, ea, eb = torch.zeros(rows), torch.zeros(rows), torch.zeros(rows)
  da, db, coef = torch.zeros(rows, G * C), torch.zeros(rows, G * C), torch.tensor([0.7])
  def fn(ops, a, b, kl, ea, eb, da, db, coef):
    ops.kl_fwd(a, b, kl, ea, eb, G, C)
    ops.kl_bwd(a, b, coef, 1.0 / rows, 0.8, da, db, G, C)
  res = both(hip, ref, fn, [a, b, kl, ea, eb, da, db, coef], [2, 3, 4, 5, 6])
  for (g, c), nm in zip(res, ['kl', 'ent_post', 'ent_prior', 'dpost', 'dprior']):
    close(g, c, rtol=1e-4, what=f'kl {nm}')


def test_losses(hip, ref):
  rows = 37
  z = rnd(rows, 8, 8, 3, seed=1)
  img = torch.randint(0, 256, (rows, 8, 8, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
  loss, dz = torch.zeros(rows), torch.zeros(rows, 8, 8, 3)
  res = both(hip, ref, lambda ops, z, img, loss, dz: ops.image_loss(z, img, loss, dz, 0.01),
             [z, img, loss, dz], [2, 3])
  for g, c in res:
    close(g, c, rtol=1e-5, what='image_loss')
  res = both(hip, ref, lambda ops, z, img, loss, dz: ops.image_loss(z, img, loss, dz, 0.01, 1, 3),
             [z, img, loss, dz], [2, 3])
  for g, c in res:
    close(g, c, rtol=1e-5, what='image_loss channel range')
  pred, tgt, dp = rnd(rows, 19, seed=3), rnd(rows, 19, seed=4), torch.zeros(rows, 19)
  res = both(hip, ref, lambda ops, pred, tgt, loss, dp: ops.mse_loss(pred, tgt, loss, dp, 0.3),
             [pred, tgt, loss, dp], [2, 3])
  for g, c in res:
    close(g, c, rtol=1e-5, what='mse_loss')
  for kind in (0, 1):
    p1, t1 = rnd(1000, seed=5, scale=3.0), rnd(1000, seed=6, scale=3.0)
    if kind == 1:
      t1 = (t1 > 0).float()
    l1, d1 = torch.zeros(1000), torch.zeros(1000)
    res = both(hip, ref, lambda ops, p1, t1, l1, d1: ops.scalar_loss(p1, t1, l1, d1, 0.2, kind),
               [p1, t1, l1, d1], [2, 3])
    for g, c in res:
      close(g, c, rtol=1e-5, what=f'scalar_loss {kind}')


@pytest.mark.parametrize('M,N,K', [(300, 512, 1030), (64, 256, 1031), (2500, 512, 1030)])
def test_gemm_deferred_on_a_ragged_contraction_axis(hip, M, N, K):
  """HipOps.gemm(..., defer=True) with K % 4 != 0 takes the K-peel path (remainder
  call into C, then the aligned bulk with beta = 1 and the deferred split-K sum); its Slabs handle
  fed to ln_act_fwd `pre=` must give what the complete product followed by the same LayerNorm
  gives, also with a caller beta."""
  x, W = rnd(M, K, seed=1).cuda(), rnd(K, N, seed=2).cuda()
  gamma, beta_ln = (1 + 0.1 * rnd(N, seed=3)).cuda(), (0.1 * rnd(N, seed=4)).cuda()
  c0 = rnd(M, N, seed=5).cuda()
  for beta in (0.0, 1.0):
    za, zb = c0.clone(), c0.clone()
    oa, ob = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
    sa, sb = torch.zeros(M, 2, device='cuda'), torch.zeros(M, 2, device='cuda')
    assert hip.gemm(x, W, za, beta=beta) is None
    hip.ln_act_fwd(za, gamma, beta_ln, oa, sa, True)
    pre = hip.gemm(x, W, zb, beta=beta, defer=True)
    hip.ln_act_fwd(zb, gamma, beta_ln, ob, sb, True, pre=pre)
    torch.cuda.synchronize()
    # (pre may be None when the bulk did not split K: then zb is complete like za)
    close(zb, za.cpu(), rtol=1e-6, what=f'deferred z beta{beta} (pre {"set" if pre is not None else "none"})')
    close(ob, oa.cpu(), rtol=1e-5, what=f'deferred LayerNorm output beta{beta}')
    close(sb, sa.cpu(), rtol=1e-5, what=f'deferred LayerNorm statistics beta{beta}')


@pytest.mark.parametrize('rows,C', [(2500, 512), (40000, 512), (37, 128), (1000, 1024)])
def test_ln_act_with_a_folded_one_unit_head(hip, ref, rows, C):
  """dd_ln_act_fwd_head / dd_ln_act_bwd_head: a LayerNorm + ELU layer with the one-unit output layer
  behind it folded in (scalar heads).  Forward: the layer's own outputs are bit-identical to
  dd_ln_act_fwd and head_out = out @ w + b; backward: equal to dd_ln_act_bwd on the explicit
  outer product dy x w, parameter-gradient partials included."""
  z = rnd(rows, C, seed=1, scale=2.0)
  gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
  w, b, dy = rnd(C, 1, seed=4, scale=0.1), rnd(1, seed=5), rnd(rows, 1, seed=6)
  def fwd(ops, z, gamma, beta, out, st, w, b, y, out2, st2):
    ops.ln_act_fwd(z.clone(), gamma, beta, out, st, True, head=(w, b, y))
    ops.ln_act_fwd(z.clone(), gamma, beta, out2, st2, True)
  out, st, y, out2, st2 = (torch.zeros(rows, C), torch.zeros(rows, 2), torch.zeros(rows, 1), torch.zeros(rows, C),
                           torch.zeros(rows, 2))
  res = both(hip, ref, fwd, [z, gamma, beta, out, st, w, b, y, out2, st2], [3, 4, 7, 8, 9])
  (og, oc), (sg, sc), (yg, yc), (o2g, _), (s2g, _) = res
  assert torch.equal(og, o2g) and torch.equal(sg, s2g)            # the layer itself: untouched by the fold
  close(og, oc, rtol=1e-5, what='out'); close(yg, yc, rtol=1e-5, what='head output')
  want = (og.double() @ w.double().cuda() + b.double().cuda())
  close(yg, want.cpu(), rtol=2e-6, what='head output vs fp64 of the device layer output')
  def bwd(ops, z, out, st, gamma, beta, w, dy, dz, dg, db, dz2, dg2, db2):
    ops.ln_act_bwd_head(dy, w, z, out, st, gamma, dz, dg, db, False, True, beta=beta)
    dout = dy.reshape(-1, 1) * w.reshape(1, -1)
    ops.ln_act_bwd(dout.contiguous(), z, out, st, gamma, dz2, dg2, db2, False, True, beta=beta)
  zs = [torch.zeros(rows, C), torch.zeros(C), torch.zeros(C), torch.zeros(rows, C), torch.zeros(C), torch.zeros(C)]
  res = both(hip, ref, bwd, [z, oc, sc, gamma, beta, w, dy] + zs, [7, 8, 9, 10, 11, 12])
  (dzg, dzc), (dgg, dgc), (dbg, dbc), (dz2g, _), (dg2g, _), (db2g, _) = res
  close(dzg, dz2g.cpu(), rtol=1e-6, what='dz vs explicit outer product (device)')
  close(dgg, dg2g.cpu(), rtol=1e-5, what='dgamma vs explicit'); close(dbg, db2g.cpu(), rtol=1e-5, what='dbeta vs explicit')
  close(dzg, dzc, rtol=1e-5, what='dz'); close(dgg, dgc, rtol=2e-4, what='dgamma'); close(dbg, dbc, rtol=2e-4, what='dbeta')


def test_reduce_stats_multi_equals_single_launches(hip):
  """dd_reduce_stats_multi: the step's metric statistics as one launch - the same sums and
  extrema as dd_reduce_stats per vector, bit for bit (same block shape, same summation order),
  strided views and more than 16 vectors included."""
  xs = [rnd(n, seed=i, scale=1.0 + i).cuda() for i, n in enumerate([2500, 40000, 37500, 1, 7, 2500, 1024] + [300] * 12)]
  xs[5] = rnd(2500, 3, seed=50).cuda()[:, 1]          # a column view (stride 3)
  one = [(torch.zeros(3, dtype=torch.float64, device='cuda'), torch.zeros(3, device='cuda')) for _ in xs]
  many = [(torch.zeros(3, dtype=torch.float64, device='cuda'), torch.zeros(3, device='cuda')) for _ in xs]
  for x, (s_, m) in zip(xs, one):
    hip.reduce_stats(x, s_, m)
  hip.reduce_stats_multi([(x, s_, m) for x, (s_, m) in zip(xs, many)])
  torch.cuda.synchronize()
  for i, ((s1, m1), (s2, m2)) in enumerate(zip(one, many)):
    assert torch.equal(s1, s2) and torch.equal(m1, m2), i
  assert abs(float(one[1][0][0]) - float(xs[1].double().sum())) < 1e-6 * 40000


def test_video_grid(hip, ref):
  """dd_video_grid vs its restatement and vs the host form Agent.report used before (numpy,
  float64: agent.py:266-282 / tfutils.video_grid tfutils.py:390-392): batch-major and time-major
  image order, a channel slice of a two-camera tensor, with and without the truth sections."""
  B, T, hw, ct = 5, 4, 8, 6
  z = rnd(B * T, hw, hw, ct, seed=1, scale=2.0)
  img = torch.randint(0, 256, (B * T, hw, hw, ct), dtype=torch.uint8, generator=torch.Generator().manual_seed(2))
  nb = 3
  for c0, c1 in ((0, 6), (3, 6)):
    cn = c1 - c0
    for truth in (True, False):
      for zsb, zst, order in ((T, 1, 'bt'), (1, B, 'tb')):
        out = torch.zeros(T, (3 if truth else 1) * hw, nb * hw, cn)
        res = both(hip, ref, lambda ops, z, img, out: ops.video_grid(z, img if truth else None, out, nb, T, c0, c1, zsb, zst),
                   [z, img, out], [2])
        for g, c in res:
          close(g, c, rtol=1e-6, what=f'video_grid {order} truth={truth} channels {c0}:{c1}')
        zz = z.numpy().astype(np.float64).reshape(((B, T) if order == 'bt' else (T, B)) + (hw, hw, ct))
        ii = img.numpy().reshape(zz.shape)
        if order == 'tb':
          zz, ii = zz.transpose(1, 0, 2, 3, 4), ii.transpose(1, 0, 2, 3, 4)
        m = 1.0 / (1.0 + np.exp(-zz[:nb, ..., c0:c1]))
        video = m
        if truth:
          tr = ii[:nb, ..., c0:c1].astype(np.float64) / 255.0
          video = np.concatenate([tr, m, (m - tr + 1) / 2], 2)
        b_, t_, h_, w_, c_ = video.shape
        want = video.transpose(1, 2, 0, 3, 4).reshape(t_, h_, b_ * w_, c_)
        assert np.abs(res[0][0].cpu().numpy() - want).max() < 1e-6


def test_normal_head(hip, ref):
  rows, A, rows_ent = 400, 16, 300
  om, os, eps = rnd(rows, A, seed=1), rnd(rows, A, seed=2), rnd(rows, A, seed=3)
  act, dact, w = torch.zeros(rows, A + 8), rnd(rows, A, seed=4), torch.rand(rows)
  scale, dom, dos, er = torch.rand(A), torch.zeros(rows, A), torch.zeros(rows, A), torch.zeros(rows)
  lo, hi = 0.1, 1.0
  ent_lo, ent_div = float(np.log(lo)), float(np.log(hi) - np.log(lo))
  out = torch.zeros(2 * A, dtype=torch.float64)
  def fn(ops, om, os, eps, act, dact, w, scale, dom, dos, er, out):
    ops.normal_head_fwd(om, os, eps, act[:, 4:4 + A], lo, hi)
    ops.normal_head_bwd(om, os, eps, dact, w, scale, dom, dos, er, rows_ent, lo, hi,
                        1.0 / (rows_ent * ent_div), ent_lo, ent_div)
    ops.actent_stats(os, rows_ent, lo, hi, ent_lo, ent_div, out)
  res = both(hip, ref, fn, [om, os, eps, act, dact, w, scale, dom, dos, er, out], [3, 7, 8, 9, 10])
  for (g, c), nm in zip(res, ['action', 'dom', 'dos', 'ent_row', 'actent']):
    close(g, c, rtol=2e-5, what=f'normal_head {nm}')


@pytest.mark.parametrize('H,N', [(15, 2500), (5, 256), (3, 7)])
def test_imag_returns(hip, ref, H, N):
  rr, vr, cr = rnd(H + 1, N, seed=1), rnd(H + 1, N, seed=2), rnd(H + 1, N, seed=3, scale=2.0) + 2
  fc = (torch.rand(N) > 0.1).float()
  reward, value, cont = torch.zeros(H, N), torch.zeros(H + 1, N), torch.zeros(H + 1, N)
  weight, ret = torch.zeros(H + 1, N), torch.zeros(H, N)
  dret, dbase = rnd(H, N, seed=4), rnd(H, N, seed=5)
  drr, dvr, dcr = torch.zeros(H + 1, N), torch.zeros(H + 1, N), torch.zeros(H + 1, N)
  def fn(ops, rr, vr, cr, fc, reward, value, cont, weight, ret, dret, dbase, drr, dvr, dcr):
    ops.imag_returns_fwd(rr, vr, cr, fc, reward, value, cont, weight, ret, H, N, 0.995, 0.95)
    ops.imag_returns_bwd(dret, dbase, rr, vr, cr, value, ret, drr, dvr, dcr, H, N, 0.995, 0.95)
  res = both(hip, ref, fn, [rr, vr, cr, fc, reward, value, cont, weight, ret, dret, dbase, drr, dvr, dcr],
             [4, 5, 6, 7, 8, 11, 12, 13])
  for (g, c), nm in zip(res, ['reward', 'value', 'cont', 'weight', 'ret', 'drr', 'dvr', 'dcr']):
    close(g, c, rtol=2e-5, what=f'imag_returns {nm}')


def test_critic_actor_seed(hip, ref):
  H, N = 5, 300
  out, ret, w = rnd(H, N, seed=1), rnd(H, N, seed=2, scale=3.0), torch.rand(H + 1, N)
  loss, dout = torch.zeros(H, N), torch.zeros(H, N)
  res = both(hip, ref, lambda ops, out, ret, w, loss, dout: ops.critic_loss(out, ret, w, loss, dout, 0.1),
             [out, ret, w, loss, dout], [3, 4])
  for g, c in res:
    close(g, c, rtol=1e-5, what='critic_loss')
  base, er, sc = rnd(H + 1, N, seed=3), rnd(H + 1, N, seed=4), torch.tensor([1.3, 0.2, 0.7])
  dret, dbase = torch.zeros(H, N), torch.zeros(H, N)
  res = both(hip, ref, lambda ops, ret, base, w, er, sc, loss, dret, dbase:
             ops.actor_seed(ret, base, w, er, sc, loss, dret, dbase, 0.01),
             [ret, base, w, er, sc, loss, dret, dbase], [5, 6, 7])
  for g, c in res:
    close(g, c, rtol=1e-5, what='actor_seed')


def test_onehot_policy(hip, ref):
  rows, A, n = 300, 6, 200
  logit = torch.log_softmax(rnd(rows, A, seed=1), -1)
  act = torch.nn.functional.one_hot(torch.randint(0, A, (rows,), generator=torch.Generator().manual_seed(2)), A).float()
  wide = torch.zeros(rows, A + 10); wide[:, 3:3 + A] = act
  ret, base, w = rnd(n, seed=3), rnd(rows, seed=4), torch.rand(rows)
  sc, scale = torch.tensor([1.2, 0.1, 0.8]), torch.tensor([0.3])
  ent, dl, lpg, lent = torch.zeros(rows), torch.zeros(rows, A), torch.zeros(n), torch.zeros(n)
  ed = float(np.log(A))
  def fn(ops, logit, wide, ret, base, w, sc, scale, ent, dl, lpg, lent):
    ops.onehot_entropy(logit, ent, ed)
    ops.onehot_policy_grad(logit, wide[:, 3:3 + A], ret, base, w, sc, scale, dl, lpg, lent, n, 0.01, ed)
  res = both(hip, ref, fn, [logit, wide, ret, base, w, sc, scale, ent, dl, lpg, lent], [7, 8, 9, 10])
  for (g, c), nm in zip(res, ['ent', 'dlogit', 'loss_pg', 'loss_ent']):
    close(g, c, rtol=2e-5, what=f'onehot {nm}')


def test_philox(hip, ref):
  step = torch.tensor([12345], dtype=torch.int64)
  for kind, cols in ((0, 32), (1, 16), (0, 7), (1, 6)):
    out = torch.zeros(3, 50, cols)
    res = both(hip, ref, lambda ops, out, step: ops.philox(out, 3, 50, cols, 80, 30, 0x1234567890, step, 5, kind),
               [out, step], [0])
    g, c = res[0]
    if kind == 0:
      assert torch.equal(g.cpu(), c), 'philox uniforms must be bit-exact'
    else:
      close(g, c, rtol=1e-5, atol=2e-5, what='philox normal')
  # sharding invariance: rows [30,80) of the global field equal a shard with offset 30
  full, shard = torch.zeros(3, 80, 32).cuda(), torch.zeros(3, 50, 32).cuda()
  hip.philox(full, 3, 80, 32, 80, 0, 7, step.cuda(), 1, 0)
  hip.philox(shard, 3, 50, 32, 80, 30, 7, step.cuda(), 1, 0)
  assert torch.equal(full[:, 30:], shard)


def test_state_kernels(hip, ref):
  x = rnd(40000, seed=1)
  sums, maxs = torch.zeros(3, dtype=torch.float64), torch.zeros(3)
  res = both(hip, ref, lambda ops, x, sums, maxs: ops.reduce_stats(x, sums, maxs), [x, sums, maxs], [1, 2])
  close(*res[0], rtol=1e-9, what='reduce sums')
  close(*res[1], rtol=0, atol=0, what='reduce maxs')
  xs = rnd(300, 7, seed=2)
  res = both(hip, ref, lambda ops, xs, sums, maxs: ops.reduce_stats(xs[:, 3], sums, maxs), [xs, sums, maxs], [1, 2])
  close(*res[0], rtol=1e-9, what='reduce strided')
  # optimizer
  n, nd = 100000, 60000
  p, g, m, v = rnd(n, seed=3), rnd(n, seed=4, scale=5.0), rnd(n, seed=5).abs() * 0.1, rnd(n, seed=6).abs() * 0.1
  st = torch.tensor([3.0, 0.0, 0.0, 1e4, 7.0], dtype=torch.float64)
  def fn(ops, p, g, m, v, st):
    ops.grad_norm(g, st)
    ops.adam_step(p, g, m, v, nd, st, 1e-3, 1e-2, 1e-6, 0.9, 0.999, 100.0)
  res = both(hip, ref, fn, [p, g, m, v, st], [0, 2, 3, 4])
  for (a, b), nm in zip(res, ['p', 'm', 'v', 'state']):
    close(a, b, rtol=1e-5, what=f'adam {nm}')
  # learning-rate warm-up (tfutils.py:160-162): the advanced step count is 4 -> Adam at 4/10 of
  # lr, the decay at 3/10; past the warm-up (count 4 of 2) nothing changes
  for warm in (10, 2):
    st = torch.tensor([3.0, 0.0, 0.0, 1e4, 7.0], dtype=torch.float64)
    def fnw(ops, p, g, m, v, st):
      ops.grad_norm(g, st)
      ops.adam_step(p, g, m, v, nd, st, 1e-1, 1e-1, 1e-6, 0.9, 0.999, 100.0, warm)
    resw = both(hip, ref, fnw, [p, g, m, v, st], [0, 2, 3])
    for (a, b), nm in zip(resw, ['p', 'm', 'v']):
      close(a, b, rtol=1e-5, what=f'adam warmup {warm} {nm}')
    if warm == 2:
      def fn0(ops, p, g, m, v, st):
        ops.grad_norm(g, st)
        ops.adam_step(p, g, m, v, nd, st, 1e-1, 1e-1, 1e-6, 0.9, 0.999, 100.0)
      st0 = torch.tensor([3.0, 0.0, 0.0, 1e4, 7.0], dtype=torch.float64)
      res0 = both(hip, ref, fn0, [p, g, m, v, st0], [0])
      assert torch.equal(resw[0][0].cpu(), res0[0][0].cpu())
  # reduced-precision mode's loss-scale controller (tfutils.py:225-240): good step counts up,
  # 1000 good steps double the scale (clipped at 1e4), an overflow halves it, resets the count,
  # leaves the step number and the parameters alone
  for st0, gbad, want in (([3.0, 0, 0, 1e3, 7.0], False, [4.0, 1.0, 1e3, 8.0]),
                          ([3.0, 0, 0, 1e3, 1000.0], False, [4.0, 1.0, 2e3, 0.0]),
                          ([3.0, 0, 0, 1e4, 1000.0], False, [4.0, 1.0, 1e4, 0.0]),
                          ([3.0, 0, 0, 1e3, 500.0], True, [3.0, 0.0, 5e2, 0.0])):
    st = torch.tensor(st0, dtype=torch.float64)
    g2 = g.clone()
    if gbad:
      g2[777] = float('inf')
    def fn2(ops, p, g2, m, v, st):
      ops.grad_norm(g2, st, mixed=True)
      ops.adam_step(p, g2, m, v, nd, st, 1e-3, 1e-2, 1e-6, 0.9, 0.999, 100.0)
    res = both(hip, ref, fn2, [p, g2, m, v, st], [0, 4])
    for a in res[1]:
      got = a.cpu().numpy()
      assert [got[0], got[2], got[3], got[4]] == want, (st0, got)
    if gbad:
      assert torch.equal(res[0][0].cpu(), p) and torch.equal(res[0][1].cpu(), p)
  # autoadapt + normalize
  scale, s2 = torch.tensor([1.0, 0.5, 0.02]), torch.tensor([30.0, 5.0, 10.0], dtype=torch.float64)
  res = both(hip, ref, lambda ops, scale, s2: ops.autoadapt_update(scale, s2, 10.0, 1.0, 0.1, 0.1, 1e-3, 1.0, True),
             [scale, s2], [0])
  close(*res[0], rtol=1e-6, what='autoadapt')
  state, sm, outv = torch.tensor([0.1, 0.5, 3.0], dtype=torch.float64), torch.tensor([20.0, 90.0, 0.0], dtype=torch.float64), torch.zeros(2)
  insc = torch.tensor([1.5])
  res = both(hip, ref, lambda ops, state, sm, insc, outv: ops.normalize_update(state, sm, 50.0, insc, 0.99, 1e8, 1, True, outv),
             [state, sm, insc, outv], [0, 3])
  close(*res[0], rtol=1e-12, what='normalize state')
  close(*res[1], rtol=1e-6, what='normalize out')


def test_misc(hip, ref):
  prev, first, init, out = rnd(20, 64, seed=1), (torch.rand(20, 5) > 0.5).float(), rnd(64, seed=2), torch.zeros(20, 100)
  dprev = rnd(20, 64, seed=3)
  def fn(ops, prev, first, init, out, dprev):
    ops.reset_mask(prev, first[:, 2], init, out[:, 10:74])
    ops.reset_mask_bwd(out[:, 10:74], first[:, 2], dprev)
  res = both(hip, ref, fn, [prev, first, init, out, dprev], [3, 4])
  for g, c in res:
    close(g, c, rtol=1e-6, what='reset_mask')
  isf, ist = torch.rand(6, 9) > 0.7, torch.rand(6, 9) > 0.8
  act, ff, cf, am = rnd(6, 9, 5, seed=4), torch.zeros(6, 9), torch.zeros(6, 9), torch.zeros(54, 12)
  res = both(hip, ref, lambda ops, isf, ist, act, ff, cf, am: ops.batch_prep(isf, ist, act, ff, cf, am[:, 7:12]),
             [isf, ist, act, ff, cf, am], [3, 4, 5])
  for g, c in res:
    close(g, c, rtol=0, atol=0, what='batch_prep')
  x, dy, dx = rnd(256, seed=5), rnd(256, seed=6), rnd(256, seed=7)
  res = both(hip, ref, lambda ops, x, dy, dx: ops.tanh_bwd(x, dy, dx, 1.0), [x, dy, dx], [2])
  close(*res[0], rtol=1e-5, what='tanh_bwd')


@pytest.mark.parametrize('shape,dtype', [((64, 64, 3), torch.uint8), ((16,), torch.float32),
                                         ((), torch.float32), ((7,), torch.int64), ((), torch.bool)])
def test_replay_gather(hip, ref, shape, dtype):
  """dd_replay_gather: byte-exact rows for every wire dtype and row size (16-byte
  lanes for images / vectors, bytes for 4- and 1-byte rows)."""
  g = torch.Generator().manual_seed(3)
  ring = torch.randint(0, 2 if dtype == torch.bool else 200, (300,) + shape, generator=g).to(dtype)
  starts = torch.tensor([0, 288, 17, 100, 33], dtype=torch.int64)
  out_c = torch.zeros((5, 12) + shape, dtype=dtype)
  ref.replay_gather(ring, starts, out_c)
  out_g = torch.zeros((5, 12) + shape, dtype=dtype, device='cuda')
  hip.replay_gather(ring.cuda(), starts.cuda(), out_g)
  torch.cuda.synchronize()
  assert torch.equal(out_g.cpu(), out_c)
  f_c = torch.ones(5, 12, dtype=torch.bool)
  f_g = torch.ones(5, 12, dtype=torch.bool, device='cuda')
  ref.replay_gather(None, starts, f_c, first_flag=True)
  hip.replay_gather(None, starts.cuda(), f_g, first_flag=True)
  assert torch.equal(f_g.cpu(), f_c) and bool(f_c[:, 0].all()) and not bool(f_c[:, 1:].any())


def test_reset_mask_pairs_and_actent_large(hip, ref):
  """Two-segment reset kernels (views with leading dimensions, null prev) and the
  two-stage entropy statistics at the step's size and an odd action width."""
  post = rnd(50, 40, seed=1)
  first = (torch.rand(50, 4) > 0.6).float()
  ia, ib = rnd(24, seed=2), rnd(16, seed=3)
  oa, ob = torch.zeros(50, 30), torch.zeros(50, 64)
  da, db = rnd(50, 24, seed=4), rnd(50, 16, seed=5)
  def fn(ops, post, first, ia, ib, oa, ob, da, db):
    ops.reset_mask2(post[:, :24], ia, oa[:, 3:27], post[:, 24:], ib, ob[:, 40:56], first[:, 1])
    ops.reset_mask_bwd2(da, post[:, :24], db, post[:, 24:], first[:, 1])
    ops.reset_mask2(None, ia, oa[:, 3:27], None, None, ob[:, :16], first[:, 2])
  res = both(hip, ref, fn, [post, first, ia, ib, oa, ob, da, db], [0, 4, 5])
  for g, c in res:
    close(g, c, rtol=0, atol=0, what='reset_mask2')
  for rows, A in ((40000, 16), (37, 6), (5000, 3)):
    os_ = rnd(rows + 5, 2 * A, seed=6)
    out = torch.zeros(2 * A, dtype=torch.float64)
    res = both(hip, ref, lambda ops, os_, out: ops.actent_stats(os_[:, A:], rows, 0.1, 1.0, -2.3, 2.3, out),
               [os_, out], [1])
    close(*res[0], rtol=1e-6, what=f'actent_stats {rows}x{A}')


def test_deferred_splitk_sum(hip):
  """gemm(defer=True) leaves the split-K partial sums to the consumer kernel; every
  consumer must produce exactly (bitwise) what the reduce pass + plain consumer gives,
  including the written-back input buffer."""
  torch.manual_seed(0)
  dev = 'cuda'
  rows, K, D = 50, 512, 256
  x = torch.randn(rows, K, device=dev)
  def pair(N, consume, beta=0.0, bias=None):
    W = torch.randn(K, N, device=dev) * 0.05
    z0 = torch.randn(rows, N, device=dev)
    za, zb = z0.clone(), z0.clone()
    hip.gemm(x, W, za, beta=beta, bias=bias)
    ra = consume(za, None)
    pre = hip.gemm(x, W, zb, beta=beta, bias=bias, defer=True)
    assert pre is not None and pre.n > 1, 'shape no longer splits K: pick another'
    rb = consume(zb, pre)
    torch.cuda.synchronize()
    assert torch.equal(za, zb), 'written-back input'
    for a, b in zip(ra, rb):
      assert torch.equal(a, b)
  g, bt = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
  def ln(z, pre):
    out, st = torch.empty_like(z), torch.empty(rows, 2, device=dev)
    hip.ln_act_fwd(z, g, bt, out, st, True, pre=pre)
    return out, st
  pair(D, ln)
  pair(D, ln, beta=1.0)
  g3, b3, h = torch.rand(3 * D, device=dev) + 0.5, torch.randn(3 * D, device=dev), torch.randn(rows, D, device=dev)
  def gru(z, pre):
    hn, st = torch.empty(rows, D, device=dev), torch.empty(rows, 2, device=dev)
    hip.gru_fwd(z, g3, b3, h, hn, st, pre=pre)
    return hn, st
  pair(3 * D, gru)
  G, C = 32, 32
  u = torch.rand(rows, G, device=dev)
  def stats(z, pre):
    lg, sm = torch.empty(rows, G * C, device=dev), torch.empty(rows, G * C, device=dev)
    hip.stats_fwd(z, u, lg, sm, G, C, 0.01, 0, pre=pre)
    return lg, sm
  pair(G * C, stats, bias=torch.randn(G * C, device=dev))
  zz, oo = torch.randn(rows, D, device=dev), torch.randn(rows, D, device=dev)
  st = torch.stack([zz.mean(1), 1.0 / (zz.var(1, unbiased=False) + 1e-3).sqrt()], 1).contiguous()
  def lnb(dout, pre):
    dz = torch.empty(rows, D, device=dev)
    hip.ln_act_bwd(dout, zz, oo, st, g, dz, None, None, False, True, pre=pre)
    return (dz,)
  pair(D, lnb)


def test_deferred_splitk_sum_with_peeled_remainder(hip, ref):
  """K = stoch + action = 1030 (one-hot-action configs): HipOps.gemm runs k = 1028.. as a small call
  into C first (beta applied there) and the multiple-of-four bulk deferred with beta = 1, so the
  consumer's `pre` sum still yields beta * C + x @ W + bias followed by the LayerNorm."""
  torch.manual_seed(1)
  rows, K, D = 50, 1030, 512
  x, W = torch.randn(rows, K + 2), torch.randn(K, D) * 0.05   # row stride 1032: 16-byte aligned rows
  z0, bias = torch.randn(rows, D), torch.randn(D)
  g, bt = torch.rand(D) + 0.5, torch.randn(D)
  for beta in (0.0, 1.0):
    def fn(ops, x, W, z, bias, g, bt, out, st):
      pre = ops.gemm(x[:, :K], W, z, beta=beta, bias=bias, defer=True)
      if ops is hip:
        assert pre is not None and pre.n > 1 and pre.beta == 1.0
      ops.ln_act_fwd(z, g, bt, out, st, True, pre=pre)
    res = both(hip, ref, fn, [x, W, z0, bias, g, bt, torch.zeros(rows, D), torch.zeros(rows, 2)], [2, 6])
    close(*res[0], what=f'z beta{beta}')
    close(*res[1], what=f'ln(z) beta{beta}')


def test_native_fp32_mode(hip, ref):
  """dd_gemm_set_mode(0): the native v_mfma_f32_32x32x2_f32 loop stays parity-green next to
  the default split-bf16 loop; an invalid mode is rejected."""
  assert hip.lib.dd_gemm_set_mode(7) == -1
  prev = hip.lib.dd_gemm_set_mode(0)
  try:
    assert prev == 6
    for M, N, K, ta, tb in ((300, 260, 1999, 0, 0), (2500, 512, 1280, 0, 1), (50, 768, 512, 1, 0)):
      A = rnd(*((K, M) if ta else (M, K)), seed=1)
      B = rnd(*((N, K) if tb else (K, N)), seed=2)
      C = torch.zeros(M, N)
      res = both(hip, ref, lambda ops, A, B, C: ops.gemm(A, B, C, bool(ta), bool(tb)), [A, B, C], [2])
      close(*res[0], what=f'native gemm {M}x{N}x{K}')
    big, w, small = rnd(6, 30, 30, 16, seed=3), rnd(6, 6, 16, 32, seed=4) * 0.1, torch.zeros(6, 13, 13, 32)
    res = both(hip, ref, lambda ops, big, w, small: ops.conv_down(big, w, None, small, 6), [big, w, small], [2])
    close(*res[0], what='native conv_down')
  finally:
    assert hip.lib.dd_gemm_set_mode(6) == 0


def test_symexp(hip, ref):
  x = rnd(5000, seed=3) * 4.0
  x[:4] = torch.tensor([0.0, -0.0, 1e-8, -30.0])
  res = both(hip, ref, lambda ops, x, o: ops.symexp(x, o), [x, torch.zeros(5000)], [1])
  close(*res[0], rtol=2e-6, what='symexp')
  assert float(res[0][0][0]) == 0.0 and float(res[0][0][3]) == pytest.approx(-(np.exp(30.0) - 1), rel=1e-5)


def test_axpy_and_balance_stats(hip, ref):
  x, y, s = rnd(5000, seed=1), rnd(5000, seed=2), torch.tensor([0.37])
  def f1(ops, x, y, s):
    ops.axpy(x, 2.0, s, y, accumulate=False)
    ops.axpy(x, -0.5, None, y)
  res = both(hip, ref, f1, [x, y, s], [1])
  close(*res[0], rtol=1e-6, what='axpy')
  for kind, thres in ((0, 0.1), (1, 0.5)):
    out = rnd(40000, seed=3)
    tgt = rnd(40000, seed=4) if kind == 0 else (torch.rand(40000) > 0.02).float()
    loss = rnd(40000, seed=5).abs()
    o7 = torch.zeros(7, dtype=torch.float64)
    res = both(hip, ref, lambda ops, out, tgt, loss, o7: ops.balance_stats(out, tgt, loss, thres, kind, o7),
               [out, tgt, loss, o7], [3])
    g, c = res[0]
    assert torch.equal(g.cpu()[4], c[4]) and torch.equal(g.cpu()[2:4], c[2:4])  # counts are exact
    close(g, c, rtol=1e-6, what=f'balance_stats kind {kind}')


def test_scan_weight_planes_are_fragment_major(hip):
  """dd_scan_wprep / dd_scan_wprep_rows (include/daydreamer_hip.h): three bf16 planes of the exact
  3-way split of every weight, stored [column tile n/16][k-step k/128][plane][wave][lane] x 8 values
  with lane = ((k % 32) / 8) * 16 + n % 16 - the order in which the scans' workgroups read their
  MFMA B fragments (a wave's load = 1 KB contiguous).  Every element is checked against a numpy
  restatement of the split and of the index; columns beyond K are zero."""
  rng = np.random.RandomState(0)

  def split3(x):
    bits = x.view(np.uint32)
    h = bits & 0xFFFF0000
    r1 = x - h.view(np.float32)
    m = r1.view(np.uint32) & 0xFFFF0000
    lo = (r1 - m.view(np.float32)).view(np.uint32)
    return (h >> 16).astype(np.uint16), (m >> 16).astype(np.uint16), (lo >> 16).astype(np.uint16)

  def index(n, k, p, Kp):
    tile, r, it, kk = n >> 4, n & 15, k >> 7, k & 127
    w, q, e = kk >> 5, (kk & 31) >> 3, kk & 7
    return ((((tile * (Kp >> 7) + it) * 3 + p) * 4 + w) * 64 + (q * 16 + r)) * 8 + e

  for K, N, Kp in ((200, 48, 256), (384, 32, 384)):
    W = (rng.randn(K, N) * np.exp(rng.randn(K, N) * 3)).astype(np.float32)
    planes = torch.zeros(3 * N * Kp, dtype=torch.int16, device='cuda:0')
    hip.scan_wprep(torch.from_numpy(W).cuda(), planes, Kp)
    got = planes.cpu().numpy().view(np.uint16)
    Wp = np.zeros((Kp, N), np.float32)
    Wp[:K] = W
    n, k = np.meshgrid(np.arange(N), np.arange(Kp))
    for p, pl in enumerate(split3(Wp)):
      assert np.array_equal(got[index(n, k, p, Kp)], pl), (K, N, p)
    # the three planes add up to the weight exactly
    parts = [(got[index(n, k, p, Kp)].astype(np.uint32) << 16).view(np.float32) for p in range(3)]
    assert np.array_equal((parts[0] + parts[1]) + parts[2], Wp)
  # the reverse scan's cache: column n of the operand = row n of W
  Nr, Kr = 32, 256
  W = rng.randn(Nr, Kr).astype(np.float32)
  planes = torch.zeros(3 * Nr * Kr, dtype=torch.int16, device='cuda:0')
  hip.scan_wprep_rows(torch.from_numpy(W).cuda(), planes)
  got = planes.cpu().numpy().view(np.uint16)
  k, n = np.meshgrid(np.arange(Kr), np.arange(Nr))
  for p, pl in enumerate(split3(W)):
    assert np.array_equal(got[index(n, k, p, Kr)], pl), p
