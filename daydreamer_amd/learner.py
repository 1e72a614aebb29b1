"""The DreamerV2+ learner step on MI355X: explicit forward and hand-derived
backward over HIP kernels (no autograd, no tracing compiler).

Host-side mirror of the reference's `Agent.train` (reference
embodied/agents/dreamerv2plus/agent.py:67-93): world-model update
(WorldModel.train/loss agent.py:157-212, RSSM.observe nets.py:66-117), then
imagination with the *updated* world model (WorldModel.imagine agent.py:234-261),
critic update (VFunction.train agent.py:398-417, update_slow 444-454) and actor
update by backprop through the rollout (ImagActorCritic.update/loss
agent.py:326-381); optimizers as tfutils.py:180-283.

Design (MI355X-first):
  * every tensor lives in preallocated HBM buffers; the three optimizer groups
    are flat arenas (params / grads / Adam m / v) so clipping, weight decay,
    Adam and the data-parallel all-reduce are single passes over one buffer;
  * replay tensors stay batch-major [B,T,...]; a scan step addresses row
    b*T+t through the leading dimension, so the posterior states land directly
    in the [B*T, feat] matrix the heads and the imagination consume;
  * exact-algebra restructurings of the reference graph: the `embed` and
    action slices of the obs_out / img_in matmuls are hoisted out of the T-step
    scan; `initial()` is evaluated once per step on one row; weight gradients
    of scan layers are one bulk GEMM over all steps; reward / cont heads are
    evaluated once per trajectory;
  * everything that varies from step to step (RNG step, Adam step, controller
    scales) is device-resident, so the step is capturable into HIP graphs.

`ops` is the kernel backend (daydreamer_amd.hipops.HipOps in the product).
"""

import contextlib
import os
import math

import numpy as np
import torch

from . import graphs
from . import spec as specs

F32 = torch.float32

# noise sites for the Philox counter
SITE_OBS_PRIOR, SITE_OBS_POST, SITE_IMG, SITE_ACT, SITE_POLICY = 0, 1, 2, 3, 4


class ParamGroup:
  """Flat arena for one optimizer: params, grads and Adam moments; the
  weight-decayed tensors ('kernel') come first so decay is a prefix."""

  def __init__(self, plist, device, trainable=True, dtype=F32):
    F32 = dtype
    order = [p for p in plist if p.decay] + [p for p in plist if not p.decay]
    self.specs = order
    # every tensor starts on a 16-byte boundary (4 elements): a 3-element bias in the
    # middle of the arena would otherwise knock every later scale / bias vector off the
    # float4 paths of the LayerNorm kernels.  The padding stays zero (zero gradient).
    self.offset, off = {}, 0
    for p in order:
      self.offset[p.name] = off
      off += (p.size + 3) // 4 * 4
      if p.decay:
        self.n_decay = off
    if not any(p.decay for p in order):
      self.n_decay = 0
    self.n = off
    self.flat = torch.zeros(self.n, dtype=F32, device=device)
    self.p, self.g = {}, {}
    if trainable:
      self.gflat = torch.zeros(self.n, dtype=F32, device=device)
      self.m = torch.zeros(self.n, dtype=F32, device=device)
      self.v = torch.zeros(self.n, dtype=F32, device=device)
      # step, grad norm, finite flag, loss scale (1e4, tfutils.py:165), good steps
      self.opt_state = torch.tensor([0.0, 0.0, 1.0, 1e4, 0.0], dtype=torch.float64, device=device)
    for p in order:
      off = self.offset[p.name]
      self.p[p.name] = self.flat[off:off + p.size].view(p.shape)
      if trainable:
        self.g[p.name] = self.gflat[off:off + p.size].view(p.shape)

  def load(self, arrays):
    for p in self.specs:
      self.p[p.name].copy_(torch.as_tensor(np.asarray(arrays[p.name])
                                           ).reshape(p.shape))


class Lin:
  """Parameters of one Linear layer (+ LayerNorm when norm)."""

  def __init__(self, grp, prefix, norm, rows=None):
    self.norm = norm
    W, dW = grp.p[f'{prefix}/kernel'], grp.g.get(f'{prefix}/kernel')
    if rows is not None:  # row slice of a concatenated-input kernel
      W, dW = W[rows[0]:rows[1]], (dW[rows[0]:rows[1]] if dW is not None else None)
    self.W, self.dW = W, dW
    if norm:
      self.gamma, self.beta = grp.p[f'{prefix}/norm/scale'], grp.p[f'{prefix}/norm/bias']
      self.dgamma, self.dbeta = grp.g.get(f'{prefix}/norm/scale'), grp.g.get(f'{prefix}/norm/bias')
    else:
      self.bias, self.dbias = grp.p[f'{prefix}/bias'], grp.g.get(f'{prefix}/bias')
    self.units = W.shape[1]


class Act:
  """Activations (and their gradients) of one Lin in one row space."""

  def __init__(self, L, rows, units, norm, grads=True, out=None, dout=None):
    """out / dout may be column slices of a wider buffer (concat-free inputs
    of the next layer)."""
    self.z = L.zeros(rows, units)
    if norm:
      self.out = out if out is not None else L.zeros(rows, units)
      self.stats = L.zeros(rows, 2)
      if grads:
        self.dz = L.zeros(rows, units)
    if grads:
      self.dout = dout if dout is not None else L.zeros(rows, units)


class Learner:

  def __init__(self, spec, ops, device, batch, length, params=None, seed=0,
               rank=0, world=1, comm=None, noise_seed=0, dtype=F32,
               groups=None, ops2=None, ops_b=None, comm_b=None, ops_b2=None, dp_overlap=False):
    self.spec, self.ops, self.device = spec, ops, torch.device(device)
    # data parallel, default schedule: the all-reduce of the decoder / head gradients (the
    # contiguous [dec | reward | cont] kernel range of the model arena, 71 % of it at
    # configs[1]) is issued as soon as they are complete - after the reverse observe scan,
    # on the process's `comm` stream - and runs next to the encoder's backward pass; the rest
    # of the arena follows in front of the optimizer (tfutils.py:221-223 reduces after the whole
    # backward).  Same sums, same order of collectives on every rank.
    self.dp_overlap = bool(dp_overlap) and comm is not None
    self._early = None
    # ops_b: launch context (own scratch workspace) of the behaviour phase, so that it
    # can run on its own stream next to the next step's world-model phase (pipeline)
    self.ops_a, self.ops_b = ops, (ops_b if ops_b is not None else ops)
    self._in_b = False
    # tfutils.balance_stats sums of the reward / cont heads (read_metrics)
    self.bal = torch.zeros(2, 7, dtype=torch.float64, device=self.device)
    self.slow_copied = True
    self.stat_b_slots = set()  # metric slots written by the behaviour phase
    self.comm_a, self.comm_b = comm, (comm_b if comm_b is not None else comm)
    # ops2: a second kernel-launch context with its own scratch workspace, used on
    # a side HIP stream to overlap weight-gradient contractions with the
    # latency-bound reverse scan (None: everything runs in program order)
    self.ops2 = ops2
    self.side_stream = (graphs.stream(self.device, 'side')
                        if ops2 is not None and self.device.type == 'cuda' else None)
    # ops_b2: the behaviour phase's own side context: the reward / cont / slow-critic heads of
    # finished time chunks run next to the rest of the (latency-bound) imagination rollout
    self.ops_b2 = ops_b2
    # (off while the pipelined schedule's behaviour phase is captured: there the next step's
    # world-model phase fills the chip next to the rollout, a third stream only competes)
    self.overlap_b = True
    # one-unit output layers folded into the LayerNorm kernels (fold_head; hip.fold_heads)
    self._fold_heads = (bool(spec.cfg.get('hip', {}).get('fold_heads', True)) and
                        hasattr(ops, 'ln_act_bwd_head'))
    self.side_stream_b = (graphs.stream(self.device, 'side_b')
                          if ops_b2 is not None and self.device.type == 'cuda' else None)
    self.dtype = dtype  # float32 in the product; tests may use float64
    # Non-finite gradients.  The reference enables its loss-scale / skip-on-overflow controller only
    # for COMPUTE_DTYPE == float16 (tfutils.py:164 `self._mixed`); in float32 AND bfloat16 it runs
    # check_numerics on the norm and raises (tfutils.py:249).  bf16 has fp32's exponent range, so a
    # non-finite gradient there is real divergence, not a scaling artefact: both precisions of this
    # learner raise FloatingPointError.  `hip.loss_scale: true` opts into the float16 contract
    # (tfutils.py:225-240, 246-260: `*_grad_scale` / `*_grad_overflow` metrics, update skipped
    # instead of an exception, controller state saved / loaded) for callers that want a run to
    # survive an isolated overflow.  The reported `*_grad_scale` is the value AFTER the controller
    # update: the reference puts the tf.Variable itself into the metrics dict (tfutils.py:231),
    # which tf.function reads when it converts its outputs - after the assign (automatic control
    # dependencies order a resource read behind earlier writes).
    self.mixed = bool(spec.cfg.get('hip', {}).get('loss_scale', False))
    self.cfg = cfg = spec.cfg
    self.B, self.T = batch, length
    self.N = batch * length
    self.H = cfg['imag_horizon']
    self.M = (self.H + 1) * self.N
    self.rank, self.world, self.comm = rank, world, comm
    self.Bg = batch * world
    self.Ng = self.N * world
    self.noise_seed = noise_seed
    self._nbytes = 0
    s = spec
    self.D, self.U, self.S, self.A = s.deter, s.units, s.stoch, s.act_dim
    self.G, self.C, self.F = s.groups, s.classes, s.feat
    self.unimix = float(cfg['rssm']['unimix'])
    # several image keys (e.g. image|depth) are concatenated on the channel axis
    # (reference nets.py:220, 274); encoder and decoder must agree on the set
    assert not s.dec_convs or list(s.dec_cnn_keys) == list(s.enc_cnn_keys), \
        (list(s.dec_cnn_keys), s.enc_cnn_keys)
    self.discrete = bool(s.act_discrete)
    assert cfg['critic_type'] == 'vfunction'
    # options of the reference this learner does not implement are rejected, never ignored
    assert cfg.get('transform_rewards', 'off') in ('off', False), cfg['transform_rewards']
    assert cfg.get('train_wm', True), 'train_wm: False'
    assert str(cfg.get('expl_behavior', 'None')) == 'None', cfg['expl_behavior']
    assert not cfg.get('priority_correct', 0.0), 'priority_correct'
    if self.discrete:
      assert cfg['actor_grad_disc'] == 'reinforce' and cfg['actor_dist_disc'] == 'onehot'
    else:
      assert cfg['actor_grad_cont'] == 'backprop' and cfg['actor_dist_cont'] == 'normal'
    assert cfg['actor_return'] in ('gve', 'gae') and cfg['critic_return'] in ('gve', 'gae')
    assert cfg['scorenorm']['impl'] in ('off', 'std')
    for k in ('model_opt', 'actor_opt', 'critic_opt'):
      assert cfg[k]['opt'] == 'adam'
    # ---- parameters
    if groups is not None:
      self.groups = groups  # shared with another Learner (policy runner)
    else:
      self.groups = {
          'model': ParamGroup(s.group('model'), self.device, dtype=dtype),
          'actor': ParamGroup(s.group('actor'), self.device, dtype=dtype),
          'critic': ParamGroup(s.group('critic'), self.device, dtype=dtype),
          'critic_target': ParamGroup(s.group('critic_target'), self.device,
                                      trainable=False, dtype=dtype)}
      init = params if params is not None else specs.init_params(s, seed)
      for g in self.groups.values():
        g.load(init)
    # ---- device scalars
    self.step_ctr = torch.zeros(1, dtype=torch.int64, device=self.device)
    # the behaviour phase keeps its own copy of the step number (it may run while the
    # next world-model phase has already advanced step_ctr)
    self.step_ctr_b = torch.zeros(1, dtype=torch.int64, device=self.device)
    # AutoAdapt state (tfutils.py:427-432): 'mult' / 'prop' start at one, 'fixed' is the
    # configured scale and never updated
    for k in ('wmkl', 'actent'):
      assert cfg[k]['impl'] in ('mult', 'prop', 'fixed'), (k, cfg[k]['impl'])
    init = lambda c: float(c['scale']) if c['impl'] == 'fixed' else 1.0
    self.wmkl_scale = torch.full((1,), init(cfg['wmkl']), dtype=dtype, device=self.device)
    self.actent_scale = torch.full((self.A,), init(cfg['actent']), dtype=dtype, device=self.device)
    self.norm_state = {k: torch.zeros(3, dtype=torch.float64, device=self.device)
                       for k in ('ret', 'score', 'adv')}
    self.norm_os = {k: torch.zeros(2, dtype=dtype, device=self.device)
                    for k in ('ret', 'score', 'adv')}
    self.sc = torch.zeros(3, dtype=dtype, device=self.device)
    self.slow_updates = -1
    self.stat_names = []
    self._stat_queue = []
    self.stat_prereduced = set()  # slots already summed over ranks in-step
    self.stat_sums = torch.zeros(64, 3, dtype=torch.float64, device=self.device)
    self.stat_maxs = torch.zeros(64, 3, dtype=dtype, device=self.device)
    self.actent_sums = torch.zeros(2 * self.A, dtype=torch.float64, device=self.device)
    self.zero_rows = None
    self.plan = graphs.EagerPlan()
    self._build_layers()
    self._build_buffers()

  # ------------------------------------------------------------------ memory

  def zeros(self, *shape, dtype=None):
    t = torch.zeros(*shape, dtype=dtype or self.dtype, device=self.device)
    self._nbytes += t.numel() * t.element_size()
    return t

  def _build_layers(self):
    s, m = self.spec, self.groups['model']
    cfg = self.cfg
    D, S, A = self.D, self.S, self.A
    self.P = P = {}
    for i in range(cfg['encoder']['mlp_layers'] if s.enc_mlp_keys else 0):
      P[f'enc/mlp/dense{i}'] = Lin(m, f'enc/mlp/dense{i}', True)
    P['img_in'] = Lin(m, 'rssm/img_in', True)
    P['img_in_s'] = Lin(m, 'rssm/img_in', True, rows=(0, S))
    P['gru'] = Lin(m, 'rssm/gru_out', True)
    P['gru_h'] = Lin(m, 'rssm/gru_out', True, rows=(0, D))
    P['gru_x'] = Lin(m, 'rssm/gru_out', True, rows=(D, D + self.U))
    self.n_prior = cfg['rssm']['prior_layers']
    for i in range(self.n_prior):
      P[f'img_out_{i}'] = Lin(m, f'rssm/img_out_{i}', True)
    P['img_stats'] = Lin(m, 'rssm/img_stats', False)
    P['obs_out_h'] = Lin(m, 'rssm/obs_out', True, rows=(0, D))
    e0 = D
    self.cnn_feat = 0
    if s.enc_convs or s.enc_res:
      if s.enc_res:   # the residual encoder ends in Linear(1024) (reference nets.py:348)
        self.cnn_feat = s.enc_res.units
      else:
        last = s.enc_convs[-1]
        self.cnn_feat = last.h_small * last.h_small * last.c_small
      P['obs_out_cnn'] = Lin(m, 'rssm/obs_out', True, rows=(e0, e0 + self.cnn_feat))
      e0 += self.cnn_feat
    if s.enc_mlp_keys:
      P['obs_out_mlp'] = Lin(m, 'rssm/obs_out', True, rows=(e0, D + s.embed))
    P['obs_stats'] = Lin(m, 'rssm/obs_stats', False)

    def head(name, group, cfgkey, outs):
      g = self.groups[group]
      layers = [Lin(g, f'{name}/dense{i}', True)
                for i in range(cfg[cfgkey]['layers'])]
      return layers, [Lin(g, f'{name}/{o}', False) for o in outs]
    self.heads = {
        'reward': head('reward', 'model', 'reward_head', ['dist_out/out']),
        'cont': head('cont', 'model', 'cont_head', ['dist_out/out']),
        'actor': head('actor', 'actor', 'actor',
                      ['dist_out/out'] if s.act_discrete else
                      ['dist_out/out', 'dist_out/std']),
        'critic': head('critic', 'critic', 'critic', ['dist_out/out']),
        'critic_target': head('critic_target', 'critic_target', 'critic',
                              ['dist_out/out'])}
    if not cfg['slow_target']:  # VFunction: target_net = net (reference agent.py:395-396)
      self.heads['critic_target'] = self.heads['critic']
    if s.dec_mlp_keys:
      g = m
      n = cfg['decoder']['mlp_layers']
      self.heads['dec_mlp'] = (
          [Lin(g, f'dec/mlp/dense{i}', True) for i in range(n)],
          [Lin(g, f'dec/mlp/dist_{k}/out', False) for k in s.dec_mlp_keys])

  def _head_acts(self, name, rows, grads=True):
    layers, outs = self.heads[name]
    return ([Act(self, rows, l.units, True, grads) for l in layers],
            [Act(self, rows, o.units, False, grads) for o in outs])

  def _build_buffers(self):
    s = self.spec
    B, T, N, H, M = self.B, self.T, self.N, self.H, self.M
    D, U, S, A, F, G = self.D, self.U, self.S, self.A, self.F, self.G
    z = self.zeros
    b = self.b = {}
    # ---- inputs (wire format, batch-major)
    if s.enc_cnn_keys:
      hw, c = s.image_hw, s.image_c
      b['image'] = z(N, hw, hw, c, dtype=torch.uint8)
    if s.enc_mlp_keys:
      b['vec_in'] = z(N, s.enc_mlp_in)
    if s.dec_mlp_keys:
      b['vec_tgt'] = {k: z(N, int(np.prod(v))) for k, v in s.dec_mlp_keys.items()}
    b['action'] = z(N, A)
    b['reward'] = z(N)
    b['is_first'] = z(N, dtype=torch.uint8)
    b['is_terminal'] = z(N, dtype=torch.uint8)
    b['first'] = z(N)
    b['cont'] = z(N)
    b['cont_b'] = z(N)  # snapshot for the behaviour phase
    b['loss_total'] = z(N)  # summed, scaled world-model loss map (its std is a metric)
    # ---- noise
    b['u_prior'] = z(B, T, G)      # batch-major: consumed in bulk after the scan
    b['u_post'] = z(T, B, G)
    b['u_img'] = z(max(H, 1), N, G)
    b['eps'] = z(H + 1, N, A)
    if self.discrete:
      b['u_act'] = z(H + 1, N, 1)
      b['alogit'] = z(self.M, A)      # normalised actor log-probs
      b['dalogit'] = z(self.M, A)
      b['ent_norm'] = z(self.M)
    # ---- encoder
    self.enc_act = []
    for cl in s.enc_convs:
      rows = N * cl.h_small * cl.h_small
      self.enc_act.append(dict(
          z=z(N, cl.h_small, cl.h_small, cl.c_small),
          out=z(N, cl.h_small, cl.h_small, cl.c_small),
          stats=z(rows, 2), dz=z(N, cl.h_small, cl.h_small, cl.c_small),
          dout=z(N, cl.h_small, cl.h_small, cl.c_small)))
    self.enc_res = self._res_buffers(s.enc_res, True) if s.enc_res else None
    self.enc_mlp_act = [Act(self, N, self.P[f'enc/mlp/dense{i}'].units, True)
                        for i in range(self.cfg['encoder']['mlp_layers']
                                       if s.enc_mlp_keys else 0)]
    # ---- observe scan (batch-major [B,T,...] buffers)
    b['post'] = z(N, F)            # [deter | stoch] of the posterior
    # GRU input [deter_prev | img_in output] and its gradient as one buffer each:
    # the cell's matmul is a single K = D+U GEMM in both directions
    b['gin'] = z(N, D + U)
    b['dgin'] = z(N, D + U)
    b['hprev'] = b['gin'][:, :D]
    b['xin'] = z(N, S + A)
    self.a_img_in = Act(self, N, U, True, out=b['gin'][:, D:], dout=b['dgin'][:, D:])
    b['z3'] = z(N, 3 * D)
    b['gstats'] = z(N, 2)
    b['dz3'] = z(N, 3 * D)
    b['dy3'] = z(N, 3 * D)
    b['dhprev'] = b['dgin'][:, :D]
    b['dxin_s'] = z(N, S)
    self.a_img_out = [Act(self, N, U, True) for _ in range(self.n_prior)]
    self.a_img_stats = Act(self, N, S, False)
    self.a_obs_out = Act(self, N, U, True)
    self.a_obs_stats = Act(self, N, S, False)
    b['prior_logit'] = z(N, S)
    b['post_logit'] = z(N, S)
    b['prior_stoch'] = z(N, S)
    b['dprior_logit'] = z(N, S)
    b['dpost_logit'] = z(N, S)
    b['dfeat'] = z(N, F)
    b['kl'] = z(N)
    b['ent_post'] = z(N)
    b['ent_prior'] = z(N)
    # initial(): one row
    b['init_deter'] = z(1, D)
    b['init_stoch'] = z(1, S)
    b['init_logit'] = z(1, S)
    b['dinit_deter'] = z(1, D)
    self.a_init_out = [Act(self, 1, U, True, grads=False) for _ in range(self.n_prior)]
    self.a_init_stats = Act(self, 1, S, False, grads=False)
    # carry state between train calls
    b['carry'] = z(B, F)
    # fused observe scan (csrc/scan.hip): transposed bf16 weight planes + barrier words
    self.fused_scan_bwd = False
    self.fused_scan = (bool(self.cfg.get('hip', {}).get('fused_scan', True)) and
                       self.dtype == torch.float32 and hasattr(self.ops, 'observe_scan_fwd') and
                       self.ops.observe_scan_supported(B, D, U, G, self.C, A))
    if self.fused_scan:
      xkp = (S + A + 31) // 32 * 32
      i16 = lambda n: torch.zeros(n, dtype=torch.int16, device=self.device)
      self.scan_w = [(self.P['img_in'].W, i16(3 * U * xkp), xkp),
                     (self.P['gru'].W, i16(3 * 3 * D * (D + U)), D + U),
                     (self.P['obs_out_h'].W, i16(3 * U * D), D),
                     (self.P['obs_stats'].W, i16(3 * S * U), U)]
      self.fused_scan_bwd = (bool(self.cfg.get('hip', {}).get('fused_scan_bwd', True)) and
                             hasattr(self.ops, 'observe_scan_bwd') and
                             self.ops.observe_scan_bwd_supported(B, D, U, G, self.C))
      self.scan_wb = [(self.P['obs_stats'].W, i16(3 * U * S)), (self.P['obs_out_h'].W, i16(3 * D * U)),
                      (self.P['gru'].W, i16(3 * (D + U) * 3 * D)), (self.P['img_in_s'].W, i16(3 * S * U))]
      self.scan_sync = torch.zeros(getattr(self.ops, 'SCAN_SYNC_WORDS', 1088),
                                   dtype=torch.int32, device=self.device)   # counter, error word, debug stamps, row-block counters
      self.scan_idx = torch.zeros((N + B + 1) * G, dtype=torch.int32, device=self.device)
    # World-model forward as two batch halves, software-pipelined (opt-in, hip.split_fwd /
    # DD_SPLIT_FWD=1): the latency-bound observe scan of one half (64 workgroups) on the main
    # stream next to the GPU-filling encoder of the other half / decoder + heads of the first on
    # the side stream.  The rows of a minibatch never interact in the forward pass (tfagent.py:
    # 105-116 shards the same axis), every half writes its own row range of the full-size buffers,
    # so the backward pass is unchanged; noise is keyed by global row, so the draws are too (the
    # whole parity suite passes with it on).  MEASURED SLOWER (round 4, same box, alternating
    # runs: 36.26 / 36.09 ms with, 35.84 / 35.76 ms without): the scan's 64 persistent workgroups
    # need > 256 registers per lane and are only placed as the other stream's convolution grids
    # drain, so the two halves of the pipeline barely overlap while the half-size convolutions
    # are less efficient.  Off by default.
    self.split_fwd = (bool(int(os.environ.get('DD_SPLIT_FWD', self.cfg.get('hip', {}).get('split_fwd', False)))) and self.fused_scan and
                      self.ops2 is not None and self.side_stream is not None and B >= 2 and
                      not s.enc_res and not s.dec_res)
    if self.split_fwd:
      bm = (B + 1) // 2
      self.halves = [(0, bm), (bm, B)]
      self.u_post_h = [z(T, b1 - b0, G) for b0, b1 in self.halves]
      self.scan_idx_h = [torch.zeros(((b1 - b0) * T + (b1 - b0) + 1) * G, dtype=torch.int32, device=self.device)
                         for b0, b1 in self.halves]
    # ---- heads on the posterior
    self.acts_wm = {k: self._head_acts(k, N) for k in ('reward', 'cont')}
    if s.dec_mlp_keys:
      self.acts_wm['dec_mlp'] = self._head_acts('dec_mlp', N)
    b['loss_reward'] = z(N)
    b['loss_cont'] = z(N)
    b['loss_vec'] = {k: z(N) for k in s.dec_mlp_keys}
    self.dec_act = []
    for cl in s.dec_convs:
      shp = (N, cl.h_big, cl.h_big, cl.c_big)
      d = dict(z=z(*shp), dz=z(*shp))
      if cl.norm:
        d.update(out=z(*shp), stats=z(N * cl.h_big * cl.h_big, 2), dout=z(*shp))
      self.dec_act.append(d)
    self.dec_res = self._res_buffers(s.dec_res, False) if s.dec_res else None
    if s.dec_convs:
      b['loss_image'] = {k: z(N) for k in s.dec_cnn_keys}
      c0 = s.dec_convs[0]
      b['bias_fold'] = z(64 * s.dec_convs[-1].c_big)
      b['bias_tiled'] = z(c0.k * c0.k * c0.c_big)
      b['dbias_tiled'] = z(c0.k * c0.k * c0.c_big)
    # ---- imagination (time-major [H+1, N, ...])
    # Row width of the trajectory: F + A, padded with zero columns to a multiple of four floats
    # when it is not one (one-hot action spaces: xarm / ur5 have A = 6, F + A = 1542): every row of a
    # column slice then starts on a 16-byte boundary and the contractions over traj[:, :F] (all heads,
    # the actor's first layer, their weight gradients) take the branch-free 16-byte loaders instead
    # of the bounds-checked scalar ones (5.2 ms of the 30.5 ms of kernel time per step at the xarm
    # shard ran on those, profiles/r04_rocprof_kernel_stats_xarm_shard.csv).  The fused imagination
    # kernels index rows by F + A themselves, so shapes they cover keep the exact width.
    ca = self.cfg['actor']
    self.fused_imag = (bool(self.cfg.get('hip', {}).get('fused_imag', True)) and
                       self.dtype == torch.float32 and H >= 1 and
                       hasattr(self.ops, 'imagine_rollout_oh_fwd' if self.discrete else 'imagine_rollout_fwd') and
                       self.ops.imagine_rollout_supported(D, U, G, self.C, A, ca['units'], ca['layers'],
                                                          self.n_prior, self.discrete))
    self._pipelined_capture = False   # set while capture_pipeline records the phase plans
    self.stamps = (torch.zeros(16 * 9, dtype=torch.int64, device=self.device)
                   if os.environ.get('DD_STAMPS') == '1' and hasattr(self.ops, 'stamp') else None)
    self._fuse_img_ln = bool(self.cfg.get('hip', {}).get('fuse_image_ln', True)) and self.dtype == torch.float32
    # (the one-hot kernel takes the row width as an argument and keeps the padded rows)
    self.TW = F + A if (self.fused_imag and not self.discrete) else (F + A + 3) // 4 * 4
    W = self.TW
    b['traj'] = z(H + 1, N, W)     # [deter | stoch | action | zero padding]
    # the stoch columns of both feature matrices are one-hot classes (nets.py:88-97): exact in one
    # bfloat16 plane, so contractions over them need three plane products, not six (dd_gemm_f32_x)
    if hasattr(self.ops, 'mark_exact') and bool(self.cfg.get('hip', {}).get('exact_planes', True)):
      self.ops.mark_exact(b['post'], D, F)
      self.ops.mark_exact(b['traj'], D, F)
    b['dtraj'] = z(H + 1, N, W)
    self.ai_img_in = Act(self, H * N, U, True)
    b['iz3'] = z(H * N, 3 * D)
    b['igstats'] = z(H * N, 2)
    b['idz3'] = z(N, 3 * D)
    b['idy3'] = z(N, 3 * D)
    b['idh'] = z(N, D)
    self.ai_img_out = [Act(self, H * N, U, True) for _ in range(self.n_prior)]
    self.ai_img_stats = Act(self, H * N, S, False)
    b['ilogit'] = z(N, S)
    self.acts_im = {
        'actor': self._head_acts('actor', M),
        'reward': self._head_acts('reward', M),
        'cont': self._head_acts('cont', M),
        'critic': self._head_acts('critic', H * N),
        'critic_target': self._head_acts('critic_target', M)}
    # fused imagination rollout (csrc/imag.hip): fragment-major bf16 weight planes
    self.fused_imag_bwd = False
    if self.fused_imag:
      i16 = lambda k, n: torch.zeros(3 * k * ((n + 15) // 16 * 16), dtype=torch.int16, device=self.device)
      layers, outs = self.heads['actor']
      pl = self.imag_planes = {}
      pl['actor0'] = (layers[0].W[:D], i16(D, ca['units']), 0)
      for i in range(1, ca['layers']):
        pl[f'actor{i}'] = (layers[i].W, i16(ca['units'], ca['units']), 0)
      if self.discrete:
        pl['head'] = (outs[0].W, i16(ca['units'], A), 0)
      else:
        head = i16(ca['units'], 2 * A)
        pl['head_m'] = (outs[0].W, head, 0)
        pl['head_s'] = (outs[1].W, head, A)
      pl['gru'] = (self.P['gru'].W, i16(D + U, 3 * D), 0)
      for i in range(self.n_prior):
        pl[f'img_out{i}'] = (self.P[f'img_out_{i}'].W, i16(D if i == 0 else U, U), 0)
      pl['stats'] = (self.P['img_stats'].W, i16(U, S), 0)
      # transposed caches of the reverse pass (dd_imag_wprep_t: W [n, K] -> operand [K, n])
      # (one-hot actions: actor_grad reinforce has no reverse pass through the rollout)
      self.fused_imag_bwd = (bool(self.cfg.get('hip', {}).get('fused_imag_bwd', True)) and
                             hasattr(self.ops, 'imagine_rollout_bwd') and not self.discrete)
      plt = self.imag_planes_t = {}
      if self.fused_imag_bwd:
        plt['stats'] = (self.P['img_stats'].W, i16(S, U))            # [U, S] -> K = S, n = U
        for i in range(self.n_prior):
          Wl = self.P[f'img_out_{i}'].W
          plt[f'img_out{i}'] = (Wl, i16(Wl.shape[1], Wl.shape[0]))
        plt['gru'] = (self.P['gru'].W, i16(3 * D, D + U))             # [D+U, 3D] -> K = 3D, n = D+U
        plt['img_in'] = (self.P['img_in'].W, i16(U, S + A))           # [S+A, U] -> K = U, n = S+A
    for k in ('value', 'cont', 'weight', 'value2', 'ent_row'):
      b['i_' + k] = z(M)
    for k in ('reward', 'ret', 'ret2', 'diff', 'crit_loss', 'critic', 'actor_loss',
              'dret', 'dbase'):
      b['i_' + k] = z(max(H * N, 1))
    b['dom'] = z(M, A)
    b['dos'] = z(M, A)
    self.zero_rows = z(max(N, B))

  # ----------------------------------------------------------- small helpers

  def _res_buffers(self, net, encoder):
    """Activations of a residual encoder / decoder (spec.ResNet).  Per block: `a1` = act(LN(x))
    and its statistics, `za` = first convolution, `a2` = act(LN(za)), `o` = block output, `dx` =
    gradient at the block input (the next-lower block's output gradient).  Per stage: `do` =
    gradient at the last block's output; encoder `x` = pooled stage input, decoder `up` = the
    stage output repeated 2x.  t1 / t2 / t3: gradient scratch shared by all blocks."""
    z, N = self.zeros, self.N
    R = dict(stages=[], do=[], x=[], up=[])
    maxd = maxc = 0
    for blocks in net.stages:
      acts = []
      for blk in blocks:
        sc, sd = (N, blk.h, blk.h, blk.cin), (N, blk.h, blk.h, blk.depth)
        rows = N * blk.h * blk.h
        acts.append(dict(a1=z(*sc), st_a=z(rows, 2), za=z(*sd), a2=z(*sd), st_b=z(rows, 2),
                         o=z(*sd), dx=z(*sc)))
        maxd, maxc = max(maxd, rows * blk.depth), max(maxc, rows * blk.cin)
      R['stages'].append(acts)
      first, last = blocks[0], blocks[-1]
      R['do'].append(z(N, last.h, last.h, last.depth))
      if encoder:
        R['x'].append(z(N, first.h, first.h, first.cin))
      else:
        R['up'].append(z(N, 2 * last.h, 2 * last.h, last.depth))
    R['t1'], R['t2'], R['t3'] = z(maxd), z(maxd), z(maxc)
    R['bias01'] = z(max(blk.depth for blocks in net.stages for blk in blocks))
    if encoder:
      R['zin'] = z(N, net.hw, net.hw, net.depth)
      R['dzin'] = z(N, net.hw, net.hw, net.depth)
      R['emb'], R['demb'] = z(N, net.units), z(N, net.units)
    else:
      R['xin'] = z(N, net.feat_h, net.feat_h, net.feat_c)
      R['dup'] = z(N, net.hw, net.hw, net.depth)
    return R

  def stat(self, name, x, count=None):
    """Register batch statistics of a vector into a metric slot.  The reduction itself is queued:
    flush_stats() runs all queued ones as ONE launch (dd_reduce_stats_multi; they are ~17
    independent single-block reductions per step, 13 us each when launched one by one).  `x` must
    stay untouched until then; a caller that consumes the sums right away flushes right away."""
    if name not in self.stat_names:
      self.stat_names.append(name)
    k = self.stat_names.index(name)
    if self._in_b:
      self.stat_b_slots.add(k)
    self._stat_queue.append((x, self.stat_sums[k], self.stat_maxs[k]))
    return k

  def flush_stats(self):
    queue, self._stat_queue = self._stat_queue, []
    if not queue:
      return
    if hasattr(self.ops, 'reduce_stats_multi'):
      self.ops.reduce_stats_multi(queue)
    else:
      for x, sums, maxs in queue:
        self.ops.reduce_stats(x, sums, maxs)

  def fork(self, stream=None):
    """Context: the body runs on the side stream, ordered after everything issued
    so far on the main stream (capturable: becomes a parallel graph branch)."""
    stream = stream or self.side_stream
    if stream is None:
      return contextlib.nullcontext()
    stream.wait_stream(torch.cuda.current_stream(self.device))
    return torch.cuda.stream(stream)

  def join(self, stream=None):
    stream = stream or self.side_stream
    if stream is not None:
      torch.cuda.current_stream(self.device).wait_stream(stream)

  def early_range(self):
    """[start, stop) of the model gradient arena that is complete once the head / decoder
    weight gradients are: the longest run of consecutive dec / reward / cont tensors in arena
    order (with weight decay: their kernels, which lie together in the decayed prefix; the few
    scale / bias vectors of those modules in the other part go with the rest)."""
    g = self.groups['model']
    mods = ('dec', 'reward', 'cont')
    specs = list(g.specs)            # arena order (offsets ascending)
    best, i = None, 0
    while i < len(specs):
      if specs[i].name.split('/')[0] not in mods:
        i += 1
        continue
      j = i
      while j < len(specs) and specs[j].name.split('/')[0] in mods:
        j += 1
      start = g.offset[specs[i].name]
      stop = g.offset[specs[j].name] if j < len(specs) else g.gflat.numel()   # (incl. zero padding)
      if best is None or stop - start > best[1] - best[0]:
        best = (start, stop)
      i = j
    return best

  def allreduce_early(self):
    """Issue the all-reduce of the early range on the comm stream (a graph cut point)."""
    rng = self.early_range()
    if rng is None:
      return
    g, comm = self.groups['model'], self.comm
    view = g.gflat[rng[0]:rng[1]]
    cs = graphs.stream(self.device, 'comm') if self.device.type == 'cuda' else None
    def issue():
      if cs is None:
        comm.allreduce_sum(view)
        return
      cs.wait_stream(torch.cuda.current_stream(self.device))
      with torch.cuda.stream(cs):
        comm.allreduce_sum(view)
    self.plan.cut(issue)
    self._early = (rng, cs)

  def allreduce(self, t):
    """Sum over data-parallel ranks (RCCL); a graph cut point."""
    if self.comm is not None:
      comm = self.comm  # the phase's communicator at issue time (the cut runs at replay)
      if os.environ.get('DD_DIST_CAPTURE') == '1':  # experiment: collective inside the graph
        comm.allreduce_sum(t)
      else:
        self.plan.cut(lambda: comm.allreduce_sum(t))

  def lin_fwd(self, P, A, x, sel=None, defer=False, head=None):
    """Linear (+ LayerNorm + ELU).  The GEMM of a normed layer leaves a split-K sum to
    the LayerNorm kernel (one launch less); for a plain layer `defer=True` returns
    (z, pending sum) for a consumer that takes `pre=`.  head = (P_out, A_out): the one-unit
    output layer behind this layer is evaluated inside the LayerNorm kernel (fold_head)."""
    sel = sel or (lambda t: t)
    zv = sel(A.z)
    if not P.norm:
      pre = self.ops.gemm(x, P.W, zv, bias=P.bias, defer=defer)
      return (zv, pre) if defer else zv
    pre = self.ops.gemm(x, P.W, zv, defer=True)
    if head is not None:
      Po, Ao = head
      self.ops.ln_act_fwd(zv, P.gamma, P.beta, sel(A.out), sel(A.stats), True, pre=pre,
                          head=(Po.W, Po.bias, sel(Ao.z)))
    else:
      self.ops.ln_act_fwd(zv, P.gamma, P.beta, sel(A.out), sel(A.stats), True, pre=pre)
    return sel(A.out)

  def fold_head(self, layers, outs):
    """The scalar heads (reward / cont / critic: MLP -> one-unit output layer, nets.py:428-492):
    the output layer as a 1-column contraction is a GEMV at ~1 TFLOP/s and its data gradient an
    outer product through HBM; both are folded into the last hidden layer's LayerNorm kernels
    (dd_ln_act_fwd_head / dd_ln_act_bwd_head)."""
    return (self._fold_heads and len(outs) == 1 and outs[0].units == 1 and len(layers) >= 1 and
            layers[-1].norm and layers[-1].units % 4 == 0 and layers[-1].units <= 1024)

  def lin_bwd(self, P, A, x, sel=None, dx=None, dx_beta=0.0, params=True,
              defer=None):
    """Gradient w.r.t. the layer output is in A.dout.  Writes parameter
    gradients (params=True, bulk over the selected rows) and dx.  With a
    `defer` list the weight / bias gradient contractions are queued as
    closures f(ops) instead of being launched (they only need dz)."""
    sel = sel or (lambda t: t)
    ops = self.ops
    if P.norm:
      dz = sel(A.dz)
      ops.ln_act_bwd(sel(A.dout), sel(A.z), sel(A.out), sel(A.stats), P.gamma,
                     dz, P.dgamma if params else None,
                     P.dbeta if params else None, False, True, beta=P.beta)
    else:
      dz = sel(A.dout)
      if params:
        if defer is not None:
          defer.append(lambda o, dz=dz, P=P: o.col_sum(dz, P.dbias))
        else:
          ops.col_sum(dz, P.dbias)
    if params:
      if defer is not None:
        defer.append(lambda o, x=x, dz=dz, P=P: o.gemm(x, dz, P.dW, ta=True))
      else:
        ops.gemm(x, dz, P.dW, ta=True)
    if dx is not None:
      ops.gemm(dz, P.W, dx, tb=True, beta=dx_beta)
    return dz

  def mlp_fwd(self, layers, acts, x, sel=None):
    for P, A in zip(layers, acts):
      x = self.lin_fwd(P, A, x, sel)
    return x

  def mlp_bwd(self, layers, acts, x0, sel=None, dx=None, dx_beta=0.0,
              params=True, defer=None):
    """Backward through a DenseLN stack; the top layer's A.dout must hold the
    incoming gradient.  x0 is the stack input."""
    sel = sel or (lambda t: t)
    for i in reversed(range(len(layers))):
      xin = x0 if i == 0 else sel(acts[i - 1].out)
      tgt = dx if i == 0 else sel(acts[i - 1].dout)
      self.lin_bwd(layers[i], acts[i], xin, sel, tgt,
                   dx_beta if i == 0 else 0.0, params, defer)

  def head_fwd(self, name, acts, x, sel=None):
    layers, outs = self.heads[name]
    la, oa = acts
    if self.fold_head(layers, outs):
      s_ = sel or (lambda t: t)
      h = self.mlp_fwd(layers[:-1], la[:-1], x, sel)
      self.lin_fwd(layers[-1], la[-1], h, sel, head=(outs[0], oa[0]))
      return [s_(oa[0].z)]
    h = self.mlp_fwd(layers, la, x, sel)
    return [self.lin_fwd(P, A, h, sel) for P, A in zip(outs, oa)]

  def head_bwd(self, name, acts, x, sel=None, dx=None, dx_beta=0.0,
               params=True, defer=None):
    """Output-layer gradients must be in the out-acts' .dout."""
    sel = sel or (lambda t: t)
    layers, outs = self.heads[name]
    la, oa = acts
    top = sel(la[-1].out)
    if self.fold_head(layers, outs):
      # output layer: its parameter gradients as before (bias: column sum, kernel: top^T dz); its
      # data gradient dz x W^T is never formed - the last hidden layer's LayerNorm backward takes
      # dz and W and builds it in registers
      Po, Ao, Pl, Al = outs[0], oa[0], layers[-1], la[-1]
      dzo = sel(Ao.dout)
      if params:
        if defer is not None:
          defer.append(lambda o, dzo=dzo, Po=Po: o.col_sum(dzo, Po.dbias))
          defer.append(lambda o, top=top, dzo=dzo, Po=Po: o.gemm(top, dzo, Po.dW, ta=True))
        else:
          self.ops.col_sum(dzo, Po.dbias)
          self.ops.gemm(top, dzo, Po.dW, ta=True)
      dzl = sel(Al.dz)
      self.ops.ln_act_bwd_head(dzo, Po.W, sel(Al.z), sel(Al.out), sel(Al.stats), Pl.gamma, dzl,
                               Pl.dgamma if params else None, Pl.dbeta if params else None,
                               False, True, beta=Pl.beta)
      n = len(layers)
      xin = x if n == 1 else sel(la[n - 2].out)
      tgt = dx if n == 1 else sel(la[n - 2].dout)
      if params:
        if defer is not None:
          defer.append(lambda o, xin=xin, dzl=dzl, Pl=Pl: o.gemm(xin, dzl, Pl.dW, ta=True))
        else:
          self.ops.gemm(xin, dzl, Pl.dW, ta=True)
      if tgt is not None:
        self.ops.gemm(dzl, Pl.W, tgt, tb=True, beta=dx_beta if n == 1 else 0.0)
      if n > 1:
        self.mlp_bwd(layers[:-1], la[:-1], x, sel, dx, dx_beta, params, defer)
      return
    for j, (P, A) in enumerate(zip(outs, oa)):
      self.lin_bwd(P, A, top, sel, sel(la[-1].dout), 0.0 if j == 0 else 1.0,
                   params, defer)
    self.mlp_bwd(layers, la, x, sel, dx, dx_beta, params, defer)

  # ----------------------------------------------------------------- encoder

  # residual blocks (reference nets.py:351-358 / 384-391):
  #   o = skip(x) + 0.1 * conv_b(act(LN(conv_a(act(LN(x))))))      (both convolutions 3x3 SAME
  #   with bias, pre-activation nets.py:510-513; skip = 1x1 convolution without bias when the
  #   channel count changes, else the identity)

  def res_block_fwd(self, blk, a, x, R):
    ops, m = self.ops, self.groups['model']
    nm, C, Dp = blk.name, blk.cin, blk.depth
    ops.ln_act_fwd(x.view(-1, C), m.p[f'{nm}a/norm/scale'], m.p[f'{nm}a/norm/bias'],
                   a['a1'].view(-1, C), a['st_a'], True)
    ops.conv_same(a['a1'], m.p[f'{nm}a/kernel'], m.p[f'{nm}a/bias'], a['za'], 3)
    ops.ln_act_fwd(a['za'].view(-1, Dp), m.p[f'{nm}b/norm/scale'], m.p[f'{nm}b/norm/bias'],
                   a['a2'].view(-1, Dp), a['st_b'], True)
    # the skip path goes into `o` first; the second convolution accumulates 0.1 * (conv + bias)
    # onto it in its epilogue (bias pre-scaled: the epilogue adds it unscaled)
    if blk.skip:
      ops.gemm(x.view(-1, C), m.p[f'{nm}s/kernel'].view(C, Dp), a['o'].view(-1, Dp))
    else:
      ops.copy2d(x.view(-1, C), a['o'].view(-1, Dp))
    b01 = R['bias01'][:Dp]
    ops.scalar_mul(b01, m.p[f'{nm}b/bias'], None, 0.1)
    ops.conv_same(a['a2'], m.p[f'{nm}b/kernel'], b01, a['o'], 3, alpha=0.1, beta=1.0)

  def res_block_bwd(self, blk, a, x, do, R):
    """do: gradient at the block output; writes a['dx'] (gradient at x) and the block's
    parameter gradients."""
    ops, m = self.ops, self.groups['model']
    nm, C, Dp = blk.name, blk.cin, blk.depth
    nd, nc = do.numel(), x.numel()
    t1, t2 = R['t1'][:nd].view(do.shape), R['t2'][:nd].view(do.shape)
    t3 = R['t3'][:nc].view(x.shape)
    gb = m.g[f'{nm}b/bias']
    ops.col_sum(do.view(-1, Dp), gb)
    ops.scalar_mul(gb, gb, None, 0.1)
    ops.conv_same_wgrad(a['a2'], do, m.g[f'{nm}b/kernel'], 3, alpha=0.1)
    ops.conv_same_bwd(do, m.p[f'{nm}b/kernel'], t2, 3, alpha=0.1)
    ops.ln_act_bwd(t2.view(-1, Dp), a['za'].view(-1, Dp), a['a2'].view(-1, Dp), a['st_b'],
                   m.p[f'{nm}b/norm/scale'], t1.view(-1, Dp), m.g[f'{nm}b/norm/scale'],
                   m.g[f'{nm}b/norm/bias'], False, True, m.g[f'{nm}a/bias'],
                   beta=m.p[f'{nm}b/norm/bias'])
    ops.conv_same_wgrad(a['a1'], t1, m.g[f'{nm}a/kernel'], 3)
    ops.conv_same_bwd(t1, m.p[f'{nm}a/kernel'], t3, 3)
    ops.ln_act_bwd(t3.view(-1, C), x.view(-1, C), a['a1'].view(-1, C), a['st_a'],
                   m.p[f'{nm}a/norm/scale'], a['dx'].view(-1, C), m.g[f'{nm}a/norm/scale'],
                   m.g[f'{nm}a/norm/bias'], False, True, None, beta=m.p[f'{nm}a/norm/bias'])
    if blk.skip:
      ops.gemm(x.view(-1, C), do.view(-1, Dp), m.g[f'{nm}s/kernel'].view(C, Dp), ta=True)
      ops.gemm(do.view(-1, Dp), m.p[f'{nm}s/kernel'].view(C, Dp), a['dx'].view(-1, C),
               tb=True, beta=1.0)
    else:
      ops.axpy(do, 1.0, None, a['dx'])

  def encoder_res_fwd(self):
    """ImageEncoderResnet (reference nets.py:337-349); the image's `/255` (agent.py:129-130) is
    fused into the 'in' convolution's loader."""
    s, ops, b, N = self.spec, self.ops, self.b, self.N
    m, net, R = self.groups['model'], self.spec.enc_res, self.enc_res
    ops.conv_same(b['image'], m.p['enc/cnn/in/kernel'], m.p['enc/cnn/in/bias'], R['zin'], 3,
                  1.0 / 255.0)
    prev = R['zin']
    for blocks, acts, xin in zip(net.stages, R['stages'], R['x']):
      ops.pool2(prev, xin, 0.25)
      x = xin
      for blk, a in zip(blocks, acts):
        self.res_block_fwd(blk, a, x, R)
        x = a['o']
      prev = x
    ops.gemm(prev.view(N, -1), m.p['enc/cnn/out/kernel'], R['emb'], bias=m.p['enc/cnn/out/bias'])

  def encoder_res_bwd(self):
    """Consumes enc_res['demb'] (gradient of the 1024-wide embedding)."""
    s, ops, b, N = self.spec, self.ops, self.b, self.N
    m, net, R = self.groups['model'], self.spec.enc_res, self.enc_res
    top = R['stages'][-1][-1]['o'].view(N, -1)
    ops.gemm(top, R['demb'], m.g['enc/cnn/out/kernel'], ta=True)
    ops.col_sum(R['demb'], m.g['enc/cnn/out/bias'])
    ops.gemm(R['demb'], m.p['enc/cnn/out/kernel'], R['do'][-1].view(N, -1), tb=True)
    for i in reversed(range(len(net.stages))):
      blocks, acts, do = net.stages[i], R['stages'][i], R['do'][i]
      for j in reversed(range(len(blocks))):
        x = acts[j - 1]['o'] if j > 0 else R['x'][i]
        self.res_block_bwd(blocks[j], acts[j], x, do, R)
        do = acts[j]['dx']
      # gradient of the 2x2 average pooling in front of the stage
      ops.repeat2(do, R['do'][i - 1] if i > 0 else R['dzin'], 0.25)
    C = net.depth
    ops.col_sum(R['dzin'].view(-1, C), m.g['enc/cnn/in/bias'])
    ops.conv_same_wgrad(b['image'], R['dzin'], m.g['enc/cnn/in/kernel'], 3, 1.0 / 255.0)

  def decoder_res_fwd(self, feat):
    """ImageDecoderResnet (reference nets.py:370-382) up to the pre-sigmoid image."""
    s, ops, N = self.spec, self.ops, self.N
    m, net, R = self.groups['model'], self.spec.dec_res, self.dec_res
    ops.gemm(feat, m.p['dec/cnn/in/kernel'], R['xin'].view(N, -1), bias=m.p['dec/cnn/in/bias'])
    x = R['xin']
    for blocks, acts, up in zip(net.stages, R['stages'], R['up']):
      for blk, a in zip(blocks, acts):
        self.res_block_fwd(blk, a, x, R)
        x = a['o']
      ops.repeat2(x, up, 1.0)
      x = up
    ops.conv_same(x, m.p['dec/cnn/out/kernel'], m.p['dec/cnn/out/bias'], self.dec_act[-1]['z'], 3)

  def decoder_res_bwd(self, feat, dfeat, beta):
    s, ops, N = self.spec, self.ops, self.N
    m, net, R = self.groups['model'], self.spec.dec_res, self.dec_res
    last, cl = self.dec_act[-1], s.dec_convs[-1]
    self._image_bias_grad(ops, last, cl.c_big, cl)
    ops.conv_same_wgrad(R['up'][-1], last['dz'], m.g['dec/cnn/out/kernel'], 3)
    ops.conv_same_bwd(last['dz'], m.p['dec/cnn/out/kernel'], R['dup'], 3)
    dup = R['dup']
    for i in reversed(range(len(net.stages))):
      blocks, acts, do = net.stages[i], R['stages'][i], R['do'][i]
      ops.pool2(dup, do, 1.0)   # gradient of the 2x repetition
      for j in reversed(range(len(blocks))):
        x = acts[j - 1]['o'] if j > 0 else (R['up'][i - 1] if i > 0 else R['xin'])
        self.res_block_bwd(blocks[j], acts[j], x, do, R)
        do = acts[j]['dx']
      dup = do
    dxin = dup.view(N, -1)
    ops.gemm(feat, dxin, m.g['dec/cnn/in/kernel'], ta=True)
    ops.col_sum(dxin, m.g['dec/cnn/in/bias'])
    ops.gemm(dxin, m.p['dec/cnn/in/kernel'], dfeat, tb=True, beta=beta)

  def encoder_fwd(self, rows=None):
    """rows = (r0, r1): only those minibatch rows (images / vectors r0 .. r1 - 1 of the N)."""
    s, ops, b = self.spec, self.ops, self.b
    m = self.groups['model']
    r0, r1 = rows if rows is not None else (0, self.N)
    R = lambda t: t[r0:r1]
    x = b.get('image')
    x = R(x) if x is not None else None
    if s.enc_res:
      assert rows is None
      self.encoder_res_fwd()
    for i, (cl, a) in enumerate(zip(s.enc_convs, self.enc_act)):
      C, px = cl.c_small, cl.h_small * cl.h_small
      if i == 0 and self._fuse_img_ln and hasattr(ops, 'conv_down_ln'):
        # image-side layer: LayerNorm + ELU in the convolution's epilogue where the kernel covers it
        ops.conv_down_ln(x, m.p[f'{cl.name}/kernel'], m.p[f'{cl.name}/bias'], m.p[f'{cl.name}/norm/scale'],
                         m.p[f'{cl.name}/norm/bias'], R(a['z']), R(a['out']), a['stats'][r0 * px:r1 * px],
                         cl.k, 1.0 / 255.0)
        x = R(a['out'])
        continue
      ops.conv_down(x, m.p[f'{cl.name}/kernel'], m.p[f'{cl.name}/bias'], R(a['z']),
                    cl.k, 1.0 / 255.0 if i == 0 else 1.0)
      ops.ln_act_fwd(R(a['z']).view(-1, C), m.p[f'{cl.name}/norm/scale'],
                     m.p[f'{cl.name}/norm/bias'], R(a['out']).view(-1, C),
                     a['stats'][r0 * px:r1 * px], True)
      x = R(a['out'])
    if s.enc_mlp_keys:
      layers = [self.P[f'enc/mlp/dense{i}'] for i in range(len(self.enc_mlp_act))]
      self.mlp_fwd(layers, self.enc_mlp_act, R(b['vec_in']), R)
    # hoisted embed part of obs_out: written straight into its pre-LN buffer
    zo = R(self.a_obs_out.z)
    first = True
    if s.enc_convs or s.enc_res:
      top = self.enc_res['emb'] if s.enc_res else R(self.enc_act[-1]['out']).view(r1 - r0, -1)
      ops.gemm(top, self.P['obs_out_cnn'].W, zo)
      first = False
    if s.enc_mlp_keys:
      ops.gemm(R(self.enc_mlp_act[-1].out), self.P['obs_out_mlp'].W, zo,
               beta=0.0 if first else 1.0)

  def encoder_bwd(self):
    """Consumes a_obs_out.dz (all steps) as the gradient of the hoisted
    embed matmul."""
    s, ops, b = self.spec, self.ops, self.b
    m = self.groups['model']
    dzo = self.a_obs_out.dz
    if s.enc_mlp_keys:
      P = self.P['obs_out_mlp']
      top = self.enc_mlp_act[-1]
      ops.gemm(top.out, dzo, P.dW, ta=True)
      ops.gemm(dzo, P.W, top.dout, tb=True)
      layers = [self.P[f'enc/mlp/dense{i}'] for i in range(len(self.enc_mlp_act))]
      self.mlp_bwd(layers, self.enc_mlp_act, b['vec_in'])
    if s.enc_res:
      P, R = self.P['obs_out_cnn'], self.enc_res
      ops.gemm(R['emb'], dzo, P.dW, ta=True)
      ops.gemm(dzo, P.W, R['demb'], tb=True)
      self.encoder_res_bwd()
    if s.enc_convs:
      P = self.P['obs_out_cnn']
      top = self.enc_act[-1]
      ops.gemm(top['out'].view(self.N, -1), dzo, P.dW, ta=True)
      ops.gemm(dzo, P.W, top['dout'].view(self.N, -1), tb=True)
      for i in reversed(range(len(s.enc_convs))):
        cl, a = s.enc_convs[i], self.enc_act[i]
        C = cl.c_small
        if i == 0 and self._fuse_img_ln and hasattr(ops, 'conv_wgrad_ln'):
          # the first layer's dz feeds nothing but its own filter gradient (the image is not
          # differentiated): LayerNorm backward applied while that kernel stages the rows
          ops.conv_wgrad_ln(b['image'], a['dout'], a['z'], a['stats'], m.p[f'{cl.name}/norm/scale'],
                            m.p[f'{cl.name}/norm/bias'], a['dz'], m.g[f'{cl.name}/kernel'],
                            m.g[f'{cl.name}/norm/scale'], m.g[f'{cl.name}/norm/bias'], m.g[f'{cl.name}/bias'],
                            cl.k, 1.0 / 255.0)
          continue
        ops.ln_act_bwd(a['dout'].view(-1, C), a['z'].view(-1, C),
                       a['out'].view(-1, C), a['stats'],
                       m.p[f'{cl.name}/norm/scale'], a['dz'].view(-1, C),
                       m.g[f'{cl.name}/norm/scale'], m.g[f'{cl.name}/norm/bias'],
                       False, True, m.g[f'{cl.name}/bias'], beta=m.p[f'{cl.name}/norm/bias'])
        big = b['image'] if i == 0 else self.enc_act[i - 1]['out']
        ops.conv_wgrad(big, a['dz'], m.g[f'{cl.name}/kernel'], cl.k,
                       1.0 / 255.0 if i == 0 else 1.0)
        if i > 0:
          ops.conv_up(a['dz'], m.p[f'{cl.name}/kernel'], None,
                      self.enc_act[i - 1]['dout'], cl.k)

  # ----------------------------------------------------------------- decoder

  def decoder_fwd(self, feat, rows=None):
    """rows = (r0, r1): only those minibatch rows (`feat` is the full [N, F] matrix either way)."""
    s, ops, b = self.spec, self.ops, self.b
    m = self.groups['model']
    r0, r1 = rows if rows is not None else (0, self.N)
    R = lambda t: t[r0:r1]
    n = r1 - r0
    feat = R(feat)
    if s.dec_res:
      assert rows is None
      self.decoder_res_fwd(feat)
    elif s.dec_convs:
      c0, a0 = s.dec_convs[0], self.dec_act[0]
      kk = c0.k * c0.k
      bt = b['bias_tiled'].view(kk, c0.c_big)
      ops.copy2d(m.p[f'{c0.name}/bias'].view(1, -1).expand(kk, c0.c_big), bt)
      ops.gemm(feat, m.p[f'{c0.name}/kernel'].view(kk * c0.c_big, self.F),
               R(a0['z']).view(n, -1), tb=True, bias=b['bias_tiled'])
      x = None
      for i, (cl, a) in enumerate(zip(s.dec_convs, self.dec_act)):
        if i > 0:
          ops.conv_up(x, m.p[f'{cl.name}/kernel'], m.p[f'{cl.name}/bias'],
                      R(a['z']), cl.k)
        if cl.norm:
          C, px = cl.c_big, cl.h_big * cl.h_big
          ops.ln_act_fwd(R(a['z']).view(-1, C), m.p[f'{cl.name}/norm/scale'],
                         m.p[f'{cl.name}/norm/bias'], R(a['out']).view(-1, C),
                         a['stats'][r0 * px:r1 * px], True)
          x = R(a['out'])
    if s.dec_convs:
      last = self.dec_act[-1]
      c0 = 0
      for key, shp in s.dec_cnn_keys.items():
        scale = self.cfg['loss_scales'].get(key, 1.0)
        ops.image_loss(R(last['z']), R(b['image']), R(b['loss_image'][key]), R(last['dz']),
                       scale / self.Ng, c0, c0 + shp[2])
        c0 += shp[2]
    if s.dec_mlp_keys:
      outs = self.head_fwd('dec_mlp', self.acts_wm['dec_mlp'], feat, R if rows is not None else None)
      for (k, v), o, A in zip(s.dec_mlp_keys.items(), outs,
                              self.acts_wm['dec_mlp'][1]):
        scale = self.cfg['loss_scales'].get(k, 1.0)
        ops.mse_loss(o, R(b['vec_tgt'][k]), R(b['loss_vec'][k]), R(A.dout),
                     scale / self.Ng)

  def decoder_bwd(self, feat, dfeat, beta, defer=None):
    s, ops, b = self.spec, self.ops, self.b
    m = self.groups['model']
    N = self.N
    run = (lambda f: defer.append(f)) if defer is not None else (lambda f: f(ops))
    if s.dec_mlp_keys:
      self.head_bwd('dec_mlp', self.acts_wm['dec_mlp'], feat, None, dfeat, beta,
                    defer=defer)
      beta = 1.0
    if s.dec_res:
      self.decoder_res_bwd(feat, dfeat, beta)
    elif s.dec_convs:
      fused_ln = None     # the layer whose LayerNorm backward ran in the epilogue of the data gradient behind it
      for i in reversed(range(len(s.dec_convs))):
        cl, a = s.dec_convs[i], self.dec_act[i]
        C = cl.c_big
        if cl.norm and fused_ln == i:
          pass
        elif cl.norm:
          ops.ln_act_bwd(a['dout'].view(-1, C), a['z'].view(-1, C),
                         a['out'].view(-1, C), a['stats'],
                         m.p[f'{cl.name}/norm/scale'], a['dz'].view(-1, C),
                         m.g[f'{cl.name}/norm/scale'],
                         m.g[f'{cl.name}/norm/bias'], False, True,
                         m.g[f'{cl.name}/bias'], beta=m.p[f'{cl.name}/norm/bias'])
        else:
          run(lambda o, a=a, C=C, cl=cl: self._image_bias_grad(o, a, C, cl))
        if i > 0:
          prev = self.dec_act[i - 1]
          run(lambda o, a=a, prev=prev, cl=cl: o.conv_wgrad(
              a['dz'], prev['out'], m.g[f'{cl.name}/kernel'], cl.k))
          pl = s.dec_convs[i - 1]
          if (not cl.norm and pl.norm and cl.c_big <= 4 and self._fuse_img_ln and
              hasattr(ops, 'conv_down_lnbwd')):
            # image layer: its data gradient + the LayerNorm backward of the layer in front of it
            ops.conv_down_lnbwd(a['dz'], m.p[f'{cl.name}/kernel'], prev['z'], prev['stats'],
                                m.p[f'{pl.name}/norm/scale'], m.p[f'{pl.name}/norm/bias'], prev['dout'],
                                prev['dz'], m.g[f'{pl.name}/norm/scale'], m.g[f'{pl.name}/norm/bias'],
                                m.g[f'{pl.name}/bias'], cl.k)
            fused_ln = i - 1
          else:
            ops.conv_down(a['dz'], m.p[f'{cl.name}/kernel'], None, prev['dout'],
                          cl.k)
        else:
          kk = cl.k * cl.k
          dzv = a['dz'].view(N, -1)
          run(lambda o, dzv=dzv, cl=cl, kk=kk, C=C: o.gemm(
              dzv, feat, m.g[f'{cl.name}/kernel'].view(kk * C, self.F), ta=True))
          ops.gemm(dzv, m.p[f'{cl.name}/kernel'].view(kk * C, self.F), dfeat,
                   beta=beta)

  def _image_bias_grad(self, ops, a, C, cl):
    """Bias gradient of the image layer: C = 3 columns would leave 61 of 64
    column lanes idle; sum 64-pixel groups as a [rows/64, 64*C] matrix first,
    then fold the 64 groups."""
    m = self.groups['model']
    px = a['dz'].numel() // C
    if px % 64 == 0:
      tmp = self.b['bias_fold']
      ops.col_sum(a['dz'].view(px // 64, 64 * C), tmp)
      ops.col_sum(tmp.view(64, C), m.g[f'{cl.name}/bias'])
    else:
      ops.col_sum(a['dz'].view(-1, C), m.g[f'{cl.name}/bias'])

  # ------------------------------------------------------------------- RSSM

  def initial_fwd(self):
    """RSSM.initial 'learned2' (reference nets.py:54-62) on one row."""
    ops, b = self.ops, self.b
    m = self.groups['model']
    ops.tanh_fwd(m.p['rssm/initial_deter'], b['init_deter'])
    x = b['init_deter']
    for i in range(self.n_prior):
      x = self.lin_fwd(self.P[f'img_out_{i}'], self.a_init_out[i], x)
    xs = self.lin_fwd(self.P['img_stats'], self.a_init_stats, x)
    ops.stats_fwd(xs, None, b['init_logit'], b['init_stoch'], self.G, self.C,
                  self.unimix, 1)

  def core_fwd(self, xin, hprev, hn, A_in, z3, gstats, sel, gin=None):
    """RSSM.img_step up to the new deter (reference nets.py:119-130):
    img_in Linear+LN+ELU, then the GRU cell.  xin [rows,S+A], hprev [rows,D]."""
    ops = self.ops
    x1 = self.lin_fwd(self.P['img_in'], A_in, xin, sel)
    z3v = sel(z3)
    if gin is not None:  # hprev and x1 are adjacent column blocks of gin
      pre = ops.gemm(sel(gin), self.P['gru'].W, z3v, defer=True)
    else:
      ops.gemm(hprev, self.P['gru_h'].W, z3v)
      pre = ops.gemm(x1, self.P['gru_x'].W, z3v, beta=1.0, defer=True)
    g = self.P['gru_h']
    ops.gru_fwd(z3v, g.gamma, g.beta, hprev, hn, sel(gstats), pre=pre)

  def prior_fwd(self, deter, A_out, A_stats, sel=None):
    """img_out layers + img_stats on the new deter (reference nets.py:131-134);
    returns the raw prior statistics."""
    x = deter
    for i in range(self.n_prior):
      x = self.lin_fwd(self.P[f'img_out_{i}'], A_out[i], x, sel)
    return self.lin_fwd(self.P['img_stats'], A_stats, x, sel)

  def prior_bwd(self, deter, ddeter, A_out, A_stats, sel=None, params=False):
    """Backward of prior_fwd: A_stats.dout holds the gradient w.r.t. the raw
    statistics; accumulates into ddeter (and parameter gradients if asked)."""
    sel = sel or (lambda t: t)
    n = self.n_prior
    self.lin_bwd(self.P['img_stats'], A_stats, sel(A_out[-1].out), sel,
                 sel(A_out[-1].dout), 0.0, params)
    for i in reversed(range(n)):
      xin = deter if i == 0 else sel(A_out[i - 1].out)
      tgt = ddeter if i == 0 else sel(A_out[i - 1].dout)
      self.lin_bwd(self.P[f'img_out_{i}'], A_out[i], xin, sel, tgt,
                   1.0 if i == 0 else 0.0, params)

  def core_bwd(self, dhn_total, hprev, A_in, z3, gstats, sel, dz3, dy3,
               dh_direct, dxin, dxin_beta, dxin_P, dgin=None):
    """Data-gradient backward of core_fwd.  dhn_total [rows,D] is the gradient
    w.r.t. the new deter.  Outputs: dz3, dy3 (GRU), dh_direct = (1-update)*dhn
    + dz3 @ Wg_h^T, and dxin (+)= dz_in @ W_in^T restricted to dxin_P's rows."""
    ops = self.ops
    g = self.P['gru_h']
    if dgin is not None:  # [dh | dx1] in one buffer, one GEMM
      ops.gru_bwd(dhn_total, sel(z3), sel(gstats), g.gamma, g.beta, hprev, dz3,
                  dh_direct, dy3, zero=sel(A_in.dout))
      ops.gemm(dz3, self.P['gru'].W, sel(dgin), tb=True, beta=1.0)
    else:
      ops.gru_bwd(dhn_total, sel(z3), sel(gstats), g.gamma, g.beta, hprev, dz3,
                  dh_direct, dy3)
      ops.gemm(dz3, self.P['gru_h'].W, dh_direct, tb=True, beta=1.0)
      ops.gemm(dz3, self.P['gru_x'].W, sel(A_in.dout), tb=True)
    self.lin_bwd(self.P['img_in'], A_in, None, sel, None, params=False)
    ops.gemm(sel(A_in.dz), dxin_P.W, dxin, tb=True, beta=dxin_beta)

  def reset_carry(self):
    """state=None: the carried state is RSSM.initial() (reference
    agent.py:39-40, 71-72)."""
    b, D = self.b, self.D
    self.initial_fwd()
    self.ops.copy2d(b['init_deter'].expand(self.B, D), b['carry'][:, :D])
    self.ops.copy2d(b['init_stoch'].expand(self.B, self.S), b['carry'][:, D:])

  def observe_scan_fused(self, use_carry, half=None):
    """The T obs_steps as one persistent launch (dd_observe_scan_fwd): same buffers, same
    values up to the summation order of the small contractions.  half: index into self.halves -
    the scan of that batch range only (its rows of every batch-major buffer, its own noise tensor
    and draw-index workspace)."""
    ops, b, P = self.ops, self.b, self.P
    if half in (None, 0):
      for W, planes, kp in self.scan_w[1:]:   # the weights changed in the last optimizer step (P1 gathers img_in rows in fp32: no planes)
        ops.scan_wprep(W, planes, kp)
    g = P['gru_h']
    T = self.T
    if half is None:
      B, rows, u_post, idx = self.B, (lambda t: t), b['u_post'], self.scan_idx
      carry = b['carry']
    else:
      b0, b1 = self.halves[half]
      B, u_post, idx = b1 - b0, self.u_post_h[half], self.scan_idx_h[half]
      rows = lambda t: t[b0 * T:b1 * T]
      carry = b['carry'][b0:b1]
    ops.observe_scan_fwd(
        B, T, self.D, self.U, self.G, self.C, self.A, use_carry, self.unimix,
        rows(b['first']), carry if use_carry else None, b['init_deter'], b['init_stoch'],
        u_post, [w[1] for w in self.scan_w],
        [P['img_in'].gamma, P['img_in'].beta, g.gamma, g.beta, P['obs_out_h'].gamma,
         P['obs_out_h'].beta, P['obs_stats'].bias],
        [rows(t) for t in (b['xin'], self.a_img_in.z, self.a_img_in.stats, b['gin'], b['z3'], b['gstats'],
                           b['post'], self.a_obs_out.z, self.a_obs_out.out, self.a_obs_out.stats,
                           self.a_obs_stats.z, b['post_logit'])],
        P['img_in'].W, idx, self.scan_sync)

  def observe_scan_bwd_fused(self):
    """The data gradient of the T obs_steps as one persistent launch (dd_observe_scan_bwd): same
    buffers as the launch sequence below, same values up to the summation order of the small
    contractions."""
    ops, b, P = self.ops, self.b, self.P
    B, T, D = self.B, self.T, self.D
    last = lambda buf: buf.view(B, T, -1)[:, T - 1]
    Aq, Ao, Ai = self.a_obs_stats, self.a_obs_out, self.a_img_in
    ops.stats_bwd(last(Aq.z), last(b['dpost_logit']), b['dfeat'].view(B, T, -1)[:, T - 1, D:],
                  last(Aq.dout), self.G, self.C, self.unimix)
    for W, planes in self.scan_wb:   # (the weights of this step: unchanged since the forward scan)
      ops.scan_wprep_rows(W, planes)
    g = P['gru_h']
    ops.observe_scan_bwd(
        B, T, D, self.U, self.G, self.C, 0, self.unimix, b['first'],
        [Aq.z, Ao.z, Ao.out, Ao.stats, b['z3'], b['gstats'], b['gin'], Ai.z, Ai.stats],
        b['dpost_logit'], [w[1] for w in self.scan_wb],
        [P['obs_out_h'].gamma, g.gamma, g.beta, P['img_in'].gamma],
        [b['dfeat'], Aq.dout, Ao.dout, Ao.dz, b['dz3'], b['dy3'], b['dgin'], Ai.dz, b['dxin_s']],
        self.scan_sync)

  def observe_fwd(self, use_carry=True, prior=True):
    """prior=False: with the fused scan the caller computes the prior of the observe steps
    (phase_wm_fwd's post_fwd); the launch sequence always computes it inside its time loop."""
    ops, b = self.ops, self.b
    B, T, D, S, F = self.B, self.T, self.D, self.S, self.F
    if self.fused_scan:
      self.observe_scan_fused(use_carry)
      if prior:
        xs = self.prior_fwd(b['post'][:, :D], self.a_img_out, self.a_img_stats)
        ops.stats_fwd(xs, b['u_prior'].view(self.N, self.G), b['prior_logit'],
                      b['prior_stoch'], self.G, self.C, self.unimix, 0)
      return
    bt = lambda t_: (lambda buf: buf.view(B, T, -1)[:, t_])
    first = b['first'].view(B, T)
    post = b['post'].view(B, T, F)
    for t in range(T):
      sel = bt(t)
      if t == 0:
        pd = b['carry'][:, :D] if use_carry else None
        ps = b['carry'][:, D:] if use_carry else None
      else:
        pd, ps = post[:, t - 1, :D], post[:, t - 1, D:]
      hprev = sel(b['hprev'])
      xin = sel(b['xin'])
      ops.reset_mask2(pd, b['init_deter'], hprev, ps, b['init_stoch'], xin[:, :S],
                      first[:, t])
      self.core_fwd(xin, hprev, post[:, t, :D], self.a_img_in, b['z3'],
                    b['gstats'], sel, gin=b['gin'])
      # posterior: obs_out on concat[deter, embed]; embed part already in z
      Ao = self.a_obs_out
      pre = ops.gemm(post[:, t, :D], self.P['obs_out_h'].W, sel(Ao.z), beta=1.0, defer=True)
      Po = self.P['obs_out_h']
      ops.ln_act_fwd(sel(Ao.z), Po.gamma, Po.beta, sel(Ao.out), sel(Ao.stats), True, pre=pre)
      xq, pre = self.lin_fwd(self.P['obs_stats'], self.a_obs_stats, sel(Ao.out), sel, defer=True)
      ops.stats_fwd(xq, b['u_post'][t], sel(b['post_logit']), post[:, t, D:],
                    self.G, self.C, self.unimix, 0, pre=pre)
    # The prior statistics depend only on deter_t and feed nothing inside the
    # scan (the prior sample is unused by obs_step): evaluate them for all T
    # steps at once, off the sequential critical path.
    xs = self.prior_fwd(b['post'][:, :D], self.a_img_out, self.a_img_stats)
    ops.stats_fwd(xs, b['u_prior'].view(self.N, self.G), b['prior_logit'],
                  b['prior_stoch'], self.G, self.C, self.unimix, 0)

  def observe_bwd(self):
    """Reverse scan.  On entry dfeat holds the heads' gradient w.r.t. every
    posterior [deter | stoch]; dpost_logit / dprior_logit hold the KL
    gradients."""
    ops, b = self.ops, self.b
    B, T, D, S, F = self.B, self.T, self.D, self.S, self.F
    bt = lambda t_: (lambda buf: buf.view(B, T, -1)[:, t_])
    first = b['first'].view(B, T)
    post = b['post'].view(B, T, F)
    dfeat = b['dfeat'].view(B, T, F)
    Ao, Po = self.a_obs_out, self.P['obs_out_h']
    P = self.P
    # prior head (KL gradient only): bulk over all steps, with parameter grads
    ops.stats_bwd(self.a_img_stats.z, b['dprior_logit'], None,
                  self.a_img_stats.dout, self.G, self.C, self.unimix)
    self.prior_bwd(b['post'][:, :D], b['dfeat'][:, :D], self.a_img_out,
                   self.a_img_stats, None, params=True)
    if self.fused_scan_bwd:
      self.observe_scan_bwd_fused()
    for t in (() if self.fused_scan_bwd else reversed(range(T))):
      sel = bt(t)
      ddeter, dstoch = dfeat[:, t, :D], dfeat[:, t, D:]
      # posterior sample + statistics
      ops.stats_bwd(sel(self.a_obs_stats.z), sel(b['dpost_logit']), dstoch,
                    sel(self.a_obs_stats.dout), self.G, self.C, self.unimix)
      pre = ops.gemm(sel(self.a_obs_stats.dout), P['obs_stats'].W, sel(Ao.dout), tb=True,
                     defer=True)
      ops.ln_act_bwd(sel(Ao.dout), sel(Ao.z), sel(Ao.out), sel(Ao.stats),
                     Po.gamma, sel(Ao.dz), None, None, False, True, pre=pre, beta=Po.beta)
      ops.gemm(sel(Ao.dz), Po.W, ddeter, tb=True, beta=1.0)
      self.core_bwd(ddeter, sel(b['hprev']), self.a_img_in, b['z3'],
                    b['gstats'], sel, sel(b['dz3']), sel(b['dy3']),
                    sel(b['dhprev']), sel(b['dxin_s']), 0.0, P['img_in_s'],
                    dgin=b['dgin'])
      if t > 0:
        ops.reset_mask_bwd2(sel(b['dhprev']), dfeat[:, t - 1, :D],
                            sel(b['dxin_s']), dfeat[:, t - 1, D:], first[:, t])
    # ---- bulk parameter gradients of the scan layers over all T steps
    m = self.groups['model']
    lnp = lambda L, A: ops.ln_param_grad(A.dout, A.z, A.out, A.stats, L.dgamma,
                                         L.dbeta, False, True)
    Aq = self.a_obs_stats
    ops.gemm(Ao.out, Aq.dout, P['obs_stats'].dW, ta=True)
    ops.col_sum(Aq.dout, P['obs_stats'].dbias)
    ops.gemm(b['post'][:, :D], Ao.dz, P['obs_out_h'].dW, ta=True)
    lnp(P['obs_out_h'], Ao)
    ops.gemm(b['gin'], b['dz3'], P['gru'].dW, ta=True)
    g = P['gru_h']
    ops.ln_param_grad(b['dy3'], b['z3'], None, b['gstats'], g.dgamma, g.dbeta,
                      False, False)
    ops.gemm(b['xin'], self.a_img_in.dz, P['img_in'].dW, ta=True)
    lnp(P['img_in'], self.a_img_in)
    # learned initial deter: rows with is_first take init_deter
    ops.gemm(b['first'].view(-1, 1), b['dhprev'], b['dinit_deter'], ta=True)
    ops.tanh_bwd(m.p['rssm/initial_deter'], b['dinit_deter'].view(-1),
                 m.g['rssm/initial_deter'], 0.0)

  # --------------------------------------------------------------- optimizer

  def opt_step(self, name, cfgkey):
    g = self.groups[name]
    c = self.cfg[cfgkey]
    if name == 'model' and self._early is not None:
      (lo, hi), cs = self._early
      self._early = None
      comm, head, tail = self.comm, g.gflat[:lo], g.gflat[hi:]
      def rest():
        if head.numel():
          comm.allreduce_sum(head)
        if tail.numel():
          comm.allreduce_sum(tail)
        if cs is not None:   # the early range's sum must have landed before the norm reads it
          torch.cuda.current_stream(self.device).wait_stream(cs)
      self.plan.cut(rest)
    else:
      self.allreduce(g.gflat)
    self.ops.grad_norm(g.gflat, g.opt_state, self.mixed)
    self.ops.adam_step(g.flat, g.gflat, g.m, g.v, g.n_decay, g.opt_state,
                       c['lr'], c['wd'], c['eps'], 0.9, 0.999, c['clip'], c.get('warmup', 0))

  # ------------------------------------------------------------------ phases

  INPUT_KEYS = ('image', 'vec_in', 'vec_tgt', 'action', 'reward', 'is_first', 'is_terminal')

  def input_stage(self):
    """A second set of the input buffers (same shapes / dtypes as b[...]): the pipelined agent
    uploads step k + 1's minibatch into it while step k's world-model phase still reads b[...],
    and commit_inputs copies it over in stream order (agent.Pipeline)."""
    st = {}
    for k in self.INPUT_KEYS:
      v = self.b.get(k)
      if isinstance(v, dict):
        st[k] = {kk: torch.empty_like(vv) for kk, vv in v.items()}
      elif v is not None:
        st[k] = torch.empty_like(v)
    return st

  def commit_inputs(self, stage):
    """b[...] <- the staged minibatch (device-to-device, on the current stream)."""
    for k, v in stage.items():
      if isinstance(v, dict):
        for kk, vv in v.items():
          self.b[k][kk].copy_(vv)
      else:
        self.b[k].copy_(v)

  def upload(self, data, dst=None):
    """Stage one replay minibatch (wire format) into the learner's HBM buffers (dst: into a
    set of buffers from input_stage() instead).
    Values are host numpy arrays (copied over PCIe) or device tensors, e.g. from
    replay.DeviceReplay.sample_batch (device-to-device)."""
    s, b = self.spec, (dst if dst is not None else self.b)
    dev = self.device
    def tens(x):
      if isinstance(x, torch.Tensor):
        return x
      return torch.from_numpy(np.ascontiguousarray(x)).to(dev, non_blocking=True)
    def put(dst, t):
      dst.copy_(t.reshape(dst.shape))  # copy_ converts dtype (bool -> uint8, f64 -> f32)
    if s.enc_cnn_keys:
      imgs = [tens(data[k]) for k in s.enc_cnn_keys]
      put(b['image'], imgs[0] if len(imgs) == 1 else torch.cat(imgs, -1))
    if s.enc_mlp_keys:
      cols = [tens(data[k]).reshape(self.N, -1).to(b['vec_in'].dtype) for k in s.enc_mlp_keys]
      put(b['vec_in'], cols[0] if len(cols) == 1 else torch.cat(cols, -1))
    for k in s.dec_mlp_keys:
      put(b['vec_tgt'][k], tens(data[k]))
    put(b['action'], tens(data['action']))
    put(b['reward'], tens(data['reward']))
    put(b['is_first'], tens(data['is_first']))
    put(b['is_terminal'], tens(data['is_terminal']))

  def phase_prep(self):
    """Per-step preparation of the world-model phase: step number, wire-format flags,
    the observe scan's sampling noise."""
    ops, b = self.ops, self.b
    B, T, G = self.B, self.T, self.G
    ops.counter_add(self.step_ctr, 1)
    ops.batch_prep(b['is_first'], b['is_terminal'], b['action'], b['first'],
                   b['cont'], b['xin'][:, self.S:])
    r0 = self.rank * B
    ops.philox(b['u_prior'], B, T, G, T, r0 * T, self.noise_seed, self.step_ctr, SITE_OBS_PRIOR, 0)
    ops.philox(b['u_post'], T, B, G, self.Bg, r0, self.noise_seed, self.step_ctr, SITE_OBS_POST, 0)
    if getattr(self, 'split_fwd', False):   # the same uniforms (keyed by global row) in per-half tensors
      for (b0, b1), u in zip(self.halves, self.u_post_h):
        ops.philox(u, T, b1 - b0, G, self.Bg, r0 + b0, self.noise_seed, self.step_ctr, SITE_OBS_POST, 0)

  def phase_prep_b(self):
    """The behaviour phase's own step number and sampling noise (same Philox keys as
    if drawn in phase_prep: (seed, step, site, global row))."""
    ops, b = self.ops, self.b
    B, T, N, H, G, A = self.B, self.T, self.N, self.H, self.G, self.A
    ops.counter_add(self.step_ctr_b, 1)
    r0 = self.rank * B
    if H > 0:
      ops.philox(b['u_img'], H, N, G, self.Ng, r0 * T, self.noise_seed, self.step_ctr_b, SITE_IMG, 0)
    if self.discrete:
      ops.philox(b['u_act'], H + 1, N, 1, self.Ng, r0 * T, self.noise_seed, self.step_ctr_b, SITE_ACT, 0)
    else:
      ops.philox(b['eps'], H + 1, N, A, self.Ng, r0 * T, self.noise_seed, self.step_ctr_b, SITE_ACT, 1)

  def phase_wm_fwd(self, use_carry=True, training=True):
    ops, b, cfg = self.ops, self.b, self.cfg
    N = self.N
    feat = b['post']
    ls = cfg['loss_scales']
    def post_fwd(rows=None):
      """Everything of the forward pass that consumes the posterior of a row range: prior of the
      observe steps, decoder, reward / cont heads and their losses (uses self.ops: the caller's
      launch context)."""
      o = self.ops
      R = (lambda t: t[rows[0]:rows[1]]) if rows is not None else (lambda t: t)
      sel = R if rows is not None else None
      if self.fused_scan:   # (the launch sequence computes the prior inside its time loop)
        xs = self.prior_fwd(R(feat)[:, :self.D], self.a_img_out, self.a_img_stats, sel)
        o.stats_fwd(xs, R(b['u_prior'].view(N, self.G)), R(b['prior_logit']),
                    R(b['prior_stoch']), self.G, self.C, self.unimix, 0)
      self.decoder_fwd(feat, rows)
      (rew,) = self.head_fwd('reward', self.acts_wm['reward'], R(feat), sel)
      o.scalar_loss(rew.view(-1), R(b['reward'].view(-1)), R(b['loss_reward']),
                    R(self.acts_wm['reward'][1][0].dout).view(-1),
                    ls.get('reward', 1.0) / self.Ng, 0)
      (cont,) = self.head_fwd('cont', self.acts_wm['cont'], R(feat), sel)
      o.scalar_loss(cont.view(-1), R(b['cont'].view(-1)), R(b['loss_cont']),
                    R(self.acts_wm['cont'][1][0].dout).view(-1),
                    ls.get('cont', 1.0) / self.Ng, 1)
    if self.split_fwd and self.ops2 is not None:
      T = self.T
      (a0, a1), (c0, c1) = self.halves
      main = self.ops
      def on_side(fn):   # fn on the side stream with the side launch context, after all work issued so far
        with self.fork():
          self.ops = self.ops2
          try:
            fn()
          finally:
            self.ops = main
      self.encoder_fwd((a0 * T, a1 * T))
      on_side(lambda: self.encoder_fwd((c0 * T, c1 * T)))       # || scan of the first half
      self.initial_fwd()
      self.observe_scan_fused(use_carry, 0)
      self.join()
      on_side(lambda: post_fwd((a0 * T, a1 * T)))               # || scan of the second half
      self.observe_scan_fused(use_carry, 1)
      self.join()
      post_fwd((c0 * T, c1 * T))
    else:
      self.encoder_fwd()
      self.initial_fwd()
      self.observe_fwd(use_carry, prior=False)
      post_fwd()
    rew = self.acts_wm['reward'][1][0].z
    cont = self.acts_wm['cont'][1][0].z
    ops.kl_fwd(b['post_logit'], b['prior_logit'], b['kl'], b['ent_post'],
               b['ent_prior'], self.G, self.C)
    k = self.stat('kl_loss', b['kl'])
    self.flush_stats()
    # AutoAdapt is updated before it is used (reference tfutils.py:441-442)
    self.allreduce(self.stat_sums[k])
    self.stat_prereduced.add(k)
    c = cfg['wmkl']
    if c['impl'] != 'fixed' and training:
      ops.autoadapt_update(self.wmkl_scale, self.stat_sums[k], float(self.Ng),
                           c['target'], 0.1, c['vel'], c['min'], c['max'], False, c['impl'])
    self.stat('post_ent', b['ent_post'])
    self.stat('prior_ent', b['ent_prior'])
    self.stat('reward_loss', b['loss_reward'])
    self.stat('cont_loss', b['loss_cont'])
    for kk in (self.spec.dec_cnn_keys if self.spec.dec_convs else ()):
      self.stat(f'{kk}_loss', b['loss_image'][kk])
    for kk in self.spec.dec_mlp_keys:
      self.stat(f'{kk}_loss', b['loss_vec'][kk])
    # metrics only: balance statistics of the two scalar heads (agent.py:204-209) and the
    # summed scaled loss map whose std is reported (agent.py:186-190, 202-203)
    ops.balance_stats(rew.view(-1), b['reward'], b['loss_reward'], 0.1, 0, self.bal[0])
    ops.balance_stats(cont.view(-1), b['cont'], b['loss_cont'], 0.5, 1, self.bal[1])
    tot = b['loss_total']
    ops.axpy(b['kl'], ls.get('kl', 1.0), self.wmkl_scale, tot, accumulate=False)
    ops.axpy(b['loss_reward'], ls.get('reward', 1.0), None, tot)
    ops.axpy(b['loss_cont'], ls.get('cont', 1.0), None, tot)
    for kk in (self.spec.dec_cnn_keys if self.spec.dec_convs else ()):
      ops.axpy(b['loss_image'][kk], ls.get(kk, 1.0), None, tot)
    for kk in self.spec.dec_mlp_keys:
      ops.axpy(b['loss_vec'][kk], ls.get(kk, 1.0), None, tot)
    self.stat('model_total', tot)
    self.flush_stats()

  def phase_wm_bwd(self):
    ops, b, cfg = self.ops, self.b, self.cfg
    self._early = None   # (a step that aborted between allreduce_early and opt_step left it set)
    feat, dfeat = b['post'], b['dfeat']
    gh = cfg['grad_heads']
    # Heads / decoder: the data-gradient chain (-> dfeat) runs first on the main
    # stream; their weight-gradient contractions only need the dz buffers and are
    # queued, then issued on the side stream so the MFMA-heavy filter gradients
    # overlap the latency-bound reverse scan below.
    defer = [] if self.ops2 is not None else None
    beta = 0.0
    for name in ('reward', 'cont'):
      flow = name in gh
      self.head_bwd(name, self.acts_wm[name], feat, None,
                    dfeat if flow else None, beta, defer=defer)
      if flow:
        beta = 1.0
    if 'decoder' in gh:
      self.decoder_bwd(feat, dfeat, beta, defer)
    else:
      if beta == 0.0:
        ops.fill(dfeat, 0.0)
      tmp = self.b.get('dfeat_sink')
      if tmp is None:   # (first eager step: never allocated inside a captured segment)
        tmp = self.b['dfeat_sink'] = self.zeros(self.N, self.F)
      self.decoder_bwd(feat, tmp, 0.0, defer)
    ops.kl_bwd(b['post_logit'], b['prior_logit'], self.wmkl_scale,
               cfg['loss_scales'].get('kl', 1.0) / self.Ng, cfg['wmkl_balance'],
               b['dpost_logit'], b['dprior_logit'], self.G, self.C)
    if defer:
      with self.fork():
        for f in defer:
          f(self.ops2)
    self.observe_bwd()
    if self.dp_overlap:
      self.join()            # head / decoder weight gradients complete
      self.allreduce_early()
    self.encoder_bwd()
    self.join()

  def phase_wm_opt(self):
    """World-model optimizer step and the hand-over to the behaviour phase.  Everything
    the behaviour phase needs from this step's buffers (imagination start states, the
    start continuation flags) is copied here, so it never reads world-model-phase
    buffers and the next step's world-model phase may overwrite them while it runs."""
    ops, b = self.ops, self.b
    self.opt_step('model', 'model_opt')
    # carry the last posterior to the next call (reference agent.py:211)
    post = b['post'].view(self.B, self.T, self.F)
    ops.copy2d(post[:, self.T - 1], b['carry'])
    ops.copy2d(b['post'], b['traj'][0][:, :self.F])
    ops.copy2d(b['cont'].view(1, -1), b['cont_b'].view(1, -1))
    self._stamp(3)

  def imagine_rollout(self, on_state=None):
    """WorldModel.imagine (reference agent.py:234-254): H img_steps from the start states in
    traj[0][:, :F] with the actor's sampled actions; fills traj [H+1, N, deter|stoch|action].
    on_state(t): called once the latent state of time row t is complete (t = 0..H)."""
    ops, b, cfg = self.ops, self.b, self.cfg
    N, H, M, D, S, A, F = self.N, self.H, self.M, self.D, self.S, self.A, self.F
    traj = b['traj']
    ca = cfg['actor']
    lo, hi = ca['minstd'], ca['maxstd']
    if self.fused_imag:
      # one persistent launch.  (hip.imag_split: two launches split at a head-chunk boundary, so
      # that the overlapped head evaluation of phase_imagine works on the first half meanwhile -
      # measured at configs[1]: no gain, 7.60 vs 7.48 ms for phase_imagine: the persistent launch
      # holds 157 of the 256 CUs, the heads crawl on the rest and their second half still runs
      # after the rollout.  Off by default.)
      cut = (H + 1) // 2 // self.HEAD_CHUNK * self.HEAD_CHUNK
      if on_state and 0 < cut <= H and self.cfg.get('hip', {}).get('imag_split', False):
        self.imagine_rollout_fused(0, cut)     # states 0 .. cut complete
        on_state(cut - 1)
        self.imagine_rollout_fused(cut, H + 1, prep=False)
      else:
        self.imagine_rollout_fused()
      if on_state:
        on_state(H)
      return
    if on_state:
      on_state(0)
    for t in range(H + 1):
      st = lambda buf, t_=t: buf.view(H + 1, N, -1)[t_]
      if self.discrete:
        (xa,) = self.head_fwd('actor', self.acts_im['actor'], traj[t][:, :F], st)
        ops.stats_fwd(xa, b['u_act'][t], st(b['alogit']), traj[t][:, F:F + A], 1, A,
                      float(ca['unimix']), 0)
      else:
        om, os_ = self.head_fwd('actor', self.acts_im['actor'], traj[t][:, :F], st)
        ops.normal_head_fwd(om, os_, b['eps'][t], traj[t][:, F:F + A], lo, hi)
      if t < H:
        si = lambda buf, t_=t: buf.view(H, N, -1)[t_]
        self.core_fwd(traj[t][:, D:F + A], traj[t][:, :D], traj[t + 1][:, :D],
                      self.ai_img_in, b['iz3'], b['igstats'], si)
        xs = self.prior_fwd(traj[t + 1][:, :D], self.ai_img_out,
                            self.ai_img_stats, si)
        ops.stats_fwd(xs, b['u_img'][t], b['ilogit'], traj[t + 1][:, D:F],
                      self.G, self.C, self.unimix, 0)
        if on_state:
          on_state(t + 1)

  def _imag_rows(self, ops):
    """Rows per workgroup of the fused continuous-action rollout and its reverse pass (process-wide
    switch, set before each launch): 32 holds half the CUs for 1.23 x the time - it pays where
    another stream has work for the freed CUs (the pipelined schedule: -0.4 ms per step at
    configs[1]) and costs 0.7 ms where the step waits for the rollout (sequential plan, eager
    steps); with few rows the 16-row form's shorter step wins (8 instead of 16 workgroups free nothing)."""
    if hasattr(ops, 'imag_set_rows'):
      rows = self.cfg.get('hip', {}).get('imag_rows', 'auto')
      rows = (32 if (self._pipelined_capture and self.N >= 1024) else 16) if str(rows) == 'auto' else int(rows)
      ops.imag_set_rows(rows)

  def imagine_rollout_fused(self, t0=0, t1=None, prep=True):
    """The H img_steps and H + 1 policy evaluations as one persistent launch
    (dd_imagine_rollout_fwd): same buffers as the launch sequence above, same values up to the
    summation order of the contractions (the one-hot stoch inputs are gathered, not multiplied)."""
    ops, b, cfg = self.ops, self.b, self.cfg
    ca = cfg['actor']
    if prep:
      for W, planes, col0 in self.imag_planes.values():   # the weights changed in the optimizer steps
        ops.imag_wprep(W, planes, col0)
    pl = self.imag_planes
    layers, outs = self.heads['actor']
    acts, oacts = self.acts_im['actor']
    if self.discrete:
      return self._imagine_rollout_fused_onehot(t0, t1)
    t = [b['traj'], b['u_img'], b['eps']]
    for i in range(ca['layers']):
      t += [pl[f'actor{i}'][1], layers[i].gamma, layers[i].beta, acts[i].z, acts[i].stats, acts[i].out]
    t += [layers[0].W, pl['head_m'][1], outs[0].bias, outs[1].bias, oacts[0].z, oacts[1].z]
    P, ai = self.P, self.ai_img_in
    t += [P['img_in'].W, P['img_in'].gamma, P['img_in'].beta, ai.z, ai.stats, ai.out]
    t += [pl['gru'][1], P['gru_h'].gamma, P['gru_h'].beta, b['iz3'], b['igstats']]
    for i in range(self.n_prior):
      L, a = P[f'img_out_{i}'], self.ai_img_out[i]
      t += [pl[f'img_out{i}'][1], L.gamma, L.beta, a.z, a.stats, a.out]
    t += [pl['stats'][1], P['img_stats'].bias, self.ai_img_stats.z]
    if getattr(self, 'imag_stamps', None) is not None:   # measurement aid (tools/imag_time.py)
      t.append(self.imag_stamps)
    self._stamp(5)
    self._imag_rows(ops)
    ops.imagine_rollout_fwd(self.N, self.H, self.D, self.U, self.G, self.C, self.A, ca['units'],
                            self.unimix, ca['minstd'], ca['maxstd'], t, t0, t1)
    self._stamp(6)

  def _imagine_rollout_fused_onehot(self, t0=0, t1=None):
    """dd_imagine_rollout_oh_fwd (csrc/imag_oh.hip): the one-hot / REINFORCE rollout at deter =
    units = 512 as one persistent launch, forward only."""
    ops, b, cfg, P = self.ops, self.b, self.cfg, self.P
    ca, pl = cfg['actor'], self.imag_planes
    layers, outs = self.heads['actor']
    acts, oacts = self.acts_im['actor']
    t = [b['traj'], b['u_img'], b['u_act']]
    for i in range(ca['layers']):
      t += [pl[f'actor{i}'][1], layers[i].gamma, layers[i].beta, acts[i].z, acts[i].stats, acts[i].out]
    t += [layers[0].W, pl['head'][1], outs[0].bias, oacts[0].z, b['alogit']]
    ai = self.ai_img_in
    t += [P['img_in'].W, P['img_in'].gamma, P['img_in'].beta, ai.z, ai.stats, ai.out]
    t += [pl['gru'][1], P['gru_h'].gamma, P['gru_h'].beta, b['iz3'], b['igstats']]
    for i in range(self.n_prior):
      L, a = P[f'img_out_{i}'], self.ai_img_out[i]
      t += [pl[f'img_out{i}'][1], L.gamma, L.beta, a.z, a.stats, a.out]
    t += [pl['stats'][1], P['img_stats'].bias, self.ai_img_stats.z]
    ops.imagine_rollout_oh_fwd(self.N, self.H, self.D, self.U, self.G, self.C, self.A, ca['units'], self.TW,
                               self.unimix, float(ca['unimix']), t, t0, t1)

  def imagine_reverse_fused(self):
    """Steps t = H .. 1 of the reverse imagination scan (stats / img_out / GRU / img_in backward)
    as one persistent launch: same dtraj as the launch sequence of _actor_backprop.scan_step."""
    ops, b, P = self.ops, self.b, self.P
    for W, planes in self.imag_planes_t.values():
      ops.imag_wprep_t(W, planes)
    plt = self.imag_planes_t
    t = [b['traj'], b['dtraj'], self.ai_img_stats.z, plt['stats'][1]]
    for i in range(self.n_prior):
      a = self.ai_img_out[i]
      t += [plt[f'img_out{i}'][1], P[f'img_out_{i}'].gamma, a.z, a.stats, a.out]
    t += [plt['gru'][1], P['gru_h'].gamma, P['gru_h'].beta, b['iz3'], b['igstats']]
    ai = self.ai_img_in
    t += [plt['img_in'][1], P['img_in'].gamma, ai.z, ai.stats, ai.out]
    if getattr(self, 'imag_stamps_b', None) is not None:   # measurement aid (tools/imag_time.py)
      t.append(self.imag_stamps_b)
    self._imag_rows(ops)
    ops.imagine_rollout_bwd(self.N, self.H, self.D, self.U, self.G, self.C, self.A, self.unimix, t)

  HEAD_CHUNK = 4  # time rows per chunk of the overlapped head evaluation

  def phase_imagine(self):
    ops, b, cfg = self.ops, self.b, self.cfg
    N, H, M, D, S, A, F = self.N, self.H, self.M, self.D, self.S, self.A, self.F
    traj = b['traj']  # traj[0][:, :F] = start states, set by phase_wm_opt
    feat = traj.view(M, self.TW)[:, :F]
    # reward / cont / target-critic heads over the trajectory (rewfn, cont head, target net:
    # agent.py:255-257, 400-403, 426).  They only read finished latent states, so each chunk
    # of time rows is evaluated on the side stream while the rollout - a chain of mid-size
    # dependent launches that leaves most of the chip idle - continues on the main stream.
    def heads(r0, r1):
      sel = lambda buf: buf[r0:r1]
      x = feat[r0:r1]
      self.head_fwd('reward', self.acts_im['reward'], x, sel)
      self.head_fwd('cont', self.acts_im['cont'], x, sel)
      # target network: the slow copy, or the online critic itself with slow_target: False
      # (agent.py:391-396: then heads['critic_target'] aliases the online parameters)
      self.head_fwd('critic_target', self.acts_im['critic_target'], x, sel)
    side = self.side_stream_b   # (None on CPU: the chunks then run inline, same arithmetic)
    if self.ops_b2 is None or not self._in_b or not self.overlap_b:
      self.imagine_rollout()
      heads(0, M)
    else:
      done = [0]
      def on_state(t):
        if (t + 1) % self.HEAD_CHUNK == 0 or t == H:
          r0, r1 = done[0] * N, (t + 1) * N
          done[0] = t + 1
          if t == H:   # last chunk: nothing left to overlap with, stay on the main stream
            if side is not None:
              self.join(side)
            heads(r0, r1)
            return
          with (self.fork(side) if side is not None else contextlib.nullcontext()):
            keep, self.ops = self.ops, self.ops_b2
            try:
              heads(r0, r1)
            finally:
              self.ops = keep
      self.imagine_rollout(on_state)
    rew = self.acts_im['reward'][1][0].z
    cont = self.acts_im['cont'][1][0].z
    val = self.acts_im['critic_target'][1][0].z
    ops.imag_returns_fwd(rew.view(-1), val.view(-1), cont.view(-1), b['cont_b'],
                         b['i_reward'], b['i_value'], b['i_cont'], b['i_weight'],
                         b['i_ret'], H, N, cfg['discount'], cfg['return_lambda'],
                         cfg['critic_return'])
    # ---- critic update (reference agent.py:398-417)
    HN = H * N
    (cout,) = self.head_fwd('critic', self.acts_im['critic'], feat[:HN])
    ops.critic_loss(cout.view(-1), b['i_ret'], b['i_weight'], b['i_crit_loss'],
                    self.acts_im['critic'][1][0].dout.view(-1), 1.0 / (H * self.Ng))
    self.head_bwd('critic', self.acts_im['critic'], feat[:HN])
    self.stat('critic_loss', b['i_crit_loss'][:HN])
    ops.symexp(cout.view(-1), b['i_critic'][:HN])     # dist.mean() of the critic's own prediction
    self.stat('imag_critic', b['i_critic'][:HN])
    self.stat('imag_reward', b['i_reward'][:HN])
    self.stat('imag_return', b['i_ret'][:HN])
    self.stat('imag_value', b['i_value'])
    self.flush_stats()
    self.opt_step('critic', 'critic_opt')

  def update_slow(self):
    """VFunction.update_slow (reference agent.py:444-454); host-side counter."""
    cfg = self.cfg
    if not cfg['slow_target']:
      self.slow_copied = True  # the target IS the freshly updated critic: re-evaluate it
      return
    init = self.slow_updates == -1
    self.slow_copied = bool(init or self.slow_updates >= cfg['slow_target_update'])
    if self.slow_copied:
      self.slow_updates = 0
      mix = 1.0 if init else cfg['slow_target_fraction']
      src, dst = self.groups['critic'], self.groups['critic_target']
      assert src.n == dst.n
      if mix == 1.0:
        # both arenas have the same layout: one flat copy
        self.ops.copy2d(src.flat.view(1, -1), dst.flat.view(1, -1))
      else:
        # d = mix * s + (1 - mix) * d (agent.py:452-453) over the flat arenas, through a copy of
        # d (the elementwise kernel's operands must not alias)
        tmp = self.b.get('slow_tmp')
        if tmp is None:
          tmp = self.b['slow_tmp'] = torch.empty_like(dst.flat)
        self.ops.copy2d(dst.flat.view(1, -1), tmp.view(1, -1))
        self.ops.axpy(tmp, 1.0 - mix, None, dst.flat, accumulate=False)
        self.ops.axpy(src.flat, mix, None, dst.flat, accumulate=True)
    self.slow_updates += 1

  def phase_actor(self):
    ops, b, cfg = self.ops, self.b, self.cfg
    N, H, M, D, S, A, F = self.N, self.H, self.M, self.D, self.S, self.A, self.F
    HN = H * N
    traj, dtraj = b['traj'], b['dtraj']
    feat = traj.view(M, self.TW)[:, :F]
    dfeat = dtraj.view(M, self.TW)[:, :F]
    ca = cfg['actor']
    lo, hi = ca['minstd'], ca['maxstd']
    # score with the post-update slow critic (reference agent.py:329-344).  It changes only
    # when update_slow copied (every slow_target_update steps): otherwise its values over
    # the trajectory are the ones phase_imagine left in the activation buffers, bit for bit
    self.plan.conditional(
        lambda: self.slow_copied,
        lambda: self.head_fwd('critic_target', self.acts_im['critic_target'], feat))
    val = self.acts_im['critic_target'][1][0].z
    rew = self.acts_im['reward'][1][0].z
    cont = self.acts_im['cont'][1][0].z
    ops.imag_returns_fwd(rew.view(-1), val.view(-1), cont.view(-1), b['cont_b'],
                         b['i_reward'], b['i_value2'], None, None, b['i_ret2'],
                         H, N, cfg['discount'], cfg['return_lambda'], cfg['actor_return'])
    ops.sub(b['i_ret2'], b['i_value2'], b['i_diff'][:HN])
    kr = self.stat('ret2', b['i_ret2'][:HN])
    kd = self.stat('diff', b['i_diff'][:HN])
    self.flush_stats()
    assert kd == kr + 1  # adjacent slots: one collective
    self.allreduce(self.stat_sums[kr:kd + 1])
    self.stat_prereduced.update((kr, kd))
    cnt = float(H * self.Ng)
    impl = {'off': 0, 'mean_std': 1, 'std': 2}
    c = cfg['retnorm']
    ops.normalize_update(self.norm_state['ret'], self.stat_sums[kr], cnt, None,
                         c['decay'], c['max'], impl[c['impl']], True, self.norm_os['ret'])
    c = cfg['scorenorm']
    ops.normalize_update(self.norm_state['score'], self.stat_sums[kd], cnt,
                         self.norm_os['ret'][1:2], c['decay'], c['max'],
                         impl[c['impl']], True, self.norm_os['score'])
    ops.scalar_mul(self.sc[0:1], self.norm_os['ret'][1:2], self.norm_os['score'][1:2], 1.0)
    c = cfg['advnorm']
    ops.normalize_update(self.norm_state['adv'], self.stat_sums[kd], cnt,
                         self.sc[0:1], c['decay'], c['max'], impl[c['impl']],
                         True, self.norm_os['adv'])
    ops.copy2d(self.norm_os['adv'].view(1, 2), self.sc[1:3].view(1, 2))
    if self.discrete:
      self._actor_reinforce(cnt)
    else:
      self._actor_backprop(cnt, rew, val, cont)
    self.stat('actor_loss_score', b['i_actor_loss'][:HN])
    self.stat('actor_loss_ent', b['i_ent_row'][:HN])
    self.flush_stats()
    self.opt_step('actor', 'actor_opt')

  def _actor_reinforce(self, cnt):
    """actor_grad 'reinforce' (reference agent.py:357-358): -log pi(a) * sg(score)
    plus the entropy regulariser (:372-377); no gradient through the rollout."""
    ops, b, cfg = self.ops, self.b, self.cfg
    N, H, M, A, F = self.N, self.H, self.M, self.A, self.F
    HN = H * N
    feat = b['traj'].view(M, self.TW)[:, :F]
    act = b['traj'].view(M, self.TW)[:, F:F + A]
    ent_div = math.log(A) if cfg['actent_norm'] else 1.0
    ops.onehot_entropy(b['alogit'], b['ent_norm'], ent_div)
    k = self.stat('actent', b['ent_norm'][:HN])
    self.flush_stats()
    self.allreduce(self.stat_sums[k])
    self.stat_prereduced.add(k)
    c = cfg['actent']
    if c['impl'] != 'fixed':
      ops.autoadapt_update(self.actent_scale[0:1], self.stat_sums[k], cnt,
                           c['target'], 0.1, c['vel'], c['min'], c['max'], True, c['impl'])
    ops.onehot_policy_grad(b['alogit'], act, b['i_ret2'], b['i_value2'],
                           b['i_weight'], self.sc, self.actent_scale[0:1],
                           b['dalogit'], b['i_actor_loss'], b['i_ent_row'], HN,
                           1.0 / cnt, ent_div)
    oa = self.acts_im['actor'][1][0]
    ops.stats_bwd(oa.z, b['dalogit'], None, oa.dout, 1, A,
                  float(cfg['actor']['unimix']))
    self.head_bwd('actor', self.acts_im['actor'], feat)

  def _actor_backprop(self, cnt, rew, val, cont):
    """actor_grad 'backprop' (reference agent.py:355-356, 361-371): gradient of
    -score through the heads and the imagined world model down to the actions."""
    ops, b, cfg = self.ops, self.b, self.cfg
    N, H, M, D, S, A, F = self.N, self.H, self.M, self.D, self.S, self.A, self.F
    HN = H * N
    traj, dtraj = b['traj'], b['dtraj']
    feat = traj.view(M, self.TW)[:, :F]
    dfeat = dtraj.view(M, self.TW)[:, :F]
    ca = cfg['actor']
    lo, hi = ca['minstd'], ca['maxstd']
    # entropy regulariser scale (AutoAdapt, inverse; reference agent.py:361-371)
    if cfg['actent_norm']:
      ent_lo, ent_div = math.log(lo), math.log(hi) - math.log(lo)
    else:
      ent_lo, ent_div = -0.5 * math.log(2 * math.pi * math.e), 1.0
    os_all = self.acts_im['actor'][1][1].z
    om_all = self.acts_im['actor'][1][0].z
    ops.actent_stats(os_all, HN, lo, hi, ent_lo, ent_div, self.actent_sums)
    self.allreduce(self.actent_sums)
    c = cfg['actent']
    if c['impl'] != 'fixed':
      ops.autoadapt_update(self.actent_scale, self.actent_sums, cnt, c['target'],
                           0.1, c['vel'], c['min'], c['max'], True, c['impl'])
    # seeds: d loss / d ret, d loss / d baseline
    ops.actor_seed(b['i_ret2'][:HN], b['i_value2'][:HN], b['i_weight'][:HN],
                   None, self.sc, b['i_actor_loss'][:HN], b['i_dret'][:HN],
                   b['i_dbase'][:HN], 1.0 / cnt)
    ops.imag_returns_bwd(b['i_dret'], b['i_dbase'], rew.view(-1), val.view(-1),
                         cont.view(-1), b['i_value2'], b['i_ret2'],
                         self.acts_im['reward'][1][0].dout.view(-1),
                         self.acts_im['critic_target'][1][0].dout.view(-1),
                         self.acts_im['cont'][1][0].dout.view(-1), H, N,
                         cfg['discount'], cfg['return_lambda'])
    ops.fill(dtraj, 0.0)
    # Gradient of the score w.r.t. every imagined latent state through the reward / cont /
    # target-critic heads (data gradients only), then the reverse scan through the imagined
    # world model.  Scan step t adds into row t-1, so the heads of row t-1 must be done before
    # it; apart from that the heads of a time chunk are independent of the scan: with a side
    # context the heads of chunk c-1 run next to the (latency-bound) scan over chunk c.
    def heads_bwd(t0, t1):
      r0, r1 = t0 * N, t1 * N
      sel = lambda buf: buf[r0:r1]
      x, dx = feat[r0:r1], dfeat[r0:r1]
      self.head_bwd('reward', self.acts_im['reward'], x, sel, dx, 0.0, params=False)
      self.head_bwd('cont', self.acts_im['cont'], x, sel, dx, 1.0, params=False)
      self.head_bwd('critic_target', self.acts_im['critic_target'], x, sel, dx, 1.0, params=False)
    zr = self.zero_rows[:N]
    def scan_step(t):
      si = lambda buf, t_=t - 1: buf.view(H, N, -1)[t_]
      ops.stats_bwd(si(self.ai_img_stats.z), None, dtraj[t][:, D:F],
                    si(self.ai_img_stats.dout), self.G, self.C, self.unimix)
      self.prior_bwd(None, dtraj[t][:, :D], self.ai_img_out, self.ai_img_stats,
                     si, params=False)
      self.core_bwd(dtraj[t][:, :D], traj[t - 1][:, :D], self.ai_img_in,
                    b['iz3'], b['igstats'], si, b['idz3'], b['idy3'], b['idh'],
                    dtraj[t - 1][:, D:F + A], 1.0, self.P['img_in'])
      ops.reset_mask_bwd(b['idh'], zr, dtraj[t - 1][:, :D])
    side = self.side_stream_b
    if self.fused_imag_bwd:
      # the reverse scan as one persistent launch (dd_imagine_rollout_bwd) after the bulk heads
      heads_bwd(0, H + 1)
      self.imagine_reverse_fused()
    elif self.ops_b2 is None or not self._in_b or not self.overlap_b:
      heads_bwd(0, H + 1)
      for t in reversed(range(1, H + 1)):
        scan_step(t)
    else:
      CH = self.HEAD_CHUNK
      bounds = list(range(0, H + 1, CH)) + [H + 1]
      chunks = list(zip(bounds[:-1], bounds[1:]))        # [(t0, t1)], ascending
      heads_bwd(*chunks[-1])
      for c in reversed(range(len(chunks))):
        t0, t1 = chunks[c]
        if c > 0:
          with (self.fork(side) if side is not None else contextlib.nullcontext()):
            keep, self.ops = self.ops, self.ops_b2
            try:
              heads_bwd(*chunks[c - 1])
            finally:
              self.ops = keep
        for t in reversed(range(max(t0, 1), t1)):
          if t == t0 and c > 0 and side is not None:
            self.join(side)        # row t0-1 belongs to the chunk evaluated on the side stream
          scan_step(t)
    # policy head + entropy bonus, then the actor network (bulk)
    dact = dtraj.view(M, self.TW)[:, F:F + A]
    oa = self.acts_im['actor'][1]
    ops.normal_head_bwd(om_all, os_all, b['eps'].view(M, A), dact, b['i_weight'],
                        self.actent_scale, oa[0].dout, oa[1].dout, b['i_ent_row'],
                        HN, lo, hi, 1.0 / (cnt * ent_div), ent_lo, ent_div)
    self.head_bwd('actor', self.acts_im['actor'], feat)

  # ------------------------------------------------------------------ report

  def openloop_device(self, ctx):
    """WorldModel.report's open-loop prediction (reference agent.py:266-282):
    after observe_fwd on this learner's batch, keep the posterior for the first
    `ctx` steps and roll the prior forward with the recorded actions for the
    rest (RSSM.imagine, nets.py:78-86); decode everything.  Returns the decoder's
    pre-sigmoid image tensor [B, T, H, W, C] (device)."""
    ops, b = self.ops, self.b
    B, T, D, S, A, F = self.B, self.T, self.D, self.S, self.A, self.F
    assert self.H * self.N >= (T - ctx) * B and self.spec.dec_convs
    post = b['post'].view(B, T, F)
    feat = b.get('report_feat')
    if feat is None:   # (first eager report: never allocated inside a captured segment)
      feat = b['report_feat'] = self.zeros(B, T, F)
    ops.copy2d(b['post'], feat.view(B * T, F))
    act = b['action'].view(B, T, A)
    tr = b['traj'].view(-1, self.TW)          # scratch rows: [deter | stoch | action]
    state = tr[:B]
    ops.copy2d(post[:, ctx - 1], state[:, :F])
    for i, t in enumerate(range(ctx, T)):
      nxt = tr[(i + 1) * B:(i + 2) * B]
      ops.copy2d(act[:, t], state[:, F:F + A])
      si = lambda buf, i_=i: buf[i_ * B:(i_ + 1) * B]
      self.core_fwd(state[:, D:F + A], state[:, :D], nxt[:, :D], self.ai_img_in,
                    b['iz3'], b['igstats'], si)
      xs = self.prior_fwd(nxt[:, :D], self.ai_img_out, self.ai_img_stats, si)
      ops.stats_fwd(xs, b['u_img'].view(-1, self.G)[i * B:(i + 1) * B],
                    b['ilogit'][:B], nxt[:, D:F], self.G, self.C, self.unimix, 0)
      ops.copy2d(nxt[:, :F], feat[:, t])
      state = nxt
    self.decoder_fwd(feat.view(B * T, F))
    z = self.dec_act[-1]['z']
    return z.view(B, T, *z.shape[1:])

  # ------------------------------------------------------------------ policy

  def policy_device(self, sample, noise=0.0):
    """Agent.policy (reference agent.py:42-65) on a [n, 1] batch held by this
    learner: encoder, one obs_step from the carried latent, actor, then sample
    (train / explore) or mode (eval), then tfutils.action_noise with amount `noise`
    (expl_noise / eval_noise, agent.py:50-63).  The new latent replaces the carry; the
    action is returned as a device view [n, A]."""
    ops, b, cfg = self.ops, self.b, self.cfg
    B, G, A, F, S = self.B, self.G, self.A, self.F, self.S
    assert self.T == 1
    ops.counter_add(self.step_ctr, 1)
    ops.batch_prep(b['is_first'], b['is_terminal'], b['action'], b['first'],
                   b['cont'], b['xin'][:, S:])
    ops.philox(b['u_prior'], B, 1, G, 1, 0, self.noise_seed, self.step_ctr, SITE_POLICY, 0)
    ops.philox(b['u_post'], 1, B, G, B, 0, self.noise_seed, self.step_ctr, SITE_POLICY + 1, 0)
    ops.philox(b['eps'][0], 1, B, A, B, 0, self.noise_seed, self.step_ctr, SITE_POLICY + 2, 1)
    self.encoder_fwd()
    self.initial_fwd()
    self.observe_fwd(True)
    t0 = b['traj'][0]
    ops.copy2d(b['post'], t0[:, :F])
    sel = lambda buf: buf.view(self.H + 1, self.N, -1)[0]
    ca = cfg['actor']
    if self.discrete:
      ops.philox(b['u_act'][0], 1, B, 1, B, 0, self.noise_seed, self.step_ctr, SITE_POLICY + 2, 0)
      (xa,) = self.head_fwd('actor', self.acts_im['actor'], t0[:, :F], sel)
      ops.stats_fwd(xa, b['u_act'][0], sel(b['alogit']), t0[:, F:F + A], 1, A,
                    float(ca['unimix']), 0 if sample else 1)
    else:
      om, os_ = self.head_fwd('actor', self.acts_im['actor'], t0[:, :F], sel)
      ops.normal_head_fwd(om, os_, b['eps'][0] if sample else None, t0[:, F:F + A],
                          ca['minstd'], ca['maxstd'])
    if noise:  # tfutils.py:85-93
      nz = b.get('act_noise')
      if nz is None:
        nz = b['act_noise'] = self.zeros(B, A)
      ops.philox(nz, 1, B, A, B, 0, self.noise_seed, self.step_ctr, SITE_POLICY + 3,
                 0 if self.discrete else 1)
      ops.action_noise(t0[:, F:F + A], nz, float(noise), self.discrete)
    ops.copy2d(b['post'], b['carry'])
    return t0[:, F:F + A]

  # --------------------------------------------------------------- train step

  def _stamp(self, slot):
    """Measurement aid (DD_STAMPS=1, tools/phase_timeline.py): the device clock at this point of
    the launch sequence, captured with it."""
    if self.stamps is not None:
      self.ops.stamp(self.stamps, slot)

  def phase_a1(self, use_carry=True):
    """World-model phase up to the gradients."""
    self._stamp(0)
    self.phase_prep()
    self.phase_wm_fwd(use_carry)
    self._stamp(1)
    self.phase_wm_bwd()
    self._stamp(2)

  def phase_b(self):
    """Behaviour phase: imagination, critic update, slow-critic copy, actor update.
    Reads world-model weights, writes actor / critic state only."""
    self.ops = self.ops_b
    self.comm = self.comm_b
    self._in_b = True
    try:
      self._stamp(4)
      self.phase_prep_b()
      self.phase_imagine()
      self._stamp(7)
      self.plan.cut(self.update_slow)
      self.phase_actor()
      self._stamp(8)
    finally:
      self.ops = self.ops_a
      self.comm = self.comm_a
      self._in_b = False

  def train_step_device(self, use_carry=True):
    """All device work of one Agent.train call (inputs already uploaded)."""
    if not use_carry:
      self.reset_carry()
      use_carry = True
    self.phase_a1(use_carry)
    self.phase_wm_opt()
    self.phase_b()

  def capture(self):
    """Capture the whole step into HIP graphs (after at least one eager step,
    so lazily registered statistics slots exist)."""
    plan = graphs.GraphPlan(self.device)
    self.plan = plan
    plan.capture(lambda: self.train_step_device(True))
    return plan

  def capture_pipeline(self):
    """The step as three separately replayable plans (world-model gradients | world-model
    optimizer + hand-over | behaviour phase) for the two-stream software pipeline of
    agent.Agent: step k's behaviour phase runs next to step k+1's world-model phase."""
    plans = []
    keep, self.overlap_b = self.overlap_b, False
    self._pipelined_capture = True
    try:
      for fn in (lambda: self.phase_a1(True), self.phase_wm_opt, self.phase_b):
        plan = graphs.GraphPlan(self.device)
        self.plan = plan
        plan.capture(fn)
        plans.append(plan)
    finally:
      self.overlap_b = keep
      self._pipelined_capture = False
    return plans

  def metric_tensors(self):
    """The device tensors a metrics read-out needs (name -> tensor)."""
    t = dict(sums=self.stat_sums, maxs=self.stat_maxs, wmkl=self.wmkl_scale, sc=self.sc,
             actent_scale=self.actent_scale, bal=self.bal)
    if getattr(self, 'scan_sync', None) is not None:
      t['scan_err'] = self.scan_sync[1:2]   # sticky error word of the persistent scan kernels
    for g in ('model', 'critic', 'actor'):
      t[f'opt_{g}'] = self.groups[g].opt_state
    if not self.discrete:
      t['actent_sums'] = self.actent_sums
    return t

  # which of them the behaviour phase writes (the rest belong to the world-model phase;
  # sums / maxs are split by row: stat_b_slots)
  METRIC_B = ('sc', 'actent_scale', 'opt_critic', 'opt_actor', 'actent_sums')

  def read_metrics(self, host=None, wm_only=False):
    """One device->host transfer of the statistics slabs -> metrics dict with
    the reference's names (agent.py:184-203, 339-342, 407-415, tfutils.py
    :208,250,266,445-446).  `host`: already fetched numpy copies of metric_tensors()
    (the pipelined agent snapshots them per phase).  wm_only: the world-model loss metrics
    alone (WorldModel.loss's metrics dict, what Agent.report starts from, agent.py:268)."""
    cfg = self.cfg
    direct = host is None
    if host is None:
      sums = self.stat_sums.clone()
      maxs = self.stat_maxs.clone()
      if self.comm is not None:
        self.comm.allreduce_sum(sums)
        self.comm.allreduce_max(maxs)
        for k in self.stat_prereduced:  # identical on every rank already
          sums[k] /= self.world
      host = {k: v.cpu().numpy() for k, v in self.metric_tensors().items()}
      host['sums'], host['maxs'] = sums.cpu().numpy(), maxs.cpu().numpy()
      if self.comm is not None:
        bal = self.bal.clone()
        self.comm.allreduce_sum(bal)
        host['bal'] = bal.cpu().numpy()
    if 'scan_err' in host and int(host['scan_err'][0]) != 0:
      # a fused observe scan ran on after a grid-barrier timeout (bit 0) or met a carried state
      # that was not one-hot (bit 1): its outputs are garbage.  Raise like check_numerics
      # (tfutils.py:207,249) instead of training on them; the word is cleared for the next step.
      code = int(host['scan_err'][0])
      if direct:   # (a pipelined agent's per-step snapshot clears the live word itself, in stream order)
        self.scan_sync[1:2].zero_()
      raise RuntimeError(f'fused observe scan failed (error word {code}: '
                         f'{"grid-barrier timeout" if code & 1 else "carried stoch not one-hot"})')
    sums, maxs = host['sums'], host['maxs']
    N, H, w = self.N, self.H, self.world
    counts = dict(imag_value=(H + 1) * N * w)
    for k in ('critic_loss', 'imag_critic', 'imag_reward', 'imag_return', 'ret2', 'diff',
              'actor_loss_score', 'actor_loss_ent', 'actent'):
      counts[k] = H * N * w
    st = {}
    for i, name in enumerate(self.stat_names):
      n = counts.get(name, N * w)
      mean = sums[i, 0] / n
      var = max(sums[i, 1] / n - mean * mean, 0.0)
      st[name] = dict(mean=mean, std=math.sqrt(var), sum=sums[i, 0],
                      max=maxs[i, 0], min=-maxs[i, 1], absmax=maxs[i, 2],
                      absmean=sums[i, 2] / n)
    mets = {}
    f = np.float32
    ls = cfg['loss_scales']
    wmkl = float(host['wmkl'][0])
    model_loss = 0.0
    for name in st:
      if name.endswith('_loss') and name not in ('critic_loss',):
        key = name[:-5]
        mets[f'{key}_loss_mean'] = st[name]['mean']
        mets[f'{key}_loss_std'] = st[name]['std']
        scale = wmkl if key == 'kl' else 1.0
        model_loss += ls.get(key, 1.0) * scale * st[name]['mean']
    mets['kl_loss_mean'] = wmkl * st['kl_loss']['mean']
    mets['kl_loss_std'] = wmkl * st['kl_loss']['std']
    mets['wmkl_mean'] = st['kl_loss']['mean']
    mets['wmkl_std'] = st['kl_loss']['std']
    mets['wmkl_scale_mean'] = wmkl
    mets['wmkl_scale_std'] = 0.0
    mets['prior_ent_mean'] = st['prior_ent']['mean']
    mets['post_ent_mean'] = st['post_ent']['mean']
    mets['prior_ent_min'] = st['prior_ent']['min']
    mets['post_ent_min'] = st['post_ent']['min']
    mets['model_loss_mean'] = model_loss
    mets['model_loss'] = model_loss
    mets['model_loss_std'] = st['model_total']['std']
    with np.errstate(divide='ignore', invalid='ignore'):  # NaN without positives / negatives
      for i, head in enumerate(('reward', 'cont')):
        s_ = host['bal'][i].astype(np.float64)
        n_ = float(N * w)
        mets[f'{head}_pos_loss'] = np.float64(s_[0]) / s_[4]
        mets[f'{head}_neg_loss'] = np.float64(s_[1]) / (n_ - s_[4])
        mets[f'{head}_pos_acc'] = np.float64(s_[2]) / s_[4]
        mets[f'{head}_neg_acc'] = np.float64(s_[3]) / (n_ - s_[4])
        mets[f'{head}_rate'] = s_[4] / n_
        mets[f'{head}_avg'] = s_[5] / n_
        mets[f'{head}_pred'] = s_[6] / n_
    if wm_only:   # (`model_loss` itself is the optimizer's metric, tfutils.py:209: not part of a report)
      return {k: np.asarray(v, f) for k, v in mets.items() if k != 'model_loss'}
    for gname, pre in (('model', ''), ('critic', 'extr_'), ('actor', '')):
      o = host[f'opt_{gname}']
      mets[f'{pre}{gname}_grad_norm'] = o[1]
      mets[f'{pre}{gname}_grad_steps'] = o[0]
      if self.mixed:   # tfutils.py:231-232, 246-247: overflow is a metric, the norm reads NaN
        mets[f'{pre}{gname}_grad_scale'] = o[3]
        mets[f'{pre}{gname}_grad_overflow'] = 1.0 - o[2]
        if o[2] == 0.0:
          mets[f'{pre}{gname}_grad_norm'] = float('nan')
      elif o[2] == 0.0:
        raise FloatingPointError(f'{gname}_norm is not finite')
    mets['extr_critic_loss'] = st['critic_loss']['mean']
    mets['extr_imag_reward_mean'] = st['imag_reward']['mean']
    mets['extr_imag_reward_std'] = st['imag_reward']['std']
    mets['extr_imag_critic_mean'] = st['imag_critic']['mean']    # agent.py:411-412
    mets['extr_imag_critic_std'] = st['imag_critic']['std']
    mets['extr_imag_return_mean'] = st['imag_return']['mean']
    mets['extr_imag_return_std'] = st['imag_return']['std']
    sc = host['sc']
    d = st['diff']
    mets['extr_score_mean'] = d['mean'] * sc[0]
    mets['extr_score_std'] = d['std'] * sc[0]
    mets['extr_score_mag'] = d['absmean'] * sc[0]
    mets['extr_score_max'] = d['absmax'] * sc[0]
    mets['actor_loss'] = (st['actor_loss_score']['mean'] +
                          st['actor_loss_ent']['mean'])
    cnt = H * N * w
    A = self.A
    if self.discrete:
      mets['actent_mean'] = st['actent']['mean']
      mets['actent_std'] = st['actent']['std']
      a = host['actent_scale'][0:1]
    else:
      asum = host['actent_sums']
      emean = asum[:A].sum() / (cnt * A)
      mets['actent_mean'] = emean
      mets['actent_std'] = math.sqrt(max(asum[A:].sum() / (cnt * A) - emean ** 2, 0.0))
      a = host['actent_scale']
    mets['actent_scale_mean'] = a.mean()
    mets['actent_scale_std'] = a.std()
    for k in ('model_loss', 'extr_critic_loss', 'actor_loss'):
      if not np.isfinite(mets[k]):
        raise FloatingPointError(f'{k} is not finite')
    return {k: np.asarray(v, f) for k, v in mets.items()}

  # ----------------------------------------------------------------- export

  CONTROLLER = ('wmkl_scale', 'actent_scale', 'step_ctr', 'step_ctr_b')

  def export_state(self):
    """Controller state that lives in this learner instance (not in the shared parameter
    arenas): AutoAdapt scales, Normalize moments, slow-critic counter, noise step."""
    st = {k: getattr(self, k).clone() for k in self.CONTROLLER}
    st.update({f'norm/{k}': v.clone() for k, v in self.norm_state.items()})
    st['slow_updates'] = self.slow_updates
    return st

  def import_state(self, st):
    for k in self.CONTROLLER:
      getattr(self, k).copy_(st[k])
    for k, v in self.norm_state.items():
      v.copy_(st[f'norm/{k}'])
    self.slow_updates = int(st['slow_updates'])

  def export_params(self):
    out = {}
    for g in self.groups.values():
      for p in g.specs:
        out[p.name] = g.p[p.name].detach().cpu().numpy().copy()
    return out

  def export_grads(self):
    out = {}
    for n in ('model', 'actor', 'critic'):
      g = self.groups[n]
      for p in g.specs:
        out[p.name] = g.g[p.name].detach().cpu().numpy().copy()
    return out
