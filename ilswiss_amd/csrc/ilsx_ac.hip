// ilsx_ac.hip — the two other off-policy actor-critic trainers of the hot path, on the same kernels as SAC-alpha:
//   TD3    rlkit/torch/algorithms/td3/td3.py:21-124,180-183 + MlpGaussianNoisePolicy (common/policies.py:130-188)
//   SAC-V  rlkit/torch/algorithms/sac/sac.py:23-179,242-243 (state-value variant, fixed temperature)
//
// One agent = one trainable arena P = [Q1 | Q2 | (V |) pi] with gradient / Adam-moment arenas of the same layout and a
// target arena T of the same layout (only the slices the algorithm has targets for are ever read).  Adam runs in the
// epilogue of the weight-gradient kernel (AdamFuse); V's launch also writes its Polyak target there; TD3's delayed
// soft update of all three targets is one streaming pass over the arena (k_ac_polyak).
//
//   TD3 step   : fwd{pi_tgt(s') + clipped noise} ; fwd{TQ1,TQ2(s',a'), Q1,Q2(s,a)} ; bwd{Q1,Q2: MSE to y} ; dW+Adam{Q1,Q2} ;
//                every `period`-th step: fwd{pi(s)} ; fwd{Q1(s,pi(s))} ; bwd{Q1 -> dQ/da} ; bwd{pi through tanh} ;
//                dW+Adam{pi} ; polyak{pi,Q1,Q2}
//   SAC-V step : fwd{Q1,Q2(s,a), TV(s'), V(s)} ; fwd{pi(s) sample} ; fwd{Q1,Q2(s,a~)} ; bwd{Q1,Q2: half-MSE to y} ;
//                bwd{V: half-MSE to min Q - alpha log pi} ; dW+Adam{Q1,Q2} ; dW+Adam+Polyak{V} ;
//                fwd{Q1,Q2(s,a~) updated} ; bwd{Q1,Q2 -> d(-min Q)/da~} ; bwd{pi: tanh-Gaussian head} ; dW+Adam{pi}
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "host_common.h"

#define AC_MAX_NETS 4
#define AC_MAX_OPT 3

struct AcScalars {
  float step[4], bc2s[4];
  int t[4];
};

// Adam bias-correction scalars of every optimiser's NEXT step; mask = optimisers that just stepped
__global__ void k_ac_tick(AcScalars* sc, DevScalars* dsc, int mask, int advance, float lr0, float lr1, float lr2, float b1) {
  const float lr[3] = {lr0, lr1, lr2};
  for (int i = 0; i < 3; ++i) {
    if (mask & (1 << i)) sc->t[i] += 1;
    const double t = (double)(sc->t[i] + 1);
    sc->step[i] = (float)((double)lr[i] / (1.0 - pow((double)b1, t)));
    sc->bc2s[i] = (float)sqrt(1.0 - pow(0.999, t));
  }
  if (advance) { dsc->step += 1; dsc->gather_step += 1; }
}

// ptu.soft_update_from_to (pytorch_util.py:10-12) over a contiguous arena slice
__global__ __launch_bounds__(256) void k_ac_polyak(const float* __restrict__ p, float* __restrict__ t, int n, float tau) {
  const float otau = 1.0f - tau;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) t[i] = t[i] * otau + p[i] * tau;
}

struct AcAgent {
  ilsx_ctx* ctx = nullptr;
  int nnets = 0, max_batch = 0, o = 0, a = 0, H = 0, act = 0, B = 0;
  NetLayout L[AC_MAX_NETS];
  ilsx_net* adopted[AC_MAX_NETS] = {nullptr, nullptr, nullptr, nullptr};
  size_t off[AC_MAX_NETS + 1] = {0, 0, 0, 0, 0};
  int opt_of[AC_MAX_NETS] = {0, 0, 0, 0};
  float lr[AC_MAX_OPT] = {0, 0, 0};
  float beta_1 = 0.9f;
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *T = nullptr;
  AcScalars* sc = nullptr;
  DevScalars* dsc = nullptr;
  uint32_t rng_stream = 0;
  int cs = 1;   // column-split factor of the 2-hidden-layer fast path (1 = generic kernels)
  // batch staging + per-net activation workspaces
  float *s = nullptr, *ac = nullptr, *r = nullptr, *d = nullptr, *s2 = nullptr, *eps = nullptr;
  bool eps_explicit = false;
  float *x[AC_MAX_NETS], *h[AC_MAX_NETS][ILSX_MAX_HID], *dl[AC_MAX_NETS][ILSX_MAX_HID], *dh[AC_MAX_NETS];
  DwArgs jobs[AC_MAX_NETS];
  float* p(int i) const { return P + off[i]; }
  float* tgt(int i) const { return T + off[i]; }
  NetView view(int i, bool target = false) const { return net_view(L[i], (target ? T : P) + off[i]); }
};

static int ac_alloc(AcAgent* g, float** q, size_t n) { return ctx_alloc(g->ctx, n * sizeof(float), (void**)q, true); }

// nets[i] are adopted into the arena in the given order; every net must share hidden width / depth / activation
static int ac_init(AcAgent* g, ilsx_ctx* ctx, ilsx_net* const* nets, int nnets, const int* opt_of, int max_batch, int o, int a) {
  g->ctx = ctx; g->nnets = nnets; g->max_batch = max_batch; g->o = o; g->a = a;
  const ilsx_mlp_cfg& c0 = nets[0]->lay.cfg;
  g->H = c0.hidden; g->act = c0.act;
  g->cs = getenv("ILSX_NO_SPLIT") ? 1 : mlp2_split_factor(c0.n_hidden, c0.hidden);
  for (int i = 0; i < nnets; ++i) {
    const ilsx_mlp_cfg& c = nets[i]->lay.cfg;
    if (nets[i]->ctx != ctx) ILSX_FAIL(ILSX_ERR_ARG, "a network belongs to another ctx");
    if (!nets[i]->owns) ILSX_FAIL(ILSX_ERR_STATE, "a network already belongs to an agent");
    if (c.hidden != c0.hidden || c.n_hidden != c0.n_hidden || c.act != c0.act)
      ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "all networks of an agent must share hidden width/depth/activation");
    g->L[i] = nets[i]->lay;
    g->off[i + 1] = g->off[i] + g->L[i].n_int;
    g->opt_of[i] = opt_of[i];
  }
  const size_t n = g->off[nnets], B = (size_t)max_batch, H = (size_t)g->H;
  ILSX_TRY(ac_alloc(g, &g->P, n)); ILSX_TRY(ac_alloc(g, &g->G, n)); ILSX_TRY(ac_alloc(g, &g->M, n));
  ILSX_TRY(ac_alloc(g, &g->V, n)); ILSX_TRY(ac_alloc(g, &g->T, n));
  ILSX_TRY(ctx_alloc(ctx, sizeof(AcScalars), (void**)&g->sc));
  ILSX_TRY(ctx_alloc(ctx, sizeof(DevScalars), (void**)&g->dsc));
  ILSX_TRY(ac_alloc(g, &g->s, B * o)); ILSX_TRY(ac_alloc(g, &g->ac, B * a)); ILSX_TRY(ac_alloc(g, &g->r, B));
  ILSX_TRY(ac_alloc(g, &g->d, B)); ILSX_TRY(ac_alloc(g, &g->s2, B * o)); ILSX_TRY(ac_alloc(g, &g->eps, B * a));
  hipStream_t st = ctx->stream;
  for (int i = 0; i < nnets; ++i) {
    ILSX_TRY(ac_alloc(g, &g->x[i], B * g->L[i].KP));
    for (int l = 0; l < g->L[i].cfg.n_hidden; ++l) { ILSX_TRY(ac_alloc(g, &g->h[i][l], B * H)); ILSX_TRY(ac_alloc(g, &g->dl[i][l], B * H)); }
    ILSX_TRY(ac_alloc(g, &g->dh[i], B * std::max(4, g->L[i].NO)));
    HIPCHK(hipMemcpyAsync(g->p(i), nets[i]->base, g->L[i].n_int * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(g->tgt(i), nets[i]->base, g->L[i].n_int * 4, hipMemcpyDeviceToDevice, st));   // net.copy()
    memset(&g->jobs[i], 0, sizeof g->jobs[i]);
    ILSX_TRY(build_dw_jobs(g->L[i], g->G + g->off[i], g->x[i], g->h[i], g->dl[i], g->dh[i], &g->jobs[i]));
  }
  HIPCHK(hipStreamSynchronize(st));
  for (int i = 0; i < nnets; ++i) {
    ILSX_TRY(ctx_free(ctx, nets[i]->base));
    nets[i]->base = g->p(i);
    nets[i]->owns = false;
    g->adopted[i] = nets[i];
  }
  g->rng_stream = ctx->next_rng_stream;
  ctx->next_rng_stream += 2;
  return ILSX_OK;
}

static void ac_release(AcAgent* g) {
  hipSetDevice(g->ctx->device);
  hipStreamSynchronize(g->ctx->stream);
  for (int i = 0; i < g->nnets; ++i) {  // give the networks private storage back so their handles stay usable
    float* nb = nullptr;
    if (g->adopted[i] && ctx_alloc(g->ctx, g->L[i].n_int * 4, (void**)&nb, false) == ILSX_OK) {
      hipMemcpyAsync(nb, g->p(i), g->L[i].n_int * 4, hipMemcpyDeviceToDevice, g->ctx->stream);
      hipStreamSynchronize(g->ctx->stream);
      g->adopted[i]->base = nb;
      g->adopted[i]->owns = true;
    }
  }
  ctx_free(g->ctx, g->P); ctx_free(g->ctx, g->G); ctx_free(g->ctx, g->M); ctx_free(g->ctx, g->V); ctx_free(g->ctx, g->T);
}

static int ac_tick(AcAgent* g, int mask, int advance) {
  hipLaunchKernelGGL(k_ac_tick, dim3(1), dim3(1), 0, g->ctx->stream, g->sc, g->dsc, mask, advance, g->lr[0], g->lr[1], g->lr[2], g->beta_1);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

static int ac_stage(AcAgent* g, const float* obs, const float* act, const float* rew, const float* done, const float* nobs, int B,
                    const float* eps) {
  if (!obs || !act || !rew || !done || !nobs) ILSX_FAIL(ILSX_ERR_ARG, "train_step: NULL batch pointer");
  if (B < 1 || B > g->max_batch) ILSX_FAIL(ILSX_ERR_ARG, "B=%d not in 1..max_batch=%d", B, g->max_batch);
  HIPCHK(hipSetDevice(g->ctx->device));
  hipStream_t st = g->ctx->stream;
  const size_t b = (size_t)B * 4;
  HIPCHK(hipMemcpyAsync(g->s, obs, b * g->o, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(g->ac, act, b * g->a, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(g->r, rew, b, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(g->d, done, b, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(g->s2, nobs, b * g->o, hipMemcpyDeviceToDevice, st));
  g->eps_explicit = eps != nullptr;
  if (eps) HIPCHK(hipMemcpyAsync(g->eps, eps, b * g->a, hipMemcpyDeviceToDevice, st));
  g->B = B;
  return ILSX_OK;
}

// forward task builders
static void ac_fwd(AcAgent* g, FwdTask& t, int net, bool target, const float* x0, int d0, const float* x1, int d1, bool save) {
  t.net = g->view(net, target);
  t.x0 = x0; t.d0 = d0; t.s0 = d0; t.x1 = x1; t.d1 = d1; t.s1 = d1;
  if (save) {
    t.xsave = g->x[net];
    for (int l = 0; l < g->L[net].cfg.n_hidden; ++l) t.hsave[l] = g->h[net][l];
  }
  t.head = HEAD_RAW;
}
static FwdArgs ac_fwd_args(AcAgent* g, int ntasks) {
  FwdArgs A;
  memset(&A, 0, sizeof A);
  A.rows = g->B; A.ntasks = ntasks; A.seed = g->ctx->seed; A.scal = g->dsc; A.part_stride = g->max_batch;
  return A;
}
// a network's scalar / vector output: whole (generic kernels) or as cs column-slice partial slabs (split kernels)
static void ac_out(AcAgent* g, FwdTask& t, float* buf) { if (g->cs > 1) t.part = buf; else t.out = buf; }
static int ac_launch_fwd(AcAgent* g, const FwdArgs& A, int KP) { return launch_fwd(g->ctx, A, g->H, g->act, KP, g->cs); }
static int ac_launch_bwd(AcAgent* g, const BwdArgs& A) { return launch_bwd_dx(g->ctx, A, g->H, g->act, g->cs); }
static BwdArgs ac_bwd_args(AcAgent* g, int ntasks, float gamma, float reward_scale) {
  BwdArgs A;
  memset(&A, 0, sizeof A);
  A.rows = g->B; A.ntasks = ntasks; A.inv_B = 1.0f / (float)g->B; A.gamma = gamma; A.reward_scale = reward_scale;
  A.scal = g->dsc; A.ga_parts = g->cs; A.ga_stride = g->max_batch; A.part_stride = g->max_batch;
  return A;
}
static void ac_bwd(AcAgent* g, BwdTask& t, int net, bool save_d) {
  t.net = g->view(net);
  for (int l = 0; l < g->L[net].cfg.n_hidden; ++l) { t.hsave[l] = g->h[net][l]; if (save_d) t.dsave[l] = g->dl[net][l]; }
  if (save_d) t.dhead = g->dh[net];
}
static int ac_dw_adam(AcAgent* g, int net0, int nnet, bool polyak, float tau) {
  DwArgs table;
  memset(&table, 0, sizeof table);
  for (int i = net0; i < net0 + nnet; ++i)
    ILSX_TRY(build_dw_jobs(g->L[i], g->G + g->off[i], g->x[i], g->h[i], g->dl[i], g->dh[i], &table));
  const int op = g->opt_of[net0];
  AdamFuse F;
  memset(&F, 0, sizeof F);
  F.on = 1; F.Gbase = g->G; F.P = g->P; F.M = g->M; F.V = g->V; F.T = polyak ? g->T : nullptr;
  F.b1 = g->beta_1; F.b2 = 0.999f; F.eps = 1e-8f; F.tau = tau;
  F.step_size = &g->sc->step[op]; F.bc2_sqrt = &g->sc->bc2s[op];
  return launch_bwd_dw(g->ctx, table, g->B, &F);
}
static PartVal pv(AcAgent* g, const float* p) { return PartVal{p, g->cs, g->max_batch}; }

static int ac_params(AcAgent* g, int which, bool set, float* host, size_t n) {
  if (!host || which < 0 || which >= 2 * g->nnets) ILSX_FAIL(ILSX_ERR_ARG, "parameter block %d out of range", which);
  HIPCHK(hipSetDevice(g->ctx->device));
  const int i = which % g->nnets;
  float* base = which < g->nnets ? g->p(i) : g->tgt(i);
  return set ? net_upload_flat(g->ctx, g->L[i], base, host, n, 0) : net_download_flat(g->ctx, g->L[i], base, host, n, 0);
}

// optimiser state of trainable block `which` (snapshots / resume): Adam moments in the block's flat ABI layout + the step count
// of the optimiser that owns it + the agent's Philox counter
static int ac_opt(AcAgent* g, int which, bool set, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!m_host || !v_host || which < 0 || which >= g->nnets) ILSX_FAIL(ILSX_ERR_ARG, "optimiser block %d out of range", which);
  HIPCHK(hipSetDevice(g->ctx->device));
  hipStream_t st = g->ctx->stream;
  float *M = g->M + g->off[which], *V = g->V + g->off[which];
  if (set) {
    ILSX_TRY(net_upload_flat(g->ctx, g->L[which], M, m_host, n, 0));
    ILSX_TRY(net_upload_flat(g->ctx, g->L[which], V, v_host, n, 0));
  } else {
    ILSX_TRY(net_download_flat(g->ctx, g->L[which], M, m_host, n, 0));
    ILSX_TRY(net_download_flat(g->ctx, g->L[which], V, v_host, n, 0));
  }
  if (!meta) return ILSX_OK;
  AcScalars hs; DevScalars hd;
  HIPCHK(hipMemcpyAsync(&hs, g->sc, sizeof hs, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&hd, g->dsc, sizeof hd, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!set) { meta->t = hs.t[g->opt_of[which]]; meta->rng_step = hd.step; return ILSX_OK; }
  hs.t[g->opt_of[which]] = (int)meta->t;
  hd.step = meta->rng_step; hd.gather_step = meta->rng_step;
  HIPCHK(hipMemcpyAsync(g->sc, &hs, sizeof hs, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(g->dsc, &hd, sizeof hd, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return ac_tick(g, 0, 0);   // bias-correction scalars of the next step from the restored counters
}

static int ac_fetch(AcAgent* g, const float* dev, size_t n, std::vector<float>& out) {
  out.resize(n);
  HIPCHK(hipMemcpyAsync(out.data(), dev, n * 4, hipMemcpyDeviceToHost, g->ctx->stream));
  return ILSX_OK;
}
// per-row scalar that may sit in cs partial slabs of max_batch rows: fetch and combine in slab order (synchronises)
static int ac_fetch_pv(AcAgent* g, const float* dev, size_t B, std::vector<float>& out) {
  std::vector<float> raw((size_t)g->cs * g->max_batch);
  HIPCHK(hipMemcpyAsync(raw.data(), dev, raw.size() * 4, hipMemcpyDeviceToHost, g->ctx->stream));
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  out.assign(B, 0.0f);
  for (size_t i = 0; i < B; ++i) {
    float v = raw[i];
    for (int c = 1; c < g->cs; ++c) v += raw[(size_t)c * g->max_batch + i];
    out[i] = v;
  }
  return ILSX_OK;
}
// create_stats_ordered_dict (core/eval_util.py): mean, population std, max, min
static void msmm(const float* v, size_t n, float out[4]) {
  double s = 0, ss = 0;
  float mx = -INFINITY, mn = INFINITY;
  for (size_t i = 0; i < n; ++i) { s += v[i]; mx = std::max(mx, v[i]); mn = std::min(mn, v[i]); }
  const double mean = s / (double)n;
  for (size_t i = 0; i < n; ++i) ss += (v[i] - mean) * (v[i] - mean);
  out[0] = (float)mean; out[1] = (float)std::sqrt(ss / (double)n); out[2] = mx; out[3] = mn;
}
static float mean_sq_diff(const std::vector<float>& a, const std::vector<float>& b, std::vector<float>* err = nullptr) {
  double s = 0;
  if (err) err->resize(a.size());
  for (size_t i = 0; i < a.size(); ++i) {
    const float e = (a[i] - b[i]) * (a[i] - b[i]);
    if (err) (*err)[i] = e;
    s += e;
  }
  return (float)(s / (double)a.size());
}

static void ac_policy_fin(AcAgent* g, FwdArgs& A, int head, const float* part, const float* eps, float* raw, float* action,
                          float* logp, float* eps_save, float noise, float noise_clip, float max_act) {
  if (g->cs == 1) return;   // generic kernels finish the head in their own epilogue
  PolicyFinishArgs& P = A.fin;
  memset(&P, 0, sizeof P);
  P.part = part; P.cs = g->cs; P.part_stride = g->max_batch; P.rows = g->B; P.a = g->a; P.head = head;
  P.rng_stream = g->rng_stream; P.seed = g->ctx->seed; P.scal = g->dsc;
  P.eps = eps; P.raw = raw; P.action = action; P.logp = logp; P.eps_save = eps_save;
  P.noise = noise; P.noise_clip = noise_clip; P.max_act = max_act;
  A.fin_on = 1;
}

// ================================================================================================ TD3
enum { T3_Q1 = 0, T3_Q2 = 1, T3_PI = 2 };
struct ilsx_td3 {
  AcAgent g;
  ilsx_td3_cfg cfg;
  long long n_steps = 0;
  float *a2 = nullptr, *q1 = nullptr, *q2 = nullptr, *tq1 = nullptr, *tq2 = nullptr;   // critic phase
  float *pa = nullptr, *pre = nullptr, *qn = nullptr, *ga = nullptr;                     // actor phase
  float *ppt = nullptr, *ppc = nullptr;   // head partials of pi_tgt(s') / pi(s) (column-split path)
  float* tqc = nullptr;                   // HER: clip(min(TQ1, TQ2), l, r) per row
  bool out_linear = false;                // the policy's output activation is the identity (ilsx_net_set_output_linear before create)
  int det_head() const { return out_linear ? HEAD_DET_LIN_NOISE : HEAD_DET_TANH_NOISE; }
};

extern "C" int ilsx_td3_create(ilsx_ctx* ctx, const ilsx_td3_cfg* cfg, ilsx_net* pi, ilsx_net* q1, ilsx_net* q2, ilsx_td3** out) {
  if (!ctx || !cfg || !pi || !q1 || !q2 || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_td3_create: NULL argument");
  const ilsx_mlp_cfg &cp = pi->lay.cfg, &c1 = q1->lay.cfg, &c2 = q2->lay.cfg;
  if (cp.n_heads != 1) ILSX_FAIL(ILSX_ERR_ARG, "TD3 policy is a single-head Mlp with tanh output (policies.py:130-188)");
  if (c1.n_heads != 1 || c1.out_dim != 1 || memcmp(&c1, &c2, sizeof c1) != 0) ILSX_FAIL(ILSX_ERR_ARG, "qf1/qf2 must be identical single-output FlattenMlp's");
  if (c1.in_dim != cp.in_dim + cp.out_dim) ILSX_FAIL(ILSX_ERR_ARG, "qf input %d != obs %d + act %d", c1.in_dim, cp.in_dim, cp.out_dim);
  if (cfg->max_batch < 1 || cfg->max_batch > (1 << 20)) ILSX_FAIL(ILSX_ERR_ARG, "max_batch=%d out of range", cfg->max_batch);
  if (cfg->policy_and_target_update_period < 1) ILSX_FAIL(ILSX_ERR_ARG, "policy_and_target_update_period must be >= 1");
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_td3* t = new ilsx_td3();
  t->cfg = *cfg;
  ilsx_net* nets[3] = {q1, q2, pi};
  const int opt_of[3] = {0, 0, 1};
  AcAgent* g = &t->g;
  g->lr[0] = cfg->qf_lr; g->lr[1] = cfg->policy_lr; g->beta_1 = 0.9f;   // optimizer_class defaults, td3.py:56-67
  int rc = ac_init(g, ctx, nets, 3, opt_of, cfg->max_batch, cp.in_dim, cp.out_dim);
  const size_t B = (size_t)cfg->max_batch, a = (size_t)cp.out_dim;
  const size_t CS = (size_t)g->cs;
  float** bufs[] = {&t->q1, &t->q2, &t->tq1, &t->tq2, &t->qn};
  for (float** b : bufs) if (rc == ILSX_OK) rc = ac_alloc(g, b, CS * B);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->a2, B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->pa, B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->pre, B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->ga, CS * B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->ppt, CS * B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &t->ppc, CS * B * a);
  if (rc == ILSX_OK && cfg->her) rc = ac_alloc(g, &t->tqc, B);
  if (rc == ILSX_OK) rc = ac_tick(g, 0, 0);
  if (rc != ILSX_OK) { delete t; return rc; }
  pi->noise = cfg->policy_noise; pi->noise_clip = cfg->policy_noise_clip; pi->max_act = cfg->max_act; pi->noise_policy = true;
  t->out_linear = pi->out_linear;
  *out = t;
  return ILSX_OK;
}
extern "C" int ilsx_td3_destroy(ilsx_td3* t) {
  if (!t) return ILSX_OK;
  ac_release(&t->g);
  delete t;
  return ILSX_OK;
}

// ---- HER-TD3 (rlkit/torch/algorithms/her/td3.py:100-170), the three places where it departs from td3.py, as side kernels so that
// the fused MLP kernels stay as they are:
//  * target action (:104-114): the reference adds sigma * N(0,1) to the target policy's output and then clamps THE NOISE, not the sum
//    (`torch.clamp(noise, min_act, max_act)`), so the action the target critics see is clamp(sigma * eps) and the target policy's
//    forward pass has no effect — followed as written;
//  * target value (:116-122): clip(min(TQ1, TQ2), clip_return_l, clip_return_r);
//  * policy loss (:148-152): -mean(Q1(s, pi(s))) + mean(pi(s)^2), i.e. dL/da += 2 a / (B * A).
__global__ void k_her_target_action(float* __restrict__ a2, const float* __restrict__ eps, int B, int a, float sigma, float max_act,
                                    uint64_t seed, const DevScalars* scal, uint32_t stream) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * a) return;
  const int gr = i / a, j = i - gr * a;
  float e;
  if (eps) {
    e = eps[i];
  } else {
    float z4[4];
    philox_normal4(seed, scal->step, stream, gr, j >> 2, z4);
    const int q = j & 3;
    e = q == 0 ? z4[0] : q == 1 ? z4[1] : q == 2 ? z4[2] : z4[3];
  }
  a2[i] = fminf(fmaxf(sigma * e, -max_act), max_act);
}
__global__ void k_her_clip_target(PartVal tq1, PartVal tq2, float* __restrict__ out, int B, float lo, float hi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = fminf(fmaxf(fminf(tq1.get(i), tq2.get(i)), lo), hi);
}
__global__ void k_her_l2_grad(float* __restrict__ ga, const float* __restrict__ pa, int n, float coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ga[i] += coef * pa[i];
}

// policy trunk as a forward task; generic kernels finish the head (tanh + clipped noise) in their epilogue, the column-split
// kernels leave head partials that the consumer launch finishes (td3_policy_fin)
static void td3_policy_task(ilsx_td3* t, FwdTask& f, bool target, const float* obs, bool noisy, bool save, float* action, float* pre,
                            float* part) {
  AcAgent* g = &t->g;
  ac_fwd(g, f, T3_PI, target, obs, g->o, nullptr, 0, save);
  f.head = t->det_head(); f.rng_stream = g->rng_stream;
  if (g->cs > 1) { f.part = part; return; }
  f.action = action; f.out = pre;
  f.max_act = t->cfg.max_act; f.noise = noisy ? t->cfg.policy_noise : 0.0f; f.noise_clip = t->cfg.policy_noise_clip;
  f.eps = (noisy && g->eps_explicit) ? g->eps : nullptr;
}
static void td3_policy_fin(ilsx_td3* t, FwdArgs& A, bool noisy, const float* part, float* action, float* pre) {
  AcAgent* g = &t->g;
  ac_policy_fin(g, A, t->det_head(), part, (noisy && g->eps_explicit) ? g->eps : nullptr, pre, action, nullptr, nullptr,
                noisy ? t->cfg.policy_noise : 0.0f, t->cfg.policy_noise_clip, t->cfg.max_act);
}

// Q1(s, pi(s)) with the updated qf1 (td3.py:111-113); pi(s)'s trunk already ran in the step's first launch
static int td3_actor_q(ilsx_td3* t, bool save) {
  AcAgent* g = &t->g;
  FwdArgs A = ac_fwd_args(g, 1);
  ac_fwd(g, A.t[0], T3_Q1, false, g->s, g->o, t->pa, g->a, false);
  if (save) for (int l = 0; l < g->L[T3_Q1].cfg.n_hidden; ++l) A.t[0].hsave[l] = g->h[T3_Q1][l];
  ac_out(g, A.t[0], t->qn);
  td3_policy_fin(t, A, false, t->ppc, t->pa, t->pre);   // deterministic=True
  return ac_launch_fwd(g, A, g->L[T3_Q1].KP);
}

static int td3_step(ilsx_td3* t, ilsx_td3_stats* stats) {
  AcAgent* g = &t->g;
  const ilsx_td3_cfg& c = t->cfg;
  const bool update = t->n_steps % c.policy_and_target_update_period == 0;
  const bool actor = update || stats != nullptr;
  const int na = g->B * g->a;
  if (c.her) {   // her/td3.py:104-114: the target action is the clamped noise alone ; pi(s) still runs when the actor needs it
    hipLaunchKernelGGL(k_her_target_action, dim3((na + 255) / 256), dim3(256), 0, g->ctx->stream, t->a2, g->eps_explicit ? (const float*)g->eps : nullptr,
                       g->B, g->a, c.policy_noise, c.max_act, g->ctx->seed, (const DevScalars*)g->dsc, g->rng_stream);
    HIPCHK(hipGetLastError());
    if (actor) {
      FwdArgs A = ac_fwd_args(g, 1);
      td3_policy_task(t, A.t[0], false, g->s, false, update, t->pa, t->pre, t->ppc);
      ILSX_TRY(ac_launch_fwd(g, A, g->L[T3_PI].KP));
    }
  } else {  // noisy target action (td3.py:84-85; the target is policy.copy() and keeps its noise) ; pi(s) rides along when needed
    FwdArgs A = ac_fwd_args(g, actor ? 2 : 1);
    td3_policy_task(t, A.t[0], true, g->s2, true, false, t->a2, nullptr, t->ppt);
    if (actor) td3_policy_task(t, A.t[1], false, g->s, false, update, t->pa, t->pre, t->ppc);
    ILSX_TRY(ac_launch_fwd(g, A, g->L[T3_PI].KP));
  }
  {
    FwdArgs A = ac_fwd_args(g, 4);
    ac_fwd(g, A.t[0], T3_Q1, true, g->s2, g->o, t->a2, g->a, false); ac_out(g, A.t[0], t->tq1);
    ac_fwd(g, A.t[1], T3_Q2, true, g->s2, g->o, t->a2, g->a, false); ac_out(g, A.t[1], t->tq2);
    ac_fwd(g, A.t[2], T3_Q1, false, g->s, g->o, g->ac, g->a, true); ac_out(g, A.t[2], t->q1); A.t[2].no_fin = 1;
    ac_fwd(g, A.t[3], T3_Q2, false, g->s, g->o, g->ac, g->a, true); ac_out(g, A.t[3], t->q2); A.t[3].no_fin = 1;
    if (!c.her) td3_policy_fin(t, A, true, t->ppt, t->a2, nullptr);
    ILSX_TRY(ac_launch_fwd(g, A, g->L[T3_Q1].KP));
  }
  if (c.her) {
    hipLaunchKernelGGL(k_her_clip_target, dim3((g->B + 255) / 256), dim3(256), 0, g->ctx->stream, pv(g, t->tq1), pv(g, t->tq2), t->tqc, g->B,
                       c.clip_return_l, c.clip_return_r);
    HIPCHK(hipGetLastError());
  }
  {
    BwdArgs A = ac_bwd_args(g, 2, c.discount, c.reward_scale);
    for (int i = 0; i < 2; ++i) {
      BwdTask& b = A.t[i];
      ac_bwd(g, b, i, true);
      b.loss = LOSS_TD_CRITIC; b.coef = 2.0f;   // nn.MSELoss: no 1/2 (td3.py:36-37,92-97)
      b.q = pv(g, i == 0 ? t->q1 : t->q2);
      if (c.her) b.tq1 = b.tq2 = PartVal{t->tqc, 1, g->max_batch};
      else { b.tq1 = pv(g, t->tq1); b.tq2 = pv(g, t->tq2); }
      b.rew = g->r; b.done = g->d;
    }
    ILSX_TRY(ac_launch_bwd(g, A));
  }
  ILSX_TRY(ac_dw_adam(g, T3_Q1, 2, false, 0.f));
  if (actor) ILSX_TRY(td3_actor_q(t, update));
  if (update) {  // td3.py:109-122
    {
      BwdArgs A = ac_bwd_args(g, 1, 0.f, 0.f);
      BwdTask& b = A.t[0];
      ac_bwd(g, b, T3_Q1, false);
      b.loss = LOSS_CONST; b.coef = -1.0f;
      b.dx = t->ga; b.dx_col0 = g->o; b.dx_cols = g->a;
      ILSX_TRY(ac_launch_bwd(g, A));
    }
    if (c.her) {   // + mean(a^2): 2 a / (B A) added to the first partial slab of dQ/da
      hipLaunchKernelGGL(k_her_l2_grad, dim3((na + 255) / 256), dim3(256), 0, g->ctx->stream, t->ga, (const float*)t->pa, na, 2.0f / (float)na);
      HIPCHK(hipGetLastError());
    }
    {
      BwdArgs A = ac_bwd_args(g, 1, 0.f, 0.f);
      BwdTask& b = A.t[0];
      ac_bwd(g, b, T3_PI, true);
      b.loss = LOSS_TD3_POLICY; b.coef = c.max_act; b.which = t->out_linear ? 1 : 0; b.raw = t->pre; b.ga1 = t->ga;
      ILSX_TRY(ac_launch_bwd(g, A));
    }
    ILSX_TRY(ac_dw_adam(g, T3_PI, 1, false, 0.f));
    hipLaunchKernelGGL(k_ac_polyak, dim3(256), dim3(256), 0, g->ctx->stream, (const float*)g->P, g->T, (int)g->off[3], c.soft_target_tau);
    HIPCHK(hipGetLastError());
  }
  ILSX_TRY(ac_tick(g, update ? 3 : 1, 1));
  t->n_steps += 1;
  if (stats) {  // td3.py:124-176; the device buffers still hold this step's predictions (taken before the Adam steps)
    const size_t B = (size_t)g->B;
    std::vector<float> q1, q2, tq1, tq2, r, d, qn, pa;
    ILSX_TRY(ac_fetch_pv(g, t->q1, B, q1)); ILSX_TRY(ac_fetch_pv(g, t->q2, B, q2)); ILSX_TRY(ac_fetch_pv(g, t->tq1, B, tq1));
    ILSX_TRY(ac_fetch_pv(g, t->tq2, B, tq2)); ILSX_TRY(ac_fetch(g, g->r, B, r)); ILSX_TRY(ac_fetch(g, g->d, B, d));
    ILSX_TRY(ac_fetch_pv(g, t->qn, B, qn)); ILSX_TRY(ac_fetch(g, t->pa, B * g->a, pa));
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    std::vector<float> y(B), e1, e2;
    for (size_t i = 0; i < B; ++i) {
      float tq = std::min(tq1[i], tq2[i]);
      if (c.her) tq = std::min(std::max(tq, c.clip_return_l), c.clip_return_r);
      y[i] = c.reward_scale * r[i] + (1.0f - d[i]) * c.discount * tq;
    }
    stats->qf1_loss = mean_sq_diff(q1, y, &e1);
    stats->qf2_loss = mean_sq_diff(q2, y, &e2);
    double s = 0;
    for (float v : qn) s += v;
    stats->policy_loss = (float)(-s / (double)B);
    if (c.her && update) {   // her/td3.py:150-152; a step without a policy update reports -mean(Q) alone (:165-170)
      double l2 = 0;
      for (float v : pa) l2 += (double)v * v;
      stats->policy_loss += (float)(l2 / (double)pa.size());
    }
    msmm(q1.data(), B, stats->q1_pred); msmm(q2.data(), B, stats->q2_pred); msmm(y.data(), B, stats->q_target);
    msmm(e1.data(), B, stats->bellman1); msmm(e2.data(), B, stats->bellman2); msmm(pa.data(), pa.size(), stats->policy_action);
  }
  return ILSX_OK;
}

extern "C" int ilsx_td3_train_step(ilsx_td3* t, const float* obs, const float* act, const float* rew, const float* done,
                                   const float* nobs, int B, const float* eps_target, ilsx_td3_stats* stats) {
  if (!t) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_td3_train_step: NULL handle");
  ILSX_TRY(ac_stage(&t->g, obs, act, rew, done, nobs, B, eps_target));
  return td3_step(t, stats);
}

static int ac_sample(AcAgent* g, ilsx_replay* rb, int B) {
  if (!rb || rb->o != g->o || rb->a != g->a) ILSX_FAIL(ILSX_ERR_ARG, "replay buffer dims do not match the agent");
  if (B < 1 || B > g->max_batch) ILSX_FAIL(ILSX_ERR_ARG, "B=%d not in 1..max_batch=%d", B, g->max_batch);
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "replay buffer is empty");
  g->B = B; g->eps_explicit = false;
  return replay_launch_sample(rb, B, nullptr, nullptr, ++rb->sample_ctr, g->s, g->ac, g->r, g->d, g->s2, nullptr);
}

extern "C" int ilsx_td3_train_from_replay(ilsx_td3* t, ilsx_replay* rb, int n_steps, int B, ilsx_td3_stats* stats) {
  if (!t || n_steps < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_td3_train_from_replay: bad argument");
  HIPCHK(hipSetDevice(t->g.ctx->device));
  for (int i = 0; i < n_steps; ++i) {
    ILSX_TRY(ac_sample(&t->g, rb, B));
    ILSX_TRY(td3_step(t, i == 0 ? stats : nullptr));   // statistics of the first batch only (td3.py:124-128)
  }
  return ILSX_OK;
}
// which: 0 qf1, 1 qf2, 2 policy, 3 target_qf1, 4 target_qf2, 5 target_policy
extern "C" int ilsx_td3_get_params(ilsx_td3* t, int which, float* dst_host, size_t n) {
  if (!t) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  return ac_params(&t->g, which, false, dst_host, n);
}
extern "C" int ilsx_td3_set_params(ilsx_td3* t, int which, const float* src_host, size_t n) {
  if (!t) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  return ac_params(&t->g, which, true, const_cast<float*>(src_host), n);
}

extern "C" int ilsx_td3_get_opt(ilsx_td3* t, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!t) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  ILSX_TRY(ac_opt(&t->g, which, false, m_host, v_host, n, meta));
  if (meta) meta->n_train_steps = t->n_steps;
  return ILSX_OK;
}
extern "C" int ilsx_td3_set_opt(ilsx_td3* t, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta) {
  if (!t) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  ilsx_opt_meta m2; if (meta) m2 = *meta;
  ILSX_TRY(ac_opt(&t->g, which, true, const_cast<float*>(m_host), const_cast<float*>(v_host), n, meta ? &m2 : nullptr));
  if (meta) t->n_steps = meta->n_train_steps;
  return ILSX_OK;
}

// ================================================================================================ SAC-V
enum { SV_Q1 = 0, SV_Q2 = 1, SV_V = 2, SV_PI = 3 };
struct ilsx_sacv {
  AcAgent g;
  ilsx_sacv_cfg cfg;
  float *q1 = nullptr, *q2 = nullptr, *tv = nullptr, *v = nullptr, *q1n0 = nullptr, *q2n0 = nullptr, *q1n = nullptr, *q2n = nullptr;
  float *raw = nullptr, *an = nullptr, *logp = nullptr, *epss = nullptr, *ga[2] = {nullptr, nullptr};
  float* ppart = nullptr;   // head partials of pi(s) (column-split path)
};

extern "C" int ilsx_sacv_create(ilsx_ctx* ctx, const ilsx_sacv_cfg* cfg, ilsx_net* pi, ilsx_net* q1, ilsx_net* q2, ilsx_net* vf,
                                ilsx_sacv** out) {
  if (!ctx || !cfg || !pi || !q1 || !q2 || !vf || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sacv_create: NULL argument");
  const ilsx_mlp_cfg &cp = pi->lay.cfg, &c1 = q1->lay.cfg, &c2 = q2->lay.cfg, &cv = vf->lay.cfg;
  if (cp.n_heads != 2) ILSX_FAIL(ILSX_ERR_ARG, "policy must have 2 heads (mean | log_std), policies.py:231-239");
  if (c1.n_heads != 1 || c1.out_dim != 1 || memcmp(&c1, &c2, sizeof c1) != 0) ILSX_FAIL(ILSX_ERR_ARG, "qf1/qf2 must be identical single-output FlattenMlp's");
  if (c1.in_dim != cp.in_dim + cp.out_dim) ILSX_FAIL(ILSX_ERR_ARG, "qf input %d != obs %d + act %d", c1.in_dim, cp.in_dim, cp.out_dim);
  if (cv.n_heads != 1 || cv.out_dim != 1 || cv.in_dim != cp.in_dim) ILSX_FAIL(ILSX_ERR_ARG, "vf must map obs -> 1");
  if (cfg->max_batch < 1 || cfg->max_batch > (1 << 20)) ILSX_FAIL(ILSX_ERR_ARG, "max_batch=%d out of range", cfg->max_batch);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_sacv* s = new ilsx_sacv();
  s->cfg = *cfg;
  ilsx_net* nets[4] = {q1, q2, vf, pi};
  const int opt_of[4] = {0, 0, 1, 2};
  AcAgent* g = &s->g;
  g->lr[0] = cfg->qf_lr; g->lr[1] = cfg->vf_lr; g->lr[2] = cfg->policy_lr; g->beta_1 = cfg->beta_1;
  int rc = ac_init(g, ctx, nets, 4, opt_of, cfg->max_batch, cp.in_dim, cp.out_dim);
  const size_t B = (size_t)cfg->max_batch, a = (size_t)cp.out_dim;
  const size_t CS = (size_t)g->cs;
  float** bufs[] = {&s->q1, &s->q2, &s->tv, &s->v, &s->q1n0, &s->q2n0, &s->q1n, &s->q2n};
  for (float** b : bufs) if (rc == ILSX_OK) rc = ac_alloc(g, b, CS * B);
  if (rc == ILSX_OK) rc = ac_alloc(g, &s->logp, B);
  if (rc == ILSX_OK) rc = ac_alloc(g, &s->raw, B * 2 * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &s->ppart, CS * B * 2 * a);
  float** abufs[] = {&s->an, &s->epss};
  for (float** b : abufs) if (rc == ILSX_OK) rc = ac_alloc(g, b, B * a);
  for (int i = 0; i < 2; ++i) if (rc == ILSX_OK) rc = ac_alloc(g, &s->ga[i], CS * B * a);
  if (rc == ILSX_OK) {  // fixed temperature (sac.py:68): the policy loss head reads it from the device scalars
    DevScalars h;
    memset(&h, 0, sizeof h);
    h.alpha = cfg->alpha; h.alpha_used = cfg->alpha; h.log_alpha = std::log((double)cfg->alpha); h.log_alpha_used = h.log_alpha;
    hipMemcpyAsync(g->dsc, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    rc = ac_tick(g, 0, 0);
  }
  if (rc != ILSX_OK) { delete s; return rc; }
  *out = s;
  return ILSX_OK;
}
extern "C" int ilsx_sacv_destroy(ilsx_sacv* s) {
  if (!s) return ILSX_OK;
  ac_release(&s->g);
  delete s;
  return ILSX_OK;
}

static int sacv_q_pair(ilsx_sacv* s, float* o1, float* o2, bool save_h) {  // Q1, Q2 at (s, a~), a~ already published
  AcAgent* g = &s->g;
  FwdArgs A = ac_fwd_args(g, 2);
  for (int i = 0; i < 2; ++i) {
    ac_fwd(g, A.t[i], i, false, g->s, g->o, s->an, g->a, false);
    if (save_h) for (int l = 0; l < g->L[i].cfg.n_hidden; ++l) A.t[i].hsave[l] = g->h[i][l];
    ac_out(g, A.t[i], i == 0 ? o1 : o2);
  }
  return ac_launch_fwd(g, A, g->L[SV_Q1].KP);
}

static int sacv_step(ilsx_sacv* s, ilsx_sacv_stats* stats) {
  AcAgent* g = &s->g;
  const ilsx_sacv_cfg& c = s->cfg;
  const float* eps = g->eps_explicit ? g->eps : nullptr;
  {  // Q1(s,a) ; Q2(s,a) ; V(s) ; pi(s): the one policy sample of the step (sac.py:123-125)
    FwdArgs A = ac_fwd_args(g, 4);
    ac_fwd(g, A.t[0], SV_Q1, false, g->s, g->o, g->ac, g->a, true); ac_out(g, A.t[0], s->q1);
    ac_fwd(g, A.t[1], SV_Q2, false, g->s, g->o, g->ac, g->a, true); ac_out(g, A.t[1], s->q2);
    ac_fwd(g, A.t[2], SV_V, false, g->s, g->o, nullptr, 0, true); ac_out(g, A.t[2], s->v);
    FwdTask& f = A.t[3];
    ac_fwd(g, f, SV_PI, false, g->s, g->o, nullptr, 0, true);
    f.head = HEAD_TANH_SAMPLE; f.rng_stream = g->rng_stream;
    if (g->cs > 1) f.part = s->ppart;
    else { f.eps = eps; f.action = s->an; f.logp = s->logp; f.out = s->raw; f.eps_save = s->epss; }
    ILSX_TRY(ac_launch_fwd(g, A, std::max(g->L[SV_Q1].KP, g->L[SV_V].KP)));
  }
  {  // pre-update critics at (s, a~) -> v_target ; target V(s') ; prologue: finish pi(s) (sample, log-prob)
    FwdArgs A = ac_fwd_args(g, 3);
    for (int i = 0; i < 2; ++i) {
      ac_fwd(g, A.t[i], i, false, g->s, g->o, s->an, g->a, false);
      ac_out(g, A.t[i], i == 0 ? s->q1n0 : s->q2n0);
    }
    ac_fwd(g, A.t[2], SV_V, true, g->s2, g->o, nullptr, 0, false); ac_out(g, A.t[2], s->tv);
    ac_policy_fin(g, A, HEAD_TANH_SAMPLE, s->ppart, eps, s->raw, s->an, s->logp, s->epss, 0.f, 0.f, 1.f);
    ILSX_TRY(ac_launch_fwd(g, A, std::max(g->L[SV_Q1].KP, g->L[SV_V].KP)));
  }
  {
    BwdArgs A = ac_bwd_args(g, 2, c.discount, c.reward_scale);
    for (int i = 0; i < 2; ++i) {
      BwdTask& b = A.t[i];
      ac_bwd(g, b, i, true);
      b.loss = LOSS_TD_CRITIC; b.coef = 1.0f;   // 0.5*mean(.)^2 (sac.py:104-105)
      b.q = pv(g, i == 0 ? s->q1 : s->q2); b.tq1 = pv(g, s->tv); b.tq2 = pv(g, s->tv);
      b.rew = g->r; b.done = g->d;
    }
    ILSX_TRY(ac_launch_bwd(g, A));
  }
  {
    BwdArgs A = ac_bwd_args(g, 1, 0.f, 0.f);
    BwdTask& b = A.t[0];
    ac_bwd(g, b, SV_V, true);
    b.loss = LOSS_SACV_VALUE; b.coef = c.alpha;
    b.q = pv(g, s->v); b.q1n = pv(g, s->q1n0); b.q2n = pv(g, s->q2n0); b.logp_next = s->logp;
    ILSX_TRY(ac_launch_bwd(g, A));
  }
  ILSX_TRY(ac_dw_adam(g, SV_Q1, 2, false, 0.f));
  ILSX_TRY(ac_dw_adam(g, SV_V, 1, true, c.soft_target_tau));   // target V <- post-Adam V (sac.py:179,242-243)
  ILSX_TRY(sacv_q_pair(s, s->q1n, s->q2n, true));                // updated critics, same sample (sac.py:150-152)
  {
    BwdArgs A = ac_bwd_args(g, 2, 0.f, 0.f);
    for (int i = 0; i < 2; ++i) {
      BwdTask& b = A.t[i];
      ac_bwd(g, b, i, false);
      b.loss = LOSS_SAC_ACTORQ; b.which = i; b.q1n = pv(g, s->q1n); b.q2n = pv(g, s->q2n);
      b.dx = s->ga[i]; b.dx_col0 = g->o; b.dx_cols = g->a;
    }
    ILSX_TRY(ac_launch_bwd(g, A));
  }
  {
    BwdArgs A = ac_bwd_args(g, 1, 0.f, 0.f);
    A.w_mu = c.policy_mean_reg_weight; A.w_std = c.policy_std_reg_weight;
    BwdTask& b = A.t[0];
    ac_bwd(g, b, SV_PI, true);
    b.loss = LOSS_SAC_POLICY; b.raw = s->raw; b.eps = s->epss; b.action = s->an; b.ga1 = s->ga[0]; b.ga2 = s->ga[1];
    ILSX_TRY(ac_launch_bwd(g, A));
  }
  ILSX_TRY(ac_dw_adam(g, SV_PI, 1, false, 0.f));
  ILSX_TRY(ac_tick(g, 7, 1));
  if (stats) {  // sac.py:181-240
    const size_t B = (size_t)g->B, a = (size_t)g->a;
    std::vector<float> q1, q2, tv, v, q1n0, q2n0, q1n, q2n, lp, raw, r, d;
    ILSX_TRY(ac_fetch_pv(g, s->q1, B, q1)); ILSX_TRY(ac_fetch_pv(g, s->q2, B, q2)); ILSX_TRY(ac_fetch_pv(g, s->tv, B, tv));
    ILSX_TRY(ac_fetch_pv(g, s->v, B, v)); ILSX_TRY(ac_fetch_pv(g, s->q1n0, B, q1n0)); ILSX_TRY(ac_fetch_pv(g, s->q2n0, B, q2n0));
    ILSX_TRY(ac_fetch_pv(g, s->q1n, B, q1n)); ILSX_TRY(ac_fetch_pv(g, s->q2n, B, q2n)); ILSX_TRY(ac_fetch(g, s->logp, B, lp));
    ILSX_TRY(ac_fetch(g, s->raw, B * 2 * a, raw)); ILSX_TRY(ac_fetch(g, g->r, B, r)); ILSX_TRY(ac_fetch(g, g->d, B, d));
    HIPCHK(hipStreamSynchronize(g->ctx->stream));
    std::vector<float> y(B), vt(B), mu(B * a), ls(B * a);
    double pl = 0, mu2 = 0, ls2 = 0;
    for (size_t i = 0; i < B; ++i) {
      y[i] = c.reward_scale * r[i] + (1.0f - d[i]) * c.discount * tv[i];
      vt[i] = std::min(q1n0[i], q2n0[i]) - c.alpha * lp[i];
      pl += c.alpha * lp[i] - std::min(q1n[i], q2n[i]);
      for (size_t j = 0; j < a; ++j) {
        const float m = raw[i * 2 * a + j], l = std::min(std::max(raw[i * 2 * a + a + j], -20.0f), 2.0f);
        mu[i * a + j] = m; ls[i * a + j] = l; mu2 += (double)m * m; ls2 += (double)l * l;
      }
    }
    stats->qf1_loss = 0.5f * mean_sq_diff(q1, y);
    stats->qf2_loss = 0.5f * mean_sq_diff(q2, y);
    stats->vf_loss = 0.5f * mean_sq_diff(v, vt);
    stats->policy_loss = (float)(pl / (double)B + c.policy_mean_reg_weight * mu2 / (double)(B * a) + c.policy_std_reg_weight * ls2 / (double)(B * a));
    msmm(q1.data(), B, stats->q1_pred); msmm(q2.data(), B, stats->q2_pred); msmm(v.data(), B, stats->v_pred);
    msmm(lp.data(), B, stats->log_pi); msmm(mu.data(), B * a, stats->policy_mu); msmm(ls.data(), B * a, stats->policy_log_std);
  }
  return ILSX_OK;
}

extern "C" int ilsx_sacv_train_step(ilsx_sacv* s, const float* obs, const float* act, const float* rew, const float* done,
                                    const float* nobs, int B, const float* eps, ilsx_sacv_stats* stats) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sacv_train_step: NULL handle");
  ILSX_TRY(ac_stage(&s->g, obs, act, rew, done, nobs, B, eps));
  return sacv_step(s, stats);
}
extern "C" int ilsx_sacv_train_from_replay(ilsx_sacv* s, ilsx_replay* rb, int n_steps, int B, ilsx_sacv_stats* stats) {
  if (!s || n_steps < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sacv_train_from_replay: bad argument");
  HIPCHK(hipSetDevice(s->g.ctx->device));
  for (int i = 0; i < n_steps; ++i) {
    ILSX_TRY(ac_sample(&s->g, rb, B));
    ILSX_TRY(sacv_step(s, i == 0 ? stats : nullptr));
  }
  return ILSX_OK;
}
// which: 0 qf1, 1 qf2, 2 vf, 3 policy, 6 target_vf
extern "C" int ilsx_sacv_get_params(ilsx_sacv* s, int which, float* dst_host, size_t n) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  return ac_params(&s->g, which, false, dst_host, n);
}
extern "C" int ilsx_sacv_set_params(ilsx_sacv* s, int which, const float* src_host, size_t n) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  return ac_params(&s->g, which, true, const_cast<float*>(src_host), n);
}

extern "C" int ilsx_sacv_get_opt(ilsx_sacv* s, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  if (meta) meta->n_train_steps = 0;
  return ac_opt(&s->g, which, false, m_host, v_host, n, meta);
}
extern "C" int ilsx_sacv_set_opt(ilsx_sacv* s, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  ilsx_opt_meta m2; if (meta) m2 = *meta;
  return ac_opt(&s->g, which, true, const_cast<float*>(m_host), const_cast<float*>(v_host), n, meta ? &m2 : nullptr);
}

// ================================================================================================ behaviour cloning
// rlkit/torch/algorithms/bc/bc.py:14-41,81-106: one Adam(lr, betas=(momentum, 0.999)) over a tanh-Gaussian policy;
// mode MLE = -mean(get_log_prob(obs, acts)), mode MSE = mean_rows(sum_j (sampled action - acts)^2).
struct ilsx_bc {
  AcAgent g;
  ilsx_bc_cfg cfg;
  float *raw = nullptr, *pred = nullptr, *logp = nullptr, *epss = nullptr;
};

extern "C" int ilsx_bc_create(ilsx_ctx* ctx, const ilsx_bc_cfg* cfg, ilsx_net* pi, ilsx_bc** out) {
  if (!ctx || !cfg || !pi || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_bc_create: NULL argument");
  if (pi->lay.cfg.n_heads != 2) ILSX_FAIL(ILSX_ERR_ARG, "policy must have 2 heads (mean | log_std), policies.py:231-239");
  if (cfg->mode != ILSX_BC_MLE && cfg->mode != ILSX_BC_MSE) ILSX_FAIL(ILSX_ERR_ARG, "mode must be ILSX_BC_MLE or ILSX_BC_MSE");
  if (cfg->max_batch < 1 || cfg->max_batch > (1 << 20)) ILSX_FAIL(ILSX_ERR_ARG, "max_batch=%d out of range", cfg->max_batch);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_bc* b = new ilsx_bc();
  b->cfg = *cfg;
  ilsx_net* nets[1] = {pi};
  const int opt_of[1] = {0};
  AcAgent* g = &b->g;
  g->lr[0] = cfg->lr; g->beta_1 = cfg->momentum;
  int rc = ac_init(g, ctx, nets, 1, opt_of, cfg->max_batch, pi->lay.cfg.in_dim, pi->lay.cfg.out_dim);
  const size_t B = (size_t)cfg->max_batch, a = (size_t)pi->lay.cfg.out_dim;
  if (rc == ILSX_OK) rc = ac_alloc(g, &b->raw, B * 2 * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &b->pred, B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &b->epss, B * a);
  if (rc == ILSX_OK) rc = ac_alloc(g, &b->logp, B);
  if (rc == ILSX_OK) rc = ac_tick(g, 0, 0);
  if (rc != ILSX_OK) { delete b; return rc; }
  *out = b;
  return ILSX_OK;
}
extern "C" int ilsx_bc_destroy(ilsx_bc* b) {
  if (!b) return ILSX_OK;
  ac_release(&b->g);
  delete b;
  return ILSX_OK;
}

static int bc_step(ilsx_bc* b, float* stat) {
  AcAgent* g = &b->g;
  const bool mle = b->cfg.mode == ILSX_BC_MLE;
  {
    FwdArgs A = ac_fwd_args(g, 1);
    FwdTask& f = A.t[0];
    ac_fwd(g, f, 0, false, g->s, g->o, nullptr, 0, true);
    f.out = b->raw; f.logp = b->logp; f.rng_stream = g->rng_stream;
    if (mle) { f.head = HEAD_TANH_LOGP_OF_ACT; f.act_in = g->ac; }
    else { f.head = HEAD_TANH_SAMPLE; f.eps = g->eps_explicit ? g->eps : nullptr; f.action = b->pred; f.eps_save = b->epss; }
    ILSX_TRY(launch_fwd(g->ctx, A, g->H, g->act, g->L[0].KP));
  }
  {
    BwdArgs A = ac_bwd_args(g, 1, 0.f, 0.f);
    BwdTask& t = A.t[0];
    ac_bwd(g, t, 0, true);
    t.loss = mle ? LOSS_BC_MLE : LOSS_BC_MSE; t.raw = b->raw; t.act_all = g->ac; t.action = b->pred; t.eps = b->epss;
    ILSX_TRY(launch_bwd_dx(g->ctx, A, g->H, g->act));
  }
  ILSX_TRY(ac_dw_adam(g, 0, 1, false, 0.f));
  ILSX_TRY(ac_tick(g, 1, 1));
  if (stat) {   // "Log-Likelihood" / "MSE" of this batch (bc.py:92-102)
    const size_t B = (size_t)g->B, a = (size_t)g->a;
    if (mle) {
      std::vector<float> lp;
      ILSX_TRY(ac_fetch(g, b->logp, B, lp));
      HIPCHK(hipStreamSynchronize(g->ctx->stream));
      double s = 0;
      for (float v : lp) s += v;
      *stat = (float)(s / (double)B);
    } else {
      std::vector<float> p, t;
      ILSX_TRY(ac_fetch(g, b->pred, B * a, p)); ILSX_TRY(ac_fetch(g, g->ac, B * a, t));
      HIPCHK(hipStreamSynchronize(g->ctx->stream));
      double s = 0;
      for (size_t i = 0; i < B * a; ++i) s += (double)(p[i] - t[i]) * (p[i] - t[i]);
      *stat = (float)(s / (double)B);
    }
  }
  return ILSX_OK;
}

extern "C" int ilsx_bc_train_step(ilsx_bc* b, const float* obs, const float* act, int B, const float* eps, float* stat) {
  if (!b || !obs || !act) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_bc_train_step: NULL argument");
  AcAgent* g = &b->g;
  if (B < 1 || B > g->max_batch) ILSX_FAIL(ILSX_ERR_ARG, "B=%d not in 1..max_batch=%d", B, g->max_batch);
  HIPCHK(hipSetDevice(g->ctx->device));
  hipStream_t st = g->ctx->stream;
  HIPCHK(hipMemcpyAsync(g->s, obs, (size_t)B * g->o * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(g->ac, act, (size_t)B * g->a * 4, hipMemcpyDeviceToDevice, st));
  g->eps_explicit = eps != nullptr;
  if (eps) HIPCHK(hipMemcpyAsync(g->eps, eps, (size_t)B * g->a * 4, hipMemcpyDeviceToDevice, st));
  g->B = B;
  return bc_step(b, stat);
}
// BC._do_training (bc.py:77-79): n_updates x (expert batch with keys observations / actions -> update)
extern "C" int ilsx_bc_train_from_replay(ilsx_bc* b, ilsx_replay* expert_rb, int n_updates, int B, float* stat) {
  if (!b || n_updates < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_bc_train_from_replay: bad argument");
  HIPCHK(hipSetDevice(b->g.ctx->device));
  for (int i = 0; i < n_updates; ++i) {
    ILSX_TRY(ac_sample(&b->g, expert_rb, B));
    ILSX_TRY(bc_step(b, i == 0 ? stat : nullptr));
  }
  return ILSX_OK;
}

extern "C" int ilsx_bc_get_opt(ilsx_bc* b, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!b) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  if (meta) meta->n_train_steps = 0;
  return ac_opt(&b->g, 0, false, m_host, v_host, n, meta);
}
extern "C" int ilsx_bc_set_opt(ilsx_bc* b, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta) {
  if (!b) ILSX_FAIL(ILSX_ERR_ARG, "NULL handle");
  ilsx_opt_meta m2; if (meta) m2 = *meta;
  return ac_opt(&b->g, 0, true, const_cast<float*>(m_host), const_cast<float*>(v_host), n, meta ? &m2 : nullptr);
}
