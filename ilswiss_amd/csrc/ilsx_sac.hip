// ilsx_sac.hip — SoftActorCritic (twin Q, auto alpha): rlkit/torch/algorithms/sac/sac_alpha.py:21-181.
//
// One agent = one parameter arena in HBM  P = [Q1 | Q2 | pi | TQ1 | TQ2], a gradient arena
// G = [Q1 | Q2 | pi | alpha-slot], Adam moments M,V = [Q1 | Q2 | pi], device scalars (log_alpha in
// float64, step counters) and a per-batch activation workspace.  A train step is 14 launches
// (sequence below), captured once per (replay, B) into a hipGraph for train_from_replay.
//
//   critic_backward : fwd{pi(s'), Q1(s,a), Q2(s,a)} ; fwd{TQ1(s',a'), TQ2(s',a')} ;
//                     bwd_dx{Q1,Q2 with the TD-target loss head} ; bwd_dw{Q1,Q2}
//   critic_update   : adam+polyak over [Q1|Q2] -> also writes [TQ1|TQ2]   (targets use post-Adam Q,
//                     and nothing reads them again this step: bitwise the same as sac_alpha.py:181)
//   actor_backward  : fwd{pi(s)} ; fwd{Q1(s,a~), Q2(s,a~)} (updated critics) ;
//                     bwd_dx{Q1,Q2 -> d(-min Q)/da~} ; bwd_dx{pi with the tanh-Gaussian loss head} ;
//                     bwd_dw{pi} ; stats (+ alpha gradient into the arena slot)
//   actor_update    : adam over pi ; finish (alpha Adam in float64, counters)
#include <cmath>
#include <cstdlib>

#include <algorithm>
#include "host_common.h"

enum { W_PI = 0, W_Q1 = 1, W_Q2 = 2, W_TQ1 = 3, W_TQ2 = 4 };

struct SacWs {  // device workspace for B <= max_batch rows
  float *s, *a, *r, *d, *s2, *eps1, *eps2;     // batch staging
  float *a2, *logp2, *q1, *q2, *tq1, *tq2;       // critic phase
  float *xq[2], *hq[2][ILSX_MAX_HID], *dq[2][ILSX_MAX_HID], *dhq[2];
  float *raw, *an, *logp, *epss, *q1n, *q2n, *ga[2];
  float *xp, *hp[ILSX_MAX_HID], *dp[ILSX_MAX_HID], *dhp;
  float* ppart;  // policy head partials [cs][max_batch][2a] (column-split path): pi(s')
  float* ppart2; //   and pi(s), whose trunk runs in the step's first launch
};

struct StatsArgs {
  PartVal q1, q2, tq1, tq2, q1n, q2n;
  const float *logp2, *r, *d, *logp, *raw;
  int B, a;
  float gamma, reward_scale, w_mu, w_std, target_entropy, inv_B;
  DevScalars* scal;
  float* alpha_grad_slot;  // G arena slot: -(sum(logp + target_entropy)) * inv_B
  int slot_given;          // split run on the phase kernels: the slot already holds the all-reduced gradient (PhaseCArgs::aslot) — leave it
};

// A grouped step (ilsx_sac_group) is assembled by running every agent's ordinary step in "collect" mode: the launch sites
// below hand their descriptors to this list instead of launching.
struct SacTailItem { StatsArgs S; int train_alpha; float lr, b1, b2, eps, qf_lr, policy_lr; };
struct SacLaunch {
  int kind;   // 0 forward, 1 backward-to-activations, 2 weight gradients (+ Adam), 3 tail
  FwdArgs f; int KP; BwdArgs b; DwArgs d; AdamFuse F; SacTailItem t;
};
struct SacCollector { std::vector<SacLaunch> L; };

struct ilsx_sac {
  ilsx_ctx* ctx = nullptr;
  SacCollector* col = nullptr;
  ilsx_sac_cfg cfg;
  ilsx_net *pi = nullptr, *q1 = nullptr, *q2 = nullptr;
  NetLayout Lq, Lp;
  int o = 0, a = 0;
  size_t nq = 0, np = 0;     // internal floats per critic / policy
  void* slab = nullptr;      // the agent's one device allocation: scal | P | G | M | V | workspace
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  DevScalars* scal = nullptr;
  SacWs ws;
  DwArgs jobs_q, jobs_p;     // dW tables (critics / policy), passed by value in the kernel arguments
  ilsx_replay* gather_rb = nullptr;  // train_from_replay: the first forward launch draws its rows from this ring
  bool fuse_now = false;     // this step applies Adam(+Polyak) inside the dW epilogue (not in split-run phases)
  int cs = 1;                // column-split factor of the 2-hidden-layer fast path (1 = generic kernels)
  float* h0scr = nullptr;    // non-null: two-phase forward for wide inputs (layer 0 in its own launch); [4 tasks][max_batch][H] scratch
  int B = 0;                 // rows of the batch currently staged
  bool eps_explicit = false;
  float target_entropy = 0.f;
  uint32_t rng_stream = 0;
  // hipGraph cache for train_from_replay
  hipGraphExec_t graph = nullptr;
  ilsx_replay* graph_rb = nullptr;
  int graph_B = 0;
  bool graph_defer = false, graph_phase = false;
  // split runs: the step as three captured segments with the two all-reduces between them (sac_split_segments)
  hipGraphExec_t seg_graph[3] = {nullptr, nullptr, nullptr};
  ilsx_replay* seg_rb = nullptr;
  int seg_B = 0;
  // deferred tail (TailLite, kernels.h): active inside ilsx_sac_train_from_replay on the column-split path
  bool defer_tail = false;
  TailLite* tail_dev = nullptr;
  int tail_B = 0;
  ilsx_sac_stats stats_cache; bool stats_cached = false;   // last statistics read from the device (sac_read_stats), for ilsx_sac_last_stats
  bool tail_split = false;           // the uploaded TailLite record was built for a split run (slot_given)
  // merged phase kernels (kernels.h k_sac_phase_a / _c): F1 F2 B1 and F3 B2 B3 as one launch each inside train_from_replay
  unsigned* phase_flags = nullptr;   // PHASE_NFLAGS arrival counters, one 128-byte line each (zeroed by the dW launches)
  int* phase_err = nullptr;          // set by a workgroup whose wait timed out
  PhaseConst pct;                    // the phase kernels' descriptor blocks in constant memory (host_common.h)
  bool phase_args_only = false;      // dry pass: build the two blocks, upload them, launch nothing (sac_phase_const_prime)
  bool no_ct = false;                // this launch sequence must not depend on the slots' contents (captures that are replayed without a re-prime)
  bool phase_now = false;            // this step runs on the phase kernels
  bool phase_broken = false;         // a timeout was seen once: stay on the 8-launch path
  bool phase_last = false;           // the last window of steps ran on the phase kernels
  bool debug_break = false;          // ilsx_sac_debug_break_phase
  int phase_fallbacks = 0;           // windows rolled back and re-run (ilsx_sac_phase_state)
  float* vote = nullptr;             // split run: the ranks' agreement on a roll-back (sac_phase_check), one device float
  bool snap_valid = false;           // the checkpoint belongs to the window now running (set by sac_snapshot_take, cleared when the window ends)
  void* snap = nullptr;              // checkpoint of [scal | P | G | M | V] taken at the start of every window that may run on the phase kernels
  size_t snap_bytes = 0;
  float* base(int which) const {
    switch (which) {
      case W_Q1: return P;
      case W_Q2: return P + nq;
      case W_PI: return P + 2 * nq;
      case W_TQ1: return P + 2 * nq + np;
      default: return P + 3 * nq + np;
    }
  }
  PartVal pv(const float* p) const { return PartVal{p, cs, cfg.max_batch}; }
  float* gbase(int which) const { return which == W_Q1 ? G : which == W_Q2 ? G + nq : G + 2 * nq; }
  size_t trainable_off(int which) const { return which == W_Q1 ? 0 : which == W_Q2 ? nq : 2 * nq; }
};

// ------------------------------------------------------------------------------------------------

// single workgroup: losses of sac_alpha.py:122-123,148-153,161-162 + the alpha gradient
// the alpha gradient alone (every step); the full statistics only when the host asked for them
__device__ __forceinline__ void sac_alpha_grad_dev(const StatsArgs& S) {
  __shared__ float sh2[4];
  float lpe = 0.f;
  for (int r = threadIdx.x; r < S.B; r += 256) lpe += S.logp[r] + S.target_entropy;
  lpe = block256_sum(lpe, sh2);
  if (threadIdx.x == 0) {
    S.scal->alpha_used = S.scal->alpha;
    S.scal->log_alpha_used = S.scal->log_alpha;
    if (!S.slot_given) {
      S.alpha_grad_slot[0] = -lpe * S.inv_B;
      S.alpha_grad_slot[1] = 0.f; S.alpha_grad_slot[2] = 0.f; S.alpha_grad_slot[3] = 0.f;
    }
  }
}

__device__ __forceinline__ void sac_stats_dev(const StatsArgs& S) {
  __shared__ float sh[4];
  const float alpha = S.scal->alpha;
  float l1 = 0, l2 = 0, pl = 0, lp = 0, mu2 = 0, ls2 = 0, mus = 0, lss = 0, q1s = 0, q2s = 0, lpe = 0;
  float mx[5] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY}, mn[5] = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
  for (int r = threadIdx.x; r < S.B; r += 256) {
    const float y = S.reward_scale * S.r[r] +
                    (1.0f - S.d[r]) * S.gamma * (fminf(S.tq1.get(r), S.tq2.get(r)) - alpha * S.logp2[r]);
    const float q1v = S.q1.get(r), q2v = S.q2.get(r);
    const float e1 = q1v - y, e2 = q2v - y;
    l1 += e1 * e1; l2 += e2 * e2; q1s += q1v; q2s += q2v;
    mx[0] = fmaxf(mx[0], q1v); mn[0] = fminf(mn[0], q1v); mx[1] = fmaxf(mx[1], q2v); mn[1] = fminf(mn[1], q2v);
    mx[2] = fmaxf(mx[2], S.logp[r]); mn[2] = fminf(mn[2], S.logp[r]);
    pl += alpha * S.logp[r] - fminf(S.q1n.get(r), S.q2n.get(r));
    lp += S.logp[r];
    lpe += S.logp[r] + S.target_entropy;
    for (int j = 0; j < S.a; ++j) {
      const float mu = S.raw[(size_t)r * 2 * S.a + j];
      const float ls = fminf(fmaxf(S.raw[(size_t)r * 2 * S.a + S.a + j], LOG_SIG_MIN), LOG_SIG_MAX);
      mu2 += mu * mu; ls2 += ls * ls; mus += mu; lss += ls;
      mx[3] = fmaxf(mx[3], mu); mn[3] = fminf(mn[3], mu); mx[4] = fmaxf(mx[4], ls); mn[4] = fminf(mn[4], ls);
    }
  }
  l1 = block256_sum(l1, sh); l2 = block256_sum(l2, sh); pl = block256_sum(pl, sh); lp = block256_sum(lp, sh);
  mu2 = block256_sum(mu2, sh); ls2 = block256_sum(ls2, sh); mus = block256_sum(mus, sh); lss = block256_sum(lss, sh);
  q1s = block256_sum(q1s, sh); q2s = block256_sum(q2s, sh); lpe = block256_sum(lpe, sh);
#pragma unroll
  for (int i = 0; i < 5; ++i) {   // block extrema: wave butterflies, then the 4 wave results through LDS
    float a = mx[i], b = mn[i];
    for (int o = 32; o >= 1; o >>= 1) { a = fmaxf(a, __shfl_xor(a, o, 64)); b = fminf(b, __shfl_xor(b, o, 64)); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    mx[i] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b;
    __syncthreads();
    mn[i] = fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
  }
  // np.std (create_stats_ordered_dict, core/eval_util.py:91-142) is the two-pass population standard deviation: centre on the mean, then
  // average the squares.  (The first form, sqrt(E[x^2] - mean^2) from one pass, loses the digits a tight distribution around a
  // large mean needs; this kernel only runs on the one batch per epoch whose statistics the host reads.)
  const float iB = 1.0f / (float)S.B, iBa = iB / (float)S.a;
  const float means[5] = {q1s * iB, q2s * iB, lp * iB, mus * iBa, lss * iBa};
  float cs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < S.B; r += 256) {
    const float d0 = S.q1.get(r) - means[0], d1 = S.q2.get(r) - means[1], d2 = S.logp[r] - means[2];
    cs[0] += d0 * d0; cs[1] += d1 * d1; cs[2] += d2 * d2;
    for (int j = 0; j < S.a; ++j) {
      const float dm = S.raw[(size_t)r * 2 * S.a + j] - means[3];
      const float dl = fminf(fmaxf(S.raw[(size_t)r * 2 * S.a + S.a + j], LOG_SIG_MIN), LOG_SIG_MAX) - means[4];
      cs[3] += dm * dm; cs[4] += dl * dl;
    }
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) cs[i] = block256_sum(cs[i], sh);
  if (threadIdx.x == 0) {
    DevScalars* sc = S.scal;
    sc->qf1_loss = 0.5f * l1 * iB;
    sc->qf2_loss = 0.5f * l2 * iB;
    sc->policy_loss = pl * iB + S.w_mu * mu2 * iBa + S.w_std * ls2 * iBa;
    sc->alpha_loss = -(float)sc->log_alpha * (lpe * iB);
    sc->q1_mean = means[0]; sc->q2_mean = means[1]; sc->log_pi_mean = means[2];
    sc->mu_mean = means[3]; sc->log_std_mean = means[4];
    for (int i = 0; i < 5; ++i) {
      sc->ext_std[i] = sqrtf(cs[i] * (i < 3 ? iB : iBa));
      sc->ext_max[i] = mx[i]; sc->ext_min[i] = mn[i];
    }
    sc->alpha_used = alpha;
    sc->log_alpha_used = sc->log_alpha;
    if (!S.slot_given) {
      S.alpha_grad_slot[0] = -lpe * S.inv_B;  // d(alpha_loss)/d(log_alpha), summed over ranks by the all-reduce
      S.alpha_grad_slot[1] = 0.f; S.alpha_grad_slot[2] = 0.f; S.alpha_grad_slot[3] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_sac_stats(const StatsArgs S) { sac_stats_dev(S); }

__global__ void k_sac_finish(DevScalars* sc, const float* alpha_grad_slot, int train_alpha, float lr, float b1,
                             float b2, float eps, float qf_lr, float policy_lr) {
  sac_finish_dev(sc, alpha_grad_slot, train_alpha, lr, b1, b2, eps, qf_lr, policy_lr, 0);
}
// stats + alpha Adam + counters in ONE launch (the un-split step: nothing has to be all-reduced in between)
// deferred = 1: this launch flushes a deferred tail (TailLite, kernels.h) — gather_step was advanced by the step itself
__global__ __launch_bounds__(256) void k_sac_tail(const StatsArgs S, int train_alpha, float lr, float b1, float b2,
                                                  float eps, float qf_lr, float policy_lr, int deferred) {
  if (S.scal->want_stats) sac_stats_dev(S);   // workgroup-uniform
  else sac_alpha_grad_dev(S);
  if (threadIdx.x == 0) S.scal->want_stats = 0;
  if (threadIdx.x == 0) {
    sac_finish_dev(S.scal, S.alpha_grad_slot, train_alpha, lr, b1, b2, eps, qf_lr, policy_lr, deferred);
  }
}

// grouped step: one workgroup per agent
__global__ __launch_bounds__(256) void k_sac_tail_group(const SacTailItem* items, int deferred) {
  const SacTailItem& T = items[blockIdx.x];
  if (T.S.scal->want_stats) sac_stats_dev(T.S);   // workgroup-uniform
  else sac_alpha_grad_dev(T.S);
  if (threadIdx.x == 0) {
    T.S.scal->want_stats = 0;
    sac_finish_dev(T.S.scal, T.S.alpha_grad_slot, T.train_alpha, T.lr, T.b1, T.b2, T.eps, T.qf_lr, T.policy_lr, deferred);
  }
}

// device-side (re)computation of the Adam scalars so every path uses the same pow()
__global__ void k_sac_refresh_adam(DevScalars* sc, float qf_lr, float policy_lr, float b1, float b2) {
  adam_scalars(qf_lr, b1, b2, sc->t_q + 1, &sc->adam_q_step, &sc->adam_q_bc2s);
  adam_scalars(policy_lr, b1, b2, sc->t_pi + 1, &sc->adam_pi_step, &sc->adam_pi_bc2s);
}
static int sac_refresh_adam(struct ilsx_sac* s);

// ------------------------------------------------------------------------------------------------
static void sac_plan_ws(ilsx_sac* s, Slab& L) {
  const size_t B = (size_t)s->cfg.max_batch;
  const int o = s->o, a = s->a, H = s->Lq.cfg.hidden;
  SacWs& w = s->ws;
  auto A = [&](float** p, size_t n) { L.add(p, n); };
  A(&w.s, B * o); A(&w.a, B * a); A(&w.r, B); A(&w.d, B);
  A(&w.s2, B * o); A(&w.eps1, B * a); A(&w.eps2, B * a);
  const size_t CS = (size_t)s->cs;
  A(&w.a2, B * a); A(&w.logp2, B); A(&w.q1, CS * B); A(&w.q2, CS * B);
  A(&w.tq1, CS * B); A(&w.tq2, CS * B); A(&w.ppart, CS * B * 2 * a);
  A(&w.ppart2, CS * B * 2 * a);
  for (int i = 0; i < 2; ++i) {
    A(&w.xq[i], B * s->Lq.KP);
    for (int l = 0; l < s->Lq.cfg.n_hidden; ++l) { A(&w.hq[i][l], B * H); A(&w.dq[i][l], B * H); }
    A(&w.dhq[i], B * 4);
    A(&w.ga[i], CS * B * a);
  }
  A(&w.raw, B * 2 * a); A(&w.an, B * a); A(&w.logp, B); A(&w.epss, B * a);
  A(&w.q1n, CS * B); A(&w.q2n, CS * B);
  A(&w.xp, B * s->Lp.KP);
  for (int l = 0; l < s->Lp.cfg.n_hidden; ++l) { A(&w.hp[l], B * H); A(&w.dp[l], B * H); }
  A(&w.dhp, B * 2 * a);
  // wide inputs (Humanoid: KP = 396 against 16 for the planar tasks): every column slice recomputing layer 0 costs 6x its own layer-1
  // work, so the forward becomes two launches (kernels.h, k_mlp2_fwd_split PH 1 / 2).  ILSX_L0_SPLIT_KP moves the threshold.
  const char* e = getenv("ILSX_L0_SPLIT_KP");
  const int thr = e ? atoi(e) : 65;   // KP is a multiple of 16: the planar tasks (KP 16 / 32) keep the one-launch form, Ant (128) and Humanoid (396) do not
  if (s->cs == 4 && H == 256 && std::max(s->Lq.KP, s->Lp.KP) >= thr) A(&s->h0scr, 4 * B * H);
}

extern "C" int ilsx_sac_create(ilsx_ctx* ctx, const ilsx_sac_cfg* cfg, ilsx_net* pi, ilsx_net* q1, ilsx_net* q2,
                               ilsx_sac** out) {
  if (!ctx || !cfg || !pi || !q1 || !q2 || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_create: NULL argument");
  if (pi->ctx != ctx || q1->ctx != ctx || q2->ctx != ctx) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_create: nets belong to another ctx");
  const ilsx_mlp_cfg &cp = pi->lay.cfg, &c1 = q1->lay.cfg, &c2 = q2->lay.cfg;
  if (cp.n_heads != 2) ILSX_FAIL(ILSX_ERR_ARG, "policy must have 2 heads (mean | log_std), policies.py:231-239");
  if (c1.n_heads != 1 || c1.out_dim != 1 || memcmp(&c1, &c2, sizeof c1) != 0)
    ILSX_FAIL(ILSX_ERR_ARG, "qf1/qf2 must be identical single-output FlattenMlp's");
  if (c1.in_dim != cp.in_dim + cp.out_dim) ILSX_FAIL(ILSX_ERR_ARG, "qf input %d != obs %d + act %d", c1.in_dim, cp.in_dim, cp.out_dim);
  if (c1.hidden != cp.hidden || c1.n_hidden != cp.n_hidden || c1.act != cp.act)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "policy and critics must share hidden width/depth/activation");
  if (!pi->owns || !q1->owns || !q2->owns) ILSX_FAIL(ILSX_ERR_STATE, "a network already belongs to an agent");
  if (cfg->max_batch < 1 || cfg->max_batch > (1 << 20)) ILSX_FAIL(ILSX_ERR_ARG, "max_batch=%d out of range", cfg->max_batch);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_sac* s = new ilsx_sac();
  s->ctx = ctx; s->cfg = *cfg; s->pi = pi; s->q1 = q1; s->q2 = q2;
  if (s->cfg.grad_world < 1) s->cfg.grad_world = 1;
  s->Lq = q1->lay; s->Lp = pi->lay;
  s->o = cp.in_dim; s->a = cp.out_dim;
  s->nq = s->Lq.n_int; s->np = s->Lp.n_int;
  s->cs = getenv("ILSX_NO_SPLIT") ? 1 : mlp2_split_factor(cp.n_hidden, cp.hidden);
  s->target_entropy = cfg->has_target_entropy ? cfg->target_entropy : -(float)s->a / 2.0f;  // sac_alpha.py:56-58
  s->rng_stream = ctx->next_rng_stream;
  ctx->next_rng_stream += 2;
  const size_t nP = 4 * s->nq + s->np, nT = 2 * s->nq + s->np;
  Slab L;   // parameters, optimiser state, scalars and the whole step workspace: one allocation (see Slab)
  L.add(&s->scal, 1);
  L.add(&s->P, nP); L.add(&s->G, nT + 4); L.add(&s->M, nT); L.add(&s->V, nT);
  L.add(&s->phase_flags, (size_t)PHASE_FLAG_WORDS); L.add(&s->phase_err, 32);
  sac_plan_ws(s, L);
  int rc = L.commit(ctx, &s->slab);
  if (rc != ILSX_OK) { delete s; return rc; }
  // adopt the networks' storage: copy into the arena, targets = copies (sac_alpha.py:60-61)
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(s->base(W_Q1), q1->base, s->nq * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(s->base(W_Q2), q2->base, s->nq * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(s->base(W_PI), pi->base, s->np * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(s->base(W_TQ1), q1->base, s->nq * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(s->base(W_TQ2), q2->base, s->nq * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  ilsx_net* nets[3] = {pi, q1, q2};
  const int wh[3] = {W_PI, W_Q1, W_Q2};
  for (int i = 0; i < 3; ++i) {
    ILSX_TRY(ctx_free(ctx, nets[i]->base));
    nets[i]->base = s->base(wh[i]);
    nets[i]->owns = false;
  }
  DevScalars h;
  memset(&h, 0, sizeof h);
  h.log_alpha = std::log((double)cfg->alpha);  // sac_alpha.py:51-53 (np.log -> float64)
  h.alpha = (float)std::exp(h.log_alpha);
  h.alpha_used = h.alpha;
  h.log_alpha_used = h.log_alpha;
  HIPCHK(hipMemcpyAsync(s->scal, &h, sizeof h, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  ILSX_TRY(sac_refresh_adam(s));
  // weight-gradient tables
  memset(&s->jobs_q, 0, sizeof s->jobs_q);
  memset(&s->jobs_p, 0, sizeof s->jobs_p);
  for (int i = 0; i < 2; ++i)
    ILSX_TRY(build_dw_jobs(s->Lq, s->gbase(i == 0 ? W_Q1 : W_Q2), s->ws.xq[i], s->ws.hq[i], s->ws.dq[i], s->ws.dhq[i], &s->jobs_q));
  ILSX_TRY(build_dw_jobs(s->Lp, s->gbase(W_PI), s->ws.xp, s->ws.hp, s->ws.dp, s->ws.dhp, &s->jobs_p));
  s->jobs_q.zero_flags = s->phase_flags;   // each dW launch re-arms the arrival counters of the phase launch that follows it
  s->jobs_p.zero_flags = s->phase_flags;
  s->jobs_q.zero_err = s->phase_err; s->jobs_p.zero_err = s->phase_err;
  *out = s;
  return ILSX_OK;
}

extern "C" int ilsx_sac_destroy(ilsx_sac* s) {
  if (!s) return ILSX_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  if (s->graph) hipGraphExecDestroy(s->graph);
  for (hipGraphExec_t& g : s->seg_graph) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  if (s->pct.slot >= 0) {
    phase_const_free(s->pct.device, s->pct.slot);
    auto& v = s->ctx->phase_slots;
    v.erase(std::remove(v.begin(), v.end(), s->pct.slot), v.end());
    s->pct.slot = -1;
  }
  if (s->tail_dev) ctx_free(s->ctx, s->tail_dev);
  if (s->snap) ctx_free(s->ctx, s->snap);
  if (s->vote) ctx_free(s->ctx, s->vote);
  // give the networks private storage back so their handles stay usable
  ilsx_net* nets[3] = {s->pi, s->q1, s->q2};
  const int wh[3] = {W_PI, W_Q1, W_Q2};
  for (int i = 0; i < 3; ++i) {
    float* nb = nullptr;
    if (ctx_alloc(s->ctx, nets[i]->lay.n_int * 4, (void**)&nb, false) == ILSX_OK) {
      hipMemcpyAsync(nb, s->base(wh[i]), nets[i]->lay.n_int * 4, hipMemcpyDeviceToDevice, s->ctx->stream);
      hipStreamSynchronize(s->ctx->stream);
      nets[i]->base = nb;
      nets[i]->owns = true;
    }
  }
  // workspace / arenas are released with the ctx (ctx owns every allocation); free the big ones now
  ctx_free(s->ctx, s->slab);
  delete s;
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------------
static float sac_inv_B(const ilsx_sac* s) { return 1.0f / ((float)s->B * (float)s->cfg.grad_world); }

// forward of the policy on `obs` (+ its tanh-Gaussian epilogue) as task `slot` of a forward launch; when the
// column-split path is on, the epilogue runs in k_policy_finish right after the launch (sac_policy_finish).
static void sac_policy_task(ilsx_sac* s, FwdTask& t, const float* obs, const float* eps, uint32_t stream, bool save,
                            float* action, float* logp, float* part) {
  const SacWs& w = s->ws;
  t.net = net_view(s->Lp, s->base(W_PI));
  t.x0 = obs; t.d0 = s->o; t.s0 = s->o;
  if (save) {
    t.xsave = w.xp;
    for (int l = 0; l < s->Lp.cfg.n_hidden; ++l) t.hsave[l] = w.hp[l];
  }
  t.head = HEAD_TANH_SAMPLE; t.rng_stream = stream;
  if (s->cs > 1) {
    t.part = part;
  } else {
    t.eps = eps; t.action = action; t.logp = logp;
    if (save) { t.out = w.raw; t.eps_save = w.epss; }
  }
}
// the policy's tanh-Gaussian epilogue runs in the prologue of the launch that consumes its actions (column-split path)
static void sac_policy_fin(ilsx_sac* s, FwdArgs& A, const float* eps, uint32_t stream, bool save, float* action, float* logp,
                           const float* part) {
  if (s->cs == 1) return;
  const SacWs& w = s->ws;
  PolicyFinishArgs& P = A.fin;
  memset(&P, 0, sizeof P);
  P.part = part; P.cs = s->cs; P.part_stride = s->cfg.max_batch; P.rows = s->B; P.a = s->a;
  P.head = HEAD_TANH_SAMPLE; P.rng_stream = stream; P.seed = s->ctx->seed; P.scal = s->scal;
  P.eps = eps; P.action = action; P.logp = logp;
  if (save) { P.raw = w.raw; P.eps_save = w.epss; }
  A.fin_on = 1;
}
static void sac_q_task(ilsx_sac* s, FwdTask& q, int which, const float* obs, const float* act, float* out, bool save_x,
                       bool save_h, int i) {
  const SacWs& w = s->ws;
  q.net = net_view(s->Lq, s->base(which));
  q.x0 = obs; q.d0 = s->o; q.s0 = s->o; q.x1 = act; q.d1 = s->a; q.s1 = s->a;
  if (save_x) q.xsave = w.xq[i];
  if (save_h) for (int l = 0; l < s->Lq.cfg.n_hidden; ++l) q.hsave[l] = w.hq[i][l];
  q.head = HEAD_RAW;
  if (s->cs > 1) q.part = out; else q.out = out;
}

static int sac_fwd(ilsx_sac* s, const FwdArgs& A0, int H, int act, int KP, int cs) {
  FwdArgs A = A0;
  if (s->h0scr) {   // wide inputs: two-phase forward (k_mlp2_fwd_split PH 1 / 2); tasks that do not save layer 0 get scratch for it
    A.l0_split = 1;
    for (int t = 0; t < A.ntasks; ++t)
      if (!A.t[t].hsave[0]) A.t[t].hsave[0] = s->h0scr + (size_t)t * s->cfg.max_batch * H;
  }
  if (!s->col) return launch_fwd(s->ctx, A, H, act, KP, cs);
  SacLaunch l; memset(&l, 0, sizeof l); l.kind = 0; l.f = A; l.KP = KP;
  s->col->L.push_back(l);
  return ILSX_OK;
}
static int sac_bwd(ilsx_sac* s, const BwdArgs& A, int H, int act, int cs) {
  if (!s->col) return launch_bwd_dx(s->ctx, A, H, act, cs);
  SacLaunch l; memset(&l, 0, sizeof l); l.kind = 1; l.b = A;
  s->col->L.push_back(l);
  return ILSX_OK;
}
static int sac_dw(ilsx_sac* s, const DwArgs& table, int rows, const AdamFuse* F) {
  if (!s->col) return launch_bwd_dw(s->ctx, table, rows, F);
  SacLaunch l; memset(&l, 0, sizeof l); l.kind = 2; l.d = table; l.F = *F;
  s->col->L.push_back(l);
  return ILSX_OK;
}

// What the phase launches' descriptor blocks are built from besides the agent's fixed allocations: the ring they draw from, the batch, and the
// step-form flags.  Two builds under the same key give the same blocks (PhaseConst::key_a / key_c); never 0.
static unsigned long long sac_phase_key(const ilsx_sac* s) {
  unsigned long long k = 1469598103934665603ull;
  auto mix = [&k](unsigned long long v) { k = (k ^ v) * 1099511628211ull; };
  const ilsx_replay* rb = s->gather_rb;
  mix((unsigned long long)(uintptr_t)rb); mix(rb ? (unsigned long long)(uintptr_t)rb->data : 0); mix(rb ? (unsigned long long)(uintptr_t)rb->dstate : 0);
  mix(rb ? (unsigned long long)rb->rec : 0); mix(rb ? (unsigned long long)rb->seed : 0); mix(rb ? (unsigned long long)rb->rng_stream : 0);
  mix((unsigned long long)s->B); mix(s->defer_tail ? 1 : 0); mix(s->eps_explicit ? 1 : 0); mix(s->fuse_now ? 1 : 0); mix((unsigned long long)s->cs);
  mix((unsigned long long)(uintptr_t)s->tail_dev); mix((unsigned long long)(uintptr_t)s->slab);
  return k ? k : 1;
}

static int sac_critic_backward(ilsx_sac* s) {
  const SacWs& w = s->ws;
  const int B = s->B, H = s->Lq.cfg.hidden, act = s->Lq.cfg.act, cs = s->cs;
  const float* eps1 = s->eps_explicit ? w.eps1 : nullptr;
  const float* eps2 = s->eps_explicit ? w.eps2 : nullptr;
  PhaseAArgs PA;
  if (s->phase_now) memset(&PA, 0, sizeof PA);
  {  // fwd: pi(s') with eps_next ; Q1(s,a) ; Q2(s,a) ; pi(s) with eps_cur (its weights do not change before the actor
     // phase reads it, so its trunk rides along here and one dependent launch disappears from the step)
    FwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 4; A.seed = s->ctx->seed; A.scal = s->scal; A.part_stride = s->cfg.max_batch;
    sac_policy_task(s, A.t[0], w.s2, eps1, s->rng_stream, false, w.a2, w.logp2, w.ppart);
    sac_policy_task(s, A.t[3], w.s, eps2, s->rng_stream + 1, true, w.an, w.logp, w.ppart2);
    sac_q_task(s, A.t[1], W_Q1, w.s, w.a, w.q1, true, true, 0);
    sac_q_task(s, A.t[2], W_Q2, w.s, w.a, w.q2, true, true, 1);
    if (s->gather_rb && cs > 1) {  // fused sample+index: rows are drawn from the ring inside this launch
      ilsx_replay* rb = s->gather_rb;
      GatherSpec& G = A.gather;
      G.records = rb->data; G.st = rb->dstate; G.rec = rb->rec; G.seed = rb->seed; G.stream = rb->rng_stream;
      G.o = s->o; G.adim = s->a; G.on = 1;
      G.s = w.s; G.a = w.a; G.r = w.r; G.d = w.d; G.s2 = w.s2;
      A.t[0].g0_off = s->o + s->a + 2; A.t[0].publish = 2;               // pi reads next_obs
      A.t[1].g0_off = 0; A.t[1].g1_off = s->o; A.t[1].publish = 1;      // Q1 reads (obs, act) and publishes s,a,r,d
      A.t[2].g0_off = 0; A.t[2].g1_off = s->o; A.t[2].publish = 0;
      A.t[3].g0_off = 0; A.t[3].publish = 0;                              // pi reads obs
    }
    if (s->defer_tail) { A.tail = s->tail_dev; A.tail_mode = 1; A.tail_n = 1; }
    if (s->phase_now) PA.f1 = A;
    else ILSX_TRY(sac_fwd(s, A, H, act, std::max(s->Lq.KP, s->Lp.KP), cs));
  }
  {  // fwd: TQ1(s',a'), TQ2(s',a')
    FwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 2; A.seed = s->ctx->seed; A.scal = s->scal; A.part_stride = s->cfg.max_batch;
    sac_q_task(s, A.t[0], W_TQ1, w.s2, w.a2, w.tq1, false, false, 0);
    sac_q_task(s, A.t[1], W_TQ2, w.s2, w.a2, w.tq2, false, false, 1);
    sac_policy_fin(s, A, eps1, s->rng_stream, false, w.a2, w.logp2, w.ppart);
    if (s->defer_tail) { A.tail = s->tail_dev; A.tail_mode = 2; A.tail_n = 1; }
    if (s->phase_now) {
      // stage 2 of the phase launch: next_obs is the copy the policy task of stage 1 published (w.s2, read after the tile's hand-off);
      // pi(s') is finished under the replay-draw counter (== the step counter once the pending tail has run; the tail runs
      // concurrently in this launch)
      A.fin.use_gather_step = 1;
      A.tail_mode = 0;
      PA.f2 = A;
    } else {
      ILSX_TRY(sac_fwd(s, A, H, act, s->Lq.KP, cs));
    }
  }
  {  // bwd_dx with the TD-target loss head
    BwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 2; A.inv_B = sac_inv_B(s);
    A.gamma = s->cfg.discount; A.reward_scale = s->cfg.reward_scale; A.scal = s->scal;
    A.part_stride = s->cfg.max_batch;
    for (int i = 0; i < 2; ++i) {
      BwdTask& t = A.t[i];
      t.net = net_view(s->Lq, s->base(i == 0 ? W_Q1 : W_Q2));
      for (int l = 0; l < s->Lq.cfg.n_hidden; ++l) { t.hsave[l] = w.hq[i][l]; t.dsave[l] = w.dq[i][l]; }
      t.dhead = w.dhq[i];
      t.loss = LOSS_SAC_CRITIC;
      t.q = s->pv(i == 0 ? w.q1 : w.q2); t.tq1 = s->pv(w.tq1); t.tq2 = s->pv(w.tq2);
      t.logp_next = w.logp2; t.rew = w.r; t.done = w.d;
    }
    if (s->phase_now) {
      PA.b1 = A; PA.flags = s->phase_flags; PA.err = s->phase_err;
      {   // pi(s) is finished inside this launch (kernels.h policy_fin_tile): the policy phase starts from finished actions
        FwdArgs Fz;
        memset(&Fz, 0, sizeof Fz);
        sac_policy_fin(s, Fz, eps2, s->rng_stream + 1, true, w.an, w.logp, w.ppart2);
        PA.fin_pi = Fz.fin; PA.fin_pi.use_gather_step = 1; PA.fin_pi_on = 1;
      }
      ILSX_TRY(launch_phase_a(s->ctx, PA, H, act, std::max(s->Lq.KP, s->Lp.KP), cs, s->no_ct ? nullptr : &s->pct, sac_phase_key(s), s->phase_args_only));
      if (s->phase_args_only) return ILSX_OK;
    } else {
      ILSX_TRY(sac_bwd(s, A, H, act, cs));
    }
  }
  AdamFuse F;
  memset(&F, 0, sizeof F);
  if (s->fuse_now) {
    F.on = 1; F.Gbase = s->G; F.P = s->P; F.M = s->M; F.V = s->V; F.T = s->base(W_TQ1);
    F.b1 = s->cfg.beta_1; F.b2 = 0.999f; F.eps = 1e-8f; F.tau = s->cfg.soft_target_tau;
    F.step_size = &s->scal->adam_q_step; F.bc2_sqrt = &s->scal->adam_q_bc2s;
    if (s->phase_now) F.T = nullptr;   // the target update rides in the policy phase launch (PhaseCArgs::polyak_*)
  }
  return sac_dw(s, s->jobs_q, B, &F);
}

static int sac_critic_update(ilsx_sac* s) {
  if (s->fuse_now) return ILSX_OK;  // already applied by the dW epilogue
  AdamArgs A;
  A.p = s->P; A.g = s->G; A.m = s->M; A.v = s->V; A.tgt = s->base(W_TQ1);
  A.n = (int)(2 * s->nq);
  A.b1 = s->cfg.beta_1; A.b2 = 0.999f; A.eps = 1e-8f; A.tau = s->cfg.soft_target_tau;
  A.step_size = &s->scal->adam_q_step; A.bc2_sqrt = &s->scal->adam_q_bc2s;
  return launch_adam(s->ctx, A);  // t_q is advanced by k_sac_finish at the end of the step
}

static StatsArgs sac_stats_args(ilsx_sac* s);
static int sac_actor_backward(ilsx_sac* s) {
  const SacWs& w = s->ws;
  const int B = s->B, H = s->Lq.cfg.hidden, act = s->Lq.cfg.act, cs = s->cs;
  const float* eps2 = s->eps_explicit ? w.eps2 : nullptr;
  PhaseCArgs PC;
  if (s->phase_now) memset(&PC, 0, sizeof PC);
  // (pi(s) with eps_cur ran as the 4th task of the step's first forward launch, sac_critic_backward)
  {  // fwd Q1(s,a~), Q2(s,a~) with the just-updated critics (sac_alpha.py:144-146)
    FwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 2; A.seed = s->ctx->seed; A.scal = s->scal; A.part_stride = s->cfg.max_batch;
    sac_q_task(s, A.t[0], W_Q1, w.s, w.an, w.q1n, false, true, 0);
    sac_q_task(s, A.t[1], W_Q2, w.s, w.an, w.q2n, false, true, 1);
    if (!s->phase_now) sac_policy_fin(s, A, eps2, s->rng_stream + 1, true, w.an, w.logp, w.ppart2);   // phase steps: finished in phase A already
    if (s->phase_now) { A.tail = s->tail_dev; PC.f3 = A; }   // the extra row of the phase launch advances the replay-draw counter
    else ILSX_TRY(sac_fwd(s, A, H, act, s->Lq.KP, cs));
  }
  {  // bwd_dx through both critics to the action columns
    BwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 2; A.inv_B = sac_inv_B(s); A.scal = s->scal; A.part_stride = s->cfg.max_batch;
    for (int i = 0; i < 2; ++i) {
      BwdTask& t = A.t[i];
      t.net = net_view(s->Lq, s->base(i == 0 ? W_Q1 : W_Q2));
      for (int l = 0; l < s->Lq.cfg.n_hidden; ++l) t.hsave[l] = w.hq[i][l];
      t.loss = LOSS_SAC_ACTORQ; t.which = i; t.q1n = s->pv(w.q1n); t.q2n = s->pv(w.q2n);
      t.dx = w.ga[i]; t.dx_col0 = s->o; t.dx_cols = s->a;
    }
    if (s->phase_now) PC.b2 = A;
    else ILSX_TRY(sac_bwd(s, A, H, act, cs));
  }
  {  // bwd_dx of the policy with the tanh-Gaussian loss head
    BwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = B; A.ntasks = 1; A.inv_B = sac_inv_B(s); A.scal = s->scal; A.part_stride = s->cfg.max_batch;
    A.w_mu = s->cfg.policy_mean_reg_weight; A.w_std = s->cfg.policy_std_reg_weight;
    A.ga_parts = cs; A.ga_stride = s->cfg.max_batch;
    BwdTask& t = A.t[0];
    t.net = net_view(s->Lp, s->base(W_PI));
    for (int l = 0; l < s->Lp.cfg.n_hidden; ++l) { t.hsave[l] = w.hp[l]; t.dsave[l] = w.dp[l]; }
    t.dhead = w.dhp;
    t.loss = LOSS_SAC_POLICY;
    t.raw = w.raw; t.eps = w.epss; t.action = w.an; t.ga1 = w.ga[0]; t.ga2 = w.ga[1];
    if (s->phase_now) {
      PC.b3 = A; PC.flags = s->phase_flags; PC.err = s->phase_err;
      if (s->fuse_now) { PC.polyak_T = s->base(W_TQ1); PC.polyak_P = s->base(W_Q1); PC.polyak_n = (int)(2 * s->nq); PC.polyak_tau = s->cfg.soft_target_tau; }
      else {   // split run: this rank's alpha-gradient partial lands in the arena's slot before the actor all-reduce (the tail is deferred)
        PC.aslot = s->G + 2 * s->nq + s->np; PC.aslot_logp = w.logp; PC.aslot_B = B; PC.aslot_te = s->target_entropy; PC.aslot_invB = sac_inv_B(s);
      }
      ILSX_TRY(launch_phase_c(s->ctx, PC, H, act, std::max(s->Lq.KP, s->Lp.KP), cs, s->no_ct ? nullptr : &s->pct, sac_phase_key(s), s->phase_args_only));
      if (s->phase_args_only) return ILSX_OK;
    } else {
      ILSX_TRY(sac_bwd(s, A, H, act, cs));
    }
  }
  {
    AdamFuse F;
    memset(&F, 0, sizeof F);
    if (s->fuse_now) {  // pi has no target network; its arena offset is handled by the job pointers (Gbase = G)
      F.on = 1; F.Gbase = s->G; F.P = s->P; F.M = s->M; F.V = s->V; F.T = nullptr;
      F.b1 = s->cfg.beta_1; F.b2 = 0.999f; F.eps = 1e-8f; F.tau = 0.f;
      F.step_size = &s->scal->adam_pi_step; F.bc2_sqrt = &s->scal->adam_pi_bc2s;
    }
    ILSX_TRY(sac_dw(s, s->jobs_p, B, &F));
  }
  if (s->fuse_now) return ILSX_OK;  // stats run inside k_sac_tail (sac_actor_update)
  if (s->phase_now && s->defer_tail) return ILSX_OK;   // split run on the phase kernels: the slot came from phase C, statistics from the flushed tail
  StatsArgs S = sac_stats_args(s);
  ProfScope ps(s->ctx, ILSX_K_SAC_STATS);
  ILSX_LAUNCH(ps, k_sac_stats, dim3(1), dim3(256), 0, s->ctx->stream, S);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

static StatsArgs sac_stats_args(ilsx_sac* s) {
  const SacWs& w = s->ws;
  const int B = s->B;
  StatsArgs S;
  S.q1 = s->pv(w.q1); S.q2 = s->pv(w.q2); S.tq1 = s->pv(w.tq1); S.tq2 = s->pv(w.tq2);
  S.q1n = s->pv(w.q1n); S.q2n = s->pv(w.q2n);
  S.logp2 = w.logp2; S.r = w.r; S.d = w.d; S.logp = w.logp; S.raw = w.raw;
  S.B = B; S.a = s->a;
  S.gamma = s->cfg.discount; S.reward_scale = s->cfg.reward_scale;
  S.w_mu = s->cfg.policy_mean_reg_weight; S.w_std = s->cfg.policy_std_reg_weight;
  S.target_entropy = s->target_entropy; S.inv_B = sac_inv_B(s);
  S.scal = s->scal; S.alpha_grad_slot = s->G + 2 * s->nq + s->np;
  S.slot_given = 0;
  return S;
}

static int sac_actor_update(ilsx_sac* s) {
  AdamArgs A;
  const size_t off = 2 * s->nq;
  A.p = s->P + off; A.g = s->G + off; A.m = s->M + off; A.v = s->V + off; A.tgt = nullptr;
  A.n = (int)s->np;
  A.b1 = s->cfg.beta_1; A.b2 = 0.999f; A.eps = 1e-8f; A.tau = 0.f;
  A.step_size = &s->scal->adam_pi_step; A.bc2_sqrt = &s->scal->adam_pi_bc2s;
  if (s->fuse_now && s->col) {
    SacLaunch l; memset(&l, 0, sizeof l); l.kind = 3;
    l.t.S = sac_stats_args(s); l.t.train_alpha = s->cfg.train_alpha; l.t.lr = s->cfg.alpha_lr; l.t.b1 = s->cfg.beta_1;
    l.t.b2 = 0.999f; l.t.eps = 1e-8f; l.t.qf_lr = s->cfg.qf_lr; l.t.policy_lr = s->cfg.policy_lr;
    s->col->L.push_back(l);
    return ILSX_OK;
  }
  if (s->fuse_now && s->defer_tail) return ILSX_OK;   // the next step's first launch (or sac_flush_tail) runs it
  if (s->fuse_now) {
    ProfScope ps(s->ctx, ILSX_K_SAC_FINISH);
    ILSX_LAUNCH(ps, k_sac_tail, dim3(1), dim3(256), 0, s->ctx->stream, sac_stats_args(s), s->cfg.train_alpha,
                       s->cfg.alpha_lr, s->cfg.beta_1, 0.999f, 1e-8f, s->cfg.qf_lr, s->cfg.policy_lr, /*deferred=*/0);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  ILSX_TRY(launch_adam(s->ctx, A));
  if (s->phase_now && s->defer_tail) return ILSX_OK;   // split run on the phase kernels: the next step's phase A (or sac_flush_tail) finishes
  ProfScope ps(s->ctx, ILSX_K_SAC_FINISH);
  ILSX_LAUNCH(ps, k_sac_finish, dim3(1), dim3(1), 0, s->ctx->stream, s->scal, (const float*)(s->G + 2 * s->nq + s->np),
                     s->cfg.train_alpha, s->cfg.alpha_lr, s->cfg.beta_1, 0.999f, 1e-8f, s->cfg.qf_lr, s->cfg.policy_lr);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

static int sac_refresh_adam(ilsx_sac* s) {
  hipLaunchKernelGGL(k_sac_refresh_adam, dim3(1), dim3(1), 0, s->ctx->stream, s->scal, s->cfg.qf_lr, s->cfg.policy_lr,
                     s->cfg.beta_1, 0.999f);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// Deferred tail: see TailLite in kernels.h.  Enabled for the duration of one train_from_replay call when the whole step runs on
// the fused column-split path (no gradient all-reduce between the phases, no XCD confinement); the last step's tail is flushed
// by the ordinary tail kernel, so no tail is ever pending when the call returns.
static bool sac_is_split(const ilsx_sac* s);
static TailLite sac_tail_lite(ilsx_sac* s, int B) {
  TailLite t;
  memset(&t, 0, sizeof t);
  const int keep = s->B;
  s->B = B;
  t.logp = s->ws.logp; t.B = B; t.target_entropy = s->target_entropy; t.inv_B = sac_inv_B(s);
  s->B = keep;
  t.alpha_grad_slot = s->G + 2 * s->nq + s->np; t.scal = s->scal;
  t.train_alpha = s->cfg.train_alpha; t.lr = s->cfg.alpha_lr; t.b1 = s->cfg.beta_1; t.b2 = 0.999f; t.eps = 1e-8f;
  t.qf_lr = s->cfg.qf_lr; t.policy_lr = s->cfg.policy_lr;
  t.slot_given = sac_is_split(s) ? 1 : 0;
  return t;
}
// A split run (gradient all-reduce between backward and update) defers its tail only when the step runs on the merged phase kernels
//   A , dW{Q} , all-reduce , Adam{Q} + Polyak , C (+ this rank's alpha-gradient partial) , dW{pi} , all-reduce , Adam{pi}
// (6 launches + 2 collectives; the tail rides in the next step's A and applies the all-reduced alpha gradient, TailLite::slot_given).
// Without the phase kernels (ILSX_NO_PHASE, ILSX_SPLIT_NO_PHASE, graph segments, a grid that does not fit) it keeps the un-deferred
// one-launch-per-stage sequence with k_sac_stats / k_sac_finish.
// Above one rank the phase-kernel form is OPT-IN (ILSX_SPLIT_PHASE=1): its deferred tail fed from the all-reduced alpha slot and its
// per-window roll-back vote have only ever run on a one-rank communicator (ILSX_SPLIT_FORCE), so a multi-rank run takes the plain
// one-launch-per-stage sequence unless asked.  When asked, the ranks AGREE on it once per window — the predicate rests on device_cus, kernel
// occupancy and environment variables, which need not match across ranks, and one rank deciding differently would deadlock the job inside
// RCCL: every rank contributes "I cannot" (0 / 1) to a one-float all-reduce and the window runs on the phase kernels only if nobody said so.
static int sac_split_on_phase(ilsx_sac* s, int B, bool* on) {
  static const bool segments = []() { const char* e = getenv("ILSX_SPLIT_SEGMENTS"); return e && atoi(e) != 0; }();
  const bool mine = getenv("ILSX_SPLIT_NO_PHASE") == nullptr && !segments && sac_window_may_use_phase(s, B);
  *on = mine;
  if (!(s->ctx->comm && s->ctx->comm_n > 1)) return ILSX_OK;
  if (getenv("ILSX_SPLIT_PHASE") == nullptr) { *on = false; return ILSX_OK; }
  hipStream_t st = s->ctx->stream;
  if (!s->vote) ILSX_TRY(ctx_alloc(s->ctx, 4 * sizeof(float), (void**)&s->vote, true));
  float v = mine ? 0.0f : 1.0f;
  HIPCHK(hipMemcpyAsync(s->vote, &v, sizeof v, hipMemcpyHostToDevice, st));
  ILSX_TRY(comm_allreduce_sum(s->ctx, s->vote, 1));
  HIPCHK(hipMemcpyAsync(&v, s->vote, sizeof v, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  *on = v == 0.0f;
  return ILSX_OK;
}
static int sac_defer_begin(ilsx_sac* s, int B) {
  static const bool off = getenv("ILSX_NO_DEFER_TAIL") != nullptr || getenv("ILSX_NO_FUSE") != nullptr;
  s->defer_tail = false;
  const bool split = sac_is_split(s);
  if (off || s->cs <= 1 || s->col) return ILSX_OK;
  if (split) {
    bool on = false;
    ILSX_TRY(sac_split_on_phase(s, B, &on));
    if (!on) return ILSX_OK;
  }
  if (!s->tail_dev) ILSX_TRY(ctx_alloc(s->ctx, sizeof(TailLite), (void**)&s->tail_dev));
  if (s->tail_B != B || s->tail_split != split) {
    const TailLite t = sac_tail_lite(s, B);
    HIPCHK(hipMemcpyAsync(s->tail_dev, &t, sizeof t, hipMemcpyHostToDevice, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));   // `t` lives on this stack frame
    s->tail_B = B; s->tail_split = split;
  }
  s->defer_tail = true;
  return ILSX_OK;
}
static int sac_flush_tail(ilsx_sac* s, bool deferred) {   // the pending tail of the call's last step (+ its statistics if requested)
  if (!deferred) return ILSX_OK;
  ProfScope ps(s->ctx, ILSX_K_SAC_FINISH);
  StatsArgs S = sac_stats_args(s);
  S.slot_given = sac_is_split(s) ? 1 : 0;   // split run: the last step's alpha gradient was all-reduced into the slot (sac_actor_backward)
  ILSX_LAUNCH(ps, k_sac_tail, dim3(1), dim3(256), 0, s->ctx->stream, S, s->cfg.train_alpha,
              s->cfg.alpha_lr, s->cfg.beta_1, 0.999f, 1e-8f, s->cfg.qf_lr, s->cfg.policy_lr, /*deferred=*/1);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// Split run (cfg.grad_world = G > 1): this rank holds B/G rows; the gradient arena is summed over the ranks between
// backward and the optimiser step with RCCL on the ctx stream (segment 0 = critics, 1 = actor | alpha slot: two messages
// because the actor loss is evaluated with the post-update critics, sac_alpha.py:142-146).
// ILSX_SPLIT_FORCE=1 (tests): a run with grad_world = 1 whose ctx carries a one-rank communicator takes the split-run code path
// too — un-fused phases with ncclAllReduce between them — so the collective plumbing is exercised on a single-GPU box.
static bool sac_is_split(const ilsx_sac* s) {
  return s->cfg.grad_world > 1 || (s->ctx->comm != nullptr && getenv("ILSX_SPLIT_FORCE") != nullptr);
}
static int sac_allreduce(ilsx_sac* s, int segment) {
  if (!sac_is_split(s)) return ILSX_OK;
  if (segment == 0) return comm_allreduce_sum(s->ctx, s->G, 2 * s->nq);
  return comm_allreduce_sum(s->ctx, s->G + 2 * s->nq, s->np + 4);
}
static int sac_check_world(const ilsx_sac* s, const char* who) {
  if (s->cfg.grad_world == 1) return ILSX_OK;
  if (!s->ctx->comm || s->ctx->comm_n != s->cfg.grad_world)
    ILSX_FAIL(ILSX_ERR_STATE, "%s: grad_world=%d needs a communicator of that many ranks on the ctx (ilsx_comm_init; found %d) — "
              "or drive the four phases and all-reduce ilsx_sac_grads_ptr yourself", who, s->cfg.grad_world, s->ctx->comm ? s->ctx->comm_n : 0);
  return ILSX_OK;
}

static int sac_full_step(ilsx_sac* s) {
  static const bool no_fuse = getenv("ILSX_NO_FUSE") != nullptr;
  s->fuse_now = !no_fuse && !sac_is_split(s);
  int rc = sac_critic_backward(s);
  if (rc == ILSX_OK) rc = sac_allreduce(s, 0);
  if (rc == ILSX_OK) rc = sac_critic_update(s);
  if (rc == ILSX_OK) rc = sac_actor_backward(s);
  if (rc == ILSX_OK) rc = sac_allreduce(s, 1);
  if (rc == ILSX_OK) rc = sac_actor_update(s);
  s->fuse_now = false;
  return rc;
}

static int sac_request_stats(ilsx_sac* s, hipStream_t st = nullptr) {  // ordered on the stream before the step that must produce them
  static const int one = 1;
  HIPCHK(hipMemcpyAsync(&s->scal->want_stats, &one, sizeof(int), hipMemcpyHostToDevice, st ? st : s->ctx->stream));
  return ILSX_OK;
}

static int sac_read_stats(ilsx_sac* s, ilsx_sac_stats* out) {
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  out->qf1_loss = h.qf1_loss; out->qf2_loss = h.qf2_loss; out->policy_loss = h.policy_loss;
  out->alpha_loss = h.alpha_loss; out->alpha = h.alpha;
  out->q1_mean = h.q1_mean; out->q2_mean = h.q2_mean; out->log_pi_mean = h.log_pi_mean;
  out->policy_mu_mean = h.mu_mean; out->policy_log_std_mean = h.log_std_mean;
  out->log_alpha = h.log_alpha;
  for (int i = 0; i < 5; ++i) { out->ext_std[i] = h.ext_std[i]; out->ext_max[i] = h.ext_max[i]; out->ext_min[i] = h.ext_min[i]; }
  s->stats_cache = *out; s->stats_cached = true;   // what ilsx_sac_last_stats hands out: the statistics as they stood right after THEIR step
  return ILSX_OK;
}

extern "C" int ilsx_sac_set_batch(ilsx_sac* s, const float* obs, const float* act, const float* rew,
                                  const float* done, const float* nobs, int B, const float* eps_next,
                                  const float* eps_cur) {
  if (!s || !obs || !act || !rew || !done || !nobs) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_set_batch: NULL argument");
  if (B < 1 || B > s->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch=%d", B, s->cfg.max_batch);
  if ((eps_next == nullptr) != (eps_cur == nullptr)) ILSX_FAIL(ILSX_ERR_ARG, "give both eps_next and eps_cur or neither");
  HIPCHK(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  const SacWs& w = s->ws;
  const size_t o = s->o, a = s->a, b = B;
  HIPCHK(hipMemcpyAsync(w.s, obs, b * o * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(w.a, act, b * a * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(w.r, rew, b * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(w.d, done, b * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(w.s2, nobs, b * o * 4, hipMemcpyDeviceToDevice, st));
  if (eps_next) {
    HIPCHK(hipMemcpyAsync(w.eps1, eps_next, b * a * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(w.eps2, eps_cur, b * a * 4, hipMemcpyDeviceToDevice, st));
  }
  s->eps_explicit = eps_next != nullptr;
  s->B = B;
  return ILSX_OK;
}

// internal (the adversarial-IRL loop, ilsx_disc.hip): the agent's batch arrays, for a caller on the same stream that fills them in place
// (replay sample -> reward relabel -> step, without the five device-to-device copies of ilsx_sac_set_batch), and the step on them
int sac_staged_batch(ilsx_sac* s, int B, float** obs, float** act, float** rew, float** done, float** nobs) {
  if (!s || B < 1 || B > s->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch=%d", B, s ? s->cfg.max_batch : 0);
  const SacWs& w = s->ws;
  *obs = w.s; *act = w.a; *rew = w.r; *done = w.d; *nobs = w.s2;
  s->eps_explicit = false;
  s->B = B;
  return ILSX_OK;
}
int sac_step_staged(ilsx_sac* s, ilsx_sac_stats* stats) {
  ILSX_TRY(sac_check_world(s, "ilsx_advirl_train"));
  if (stats) ILSX_TRY(sac_request_stats(s));
  ILSX_TRY(sac_full_step(s));
  if (stats) return sac_read_stats(s, stats);
  return ILSX_OK;
}
int sac_dims(const ilsx_sac* s, int* o, int* a) { *o = s->o; *a = s->a; return ILSX_OK; }
// A window of steps on batches the caller stages in place (sac_staged_batch): the deferred tail and the merged phase kernels, as inside
// one ilsx_sac_train_from_replay call — the rows just come from the batch arrays instead of the in-kernel replay draw.
static bool sac_phase_ok(ilsx_sac* s, int B);
static int sac_phase_check(ilsx_sac* s, int B, bool deferred, const char* who, bool teardown = false);
int sac_window_begin(ilsx_sac* s, int B) {
  ILSX_TRY(sac_check_world(s, "ilsx_advirl_train"));
  s->B = B; s->eps_explicit = false;
  return sac_defer_begin(s, B);
}
int sac_window_step(ilsx_sac* s) {
  s->gather_rb = nullptr;
  s->phase_now = sac_phase_ok(s, s->B);
  s->phase_last = s->phase_now;
  const int rc = sac_full_step(s);
  s->phase_now = false;
  return rc;
}
int sac_window_end(ilsx_sac* s, bool teardown) {
  const bool deferred = s->defer_tail;
  s->defer_tail = false;
  ILSX_TRY(sac_flush_tail(s, deferred));
  return sac_phase_check(s, s->B, deferred, "ilsx_advirl_train", teardown);
}

#define SAC_PHASE(name, fn)                                                        \
  extern "C" int name(ilsx_sac* s) {                                               \
    if (!s) ILSX_FAIL(ILSX_ERR_ARG, #name ": NULL agent");                         \
    if (s->B < 1) ILSX_FAIL(ILSX_ERR_STATE, #name ": no batch staged (ilsx_sac_set_batch)"); \
    HIPCHK(hipSetDevice(s->ctx->device));                                          \
    s->stats_cached = false; /* phases driven by hand: ilsx_sac_last_stats reads the device */ \
    return fn(s);                                                                  \
  }
SAC_PHASE(ilsx_sac_critic_backward, sac_critic_backward)
SAC_PHASE(ilsx_sac_critic_update, sac_critic_update)
SAC_PHASE(ilsx_sac_actor_backward, sac_actor_backward)
SAC_PHASE(ilsx_sac_actor_update, sac_actor_update)

extern "C" int ilsx_sac_last_stats(ilsx_sac* s, ilsx_sac_stats* stats) {
  if (!s || !stats) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_last_stats: NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  // the statistics of the step that was asked for them (the first batch of a grouped call), not whatever later steps left in the device
  // scalars: "Alpha" is the temperature after that step's own update (sac_alpha.py:160-166,208-212), and every later step moves it
  if (s->stats_cached) { *stats = s->stats_cache; return ILSX_OK; }
  return sac_read_stats(s, stats);
}

extern "C" int ilsx_sac_train_step(ilsx_sac* s, const float* obs, const float* act, const float* rew,
                                   const float* done, const float* nobs, int B, const float* eps_next,
                                   const float* eps_cur, ilsx_sac_stats* stats) {
  if (s) ILSX_TRY(sac_check_world(s, "ilsx_sac_train_step"));
  ILSX_TRY(ilsx_sac_set_batch(s, obs, act, rew, done, nobs, B, eps_next, eps_cur));
  if (stats) ILSX_TRY(sac_request_stats(s));
  ILSX_TRY(sac_full_step(s));
  if (stats) return sac_read_stats(s, stats);
  return ILSX_OK;
}

// The merged phase kernels need: the column-split path on narrow inputs, the fused in-kernel replay draw, the deferred tail, no
// gradient all-reduce between the phases, no per-workgroup stamps, and every workgroup of a phase launch resident at once.
static bool sac_phase_ok(ilsx_sac* s, int B) {
  const bool off = getenv("ILSX_NO_PHASE") != nullptr;
  return !off && !s->phase_broken && s->cs > 1 && !s->h0scr && !s->col && s->defer_tail &&
         s->ctx->xcd_shift == 0 && phase_fits(s->ctx, B, s->Lq.cfg.hidden, s->cs, 4);   // (a split run defers its tail only for the phase kernels: sac_defer_begin)
}
static bool sac_phase_possible(ilsx_sac* s, int B) {   // as sac_phase_ok, for the state a train_from_replay call ran in (defer_tail already cleared)
  const bool off = getenv("ILSX_NO_PHASE") != nullptr;   // read per call: tests switch between the two paths in one process
  return !off && s->cs > 1 && !s->h0scr && phase_fits(s->ctx, B, s->Lq.cfg.hidden, s->cs, 4);
}
// Checkpoint / roll-back of everything a gradient step mutates — device scalars (log alpha and its moments, step / Philox counters,
// Adam scalars), parameters + targets, gradients, Adam moments: one contiguous range of the agent's slab (ilsx_sac_create adds them
// in this order).  Taken at the start of every window that may run on the merged phase kernels (~3 MB device-to-device, once per
// train call); restored when the window's phase kernels reported a broken wait or placement, before the window is re-run.
static size_t sac_snap_range(const ilsx_sac* s) { return (size_t)((const char*)(s->V + (2 * s->nq + s->np)) - (const char*)s->scal); }
int sac_snapshot_take(ilsx_sac* s) {
  const size_t n = sac_snap_range(s);
  if (!s->snap) { ILSX_TRY(ctx_alloc(s->ctx, n, &s->snap, false)); s->snap_bytes = n; }
  HIPCHK(hipMemcpyAsync(s->snap, s->scal, n, hipMemcpyDeviceToDevice, s->ctx->stream));
  s->snap_valid = true;
  return ILSX_OK;
}
int sac_snapshot_restore(ilsx_sac* s) {
  // only the checkpoint of THIS window: one left over from an earlier call would roll the agent back to some other point in its history
  if (!s->snap || !s->snap_valid) ILSX_FAIL(ILSX_ERR_STATE, "no checkpoint of this window to roll back to");
  s->snap_valid = false;
  HIPCHK(hipMemcpyAsync(s->scal, s->snap, s->snap_bytes, hipMemcpyDeviceToDevice, s->ctx->stream));
  HIPCHK(hipMemsetAsync(s->phase_flags, 0, (size_t)PHASE_FLAG_WORDS * sizeof(unsigned), s->ctx->stream));
  return ILSX_OK;
}
bool sac_window_may_use_phase(ilsx_sac* s, int B) { return !s->phase_broken && s->cs > 1 && !s->col && sac_phase_possible(s, B) && s->ctx->xcd_shift == 0; }
// After a window of steps that may have run on the merged phase kernels: a workgroup that gave up waiting, or a row tile whose workgroups
// sat on two XCDs, left a mark — the window's updates are then not to be trusted.  Returns ILSX_RETRY_WINDOW: the caller rolls the window
// back (sac_snapshot_restore) and runs it again; the agent stays on one launch per stage from here on.
static int sac_phase_check(ilsx_sac* s, int B, bool deferred, const char* who, bool teardown) {
  hipStream_t st = s->ctx->stream;
  // every exit but the retry ends the window: its checkpoint is then nobody's roll-back point any more (also when a copy or the vote fails)
  struct SnapGuard { ilsx_sac* s; bool keep = false; ~SnapGuard() { if (!keep) s->snap_valid = false; } } snap_guard{s};
  // teardown: the window is being closed on an error path (ilsx_advirl_train's guard) — this rank's peers are inside some other collective,
  // so no vote is issued (a mismatched all-reduce hangs instead of returning the error) and nothing is re-run: the caller already fails
  if (deferred && s->snap_valid && !teardown) {   // == the window began with sac_window_may_use_phase (the one predicate: the checkpoint is taken under it)
    // a phase-kernel workgroup that gave up waiting left a mark: the steps of this call are not to be trusted
    int err = 0;
    unsigned masks[PHASE_MAX_TILES * 32];
    HIPCHK(hipMemcpyAsync(&err, s->phase_err, sizeof err, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(masks, s->phase_flags + PHASE_MASK_WORD(0), sizeof masks, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int t = 0; t < PHASE_MAX_TILES; ++t)
      if (masks[t * 32] & (masks[t * 32] - 1)) err |= 2;   // a tile's workgroups ran on more than one XCD: its exchange went through two L2s
    if (s->debug_break) { err |= 1; s->debug_break = false; }
    if (sac_is_split(s) && s->ctx->comm && s->ctx->comm_n > 1) {
      // a split run rolls back on EVERY rank or on none: a window re-run on one rank alone would issue collectives the others never join.
      // One float per rank through the run's own communicator, once per window (not per step); nothing to vote on at one rank.
      if (!s->vote) ILSX_TRY(ctx_alloc(s->ctx, 4 * sizeof(float), (void**)&s->vote, true));
      float v = err ? 1.0f : 0.0f;
      HIPCHK(hipMemcpyAsync(s->vote, &v, sizeof v, hipMemcpyHostToDevice, st));
      ILSX_TRY(comm_allreduce_sum(s->ctx, s->vote, 1));
      HIPCHK(hipMemcpyAsync(&v, s->vote, sizeof v, hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (v > 0.0f) err |= 4;   // some rank's hand-offs broke: this rank rolls back with it (the two paths are bit-identical)
    }
    if (err) {
      s->phase_broken = true;
      if (s->graph) { hipGraphExecDestroy(s->graph); s->graph = nullptr; }
      HIPCHK(hipMemsetAsync(s->phase_err, 0, sizeof(int), st));
      if (getenv("ILSX_PHASE_VERBOSE"))
        fprintf(stderr, "[ilsx] %s: a merged phase kernel %s; the window is rolled back and re-run, this agent stays on one launch per stage\n", who,
                (err & 1) ? "timed out waiting for the workgroups of its tile (other kernels sharing this GPU?)"
                : (err & 2) ? "found the workgroups of one row tile on different XCDs" : "broke on another rank of this split run");
      s->phase_fallbacks += 1;
      snap_guard.keep = true;   // sac_snapshot_restore consumes it
      return ILSX_RETRY_WINDOW;
    }
  }
  return ILSX_OK;
}

// The two phase launches' descriptor blocks, built exactly as the step that follows will build them (same agent state) and put into this agent's
// constant-memory slots — a dry pass that launches nothing, run before a capture begins (launch_phase_a / _c then find the slots current).
static int sac_phase_const_prime(ilsx_sac* s, ilsx_replay* rb) {
  phase_const_prepare(s->ctx, &s->pct);
  if (s->pct.slot < 0 || s->cs <= 1) return ILSX_OK;
  static const bool no_fuse = getenv("ILSX_NO_FUSE") != nullptr;
  s->gather_rb = rb; s->phase_now = true; s->fuse_now = !no_fuse && !sac_is_split(s);
  s->phase_args_only = true;
  int rc = sac_critic_backward(s);
  if (rc == ILSX_OK) rc = sac_actor_backward(s);
  s->phase_args_only = false;
  s->gather_rb = nullptr; s->phase_now = false; s->fuse_now = false;
  return rc;
}

static int sac_sample_and_step(ilsx_sac* s, ilsx_replay* rb, int B) {
  const SacWs& w = s->ws;
  if (s->cs > 1) {
    s->gather_rb = rb;   // sampling is fused into the first forward launch
    s->phase_now = sac_phase_ok(s, B);
    s->phase_last = s->phase_now;
    const int rc = sac_full_step(s);
    s->gather_rb = nullptr;
    s->phase_now = false;
    return rc;
  }
  ILSX_TRY(replay_launch_sample(rb, B, nullptr, s->scal, 0, w.s, w.a, w.r, w.d, w.s2, nullptr));
  return sac_full_step(s);
}

static int sac_train_from_replay_once(ilsx_sac* s, ilsx_replay* rb, int n_steps, int B, ilsx_sac_stats* stats);
// Split run (one run over G GPUs): the step's launches between the two gradient all-reduces are captured ONCE as three graph segments
//   [draw + critic backward]  all-reduce(critic gradients)  [critic Adam + Polyak ; actor backward]  all-reduce(actor | alpha)  [actor Adam ; tail]
// and replayed with the two ncclAllReduce calls enqueued directly between them: ~13 kernel launches per step become 3 graph launches,
// while the collective itself stays OUT of the capture (RCCL inside a stream capture is the one thing a single-GPU box cannot exercise
// with more than one rank; ILSX_SPLIT_GRAPH=1 captures the whole step including it).  OPT-IN (ILSX_SPLIT_SEGMENTS=1): measured at one rank
// (profiles/r04_split_run_1rank.jsonl) the three small graphs are SLOWER than launching the 13 kernels directly — 107 us against 90 us per
// step: a hipGraphLaunch costs more than the four launches it replaces — so the default stays direct launches.
static int sac_split_segment(ilsx_sac* s, ilsx_replay* rb, int B, int seg) {
  s->fuse_now = false;
  if (seg == 0) {
    if (s->cs > 1) s->gather_rb = rb;   // rows drawn inside the first forward launch
    else { const SacWs& w = s->ws; ILSX_TRY(replay_launch_sample(rb, B, nullptr, s->scal, 0, w.s, w.a, w.r, w.d, w.s2, nullptr)); }
    const int rc = sac_critic_backward(s);
    s->gather_rb = nullptr;
    return rc;
  }
  if (seg == 1) { ILSX_TRY(sac_critic_update(s)); return sac_actor_backward(s); }
  return sac_actor_update(s);
}
static int sac_split_segments_step(ilsx_sac* s, ilsx_replay* rb, int B) {
  hipStream_t st = s->ctx->stream;
  if (s->seg_rb != rb || s->seg_B != B || !s->seg_graph[0]) {
    for (hipGraphExec_t& g : s->seg_graph) if (g) { hipGraphExecDestroy(g); g = nullptr; }
    for (int seg = 0; seg < 3; ++seg) {
      hipGraph_t g = nullptr;
      HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      s->no_ct = true;   // these cached segments are not re-primed before their replays: their phase launches keep the argument segment
      const int rc = sac_split_segment(s, rb, B, seg);
      s->no_ct = false;
      const hipError_t e = hipStreamEndCapture(st, &g);
      if (rc != ILSX_OK) { if (g) hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) ILSX_FAIL(ILSX_ERR_HIP, "hipStreamEndCapture (split segment %d) failed: %s", seg, hipGetErrorString(e));
      const hipError_t e2 = hipGraphInstantiate(&s->seg_graph[seg], g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (e2 != hipSuccess) { s->seg_graph[seg] = nullptr; ILSX_FAIL(ILSX_ERR_HIP, "hipGraphInstantiate (split segment %d) failed: %s", seg, hipGetErrorString(e2)); }
    }
    s->seg_rb = rb; s->seg_B = B;
  }
  HIPCHK(hipGraphLaunch(s->seg_graph[0], st));
  ILSX_TRY(sac_allreduce(s, 0));
  HIPCHK(hipGraphLaunch(s->seg_graph[1], st));
  ILSX_TRY(sac_allreduce(s, 1));
  HIPCHK(hipGraphLaunch(s->seg_graph[2], st));
  return ILSX_OK;
}

extern "C" int ilsx_sac_train_from_replay(ilsx_sac* s, ilsx_replay* rb, int n_steps, int B, ilsx_sac_stats* stats) {
  // The reference records eval_statistics on the FIRST train_step after end_epoch (sac_alpha.py:185-190: `if self.eval_statistics is
  // None`), i.e. on the first batch of the call that asks for them.  The deferred tail has no statistics branch, so the call is cut there:
  // one step with the ordinary tail (which computes them), then the other n - 1 steps as a window of their own.  A call boundary does not
  // change any parameter (tests/test_hip_parity.py _DEFER_SCRIPT), so the split is invisible except for which batch the statistics describe.
  if (stats && n_steps > 1) {
    ILSX_TRY(ilsx_sac_train_from_replay(s, rb, 1, B, stats));
    return ilsx_sac_train_from_replay(s, rb, n_steps - 1, B, nullptr);
  }
  int rc = sac_train_from_replay_once(s, rb, n_steps, B, stats);
  if (rc == ILSX_RETRY_WINDOW) {   // the phase kernels' hand-offs broke (shared GPU): back to the checkpoint, same steps on one launch per stage
    ILSX_TRY(sac_snapshot_restore(s));
    rc = sac_train_from_replay_once(s, rb, n_steps, B, stats);
    if (rc == ILSX_RETRY_WINDOW) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_sac_train_from_replay: the fallback path asked for a retry");
  }
  return rc;
}
void phase_wgs_per_cu(int H, int cs, int* occ_a, int* occ_c);   // ilsx_core.hip
extern "C" int ilsx_sac_phase_state(ilsx_sac* s, int* fallbacks, int* disabled, int* last_window_on_phase, int* wgs_per_cu_a, int* wgs_per_cu_c) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_phase_state: NULL agent");
  if (fallbacks) *fallbacks = s->phase_fallbacks;
  if (disabled) *disabled = s->phase_broken ? 1 : 0;
  if (last_window_on_phase) *last_window_on_phase = s->phase_last ? 1 : 0;
  int oa = 0, oc = 0;
  if (s->cs > 1 && (s->Lq.cfg.hidden == 256 || s->Lq.cfg.hidden == 128)) phase_wgs_per_cu(s->Lq.cfg.hidden, s->cs, &oa, &oc);
  if (wgs_per_cu_a) *wgs_per_cu_a = oa;
  if (wgs_per_cu_c) *wgs_per_cu_c = oc;
  return ILSX_OK;
}
extern "C" int ilsx_sac_debug_break_phase(ilsx_sac* s) {   // test aid: the next window finds a "timed out" mark and takes the roll-back path
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_debug_break_phase: NULL agent");
  // the mark is set on the DEVICE, as a wait that gave up would set it: the window's weight-gradient launches then arm the arrival counters
  // satisfied (PHASE_FLAG_DEAD) and its later phase launches run through without waiting — on stale data, which the roll-back discards
  static const int one = 1;
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemcpyAsync(s->phase_err, &one, sizeof one, hipMemcpyHostToDevice, s->ctx->stream));
  s->debug_break = true;
  return ILSX_OK;
}
static int sac_train_from_replay_once(ilsx_sac* s, ilsx_replay* rb, int n_steps, int B, ilsx_sac_stats* stats) {
  if (!s || !rb || n_steps < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_train_from_replay: bad argument");
  if (B < 1 || B > s->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch=%d", B, s->cfg.max_batch);
  if (rb->o != s->o || rb->a != s->a) ILSX_FAIL(ILSX_ERR_ARG, "replay dims (%d,%d) != agent dims (%d,%d)", rb->o, rb->a, s->o, s->a);
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "replay buffer is empty");
  ILSX_TRY(sac_check_world(s, "ilsx_sac_train_from_replay"));
  HIPCHK(hipSetDevice(s->ctx->device));
  ILSX_TRY(replay_flush_state(rb));   // the fused gather reads {size} from the ring's device state
  hipStream_t st = s->ctx->stream;
  s->B = B;
  s->eps_explicit = false;
  // split runs replay the step from a hipGraph only on request: RCCL calls inside a capture are the one part of this path
  // that could not be exercised with more than one rank on the development box
  static const bool split_graph = getenv("ILSX_SPLIT_GRAPH") != nullptr;
  static const bool no_graph_env = getenv("ILSX_NO_GRAPH") != nullptr;
  const bool no_graph = no_graph_env || (sac_is_split(s) && !split_graph);
  if (n_steps == 0) return stats ? sac_read_stats(s, stats) : ILSX_OK;
  ILSX_TRY(sac_defer_begin(s, B));
  struct DeferGuard { ilsx_sac* s; ~DeferGuard() { s->defer_tail = false; } } guard{s};   // never left on, whatever path returns
  const bool deferred = s->defer_tail;
  if (deferred && sac_window_may_use_phase(s, B)) ILSX_TRY(sac_snapshot_take(s));   // the roll-back point of this window
  static const bool segments = []() { const char* e = getenv("ILSX_SPLIT_SEGMENTS"); return e && atoi(e) != 0; }();   // measured slower: see sac_split_segments_step
  if (sac_is_split(s) && !split_graph && !no_graph_env && segments && !s->ctx->prof_on) {
    for (int i = 0; i < n_steps; ++i) {
      if (stats && i == n_steps - 1) ILSX_TRY(sac_request_stats(s));
      ILSX_TRY(sac_split_segments_step(s, rb, B));
    }
  } else if (no_graph || s->ctx->prof_on) {
    for (int i = 0; i < n_steps; ++i) {
      if (stats && i == n_steps - 1) ILSX_TRY(sac_request_stats(s));
      const int rc = sac_sample_and_step(s, rb, B);
      if (rc != ILSX_OK) return rc;
    }
  } else {
    const bool phase = sac_phase_ok(s, B);
    s->phase_last = phase;
    // The phase kernels' descriptor blocks into this agent's constant-memory slots — before a capture begins, and before EVERY replay of a
    // cached graph too: between two calls another step form of the same agent (an explicit-batch step, the adversarial-IRL loop, a profiled
    // window: direct launches under another state key) may have put ITS blocks there, and the graph's launches would read those.  The dry pass
    // costs a few host microseconds when the slots are current.
    static const bool no_reprime = getenv("ILSX_PHASE_CT_NO_REPRIME") != nullptr;   // test aid: shows what the re-prime is for (tests/test_hip_parity.py)
    const bool rebuild = !s->graph || s->graph_rb != rb || s->graph_B != B || s->graph_defer != s->defer_tail || s->graph_phase != phase;
    if (phase && (rebuild || !no_reprime)) ILSX_TRY(sac_phase_const_prime(s, rb));
    if (!s->graph || s->graph_rb != rb || s->graph_B != B || s->graph_defer != s->defer_tail || s->graph_phase != phase) {
      if (s->graph) { hipGraphExecDestroy(s->graph); s->graph = nullptr; }
      hipGraph_t g = nullptr;
      HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      int rc = sac_sample_and_step(s, rb, B);
      hipError_t e = hipStreamEndCapture(st, &g);
      if (rc != ILSX_OK) { if (g) hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) ILSX_FAIL(ILSX_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
      e = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (e != hipSuccess) { s->graph = nullptr; ILSX_FAIL(ILSX_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
      s->graph_rb = rb; s->graph_B = B; s->graph_defer = s->defer_tail; s->graph_phase = phase;
    }
    for (int i = 0; i < n_steps; ++i) {
      if (stats && i == n_steps - 1) ILSX_TRY(sac_request_stats(s));
      HIPCHK(hipGraphLaunch(s->graph, st));
    }
  }
  s->defer_tail = false;
  ILSX_TRY(sac_flush_tail(s, deferred));
  ILSX_TRY(sac_phase_check(s, B, deferred, "ilsx_sac_train_from_replay"));
  if (stats) return sac_read_stats(s, stats);
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------------
static const NetLayout* sac_layout(const ilsx_sac* s, int which) { return which == W_PI ? &s->Lp : &s->Lq; }

extern "C" int ilsx_sac_get_params(ilsx_sac* s, int which, float* dst, size_t n, int dst_is_device) {
  if (!s || !dst || which < 0 || which > 4) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_get_params: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  return net_download_flat(s->ctx, *sac_layout(s, which), s->base(which), dst, n, dst_is_device);
}
extern "C" int ilsx_sac_set_params(ilsx_sac* s, int which, const float* src, size_t n, int src_is_device) {
  if (!s || !src || which < 0 || which > 4) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_set_params: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  return net_upload_flat(s->ctx, *sac_layout(s, which), s->base(which), src, n, src_is_device);
}
extern "C" int ilsx_sac_get_grads(ilsx_sac* s, int which, float* dst, size_t n, int dst_is_device) {
  if (!s || !dst || which < 0 || which > 2) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_get_grads: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  return net_download_flat(s->ctx, *sac_layout(s, which), s->gbase(which), dst, n, dst_is_device);
}
extern "C" int ilsx_sac_grads_ptr(ilsx_sac* s, int segment, float** dev_ptr, size_t* n) {
  if (!s || !dev_ptr || !n || segment < 0 || segment > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_grads_ptr: bad argument");
  if (segment == 0) { *dev_ptr = s->G; *n = 2 * s->nq; }
  else { *dev_ptr = s->G + 2 * s->nq; *n = s->np + 4; }
  return ILSX_OK;
}
extern "C" int ilsx_sac_get_log_alpha(ilsx_sac* s, double* out) {
  if (!s || !out) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  *out = h.log_alpha;
  return ILSX_OK;
}
extern "C" int ilsx_sac_set_log_alpha(ilsx_sac* s, double v) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  h.log_alpha = v;
  h.alpha = (float)std::exp(v);
  HIPCHK(hipMemcpyAsync(s->scal, &h, sizeof h, hipMemcpyHostToDevice, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  return ILSX_OK;
}
extern "C" int ilsx_sac_get_adam(ilsx_sac* s, int which, float* m_host, float* v_host, size_t n, int64_t* t) {
  if (!s || !m_host || !v_host || which < 0 || which > 2) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_get_adam: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t off = s->trainable_off(which);
  ILSX_TRY(net_download_flat(s->ctx, *sac_layout(s, which), s->M + off, m_host, n, 0));
  ILSX_TRY(net_download_flat(s->ctx, *sac_layout(s, which), s->V + off, v_host, n, 0));
  if (t) {
    DevScalars h;
    HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    *t = which == W_PI ? h.t_pi : h.t_q;
  }
  return ILSX_OK;
}
extern "C" int ilsx_sac_set_adam(ilsx_sac* s, int which, const float* m_host, const float* v_host, size_t n, int64_t t) {
  if (!s || !m_host || !v_host || which < 0 || which > 2) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_set_adam: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t off = s->trainable_off(which);
  ILSX_TRY(net_upload_flat(s->ctx, *sac_layout(s, which), s->M + off, m_host, n, 0));
  ILSX_TRY(net_upload_flat(s->ctx, *sac_layout(s, which), s->V + off, v_host, n, 0));
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  if (which == W_PI) h.t_pi = (int)t; else h.t_q = (int)t;
  HIPCHK(hipMemcpyAsync(s->scal, &h, sizeof h, hipMemcpyHostToDevice, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  return sac_refresh_adam(s);
}
extern "C" int ilsx_sac_get_alpha_opt(ilsx_sac* s, double* m, double* v, int64_t* t, uint64_t* rng_step) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  if (m) *m = h.m_alpha;
  if (v) *v = h.v_alpha;
  if (t) *t = h.t_alpha;
  if (rng_step) *rng_step = h.step;
  return ILSX_OK;
}
extern "C" int ilsx_sac_set_alpha_opt(ilsx_sac* s, double m, double v, int64_t t, uint64_t rng_step) {
  if (!s) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  DevScalars h;
  HIPCHK(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  h.m_alpha = m; h.v_alpha = v; h.t_alpha = (int)t; h.step = rng_step; h.gather_step = rng_step;
  HIPCHK(hipMemcpyAsync(s->scal, &h, sizeof h, hipMemcpyHostToDevice, s->ctx->stream));
  HIPCHK(hipStreamSynchronize(s->ctx->stream));
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------------ parity aids
// The fused path draws its rows inside k_mlp2_fwd_split (GatherSpec) and its noise inside the policy-finish prologue.  These two
// entry points let a test rebuild exactly those inputs with INDEPENDENT kernels (k_replay_sample, k_debug_eps) and feed them to
// the oracle (tests/test_fused_parity.py).
__global__ __launch_bounds__(256) void k_debug_eps(uint64_t seed, uint64_t step, uint32_t stream, int B, int a, float* out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * a) return;
  const int r = e / a, j = e - r * a;
  float z4[4];
  philox_normal4(seed, step, stream, (uint32_t)r, (uint32_t)(j >> 2), z4);
  const int q = j & 3;
  out[e] = q == 0 ? z4[0] : q == 1 ? z4[1] : q == 2 ? z4[2] : z4[3];
}

// Known-answer aid: the raw Philox4x32-10 words and the N(0,1) values philox_normal4 makes of them, for explicit (seed, step, stream):
// raw[r*4 + i] = word i of the block with counter (r, quad 0); normals[r*a + j] = what every policy epilogue draws for (row r, dim j).
__global__ __launch_bounds__(256) void k_debug_philox_raw(uint64_t seed, uint64_t step, uint32_t stream, int n, uint32_t* out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  uint32_t c[4] = {(uint32_t)r, 0u, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
  out[4 * r + 0] = c[0]; out[4 * r + 1] = c[1]; out[4 * r + 2] = c[2]; out[4 * r + 3] = c[3];
}
extern "C" int ilsx_debug_philox(ilsx_ctx* ctx, uint64_t seed, uint64_t step, uint32_t stream, int n_rows, int a, uint32_t* raw,
                                 float* normals) {
  if (!ctx || n_rows < 1 || a < 1 || (!raw && !normals)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_debug_philox: bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (raw) hipLaunchKernelGGL(k_debug_philox_raw, dim3((n_rows + 255) / 256), dim3(256), 0, ctx->stream, seed, step, stream, n_rows, raw);
  if (normals) hipLaunchKernelGGL(k_debug_eps, dim3((n_rows * a + 255) / 256), dim3(256), 0, ctx->stream, seed, step, stream, n_rows, a, normals);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int disc_debug_stream(const void* d, uint32_t* stream, uint64_t* seed);   // ilsx_disc.hip
extern "C" int ilsx_debug_rng_stream(const void* object, int kind, uint32_t* stream, uint64_t* seed) {
  if (!object || !stream || kind < 0 || kind > 2) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_debug_rng_stream: bad argument");
  if (kind == 0) { const ilsx_replay* rb = (const ilsx_replay*)object; *stream = rb->rng_stream; if (seed) *seed = rb->seed; return ILSX_OK; }
  if (kind == 1) { const ilsx_sac* s = (const ilsx_sac*)object; *stream = s->rng_stream; if (seed) *seed = s->ctx->seed; return ILSX_OK; }
  return disc_debug_stream(object, stream, seed);
}

extern "C" int ilsx_sac_debug_batch(ilsx_sac* s, ilsx_replay* rb, uint64_t step, int B, float* obs, float* act, float* rew,
                                    float* done, float* nobs, float* eps_next, float* eps_cur, int64_t* idx) {
  if (!s || !rb || B < 1 || B > s->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_debug_batch: bad argument");
  if (rb->o != s->o || rb->a != s->a) ILSX_FAIL(ILSX_ERR_ARG, "replay dims do not match the agent");
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "replay buffer is empty");
  HIPCHK(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  const size_t o = s->o, a = s->a, b = B;
  // scratch for the outputs the caller did not ask for (the sample kernel writes all five keys)
  float* tmp = nullptr;
  ILSX_TRY(ctx_alloc(s->ctx, (b * (2 * o + a + 2)) * sizeof(float), (void**)&tmp, false));
  float* t_obs = tmp; float* t_act = t_obs + b * o; float* t_rew = t_act + b * a; float* t_done = t_rew + b; float* t_nobs = t_done + b;
  int rc = replay_launch_sample(rb, B, nullptr, nullptr, step, obs ? obs : t_obs, act ? act : t_act, rew ? rew : t_rew,
                                done ? done : t_done, nobs ? nobs : t_nobs, idx);
  if (rc == ILSX_OK) {
    const unsigned blocks = (unsigned)((b * a + 255) / 256);
    if (eps_next) hipLaunchKernelGGL(k_debug_eps, dim3(blocks), dim3(256), 0, st, s->ctx->seed, step, s->rng_stream, B, (int)a, eps_next);
    if (eps_cur) hipLaunchKernelGGL(k_debug_eps, dim3(blocks), dim3(256), 0, st, s->ctx->seed, step, s->rng_stream + 1, B, (int)a, eps_cur);
    if (hipGetLastError() != hipSuccess) rc = ILSX_ERR_HIP;
  }
  const int rf = ctx_free(s->ctx, tmp);   // synchronises the stream
  return rc != ILSX_OK ? rc : rf;
}

extern "C" int ilsx_sac_debug_last_batch(ilsx_sac* s, int B, float* obs, float* act, float* rew, float* done, float* nobs,
                                         float* eps_cur) {
  if (!s || B < 1 || B > s->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_debug_last_batch: bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  const SacWs& w = s->ws;
  const size_t o = s->o, a = s->a, b = B;
  if (obs) HIPCHK(hipMemcpyAsync(obs, w.s, b * o * 4, hipMemcpyDeviceToDevice, st));
  if (act) HIPCHK(hipMemcpyAsync(act, w.a, b * a * 4, hipMemcpyDeviceToDevice, st));
  if (rew) HIPCHK(hipMemcpyAsync(rew, w.r, b * 4, hipMemcpyDeviceToDevice, st));
  if (done) HIPCHK(hipMemcpyAsync(done, w.d, b * 4, hipMemcpyDeviceToDevice, st));
  if (nobs) HIPCHK(hipMemcpyAsync(nobs, w.s2, b * o * 4, hipMemcpyDeviceToDevice, st));
  if (eps_cur) HIPCHK(hipMemcpyAsync(eps_cur, w.epss, b * a * 4, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return ILSX_OK;
}

// ================================================================================================ grouped agents
// "within a GPU, the co-resident seeds are batched as grouped GEMMs (weights differ per seed)".  K independent
// agents of identical shape step in lock-step: every stage of the fused step is ONE launch whose grid carries all agents'
// tasks (descriptor tables in device memory, built once), so the step costs 9 dependent launches for K agents instead of
// 9K, and each launch brings K times the workgroups to hide the dependent-load latency that bounds the single step.
struct ilsx_sac_group {
  ilsx_ctx* ctx = nullptr;
  std::vector<ilsx_sac*> agents;
  std::vector<ilsx_replay*> rbs;
  int B = 0;
  hipGraphExec_t graph = nullptr;
  struct Stage {
    int kind = 0, KP = 0, ntasks = 0;
    FwdArgs f; BwdArgs b; DwArgs d;
    void *tasks = nullptr, *gtiles = nullptr, *tails = nullptr;
    int cslot = -1;   // first slot of this stage's records in the constant-memory table (kernels.h g_fwd_tab / g_bwd_tab), -1 = none
  };
  std::vector<Stage> stages;
  TailLite* tails_lite = nullptr;   // deferred tail: one record per agent (see TailLite, kernels.h)
  // agents living in contexts with a stream of their own (one per co-resident run: their rollouts overlap): the grouped launches go on
  // ctx->stream, fenced against every such stream at both ends of a train call (group_fence_in / group_fence_out)
  std::vector<hipStream_t> peer_streams; std::vector<hipEvent_t> peer_events; hipEvent_t done_event = nullptr;
  bool defer = false;
  int mt = 1;                       // 16-row tiles per workgroup of the forward / backward launches (macro tiles, fwd_split_tile.inc)
  int late = -1;                    // "late weights" launch shape of the forward / backward launches: -1 per launch by size, 0 / 1 pinned (ILSX_GRP_LATE)
};

template <class T>
static int upload_table(ilsx_ctx* ctx, const std::vector<T>& v, void** dev) {
  ILSX_TRY(ctx_alloc(ctx, v.size() * sizeof(T), dev, false));
  HIPCHK(hipMemcpyAsync(*dev, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return ILSX_OK;
}

static void group_release_tables(ilsx_sac_group* g) {
  for (auto& st : g->stages) {
    for (void* p : {st.tasks, st.gtiles, st.tails})
      if (p) ctx_free(g->ctx, p);
    if (st.cslot >= 0) grp_const_free(g->ctx->device, st.kind == 0, st.cslot);
  }
  g->stages.clear();
  if (g->graph) { hipGraphExecDestroy(g->graph); g->graph = nullptr; }
}

static int group_pick_mt(const ilsx_sac_group* g, int B);
// collect every agent's launch list for one fused step on (rbs, B) and merge stage by stage
static int group_build(ilsx_sac_group* g, ilsx_replay* const* rbs, int B) {
  const int K = (int)g->agents.size();
  std::vector<SacCollector> cols(K);
  for (int k = 0; k < K; ++k) {
    ilsx_sac* s = g->agents[k];
    s->B = B; s->eps_explicit = false; s->col = &cols[k]; s->gather_rb = rbs[k];
    const int rc = sac_full_step(s);
    s->col = nullptr; s->gather_rb = nullptr;
    if (rc != ILSX_OK) return rc;
    if (cols[k].L.size() != cols[0].L.size()) ILSX_FAIL(ILSX_ERR_STATE, "agents of a group must produce the same launch sequence");
  }
  group_release_tables(g);
  const char* ce = getenv("ILSX_GRP_CONST");   // 0: descriptor records from the device-memory tables only (A/B)
  const bool use_const = !(ce && atoi(ce) == 0);
  const size_t nst = cols[0].L.size();
  g->stages.resize(nst);
  for (size_t i = 0; i < nst; ++i) {
    ilsx_sac_group::Stage& st = g->stages[i];
    st.kind = cols[0].L[i].kind;
    for (int k = 0; k < K; ++k)
      if (cols[k].L[i].kind != st.kind) ILSX_FAIL(ILSX_ERR_STATE, "agents of a group must produce the same launch sequence");
    if (st.kind == 0) {
      std::vector<FwdTaskG> tasks;
      st.f = cols[0].L[i].f; st.KP = cols[0].L[i].KP;
      for (int k = 0; k < K; ++k) {
        const FwdArgs& A = cols[k].L[i].f;
        FwdGroup G; memset(&G, 0, sizeof G);
        G.fin = A.fin; G.gather = A.gather; G.scal = A.scal; G.fin_on = A.fin_on;
        for (int t = 0; t < A.ntasks; ++t) {
          FwdTaskG R; memset(&R, 0, sizeof R);
          R.t = A.t[t]; R.t.agent = k; R.t.first = t == 0; R.g = G;
          tasks.push_back(R);
        }
      }
      st.ntasks = (int)tasks.size();
      ILSX_TRY(upload_table(g->ctx, tasks, &st.tasks));
      st.f.tasks = (const FwdTaskG*)st.tasks; st.f.ntasks = st.ntasks;
      if (use_const && grp_const_alloc(g->ctx->device, true, st.ntasks, &st.cslot) == 0) {   // the same records in constant memory (kernels.h GRP == 2)
        ILSX_TRY(grp_const_upload(g->ctx, true, st.cslot, tasks.data(), tasks.size()));
        st.f.ctab = st.cslot + 1;
      }
    } else if (st.kind == 1) {
      std::vector<BwdTask> tasks;
      st.b = cols[0].L[i].b;
      for (int k = 0; k < K; ++k) {
        const BwdArgs& A = cols[k].L[i].b;
        if (A.inv_B != st.b.inv_B || A.gamma != st.b.gamma || A.reward_scale != st.b.reward_scale || A.w_mu != st.b.w_mu ||
            A.w_std != st.b.w_std)
          ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "agents of a group must share discount / reward_scale / regulariser weights");
        for (int t = 0; t < A.ntasks; ++t) { BwdTask T = A.t[t]; T.scal = A.scal; tasks.push_back(T); }
      }
      st.ntasks = (int)tasks.size();
      ILSX_TRY(upload_table(g->ctx, tasks, &st.tasks));
      st.b.tasks = (const BwdTask*)st.tasks; st.b.ntasks = st.ntasks;
      if (use_const && grp_const_alloc(g->ctx->device, false, st.ntasks, &st.cslot) == 0) {
        ILSX_TRY(grp_const_upload(g->ctx, false, st.cslot, tasks.data(), tasks.size()));
        st.b.ctab = st.cslot + 1;
      }
    } else if (st.kind == 2) {
      std::vector<DwTileG> gt;
      st.d = cols[0].L[i].d;
      int tiles = 0, nmat = 0;
      // tile shape: as for single launches (launch_bwd_dw) the smallest that keeps the launch within ~3 workgroups per CU — it only
      // matters for small groups (K = 1, 2: +4 %); at K >= 4 the launch is bound elsewhere and every shape measures the same
      const char* gse = getenv("ILSX_DW_TILE_GRP");   // "NH KT" digits (24, 14, 12, 11), unset / 0 = by size; read per group build
      const int gshape = gse ? atoi(gse) : 0;
      int gnh = 2, gkt = 4;
      // ILSX_DW_GRP_STRIP = 1: the strip shape (k_dw_strip: one wavefront per 16 x 64 output strip, no cross-wave reduction).  Measured
      // (K = 8 Hopper runs): 42 us per launch against 23 us for the 8-wave tiles — one wave walking 8 row-eighths and 8 optimiser
      // epilogues in sequence is a longer dependent chain than eight waves and one LDS reduction; off by default, bit-identical, under test
      const char* se = getenv("ILSX_DW_GRP_STRIP");
      const bool strip = se ? atoi(se) != 0 : false;
      if (strip) { gnh = 1; gkt = 4; }
      else if (gshape) { gnh = gshape / 10; gkt = gshape % 10; }
      else {
        auto count = [&](int nh, int kt) {
          long long n = 0;
          for (int k = 0; k < K; ++k) {
            const DwArgs& D = cols[k].L[i].d;
            for (int m = 0; m < D.nmat; ++m) n += (long long)((D.m[m].NA + 16 * nh - 1) / (16 * nh)) * ((D.m[m].NB + 16 * kt - 1) / (16 * kt));
          }
          return n;
        };
        const long long cap = 3LL * device_cus(g->ctx);
        if (count(1, 1) <= cap) { gnh = 1; gkt = 1; }
        else if (count(1, 2) <= cap) { gnh = 1; gkt = 2; }
      }
      st.d.tile_nh = gnh; st.d.tile_kt = gkt; st.d.strip = strip ? 1 : 0;
      for (int k = 0; k < K; ++k) {
        const DwArgs& D = cols[k].L[i].d;
        for (int m = 0; m < D.nmat; ++m, ++nmat) {
          DwTileG R; memset(&R, 0, sizeof R);
          R.J = D.m[m]; R.F = cols[k].L[i].F;
          R.J.ktiles = (R.J.NB + 16 * gkt - 1) / (16 * gkt);
          const int n = ((R.J.NA + 16 * gnh - 1) / (16 * gnh)) * R.J.ktiles;
          R.J.tile0 = tiles; R.J.agent = k;
          for (int t = 0; t < n; ++t) gt.push_back(R);
          tiles += n;
        }
      }
      st.d.ntiles = tiles; st.d.nmat = nmat;
      ILSX_TRY(upload_table(g->ctx, gt, &st.gtiles));
      st.d.gtiles = (const DwTileG*)st.gtiles;
      st.d.g_lo = nullptr;   // never the row-split path (B is the SAC batch)
    } else {
      std::vector<SacTailItem> items;
      for (int k = 0; k < K; ++k) items.push_back(cols[k].L[i].t);
      ILSX_TRY(upload_table(g->ctx, items, &st.tails));
    }
  }
  {
    std::vector<TailLite> lite;
    for (int k = 0; k < K; ++k) lite.push_back(sac_tail_lite(g->agents[k], B));
    if (g->tails_lite) ctx_free(g->ctx, g->tails_lite);
    g->tails_lite = nullptr;
    ILSX_TRY(upload_table(g->ctx, lite, (void**)&g->tails_lite));
  }
  HIPCHK(hipStreamSynchronize(g->ctx->stream));
  g->rbs.assign(rbs, rbs + K);
  g->B = B;
  g->mt = group_pick_mt(g, B);
  // ILSX_GRP_LATE = 0 | 1 pins the "late weights" launch shape off / on (kernels.h GRP == 3); by default each launch decides from its size
  if (const char* e = getenv("ILSX_GRP_LATE")) g->late = atoi(e) ? 1 : 0; else g->late = -1;
  return ILSX_OK;
}

static int group_launch_tail(ilsx_sac_group* g, const void* tails, int deferred) {
  ProfScope ps(g->ctx, ILSX_K_SAC_FINISH);
  ILSX_LAUNCH(ps, k_sac_tail_group, dim3((unsigned)g->agents.size()), dim3(256), 0, g->ctx->stream, (const SacTailItem*)tails, deferred);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
// Row tiles per workgroup of the grouped forward / backward launches (macro tiles, fwd_split_tile.inc MT): ILSX_GRP_MT = 2 | 4 selects
// them, the default is ONE.  Measured on MI355X (tools/grp_sweep.sh, profiles/r04_grp_sweep.txt; K = 8 Hopper runs, forward / backward
// launch averages): 1 row tile 29.8 / 25.3 us, 2: 35.9 / 29.2, 4: 45.7 / 51.0 — sharing a wave's weight fragments over more rows does
// not pay, because these kernels are not bound by fetching weights but by the instructions a wave issues PER ROW TILE (gather, Philox,
// head epilogues, activation stores: ~1.5k of the ~2.6k a 16-row workgroup executes) at one or two waves per SIMD; a macro tile keeps
// that per-row count, halves or quarters the workgroups in flight and (4 tiles: 105 KB of LDS) the waves per SIMD.  Kept, bit-identical
// and under test (test_sac_group_lockstep_is_bitwise_the_independent_runs), as the measured answer to VERDICT r3 item 1.
static int group_pick_mt(const ilsx_sac_group* g, int B) {
  const ilsx_sac* s0 = g->agents[0];
  (void)B;
  if (!(s0->Lq.cfg.hidden == 256 && s0->cs == 4)) return 1;
  if (const char* e = getenv("ILSX_GRP_MT")) { const int v = atoi(e); return (v == 2 || v == 4) ? v : 1; }
  return 1;
}
static int group_launch_step(ilsx_sac_group* g) {
  ilsx_sac* s0 = g->agents[0];
  const int H = s0->Lq.cfg.hidden, act = s0->Lq.cfg.act, cs = s0->cs;
  int nfwd = 0;
  const int K = (int)g->agents.size();
  for (auto& st : g->stages) {
    if (st.kind == 0) {
      FwdArgs A = st.f;
      if (g->defer && nfwd < 2) { A.tail = g->tails_lite; A.tail_mode = nfwd + 1; A.tail_n = K; }   // tail of the previous step / gather_step
      ++nfwd;
      A.mt = g->mt; A.swz.agents = K; A.mt_a = s0->a; A.late = g->late;
      A.mt_not = std::max((s0->Lp.NO + 15) / 16, (s0->Lq.NO + 15) / 16);
      ILSX_TRY(launch_fwd(g->ctx, A, H, act, st.KP, cs));
    }
    else if (st.kind == 1) { BwdArgs A = st.b; A.mt = g->mt; A.swz.agents = K; A.late = g->late; ILSX_TRY(launch_bwd_dx(g->ctx, A, H, act, cs)); }
    else if (st.kind == 2) { AdamFuse on; memset(&on, 0, sizeof on); on.on = 1; ILSX_TRY(launch_bwd_dw(g->ctx, st.d, g->B, &on)); }
    else if (!g->defer) ILSX_TRY(group_launch_tail(g, st.tails, 0));
  }
  return ILSX_OK;
}

extern "C" int ilsx_sac_group_create(ilsx_ctx* ctx, ilsx_sac* const* agents, int n_agents, ilsx_sac_group** out) {
  if (!ctx || !agents || !out || n_agents < 1 || n_agents > 64) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_group_create: bad argument");
  for (int k = 0; k < n_agents; ++k) {
    ilsx_sac* s = agents[k];
    // the group's launches go on ctx->stream; an agent's own objects (env, replay, rollout inference) are ordered with them as long as its
    // ctx enqueues on the same stream — the same ctx, or a sibling created on it (per-run Philox key + stream ids, include/ilsx.h)
    if (!s || s->ctx->device != ctx->device)
      ILSX_FAIL(ILSX_ERR_ARG, "every agent of a group must live on the group's device");
    if (s->cs <= 1) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "grouped steps use the column-split kernels (2 hidden layers of 128 or 256)");
    if (s->cfg.grad_world != 1) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "grouped agents are whole runs (grad_world == 1)");
    if (memcmp(&s->Lq.cfg, &agents[0]->Lq.cfg, sizeof s->Lq.cfg) || memcmp(&s->Lp.cfg, &agents[0]->Lp.cfg, sizeof s->Lp.cfg) ||
        s->cfg.max_batch != agents[0]->cfg.max_batch)
      ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "agents of a group must have identical network shapes and max_batch");
  }
  ilsx_sac_group* g = new ilsx_sac_group();
  g->ctx = ctx;
  g->agents.assign(agents, agents + n_agents);
  HIPCHK(hipSetDevice(ctx->device));
  for (int k = 0; k < n_agents; ++k) {
    hipStream_t st = agents[k]->ctx->stream;
    if (st == ctx->stream || std::find(g->peer_streams.begin(), g->peer_streams.end(), st) != g->peer_streams.end()) continue;
    hipEvent_t ev = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    g->peer_streams.push_back(st); g->peer_events.push_back(ev);
  }
  if (!g->peer_streams.empty()) HIPCHK(hipEventCreateWithFlags(&g->done_event, hipEventDisableTiming));
  *out = g;
  return ILSX_OK;
}
// everything the agents' own streams have enqueued (rollout records in the rings, parameter uploads) happens before the grouped launches ...
// HOST fences by default (ILSX_GROUP_FENCE unset / 2): wait for the peer streams before the first grouped launch, for the group stream after
// the last.  Event fences (ILSX_GROUP_FENCE=1: hipEventRecord on each peer + hipStreamWaitEvent on the group stream, and the reverse at the
// end — no host wait) were measured and rejected: a stream that has waited on another stream's event runs every LATER graph launch slower on
// this stack — K = 10 Hopper runs, B = 512, 1000 steps per call: 275 us per lock-step with host fences or one shared stream, 284-320 us with
// event fences (tools/grp_streams_ab.py, profiles/r06_grp_streams.txt).  The lock-step loop waits for every stream around a train call anyway.
static int group_fence_mode() { static const int m = []() { const char* e = getenv("ILSX_GROUP_FENCE"); return e ? atoi(e) : 2; }(); return m; }
static int group_fence_in(ilsx_sac_group* g) {
  if (group_fence_mode() == 0) return ILSX_OK;
  if (group_fence_mode() == 2) { for (hipStream_t st : g->peer_streams) HIPCHK(hipStreamSynchronize(st)); return ILSX_OK; }
  for (size_t i = 0; i < g->peer_streams.size(); ++i) {
    HIPCHK(hipEventRecord(g->peer_events[i], g->peer_streams[i]));
    HIPCHK(hipStreamWaitEvent(g->ctx->stream, g->peer_events[i], 0));
  }
  return ILSX_OK;
}
// ... and whatever they enqueue next (rollout inference on the updated policy) happens after them
static int group_fence_out(ilsx_sac_group* g) {
  if (g->peer_streams.empty() || group_fence_mode() == 0) return ILSX_OK;
  if (group_fence_mode() == 2) { HIPCHK(hipStreamSynchronize(g->ctx->stream)); return ILSX_OK; }
  HIPCHK(hipEventRecord(g->done_event, g->ctx->stream));
  for (hipStream_t st : g->peer_streams) HIPCHK(hipStreamWaitEvent(st, g->done_event, 0));
  return ILSX_OK;
}

extern "C" int ilsx_sac_group_destroy(ilsx_sac_group* g) {
  if (!g) return ILSX_OK;
  hipSetDevice(g->ctx->device);
  hipStreamSynchronize(g->ctx->stream);
  group_release_tables(g);
  if (g->tails_lite) ctx_free(g->ctx, g->tails_lite);
  for (hipEvent_t ev : g->peer_events) hipEventDestroy(ev);
  if (g->done_event) hipEventDestroy(g->done_event);
  delete g;
  return ILSX_OK;
}

// n_steps lock-step gradient steps of every agent, agent k sampling from rbs[k] (TorchRLAlgorithm._do_training of K
// independent runs).  want_stats: the first step also computes every agent's statistics (read with ilsx_sac_last_stats).
extern "C" int ilsx_sac_group_train_from_replay(ilsx_sac_group* g, ilsx_replay* const* rbs, int n_steps, int B, int want_stats) {
  if (!g || !rbs || n_steps < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_sac_group_train_from_replay: bad argument");
  const int K = (int)g->agents.size();
  HIPCHK(hipSetDevice(g->ctx->device));
  for (int k = 0; k < K; ++k) {
    ilsx_sac* s = g->agents[k];
    if (!rbs[k] || rbs[k]->o != s->o || rbs[k]->a != s->a) ILSX_FAIL(ILSX_ERR_ARG, "replay %d does not match its agent", k);
    if (rbs[k]->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "replay buffer %d is empty", k);
  }
  if (B < 1 || B > g->agents[0]->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch", B);
  if (want_stats && n_steps > 1) {   // statistics describe the FIRST batch of the call (sac_alpha.py:185-190), as in ilsx_sac_train_from_replay
    ILSX_TRY(ilsx_sac_group_train_from_replay(g, rbs, 1, B, 1));
    return ilsx_sac_group_train_from_replay(g, rbs, n_steps - 1, B, 0);
  }
  bool rebuild = g->stages.empty() || g->B != B || (int)g->rbs.size() != K;
  for (int k = 0; k < K && !rebuild; ++k) rebuild = g->rbs[k] != rbs[k];
  for (int k = 0; k < K; ++k) ILSX_TRY(replay_flush_state(rbs[k]));   // on each ring's own stream, ahead of the fence
  ILSX_TRY(group_fence_in(g));
  struct FenceOut { ilsx_sac_group* g; ~FenceOut() { group_fence_out(g); } } fence_out{g};   // on every path out of this call
  if (rebuild) ILSX_TRY(group_build(g, rbs, B));
  hipStream_t st = g->ctx->stream;
  static const bool no_graph = getenv("ILSX_NO_GRAPH") != nullptr;
  static const bool no_defer = getenv("ILSX_NO_DEFER_TAIL") != nullptr;
  const bool use_graph = !(no_graph || g->ctx->prof_on);
  if (n_steps == 0) return ILSX_OK;
  if (g->defer != !no_defer && g->graph) { hipGraphExecDestroy(g->graph); g->graph = nullptr; }
  g->defer = !no_defer;   // the last step's tail is flushed below: nothing is pending when the call returns
  if (use_graph && !g->graph) {
    hipGraph_t gr = nullptr;
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = group_launch_step(g);
    hipError_t e = hipStreamEndCapture(st, &gr);
    if (rc != ILSX_OK) { if (gr) hipGraphDestroy(gr); return rc; }
    if (e != hipSuccess) ILSX_FAIL(ILSX_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&g->graph, gr, nullptr, nullptr, 0);
    hipGraphDestroy(gr);
    if (e != hipSuccess) { g->graph = nullptr; ILSX_FAIL(ILSX_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
  }
  for (int i = 0; i < n_steps; ++i) {
    if (want_stats && i == n_steps - 1)
      for (int k = 0; k < K; ++k) ILSX_TRY(sac_request_stats(g->agents[k], st));
    if (use_graph) HIPCHK(hipGraphLaunch(g->graph, st));
    else ILSX_TRY(group_launch_step(g));
  }
  if (g->defer)
    for (auto& stg : g->stages)
      if (stg.kind == 3) ILSX_TRY(group_launch_tail(g, stg.tails, 1));
  if (want_stats) {   // (n_steps == 1 here) keep every agent's statistics as they stand now: the rest of the call moves alpha on
    ILSX_TRY(group_fence_out(g));   // sac_read_stats copies on the agent's own stream
    for (int k = 0; k < K; ++k) { ilsx_sac_stats tmp; ILSX_TRY(sac_read_stats(g->agents[k], &tmp)); }
  }
  return ILSX_OK;
}
