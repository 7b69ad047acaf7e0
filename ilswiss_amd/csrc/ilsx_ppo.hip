// ilsx_ppo.hip — PPO: rlkit/torch/algorithms/ppo/ppo.py:57-100 (calc_adv: per-trajectory GAE with zero
// bootstrap + per-trajectory advantage standardisation with the unbiased std) and :102-170 (train_step:
// update_epoch x shuffled minibatches; value MSE + L2 over all vf parameters; clipped surrogate; grad-norm clip
// 20), policy = ReparamMultivariateGaussianPolicy(conditioned_std=False) (policies.py:348-478: tanh-hidden mean
// network + state-independent action_log_std parameter), value net = tanh FlattenMlp (ppo_exp_script.py:82-96).
//
// Device pipeline for one train_step over N on-policy samples (T trajectories):
//   forward(vf) over all N rows -> k_ppo_gae (one lane per trajectory: reverse scan, mean/std, normalise)
//   forward(pi) over all N rows with the Gaussian log-prob head -> fixed log-probs
//   per minibatch (rows gathered through an index list, nothing is copied) — five launches (round 6; ten before):
//     forward {value ; policy with the log-prob head} ; backward {LOSS_MSE head ; LOSS_PPO_POLICY head, which also emits d/d log_std rows} ;
//     weight gradients of both nets (one table) ; k_ppo_norm (sum of squares of the whole policy gradient, log_std column sums) ;
//     k_ppo_update (clip_grad_norm_(20) scale + Adam over the policy arena, Adam + L2 over the value arena, step counters)
// HBM-bound pieces: k_ppo_gae streams 5 fp32 arrays once (20 B per sample).
#include <cmath>

#include "host_common.h"

struct PpoScalars {
  float v_step, v_bc2s, p_step, p_bc2s;
  float vf_loss, pg_loss, grad_norm, pad;
  int t_v, t_p, pad2[2];
};

// ---- GAE: one wavefront per trajectory (ppo.py:73-86).  The trajectory is walked from its end in 64-sample chunks:
// every lane loads one sample (coalesced 256-B rows of values / rewards); delta_t = r_t + g*V_{t+1} - V_t takes V_{t+1}
// from the neighbouring lane; the recurrence A_t = delta_t + (g*l)*A_{t+1} is a 6-step Kogge-Stone suffix scan with the
// constant ratio g*l (a lane-serial walk in the reference's exact order costs 64 dependent VALU steps per chunk with all
// lanes computing the same scalar and ran 5x slower); chunks chain through (A, V) of their first sample.  The raw
// advantages stay in registers (up to 1024 samples = max_path_length of every config; longer trajectories park them
// in `adv`), then mean / unbiased std and the normalised write.  HBM traffic: values + rewards in, returns + advantages
// out = 16 B per sample.  Differences to the sequential fp32 order are a few ulp (tests: rtol 1e-5 on returns).
#define GAE_KEEP 16
__global__ __launch_bounds__(256) void k_ppo_gae(const float* __restrict__ values, const float* __restrict__ rewards,
                                                 const int* __restrict__ offs, const float* __restrict__ boot, int n_traj,
                                                 float reward_scale, float gamma, float tau, float* __restrict__ returns,
                                                 float* __restrict__ adv) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_traj) return;   // wave-uniform
  const int b = offs[t], e = offs[t + 1], T = e - b;
  const int nchunk = (T + 63) >> 6;
  const float c = gamma * tau;
  float cp[7];   // c^(2^s)
  cp[0] = c;
#pragma unroll
  for (int s = 1; s < 7; ++s) cp[s] = cp[s - 1] * cp[s - 1];
  float keep[GAE_KEEP];
#pragma unroll
  for (int k = 0; k < GAE_KEEP; ++k) keep[k] = 0.0f;
  float carry_v = boot ? boot[t] : 0.0f, carry_a = 0.0f, sum = 0.0f;   // V after the last sample: 0 = the reference (ppo.py:74)
  for (int ch = nchunk - 1, k = 0; ch >= 0; --ch, ++k) {
    const int i0 = b + 64 * ch, idx = i0 + lane;
    const int last = min(63, e - 1 - i0);
    const bool ok = lane <= last;
    const float v = ok ? values[idx] : 0.0f;
    const float r = ok ? reward_scale * rewards[idx] : 0.0f;
    float vn = __shfl_down(v, 1);
    if (lane == last) vn = carry_v;
    float a = ok ? r + gamma * vn - v : 0.0f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const float up = __shfl_down(a, 1 << s);
      if (lane + (1 << s) <= last) a = a + cp[s] * up;
    }
    // carry from the later chunk: c^(last - lane + 1) * A_first(later chunk)
    const int m = last - lane + 1;
    float pw = 1.0f;
#pragma unroll
    for (int s = 0; s < 7; ++s) if (m & (1 << s)) pw *= cp[s];
    if (ok) a = a + pw * carry_a;
    carry_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
    carry_v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
    if (ok) { returns[idx] = v + a; sum += a; }
    bool kept = false;
#pragma unroll
    for (int q = 0; q < GAE_KEEP; ++q)
      if (q == k) { keep[q] = a; kept = true; }
    if (!kept && ok) adv[idx] = a;   // long trajectory: park the raw advantage, normalised in place below
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)T;
  double ss = 0.0;
  for (int ch = nchunk - 1, k = 0; ch >= 0; --ch, ++k) {
    const int idx = b + 64 * ch + lane;
    float a = 0.0f;
#pragma unroll
    for (int q = 0; q < GAE_KEEP; ++q) if (q == k) a = keep[q];
    if (k >= GAE_KEEP && idx < e) a = adv[idx];
    const double dd = idx < e ? (double)a - (double)mean : 0.0;
    ss += dd * dd;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  float sd = (float)sqrt(ss / (double)(T - 1));   // torch.std(): unbiased; T == 1 -> nan like the reference
  // segment mode (bootstrap values given; not a reference code path): a rollout boundary can leave a one-sample segment,
  // whose standardised advantage is undefined -> 0 (no policy gradient from it) instead of poisoning the minibatch
  if (boot && T == 1) sd = INFINITY;
  for (int ch = nchunk - 1, k = 0; ch >= 0; --ch, ++k) {
    const int idx = b + 64 * ch + lane;
    float a = 0.0f;
#pragma unroll
    for (int q = 0; q < GAE_KEEP; ++q) if (q == k) a = keep[q];
    if (k >= GAE_KEEP && idx < e) a = adv[idx];
    if (idx < e) adv[idx] = (a - mean) / sd;
  }
}

// ---- sum of squares of the policy gradient (every parameter once) + d/d log_std = column sums of the aux rows
#define PPO_MAX_A 32
struct PpoNormArgs {
  const float* G; int n;
  int skip0[2], skip1[2];                    // [skip0, skip1): second packings of hidden->hidden matrices (not parameters)
  const float* aux; int rows, a;
  float* partial;                            // [gridDim.x] sum of squares of this block's share of the mean-net gradient
  float* partial_ls;                         // [gridDim.x][a] this block's share of the log_std gradient (column sums of aux)
};
__global__ __launch_bounds__(256) void k_ppo_norm(const PpoNormArgs P) {
  __shared__ float sh[4];
  float s = 0.0f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P.n; i += gridDim.x * 256)
    if ((i < P.skip0[0] || i >= P.skip1[0]) && (i < P.skip0[1] || i >= P.skip1[1])) { const float g = P.G[i]; s += g * g; }
  s = block256_sum(s, sh);
  if (threadIdx.x == 0) P.partial[blockIdx.x] = s;
  // log_std gradient: rows strided over (block, thread), summed per column in a fixed order
  for (int j = 0; j < P.a; ++j) {
    float c = 0.0f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < P.rows; r += gridDim.x * 256) c += P.aux[(size_t)r * P.a + j];
    c = block256_sum(c, sh);
    if (threadIdx.x == 0) P.partial_ls[blockIdx.x * P.a + j] = c;
  }
}

// torch.randperm stand-in (ppo.py:116) without a host shuffle: a keyed 4-round Feistel network over the 2^k >= N index
// space with cycle-walking is a bijection of [0, N); one thread per output slot.
__device__ __forceinline__ uint32_t feistel_perm(uint32_t x, int half_bits, uint32_t k0, uint32_t k1) {
  const uint32_t mask = (1u << half_bits) - 1u;
  uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    uint32_t c[4] = {r, (uint32_t)rd, k1, 0x9E3779B9u};
    philox4x32_10(c, k0, k1 ^ (uint32_t)rd);
    const uint32_t nl = r, nr = l ^ (c[0] & mask);
    l = nl; r = nr;
  }
  return (l << half_bits) | r;
}
__global__ __launch_bounds__(256) void k_ppo_perm(int* __restrict__ perm, int n, int half_bits, uint32_t k0, uint32_t k1) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t x = (uint32_t)i;
  do { x = feistel_perm(x, half_bits, k0, k1); } while (x >= (uint32_t)n);   // cycle-walk back into [0, n)
  perm[i] = (int)x;
}

// The optimiser launch of a minibatch: clip_grad_norm_(20) + Adam over the policy arena (ppo.py:166-170), Adam + the L2 term over the value arena
// (ppo.py:145-153: the two updates of a minibatch share nothing — the policy loss reads the advantages, not the value net — so they run side by side),
// and the step counters.  The bias-correction scalars are computed here, per block, from the host's step counts (double pow, as k_ppo_refresh did):
// no scalar launch between the minibatches.  Value update: adam_apply's expressions (kernels.h) in its order, element-wise (both packings of a
// hidden -> hidden matrix hold the same gradient and parameter, hence stay equal).
struct PpoUpdateArgs {
  float* P; float* G; float* M; float* V; int n;   // whole policy arena incl. log_std (and both W1 packings)
  int n_mean, a;                                    // log_std lives at [n_mean, n_mean + a)
  const float* partial; const float* partial_ls; int nparts;
  float max_norm, b1, b2, eps;
  PpoScalars* sc;
  int ls_from_G;   // split run: the log_std gradient sits in G already (k_ppo_ls_fold, summed over the ranks); single run: built here from the column sums
  float* Pv; const float* Gv; float* Mv; float* Vv; int nv;   // value arena
  float l2x2, v_lr, p_lr;
  int t_v, t_p;    // optimiser steps taken BEFORE this one (the host's counts; sc->t_v / t_p are written for the snapshot getters)
  int nblk_p;      // blocks [0, nblk_p): policy ; the rest: value
};
__global__ __launch_bounds__(256) void k_ppo_update(const PpoUpdateArgs A) {
  __shared__ float s_coef, s_step, s_bc2s;
  __shared__ float s_gls[PPO_MAX_A];
  const bool policy = (int)blockIdx.x < A.nblk_p;
  if (threadIdx.x == 0) {
    const double b1 = 0.9, b2 = 0.999;
    const int t = policy ? A.t_p : A.t_v;
    s_step = (float)((double)(policy ? A.p_lr : A.v_lr) / (1.0 - pow(b1, (double)(t + 1))));
    s_bc2s = (float)sqrt(1.0 - pow(b2, (double)(t + 1)));
  }
  if (!policy) {
    __syncthreads();
    const float step = s_step, bc2s = s_bc2s;
    const int nb = (int)gridDim.x - A.nblk_p, b0 = (int)blockIdx.x - A.nblk_p;
    for (int i = b0 * 256 + threadIdx.x; i < A.nv; i += nb * 256) {
      const float p0 = A.Pv[i];
      float g = A.Gv[i];
      g = g + A.l2x2 * p0;
      const float m = A.Mv[i] * A.b1 + (1.0f - A.b1) * g;
      const float v = A.Vv[i] * A.b2 + (1.0f - A.b2) * g * g;
      A.Mv[i] = m; A.Vv[i] = v;
      A.Pv[i] = p0 - step * (m / (sqrtf(v) / bc2s + A.eps));
    }
    if (b0 == 0 && threadIdx.x == 0) { A.sc->t_v = A.t_v + 1; A.sc->v_step = step; A.sc->v_bc2s = bc2s; }
    return;
  }
  if (threadIdx.x < A.a) {   // every block rebuilds the log_std gradient from the per-block column sums (fixed order)
    float c = 0.0f;
    if (A.ls_from_G) c = A.G[A.n_mean + threadIdx.x];
    else for (int i = 0; i < A.nparts; ++i) c += A.partial_ls[i * A.a + threadIdx.x];
    s_gls[threadIdx.x] = c;
    if (blockIdx.x == 0 && !A.ls_from_G) A.G[A.n_mean + threadIdx.x] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < A.nparts; ++i) tot += (double)A.partial[i];
    for (int j = 0; j < A.a; ++j) tot += (double)s_gls[j] * (double)s_gls[j];
    const double norm = sqrt(tot);
    const double coef = (double)A.max_norm / (norm + 1e-6);   // torch.nn.utils.clip_grad_norm_
    s_coef = coef < 1.0 ? (float)coef : 1.0f;
    if (blockIdx.x == 0) { A.sc->grad_norm = (float)norm; A.sc->t_p = A.t_p + 1; A.sc->p_step = s_step; A.sc->p_bc2s = s_bc2s; }
  }
  __syncthreads();
  const float coef = s_coef, step = s_step, bc2s = s_bc2s;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < A.n; i += A.nblk_p * 256) {
    const float graw = (i >= A.n_mean && i < A.n_mean + A.a) ? s_gls[i - A.n_mean] : (i >= A.n_mean ? 0.0f : A.G[i]);
    const float g = graw * coef;
    const float m = A.M[i] * A.b1 + (1.0f - A.b1) * g;
    const float v = A.V[i] * A.b2 + (1.0f - A.b2) * g * g;
    A.M[i] = m; A.V[i] = v;
    A.P[i] = A.P[i] - step * (m / (sqrtf(v) / bc2s + A.eps));
  }
}

// Split run (cfg.grad_world = G): the pieces the fused single-rank minibatch folds into other launches, as launches of their own
// around the all-reduce.  k_ppo_ls_fold: this rank's log_std gradient (the column sums k_ppo_norm left per block, added in k_ppo_update's order) into
// its slot of the policy arena, so that it rides in the arena's all-reduce.
__global__ void k_ppo_ls_fold(float* G, int n_mean, int a, const float* partial_ls, int nparts) {
  const int j = threadIdx.x;
  if (j >= a) return;
  float c = 0.0f;
  for (int i = 0; i < nparts; ++i) c += partial_ls[i * a + j];
  G[n_mean + j] = c;
}
// Adam bias-correction scalars of the NEXT value / policy step (which: 0 = a value step just ran, 1 = policy, -1 = init)
__global__ void k_ppo_refresh(PpoScalars* sc, int which, float v_lr, float p_lr) {
  if (which == 0) sc->t_v += 1;
  if (which == 1) sc->t_p += 1;
  const double b1 = 0.9, b2 = 0.999;
  sc->v_step = (float)((double)v_lr / (1.0 - pow(b1, (double)(sc->t_v + 1))));
  sc->v_bc2s = (float)sqrt(1.0 - pow(b2, (double)(sc->t_v + 1)));
  sc->p_step = (float)((double)p_lr / (1.0 - pow(b1, (double)(sc->t_p + 1))));
  sc->p_bc2s = (float)sqrt(1.0 - pow(b2, (double)(sc->t_p + 1)));
}

// ------------------------------------------------------------------------------------------------ host
struct ilsx_ppo {
  ilsx_ctx* ctx = nullptr;
  ilsx_ppo_cfg cfg;
  NetLayout Lp, Lv;
  int o = 0, a = 0, N = 0;
  size_t np = 0, nv = 0;       // internal floats: policy arena = mean net + log_std (padded to 4), value net
  float *Pp = nullptr, *Gp = nullptr, *Mp = nullptr, *Vp = nullptr;   // policy
  float *Pv = nullptr, *Gv = nullptr, *Mv = nullptr, *Vv = nullptr;   // value
  PpoScalars* sc = nullptr;
  float *values = nullptr, *returns = nullptr, *adv = nullptr, *lp_old = nullptr;      // [max_samples]
  // minibatch workspace (mb rows)
  float *xv = nullptr, *hv[ILSX_MAX_HID], *dv[ILSX_MAX_HID], *dhv = nullptr, *vpred = nullptr;
  float *xp = nullptr, *hp[ILSX_MAX_HID], *dp[ILSX_MAX_HID], *dhp = nullptr, *mu = nullptr, *lp = nullptr, *aux = nullptr;
  float* partial = nullptr;
  int *offs = nullptr, *perm = nullptr;
  DwArgs jobs_vp;              // both nets' weight-gradient jobs: ONE launch per minibatch (the gradient arenas are one allocation: Gv | Gp)
  size_t nv_pad = 0;           // floats from Gv to Gp
  int t_v = 0, t_p = 0;        // optimiser steps taken (host mirror of PpoScalars::t_v / t_p: k_ppo_update gets them as arguments)
  unsigned long long shuffles = 0;   // library-drawn minibatch permutations so far (key of the next one)
  bool cond() const { return cfg.conditioned_std != 0; }   // log-std from the policy net's second head (policies.py:368-374)
  int n_ls() const { return cond() ? 0 : a; }             // trailing action_log_std parameters of the policy arena
  int no() const { return cond() ? 2 * a : a; }           // head outputs of the policy net
  float* log_std() const { return cond() ? nullptr : Pp + Lp.n_int; }   // null = "read it from the head" for the kernels
  float* g_log_std() const { return Gp + Lp.n_int; }
};

static int ppo_refresh(ilsx_ppo* p, int which) {
  hipLaunchKernelGGL(k_ppo_refresh, dim3(1), dim3(1), 0, p->ctx->stream, p->sc, which, p->cfg.value_lr, p->cfg.policy_lr);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_ppo_create(ilsx_ctx* ctx, const ilsx_ppo_cfg* cfg, ilsx_ppo** out) {
  if (!ctx || !cfg || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_create: NULL argument");
  if (cfg->max_samples < 1 || cfg->mini_batch_size < 1) ILSX_FAIL(ILSX_ERR_ARG, "max_samples / mini_batch_size must be >= 1");
  if (cfg->act_dim > PPO_MAX_A) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "act_dim %d > %d", cfg->act_dim, PPO_MAX_A);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_ppo* p = new ilsx_ppo();
  p->ctx = ctx; p->cfg = *cfg; p->o = cfg->obs_dim; p->a = cfg->act_dim;
  if (p->cfg.grad_world < 1) p->cfg.grad_world = 1;
  ilsx_mlp_cfg mp = {cfg->obs_dim, cfg->n_hidden, cfg->hidden, cfg->act_dim, cfg->conditioned_std ? 2 : 1, ILSX_ACT_TANH,
                     {cfg->hidden_sizes[0], cfg->hidden_sizes[1], cfg->hidden_sizes[2]}};
  ilsx_mlp_cfg mv = {cfg->obs_dim, cfg->n_hidden, cfg->hidden, 1, 1, ILSX_ACT_TANH, {cfg->hidden_sizes[0], cfg->hidden_sizes[1], cfg->hidden_sizes[2]}};
  int rc = net_layout_build(mp, &p->Lp);
  if (rc == ILSX_OK) rc = net_layout_build(mv, &p->Lv);
  if (rc != ILSX_OK) { delete p; return rc; }
  p->np = p->Lp.n_int + ((p->n_ls() + 3) / 4) * 4;
  p->nv = p->Lv.n_int;
  const size_t N = (size_t)cfg->max_samples, mb = (size_t)cfg->mini_batch_size, H = (size_t)cfg->hidden;
  auto A = [&](float** q, size_t cnt) { return ctx_alloc(ctx, cnt * sizeof(float), (void**)q, true); };
  p->nv_pad = (p->nv + 63) / 64 * 64;
  rc = A(&p->Pp, p->np);
  if (rc == ILSX_OK) rc = A(&p->Gv, p->nv_pad + p->np);   // Gv | Gp in one allocation: one weight-gradient table, one all-reduce in a split run
  if (rc == ILSX_OK) p->Gp = p->Gv + p->nv_pad;
  if (rc == ILSX_OK) rc = A(&p->Mp, p->np);
  if (rc == ILSX_OK) rc = A(&p->Vp, p->np);
  if (rc == ILSX_OK) rc = A(&p->Pv, p->nv);
  if (rc == ILSX_OK) rc = A(&p->Mv, p->nv);
  if (rc == ILSX_OK) rc = A(&p->Vv, p->nv);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(PpoScalars), (void**)&p->sc);
  if (rc == ILSX_OK) rc = A(&p->values, N);
  if (rc == ILSX_OK) rc = A(&p->returns, N);
  if (rc == ILSX_OK) rc = A(&p->adv, N);
  if (rc == ILSX_OK) rc = A(&p->lp_old, N);
  if (rc == ILSX_OK) rc = A(&p->xv, mb * p->Lv.KP);
  if (rc == ILSX_OK) rc = A(&p->xp, mb * p->Lp.KP);
  for (int l = 0; l < cfg->n_hidden && rc == ILSX_OK; ++l) {
    rc = A(&p->hv[l], mb * H);
    if (rc == ILSX_OK) rc = A(&p->dv[l], mb * H);
    if (rc == ILSX_OK) rc = A(&p->hp[l], mb * H);
    if (rc == ILSX_OK) rc = A(&p->dp[l], mb * H);
  }
  if (rc == ILSX_OK) rc = A(&p->dhv, mb);
  if (rc == ILSX_OK) rc = A(&p->vpred, mb);
  if (rc == ILSX_OK) rc = A(&p->dhp, mb * p->no());
  if (rc == ILSX_OK) rc = A(&p->mu, mb * p->no());
  if (rc == ILSX_OK) rc = A(&p->lp, mb);
  if (rc == ILSX_OK) rc = A(&p->aux, mb * p->a);
  if (rc == ILSX_OK) rc = A(&p->partial, 64 + 64 * PPO_MAX_A);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, (N + 1) * sizeof(int), (void**)&p->offs);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, N * sizeof(int), (void**)&p->perm);
  if (rc != ILSX_OK) { delete p; return rc; }
  memset(&p->jobs_vp, 0, sizeof p->jobs_vp);
  ILSX_TRY(build_dw_jobs(p->Lv, p->Gv, p->xv, p->hv, p->dv, p->dhv, &p->jobs_vp));
  ILSX_TRY(build_dw_jobs(p->Lp, p->Gp, p->xp, p->hp, p->dp, p->dhp, &p->jobs_vp));
  ILSX_TRY(ppo_refresh(p, -1));
  *out = p;
  return ILSX_OK;
}

extern "C" int ilsx_ppo_destroy(ilsx_ppo* p) {
  if (!p) return ILSX_OK;
  hipSetDevice(p->ctx->device);
  hipStreamSynchronize(p->ctx->stream);
  // arenas and workspaces are released with the ctx
  delete p;
  return ILSX_OK;
}

// which: 0 = policy (flat: mean-net parameters | action_log_std[a]), 1 = value net
extern "C" int ilsx_ppo_num_params(const ilsx_ppo* p, int which, size_t* out) {
  if (!p || !out || which < 0 || which > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_num_params: bad argument");
  *out = which == 0 ? p->Lp.n_flat + p->n_ls() : p->Lv.n_flat;
  return ILSX_OK;
}
extern "C" int ilsx_ppo_set_params(ilsx_ppo* p, int which, const float* src, size_t n) {
  if (!p || !src || which < 0 || which > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_set_params: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  if (which == 1) return net_upload_flat(p->ctx, p->Lv, p->Pv, src, n, 0);
  if (n != p->Lp.n_flat + p->n_ls()) ILSX_FAIL(ILSX_ERR_ARG, "policy parameter count %zu != %zu", n, p->Lp.n_flat + p->n_ls());
  ILSX_TRY(net_upload_flat(p->ctx, p->Lp, p->Pp, src, p->Lp.n_flat, 0));
  if (p->n_ls()) HIPCHK(hipMemcpyAsync(p->log_std(), src + p->Lp.n_flat, p->a * sizeof(float), hipMemcpyHostToDevice, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  return ILSX_OK;
}
extern "C" int ilsx_ppo_get_params(ilsx_ppo* p, int which, float* dst, size_t n) {
  if (!p || !dst || which < 0 || which > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_get_params: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  if (which == 1) return net_download_flat(p->ctx, p->Lv, p->Pv, dst, n, 0);
  if (n != p->Lp.n_flat + p->n_ls()) ILSX_FAIL(ILSX_ERR_ARG, "policy parameter count %zu != %zu", n, p->Lp.n_flat + p->n_ls());
  ILSX_TRY(net_download_flat(p->ctx, p->Lp, p->Pp, dst, p->Lp.n_flat, 0));
  if (p->n_ls()) HIPCHK(hipMemcpyAsync(dst + p->Lp.n_flat, p->log_std(), p->a * sizeof(float), hipMemcpyDeviceToHost, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  return ILSX_OK;
}

// Adam state of the policy (mean net | action_log_std) / value optimiser (ppo.py:47-55), flat ABI layout of the block
static int ppo_opt(ilsx_ppo* p, int which, bool set, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!p || !m_host || !v_host || which < 0 || which > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_*_opt: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  const NetLayout& L = which == 0 ? p->Lp : p->Lv;
  float* arenas[2] = {which == 0 ? p->Mp : p->Mv, which == 0 ? p->Vp : p->Vv};
  float* hosts[2] = {m_host, v_host};
  const size_t want = L.n_flat + (which == 0 ? (size_t)p->n_ls() : 0);
  if (n != want) ILSX_FAIL(ILSX_ERR_ARG, "optimiser state count %zu != %zu", n, want);
  for (int k = 0; k < 2; ++k) {
    if (set) ILSX_TRY(net_upload_flat(p->ctx, L, arenas[k], hosts[k], L.n_flat, 0));
    else ILSX_TRY(net_download_flat(p->ctx, L, arenas[k], hosts[k], L.n_flat, 0));
    if (which == 0 && p->n_ls()) {   // the state-independent log-std parameter sits behind the mean net in the policy arena
      if (set) HIPCHK(hipMemcpyAsync(arenas[k] + L.n_int, hosts[k] + L.n_flat, p->a * sizeof(float), hipMemcpyHostToDevice, st));
      else HIPCHK(hipMemcpyAsync(hosts[k] + L.n_flat, arenas[k] + L.n_int, p->a * sizeof(float), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
    }
  }
  if (!meta) return ILSX_OK;
  PpoScalars h;
  HIPCHK(hipMemcpyAsync(&h, p->sc, sizeof h, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!set) { meta->t = which == 0 ? h.t_p : h.t_v; meta->rng_step = p->shuffles; meta->n_train_steps = 0; return ILSX_OK; }
  if (which == 0) { h.t_p = (int)meta->t; p->t_p = h.t_p; } else { h.t_v = (int)meta->t; p->t_v = h.t_v; }
  p->shuffles = meta->rng_step;
  HIPCHK(hipMemcpyAsync(p->sc, &h, sizeof h, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return ppo_refresh(p, -1);
}
extern "C" int ilsx_ppo_get_opt(ilsx_ppo* p, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  return ppo_opt(p, which, false, m_host, v_host, n, meta);
}
extern "C" int ilsx_ppo_set_opt(ilsx_ppo* p, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta) {
  ilsx_opt_meta m2; if (meta) m2 = *meta;
  return ppo_opt(p, which, true, const_cast<float*>(m_host), const_cast<float*>(v_host), n, meta ? &m2 : nullptr);
}

static void ppo_fwd_task(FwdTask& t, const NetLayout& L, float* base, const float* obs, int o) {
  t.net = net_view(L, base);
  t.x0 = obs; t.d0 = o; t.s0 = o;
}

// values[N] = vf(obs) ; fixed log-probs[N] = log pi(act | obs)   (ppo.py:104-113)
static int ppo_full_forward(ilsx_ppo* p, const float* obs, const float* act, int N, bool value, float* out) {
  FwdArgs A;
  memset(&A, 0, sizeof A);
  A.rows = N; A.ntasks = 1; A.seed = p->ctx->seed;
  FwdTask& t = A.t[0];
  if (value) {
    ppo_fwd_task(t, p->Lv, p->Pv, obs, p->o);
    t.head = HEAD_RAW; t.out = out;
    return launch_fwd(p->ctx, A, p->cfg.hidden, ILSX_ACT_TANH, p->Lv.KP);
  }
  ppo_fwd_task(t, p->Lp, p->Pp, obs, p->o);
  t.head = HEAD_GAUSS_LOGP_OF_ACT; t.act_in = act; t.log_std = p->log_std(); t.logp = out;
  return launch_fwd(p->ctx, A, p->cfg.hidden, ILSX_ACT_TANH, p->Lp.KP);
}

extern "C" int ilsx_ppo_values(ilsx_ppo* p, const float* obs, int n, float* values) {
  if (!p || !obs || !values || n < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_values: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  return ppo_full_forward(p, obs, nullptr, n, true, values);
}

extern "C" int ilsx_ppo_gae(ilsx_ppo* p, const float* obs, const float* act, const float* rew, const int32_t* traj_offsets_host,
                            int n_traj, const float* bootstrap_values, float* returns_out, float* adv_out, float* values_out,
                            float* logp_out) {
  if (!p || !obs || !act || !rew || !traj_offsets_host || n_traj < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_gae: bad argument");
  const int N = traj_offsets_host[n_traj];
  if (traj_offsets_host[0] != 0 || N < 1 || N > p->cfg.max_samples) ILSX_FAIL(ILSX_ERR_ARG, "trajectory offsets must span 1..max_samples rows");
  HIPCHK(hipSetDevice(p->ctx->device));
  hipStream_t st = p->ctx->stream;
  HIPCHK(hipMemcpyAsync(p->offs, traj_offsets_host, (size_t)(n_traj + 1) * sizeof(int), hipMemcpyHostToDevice, st));
  ILSX_TRY(ppo_full_forward(p, obs, act, N, true, p->values));
  {
    ProfScope ps(p->ctx, ILSX_K_PPO_GAE);
    ILSX_LAUNCH(ps, k_ppo_gae, dim3((n_traj + 3) / 4), dim3(256), 0, st, p->values, rew, p->offs, bootstrap_values, n_traj, p->cfg.reward_scale,
                       p->cfg.discount, p->cfg.gae_tau, p->returns, p->adv);
  }
  HIPCHK(hipGetLastError());
  ILSX_TRY(ppo_full_forward(p, obs, act, N, false, p->lp_old));
  p->N = N;
  const size_t nb = (size_t)N * sizeof(float);
  if (returns_out) HIPCHK(hipMemcpyAsync(returns_out, p->returns, nb, hipMemcpyDeviceToDevice, st));
  if (adv_out) HIPCHK(hipMemcpyAsync(adv_out, p->adv, nb, hipMemcpyDeviceToDevice, st));
  if (values_out) HIPCHK(hipMemcpyAsync(values_out, p->values, nb, hipMemcpyDeviceToDevice, st));
  if (logp_out) HIPCHK(hipMemcpyAsync(logp_out, p->lp_old, nb, hipMemcpyDeviceToDevice, st));
  HIPCHK(hipStreamSynchronize(st));   // the host offsets buffer may be reused by the caller
  return ILSX_OK;
}

// Split run (cfg.grad_world = G > 1): see ilsx_ppo_cfg.  ILSX_SPLIT_FORCE=1 (tests): an agent with grad_world = 1 whose ctx carries a one-rank
// communicator takes the split-run code path — dW without the fused optimiser, all-reduce, Adam as launches of their own.
static bool ppo_is_split(const ilsx_ppo* p) {
  return p->cfg.grad_world > 1 || (p->ctx->comm != nullptr && getenv("ILSX_SPLIT_FORCE") != nullptr);
}
static int ppo_check_world(const ilsx_ppo* p) {
  if (p->cfg.grad_world == 1) return ILSX_OK;
  if (p->ctx->comm && p->ctx->comm_n == 1 && getenv("ILSX_SPLIT_FORCE")) return ILSX_OK;   // tests: a one-rank communicator stands in, the arena holds this rank's share
  if (!p->ctx->comm || p->ctx->comm_n != p->cfg.grad_world)
    ILSX_FAIL(ILSX_ERR_STATE, "ilsx_ppo_train: grad_world=%d needs a communicator of that many ranks on the ctx (ilsx_comm_init; found %d)",
              p->cfg.grad_world, p->ctx->comm ? p->ctx->comm_n : 0);
  return ILSX_OK;
}

// One minibatch (ppo.py:136-170) as FIVE launches: forward {value, policy}, backward {value, policy}, weight gradients of both nets, the policy
// gradient's norm, the optimiser launch (k_ppo_update).  The reference steps the value net, then the policy; the two updates share no operand (the
// policy loss reads the precomputed advantages), so stepping them side by side gives the same numbers — and at the reference's own minibatch size (64 rows:
// ppo_hopper.yaml) the update is a chain of dependent launches at ~6 us each, ten of them before round 6.
static int ppo_minibatch(ilsx_ppo* p, const float* obs, const float* act, const int* idx, int rows) {
  ilsx_ctx* ctx = p->ctx;
  const int H = p->cfg.hidden, nh = p->cfg.n_hidden;
  const bool split = ppo_is_split(p);
  const float inv = 1.0f / ((float)rows * (float)p->cfg.grad_world);
  {
    FwdArgs A;
    memset(&A, 0, sizeof A);
    A.rows = rows; A.ntasks = 2; A.seed = ctx->seed;
    FwdTask& tv = A.t[0];   // value net (ppo.py:136-153)
    ppo_fwd_task(tv, p->Lv, p->Pv, obs, p->o);
    tv.rows_idx = idx; tv.xsave = p->xv;
    for (int l = 0; l < nh; ++l) tv.hsave[l] = p->hv[l];
    tv.head = HEAD_RAW; tv.out = p->vpred;
    FwdTask& tp = A.t[1];   // policy (ppo.py:155-170)
    ppo_fwd_task(tp, p->Lp, p->Pp, obs, p->o);
    tp.rows_idx = idx; tp.xsave = p->xp;
    for (int l = 0; l < nh; ++l) tp.hsave[l] = p->hp[l];
    tp.head = HEAD_GAUSS_LOGP_OF_ACT; tp.act_in = act; tp.log_std = p->log_std(); tp.logp = p->lp; tp.out = p->mu;
    ILSX_TRY(launch_fwd(ctx, A, H, ILSX_ACT_TANH, std::max(p->Lv.KP, p->Lp.KP)));
  }
  {
    BwdArgs Bw;
    memset(&Bw, 0, sizeof Bw);
    Bw.rows = rows; Bw.ntasks = 2; Bw.inv_B = inv;
    BwdTask& bv = Bw.t[0];
    bv.net = net_view(p->Lv, p->Pv);
    for (int l = 0; l < nh; ++l) { bv.hsave[l] = p->hv[l]; bv.dsave[l] = p->dv[l]; }
    bv.dhead = p->dhv; bv.loss = LOSS_MSE; bv.rows_idx = idx; bv.pred = p->vpred; bv.target = p->returns;
    if (p->cfg.use_value_clip) { bv.lp_old = p->values; bv.clip_eps = p->cfg.clip_eps; }   // ppo.py:137-143
    BwdTask& bp = Bw.t[1];
    bp.net = net_view(p->Lp, p->Pp);
    for (int l = 0; l < nh; ++l) { bp.hsave[l] = p->hp[l]; bp.dsave[l] = p->dp[l]; }
    bp.dhead = p->dhp; bp.loss = LOSS_PPO_POLICY; bp.rows_idx = idx;
    bp.pred = p->lp; bp.target = p->adv; bp.lp_old = p->lp_old; bp.act_all = act; bp.log_std = p->log_std(); bp.mu = p->mu;
    bp.aux = p->cond() ? nullptr : p->aux; bp.clip_eps = p->cfg.clip_eps;
    ILSX_TRY(launch_bwd_dx(ctx, Bw, H, ILSX_ACT_TANH));
  }
  ILSX_TRY(launch_bwd_dw(ctx, p->jobs_vp, rows, nullptr));
  PpoNormArgs Nn;
  Nn.G = p->Gp; Nn.n = (int)p->Lp.n_int;
  for (int l = 1; l <= 2; ++l) {
    Nn.skip0[l - 1] = l < nh ? p->Lp.off_Wb[l] : 0;
    Nn.skip1[l - 1] = l < nh ? p->Lp.off_Wb[l] + H * H : 0;
  }
  Nn.aux = p->aux; Nn.rows = p->cond() ? 0 : rows; Nn.a = p->n_ls(); Nn.partial = p->partial; Nn.partial_ls = p->partial + 64;
  const int nblk = 32;
  hipLaunchKernelGGL(k_ppo_norm, dim3(nblk), dim3(256), 0, ctx->stream, Nn);
  if (split) {   // this rank's log_std gradient into the arena | ONE all-reduce of both nets' gradients (Gv | Gp) | the norm of the SUMMED policy gradient
    if (p->n_ls()) hipLaunchKernelGGL(k_ppo_ls_fold, dim3(1), dim3(64), 0, ctx->stream, p->Gp, (int)p->Lp.n_int, p->n_ls(), Nn.partial_ls, nblk);
    HIPCHK(hipGetLastError());
    ILSX_TRY(comm_allreduce_sum(ctx, p->Gv, p->nv_pad + p->np));
    Nn.rows = 0; Nn.a = 0;
    hipLaunchKernelGGL(k_ppo_norm, dim3(nblk), dim3(256), 0, ctx->stream, Nn);
  }
  PpoUpdateArgs C;
  memset(&C, 0, sizeof C);
  C.ls_from_G = split ? 1 : 0;
  C.P = p->Pp; C.G = p->Gp; C.M = p->Mp; C.V = p->Vp; C.n = (int)p->np;
  C.partial = p->partial; C.partial_ls = p->partial + 64; C.nparts = nblk; C.n_mean = (int)p->Lp.n_int; C.a = p->n_ls(); C.max_norm = 20.0f; C.b1 = 0.9f; C.b2 = 0.999f; C.eps = 1e-8f; C.sc = p->sc;
  C.Pv = p->Pv; C.Gv = p->Gv; C.Mv = p->Mv; C.Vv = p->Vv; C.nv = (int)p->nv;
  C.l2x2 = 2.0f * p->cfg.value_l2_reg; C.v_lr = p->cfg.value_lr; C.p_lr = p->cfg.policy_lr;
  C.t_v = p->t_v++; C.t_p = p->t_p++;
  C.nblk_p = 64;
  hipLaunchKernelGGL(k_ppo_update, dim3(128), dim3(256), 0, ctx->stream, C);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// PPO.train_step (ppo.py:102-170).  perms_host: update_epoch x N int32 permutations (the reference draws
// torch.randperm per epoch), or NULL = drawn here from a host Mersenne twister seeded by the ctx seed.
extern "C" int ilsx_ppo_train(ilsx_ppo* p, const float* obs, const float* act, const float* rew,
                              const int32_t* traj_offsets_host, int n_traj, const float* bootstrap_values,
                              const int32_t* perms_host) {
  if (!p) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_train: NULL agent");
  ILSX_TRY(ppo_check_world(p));
  ILSX_TRY(ilsx_ppo_gae(p, obs, act, rew, traj_offsets_host, n_traj, bootstrap_values, nullptr, nullptr, nullptr, nullptr));
  const int N = p->N, mb = p->cfg.mini_batch_size;
  int half_bits = 1;
  while ((1ll << (2 * half_bits)) < (long long)N) ++half_bits;
  for (int ep = 0; ep < p->cfg.update_epoch; ++ep) {
    if (perms_host) {
      HIPCHK(hipStreamSynchronize(p->ctx->stream));   // previous epoch's kernels still read p->perm
      HIPCHK(hipMemcpyAsync(p->perm, perms_host + (size_t)ep * N, (size_t)N * sizeof(int), hipMemcpyHostToDevice, p->ctx->stream));
      HIPCHK(hipStreamSynchronize(p->ctx->stream));
    } else {
      const unsigned long long shuffles = ++p->shuffles;   // per agent (was process-wide): resumable through ilsx_ppo_set_opt
      hipLaunchKernelGGL(k_ppo_perm, dim3((N + 255) / 256), dim3(256), 0, p->ctx->stream, p->perm, N, half_bits,
                         (uint32_t)(p->ctx->seed ^ 0x70657261u), (uint32_t)shuffles);
      HIPCHK(hipGetLastError());
    }
    for (int s = 0; s < N; s += mb) ILSX_TRY(ppo_minibatch(p, obs, act, p->perm + s, std::min(mb, N - s)));
  }
  return ILSX_OK;
}

// the policy gradient's norm of the last minibatch, as clip_grad_norm_ saw it (split run: of the summed gradient), for tests
extern "C" int ilsx_ppo_debug_grad_norm(ilsx_ppo* p, float* out) {
  if (!p || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_debug_grad_norm: NULL argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  PpoScalars h;
  HIPCHK(hipMemcpyAsync(&h, p->sc, sizeof h, hipMemcpyDeviceToHost, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  *out = h.grad_norm;
  return ILSX_OK;
}

// one library-drawn shuffle of [0, n) (what ilsx_ppo_train uses when perms_host == NULL), for tests
extern "C" int ilsx_ppo_debug_perm(ilsx_ppo* p, int n, uint32_t key, int32_t* perm_host) {
  if (!p || !perm_host || n < 1 || n > p->cfg.max_samples) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_debug_perm: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  int half_bits = 1;
  while ((1ll << (2 * half_bits)) < (long long)n) ++half_bits;
  hipLaunchKernelGGL(k_ppo_perm, dim3((n + 255) / 256), dim3(256), 0, p->ctx->stream, p->perm, n, half_bits,
                     (uint32_t)(p->ctx->seed ^ 0x70657261u), key);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(perm_host, p->perm, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  return ILSX_OK;
}

// ReparamMultivariateGaussianPolicy.get_actions (policies.py:392-417): mean + exp(log_std)*eps, un-squashed
extern "C" int ilsx_ppo_policy_act(ilsx_ppo* p, const float* obs, int n, int deterministic, const float* eps, float* act,
                                   float* logp) {
  if (!p || !obs || !act || n < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ppo_policy_act: bad argument");
  HIPCHK(hipSetDevice(p->ctx->device));
  FwdArgs A;
  memset(&A, 0, sizeof A);
  A.rows = n; A.ntasks = 1; A.seed = p->ctx->seed; A.step_host = ++p->ctx->ppo_act_calls;
  FwdTask& t = A.t[0];
  ppo_fwd_task(t, p->Lp, p->Pp, obs, p->o);
  if (deterministic) { t.head = HEAD_RAW; t.out = act; t.out_cols = p->a; }   // action = mean (policies.py:407-408): the first a head outputs
  else { t.head = HEAD_GAUSS_SAMPLE; t.eps = eps; t.action = act; t.logp = logp; t.log_std = p->log_std(); t.rng_stream = 0x50504f00u; }
  return launch_fwd(p->ctx, A, p->cfg.hidden, ILSX_ACT_TANH, p->Lp.KP);
}
