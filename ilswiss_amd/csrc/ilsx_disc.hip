// ilsx_disc.hip — adversarial-IRL discriminator: rlkit/torch/algorithms/adv_irl/adv_irl.py:133-216
// (_do_reward_training: BCE-with-logits + WGAN-GP gradient penalty), :277-298 (reward modes) and
// disc_models/simple_disc_models.py:8-48 (MLPDisc: 1-3 x (Linear, act) then Linear, clamp +-10, no BN; the fused kernel below is the
// two-block network of every reference spec, disc_step_blocks the other depths).
//
// The reference gets the gradient-penalty gradient from autograd's double backward; here it is derived by
// hand (oracle/disc.py) and fused into ONE row-tile kernel.  One step =
//   k_disc_prep   X = [expert (B) ; policy (B) ; eps*expert + (1-eps)*policy (B)]
//   forward       the shared MLP forward kernels over all 3B rows (h1, h2, raw logits kept)
//   k_disc_bwd    per 16-row tile: CE rows -> delta2, delta1 ; GP rows -> dD/dx, its norm, and the whole
//                 second-order chain (3 MFMA GEMMs with W2 / W2^T), written as ROW-STACKED operand
//                 matrices so that every weight gradient is a single A^T.B contraction
//   k_mlp_bwd_dw  3 stacked jobs: dW1 over 4B rows, dW2 over 4B rows, dw3 over 3B rows (bias rows limited);
//                 epilogue: Adam(lr, betas=(disc_momentum, 0.999)) on the tile it owns
//   k_disc_tail   losses / accuracy, step counter, Adam scalars of the next step
#include <cmath>

#include "host_common.h"
#include "disc_bn_step.h"

struct DiscScalars {
  float ce_loss, grad_pen, accuracy, pad;
  float adam_step, adam_bc2s;
  int t, pad1;
};

struct DiscBwdArgs {
  NetView net;
  int B, rows, use_gp, D;
  int world;     // split run (cfg.grad_world): the means are over B * world rows
  float clamp, gp_w;
  PartVal raw;
  float* hs0;    // [4B][H]  rows < 3B: h1 (read); rows 3B..4B: v1-bar (written)
  float* hs1;    // [3B][H]  h2 (read); GP rows are overwritten with u2-bar * phi2'
  float* xs;     // [4B][KP] rows < 3B: X; rows 3B..4B: g-bar (written, zero-padded)
  float* A2;     // [4B][H]  delta2 | z2-bar | u2
  float* A1;     // [4B][H]  delta1 | z1-bar | gate*u1
  float* dhead;  // [3B]     dL/dlogit for CE rows, 1 for GP rows
  float *ce_row, *correct, *gp_row;
};

template <int ACT> __device__ __forceinline__ float d2act_from_out(float h) {  // d phi'(z)/dz in terms of h
  if (ACT == ACT_RELU) return 0.0f;
  return -2.0f * h * (1.0f - h * h);
}

__global__ void k_disc_prep(const float* __restrict__ eo, const float* __restrict__ ea, const float* __restrict__ po,
                            const float* __restrict__ pa, const float* __restrict__ eps, int B, int o, int a, int use_gp,
                            uint64_t seed, uint32_t stream, unsigned long long step, float* __restrict__ X,
                            float* __restrict__ eps_used) {
  const int D = o + a, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * D) return;
  const int r = e / D, c = e - r * D;
  const float xe = c < o ? eo[(size_t)r * o + c] : ea[(size_t)r * a + (c - o)];
  const float xp = c < o ? po[(size_t)r * o + c] : pa[(size_t)r * a + (c - o)];
  X[(size_t)r * D + c] = xe;
  X[(size_t)(B + r) * D + c] = xp;
  if (use_gp) {
    float w;
    if (eps) {
      w = eps[r];
    } else {  // ptu.rand(B, 1): U[0,1) (adv_irl.py:184)
      uint32_t ctr[4] = {(uint32_t)r >> 2, 0x44495343u, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
      philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
      w = (float)(ctr[r & 3] >> 8) * (1.0f / 16777216.0f);
    }
    X[(size_t)(2 * B + r) * D + c] = w * xe + (1.0f - w) * xp;   // adv_irl.py:187
    if (c == 0 && eps_used) eps_used[r] = w;
  }
}

// k_disc_prep with the two batches drawn from the replay rings in place (the adversarial-IRL loop): row r of the expert / policy block is
// the record ilsx_replay_sample would have drawn under the same counters (replay_draw), so X is bit for bit what
// sample(expert) ; sample(policy) ; k_disc_prep builds — two launches and two staging round trips less per discriminator step.
struct DiscRing { const float* data; const DevReplayState* st; uint64_t seed; uint32_t stream; int rec; unsigned long long step; };
__global__ void k_disc_prep_rings(const DiscRing E, const DiscRing P, int B, int o, int a, int state_only, int use_gp, uint64_t seed,
                                  uint32_t stream, unsigned long long step, float* __restrict__ X, float* __restrict__ eps_used) {
  const int D = o + (state_only ? o : a), e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * D) return;
  const int r = e / D, c = e - r * D;
  const int col = c < o ? c : (state_only ? o + a + 2 + (c - o) : o + (c - o));   // record = [obs | act | rew | done | next_obs]
  const float xe = E.data[(size_t)replay_draw(E.seed, E.step, E.stream, (uint32_t)r, E.st->size) * E.rec + col];
  const float xp = P.data[(size_t)replay_draw(P.seed, P.step, P.stream, (uint32_t)r, P.st->size) * P.rec + col];
  X[(size_t)r * D + c] = xe;
  X[(size_t)(B + r) * D + c] = xp;
  if (use_gp) {
    uint32_t ctr[4] = {(uint32_t)r >> 2, 0x44495343u, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
    const float w = (float)(ctr[r & 3] >> 8) * (1.0f / 16777216.0f);   // ptu.rand(B, 1): U[0,1) (adv_irl.py:184)
    X[(size_t)(2 * B + r) * D + c] = w * xe + (1.0f - w) * xp;   // adv_irl.py:187
    if (c == 0 && eps_used) eps_used[r] = w;
  }
}

template <int H, int ACT>
__global__ __launch_bounds__(4 * H) void k_disc_bwd(const DiscBwdArgs A) {
  constexpr int NW = H / 16, NTH = 4 * H, NC = H / 16, KPL = H / 64, RPW = 16 / NW;
  constexpr int RPT = 16 * H / NTH, RSTEP = NTH / H;
  constexpr int LDH = H + ILSX_LDS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufA = smem;               // MFMA A operand of the current GEMM
  float* bufB = bufA + 16 * LDH;    // v1, later phi1'-bar
  float* bufC = bufB + 16 * LDH;    // u1, later z2-bar
  float* gs = bufC + 16 * LDH;      // [16][64]  dD/dx, then g-bar
  float* rowf = gs + 16 * 64;       // [16][4]   dlogit, gate, isgp
  const NetView& N = A.net;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, rows = A.rows, B = A.B, KP = N.KP, D = A.D;
  const int k1 = tid % H, rb1 = tid / H, c0 = wave * 16, col = c0 + li;
  const float* Wh = N.base + N.off_Wh;

  // ---- operands of every phase, requested up front
  float h1v[RPT], h2v[RPT], h1m[4], h2m[4];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int gr = r0 + rb1 + RSTEP * i;
    h1v[i] = gr < rows ? A.hs0[(size_t)gr * H + k1] : 0.0f;
    h2v[i] = gr < rows ? A.hs1[(size_t)gr * H + k1] : 0.0f;
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int gr = r0 + 4 * g + v;
    h1m[v] = gr < rows ? A.hs0[(size_t)gr * H + col] : 0.0f;
    h2m[v] = gr < rows ? A.hs1[(size_t)gr * H + col] : 0.0f;
  }
  const float w3k = Wh[k1], w3m = Wh[col];
  float4 wreg[NC];
  {
    const float* wp = N.base + N.off_Wb[1] + (size_t)wave * NC * 256 + 4 * lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
  }
  // ---- per-row loss head: BCE-with-logits through the clamp gate (adv_irl.py:176-179, simple_disc_models.py:45-47)
  if (tid < 16) {
    const int gr = r0 + tid;
    float dl = 0.0f, gate = 0.0f, isgp = 0.0f;
    if (gr < rows) {
      const float raw = A.raw.get(gr);
      gate = (raw >= -A.clamp && raw <= A.clamp) ? 1.0f : 0.0f;
      const float l = fminf(fmaxf(raw, -A.clamp), A.clamp);
      if (gr < 2 * B) {
        const float t = gr < B ? 1.0f : 0.0f;
        dl = gate * (1.0f / (1.0f + expf(-l)) - t) / (float)(2 * B * A.world);
        A.ce_row[gr] = fmaxf(l, 0.0f) - l * t + log1pf(expf(-fabsf(l)));
        A.correct[gr] = ((l > 0.0f) == (t > 0.5f)) ? 1.0f : 0.0f;
        A.dhead[gr] = dl;
      } else {
        isgp = 1.0f;
        A.dhead[gr] = 1.0f;   // the "ones" rows that carry (u2-bar * phi2') into dw3
      }
    }
    rowf[tid * 4 + 0] = dl; rowf[tid * 4 + 1] = gate; rowf[tid * 4 + 2] = isgp;
  }
  __syncthreads();
  // ---- GEMM 1 operand: CE rows delta2 = dlogit*w3*phi2' ; GP rows u2 = phi2'*w3
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int row = rb1 + RSTEP * i, gr = r0 + row;
    const bool gp = rowf[row * 4 + 2] > 0.5f;
    const float p2 = act_grad_from_out<ACT>(h2v[i]);
    float val = 0.0f;
    if (gr < rows) {
      val = gp ? p2 * w3k : rowf[row * 4 + 0] * w3k * p2;
      A.A2[(size_t)(gp ? gr + B : gr) * H + k1] = val;
    }
    bufA[row * LDH + k1] = val;
  }
  __syncthreads();
  auto gemm = [&](const float* src) {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* ap = src + li * LDH + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(ap + 16 * c);
      acc0 = MFMA16(a.x, wreg[c].x, acc0); acc1 = MFMA16(a.y, wreg[c].y, acc1);
      acc0 = MFMA16(a.z, wreg[c].z, acc0); acc1 = MFMA16(a.w, wreg[c].w, acc1);
    }
    return acc0 + acc1;
  };
  // ---- GEMM 1 (x W2): CE rows -> delta1 ; GP rows -> v1, u1
  {
    const f32x4 acc = gemm(bufA);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * g + v, gr = r0 + row;
      const bool gp = rowf[row * 4 + 2] > 0.5f;
      const float p1 = act_grad_from_out<ACT>(h1m[v]);
      float v1 = 0.0f, u1 = 0.0f;
      if (gr < rows) {
        if (!gp) {
          A.A1[(size_t)gr * H + col] = acc[v] * p1;
        } else {
          v1 = acc[v]; u1 = p1 * v1;
          A.A1[(size_t)(gr + B) * H + col] = rowf[row * 4 + 1] * u1;
        }
      }
      bufB[row * LDH + col] = v1;
      bufC[row * LDH + col] = u1;
    }
  }
  if (!A.use_gp || r0 + 15 < 2 * B) return;   // tile has no gradient-penalty row (workgroup-uniform)
  __syncthreads();
  // ---- g = gate * u1 W1 (dD/dx), its norm, g-bar = dGP/dg    (wave <-> rows, lanes split H)
  const float* W1 = N.base + N.off_W[0];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr, gr = r0 + row;
    float uv[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) uv[i] = bufC[row * LDH + lane + 64 * i];
    float mine = 0.0f;
    for (int d = 0; d < D; ++d) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < KPL; ++i) s = fmaf(uv[i], W1[pack_f(lane + 64 * i, d, KP)], s);
      s = wave_sum(s);
      if (lane == d) mine = s * rowf[row * 4 + 1];
    }
    float sq = wave_sum(mine * mine);
    const float n = sqrtf(sq);
    const bool gp = rowf[row * 4 + 2] > 0.5f && gr < rows;
    // adv_irl.py:201-202; a clamped interpolate has g == 0: torch's norm backward is 0 there (not 0/0), the row still
    // counts (0 - 1)^2 in the penalty value
    const float coef = (gp && n > 0.0f) ? A.gp_w / (float)(B * A.world) * 2.0f * (n - 1.0f) / n : 0.0f;
    const float gb = coef * mine;
    gs[row * 64 + lane] = gb;
    if (gp) {
      if (lane < KP) A.xs[(size_t)(gr + B) * KP + lane] = gb;
      if (lane == 0) A.gp_row[gr - 2 * B] = (n - 1.0f) * (n - 1.0f);
    }
  }
  __syncthreads();
  // ---- u1-bar = gate * g-bar W1^T ; v1-bar = u1-bar*phi1' ; phi1'-bar = u1-bar*v1   (thread <-> column k1)
  {
    const float* wp = N.base + N.off_W[1] + (size_t)wave * NC * 256 + 4 * lane;   // W2 forward-packed for GEMM 2
#pragma unroll
    for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
    float ub[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) ub[i] = 0.0f;
    for (int d4 = 0; d4 < KP; d4 += 4) {
      const float4 w = *reinterpret_cast<const float4*>(W1 + pack_f(k1, d4, KP));
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const float* gr_ = gs + (rb1 + RSTEP * i) * 64 + d4;
        ub[i] = fmaf(gr_[0], w.x, ub[i]); ub[i] = fmaf(gr_[1], w.y, ub[i]);
        ub[i] = fmaf(gr_[2], w.z, ub[i]); ub[i] = fmaf(gr_[3], w.w, ub[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = rb1 + RSTEP * i, gr = r0 + row;
      const bool gp = rowf[row * 4 + 2] > 0.5f && gr < rows;
      const float u1b = gp ? rowf[row * 4 + 1] * ub[i] : 0.0f;
      const float v1 = bufB[row * LDH + k1];
      const float v1b = u1b * act_grad_from_out<ACT>(h1v[i]);
      bufA[row * LDH + k1] = v1b;
      bufB[row * LDH + k1] = u1b * v1;           // phi1'-bar
      if (gp) A.hs0[(size_t)(gr + B) * H + k1] = v1b;
    }
  }
  __syncthreads();
  // ---- GEMM 2 (x W2^T): u2-bar ; z2-bar = u2-bar*w3*phi2'' ; (u2-bar*phi2') -> the stacked dw3 operand
  {
    const f32x4 acc = gemm(bufA);
    {
      const float* wp = N.base + N.off_Wb[1] + (size_t)wave * NC * 256 + 4 * lane;   // W2 backward-packed for GEMM 3
#pragma unroll
      for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * g + v, gr = r0 + row;
      const bool gp = rowf[row * 4 + 2] > 0.5f && gr < rows;
      float z2b = 0.0f;
      if (gp) {
        z2b = acc[v] * w3m * d2act_from_out<ACT>(h2m[v]);
        A.A2[(size_t)gr * H + col] = z2b;
        A.hs1[(size_t)gr * H + col] = acc[v] * act_grad_from_out<ACT>(h2m[v]);
      }
      bufC[row * LDH + col] = z2b;
    }
  }
  __syncthreads();
  // ---- GEMM 3 (x W2): h1-bar ; z1-bar = h1-bar*phi1' + phi1'-bar*phi1''
  {
    const f32x4 acc = gemm(bufC);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * g + v, gr = r0 + row;
      if (rowf[row * 4 + 2] > 0.5f && gr < rows)
        A.A1[(size_t)gr * H + col] = acc[v] * act_grad_from_out<ACT>(h1m[v]) + bufB[row * LDH + col] * d2act_from_out<ACT>(h1m[v]);
    }
  }
}

__global__ __launch_bounds__(256) void k_disc_tail(DiscScalars* sc, const float* ce_row, const float* correct,
                                                   const float* gp_row, int B, int use_gp, float lr, float b1, float b2) {
  __shared__ float sh[4];
  float ce = 0.f, ac = 0.f, gp = 0.f;
  for (int r = threadIdx.x; r < 2 * B; r += 256) { ce += ce_row[r]; ac += correct[r]; }
  if (use_gp) for (int r = threadIdx.x; r < B; r += 256) gp += gp_row[r];
  ce = block256_sum(ce, sh); ac = block256_sum(ac, sh); gp = block256_sum(gp, sh);
  if (threadIdx.x == 0) {
    sc->ce_loss = ce / (float)(2 * B);
    sc->accuracy = ac / (float)(2 * B);
    sc->grad_pen = use_gp ? gp / (float)B : 0.0f;
    sc->t += 1;
    const int t = sc->t + 1;
    sc->adam_step = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
    sc->adam_bc2s = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
}
__global__ void k_disc_refresh(DiscScalars* sc, float lr, float b1, float b2) {
  const int t = sc->t + 1;
  sc->adam_step = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
  sc->adam_bc2s = (float)sqrt(1.0 - pow((double)b2, (double)t));
}

// reward relabelling (adv_irl.py:277-298) from raw logits
__global__ void k_disc_reward(PartVal raw, int n, float clamp, int mode, int has_min, float rmin, int has_max, float rmax,
                              float* __restrict__ rew, float* __restrict__ logits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float x = fminf(fmaxf(raw.get(r), -clamp), clamp);
  float v;
  if (mode == ILSX_DISC_AIRL) v = x;
  else if (mode == ILSX_DISC_GAIL) v = x > 20.0f ? x : log1pf(expf(x));                 // F.softplus(x, beta=1)
  else if (mode == ILSX_DISC_GAIL2) v = -x > 20.0f ? x : -log1pf(expf(-x));             // F.softplus(x, beta=-1)
  else v = expf(x) * (-1.0f * x);                                                        // fairl
  if (has_max) v = fminf(v, rmax);
  if (has_min) v = fmaxf(v, rmin);
  if (rew) rew[r] = v;
  if (logits) logits[r] = x;
}

// ------------------------------------------------------------------------------------------------ BatchNorm discriminator (use_bn)
// The phases of csrc/disc_bn.h in the order of csrc/disc_bn_step.h, every phase one launch on the ctx stream (the host test harness
// runs the same two headers as serial loops: tests/test_disc_bn_host.py).
template <class F> __global__ __launch_bounds__(256) void k_dbn_par(int n, F f) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) f(i);
}
template <class F> __global__ __launch_bounds__(64) void k_dbn_col(F f) { f((int)blockIdx.x, (int)threadIdx.x); }   // one wavefront per feature column
// The matrix products of the step (DbnGemm, disc_bn.h): a 256-thread workgroup owns a 16 x 16 output tile; its four waves split the
// contraction into the four consecutive ranges of dbn_kq(Kd) terms, each wave staging its own 32-term chunks of both operands in its own
// LDS region (every global operand word is read once per tile, consecutive lanes along the operand's unit stride) and holding 2 x 2 outputs
// per lane; the four partial tiles are added in range order — the chain of dbn_gemm_elem, bit for bit.  Small tiles on purpose: the weight
// gradients contract over all rows into few outputs (128 x 128 outputs over 512 rows = 64 workgroups); 32 x 32 tiles without the split ran
// 23 us per launch (16 workgroups, 16 serial load -> LDS -> multiply rounds), one thread per element with a strided walk 50-100 us.
// Two independent products can share a launch.
struct DbnGemm2 { DbnGemm g[3]; int start[4], tn[3]; };   // start[i]: first workgroup of product i (start[3] = grid size)
struct DbnGemmLds { float red[4][256]; };   // the four waves' partial tiles
// One 16 x 16 output tile of one product, by the 256 threads of a workgroup: wave w contracts the w-th of the four consecutive ranges of
// dbn_kq(Kd) terms on the exact-fp32 matrix pipe — v_mfma_f32_16x16x4_f32 is an fmaf chain over its four k slots in ascending order
// (tools/ubench/mfma_32x32.hip checks it against a host fmaf chain), so a wave's partial is the chain dbn_gemm_elem states for its range, and
// the four partials are added in range order: the same bits as the host emulation (padding terms are 0 * 0 products: + 0.0f).  Operands go
// from memory straight into the fragment layout (lane = (row | column) + 16 x k slot): no LDS staging, no barrier inside the contraction —
// the round-5 form (every lane 2 x 2 outputs, both operands staged through 18 KB of LDS, two barriers per 32 terms) ran the single-XCD
// step's product phases twice as long.  Loads are issued DBN_GQ MFMAs ahead.
#define DBN_GQ 8
__device__ __forceinline__ void dbn_gemm_tile(const DbnGemm& g, int tile, int tn, DbnGemmLds& S) {
  const int i0 = (tile / tn) * 16, j0 = (tile % tn) * 16, t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int lr = lane & 15, ks = lane >> 4;   // row of the A fragment / column of the B fragment ; k slot
  const int kq = dbn_kq(g.Kd), kbeg = wave * kq, kend = kbeg + kq < g.Kd ? kbeg + kq : g.Kd;
  const bool i_ok = i0 + lr < g.M, j_ok = j0 + lr < g.N;
  const float* pa = g.A + (size_t)(i0 + lr) * g.sai;
  const float* pb = g.B + (size_t)(j0 + lr) * g.sbj;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = kbeg; k0 < kend; k0 += 4 * DBN_GQ) {   // wave-uniform bounds
    float va[DBN_GQ], vb[DBN_GQ];
#pragma unroll
    for (int q = 0; q < DBN_GQ; ++q) {
      const int k = k0 + 4 * q + ks;
      va[q] = (i_ok && k < kend) ? pa[(size_t)k * g.sak] : 0.0f;
      vb[q] = (j_ok && k < kend) ? pb[(size_t)k * g.sbk] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < DBN_GQ; ++q)
      if (k0 + 4 * q < kend) acc = MFMA16(va[q], vb[q], acc);   // wave-uniform
  }
  __syncthreads();   // (the previous tile's reduction has been read)
#pragma unroll
  for (int v = 0; v < 4; ++v) S.red[wave][(4 * ks + v) * 16 + lr] = acc[v];   // accumulator layout: lane = column + 16 x (row / 4), register = row % 4
  __syncthreads();
  const int gi = i0 + (t >> 4), gj = j0 + (t & 15);
  if (gi < g.M && gj < g.N) {
    float v = ((S.red[0][t] + S.red[1][t]) + S.red[2][t]) + S.red[3][t];
    if (g.bias) v = v + g.bias[gj];
    float* c = g.C + (size_t)gi * g.ldc + gj;
    *c = g.acc ? *c + v : v;
  }
}
__global__ __launch_bounds__(256) void k_dbn_gemm(const DbnGemm2 G2) {
  __shared__ __attribute__((aligned(16))) DbnGemmLds S;
  const int which = (int)blockIdx.x >= G2.start[2] ? 2 : (int)blockIdx.x >= G2.start[1] ? 1 : 0;
  dbn_gemm_tile(G2.g[which], (int)blockIdx.x - G2.start[which], G2.tn[which], S);
}

struct DbnLaunch {
  hipStream_t st;
  template <class F> void par(int n, F f) { if (n > 0) hipLaunchKernelGGL(k_dbn_par<F>, dim3((n + 255) / 256), dim3(256), 0, st, n, f); }
  template <class F> void col(int H, F f) { if (H > 0) hipLaunchKernelGGL(k_dbn_col<F>, dim3(H), dim3(64), 0, st, f); }
  static int tiles(const DbnGemm& g, int* tn) { *tn = (g.N + 15) / 16; return g.Kd > 0 ? ((g.M + 15) / 16) * *tn : 0; }
  void launch(const DbnGemm* gs, int n) {   // up to three products nobody of which reads what another one writes
    DbnGemm2 G2;
    int at = 0;
    for (int i = 0; i < 3; ++i) {
      G2.g[i] = gs[i < n ? i : 0];
      G2.start[i] = at;
      G2.tn[i] = 1;
      if (i < n) at += tiles(gs[i], &G2.tn[i]);
    }
    G2.start[3] = at;
    if (at > 0) hipLaunchKernelGGL(k_dbn_gemm, dim3(at), dim3(256), 0, st, G2);
  }
  void gemm(const DbnGemm& g1) { launch(&g1, 1); }
  void gemm(const DbnGemm& g1, const DbnGemm& g2) { const DbnGemm gs[2] = {g1, g2}; launch(gs, 2); }
  void gemm(const DbnGemm& g1, const DbnGemm& g2, const DbnGemm& g3) { const DbnGemm gs[3] = {g1, g2, g3}; launch(gs, 3); }
};
struct DiscBn {
  DbnNet N;
  DbnWs W;
  float* stats3 = nullptr;   // device: mean BCE, accuracy, mean (|g| - 1)^2 of the last step
  float* xcat = nullptr;     // [rows][D] cat(obs, second) of a reward call
  float* lg = nullptr;       // [rows] eval-mode logits of a reward call
  int rows = 0;              // workspace rows (3 * max_batch)
  int t = 0;                 // Adam step count
};

// ------------------------------------------------------------------------------------------------ host
struct ilsx_disc {
  ilsx_ctx* ctx = nullptr;
  ilsx_disc_cfg cfg;
  NetLayout L;
  int cs = 1, D = 0, o = 0, a = 0;
  int from_expert = 0;   // policy_optim_batch_size_from_expert (adv_irl.py:239-255)
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  DiscScalars* scal = nullptr;
  float *X = nullptr, *xs = nullptr, *hs0 = nullptr, *hs1 = nullptr, *A2 = nullptr, *A1 = nullptr, *dhead = nullptr;
  float *raw = nullptr, *ce_row = nullptr, *correct = nullptr, *gp_row = nullptr, *eps_used = nullptr;
  DwArgs jobs;
  int jobs_B = -1;
  // num_layer_blocks != 2 (disc_step_blocks): per hidden layer l the activations hsb[l] [4B][H] (rows < 3B: forward; rows 3B..4B: v-bar_l),
  // the row-stacked dW operand Ab[l] [4B][H] (delta_l of the CE rows | delta''_l of the GP rows | u_l) and z-bar_l [B][H]
  int nblk = 2;
  float *hsb[ILSX_MAX_HID] = {nullptr, nullptr, nullptr}, *Ab[ILSX_MAX_HID] = {nullptr, nullptr, nullptr}, *zb[ILSX_MAX_HID] = {nullptr, nullptr, nullptr};
  float *given = nullptr, *gdx = nullptr, *dhead4 = nullptr;
  uint32_t rng_stream = 0;
  unsigned long long step_ctr = 0;
  float* snap = nullptr;   // checkpoint of P | M | V | scalars for a window that is rolled back (ilsx_advirl_train)
  DiscBn* bn = nullptr;    // use_bn: natural-layout parameters + the phase workspace (P / G / M / V above are its arenas)
  PartVal pv() const { return PartVal{raw, cs, 3 * cfg.max_batch}; }
};

static int disc_refresh(ilsx_disc* d) {
  hipLaunchKernelGGL(k_disc_refresh, dim3(1), dim3(1), 0, d->ctx->stream, d->scal, d->cfg.disc_lr, d->cfg.disc_momentum, 0.999f);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_disc_create(ilsx_ctx* ctx, const ilsx_disc_cfg* cfg, ilsx_disc** out) {
  if (!ctx || !cfg || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_create: NULL argument");
  if (cfg->obs_dim < 1 || cfg->act_dim < 1 || cfg->obs_dim + cfg->act_dim > 64)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "discriminator input obs+act=%d: this kernel supports up to 64", cfg->obs_dim + cfg->act_dim);
  if (cfg->max_batch < 1) ILSX_FAIL(ILSX_ERR_ARG, "max_batch must be >= 1");
  if (cfg->state_only && cfg->act_dim != cfg->obs_dim)
    ILSX_FAIL(ILSX_ERR_ARG, "state_only: the second input segment is next_obs, act_dim (%d) must equal obs_dim (%d)", cfg->act_dim, cfg->obs_dim);
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_disc* d = new ilsx_disc();
  d->ctx = ctx; d->cfg = *cfg; d->o = cfg->obs_dim; d->a = cfg->act_dim; d->D = d->o + d->a;
  if (d->cfg.grad_world < 1) d->cfg.grad_world = 1;
  if (d->cfg.grad_world > 1 && cfg->use_bn) {
    delete d;
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "a BatchNorm discriminator does not split over ranks (grad_world=%d): its batch statistics would have to cross them", cfg->grad_world);
  }
  d->nblk = cfg->num_layer_blocks ? cfg->num_layer_blocks : 2;
  if (d->nblk < 1 || d->nblk > ILSX_MAX_HID) { const int nb = d->nblk; delete d; ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "num_layer_blocks=%d: 1..%d", nb, ILSX_MAX_HID); }
  if (cfg->use_bn) {   // Linear -> BatchNorm1d -> act blocks: natural layout, any width, the phase chain of csrc/disc_bn_step.h
    if (cfg->hid_dim < 1 || cfg->hid_dim > 1024) { delete d; ILSX_FAIL(ILSX_ERR_ARG, "hid_dim=%d out of range", cfg->hid_dim); }
    if (cfg->hid_act != ILSX_ACT_RELU && cfg->hid_act != ILSX_ACT_TANH) { delete d; ILSX_FAIL(ILSX_ERR_ARG, "hid_act=%d unknown", cfg->hid_act); }
    DiscBn* b = d->bn = new DiscBn();
    DbnNet& N = b->N;
    N.D = d->D; N.H = cfg->hid_dim; N.nblk = d->nblk; N.act = cfg->hid_act == ILSX_ACT_TANH ? DBN_TANH : DBN_RELU; N.clampv = cfg->clamp_magnitude;
    const size_t np = (size_t)N.n_params(), H = (size_t)N.H, rows = 3 * (size_t)cfg->max_batch, wd = std::max(H, (size_t)d->D);
    b->rows = (int)rows;
    d->L.n_flat = np; d->L.n_int = np;   // flat ABI order == storage order (torch's parameters(): per block W | b | gamma | beta, then the output layer)
    d->cs = 1;
    d->rng_stream = ctx->next_rng_stream++;
    auto A = [&](float** p, size_t cnt) { return ctx_alloc(ctx, cnt * sizeof(float), (void**)p, true); };
    int rc = A(&d->P, np);
    if (rc == ILSX_OK) rc = A(&d->G, np);
    if (rc == ILSX_OK) rc = A(&d->M, np);
    if (rc == ILSX_OK) rc = A(&d->V, np);
    if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(DiscScalars), (void**)&d->scal);
    if (rc == ILSX_OK) rc = A(&d->X, rows * d->D);
    if (rc == ILSX_OK) rc = A(&d->eps_used, (size_t)cfg->max_batch);
    if (rc == ILSX_OK) rc = A(&N.rmean, (size_t)N.nblk * H);
    if (rc == ILSX_OK) rc = A(&N.rvar, (size_t)N.nblk * H);
    N.P = d->P; N.G = d->G; N.M = d->M; N.V = d->V;
    DbnWs& W = b->W;
    for (int l = 0; l < N.nblk && rc == ILSX_OK; ++l) {
      float** mats[] = {&W.ch[l], &W.ah[l], &W.h[l], &W.p[l], &W.gch[l], &W.gah[l], &W.gh[l], &W.gp[l], &W.uh[l], &W.uy[l], &W.uah[l], &W.tt[l], &W.ua[l],
                        &W.ybar[l], &W.ahbar[l]};
      for (float** m : mats) if (rc == ILSX_OK) rc = A(m, rows * H);
      float** vecs[] = {&W.s[l], &W.gs[l], &W.m2[l], &W.sbar[l]};
      for (float** v : vecs) if (rc == ILSX_OK) rc = A(v, H);
    }
    if (rc == ILSX_OK) rc = A(&W.t0, rows * wd);
    if (rc == ILSX_OK) rc = A(&W.t1, rows * wd);
    if (rc == ILSX_OK) rc = A(&W.gt0, rows * wd);
    if (rc == ILSX_OK) rc = A(&W.gt1, rows * wd);
    if (rc == ILSX_OK) rc = A(&W.bstat, (size_t)2 * N.nblk * 2 * H);
    float** rowv[] = {&W.logit, &W.dlogit, &W.gate, &W.ce_row, &W.correct, &W.gp_row, &b->lg};
    for (float** v : rowv) if (rc == ILSX_OK) rc = A(v, rows);
    if (rc == ILSX_OK) rc = A(&b->stats3, 4);
    if (rc == ILSX_OK) rc = A(&b->xcat, rows * d->D);
    if (rc == ILSX_OK) {   // running_var starts at 1 (torch.nn.BatchNorm1d), running_mean at 0
      std::vector<float> ones((size_t)N.nblk * H, 1.0f);
      hipError_t e = hipMemcpyAsync(N.rvar, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) { ilsx_set_err("ilsx_disc_create: %s", hipGetErrorString(e)); rc = ILSX_ERR_HIP; }
    }
    if (rc != ILSX_OK) { ilsx_disc_destroy(d); return rc; }   // releases every buffer allocated so far (the destroy path takes nulls)
    W.X = d->X; W.XH = d->X + 2 * (size_t)cfg->max_batch * d->D;   // re-pointed per step (the interpolates follow the 2B stacked rows)
    *out = d;
    return ILSX_OK;
  }
  ilsx_mlp_cfg mc = {d->D, d->nblk, cfg->hid_dim, 1, 1, cfg->hid_act};
  int rc = net_layout_build(mc, &d->L);
  if (rc != ILSX_OK) { delete d; return rc; }   // nothing allocated yet
  d->cs = (getenv("ILSX_NO_SPLIT") || d->nblk != 2) ? 1 : mlp2_split_factor(2, cfg->hid_dim);
  d->rng_stream = ctx->next_rng_stream++;
  const size_t n = d->L.n_int, B = (size_t)cfg->max_batch, H = (size_t)cfg->hid_dim, KP = (size_t)d->L.KP;
  auto A = [&](float** p, size_t cnt) { return ctx_alloc(ctx, cnt * sizeof(float), (void**)p, true); };
  rc = A(&d->P, n);
  if (rc == ILSX_OK) rc = A(&d->G, n);
  if (rc == ILSX_OK) rc = A(&d->M, n);
  if (rc == ILSX_OK) rc = A(&d->V, n);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(DiscScalars), (void**)&d->scal);
  if (rc == ILSX_OK) rc = A(&d->X, 3 * B * d->D);
  if (rc == ILSX_OK) rc = A(&d->xs, 4 * B * KP);
  if (rc == ILSX_OK) rc = A(&d->hs0, 4 * B * H);
  if (rc == ILSX_OK) rc = A(&d->hs1, 3 * B * H);
  if (rc == ILSX_OK) rc = A(&d->A2, 4 * B * H);
  if (rc == ILSX_OK) rc = A(&d->A1, 4 * B * H);
  if (rc == ILSX_OK) rc = A(&d->dhead, 3 * B);
  if (rc == ILSX_OK) rc = A(&d->raw, (size_t)d->cs * 3 * B);
  if (rc == ILSX_OK) rc = A(&d->ce_row, 2 * B);
  if (rc == ILSX_OK) rc = A(&d->correct, 2 * B);
  if (rc == ILSX_OK) rc = A(&d->gp_row, B);
  if (rc == ILSX_OK) rc = A(&d->eps_used, B);
  if (d->nblk != 2) {
    for (int l = 0; l < d->nblk && rc == ILSX_OK; ++l) {
      rc = A(&d->hsb[l], 4 * B * H);
      if (rc == ILSX_OK) rc = A(&d->Ab[l], 4 * B * H);
      if (rc == ILSX_OK) rc = A(&d->zb[l], B * H);
    }
    if (rc == ILSX_OK) rc = A(&d->given, 3 * B);
    if (rc == ILSX_OK) rc = A(&d->gdx, B * 64);
    if (rc == ILSX_OK) rc = A(&d->dhead4, 4 * B);
  }
  if (rc == ILSX_OK) rc = disc_refresh(d);
  if (rc != ILSX_OK) { ilsx_disc_destroy(d); return rc; }
  *out = d;
  return ILSX_OK;
}

extern "C" int ilsx_disc_destroy(ilsx_disc* d) {
  if (!d) return ILSX_OK;
  void* ps[] = {d->P, d->G, d->M, d->V, d->scal, d->X, d->xs, d->hs0, d->hs1, d->A2, d->A1, d->dhead, d->raw, d->ce_row,
                d->correct, d->gp_row, d->eps_used};
  for (void* p : ps) ctx_free(d->ctx, p);
  for (int l = 0; l < ILSX_MAX_HID; ++l) { if (d->hsb[l]) ctx_free(d->ctx, d->hsb[l]); if (d->Ab[l]) ctx_free(d->ctx, d->Ab[l]); if (d->zb[l]) ctx_free(d->ctx, d->zb[l]); }
  if (d->given) ctx_free(d->ctx, d->given);
  if (d->gdx) ctx_free(d->ctx, d->gdx);
  if (d->dhead4) ctx_free(d->ctx, d->dhead4);
  if (d->snap) ctx_free(d->ctx, d->snap);
  if (d->bn) {
    DiscBn* b = d->bn;
    DbnWs& W = b->W;
    for (int l = 0; l < b->N.nblk; ++l)
      for (float* p : {W.ch[l], W.ah[l], W.h[l], W.p[l], W.gch[l], W.gah[l], W.gh[l], W.gp[l], W.gs[l], W.uh[l], W.uy[l], W.uah[l], W.tt[l], W.ua[l], W.ybar[l],
                       W.ahbar[l], W.s[l], W.m2[l], W.sbar[l]}) ctx_free(d->ctx, p);
    for (float* p : {W.t0, W.t1, W.gt0, W.gt1, W.bstat, W.logit, W.dlogit, W.gate, W.ce_row, W.correct, W.gp_row, b->lg, b->stats3, b->xcat, b->N.rmean, b->N.rvar}) ctx_free(d->ctx, p);
    delete b;
  }
  delete d;
  return ILSX_OK;
}

extern "C" int ilsx_disc_num_params(const ilsx_disc* d, size_t* out) {
  if (!d || !out) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  *out = d->L.n_flat;
  return ILSX_OK;
}
// use_bn: the flat ABI order is the storage order — a plain copy
static int bn_copy(ilsx_disc* d, float* dev, float* host, size_t n, size_t want, bool to_dev) {
  if (n != want) ILSX_FAIL(ILSX_ERR_ARG, "count %zu != expected %zu", n, want);
  HIPCHK(hipMemcpyAsync(to_dev ? (void*)dev : (void*)host, to_dev ? (const void*)host : (const void*)dev, n * sizeof(float),
                        to_dev ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, d->ctx->stream));
  HIPCHK(hipStreamSynchronize(d->ctx->stream));
  return ILSX_OK;
}
extern "C" int ilsx_disc_set_params(ilsx_disc* d, const float* src, size_t n) {
  if (!d || !src) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->ctx->device));
  if (d->bn) return bn_copy(d, d->P, const_cast<float*>(src), n, d->L.n_flat, true);
  return net_upload_flat(d->ctx, d->L, d->P, src, n, 0);
}
extern "C" int ilsx_disc_get_params(ilsx_disc* d, float* dst, size_t n) {
  if (!d || !dst) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->ctx->device));
  if (d->bn) return bn_copy(d, d->P, dst, n, d->L.n_flat, false);
  return net_download_flat(d->ctx, d->L, d->P, dst, n, 0);
}
extern "C" int ilsx_disc_get_grads(ilsx_disc* d, float* dst, size_t n) {
  if (!d || !dst) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  HIPCHK(hipSetDevice(d->ctx->device));
  if (d->bn) return bn_copy(d, d->G, dst, n, d->L.n_flat, false);
  return net_download_flat(d->ctx, d->L, d->G, dst, n, 0);
}
extern "C" int ilsx_disc_get_bn_stats(ilsx_disc* d, float* rm_host, float* rv_host, size_t n) {
  if (!d || !rm_host || !rv_host) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_get_bn_stats: NULL argument");
  if (!d->bn) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_disc_get_bn_stats: this discriminator has no batch norm (use_bn = 0)");
  HIPCHK(hipSetDevice(d->ctx->device));
  const size_t want = (size_t)d->bn->N.nblk * d->bn->N.H;
  ILSX_TRY(bn_copy(d, d->bn->N.rmean, rm_host, n, want, false));
  return bn_copy(d, d->bn->N.rvar, rv_host, n, want, false);
}
extern "C" int ilsx_disc_set_bn_stats(ilsx_disc* d, const float* rm_host, const float* rv_host, size_t n) {
  if (!d || !rm_host || !rv_host) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_set_bn_stats: NULL argument");
  if (!d->bn) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_disc_set_bn_stats: this discriminator has no batch norm (use_bn = 0)");
  HIPCHK(hipSetDevice(d->ctx->device));
  const size_t want = (size_t)d->bn->N.nblk * d->bn->N.H;
  ILSX_TRY(bn_copy(d, d->bn->N.rmean, const_cast<float*>(rm_host), n, want, true));
  return bn_copy(d, d->bn->N.rvar, const_cast<float*>(rv_host), n, want, true);
}

// disc_optimizer state (adv_irl.py:75-77) for snapshots / resume: Adam moments in the flat ABI layout, step count, Philox counter
static int disc_opt(ilsx_disc* d, bool set, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  if (!d || !m_host || !v_host) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_*_opt: NULL argument");
  HIPCHK(hipSetDevice(d->ctx->device));
  hipStream_t st = d->ctx->stream;
  if (d->bn) {   // natural layout; the Adam step count is the host's
    ILSX_TRY(bn_copy(d, d->M, m_host, n, d->L.n_flat, set));
    ILSX_TRY(bn_copy(d, d->V, v_host, n, d->L.n_flat, set));
    if (meta) { if (set) { d->bn->t = (int)meta->t; d->step_ctr = meta->rng_step; } else { meta->t = d->bn->t; meta->rng_step = d->step_ctr; meta->n_train_steps = 0; } }
    return ILSX_OK;
  }
  if (set) { ILSX_TRY(net_upload_flat(d->ctx, d->L, d->M, m_host, n, 0)); ILSX_TRY(net_upload_flat(d->ctx, d->L, d->V, v_host, n, 0)); }
  else { ILSX_TRY(net_download_flat(d->ctx, d->L, d->M, m_host, n, 0)); ILSX_TRY(net_download_flat(d->ctx, d->L, d->V, v_host, n, 0)); }
  if (!meta) return ILSX_OK;
  DiscScalars h;
  HIPCHK(hipMemcpyAsync(&h, d->scal, sizeof h, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!set) { meta->t = h.t; meta->rng_step = d->step_ctr; meta->n_train_steps = 0; return ILSX_OK; }
  h.t = (int)meta->t;
  d->step_ctr = meta->rng_step;
  HIPCHK(hipMemcpyAsync(d->scal, &h, sizeof h, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return disc_refresh(d);
}
extern "C" int ilsx_disc_get_opt(ilsx_disc* d, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta) {
  return disc_opt(d, false, m_host, v_host, n, meta);
}
extern "C" int ilsx_disc_set_opt(ilsx_disc* d, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta) {
  ilsx_opt_meta m2; if (meta) m2 = *meta;
  return disc_opt(d, true, const_cast<float*>(m_host), const_cast<float*>(v_host), n, meta ? &m2 : nullptr);
}

// shared forward over `rows` rows of X (row stride D): raw head partials -> d->raw, optional activation saves
static int disc_forward(ilsx_disc* d, const float* x0, int d0, int s0, const float* x1, int d1, int s1, int rows, bool save) {
  FwdArgs A;
  memset(&A, 0, sizeof A);
  A.rows = rows; A.ntasks = 1; A.seed = d->ctx->seed; A.part_stride = 3 * d->cfg.max_batch;
  FwdTask& t = A.t[0];
  t.net = net_view(d->L, d->P);
  t.x0 = x0; t.d0 = d0; t.s0 = s0; t.x1 = x1; t.d1 = d1; t.s1 = s1;
  if (save && d->nblk == 2) { t.xsave = d->xs; t.hsave[0] = d->hs0; t.hsave[1] = d->hs1; }
  if (save && d->nblk != 2) { t.xsave = d->xs; for (int l = 0; l < d->nblk; ++l) t.hsave[l] = d->hsb[l]; }
  t.head = HEAD_RAW;
  if (d->cs > 1) t.part = d->raw; else t.out = d->raw;
  return launch_fwd(d->ctx, A, d->cfg.hid_dim, d->cfg.hid_act, d->L.KP, d->cs);
}

static int disc_build_jobs(ilsx_disc* d, int B) {
  if (d->jobs_B == B) return ILSX_OK;
  const int H = d->cfg.hid_dim, KP = d->L.KP, gp = d->cfg.use_grad_pen ? 1 : 0;
  const int rows_h = gp ? 4 * B : 2 * B, bias_h = gp ? 3 * B : 2 * B, rows_o = gp ? 3 * B : 2 * B;
  const NetLayout& L = d->L;
  memset(&d->jobs, 0, sizeof d->jobs);
  ILSX_TRY(dw_table_add(&d->jobs, d->A1, H, H, d->xs, KP, KP, d->G + L.off_W[0], nullptr, KP, d->G + L.off_b[0],
                        DW_OUT_PACK_F, rows_h, bias_h));
  ILSX_TRY(dw_table_add(&d->jobs, d->A2, H, H, d->hs0, H, H, d->G + L.off_W[1], d->G + L.off_Wb[1], H, d->G + L.off_b[1],
                        DW_OUT_PACK_FB, rows_h, bias_h));
  ILSX_TRY(dw_table_add(&d->jobs, d->dhead, 1, 1, d->hs1, H, H, d->G + L.off_Wh, nullptr, H, d->G + L.off_bh,
                        DW_OUT_NATURAL, rows_o, 2 * B));
  d->jobs_B = B;
  return ILSX_OK;
}

// ================================================================================================ num_layer_blocks = 1 or 3
// The same loss and the same hand-derived double backward as k_disc_bwd (oracle/disc.py train_step_blocks) for any
// depth, as a chain of small launches — a completeness path (the reference's specs all use 2 blocks), written for clarity, not speed:
//   k_disc_head_rows   per row: CE terms, dlogit through the clamp gate ; the "given" head gradients of the two backward launches
//   generic backward   CE rows: delta_l -> Ab[l] rows < 2B ;  GP rows (given = gate): u_l -> Ab[l] rows 3B..4B, dD/dx -> gdx
//   k_disc_gp_rows     |dD/dx|, the penalty value, g-bar = dGP/dg -> xs rows 3B..4B
//   k_disc_lin_layer   UP the net, l = 0..L-1: u-bar_l = W_l x_{l-1}; v-bar_l = phi'_l u-bar_l -> hsb[l] rows 3B..4B; z-bar_l = (phi''/phi')_l u_l u-bar_l
//   k_disc_inj_layer   DOWN, l = L-2..0: delta''_l = z-bar_l + phi'_l (delta''_{l+1} W_{l+1}) -> Ab[l] rows 2B..3B
//   k_mlp_bwd_dw       per layer ONE contraction over the 4B stacked rows (bias from the first 3B) + the head's, Adam in the epilogue
__global__ void k_disc_head_rows(PartVal raw, int B, int world, int use_gp, float clamp, float* __restrict__ ce_row, float* __restrict__ correct,
                                 float* __restrict__ given, float* __restrict__ dhead4) {
  const int gr = blockIdx.x * blockDim.x + threadIdx.x, rows = use_gp ? 3 * B : 2 * B;
  if (gr >= 4 * B) return;
  if (gr >= rows) { dhead4[gr] = (use_gp && gr >= 3 * B) ? 1.0f : 0.0f; return; }   // rows 3B..4B: the "ones" rows that carry v-bar_L into dw
  const float r = raw.get(gr);
  const float gate = (r >= -clamp && r <= clamp) ? 1.0f : 0.0f;
  const float l = fminf(fmaxf(r, -clamp), clamp);
  if (gr < 2 * B) {
    const float t = gr < B ? 1.0f : 0.0f;
    const float dl = gate * (1.0f / (1.0f + expf(-l)) - t) / (float)(2 * B * world);
    ce_row[gr] = fmaxf(l, 0.0f) - l * t + log1pf(expf(-fabsf(l)));
    correct[gr] = ((l > 0.0f) == (t > 0.5f)) ? 1.0f : 0.0f;
    given[gr] = dl; dhead4[gr] = dl;
  } else {
    given[gr] = gate; dhead4[gr] = 0.0f;   // GP rows: the u chain starts from the gate; their forward activations feed no head gradient
  }
}
__global__ void k_disc_gp_rows(const float* __restrict__ gdx, int B, int world, int D, int KP, float gp_w, float* __restrict__ gbar /* xs rows 3B.. */,
                               float* __restrict__ gp_row) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= B) return;
  float sq = 0.0f;
  for (int c = 0; c < D; ++c) { const float g = gdx[(size_t)r * D + c]; sq += g * g; }
  const float n = sqrtf(sq);
  const float coef = n > 0.0f ? gp_w / (float)(B * world) * 2.0f * (n - 1.0f) / n : 0.0f;   // adv_irl.py:201-202; norm backward is 0 at 0
  for (int c = 0; c < KP; ++c) gbar[(size_t)r * KP + c] = c < D ? coef * gdx[(size_t)r * D + c] : 0.0f;
  gp_row[r] = (n - 1.0f) * (n - 1.0f);
}
// one thread per (row, output column n)
template <int ACT>
__global__ void k_disc_lin_layer(const float* __restrict__ W, int K, int ldw, const float* __restrict__ xin, int ldx, const float* __restrict__ hgp,
                                 const float* __restrict__ u, int H, int B, float* __restrict__ vbar, float* __restrict__ zbar) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * H) return;
  const int r = e / H, nn = e - r * H;
  float ub = 0.0f;
  for (int k = 0; k < K; ++k) ub = fmaf(W[pack_f(nn, k, ldw)], xin[(size_t)r * ldx + k], ub);
  const float h = hgp[(size_t)r * H + nn];
  vbar[(size_t)r * H + nn] = act_grad_from_out<ACT>(h) * ub;
  zbar[(size_t)r * H + nn] = ACT == ACT_TANH ? -2.0f * h * u[(size_t)r * H + nn] * ub : 0.0f;
}
// one thread per (row, column k of layer l): delta''_l = z-bar_l + phi'_l * sum_n delta''_{l+1}[n] W_{l+1}[n][k]
template <int ACT>
__global__ void k_disc_inj_layer(const float* __restrict__ Wnext, int H, const float* __restrict__ dnext, const float* __restrict__ zbar,
                                 const float* __restrict__ hgp, int B, float* __restrict__ dout) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * H) return;
  const int r = e / H, k = e - r * H;
  float s = 0.0f;
  for (int nn = 0; nn < H; ++nn) s = fmaf(dnext[(size_t)r * H + nn], Wnext[pack_f(nn, k, H)], s);
  dout[(size_t)r * H + k] = zbar[(size_t)r * H + k] + act_grad_from_out<ACT>(hgp[(size_t)r * H + k]) * s;
}

// Weight gradients + Adam(lr, betas = (disc_momentum, 0.999)).  One rank: Adam in the epilogue of the weight-gradient tiles (kernels.h AdamFuse:
// same expressions as k_adam_polyak, one launch and one pass over the arena less).  Split run (cfg.grad_world = G, SURVEY section 8e "Disc: same"):
// backward | all-reduce(gradient arena) | Adam as a launch of its own.  ILSX_SPLIT_FORCE=1 (tests): the split path on a one-rank communicator.
static bool disc_is_split(const ilsx_disc* d) {
  return d->cfg.grad_world > 1 || (d->ctx->comm != nullptr && getenv("ILSX_SPLIT_FORCE") != nullptr);
}
static int disc_dw_adam(ilsx_disc* d, int rows) {
  ilsx_ctx* ctx = d->ctx;
  AdamFuse F;
  memset(&F, 0, sizeof F);
  F.on = 1; F.Gbase = d->G; F.P = d->P; F.M = d->M; F.V = d->V; F.T = nullptr;
  F.b1 = d->cfg.disc_momentum; F.b2 = 0.999f; F.eps = 1e-8f; F.tau = 0.f;
  F.step_size = &d->scal->adam_step; F.bc2_sqrt = &d->scal->adam_bc2s;
  if (!disc_is_split(d)) return launch_bwd_dw(ctx, d->jobs, rows, &F);
  if (d->cfg.grad_world > 1 && (!ctx->comm || ctx->comm_n != d->cfg.grad_world) &&
      !(ctx->comm && ctx->comm_n == 1 && getenv("ILSX_SPLIT_FORCE")))   // (tests: a one-rank communicator stands in, the arena holds this rank's share)
    ILSX_FAIL(ILSX_ERR_STATE, "discriminator step: grad_world=%d needs a communicator of that many ranks on the ctx (ilsx_comm_init; found %d)",
              d->cfg.grad_world, ctx->comm ? ctx->comm_n : 0);
  ILSX_TRY(launch_bwd_dw(ctx, d->jobs, rows, nullptr));
  ILSX_TRY(comm_allreduce_sum(ctx, d->G, d->L.n_int));
  AdamArgs A;
  memset(&A, 0, sizeof A);
  A.p = d->P; A.g = d->G; A.m = d->M; A.v = d->V; A.tgt = nullptr; A.n = (int)d->L.n_int;
  A.b1 = F.b1; A.b2 = F.b2; A.eps = F.eps; A.tau = 0.f; A.step_size = F.step_size; A.bc2_sqrt = F.bc2_sqrt;
  return launch_adam(ctx, A);
}

static int disc_step_blocks(ilsx_disc* d, int B, ilsx_disc_stats* stats) {
  ilsx_ctx* ctx = d->ctx;
  hipStream_t st = ctx->stream;
  const int gp = d->cfg.use_grad_pen ? 1 : 0, rows = gp ? 3 * B : 2 * B, H = d->cfg.hid_dim, L = d->nblk, KP = d->L.KP, act = d->cfg.hid_act;
  const size_t sB = (size_t)B;
  if (d->jobs_B != B) {   // row-stacked weight-gradient jobs (see the table above)
    memset(&d->jobs, 0, sizeof d->jobs);
    const int rows_h = gp ? 4 * B : 2 * B, bias_h = gp ? 3 * B : 2 * B;
    for (int l = 0; l < L; ++l)
      ILSX_TRY(dw_table_add(&d->jobs, d->Ab[l], H, H, l == 0 ? d->xs : d->hsb[l - 1], l == 0 ? KP : H, l == 0 ? KP : H, d->G + d->L.off_W[l],
                            l > 0 ? d->G + d->L.off_Wb[l] : nullptr, l == 0 ? KP : H, d->G + d->L.off_b[l], l > 0 ? DW_OUT_PACK_FB : DW_OUT_PACK_F,
                            rows_h, bias_h));
    ILSX_TRY(dw_table_add(&d->jobs, d->dhead4, 1, 1, d->hsb[L - 1], H, H, d->G + d->L.off_Wh, nullptr, H, d->G + d->L.off_bh, DW_OUT_NATURAL,
                          rows_h, 2 * B));
    d->jobs_B = B;
  }
  ILSX_TRY(disc_forward(d, d->X, d->D, d->D, nullptr, 0, 0, rows, true));
  hipLaunchKernelGGL(k_disc_head_rows, dim3((4 * B + 255) / 256), dim3(256), 0, st, d->pv(), B, d->cfg.grad_world, gp, d->cfg.clamp_magnitude, d->ce_row, d->correct,
                     d->given, d->dhead4);
  HIPCHK(hipGetLastError());
  for (int pass = 0; pass < (gp ? 2 : 1); ++pass) {   // 0: the CE rows ; 1: the interpolates (given = gate: the u chain + dD/dx)
    BwdArgs A;
    memset(&A, 0, sizeof A);
    const size_t r0 = pass ? 2 * sB : 0;
    A.rows = pass ? B : 2 * B; A.ntasks = 1; A.inv_B = 1.0f;
    BwdTask& b = A.t[0];
    b.net = net_view(d->L, d->P);
    for (int l = 0; l < L; ++l) { b.hsave[l] = d->hsb[l] + r0 * H; b.dsave[l] = d->Ab[l] + (pass ? 3 * sB : 0) * H; }
    b.loss = LOSS_GIVEN; b.given = d->given + r0;
    if (pass) { b.dx = d->gdx; b.dx_col0 = 0; b.dx_cols = d->D; }
    ILSX_TRY(launch_bwd_dx(ctx, A, H, act));
  }
  if (gp) {
    hipLaunchKernelGGL(k_disc_gp_rows, dim3((B + 255) / 256), dim3(256), 0, st, (const float*)d->gdx, B, d->cfg.grad_world, d->D, KP, d->cfg.grad_pen_weight,
                       d->xs + 3 * sB * KP, d->gp_row);
    const dim3 grid((unsigned)((sB * H + 255) / 256)), block(256);
    for (int l = 0; l < L; ++l) {
      const float* W = d->P + d->L.off_W[l];
      const float* xin = l == 0 ? d->xs + 3 * sB * KP : d->hsb[l - 1] + 3 * sB * H;
      float* zout = l == L - 1 ? d->Ab[l] + 2 * sB * H : d->zb[l];   // delta''_L = z-bar_L
      const int K = l == 0 ? KP : H;
      if (act == ILSX_ACT_TANH)
        hipLaunchKernelGGL(k_disc_lin_layer<ACT_TANH>, grid, block, 0, st, W, K, K, xin, K, (const float*)(d->hsb[l] + 2 * sB * H),
                           (const float*)(d->Ab[l] + 3 * sB * H), H, B, d->hsb[l] + 3 * sB * H, zout);
      else
        hipLaunchKernelGGL(k_disc_lin_layer<ACT_RELU>, grid, block, 0, st, W, K, K, xin, K, (const float*)(d->hsb[l] + 2 * sB * H),
                           (const float*)(d->Ab[l] + 3 * sB * H), H, B, d->hsb[l] + 3 * sB * H, zout);
    }
    for (int l = L - 2; l >= 0; --l) {
      const float* Wn = d->P + d->L.off_W[l + 1];
      if (act == ILSX_ACT_TANH)
        hipLaunchKernelGGL(k_disc_inj_layer<ACT_TANH>, grid, block, 0, st, Wn, H, (const float*)(d->Ab[l + 1] + 2 * sB * H), (const float*)d->zb[l],
                           (const float*)(d->hsb[l] + 2 * sB * H), B, d->Ab[l] + 2 * sB * H);
      else
        hipLaunchKernelGGL(k_disc_inj_layer<ACT_RELU>, grid, block, 0, st, Wn, H, (const float*)(d->Ab[l + 1] + 2 * sB * H), (const float*)d->zb[l],
                           (const float*)(d->hsb[l] + 2 * sB * H), B, d->Ab[l] + 2 * sB * H);
    }
    HIPCHK(hipGetLastError());
  }
  ILSX_TRY(disc_dw_adam(d, gp ? 4 * B : 2 * B));
  hipLaunchKernelGGL(k_disc_tail, dim3(1), dim3(256), 0, st, d->scal, d->ce_row, d->correct, d->gp_row, B, gp, d->cfg.disc_lr, d->cfg.disc_momentum,
                     0.999f);
  HIPCHK(hipGetLastError());
  if (stats) {
    DiscScalars h;
    HIPCHK(hipMemcpyAsync(&h, d->scal, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    stats->ce_loss = h.ce_loss; stats->grad_pen = h.grad_pen; stats->accuracy = h.accuracy;
  }
  return ILSX_OK;
}

static int disc_step_after_prep(ilsx_disc* d, int B, ilsx_disc_stats* stats);

extern "C" int ilsx_disc_train_step(ilsx_disc* d, const float* exp_obs, const float* exp_act, const float* pol_obs,
                                    const float* pol_act, int B, const float* eps, ilsx_disc_stats* stats) {
  if (!d || !exp_obs || !exp_act || !pol_obs || !pol_act) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_train_step: NULL argument");
  if (B < 1 || B > d->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch=%d", B, d->cfg.max_batch);
  ilsx_ctx* ctx = d->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const int gp = d->cfg.use_grad_pen ? 1 : 0;
  {
    const int tot = B * d->D;
    hipLaunchKernelGGL(k_disc_prep, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, exp_obs, exp_act, pol_obs, pol_act, eps,
                       B, d->o, d->a, gp, ctx->seed, d->rng_stream, ++d->step_ctr, d->X, d->eps_used);
    HIPCHK(hipGetLastError());
  }
  return disc_step_after_prep(d, B, stats);
}

// the discriminator step of the adversarial-IRL loop: both batches drawn from the rings inside the prep launch (k_disc_prep_rings)
static int disc_train_step_from_rings(ilsx_disc* d, ilsx_replay* expert_rb, ilsx_replay* policy_rb, int B, ilsx_disc_stats* stats) {
  if (B < 1 || B > d->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "batch %d not in 1..max_batch=%d", B, d->cfg.max_batch);
  if (expert_rb->size < 1 || policy_rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_advirl_train: a replay buffer is empty");
  ilsx_ctx* ctx = d->ctx;
  const int gp = d->cfg.use_grad_pen ? 1 : 0, tot = B * d->D;
  ILSX_TRY(replay_flush_state(expert_rb));
  ILSX_TRY(replay_flush_state(policy_rb));
  // the counters advance exactly as ilsx_replay_sample(expert) ; ilsx_replay_sample(policy) would advance them
  const DiscRing E = {expert_rb->data, expert_rb->dstate, expert_rb->seed, expert_rb->rng_stream, expert_rb->rec, ++expert_rb->sample_ctr};
  const DiscRing P = {policy_rb->data, policy_rb->dstate, policy_rb->seed, policy_rb->rng_stream, policy_rb->rec, ++policy_rb->sample_ctr};
  hipLaunchKernelGGL(k_disc_prep_rings, dim3((tot + 255) / 256), dim3(256), 0, ctx->stream, E, P, B, d->cfg.obs_dim, policy_rb->a,
                     d->cfg.state_only ? 1 : 0, gp, ctx->seed, d->rng_stream, ++d->step_ctr, d->X, d->eps_used);
  HIPCHK(hipGetLastError());
  return disc_step_after_prep(d, B, stats);
}

// use_bn: d->X holds [expert ; policy ; interpolates] (k_disc_prep / k_disc_prep_rings), row stride D
// (Round 6 ran the whole step as ONE launch on one XCD — the phases called inside a kernel, an arrival-counter barrier between them: correct
//  and 3-4x slower than the launches, profiles/r06_discbn_one_launch.txt; removed.)
static int discbn_step(ilsx_disc* d, int B, ilsx_disc_stats* stats) {
  DiscBn* b = d->bn;
  if (3 * B > b->rows) ILSX_FAIL(ILSX_ERR_ARG, "batch %d exceeds the workspace", B);
  if (B < 2) ILSX_FAIL(ILSX_ERR_ARG, "a BatchNorm discriminator needs at least 2 rows per class (torch raises on a single-row training batch)");
  DbnLaunch L{d->ctx->stream};
  DbnWs W = b->W;
  W.X = d->X; W.XH = d->X + 2 * (size_t)B * d->D;
  const int gp = d->cfg.use_grad_pen ? 1 : 0;
  dbn_backward(L, b->N, W, B, gp, d->cfg.grad_pen_weight);
  dbn_finish(L, b->N, W, B, gp, b->stats3, d->cfg.disc_lr, d->cfg.disc_momentum, ++b->t);
  HIPCHK(hipGetLastError());
  if (stats) {
    float h3[3];
    HIPCHK(hipMemcpyAsync(h3, b->stats3, sizeof h3, hipMemcpyDeviceToHost, d->ctx->stream));
    HIPCHK(hipStreamSynchronize(d->ctx->stream));
    stats->ce_loss = h3[0]; stats->accuracy = h3[1]; stats->grad_pen = h3[2];
  }
  return ILSX_OK;
}

static int disc_step_after_prep(ilsx_disc* d, int B, ilsx_disc_stats* stats) {
  if (d->bn) return discbn_step(d, B, stats);
  if (d->nblk != 2) return disc_step_blocks(d, B, stats);
  ilsx_ctx* ctx = d->ctx;
  ILSX_TRY(disc_build_jobs(d, B));
  const int gp = d->cfg.use_grad_pen ? 1 : 0, rows = gp ? 3 * B : 2 * B, H = d->cfg.hid_dim;
  ILSX_TRY(disc_forward(d, d->X, d->D, d->D, nullptr, 0, 0, rows, true));
  {
    DiscBwdArgs A;
    memset(&A, 0, sizeof A);
    A.net = net_view(d->L, d->P);
    A.B = B; A.rows = rows; A.use_gp = gp; A.D = d->D; A.world = d->cfg.grad_world;
    A.clamp = d->cfg.clamp_magnitude; A.gp_w = d->cfg.grad_pen_weight;
    A.raw = d->pv();
    A.hs0 = d->hs0; A.hs1 = d->hs1; A.xs = d->xs; A.A2 = d->A2; A.A1 = d->A1; A.dhead = d->dhead;
    A.ce_row = d->ce_row; A.correct = d->correct; A.gp_row = d->gp_row;
    const size_t lds = sizeof(float) * (3 * 16 * (H + ILSX_LDS_PAD) + 16 * 64 + 16 * 4);
    dim3 grid((rows + 15) / 16), block(4 * H);
    ProfScope ps(ctx, ILSX_K_DISC_BWD);
    if (d->cfg.hid_act == ILSX_ACT_TANH) {
      if (H == 64) ILSX_LAUNCH(ps, (k_disc_bwd<64, ACT_TANH>), grid, block, lds, ctx->stream, A);
      else if (H == 128) ILSX_LAUNCH(ps, (k_disc_bwd<128, ACT_TANH>), grid, block, lds, ctx->stream, A);
      else ILSX_LAUNCH(ps, (k_disc_bwd<256, ACT_TANH>), grid, block, lds, ctx->stream, A);
    } else {
      if (H == 64) ILSX_LAUNCH(ps, (k_disc_bwd<64, ACT_RELU>), grid, block, lds, ctx->stream, A);
      else if (H == 128) ILSX_LAUNCH(ps, (k_disc_bwd<128, ACT_RELU>), grid, block, lds, ctx->stream, A);
      else ILSX_LAUNCH(ps, (k_disc_bwd<256, ACT_RELU>), grid, block, lds, ctx->stream, A);
    }
    HIPCHK(hipGetLastError());
  }
  ILSX_TRY(disc_dw_adam(d, rows));
  hipLaunchKernelGGL(k_disc_tail, dim3(1), dim3(256), 0, ctx->stream, d->scal, d->ce_row, d->correct, d->gp_row, B, gp,
                     d->cfg.disc_lr, d->cfg.disc_momentum, 0.999f);
  HIPCHK(hipGetLastError());
  if (stats) {
    DiscScalars h;
    HIPCHK(hipMemcpyAsync(&h, d->scal, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    stats->ce_loss = h.ce_loss; stats->grad_pen = h.grad_pen; stats->accuracy = h.accuracy;
  }
  return ILSX_OK;
}

extern "C" int ilsx_disc_reward(ilsx_disc* d, const float* obs, const float* act, int n, int mode, int has_min, float rmin,
                                int has_max, float rmax, float* rew, float* logits) {
  if (!d || !obs || !act || (!rew && !logits)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_disc_reward: NULL argument");
  if (n < 1 || n > 3 * d->cfg.max_batch) ILSX_FAIL(ILSX_ERR_ARG, "n=%d not in 1..3*max_batch", n);
  if (mode < 0 || mode > 3) ILSX_FAIL(ILSX_ERR_ARG, "unknown reward mode %d", mode);
  HIPCHK(hipSetDevice(d->ctx->device));
  if (d->bn) {   // eval mode (adv_irl.py:268-274): the running statistics
    DiscBn* b = d->bn;
    if (n > b->rows) ILSX_FAIL(ILSX_ERR_ARG, "n=%d exceeds the workspace", n);
    DbnLaunch L{d->ctx->stream};
    float* xc = b->xcat;
    const int o = d->o, a = d->a, D = d->D;
    L.par(n * D, [=] __device__(int idx) { const int r = idx / D, k = idx - r * D; xc[idx] = k < o ? obs[(size_t)r * o + k] : act[(size_t)r * a + (k - o)]; });
    dbn_logits_eval(L, b->N, b->W, xc, n, b->lg);
    hipLaunchKernelGGL(k_disc_reward, dim3((n + 255) / 256), dim3(256), 0, d->ctx->stream, PartVal{b->lg, 1, n}, n, d->cfg.clamp_magnitude,
                       mode, has_min, rmin, has_max, rmax, rew, logits);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  ILSX_TRY(disc_forward(d, obs, d->o, d->o, act, d->a, d->a, n, false));
  hipLaunchKernelGGL(k_disc_reward, dim3((n + 255) / 256), dim3(256), 0, d->ctx->stream, d->pv(), n, d->cfg.clamp_magnitude,
                     mode, has_min, rmin, has_max, rmax, rew, logits);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------------
// AdvIRL._do_training (adv_irl.py:126-131): `loops` x { k discriminator steps ; m policy steps whose rewards are
// relabelled by the discriminator (:256-301) }, with every batch drawn on the device from the two HBM replay rings
// (get_batch, :106-113).  One C call per train call instead of ~6 per loop iteration from the host language.
static int advirl_train_once(ilsx_disc* d, ilsx_sac* sac, ilsx_replay* expert_rb, ilsx_replay* policy_rb, int loops,
                             int disc_updates, int policy_updates, int disc_batch, int policy_batch, int mode, int has_min,
                             float rew_clip_min, int has_max, float rew_clip_max, ilsx_disc_stats* disc_stats,
                             ilsx_sac_stats* sac_stats, float* rew_stats4);
// The policy steps of a call after the first run in a window on the agent's merged phase kernels where those fit (ilsx_sac.hip).  Their
// tile-local hand-offs need the GPU to this process; when another process's kernels break them the window reports it at its end and
// the WHOLE call is rolled back — agent and discriminator are checkpointed at entry (two device-to-device copies), the rings' and the
// discriminator's draw counters with them — and run again on one launch per stage, which the agent then keeps.
extern "C" int ilsx_advirl_train(ilsx_disc* d, ilsx_sac* sac, ilsx_replay* expert_rb, ilsx_replay* policy_rb, int loops,
                                 int disc_updates, int policy_updates, int disc_batch, int policy_batch, int mode, int has_min,
                                 float rew_clip_min, int has_max, float rew_clip_max, ilsx_disc_stats* disc_stats,
                                 ilsx_sac_stats* sac_stats, float* rew_stats4) {
  if (!d || !sac || !expert_rb || !policy_rb || loops < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_advirl_train: bad argument");
  HIPCHK(hipSetDevice(d->ctx->device));
  static const bool no_window = getenv("ILSX_ADVIRL_NO_WINDOW") != nullptr;
  const bool checkpoint = !no_window && !d->bn && policy_batch >= 1 && sac_window_may_use_phase(sac, policy_batch);
  const size_t n = d->L.n_int;
  const unsigned long long c_e = expert_rb->sample_ctr, c_p = policy_rb->sample_ctr, c_d = d->step_ctr;
  if (checkpoint) {
    hipStream_t st = d->ctx->stream;
    if (!d->snap) ILSX_TRY(ctx_alloc(d->ctx, (3 * n) * sizeof(float) + sizeof(DiscScalars), (void**)&d->snap, false));
    HIPCHK(hipMemcpyAsync(d->snap, d->P, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->snap + n, d->M, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->snap + 2 * n, d->V, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->snap + 3 * n, d->scal, sizeof(DiscScalars), hipMemcpyDeviceToDevice, st));
    ILSX_TRY(sac_snapshot_take(sac));
  }
  int rc = advirl_train_once(d, sac, expert_rb, policy_rb, loops, disc_updates, policy_updates, disc_batch, policy_batch, mode, has_min,
                             rew_clip_min, has_max, rew_clip_max, disc_stats, sac_stats, rew_stats4);
  if (rc == ILSX_RETRY_WINDOW) {
    if (!checkpoint) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_advirl_train: a window asked for a roll-back without a checkpoint");
    hipStream_t st = d->ctx->stream;
    HIPCHK(hipMemcpyAsync(d->P, d->snap, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->M, d->snap + n, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->V, d->snap + 2 * n, n * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(d->scal, d->snap + 3 * n, sizeof(DiscScalars), hipMemcpyDeviceToDevice, st));
    ILSX_TRY(sac_snapshot_restore(sac));
    expert_rb->sample_ctr = c_e; policy_rb->sample_ctr = c_p; d->step_ctr = c_d;
    rc = advirl_train_once(d, sac, expert_rb, policy_rb, loops, disc_updates, policy_updates, disc_batch, policy_batch, mode, has_min,
                           rew_clip_min, has_max, rew_clip_max, disc_stats, sac_stats, rew_stats4);
    if (rc == ILSX_RETRY_WINDOW) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_advirl_train: the fallback path asked for a retry");
  }
  return rc;
}
static int advirl_train_once(ilsx_disc* d, ilsx_sac* sac, ilsx_replay* expert_rb, ilsx_replay* policy_rb, int loops,
                             int disc_updates, int policy_updates, int disc_batch, int policy_batch, int mode, int has_min,
                             float rew_clip_min, int has_max, float rew_clip_max, ilsx_disc_stats* disc_stats,
                             ilsx_sac_stats* sac_stats, float* rew_stats4) {
  if (disc_batch < 1 || disc_batch > d->cfg.max_batch || policy_batch < 1) ILSX_FAIL(ILSX_ERR_ARG, "batch sizes out of range");
  const int o = d->cfg.obs_dim, a = policy_rb->a;   // a: the env's action width (state_only discriminators have cfg.act_dim == obs_dim)
  const bool so = d->cfg.state_only != 0;
  const int nfe = d->from_expert;
  if (nfe < 0 || nfe > policy_batch) ILSX_FAIL(ILSX_ERR_ARG, "policy_optim_batch_size_from_expert=%d not in 0..policy_batch=%d", nfe, policy_batch);
  if (expert_rb->o != o || policy_rb->o != o || expert_rb->a != a || (!so && a != d->cfg.act_dim))
    ILSX_FAIL(ILSX_ERR_ARG, "replay dims do not match the discriminator");
  bool first_disc = true, first_pol = true, in_window = false;
  static const bool no_window_env = getenv("ILSX_ADVIRL_NO_WINDOW") != nullptr;
  const bool no_window = no_window_env || d->bn != nullptr;   // (the BatchNorm discriminator's state is not part of the window checkpoint)
  struct WindowGuard { ilsx_sac* s; bool* on; ~WindowGuard() { if (*on) sac_window_end(s, /*teardown=*/true); } } window_guard{sac, &in_window};   // error paths
  for (int it = 0; it < loops; ++it) {
    for (int k = 0; k < disc_updates; ++k) {   // adv_irl.py:133-216
      ILSX_TRY(disc_train_step_from_rings(d, expert_rb, policy_rb, disc_batch, first_disc ? disc_stats : nullptr));
      first_disc = false;
    }
    for (int m = 0; m < policy_updates; ++m) {   // adv_irl.py:238-314
      const int npol = policy_batch - nfe;   // adv_irl.py:239-255: torch.cat([rows from the policy buffer, rows from the expert buffer])
      // the batch is drawn straight into the agent's own batch arrays and relabelled there (same stream): no staging copy
      float *bo, *ba, *br, *bd, *bn;
      ILSX_TRY(sac_staged_batch(sac, policy_batch, &bo, &ba, &br, &bd, &bn));
      if (npol > 0) ILSX_TRY(ilsx_replay_sample(policy_rb, npol, nullptr, bo, ba, br, bd, bn, nullptr));
      if (nfe > 0)
        ILSX_TRY(ilsx_replay_sample(expert_rb, nfe, nullptr, bo + (size_t)npol * o, ba + (size_t)npol * a, br + npol, bd + npol,
                                    bn + (size_t)npol * o, nullptr));
      ILSX_TRY(ilsx_disc_reward(d, bo, so ? bn : ba, policy_batch, mode, has_min, rew_clip_min, has_max, rew_clip_max, br, nullptr));
      // the step whose statistics the caller reads (the first one) runs with its own tail; every later step of the call sits in a window
      // (ilsx_sac.hip sac_window_*): tail deferred into the next step's first launch, merged phase kernels where they fit
      if (first_pol || no_window) {
        ILSX_TRY(sac_step_staged(sac, first_pol ? sac_stats : nullptr));
        if (first_pol && !no_window) { ILSX_TRY(sac_window_begin(sac, policy_batch)); in_window = true; }
      } else {
        ILSX_TRY(sac_window_step(sac));
      }
      if (first_pol && rew_stats4) {   // "Disc Rew Mean/Std/Max/Min" of the first relabelled batch (adv_irl.py:303-314)
        std::vector<float> r(policy_batch);
        HIPCHK(hipMemcpyAsync(r.data(), br, (size_t)policy_batch * 4, hipMemcpyDeviceToHost, d->ctx->stream));
        HIPCHK(hipStreamSynchronize(d->ctx->stream));
        double s = 0, ss = 0;
        float mx = r[0], mn = r[0];
        for (float v : r) { s += v; mx = std::max(mx, v); mn = std::min(mn, v); }
        const double mean = s / policy_batch;
        for (float v : r) ss += (v - mean) * (v - mean);
        rew_stats4[0] = (float)mean; rew_stats4[1] = (float)std::sqrt(ss / policy_batch); rew_stats4[2] = mx; rew_stats4[3] = mn;
      }
      first_pol = false;
    }
  }
  if (in_window) { in_window = false; ILSX_TRY(sac_window_end(sac)); }
  return ILSX_OK;
}

extern "C" int ilsx_advirl_set_policy_batch_from_expert(ilsx_disc* d, int n_from_expert) {
  if (!d || n_from_expert < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_advirl_set_policy_batch_from_expert: bad argument");
  d->from_expert = n_from_expert;
  return ILSX_OK;
}

int disc_debug_stream(const void* obj, uint32_t* stream, uint64_t* seed) {   // ilsx_debug_rng_stream (ilsx_sac.hip)
  const ilsx_disc* d = (const ilsx_disc*)obj;
  *stream = d->rng_stream;
  if (seed) *seed = d->ctx->seed;
  return ILSX_OK;
}
