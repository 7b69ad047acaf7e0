// kernels.h — gfx950 (CDNA4, wave64) device code for the ILSwiss hot path.
//
// All GEMM-shaped work runs on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32): results are an
// fmaf chain in fp32, so parity with the reference's fp32 PyTorch path holds to summation order.
// Row tiles are 16 batch rows per workgroup (the MFMA M), 4 waves split the hidden width; hidden
// activations never leave LDS inside a forward / backward-to-input chain.
//
// Fragment maps used below (MI355X guide §3): for D = A(16x4) * B(4x16)
//   lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
//   lane l, register v receives D[row = 4*(l>>4) + v][col = l&15].
// A k16 chunk is issued as 4 MFMAs; lane group g = l>>4 owns k = 4g+s at step s, so each lane
// fetches its four k values with ONE 16-byte load when the operand is k-contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ILSX_LDS_PAD 4
#define ILSX_MAX_HID 3
#define ILSX_MAX_NO 64   // max total head outputs (n_heads*out_dim)

enum { HEAD_RAW = 0, HEAD_TANH_SAMPLE = 1, HEAD_TANH_DET = 2, HEAD_TANH_LOGP_OF_ACT = 3 };
enum { LOSS_GIVEN = 0, LOSS_SAC_CRITIC = 1, LOSS_SAC_ACTORQ = 2, LOSS_SAC_POLICY = 3 };
enum { ACT_RELU = 0, ACT_TANH = 1 };

#define LOG_SIG_MIN (-20.0f)
#define LOG_SIG_MAX (2.0f)
#define TANH_EPS (1e-6f)
#define HALF_LOG_2PI (0.91893853320467274178f)

// View of one network inside a flat fp32 arena (internal layout: first-layer rows padded to KP).
struct NetView {
  float* base;
  int off_W[ILSX_MAX_HID], off_b[ILSX_MAX_HID], ld[ILSX_MAX_HID];
  int off_Wh, off_bh;
  int nhid, H, in_dim, KP, NO;
};

// Device-resident scalars of one SAC agent (everything a captured graph needs to re-read).
struct DevScalars {
  double log_alpha, m_alpha, v_alpha;
  double log_alpha_used;
  float alpha;        // (float)exp(log_alpha): the value tensor ops see (sac_alpha.py:54,166)
  float alpha_used;   // alpha the step just taken used
  int t_q, t_pi, t_alpha;
  int pad0;
  unsigned long long step;  // Philox step counter
  // stats of the step just taken (sac_alpha.py:186-233)
  float qf1_loss, qf2_loss, policy_loss, alpha_loss;
  float q1_mean, q2_mean, log_pi_mean, mu_mean, log_std_mean;
  float pad1;
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: no state to carry between launches.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01_open(uint32_t x) {  // (0,1)
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
// 4 standard normals for (row, quad, stream, step)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t step, uint32_t stream, uint32_t row,
                                               uint32_t quad, float (&z)[4]) {
  uint32_t c[4] = {row, quad, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
  const float r0 = sqrtf(-2.0f * logf(u01_open(c[0]))), r1 = sqrtf(-2.0f * logf(u01_open(c[2])));
  const float t0 = 6.28318530717958647692f * u01_open(c[1]), t1 = 6.28318530717958647692f * u01_open(c[3]);
  z[0] = r0 * cosf(t0); z[1] = r0 * sinf(t0); z[2] = r1 * cosf(t1); z[3] = r1 * sinf(t1);
}

template <int ACT> __device__ __forceinline__ float act_fn(float z) {
  if (ACT == ACT_RELU) return fmaxf(z, 0.0f);
  return tanhf(z);
}
template <int ACT> __device__ __forceinline__ float act_grad_from_out(float h) {
  if (ACT == ACT_RELU) return h > 0.0f ? 1.0f : 0.0f;
  return 1.0f - h * h;
}

// ------------------------------------------------------------------------------------------------
// out[r][n] += sum_k cur[r][k] * W[n][k]   (cur: LDS row tile [16][ldc]; W: global, k-contiguous)
template <int NT>
__device__ __forceinline__ void gemm_rowtile_nk(const float* cur, int ldc, int K, const float* __restrict__ W,
                                                int ldw, int n_base, int li, int g, f32x4 (&acc)[NT]) {
  const float* ap = cur + li * ldc + 4 * g;
  const float* bp = W + (size_t)(n_base + li) * ldw + 4 * g;
  for (int kc = 0; kc < K; kc += 16) {
    const float4 a = *reinterpret_cast<const float4*>(ap + kc);
    float4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const float4*>(bp + (size_t)t * 16 * ldw + kc);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
  }
}

// out[r][c] += sum_n cur[r][n] * W[n][c]   (backward-to-input: contraction over W's ROW index)
template <int NT>
__device__ __forceinline__ void gemm_rowtile_kn(const float* cur, int ldc, int Kn, const float* __restrict__ W,
                                                int ldw, int c_base, int li, int g, f32x4 (&acc)[NT]) {
  const float* ap = cur + li * ldc + 4 * g;
  const float* bp = W + (size_t)(4 * g) * ldw + c_base + li;
  for (int nc = 0; nc < Kn; nc += 16) {
    const float4 a = *reinterpret_cast<const float4*>(ap + nc);
    float b[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) b[t][s] = bp[(size_t)(nc + s) * ldw + 16 * t];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t][0], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t][1], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t][2], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t][3], acc[t], 0, 0, 0);
  }
}

// sum over the 16 lanes of a lane group (lanes sharing l>>4)
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}

// ================================================================================================
// Fused MLP forward over 16-row tiles (Mlp.forward, networks.py:85-101; FlattenMlp cat, :108-115;
// tanh-Gaussian head, policies.py:262-307 + distributions.py:23-28,43-50,74-97).
// grid = (ceil(rows/16), ntasks), block = 256.
struct FwdTask {
  NetView net;
  const float* x0; const float* x1;  // input segments (cat along dim 1)
  int d0, s0, d1, s1;                // dims and row strides
  float* xsave;                      // [rows][KP] zero-padded input (for dW0), nullable
  float* hsave[ILSX_MAX_HID];        // [rows][H] post-activation, nullable
  float* out;                        // raw head outputs [rows][NO], nullable
  int head;                          // HEAD_*
  uint32_t rng_stream;
  const float* eps;                  // [rows][a] explicit N(0,1) or null -> Philox
  const float* act_in;               // HEAD_TANH_LOGP_OF_ACT: actions [rows][a]
  float* eps_save;                   // [rows][a] nullable
  float* action;                     // [rows][a] nullable
  float* logp;                       // [rows] nullable
};
struct FwdArgs {
  FwdTask t[4];
  int rows, ntasks;
  uint64_t seed;
  const DevScalars* scal;  // nullable: step counter for Philox
  uint64_t step_host;      // used when scal == null
};

#ifdef ILSX_KERNEL_IMPL
template <int H, int ACT>
__global__ __launch_bounds__(256) void k_mlp_fwd(const FwdArgs A) {
  constexpr int NT = H / 64;
  constexpr int LDH = H + ILSX_LDS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FwdTask& T = A.t[blockIdx.y];
  const NetView& N = T.net;
  const int KP = N.KP, LDX = KP + ILSX_LDS_PAD, NO = N.NO;
  float* xs = smem;
  float* bufA = xs + 16 * LDX;
  float* bufB = bufA + 16 * LDH;
  float* hout = bufB + 16 * LDH;  // [16][NO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, rows = A.rows;

  // ---- stage the (concatenated, zero-padded) input tile
  for (int e = tid; e < 16 * KP; e += 256) {
    const int r = e / KP, k = e - r * KP, gr = r0 + r;
    float v = 0.0f;
    if (gr < rows) {
      if (k < T.d0) v = T.x0[(size_t)gr * T.s0 + k];
      else if (k < T.d0 + T.d1) v = T.x1[(size_t)gr * T.s1 + (k - T.d0)];
      if (T.xsave) T.xsave[(size_t)gr * KP + k] = v;
    }
    xs[r * LDX + k] = v;
  }
  __syncthreads();

  // ---- hidden layers on the matrix pipe
  const float* cur = xs;
  int K = KP, ldc = LDX;
  for (int l = 0; l < N.nhid; ++l) {
    float* nxt = (l & 1) ? bufB : bufA;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n_base = wave * (16 * NT);
    gemm_rowtile_nk<NT>(cur, ldc, K, N.base + N.off_W[l], N.ld[l], n_base, li, g, acc);
    const float* bias = N.base + N.off_b[l];
    float* hs = T.hsave[l];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = n_base + 16 * t + li;
      const float bv = bias[col];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 4 * g + v;
        const float h = act_fn<ACT>(acc[t][v] + bv);
        nxt[row * LDH + col] = h;
        if (hs && r0 + row < rows) hs[(size_t)(r0 + row) * H + col] = h;
      }
    }
    __syncthreads();
    cur = nxt; K = H; ldc = LDH;
  }

  // ---- heads (NO <= 64 outputs): 16 lanes per row, shuffle-reduced dot products
  {
    const int row = tid >> 4, part = tid & 15;
    const float* hrow = cur + row * ldc;
    const float* Wh = N.base + N.off_Wh;
    const float* bh = N.base + N.off_bh;
    for (int j = 0; j < NO; ++j) {
      float s = 0.0f;
      for (int k = 4 * part; k < K; k += 64) {
        const float4 hv = *reinterpret_cast<const float4*>(hrow + k);
        const float4 wv = *reinterpret_cast<const float4*>(Wh + (size_t)j * H + k);
        s = fmaf(hv.x, wv.x, s); s = fmaf(hv.y, wv.y, s); s = fmaf(hv.z, wv.z, s); s = fmaf(hv.w, wv.w, s);
      }
      s = group16_sum(s);
      if (part == 0) hout[row * ILSX_MAX_NO + j] = s + bh[j];
    }
  }
  __syncthreads();

  // ---- head epilogue
  if (T.out) {
    for (int e = tid; e < 16 * NO; e += 256) {
      const int r = e / NO, j = e - r * NO;
      if (r0 + r < rows) T.out[(size_t)(r0 + r) * NO + j] = hout[r * ILSX_MAX_NO + j];
    }
  }
  if (T.head != HEAD_RAW && tid < 16 && r0 + tid < rows) {
    const int r = tid, gr = r0 + r, a = NO >> 1;
    const float* ho = hout + r * ILSX_MAX_NO;
    const uint64_t step = A.scal ? A.scal->step : A.step_host;
    float lp_quad = 0.f, lp_ls = 0.f, lp_jac = 0.f;
    for (int j0 = 0; j0 < a; j0 += 4) {
      float z4[4] = {0.f, 0.f, 0.f, 0.f};
      if (T.head == HEAD_TANH_SAMPLE && !T.eps) philox_normal4(A.seed, step, T.rng_stream, gr, j0 >> 2, z4);
      for (int jj = 0; jj < 4 && j0 + jj < a; ++jj) {
        const int j = j0 + jj;
        const float mu = ho[j];
        const float ls = fminf(fmaxf(ho[a + j], LOG_SIG_MIN), LOG_SIG_MAX);
        const float sd = expf(ls);
        float e = 0.f, z, act;
        if (T.head == HEAD_TANH_DET) {
          z = mu; act = tanhf(mu);
        } else if (T.head == HEAD_TANH_LOGP_OF_ACT) {
          act = T.act_in[(size_t)gr * a + j];
          z = 0.5f * (logf(1.0f + act + TANH_EPS) - logf(1.0f - act + TANH_EPS));  // distributions.py:85-88
        } else {
          e = T.eps ? T.eps[(size_t)gr * a + j] : z4[jj];
          z = e * sd + mu;        // distributions.py:27
          act = tanhf(z);
        }
        const float dm = mu - z;
        lp_quad += dm * dm / expf(2.0f * ls);                 // distributions.py:45-47
        lp_ls += ls;
        lp_jac += logf(1.0f - act * act + TANH_EPS);          // distributions.py:91-93
        if (T.action) T.action[(size_t)gr * a + j] = act;
        if (T.eps_save) T.eps_save[(size_t)gr * a + j] = e;
      }
    }
    if (T.logp) T.logp[gr] = -0.5f * lp_quad - (lp_ls + HALF_LOG_2PI) - lp_jac;
  }
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Fused backward-to-activations over 16-row tiles.  Produces the head gradient from a loss functor,
// then delta_l = (delta_{l+1} W_{l+1}) * act'(h_l) down the stack (MFMA), optionally dL/dx columns.
struct BwdTask {
  NetView net;
  const float* hsave[ILSX_MAX_HID];
  float* dsave[ILSX_MAX_HID];   // delta_l [rows][H] for the dW kernel, nullable
  float* dhead;                 // [rows][NO] head gradient, nullable
  int loss;
  int which;                    // LOSS_SAC_ACTORQ: 0 -> this net is Q1, 1 -> Q2
  const float* given;           // LOSS_GIVEN: dL/dout [rows][NO]
  const float *q, *tq1, *tq2, *logp_next, *rew, *done;  // LOSS_SAC_CRITIC
  const float *q1n, *q2n;                                 // LOSS_SAC_ACTORQ
  const float *raw, *eps, *action, *ga1, *ga2;            // LOSS_SAC_POLICY (raw = mu|log_std_raw)
  float* dx;                    // [rows][dx_cols] = dL/dx[:, dx_col0:dx_col0+dx_cols], nullable
  int dx_col0, dx_cols;
};
struct BwdArgs {
  BwdTask t[2];
  int rows, ntasks;
  float inv_B;          // 1/(B*grad_world)
  float gamma, reward_scale, w_mu, w_std;
  const DevScalars* scal;
};

#ifdef ILSX_KERNEL_IMPL
template <int H, int ACT>
__global__ __launch_bounds__(256) void k_mlp_bwd_dx(const BwdArgs A) {
  constexpr int NT = H / 64;
  constexpr int LDH = H + ILSX_LDS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const BwdTask& T = A.t[blockIdx.y];
  const NetView& N = T.net;
  const int NO = N.NO;
  float* bufA = smem;
  float* bufB = bufA + 16 * LDH;
  float* dout = bufB + 16 * LDH;  // [16][ILSX_MAX_NO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, rows = A.rows;

  // ---- head gradient
  for (int e = tid; e < 16 * ILSX_MAX_NO; e += 256) dout[e] = 0.0f;
  __syncthreads();
  if (tid < 16 && r0 + tid < rows) {
    const int gr = r0 + tid;
    float* d = dout + tid * ILSX_MAX_NO;
    if (T.loss == LOSS_GIVEN) {
      for (int j = 0; j < NO; ++j) d[j] = T.given[(size_t)gr * NO + j];
    } else if (T.loss == LOSS_SAC_CRITIC) {
      // sac_alpha.py:110-123: y = r + (1-d)*gamma*(min(TQ1,TQ2) - alpha*logpi'); dL/dq = (q-y)/B
      const float alpha = A.scal->alpha;
      const float r = A.reward_scale * T.rew[gr];
      const float y = r + (1.0f - T.done[gr]) * A.gamma * (fminf(T.tq1[gr], T.tq2[gr]) - alpha * T.logp_next[gr]);
      d[0] = (T.q[gr] - y) * A.inv_B;
    } else if (T.loss == LOSS_SAC_ACTORQ) {
      // sac_alpha.py:144-148: -mean(min(Q1,Q2)); torch.minimum splits ties evenly
      const float a1 = T.q1n[gr], a2 = T.q2n[gr];
      const float w1 = a1 < a2 ? 1.0f : (a1 == a2 ? 0.5f : 0.0f);
      d[0] = -(T.which == 0 ? w1 : 1.0f - w1) * A.inv_B;
    } else {  // LOSS_SAC_POLICY: SURVEY Appendix A.1/A.2
      const int a = NO >> 1;
      const float alpha = A.scal->alpha;
      const float glp = alpha * A.inv_B;
      const float inv_Ba = A.inv_B / (float)a;
      for (int j = 0; j < a; ++j) {
        const float mu = T.raw[(size_t)gr * NO + j], lsr = T.raw[(size_t)gr * NO + a + j];
        const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
        const float sd = expf(ls), e = T.eps[(size_t)gr * a + j], act = T.action[(size_t)gr * a + j];
        const float ga = T.ga1[(size_t)gr * a + j] + T.ga2[(size_t)gr * a + j];
        const float om = 1.0f - act * act;
        const float dz = ga * om + glp * (2.0f * act * om / (om + TANH_EPS));
        const float dmu = dz + 2.0f * A.w_mu * mu * inv_Ba;
        const float dls = dz * sd * e - glp + 2.0f * A.w_std * ls * inv_Ba;
        d[j] = dmu;
        d[a + j] = (lsr >= LOG_SIG_MIN && lsr <= LOG_SIG_MAX) ? dls : 0.0f;
      }
    }
    if (T.dhead) for (int j = 0; j < NO; ++j) T.dhead[(size_t)gr * NO + j] = d[j];
  }
  __syncthreads();

  // ---- delta_L = (dout Wh) * act'(h_L): small contraction (NO <= 64) on the VALU
  const int L = N.nhid;
  {
    const int row = tid >> 4, part = tid & 15, gr = r0 + row;
    const float* Wh = N.base + N.off_Wh;
    const float* hl = T.hsave[L - 1];
    float* ds = T.dsave[L - 1];
    for (int k = part; k < H; k += 16) {
      float s = 0.0f;
      for (int j = 0; j < NO; ++j) s = fmaf(dout[row * ILSX_MAX_NO + j], Wh[(size_t)j * H + k], s);
      float dv = 0.0f;
      if (gr < rows) {
        dv = s * act_grad_from_out<ACT>(hl[(size_t)gr * H + k]);
        if (ds) ds[(size_t)gr * H + k] = dv;
      }
      bufA[row * LDH + k] = dv;
    }
  }
  __syncthreads();

  // ---- delta_{l-1} = (delta_l W_l) * act'(h_{l-1})  for l = L-1 .. 1   (matrix pipe)
  float* cur = bufA;
  for (int l = L - 1; l >= 1; --l) {
    float* nxt = (cur == bufA) ? bufB : bufA;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c_base = wave * (16 * NT);
    gemm_rowtile_kn<NT>(cur, LDH, H, N.base + N.off_W[l], N.ld[l], c_base, li, g, acc);
    const float* hp = T.hsave[l - 1];
    float* ds = T.dsave[l - 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int col = c_base + 16 * t + li;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 4 * g + v, gr = r0 + row;
        float dv = 0.0f;
        if (gr < rows) {
          dv = acc[t][v] * act_grad_from_out<ACT>(hp[(size_t)gr * H + col]);
          if (ds) ds[(size_t)gr * H + col] = dv;
        }
        nxt[row * LDH + col] = dv;
      }
    }
    __syncthreads();
    cur = nxt;
  }

  // ---- dL/dx columns (action columns for the actor; contraction over H on the VALU)
  if (T.dx) {
    const int row = tid >> 4, part = tid & 15, gr = r0 + row;
    const float* W0 = N.base + N.off_W[0];
    const int ld0 = N.ld[0];
    for (int c = 0; c < T.dx_cols; ++c) {
      float s = 0.0f;
      for (int k = part; k < H; k += 16) s = fmaf(cur[row * LDH + k], W0[(size_t)k * ld0 + T.dx_col0 + c], s);
      s = group16_sum(s);
      if (part == 0 && gr < rows) T.dx[(size_t)gr * T.dx_cols + c] = s;
    }
  }
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Weight gradients: dW[n][k] = sum_r A[r][n] * Bm[r][k], db[n] = sum_r A[r][n]  (contraction over
// the batch).  One workgroup per 64(n) x 64(k) output tile; wave w owns n rows [16w,16w+16).
struct DwJob {
  const float* A; const float* Bm; float* dW; float* db;
  int lda, NA, ldb, NB, ldw, n0, k0, pad;
};

#ifdef ILSX_KERNEL_IMPL
__global__ __launch_bounds__(256) void k_mlp_bwd_dw(const DwJob* __restrict__ jobs, int rows) {
  const DwJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int nsub = J.n0 + 16 * wave;
  if (nsub >= J.NA) return;
  const bool n_ok = nsub + li < J.NA;
  int ntk = (J.NB - J.k0 + 15) / 16;
  if (ntk > 4) ntk = 4;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.0f;
  bool k_ok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) k_ok[t] = J.k0 + 16 * t + li < J.NB;
  for (int rc = 0; rc < rows; rc += 16) {
    float a[4], b[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = rc + 4 * g + s;
      const bool r_ok = r < rows;
      a[s] = (r_ok && n_ok) ? J.A[(size_t)r * J.lda + nsub + li] : 0.0f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        b[t][s] = (r_ok && k_ok[t]) ? J.Bm[(size_t)r * J.ldb + J.k0 + 16 * t + li] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < ntk) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
      bsum += a[s];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = J.k0 + 16 * t + li;
    if (t < ntk && k < J.NB) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = nsub + 4 * g + v;
        if (n < J.NA) J.dW[(size_t)n * J.ldw + k] = acc[t][v];
      }
    }
  }
  if (J.db && J.k0 == 0) {
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (g == 0 && n_ok) J.db[nsub + li] = bsum;
  }
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Fused Adam (+ optional Polyak target update) over a flat arena segment.
// torch 1.9 Adam (sac_alpha.py:65-76) + pytorch_util.py:10-12.
struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* tgt;  // tgt nullable
  int n;
  float lr, b1, b2, eps, tau;
  const int* t_ctr;  // completed-step counter (device); this step uses t = *t_ctr + 1
};

#ifdef ILSX_KERNEL_IMPL
__global__ __launch_bounds__(256) void k_adam_polyak(const AdamArgs A) {
  __shared__ float s_step, s_bc2s;
  if (threadIdx.x == 0) {
    const int t = *A.t_ctr + 1;
    const double bc1 = 1.0 - pow((double)A.b1, (double)t);
    const double bc2 = 1.0 - pow((double)A.b2, (double)t);
    s_step = (float)((double)A.lr / bc1);
    s_bc2s = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step = s_step, bc2s = s_bc2s, b1 = A.b1, b2 = A.b2, ob1 = 1.0f - A.b1, ob2 = 1.0f - A.b2;
  const float tau = A.tau, otau = 1.0f - A.tau, eps = A.eps;
  const int n4 = A.n >> 2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    float4 p = reinterpret_cast<float4*>(A.p)[i];
    const float4 gg = reinterpret_cast<const float4*>(A.g)[i];
    float4 m = reinterpret_cast<float4*>(A.m)[i];
    float4 v = reinterpret_cast<float4*>(A.v)[i];
    float* pp = reinterpret_cast<float*>(&p);
    const float* gp = reinterpret_cast<const float*>(&gg);
    float* mp = reinterpret_cast<float*>(&m);
    float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mp[c] = mp[c] * b1 + ob1 * gp[c];
      vp[c] = vp[c] * b2 + ob2 * gp[c] * gp[c];
      const float denom = sqrtf(vp[c]) / bc2s + eps;
      pp[c] = pp[c] - step * (mp[c] / denom);
    }
    reinterpret_cast<float4*>(A.p)[i] = p;
    reinterpret_cast<float4*>(A.m)[i] = m;
    reinterpret_cast<float4*>(A.v)[i] = v;
    if (A.tgt) {
      float4 tg = reinterpret_cast<float4*>(A.tgt)[i];
      tg.x = tg.x * otau + pp[0] * tau; tg.y = tg.y * otau + pp[1] * tau;
      tg.z = tg.z * otau + pp[2] * tau; tg.w = tg.w * otau + pp[3] * tau;
      reinterpret_cast<float4*>(A.tgt)[i] = tg;
    }
  }
}
#endif  // ILSX_KERNEL_IMPL

// block-wide sum of one float per thread (256 threads); result valid in every thread
__device__ __forceinline__ float block256_sum(float v, float* sh) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
