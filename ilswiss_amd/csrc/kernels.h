// kernels.h — gfx950 (CDNA4, wave64) device code for the ILSwiss hot path.
//
// All GEMM-shaped work runs on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32): results are an
// fmaf chain in fp32, so parity with the reference's fp32 PyTorch path holds to summation order.
// Row tiles are 16 batch rows per workgroup (the MFMA M); H/16 waves (16 for H=256) each own 16 output
// columns, so a wave's whole weight slice of a hidden layer (16 x H floats = H/4 VGPRs per lane) is
// fetched with ONE burst of loads issued before the previous layer runs: the k-loop is pure
// ds_read + MFMA with 4 waves per SIMD interleaving.  Hidden activations never leave LDS inside a
// forward / backward-to-input chain.
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
//   W_l   (off_W[l])  : "forward-packed"  16x16 blocks [n/16][k/16][(k%16)/4][n%16][k%4]  (ld[l] = K of layer l)
//   W_l^b (off_Wb[l]) : "backward-packed" 16x16 blocks [k/16][n/16][(n%16)/4][k%16][n%4]  (l >= 1 only)
//   heads (off_Wh)    : natural [NO][H]
// Each 16x16 block is 1 KiB contiguous in exactly the order the 64 lanes of an MFMA B-fragment consume
// it, so a wave fetches a k16 chunk with ONE fully coalesced 16-byte-per-lane load (measured 3.6x faster
// than fetching fragments from a row-major matrix: tools/ubench/wload.hip).
struct NetView {
  float* base;
  int off_W[ILSX_MAX_HID], off_Wb[ILSX_MAX_HID], off_b[ILSX_MAX_HID], ld[ILSX_MAX_HID];
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
  // Adam bias-correction scalars for the NEXT step: lr/(1-b1^t), sqrt(1-b2^t)   (t = t_x + 1)
  float adam_q_step, adam_q_bc2s, adam_pi_step, adam_pi_bc2s;
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


// phase timestamps (shader clock) of workgroup (0,0), wave 0: debugging aid, off unless a buffer is set
#define ILSX_STAMP(dbg, i) do { if ((dbg) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) (dbg)[i] = __builtin_amdgcn_s_memtime(); } while (0)

// element (n,k) of a forward-packed matrix with K columns / of a backward-packed matrix with N rows
__host__ __device__ __forceinline__ int pack_f(int n, int k, int K) {
  return (((n >> 4) * (K >> 4) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3);
}
__host__ __device__ __forceinline__ int pack_b(int n, int k, int N) {
  return (((k >> 4) * (N >> 4) + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (k & 15)) * 4 + (n & 3);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
  return v;
}
template <int N> __device__ __forceinline__ void load_vec(const float* p, float (&v)[N]) {
  if (N == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if (N == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else { v[0] = p[0]; }
}

// ================================================================================================
// Fused MLP forward over 16-row tiles (Mlp.forward, networks.py:85-101; FlattenMlp cat, :108-115;
// tanh-Gaussian head, policies.py:262-307 + distributions.py:23-28,43-50,74-97).
// grid = (ceil(rows/16), ntasks), block = 4*H threads (H/16 waves, wave w owns columns [16w,16w+16)).
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
  unsigned long long* dbg; // nullable phase timestamps
};

#ifdef ILSX_KERNEL_IMPL
template <int H, int ACT>
__global__ __launch_bounds__(4 * H) void k_mlp_fwd(const FwdArgs A) {
  constexpr int NW = H / 16, NTH = 4 * H, NC = H / 16, KPL = H / 64, RPW = 16 / NW;
  constexpr int LDH = H + ILSX_LDS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FwdTask& T = A.t[blockIdx.y];
  const NetView& N = T.net;
  const int KP = N.KP, LDX = KP + ILSX_LDS_PAD, NO = N.NO;
  float* xs = smem;
  float* bufA = xs + 16 * LDX;
  float* bufB = bufA + 16 * LDH;
  float* hout = bufB + 16 * LDH;  // [16][ILSX_MAX_NO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, rows = A.rows, n0 = wave * 16;
  ILSX_STAMP(A.dbg, 0);

  // ---- stage the (concatenated, zero-padded) input tile
  for (int e = tid; e < 16 * KP; e += NTH) {
    const int r = e / KP, k = e - r * KP, gr = r0 + r;
    float v = 0.0f;
    if (gr < rows) {
      if (k < T.d0) v = T.x0[(size_t)gr * T.s0 + k];
      else if (k < T.d0 + T.d1) v = T.x1[(size_t)gr * T.s1 + (k - T.d0)];
      if (T.xsave) T.xsave[(size_t)gr * KP + k] = v;
    }
    xs[r * LDX + k] = v;
  }
  // ---- this wave's whole slice of hidden layer 1 (16 columns x H): one burst, lands while layer 0 runs
  float4 wreg[NC];
  if (N.nhid > 1) {
    const float* wp = N.base + N.off_W[1] + (size_t)wave * NC * 256 + 4 * lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
  }
  __syncthreads();
  ILSX_STAMP(A.dbg, 1);

  // ---- layer 0: K = KP (16 for Hopper, 400 for Humanoid critics), weights streamed with a 1-deep prefetch
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  {
    const float* ap = xs + li * LDX + 4 * g;
    const float* bp = N.base + N.off_W[0] + (size_t)wave * (KP >> 4) * 256 + 4 * lane;
    float4 b = *reinterpret_cast<const float4*>(bp);
    for (int kc = 0; kc < KP; kc += 16) {
      float4 bn = b;
      if (kc + 16 < KP) bn = *reinterpret_cast<const float4*>(bp + (kc + 16) * 16);
      const float4 a = *reinterpret_cast<const float4*>(ap + kc);
      acc0 = MFMA16(a.x, b.x, acc0); acc1 = MFMA16(a.y, b.y, acc1);
      acc0 = MFMA16(a.z, b.z, acc0); acc1 = MFMA16(a.w, b.w, acc1);
      b = bn;
    }
  }
  float* cur = bufA;
  for (int l = 0;; ++l) {
    // epilogue of layer l: bias + activation -> LDS (next layer's A operand) and the saved activations
    {
      const float bv = (N.base + N.off_b[l])[n0 + li];
      float* hs = T.hsave[l];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 4 * g + v;
        const float h = act_fn<ACT>(acc0[v] + acc1[v] + bv);
        cur[row * LDH + n0 + li] = h;
        if (hs && r0 + row < rows) hs[(size_t)(r0 + row) * H + n0 + li] = h;
      }
    }
    __syncthreads();
    ILSX_STAMP(A.dbg, 2 + l);
    if (l + 1 >= N.nhid) break;
    // layer l+1 straight out of registers
    acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1 = acc0;
    const float* ap = cur + li * LDH + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(ap + 16 * c);
      acc0 = MFMA16(a.x, wreg[c].x, acc0); acc1 = MFMA16(a.y, wreg[c].y, acc1);
      acc0 = MFMA16(a.z, wreg[c].z, acc0); acc1 = MFMA16(a.w, wreg[c].w, acc1);
    }
    if (l + 2 < N.nhid) {
      const float* wp = N.base + N.off_W[l + 2] + (size_t)wave * NC * 256 + 4 * lane;
#pragma unroll
      for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
    }
    cur = (cur == bufA) ? bufB : bufA;
  }

  // ---- heads (NO <= 64 outputs): one wave per row, lanes split K, 4 outputs in flight
  const float* Wh = N.base + N.off_Wh;
  const float* bh = N.base + N.off_bh;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr;
    float hv[KPL];
    load_vec<KPL>(cur + row * LDH + KPL * lane, hv);
    for (int j0 = 0; j0 < NO; j0 += 4) {
      float s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + u, NO - 1);
        float wv[KPL];
        load_vec<KPL>(Wh + (size_t)j * H + KPL * lane, wv);
        float t = 0.0f;
#pragma unroll
        for (int c = 0; c < KPL; ++c) t = fmaf(hv[c], wv[c], t);
        s[u] = t;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = wave_sum(s[u]);
      if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j0 + u < NO) hout[row * ILSX_MAX_NO + j0 + u] = s[u] + bh[j0 + u];
      }
    }
  }
  __syncthreads();
  ILSX_STAMP(A.dbg, 6);

  // ---- head epilogue: wave <-> row, lane <-> output / action dim
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr, gr = r0 + row;
    if (gr >= rows) continue;  // wave-uniform
    const float* ho = hout + row * ILSX_MAX_NO;
    if (T.out && lane < NO) T.out[(size_t)gr * NO + lane] = ho[lane];
    if (T.head == HEAD_RAW) continue;
    const int a = NO >> 1, j = lane;
    float lp_quad = 0.f, lp_ls = 0.f, lp_jac = 0.f;
    if (j < a) {
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
        if (T.eps) {
          e = T.eps[(size_t)gr * a + j];
        } else {
          float z4[4];
          philox_normal4(A.seed, A.scal ? A.scal->step : A.step_host, T.rng_stream, gr, j >> 2, z4);
          const int q = j & 3;
          e = q == 0 ? z4[0] : q == 1 ? z4[1] : q == 2 ? z4[2] : z4[3];
        }
        z = e * sd + mu;        // distributions.py:27
        act = tanhf(z);
      }
      const float dm = mu - z;
      lp_quad = dm * dm / expf(2.0f * ls);                 // distributions.py:45-47
      lp_ls = ls;
      lp_jac = logf(1.0f - act * act + TANH_EPS);          // distributions.py:91-93
      if (T.action) T.action[(size_t)gr * a + j] = act;
      if (T.eps_save) T.eps_save[(size_t)gr * a + j] = e;
    }
    if (T.logp) {  // wave-uniform
      lp_quad = wave_sum(lp_quad); lp_ls = wave_sum(lp_ls); lp_jac = wave_sum(lp_jac);
      if (lane == 0) T.logp[gr] = -0.5f * lp_quad - (lp_ls + HALF_LOG_2PI) - lp_jac;
    }
  }
  ILSX_STAMP(A.dbg, 7);
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
  unsigned long long* dbg;
};

#ifdef ILSX_KERNEL_IMPL
template <int H, int ACT>
__global__ __launch_bounds__(4 * H) void k_mlp_bwd_dx(const BwdArgs A) {
  constexpr int NW = H / 16, NTH = 4 * H, NC = H / 16, KPL = H / 64, RPW = 16 / NW;
  constexpr int LDH = H + ILSX_LDS_PAD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const BwdTask& T = A.t[blockIdx.y];
  const NetView& N = T.net;
  const int NO = N.NO, L = N.nhid;
  float* bufA = smem;
  float* bufB = bufA + 16 * LDH;
  float* dout = bufB + 16 * LDH;  // [16][ILSX_MAX_NO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16, rows = A.rows, c0 = wave * 16;

  // ---- this wave's slice of W_{L-1} (all H rows x its 16 columns), in flight during the head phases
  float4 wreg[NC];
  if (L > 1) {
    const float* wp = N.base + N.off_Wb[L - 1] + (size_t)wave * NC * 256 + 4 * lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
  }

  // ---- head gradient: thread <-> (row, output)
  for (int e = tid; e < 16 * ILSX_MAX_NO; e += NTH) {
    const int row = e >> 6, j = e & 63, gr = r0 + row;
    float d = 0.0f;
    if (gr < rows && j < NO) {
      if (T.loss == LOSS_GIVEN) {
        d = T.given[(size_t)gr * NO + j];
      } else if (T.loss == LOSS_SAC_CRITIC) {
        // sac_alpha.py:110-123: y = r + (1-d)*gamma*(min(TQ1,TQ2) - alpha*logpi'); dL/dq = (q-y)/B
        const float alpha = A.scal->alpha;
        const float r = A.reward_scale * T.rew[gr];
        const float y = r + (1.0f - T.done[gr]) * A.gamma * (fminf(T.tq1[gr], T.tq2[gr]) - alpha * T.logp_next[gr]);
        d = (T.q[gr] - y) * A.inv_B;
      } else if (T.loss == LOSS_SAC_ACTORQ) {
        // sac_alpha.py:144-148: -mean(min(Q1,Q2)); torch.minimum splits ties evenly
        const float a1 = T.q1n[gr], a2 = T.q2n[gr];
        const float w1 = a1 < a2 ? 1.0f : (a1 == a2 ? 0.5f : 0.0f);
        d = -(T.which == 0 ? w1 : 1.0f - w1) * A.inv_B;
      } else {  // LOSS_SAC_POLICY: SURVEY Appendix A.1/A.2 ; j < a -> d mu_j, else d log_std_raw_{j-a}
        const int a = NO >> 1, jj = j < a ? j : j - a;
        const float alpha = A.scal->alpha;
        const float glp = alpha * A.inv_B;
        const float inv_Ba = A.inv_B / (float)a;
        const float mu = T.raw[(size_t)gr * NO + jj], lsr = T.raw[(size_t)gr * NO + a + jj];
        const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
        const float sd = expf(ls), ep = T.eps[(size_t)gr * a + jj], act = T.action[(size_t)gr * a + jj];
        const float ga = T.ga1[(size_t)gr * a + jj] + T.ga2[(size_t)gr * a + jj];
        const float om = 1.0f - act * act;
        const float dz = ga * om + glp * (2.0f * act * om / (om + TANH_EPS));
        if (j < a) {
          d = dz + 2.0f * A.w_mu * mu * inv_Ba;
        } else {
          const float dls = dz * sd * ep - glp + 2.0f * A.w_std * ls * inv_Ba;
          d = (lsr >= LOG_SIG_MIN && lsr <= LOG_SIG_MAX) ? dls : 0.0f;
        }
      }
      if (T.dhead) T.dhead[(size_t)gr * NO + j] = d;
    }
    dout[e] = d;
  }
  __syncthreads();

  // ---- delta_{L-1} = (dout Wh) * act'(h_{L-1}): small contraction (NO <= 64) on the VALU
  {
    const float* Wh = N.base + N.off_Wh;
    const float* hl = T.hsave[L - 1];
    float* ds = T.dsave[L - 1];
    for (int e = tid; e < 16 * H; e += NTH) {
      const int row = e / H, k = e - row * H, gr = r0 + row;
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

  // ---- delta_{l-1} = (delta_l W_l) * act'(h_{l-1})  for l = L-1 .. 1   (matrix pipe, W_l from registers)
  float* cur = bufA;
  for (int l = L - 1; l >= 1; --l) {
    float* nxt = (cur == bufA) ? bufB : bufA;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* ap = cur + li * LDH + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(ap + 16 * c);
      acc0 = MFMA16(a.x, wreg[c].x, acc0); acc1 = MFMA16(a.y, wreg[c].y, acc1);
      acc0 = MFMA16(a.z, wreg[c].z, acc0); acc1 = MFMA16(a.w, wreg[c].w, acc1);
    }
    if (l - 1 >= 1) {
      const float* wp = N.base + N.off_Wb[l - 1] + (size_t)wave * NC * 256 + 4 * lane;
#pragma unroll
      for (int c = 0; c < NC; ++c) wreg[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
    }
    const float* hp = T.hsave[l - 1];
    float* ds = T.dsave[l - 1];
    const int col = c0 + li;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * g + v, gr = r0 + row;
      float dv = 0.0f;
      if (gr < rows) {
        dv = (acc0[v] + acc1[v]) * act_grad_from_out<ACT>(hp[(size_t)gr * H + col]);
        if (ds) ds[(size_t)gr * H + col] = dv;
      }
      nxt[row * LDH + col] = dv;
    }
    __syncthreads();
    cur = nxt;
  }

  // ---- dL/dx columns (action columns for the actor): wave <-> row, lanes split the H contraction
  if (T.dx) {
    const float* W0 = N.base + N.off_W[0];
    const int ld0 = N.ld[0];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int row = wave * RPW + rr, gr = r0 + row;
      float dv[KPL];
#pragma unroll
      for (int i = 0; i < KPL; ++i) dv[i] = cur[row * LDH + lane + 64 * i];
      for (int c = 0; c < T.dx_cols; ++c) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < KPL; ++i) s = fmaf(dv[i], W0[pack_f(lane + 64 * i, T.dx_col0 + c, ld0)], s);
        s = wave_sum(s);
        if (lane == 0 && gr < rows) T.dx[(size_t)gr * T.dx_cols + c] = s;
      }
    }
  }
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Weight gradients: dW[n][k] = sum_r A[r][n] * Bm[r][k], db[n] = sum_r A[r][n]  (contraction over
// the batch).  One 1024-thread workgroup per 32(n) x 64(k) output tile: wave = (n-half, row-eighth);
// each wave issues all its operand loads up front, runs its MFMAs, and the 8 row-partials are summed
// through LDS.  dynamic LDS = DW_LDS_BYTES.
enum { DW_OUT_NATURAL = 0, DW_OUT_PACK_F = 1, DW_OUT_PACK_FB = 2 };
struct DwJob {
  const float* A; const float* Bm; float* dW; float* dWb; float* db;
  int lda, NA, ldb, NB, ldw, n0, k0, mode;   // ldw: natural row stride, or K (PACK_F) ; NA rows for PACK_B
};
#define DW_TILE_N 32
#define DW_TILE_K 64
#define DW_LDS_BYTES ((16 * 16 * 64 + 16 * 16) * 4)

#ifdef ILSX_KERNEL_IMPL
__global__ __launch_bounds__(1024) void k_mlp_bwd_dw(const DwJob* __restrict__ jobs, int rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* part = smem;                    // [16 waves][16 acc regs][64 lanes]
  float* bpart = smem + 16 * 16 * 64;    // [16 waves][16]
  const DwJob J = jobs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int wn = wave & 1, wr = wave >> 1;
  const int nsub = J.n0 + 16 * wn;
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
  for (int rc = 16 * wr; rc < rows; rc += 128) {
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
        if (t < ntk) acc[t] = MFMA16(a[s], b[t][s], acc[t]);
      bsum += a[s];
    }
  }
  // partial tiles -> LDS
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) part[(wave * 16 + t * 4 + v) * 64 + lane] = acc[t][v];
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (g == 0) bpart[wave * 16 + li] = bsum;
  __syncthreads();
  // sum the 8 row-partials; thread <-> (n-half, tile, reg, lane)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int e = tid + 1024 * h;
    const int on = e >> 10, rest = e & 1023, t = rest >> 8, v = (rest >> 6) & 3, ol = rest & 63;
    float s = 0.0f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) s += part[((r8 * 2 + on) * 16 + t * 4 + v) * 64 + ol];
    const int n = J.n0 + 16 * on + 4 * (ol >> 4) + v, k = J.k0 + 16 * t + (ol & 15);
    if (t < ntk && n < J.NA && k < J.NB) {
      if (J.mode == DW_OUT_NATURAL) {
        J.dW[(size_t)n * J.ldw + k] = s;
      } else {
        J.dW[pack_f(n, k, J.ldw)] = s;
        if (J.mode == DW_OUT_PACK_FB) J.dWb[pack_b(n, k, J.NA)] = s;
      }
    }
  }
  if (J.db && J.k0 == 0 && tid < 32) {
    const int on = tid >> 4, ol = tid & 15;
    float s = 0.0f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) s += bpart[(r8 * 2 + on) * 16 + ol];
    if (J.n0 + 16 * on + ol < J.NA) J.db[J.n0 + 16 * on + ol] = s;
  }
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Fused Adam (+ optional Polyak target update) over a flat arena segment.
// torch 1.9 Adam (sac_alpha.py:65-76) + pytorch_util.py:10-12.  The bias-correction scalars
// step_size = lr/(1-b1^t) and sqrt(1-b2^t) are kept in device memory (float64 pow done once per step
// by the step's tail kernel) so this kernel is pure streaming.
struct AdamArgs {
  float* p; const float* g; float* m; float* v; float* tgt;  // tgt nullable
  int n;
  float b1, b2, eps, tau;
  const float* step_size;  // device scalars for THIS step
  const float* bc2_sqrt;
};

#ifdef ILSX_KERNEL_IMPL
__global__ __launch_bounds__(256) void k_adam_polyak(const AdamArgs A) {
  const float step = *A.step_size, bc2s = *A.bc2_sqrt, b1 = A.b1, b2 = A.b2, ob1 = 1.0f - A.b1, ob2 = 1.0f - A.b2;
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
