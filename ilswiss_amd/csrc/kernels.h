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
#ifndef ILSX_LATE_WAVES
#define ILSX_LATE_WAVES 4   // waves per SIMD the "late weights" grouped instantiations (GRP == 3) are compiled for
#endif
#ifndef ILSX_MT_WAVES
#define ILSX_MT_WAVES 2   // waves per SIMD the macro-tile instantiations are compiled for (register budget 512 / this)
#endif
#define ILSX_MAX_HID 3
#define ILSX_MAX_NO 64   // max total head outputs (n_heads*out_dim)

enum { HEAD_RAW = 0, HEAD_TANH_SAMPLE = 1, HEAD_TANH_DET = 2, HEAD_TANH_LOGP_OF_ACT = 3,
       HEAD_GAUSS_SAMPLE = 4, HEAD_GAUSS_LOGP_OF_ACT = 5,    // un-squashed Gaussian, state-independent log_std (PPO)
       HEAD_DET_TANH_NOISE = 6,                              // max_act*tanh(out) + clip(noise*eps) (TD3, policies.py:166-188)
       HEAD_DET_LIN_NOISE = 7 };                             // the same with Mlp's default output activation (identity, networks.py:31)
enum { LOSS_GIVEN = 0, LOSS_SAC_CRITIC = 1, LOSS_SAC_ACTORQ = 2, LOSS_SAC_POLICY = 3, LOSS_MSE = 4, LOSS_PPO_POLICY = 5,
       LOSS_TD_CRITIC = 6, LOSS_SACV_VALUE = 7, LOSS_CONST = 8, LOSS_TD3_POLICY = 9, LOSS_BC_MLE = 10, LOSS_BC_MSE = 11 };
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
  // Std / Max / Min of the five batch quantities of create_stats_ordered_dict (sac_alpha.py:202-233): q1, q2, log pi, mu, log std
  float ext_std[5], ext_max[5], ext_min[5];
  int want_stats;   // host sets 1 before the step whose statistics it will read (sac_alpha.py:186: one batch per epoch)
  int pad1;
  // Philox counter of the in-kernel replay draw.  Equal to `step` except inside a deferred-tail window (TailLite), where the
  // second forward launch of a step advances it and the tail (one step late) advances `step`: gather_step == step + 1 means
  // "the step just taken has not had its alpha / counter update yet".
  unsigned long long gather_step;
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
  // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, and v_sin_f32 / v_cos_f32, which take their argument
  // in REVOLUTIONS — sin(2*pi*u) is one instruction on u.  ~1e-6 absolute accuracy: a sample of N(0,1) needs no more, and the
  // libm forms (range reduction for the 2*pi*u arguments) were ~200 of the ~1000 instructions of the policy epilogue, each of
  // which costs 4 cycles on the one wave per SIMD these kernels run (tools/ubench/icache.hip).
  const float u0 = u01_open(c[0]), u2 = u01_open(c[2]);
  const float r0 = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u0));   // sqrt(-2 ln u) = sqrt(-2 ln2 log2 u)
  const float r1 = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u2));
  const float t0 = u01_open(c[1]), t1 = u01_open(c[3]);
  z[0] = r0 * __builtin_amdgcn_cosf(t0); z[1] = r0 * __builtin_amdgcn_sinf(t0);
  z[2] = r1 * __builtin_amdgcn_cosf(t1); z[3] = r1 * __builtin_amdgcn_sinf(t1);
}

// Workgroup barrier for LDS hand-offs: waits for the LDS / scalar queue only.  __syncthreads() also drains the vector-memory
// counter (the acknowledgement of every global STORE the phase before it issued), an ordering nothing in these kernels needs
// (no thread reads another thread's global store within a launch).  Loads whose results are still in flight stay tracked by
// the compiler's own waitcnt insertion.  (Measured neutral on the SAC step; kept because it removes a false dependency.)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// block-wide sum of one float per thread (256 threads); result valid in every thread
__device__ __forceinline__ float block256_sum(float v, float* sh) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  lds_barrier();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  lds_barrier();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- the scalar tail of a SAC step: alpha Adam in float64 (sac_alpha.py:51-53,160-166), step counters, and the Adam
// bias-correction scalars of the NEXT step.
__device__ __host__ inline void adam_scalars(double lr, double b1, double b2, int t, float* step, float* bc2s) {
  *step = (float)(lr / (1.0 - pow(b1, (double)t)));
  *bc2s = (float)sqrt(1.0 - pow(b2, (double)t));
}
__device__ __forceinline__ void sac_finish_dev(DevScalars* sc, const float* alpha_grad_slot, int train_alpha, float lr,
                                               float b1, float b2, float eps, float qf_lr, float policy_lr, int deferred) {
  if (train_alpha) {
    const double g = (double)alpha_grad_slot[0];
    const int t = sc->t_alpha + 1;
    sc->m_alpha = sc->m_alpha * (double)b1 + (1.0 - (double)b1) * g;
    sc->v_alpha = sc->v_alpha * (double)b2 + (1.0 - (double)b2) * g * g;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const double denom = sqrt(sc->v_alpha) / sqrt(bc2) + (double)eps;
    sc->log_alpha -= ((double)lr / bc1) * (sc->m_alpha / denom);
    sc->alpha = (float)exp(sc->log_alpha);
    sc->t_alpha = t;
  }
  sc->t_q += 1;
  sc->t_pi += 1;
  sc->step += 1;
  if (!deferred) sc->gather_step += 1;   // deferred: already advanced by the step's second forward launch
  adam_scalars(qf_lr, b1, b2, sc->t_q + 1, &sc->adam_q_step, &sc->adam_q_bc2s);
  adam_scalars(policy_lr, b1, b2, sc->t_pi + 1, &sc->adam_pi_step, &sc->adam_pi_bc2s);
}
// Deferred tail: inside ilsx_sac_train_from_replay the tail of step k is not a launch of its own.  ONE extra workgroup of the
// first forward launch of step k+1 (FwdArgs::tail, tail_mode 1) does the alpha update while the other workgroups run pi(s'),
// Q(s,a), pi(s) — none of which reads alpha, `step`, the Adam scalars or the t counters (first consumers: the second / third
// launch); the replay draw of that launch is keyed by `gather_step`, which the extra workgroup of the step's SECOND forward
// launch advances (tail_mode 2) — no reader of it is in flight there.  The last step of a call is flushed by the ordinary
// k_sac_tail.  The record lives in device memory, so the kernel arguments grow by one pointer and one int.
struct TailLite {
  const float* logp; int B; float target_entropy, inv_B; float* alpha_grad_slot; DevScalars* scal;
  int train_alpha; float lr, b1, b2, eps, qf_lr, policy_lr;
  int slot_given;   // split run: the alpha-gradient slot was written by the policy phase launch (PhaseCArgs::aslot) and summed over the ranks
                    // by the actor all-reduce — the tail applies it, it does not recompute this rank's partial
};
// d(alpha_loss)/d(log_alpha) of the rows this rank holds, into the gradient arena's alpha slot (one 256-thread workgroup)
__device__ __forceinline__ void alpha_slot_run(const float* logp, int B, float target_entropy, float inv_B, float* slot, float* sh4) {
  float lpe = 0.f;
  for (int r = threadIdx.x; r < B; r += 256) lpe += logp[r] + target_entropy;
  lpe = block256_sum(lpe, sh4);
  if (threadIdx.x == 0) { slot[0] = -lpe * inv_B; slot[1] = 0.f; slot[2] = 0.f; slot[3] = 0.f; }
}
__device__ __forceinline__ void tail_lite_run(const TailLite& T, float* sh4) {   // one 256-thread workgroup
  if (T.scal->gather_step != T.scal->step + 1) return;   // nothing pending (first step of a call); workgroup-uniform
  float lpe = 0.f;
  if (!T.slot_given) {
    for (int r = threadIdx.x; r < T.B; r += 256) lpe += T.logp[r] + T.target_entropy;
    lpe = block256_sum(lpe, sh4);
  }
  if (threadIdx.x == 0) {
    T.scal->alpha_used = T.scal->alpha;
    T.scal->log_alpha_used = T.scal->log_alpha;
    if (!T.slot_given) {
      T.alpha_grad_slot[0] = -lpe * T.inv_B;
      T.alpha_grad_slot[1] = 0.f; T.alpha_grad_slot[2] = 0.f; T.alpha_grad_slot[3] = 0.f;
    }
    sac_finish_dev(T.scal, T.alpha_grad_slot, T.train_alpha, T.lr, T.b1, T.b2, T.eps, T.qf_lr, T.policy_lr, 1);
  }
}

template <int ACT> __device__ __forceinline__ float act_fn(float z) {
  if (ACT == ACT_RELU) return fmaxf(z, 0.0f);
  return tanhf(z);
}
template <int ACT> __device__ __forceinline__ float act_grad_from_out(float h) {
  if (ACT == ACT_RELU) return h > 0.0f ? 1.0f : 0.0f;
  return 1.0f - h * h;
}


// Phase timestamps of EVERY workgroup of a launch (thread 0 of each): debugging aid, off unless a trace buffer is set
// (ilsx_debug_set_stamp_buffer).  Record layout: dbg[wg_linear][ILSX_TRACE_SLOTS] of wall_clock64() ticks (the 100 MHz constant
// clock every XCD shares, so start / end times of different workgroups are comparable); tools/step_gantt.py reads it.  A stamp is
// only a reliable marker right after a barrier (the compiler may sink the store past independent code).
#define ILSX_TRACE_SLOTS 8
#define ILSX_TRACE_MAXWG 2048
#define ILSX_WG_LINEAR (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))
// Compiled in only with -DILSX_STAMPS (make STAMPS=1 -> libilsx_stamps.so, what tools/step_gantt.py loads): each stamp is ~8 issue slots
// per wave plus a parked pointer pair, ~2.5 % of a forward launch at one wave per SIMD — measurement code, kept out of the product build.
// ILSX_STAMPS_FINE (measurement only, changes the timing it measures): a stamp preceded by a full memory wait — "when had everything
// requested so far arrived"
#if defined(ILSX_STAMPS) && defined(ILSX_STAMPS_FINE)
#define ILSX_STAMP_SYNC(dbg, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if ((dbg) && threadIdx.x == 0 && ILSX_WG_LINEAR < ILSX_TRACE_MAXWG) (dbg)[(size_t)ILSX_WG_LINEAR * ILSX_TRACE_SLOTS + (i)] = wall_clock64(); asm volatile("" ::: "memory"); } while (0)
#else
#define ILSX_STAMP_SYNC(dbg, i) ((void)0)
#endif
#ifdef ILSX_STAMPS
#define ILSX_STAMP(dbg, i) do { if ((dbg) && threadIdx.x == 0 && ILSX_WG_LINEAR < ILSX_TRACE_MAXWG) (dbg)[(size_t)ILSX_WG_LINEAR * ILSX_TRACE_SLOTS + (i)] = wall_clock64(); } while (0)
#else
#define ILSX_STAMP(dbg, i) ((void)0)
#endif

// element (n,k) of a forward-packed matrix with K columns / of a backward-packed matrix with N rows
__host__ __device__ __forceinline__ int pack_f(int n, int k, int K) {
  return (((n >> 4) * (K >> 4) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3);
}
__host__ __device__ __forceinline__ int pack_b(int n, int k, int N) {
  return (((k >> 4) * (N >> 4) + (n >> 4)) * 64 + ((n & 15) >> 2) * 16 + (k & 15)) * 4 + (n & 3);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Sum over the 64 lanes, the total in every lane.  Inside a 16-lane row the exchange is DPP (quad permutes, then the half-row and the
// row mirrored onto themselves: VALU modifiers, no LDS round trip), the four row sums are combined through v_readlane.  The butterfly
// of __shfl_xor this replaces is six ds_bpermute round trips per call (~60 cycles each on the one or two waves per SIMD these kernels
// run): the discriminator's dD/dx loop calls it once per input column.  All lanes must be active (every call site is wave-uniform).
template <int CTL> __device__ __forceinline__ float wave_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += wave_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += wave_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += wave_dpp<0x141>(v);   // row_half_mirror
  v += wave_dpp<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}
template <int N> __device__ __forceinline__ void load_vec(const float* p, float (&v)[N]) {
  if constexpr (N == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if constexpr (N == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else { v[0] = p[0]; }
}

// ---- replay ring state + the on-device index draw (random_batch: uniform with replacement over [0,size))
struct DevReplayState {
  long long size, top;
};
__device__ __forceinline__ long long replay_draw(uint64_t seed, uint64_t step, uint32_t stream, uint32_t r, long long size) {
  uint32_t c[4] = {r >> 2, 0x52425546u /* 'RBUF' */, (uint32_t)step, (uint32_t)(step >> 32) ^ (stream * 0x9E3779B9u)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ stream);
  const uint32_t u = c[r & 3];
  return (long long)(((unsigned long long)u * (unsigned long long)size) >> 32);
}
// Fused sample+index: the rows of a forward launch are drawn from the replay ring inside the kernel
// (record = [obs | act | rew | done | next_obs]); designated slices publish the batch keys for later kernels.
struct GatherSpec {
  const float* records; const DevReplayState* st;
  float *s, *a, *r, *d, *s2;   // staging batch written by the publishing slices
  uint64_t seed; uint32_t stream; int rec, o, adim, on;
};

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
  union {   // 16 bytes with two readings — the phase-kernel descriptors (two FwdArgs + a BwdArgs) fill the 4 KB kernel-argument segment to 112 bytes
    struct {
      float* part;                   // column-split kernels: partial head sums [CS][part_stride][NO]
      int g0_off, g1_off;            // GatherSpec (column-split kernels): record offsets of the x0 / x1 segments
    };
    struct {                         // GENERIC kernel (k_mlp_fwd, which reads neither of the above), step_t != 0: this task's noise is keyed by
      uint64_t seed_t, step_t;       //   (seed_t, step_t) instead of the launch's (FwdArgs::seed, step) — several runs' rollout inference in ONE launch,
    };                               //   each run on its own Philox key and call counter (>= 1): ilsx_rollout_steps_lockstep
  };
  int publish;                       // GatherSpec: 1 = publish s,a,r,d ; 2 = s2
  const int* rows_idx;               // nullable: row r of this launch reads source row rows_idx[r] (minibatch gather)
  const float* log_std;              // HEAD_GAUSS_*: state-independent log-std parameter [a]
  float noise, noise_clip, max_act;  // HEAD_DET_TANH_NOISE (noise == 0: deterministic)
  int no_fin;                        // this task's action segment is NOT the launch's finished policy (FwdArgs::fin)
  int agent, first;                  // grouped launches (FwdArgs::tasks): owning agent, 1 = the agent's publishing task
  int out_cols;                      // generic kernel: `out` takes only the first out_cols head outputs, row stride out_cols (0 = all NO)
};
static_assert(offsetof(FwdTask, step_t) == offsetof(FwdTask, g0_off) && sizeof(float*) == 8, "FwdTask: the generic kernel's per-task key overlays part | g0_off | g1_off");
// In-kernel exchange (the merged phase kernels, k_sac_phase_a / _c below): data one workgroup of a launch writes and ANOTHER workgroup
// of the SAME launch reads.  All workgroups that exchange sit on ONE XCD (same row tile), so the data travels through that XCD's L2:
// the producer's ordinary stores are acknowledged by the L2 (the vector L1 is write-through) before it signals; the consumer drops its
// CU's vector L1 once after the wait (buffer_inv sc0, xch_wait) and reads with ordinary loads.  No fence anywhere: a release / acquire
// pair writes back and invalidates the whole L2.  Measured per hand-off (tools/ubench/tilesync.hip): release / acquire atomics 6.4 us,
// __threadfence 9 us, every exchanged word as an agent-scope atomic (sc1: served past the L2) 1.3 us — but each such load is a trip to
// the fabric and the stage bodies ran 3.5 us slower — this way 0.95 us; a kernel boundary is 2.7 us plus a cold prologue.
// ldx / stx mark the exchanged accesses (X = the access happens inside a phase kernel); they compile to the ordinary access.
template <bool X> __device__ __forceinline__ float ldx(const float* p) { return *p; }
template <bool X> __device__ __forceinline__ void stx(float* p, float v) { *p = v; }
// a value another workgroup of this launch wrote to a uniform address: a vector load past every cache (a plain load of a uniform
// address becomes a scalar load, and the scalar cache is not dropped by buffer_inv sc0)
template <bool X> __device__ __forceinline__ float ldx_uniform(const float* p) {
  if constexpr (X) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
// A per-row scalar (Q value) that may still be split into CS column-slice partial sums: summed in a fixed
// order by whoever consumes it (the "combine in the next kernel's prologue" seam of a split-K reduction).
struct PartVal {
  const float* p; int cs; int stride;
  template <bool X = false>
  __device__ __forceinline__ float get(int r) const {  // cs <= 4; the loads are independent and issue together
    const float v0 = ldx<X>(p + r);
    const float v1 = cs > 1 ? ldx<X>(p + (size_t)stride + r) : 0.0f;
    const float v2 = cs > 2 ? ldx<X>(p + (size_t)2 * stride + r) : 0.0f;
    const float v3 = cs > 3 ? ldx<X>(p + (size_t)3 * stride + r) : 0.0f;
    float s = v0;
    if (cs > 1) s += v1;
    if (cs > 2) s += v2;
    if (cs > 3) s += v3;
    return s;
  }
};
// Combine the CS head partials of a tanh-Gaussian policy and run its epilogue (policies.py:262-307):
// one thread per row.  Launched right after k_mlp2_fwd_split for policy tasks.
struct PolicyFinishArgs {
  const float* part; int cs, part_stride, rows, a, head;
  uint32_t rng_stream; uint64_t seed; const DevScalars* scal; uint64_t step_host;
  const float* eps; const float* act_in;
  float *raw, *eps_save, *action, *logp;
  float noise, noise_clip, max_act;   // head == HEAD_DET_TANH_NOISE (TD3): raw = pre-activation [rows][a]
  int use_gather_step;                // phase kernels: the Philox step is scal->gather_step (== the step counter once the pending tail has run)
};
// Grouped launches on a 1-D grid: workgroup -> (macro tile, task, column slice) such that the XCD a workgroup lands on (the dispatcher deals
// workgroups to the 8 XCDs round-robin in linear order) depends on (agent, macro tile) ONLY — the same in every forward / backward launch
// of the lock-step whatever its task count, so the activations of a row range stay in ONE XCD's L2 from the launch that writes them to the
// launches that read them, and every XCD gets the same number of (agent, tile) pairs.  (With a (tiles, tasks, slices) grid padded to 8 in x,
// four macro tiles per agent would put all work on XCDs 0-3.)   tpa = 0: off (3-D grid).
struct GrpSwizzle { int tpa, tiles, agents, np8; };   // tasks per agent in this launch, macro tiles per agent, agents, ceil(agents*tiles/8)
// decode: false = padding workgroup
__device__ __forceinline__ bool grp_swizzle(const GrpSwizzle& S, int ncs, unsigned lin, int* bx, int* task, int* cs) {
  const int xcd = lin & 7, q = lin >> 3;
  const int pidx = q % S.np8, rest = q / S.np8;
  const int p = pidx * 8 + xcd;
  *cs = rest % ncs;
  const int tk = rest / ncs;
  if (p >= S.agents * S.tiles || tk >= S.tpa) return false;
  const int ag = p / S.tiles;
  *bx = p - ag * S.tiles;
  *task = ag * S.tpa + tk;
  return true;
}
struct FwdGroup;
struct FwdArgs {
  FwdTask t[4];
  int rows, ntasks;
  uint64_t seed;
  const DevScalars* scal;  // nullable: step counter for Philox
  uint64_t step_host;      // used when scal == null
  unsigned long long* dbg; // nullable phase timestamps
  int part_stride;         // rows of one partial slab (column-split kernels)
  int fin_on;              // column-split kernels: the x1 segment (actions) of EVERY task is the output of a
  PolicyFinishArgs fin;    //   policy whose head partials are combined + squashed here, in the consumer
  GatherSpec gather;       // rows drawn from the replay ring in-kernel (first launch of a SAC step)
  int xs;                  // XCD confinement: only workgroups with (blockIdx.x & ((1<<xs)-1)) == 0 work (grid.x <<= xs)
  // grouped launch (several agents' tasks in one grid, SURVEY §8e "co-resident seeds as grouped GEMMs"): descriptor
  // tables in device memory, built once per group; blockIdx.y indexes `tasks`
  const struct FwdTaskG* tasks;
  int rt;                  // column-split kernels: consecutive 16-row tiles per workgroup (0 = 1); grid.x = ceil(tiles / rt)
  const struct TailLite* tail;   // column-split kernels, tail_mode != 0: the launch carries one extra y row whose first
  int tail_mode, tail_n;         //   workgroups (one per agent, tail_n of them) run the deferred tail (1) or advance gather_step (2)
  int l0_split;                  // wide inputs: layer 0 in a launch of its own, column-split like layer 1 (every task then has hsave[0])
  int mt;                        // grouped launches: 16-row tiles per workgroup processed together (macro tile; 0 / 1 = one) — host side: picks the instantiation
  int mt_not, mt_a;              //   host side (LDS sizing): the widest head of the launch's tasks in 16-output tiles, the widest finished policy
  GrpSwizzle swz;                // grouped launches with a 1-D grid (see GrpSwizzle)
  int ctab;                      // grouped launches, descriptor records in CONSTANT memory (g_fwd_tab, below): 1 + first slot of this launch's table; 0 = device table `tasks`
  int late;                      // with ctab: the GRP == 3 instantiation (layer-1 weight slice fetched right before its MFMA phase, 4 waves per SIMD)
};
struct FwdGroup { PolicyFinishArgs fin; GatherSpec gather; const DevScalars* scal; int fin_on; };
// one self-contained record per grid row: the task and its agent's per-launch state side by side, so a workgroup reaches
// every pointer it needs with ONE dependent load level (a task -> group -> scalars chain cost ~2 us per level)
struct FwdTaskG { FwdTask t; FwdGroup g; };

#ifdef ILSX_KERNEL_IMPL
// ---- the NO head outputs of one row: one wave, lanes split K (lane owns H/64 consecutive features), 4 outputs in flight
template <int H>
__device__ __forceinline__ void fwd_heads_row(const float* hrow, const float* Wh, const float* bh, int NO, float* hout_row, int lane) {
  constexpr int KPL = H / 64;
  float hv[KPL];
  load_vec<KPL>(hrow + KPL * lane, hv);
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
        if (j0 + u < NO) hout_row[j0 + u] = s[u] + bh[j0 + u];
    }
  }
}
// ---- head epilogue of one row (shared by the generic and the large-batch forward kernels): one wave per row, lane <-> output /
// action dim; `ho` = the row's NO raw head outputs (LDS), gr = its global row
__device__ __forceinline__ void fwd_head_row(const FwdTask& T, const FwdArgs& A, const float* ho, int gr, int lane, int NO) {
  const int oc = T.out_cols > 0 ? T.out_cols : NO;
  if (T.out && lane < oc) T.out[(size_t)gr * oc + lane] = ho[lane];
  if (T.head == HEAD_RAW) return;
  if (T.head == HEAD_DET_TANH_NOISE || T.head == HEAD_DET_LIN_NOISE) {
    // MlpGaussianNoisePolicy.forward (policies.py:166-188): the result is NOT re-clipped to [-max_act, max_act]
    const int j = lane;
    if (j < NO && T.action) {
      float act = T.max_act * (T.head == HEAD_DET_LIN_NOISE ? ho[j] : tanhf(ho[j]));
      if (T.noise != 0.0f) {
        float e;
        if (T.eps) {
          e = T.eps[(size_t)gr * NO + j];
        } else {
          float z4[4];
          philox_normal4(T.step_t ? T.seed_t : A.seed, T.step_t ? T.step_t : (A.scal ? A.scal->step : A.step_host), T.rng_stream, gr, j >> 2, z4);
          const int qd = j & 3;
          e = qd == 0 ? z4[0] : qd == 1 ? z4[1] : qd == 2 ? z4[2] : z4[3];
        }
        act += fminf(fmaxf(T.noise * e, -T.noise_clip), T.noise_clip);
      }
      T.action[(size_t)gr * NO + j] = act;
    }
    return;
  }
  if (T.head >= HEAD_GAUSS_SAMPLE) {
    // ReparamMultivariateGaussianPolicy (policies.py:398-417,462-478 + distributions.py:43-50): log_std is the state-independent
    // parameter (conditioned_std=False, T.log_std) or, with T.log_std null, the net's second head clamped to [LOG_SIG_MIN, LOG_SIG_MAX]
    // (conditioned_std=True, :401-405: NO = 2a outputs, mean | log_std)
    const bool cond = T.log_std == nullptr;
    const int a = cond ? NO >> 1 : NO, j = lane;
    float q = 0.f, l = 0.f;
    if (j < a) {
      const float mu = ho[j], ls = cond ? fminf(fmaxf(ho[a + j], LOG_SIG_MIN), LOG_SIG_MAX) : T.log_std[j];
      float act;
      if (T.head == HEAD_GAUSS_LOGP_OF_ACT) {
        const size_t sr = T.rows_idx ? (size_t)T.rows_idx[gr] : (size_t)gr;
        act = T.act_in[sr * a + j];
      } else {
        float e;
        if (T.eps) {
          e = T.eps[(size_t)gr * a + j];
        } else {
          float z4[4];
          philox_normal4(T.step_t ? T.seed_t : A.seed, T.step_t ? T.step_t : (A.scal ? A.scal->step : A.step_host), T.rng_stream, gr, j >> 2, z4);
          const int qd = j & 3;
          e = qd == 0 ? z4[0] : qd == 1 ? z4[1] : qd == 2 ? z4[2] : z4[3];
        }
        act = e * expf(ls) + mu;
        if (T.action) T.action[(size_t)gr * a + j] = act;
      }
      const float dm = mu - act;
      q = dm * dm / expf(2.0f * ls);
      l = ls;
    }
    if (T.logp) {
      q = wave_sum(q); l = wave_sum(l);
      if (lane == 0) T.logp[gr] = -0.5f * q - (l + HALF_LOG_2PI);
    }
    return;
  }
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
        philox_normal4(T.step_t ? T.seed_t : A.seed, T.step_t ? T.step_t : (A.scal ? A.scal->step : A.step_host), T.rng_stream, gr, j >> 2, z4);
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
  const bool full_tile = r0 + 16 <= rows;   // workgroup-uniform
  ILSX_STAMP(A.dbg, 0);

  // ---- stage the (concatenated, zero-padded) input tile
  for (int e = tid; e < 16 * KP; e += NTH) {
    const int r = e / KP, k = e - r * KP, gr = r0 + r;
    float v = 0.0f;
    if (gr < rows) {
      const size_t sr = T.rows_idx ? (size_t)T.rows_idx[gr] : (size_t)gr;
      if (k < T.d0) v = T.x0[sr * T.s0 + k];
      else if (k < T.d0 + T.d1) v = T.x1[sr * T.s1 + (k - T.d0)];
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
  lds_barrier();
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
      float* const hsp = hs ? hs + (size_t)(r0 + 4 * g) * H + n0 + li : nullptr;   // one address, rows as immediate offsets; no row guards on a whole tile
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 4 * g + v;
        const float h = act_fn<ACT>(acc0[v] + acc1[v] + bv);
        cur[row * LDH + n0 + li] = h;
        if (hs && (full_tile || r0 + row < rows)) hsp[(size_t)v * H] = h;
      }
    }
    lds_barrier();
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
    fwd_heads_row<H>(cur + row * LDH, Wh, bh, NO, hout + row * ILSX_MAX_NO, lane);
  }
  lds_barrier();
  ILSX_STAMP(A.dbg, 6);

  // ---- head epilogue: wave <-> row, lane <-> output / action dim
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = wave * RPW + rr, gr = r0 + row;
    if (gr >= rows) continue;  // wave-uniform
    fwd_head_row(T, A, hout + row * ILSX_MAX_NO, gr, lane, NO);
  }
  ILSX_STAMP(A.dbg, 7);
}
#endif  // ILSX_KERNEL_IMPL

// ---- LARGE batches (PPO's 32768-row minibatches) run on k_mlp_fwd / k_mlp_bwd_dx too: 76 us per 32768-row forward of a plain Q net
// (38 % of the fp32 MFMA peak; 99 us = 29 % with PPO's gather, saved activations and log-prob epilogue).  Three large-batch forward
// kernels were built in round 4, each bit-identical to k_mlp_fwd, each measured at the SAME 72 .. 81 us, and removed again
// (profiles/r04_ppo_fwd.txt has the diagnostic breakdowns): (a) the whole hidden -> hidden matrix resident in the registers of one
// 8-wave workgroup per CU walking 32-row macro tiles (no weight traffic; MFMA phase 35 us, everything else in sequence behind it);
// (b) 4-wave workgroups, three per CU, streaming the weights with two row tiles per fetched fragment (256 MB per launch across the
// L2s; the layer-1 phase alone 51 us); (c) kernel (a) software-pipelined across the two waves of each SIMD — waves 0-3 and 4-7
// alternating, half a period apart, between an MFMA slot (layer 1, 256 MFMAs out of registers) and a VALU slot (activation epilogue,
// layer 0 of a later tile, heads of an earlier one), one barrier per slot: MFMA slots alone 140 us, VALU slots alone 151 us, together
// 267 us at 131072 rows.  The reason is a property of the part (tools/ubench/mfma_valu_overlap.hip, profiles/r04_mfma_overlap.txt):
// a wave issuing back-to-back v_mfma_f32_16x16x4_f32 and a second wave of the same SIMD doing VALU, LDS or even dependent global
// loads take the SUM of their times, not the maximum (94 + 127 -> 214 us; two VALU waves overlap perfectly, 167 + 167 -> 166), and
// within one wave two fmas between consecutive MFMAs cost +34 %, one ds_read_b128 +93 %.  For these fp32 MLPs the attainable ceiling
// is therefore MFMA time PLUS the VALU / LDS issue time of everything around it, and kernel shape does not move it; what moves it is
// instruction count (the large-batch weight-gradient kernel below is nearly pure MFMA + 4-byte LDS reads: 41 %).



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
  PartVal q, tq1, tq2;                                    // LOSS_SAC_CRITIC
  const float *logp_next, *rew, *done;
  PartVal q1n, q2n;                                       // LOSS_SAC_ACTORQ
  const float *raw, *eps, *action, *ga1, *ga2;            // LOSS_SAC_POLICY (raw = mu|log_std_raw)
  float* dx;                    // [rows][dx_cols] = dL/dx[:, dx_col0:dx_col0+dx_cols], nullable
  int dx_col0, dx_cols;
  // LOSS_MSE / LOSS_PPO_POLICY (minibatch rows gathered through rows_idx)
  const int* rows_idx;
  const float *pred, *target;          // LOSS_MSE: v_pred [rows], returns [N]  ;  PPO: logp_cur [rows], advantages [N]
  const float *lp_old, *act_all, *log_std, *mu;   // PPO: fixed log-probs [N], actions [N][a], log_std [a], mean [rows][a]
  float* aux;                          // PPO: per-row d(loss)/d(log_std) contributions [rows][a]
  float clip_eps;
  const DevScalars* scal;              // grouped launches: this task's agent scalars (else BwdArgs::scal)
  float coef;                          // LOSS_TD_CRITIC: 1 (half-MSE) or 2 (MSE); LOSS_SACV_VALUE: alpha; LOSS_CONST: value;
};                                     //   LOSS_TD3_POLICY: max_act
struct BwdArgs {
  BwdTask t[2];
  int rows, ntasks;
  float inv_B;          // 1/(B*grad_world)
  float gamma, reward_scale, w_mu, w_std;
  const DevScalars* scal;
  unsigned long long* dbg;
  int ga_parts, ga_stride;  // LOSS_SAC_POLICY: ga1/ga2 hold ga_parts partial slabs of ga_stride rows each
  int part_stride;          // column-split kernels: rows of one dx partial slab
  int xs;                   // XCD confinement (see FwdArgs)
  const BwdTask* tasks;     // grouped launch: descriptor table in device memory, indexed by blockIdx.y
  int mt;                   // grouped launches: row tiles per workgroup (see FwdArgs::mt)
  GrpSwizzle swz;
  int ctab;                 // grouped launches: 1 + first slot in g_bwd_tab, 0 = device table `tasks` (see FwdArgs::ctab)
  int late;                 // see FwdArgs::late
};

// dL/d(head output j) of row gr for the loss functor of task T (shared by the generic and the column-split
// backward kernels).
template <bool X = false>
__device__ __forceinline__ float bwd_head_grad(const BwdTask& T, const BwdArgs& A, int gr, int j, int NO) {
  float d = 0.0f;
  const DevScalars* scal = T.scal ? T.scal : A.scal;
  if (T.loss == LOSS_GIVEN) {
    d = T.given[(size_t)gr * NO + j];
  } else if (T.loss == LOSS_SAC_CRITIC) {
    // sac_alpha.py:110-123: y = r + (1-d)*gamma*(min(TQ1,TQ2) - alpha*logpi'); dL/dq = (q-y)/B
    const float alpha = ldx_uniform<X>(&scal->alpha);
    const float r = A.reward_scale * ldx<X>(T.rew + gr);
    const float y = r + (1.0f - ldx<X>(T.done + gr)) * A.gamma * (fminf(T.tq1.get<X>(gr), T.tq2.get<X>(gr)) - alpha * ldx<X>(T.logp_next + gr));
    d = (T.q.get<X>(gr) - y) * A.inv_B;
  } else if (T.loss == LOSS_SAC_ACTORQ) {
    // sac_alpha.py:144-148: -mean(min(Q1,Q2)); torch.minimum splits ties evenly
    const float a1 = T.q1n.get<X>(gr), a2 = T.q2n.get<X>(gr);
    const float w1 = a1 < a2 ? 1.0f : (a1 == a2 ? 0.5f : 0.0f);
    d = -(T.which == 0 ? w1 : 1.0f - w1) * A.inv_B;
  } else if (T.loss == LOSS_TD_CRITIC) {
    // td3.py:84-99 (coef 2: plain MSE) / sac.py:93-105 (coef 1: half MSE, tq1 == tq2 == target V): no entropy term
    const float y = A.reward_scale * T.rew[gr] + (1.0f - T.done[gr]) * A.gamma * fminf(T.tq1.get(gr), T.tq2.get(gr));
    d = T.coef * (T.q.get(gr) - y) * A.inv_B;
  } else if (T.loss == LOSS_SACV_VALUE) {
    // sac.py:120-131: v_target = min(Q1,Q2)(s,a~) - alpha*log pi (detached), L = 0.5*mean((v - v_target)^2)
    const float vt = fminf(T.q1n.get(gr), T.q2n.get(gr)) - T.coef * T.logp_next[gr];
    d = (T.q.get(gr) - vt) * A.inv_B;
  } else if (T.loss == LOSS_CONST) {
    d = T.coef * A.inv_B;   // td3.py:113-114: -mean(Q1(s, pi(s)))
  } else if (T.loss == LOSS_TD3_POLICY) {
    // action = max_act*tanh(pre): d pre = dL/da * max_act * (1 - tanh(pre)^2); identity output (T.which = 1): d pre = dL/da * max_act
    const float th = T.which ? 0.0f : tanhf(T.raw[(size_t)gr * NO + j]);
    float ga = 0.0f;   // dQ1/da, possibly in column-slice partial slabs (<= 4: all requested before any is summed; a loop with a
    float gp[4];       //   run-time trip count waits for its load on every trip)
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) { const float v = T.ga1[((size_t)(pc < A.ga_parts ? pc : 0) * A.ga_stride + gr) * NO + j]; gp[pc] = pc < A.ga_parts ? v : 0.0f; }
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) ga += gp[pc];
    d = ga * T.coef * (1.0f - th * th);
  } else if (T.loss == LOSS_BC_MLE || T.loss == LOSS_BC_MSE) {
    // bc.py:88-101.  raw = mean | log_std_raw [rows][2a]; act_all = expert actions [rows][a]
    const int a = NO >> 1, jj = j < a ? j : j - a;
    const float mu = T.raw[(size_t)gr * NO + jj], lsr = T.raw[(size_t)gr * NO + a + jj];
    const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
    const bool gate = lsr >= LOG_SIG_MIN && lsr <= LOG_SIG_MAX;
    const float tgt = T.act_all[(size_t)gr * a + jj];
    if (T.loss == LOSS_BC_MLE) {   // -mean(log_prob(acts)): z = atanh-with-epsilon of the expert action (distributions.py:85-88)
      const float z = 0.5f * (logf(1.0f + tgt + TANH_EPS) - logf(1.0f - tgt + TANH_EPS));
      const float var = expf(2.0f * ls), dm = mu - z;
      d = j < a ? dm / var * A.inv_B : (gate ? -(dm * dm / var - 1.0f) * A.inv_B : 0.0f);
    } else {                        // mean_rows(sum_j (sampled action - acts)^2): action = tanh(mu + sigma*eps)
      const float pred = T.action[(size_t)gr * a + jj];
      const float dz = 2.0f * (pred - tgt) * A.inv_B * (1.0f - pred * pred);
      d = j < a ? dz : (gate ? dz * expf(ls) * T.eps[(size_t)gr * a + jj] : 0.0f);
    }
  } else if (T.loss == LOSS_MSE) {
    // ppo.py:145: mean((v - R)^2) over the minibatch
    const size_t sr = T.rows_idx ? (size_t)T.rows_idx[gr] : (size_t)gr;
    const float v = T.pred[gr], R = T.target[sr];
    d = 2.0f * (v - R) * A.inv_B;
    if (T.lp_old) {
      // use_value_clip (ppo.py:137-143): mean(max((v - R)^2, (v_clip - R)^2)), v_clip = v_old + clamp(v - v_old, +-clip_eps);
      // lp_old carries the fixed values v_old [N].  torch.max splits ties evenly; clamp passes the gradient on [-eps, eps].
      const float vo = T.lp_old[sr], dv = v - vo;
      const float vc = vo + fminf(fmaxf(dv, -T.clip_eps), T.clip_eps);
      const float inside = (dv >= -T.clip_eps && dv <= T.clip_eps) ? 1.0f : 0.0f;
      const float l1 = (v - R) * (v - R), l2 = (vc - R) * (vc - R);
      const float w = l1 > l2 ? 1.0f : (l1 == l2 ? 0.5f : 0.0f);
      d = 2.0f * (w * (v - R) + (1.0f - w) * (vc - R) * inside) * A.inv_B;
    }
  } else if (T.loss == LOSS_PPO_POLICY) {
    // ppo.py:155-164: ratio = exp(logp - logp_old); -mean(min(ratio*A, clamp(ratio, 1+-eps)*A)); j indexes the mean
    const size_t sr = T.rows_idx ? (size_t)T.rows_idx[gr] : (size_t)gr;
    const float adv = T.target[sr];
    const float ratio = expf(T.pred[gr] - T.lp_old[sr]);
    const float lo = 1.0f - T.clip_eps, hi = 1.0f + T.clip_eps;
    const float s1 = ratio * adv, s2 = fminf(fmaxf(ratio, lo), hi) * adv;
    const float inside = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;          // clamp passes gradient on [lo,hi]
    const float w = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f);                 // torch.min tie rule
    const float dlp = -(w * adv + (1.0f - w) * adv * inside) * A.inv_B * ratio;
    if (T.log_std) {   // conditioned_std=False: j indexes the mean; the log-std parameter's gradient is the column sum of aux
      const float ls = T.log_std[j], var = expf(2.0f * ls);
      const float diff = T.act_all[sr * NO + j] - T.mu[(size_t)gr * NO + j];
      d = dlp * diff / var;
      if (T.aux) T.aux[(size_t)gr * NO + j] = dlp * (diff * diff / var - 1.0f);
    } else {           // conditioned_std=True (policies.py:401-405): T.mu holds mean | raw log-std [rows][2a]; j < a -> d mean_j, else d raw log-std
      const int a = NO >> 1, jj = j < a ? j : j - a;
      const float lsr = T.mu[(size_t)gr * NO + a + jj];
      const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX), var = expf(2.0f * ls);
      const float diff = T.act_all[sr * a + jj] - T.mu[(size_t)gr * NO + jj];
      if (j < a) d = dlp * diff / var;
      else d = (lsr >= LOG_SIG_MIN && lsr <= LOG_SIG_MAX) ? dlp * (diff * diff / var - 1.0f) : 0.0f;   // the clamp's gate
    }
  } else {  // LOSS_SAC_POLICY: SURVEY Appendix A.1/A.2 ; j < a -> d mu_j, else d log_std_raw_{j-a}
    const int a = NO >> 1, jj = j < a ? j : j - a;
    const float alpha = scal->alpha;
    const float glp = alpha * A.inv_B;
    const float inv_Ba = A.inv_B / (float)a;
    const float mu = ldx<X>(T.raw + (size_t)gr * NO + jj), lsr = ldx<X>(T.raw + (size_t)gr * NO + a + jj);
    const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
    const float sd = expf(ls), ep = ldx<X>(T.eps + (size_t)gr * a + jj), act = ldx<X>(T.action + (size_t)gr * a + jj);
    float ga = 0.0f;  // d(-min Q)/da~: both critics, each possibly in column-slice partial slabs
    float g1[4], g2[4];   // <= 4 slabs per critic: all requested before any is summed
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const size_t at = ((size_t)(pc < A.ga_parts ? pc : 0) * A.ga_stride + gr) * a + jj;
      const float v1 = ldx<X>(T.ga1 + at), v2 = ldx<X>(T.ga2 + at);
      g1[pc] = pc < A.ga_parts ? v1 : 0.0f; g2[pc] = pc < A.ga_parts ? v2 : 0.0f;
    }
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) ga += g1[pc] + g2[pc];
    const float om = 1.0f - act * act;
    const float dz = ga * om + glp * (2.0f * act * om / (om + TANH_EPS));
    if (j < a) {
      d = dz + 2.0f * A.w_mu * mu * inv_Ba;
    } else {
      const float dls = dz * sd * ep - glp + 2.0f * A.w_std * ls * inv_Ba;
      d = (lsr >= LOG_SIG_MIN && lsr <= LOG_SIG_MAX) ? dls : 0.0f;
    }
  }
  return d;
}

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

  // Whole tile inside the batch (workgroup-uniform): loads and stores of the saved activations / deltas go without row guards (guarded, each
  // was a compare, an EXEC branch and three address instructions).  The saved activations the two delta phases multiply by are requested
  // HERE, before the head gradient, instead of behind the contraction that needs them (a dependent global load per phase, exposed on a
  // workgroup whose 16 waves all wait at the same barrier).
  const bool full_tile = r0 + 16 <= rows;
  constexpr int RG = NTH / H, EPT = 16 / RG;   // delta_{L-1}: thread <-> column kq, rows rq + RG i
  const int kq = tid % H, rq = tid / H;
  float hlv[EPT];
  {
    const float* const hlp = T.hsave[L - 1] + (size_t)(r0 + rq) * H + kq;
#pragma unroll
    for (int i = 0; i < EPT; ++i) hlv[i] = (full_tile || r0 + rq + RG * i < rows) ? hlp[(size_t)(RG * i) * H] : 0.0f;
  }

  // ---- head gradient: thread <-> (row, output)
  for (int e = tid; e < 16 * NO; e += NTH) {
    const int row = e / NO, j = e - row * NO, gr = r0 + row;
    float d = 0.0f;
    if (gr < rows) {
      d = bwd_head_grad(T, A, gr, j, NO);
      if (T.dhead) T.dhead[(size_t)gr * NO + j] = d;
    }
    dout[row * ILSX_MAX_NO + j] = d;
  }
  lds_barrier();

  // ---- delta_{L-1} = (dout Wh) * act'(h_{L-1}): small contraction (NO <= 64) on the VALU.  A thread's EPT rows share its column: one
  //      head-weight load per output feeds all of them (the same fma chain over j per element as one element at a time)
  {
    const float* Wh = N.base + N.off_Wh;
    float* ds = T.dsave[L - 1];
    float sv[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) sv[i] = 0.0f;
    for (int j = 0; j < NO; ++j) {
      const float w = Wh[(size_t)j * H + kq];
#pragma unroll
      for (int i = 0; i < EPT; ++i) sv[i] = fmaf(dout[(rq + RG * i) * ILSX_MAX_NO + j], w, sv[i]);
    }
    float* const dsp = ds ? ds + (size_t)(r0 + rq) * H + kq : nullptr;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int row = rq + RG * i;
      const bool ok = full_tile || r0 + row < rows;
      const float dv = ok ? sv[i] * act_grad_from_out<ACT>(hlv[i]) : 0.0f;
      if (ds && ok) dsp[(size_t)(RG * i) * H] = dv;
      bufA[row * LDH + kq] = dv;
    }
  }
  lds_barrier();

  // ---- delta_{l-1} = (delta_l W_l) * act'(h_{l-1})  for l = L-1 .. 1   (matrix pipe, W_l from registers)
  float* cur = bufA;
  for (int l = L - 1; l >= 1; --l) {
    float* nxt = (cur == bufA) ? bufB : bufA;
    const int col = c0 + li;
    // the activations this layer's delta is multiplied by, requested before the MFMAs that produce it
    float hpv[4];
    {
      const float* const hpp = T.hsave[l - 1] + (size_t)(r0 + 4 * g) * H + col;
#pragma unroll
      for (int v = 0; v < 4; ++v) hpv[v] = (full_tile || r0 + 4 * g + v < rows) ? hpp[(size_t)v * H] : 0.0f;
    }
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
    float* ds = T.dsave[l - 1];
    float* const dsp = ds ? ds + (size_t)(r0 + 4 * g) * H + col : nullptr;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int row = 4 * g + v;
      const bool ok = full_tile || r0 + row < rows;
      const float dv = ok ? (acc0[v] + acc1[v]) * act_grad_from_out<ACT>(hpv[v]) : 0.0f;
      if (ds && ok) dsp[(size_t)v * H] = dv;
      nxt[row * LDH + col] = dv;
    }
    lds_barrier();
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
// Column-split kernels for 2-hidden-layer nets (every BASELINE config): the last hidden layer of a 16-row
// tile is split over CS workgroups (blockIdx.z), each owning H/CS columns, so the per-workgroup weight
// burst (H*H/CS floats) and MFMA time (both ~1/CS of the unsplit kernel) stop bounding the launch and
// 16*CS*ntasks workgroups spread over the chip.  Layer 0 (K = KP, tiny) is recomputed by every slice.
// Head outputs leave as CS partial sums per row (FwdTask::part) and are combined in the consumer's
// prologue (PartVal) or by k_policy_finish; backward input-gradients leave as CS partial slabs likewise.
// block = 4*H/CS threads; wave w owns column tile(s) [w*CS, (w+1)*CS) of layer 0 and tile cs*NWV+w of layer 1.
// Descriptor records of grouped launches in CONSTANT address space.  Copied by value at kernel entry out of a device-memory table (GRP == 1)
// a record is ~130 / ~100 scalar registers live at once — more than a wave has — and the compiler parks the overflow in VGPR lanes: ~370 of
// the ~2.6k instructions a grouped forward workgroup executes were v_writelane / v_readlane.  A `__constant__` table is read like the
// kernel-argument segment of the single-run launches: a field is a scalar load where it is used (the address space tells the compiler
// that no store of the kernel can alias it), loaded again instead of parked when registers run short.  GRP == 2 selects it; slots are
// handed out by the host (grp_const_alloc, ilsx_core.hip), a group that finds none falls back to GRP == 1.
// GRP == 3 is GRP == 2 with the wave's 64-register slice of the hidden->hidden matrix fetched right before the MFMA phase that consumes
// it instead of at entry: the prologue (staging, layer 0 / head gradient, delta_1) then runs in <= 128 registers and the kernel is
// compiled for FOUR waves per SIMD instead of two.  Each workgroup waits out one exposed L2 burst (slower alone), twice as many are
// resident: it pays when a launch has more workgroups than 2 per CU (K = 8 seeds: 1024), and loses when it has not (K = 4: 512) — the
// host picks per launch (FwdArgs::late).  Same MFMA chains in the same order: bit-identical.
#define GRP_CONST_SLOTS 768
#ifdef ILSX_KERNEL_IMPL
__constant__ FwdTaskG g_fwd_tab[GRP_CONST_SLOTS];
__constant__ BwdTask g_bwd_tab[GRP_CONST_SLOTS];
#endif
#ifdef ILSX_KERNEL_IMPL
// PH: 0 = the whole forward in one launch, every slice recomputing layer 0 (K = KP is tiny for the planar tasks); for wide inputs
// (Humanoid: KP = 396) that recomputation is 6x the slice's own layer-1 work, so the forward runs as two launches of the same grid:
// PH = 1 stages x (gather / policy epilogue as in PH 0) and computes THIS slice's 64 columns of layer 0 into hsave[0];
// PH = 2 starts from hsave[0] (16 KB per tile, L2-resident) and does layer 1 + heads.
template <int H, int ACT, int CS, int GRP, int PH = 0, int MT = 1>   // GRP: 0 = tasks in the kernel arguments, 1 = device table, 2 = constant table
__global__ __launch_bounds__(4 * H / CS, (MT > 1 ? ILSX_MT_WAVES : GRP == 3 ? ILSX_LATE_WAVES : 1)) void k_mlp2_fwd_split(const FwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(MT == 1 || GRP, "macro tiles are a grouped-launch shape");
  int sw_bx = 0, sw_task = 0, sw_cs = 0;
  if constexpr (GRP && MT > 1) {   // 1-D grid (GrpSwizzle); the deferred-tail workgroups follow the working ones
    const unsigned nwork = 8u * A.swz.np8 * A.swz.tpa * CS;
    if (blockIdx.x >= nwork) {
      const int ag = blockIdx.x - nwork;
      if (A.tail_mode && ag < A.tail_n) {
        const TailLite& TL = A.tail[ag];
        if (A.tail_mode == 1) tail_lite_run(TL, smem);
        else if (threadIdx.x == 0) TL.scal->gather_step += 1;
      }
      return;
    }
    if (!grp_swizzle(A.swz, CS, blockIdx.x, &sw_bx, &sw_task, &sw_cs)) return;
  } else {
  if (A.tail_mode && (int)blockIdx.y == A.ntasks) {   // the extra y row of a deferred-tail launch (an extra x column would shift
    const int ag = blockIdx.x + gridDim.x * blockIdx.z;  // the tile -> XCD mapping of every other workgroup, see launch_fwd)
    if (ag < A.tail_n) {
      const TailLite& TL = A.tail[ag];             // one record per agent of a grouped launch (tail_n = 1 otherwise)
      if (A.tail_mode == 1) tail_lite_run(TL, smem);
      else if (threadIdx.x == 0) TL.scal->gather_step += 1;
    }
    return;
  }
  }
  // GRP: descriptor records in device memory, copied by value at entry (before any store) so that their fields are
  // loaded once with scalar loads; a reference would be re-read with vector loads after every store
  // !GRP: the record stays where it is, in the kernel-argument segment (constant address space): a field is a scalar load at its
  // use and, when registers run short, is loaded again instead of being parked in a VGPR lane.  Copying the record by value here
  // made ~100 scalars live at once: 196 v_writelane + 369 v_readlane in this kernel (a quarter of a wave's issue slots in the prologue).
  FwdTaskG Rg;
  const int ty = (GRP && MT > 1) ? sw_task : (int)blockIdx.y;
  if (GRP == 1) Rg = A.tasks[ty];
  const FwdTaskG& Rc = g_fwd_tab[GRP >= 2 ? A.ctab - 1 + ty : 0];
  const FwdTask& T = GRP >= 2 ? Rc.t : GRP == 1 ? Rg.t : A.t[ty];
  const FwdGroup* GP = GRP >= 2 ? &Rc.g : GRP == 1 ? &Rg.g : nullptr;
  // the body is shared as TEXT with the merged phase kernels (see fwd_split_tile.inc for why it is not a device function)
  constexpr bool XCH = false;
  const int bx = (GRP && MT > 1) ? sw_bx : (int)blockIdx.x, cs = (GRP && MT > 1) ? sw_cs : (int)blockIdx.z;
  const bool first_task = ty == 0;
#define XCH_HOOK_STAGE
#define XCH_HOOK_FIN
#include "fwd_split_tile.inc"
#undef XCH_HOOK_STAGE
#undef XCH_HOOK_FIN
}
#endif  // ILSX_KERNEL_IMPL
#ifdef ILSX_KERNEL_IMPL
__global__ __launch_bounds__(64) void k_policy_finish(const PolicyFinishArgs P) {
  const int gr = blockIdx.x * 64 + threadIdx.x;
  if (gr >= P.rows) return;
  const int a = P.a, NO = 2 * a;
  float lp_quad = 0.f, lp_ls = 0.f, lp_jac = 0.f;
  const uint64_t step = P.scal ? P.scal->step : P.step_host;
  for (int j0 = 0; j0 < a; j0 += 4) {
    float z4[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.head == HEAD_TANH_SAMPLE && !P.eps) philox_normal4(P.seed, step, P.rng_stream, gr, j0 >> 2, z4);
    for (int jj = 0; jj < 4 && j0 + jj < a; ++jj) {
      const int j = j0 + jj;
      float mu = 0.f, lsr = 0.f;
      for (int c = 0; c < P.cs; ++c) {
        mu += P.part[((size_t)c * P.part_stride + gr) * NO + j];
        lsr += P.part[((size_t)c * P.part_stride + gr) * NO + a + j];
      }
      if (P.raw) { P.raw[(size_t)gr * NO + j] = mu; P.raw[(size_t)gr * NO + a + j] = lsr; }
      const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
      const float sd = expf(ls);
      float e = 0.f, z, act;
      if (P.head == HEAD_TANH_DET) {
        z = mu; act = tanhf(mu);
      } else if (P.head == HEAD_TANH_LOGP_OF_ACT) {
        act = P.act_in[(size_t)gr * a + j];
        z = 0.5f * (logf(1.0f + act + TANH_EPS) - logf(1.0f - act + TANH_EPS));
      } else {
        e = P.eps ? P.eps[(size_t)gr * a + j] : z4[jj];
        z = e * sd + mu;
        act = tanhf(z);
      }
      const float dm = mu - z;
      lp_quad += dm * dm / expf(2.0f * ls);
      lp_ls += ls;
      lp_jac += logf(1.0f - act * act + TANH_EPS);
      if (P.action) P.action[(size_t)gr * a + j] = act;
      if (P.eps_save) P.eps_save[(size_t)gr * a + j] = e;
    }
  }
  if (P.logp) P.logp[gr] = -0.5f * lp_quad - (lp_ls + HALF_LOG_2PI) - lp_jac;
}

template <int H, int ACT, int CS, int GRP, int MT = 1>
__global__ __launch_bounds__(4 * H / CS, (MT > 1 ? ILSX_MT_WAVES : GRP == 3 ? ILSX_LATE_WAVES : 1)) void k_mlp2_bwd_split(const BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(MT == 1 || GRP, "macro tiles are a grouped-launch shape");
  int sw_bx = 0, sw_task = 0, sw_cs = 0;
  if constexpr (GRP && MT > 1) {
    if (!grp_swizzle(A.swz, CS, blockIdx.x, &sw_bx, &sw_task, &sw_cs)) return;
  }
  const int ty = (GRP && MT > 1) ? sw_task : (int)blockIdx.y;
  BwdTask Tg;
  if (GRP == 1) Tg = A.tasks[ty];   // by value at entry (see k_mlp2_fwd_split)
  const BwdTask& T = GRP >= 2 ? g_bwd_tab[A.ctab - 1 + ty] : GRP == 1 ? Tg : A.t[ty];   // GRP 0 / 2: read in place (kernel arguments / constant table)
  constexpr bool XCH = false;
  const int bx = (GRP && MT > 1) ? sw_bx : (int)blockIdx.x, cs = (GRP && MT > 1) ? sw_cs : (int)blockIdx.z;
#define XCH_HOOK_ACT
#define XCH_HOOK_HEAD
#include "bwd_split_tile.inc"
#undef XCH_HOOK_ACT
#undef XCH_HOOK_HEAD
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Merged phase kernels of the SAC step (single run, narrow inputs, every working workgroup co-resident: <= 1 per CU).
// The step's forward / backward-to-activations launches are ROW-LOCAL: the workgroups that serve one 16-row tile (4 tasks x CS column
// slices) only exchange data with each other, and they sit on one XCD (tile = blockIdx.x, x fastest in the linear workgroup order the
// dispatcher deals round-robin to the 8 XCDs).  So F1 -> F2 -> B1 (phase A) and F3 -> B2 -> B3 (phase C) each run as ONE launch whose
// workgroups walk the stages and hand data over through per-tile arrival counters instead of kernel boundaries (ldx / stx above).
// The step becomes A, dW{Q}+Adam+Polyak, C, dW{pi}+Adam: 4 launches instead of 8 — the two dW launches stay, they contract over ALL
// rows.  Arithmetic and summation order are those of the separate launches (the stage bodies are the same text: *_split_tile.inc),
// so results are bit-identical to the 8-launch path.  A stage's batch-independent operands (its weights) are requested BEFORE the
// wait for the previous stage (the XCH_HOOK_* points of the bodies), so their latency passes while waiting.
//   flags: one 128-byte line per counter; counter 3t / 3t+1 = stage 1 / stage 2 arrivals of tile t, 3t+2 = stage 1 arrivals of the
//   policy-on-next_obs slices alone (phase A: all the target critics wait for), counter PHASE_TAIL_FLAG = "the
//   deferred tail of the previous step has run" (alpha is valid).  The preceding dW launch zeroes them (DwArgs::zero_flags).  After the
//   counters: one word per tile in which every workgroup ORs the XCD it runs on (checked by the host: one bit per tile or the call fails).
//   A workgroup that waits longer than ~50 ms sets *err and goes on; the host then ROLLS THE WINDOW BACK (parameters, optimiser state and
//   counters are checkpointed at the start of every window: one device-to-device copy) and re-runs it on one launch per stage, where it
//   stays (ilsx_sac.hip sac_phase_check).  The co-residency the protocol needs holds on a GPU this process has to itself; another
//   process's kernels on the same GPU (the reference's launcher runs all workers of a sweep on one GPU, run_experiment.py:57-78) can
//   break it, and then costs one bounded wait, not the run.
#define PHASE_MAX_TILES 64
#define PHASE_TAIL_FLAG (3 * PHASE_MAX_TILES)
#define PHASE_NFLAGS (3 * PHASE_MAX_TILES + 1)
#define PHASE_MASK_WORD(t) ((PHASE_NFLAGS + (t)) * 32)   // per-tile XCD masks live after the counters (never zeroed by the dW launches)
#define PHASE_FLAG_WORDS ((PHASE_NFLAGS + PHASE_MAX_TILES) * 32)
struct PhaseAArgs {
  FwdArgs f1, f2; BwdArgs b1; unsigned* flags; int* err; unsigned long long* dbg;
  PolicyFinishArgs fin_pi; int fin_pi_on, pad;   // finish pi(s) (task 3 of stage 1) here, on the lead slice of the first critic's backward row (policy_fin_tile)
};
struct PhaseCArgs {
  FwdArgs f3; BwdArgs b2, b3; unsigned* flags; int* err; unsigned long long* dbg;
  // Polyak update of the target critics (pytorch_util.py:10-12), run by the otherwise idle bookkeeping row while the policy phase
  // computes: polyak_n > 0 = the critics' dW launch of this step left it out (AdamFuse::T null); same expression, same operands
  float* polyak_T; const float* polyak_P; int polyak_n; float polyak_tau;
  // split run: the same row writes this rank's alpha-gradient partial into the gradient arena's slot (log pi of this step is final since
  // phase A), so that the actor all-reduce that follows carries it and the deferred tail finds the sum (TailLite::slot_given)
  float* aslot; const float* aslot_logp; int aslot_B; float aslot_te, aslot_invB;
};

static_assert(sizeof(PhaseAArgs) <= 4096 && sizeof(PhaseCArgs) <= 4096, "phase-kernel descriptors travel in the kernel-argument segment (4 KB)");
#define PHASE_CONST_SLOTS 64
#ifdef ILSX_KERNEL_IMPL
__constant__ PhaseAArgs g_phase_a_tab[PHASE_CONST_SLOTS];
__constant__ PhaseCArgs g_phase_c_tab[PHASE_CONST_SLOTS];
#endif

#ifdef ILSX_KERNEL_IMPL
__device__ __forceinline__ void xch_arrive(unsigned* flag) {   // every thread of the workgroup calls it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this thread's exchange stores are acknowledged by the L2
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Once a wait has given up (*err set), the host will roll the window back and re-run it on one launch per stage (ilsx_sac.hip
// sac_phase_check); so that the REST of the window does not sit out the bound again in every wait, the weight-gradient launch that re-arms
// the counters for the next phase launch arms them SATISFIED (PHASE_FLAG_DEAD, k_mlp_bwd_dw) when it finds *err set: later phase
// launches of the window run straight through.  (Reading *err inside the phase kernels instead — at entry, scalar or vector load — cost
// 0.8 us per phase launch: tools/ab_r03.sh.)
#define PHASE_FLAG_DEAD 0x40000000u
__device__ __forceinline__ void xch_wait(const unsigned* flag, unsigned target, int* err) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 16)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  asm volatile("buffer_inv sc0" ::: "memory");   // drop this CU's vector L1: the producers' lines are read from the XCD's L2
}
// which XCD this workgroup runs on (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
// every working workgroup of a phase launch ORs its XCD into its tile's mask (fire and forget): more than one bit in a mask = the
// placement the exchange relies on did not hold, the host fails the call (ilsx_sac_train_from_replay) and falls back
__device__ __forceinline__ void xch_mark_xcd(unsigned* mask) {
  if (threadIdx.x == 0) __hip_atomic_fetch_or(mask, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The tanh-Gaussian epilogue of a column-split policy forward for one 16-row tile (policies.py:262-283, distributions.py:23-28,43-50,
// 74-97), as a job of its own: the CS head partials are combined, the action is sampled and everything later launches need (action,
// noise, raw head, log pi) is published.  Same expressions, in the same order, as the epilogue a consuming forward runs in its prologue
// (fwd_split_tile.inc, `fin`): phase A runs it for pi(s) on a workgroup that would otherwise idle until the TD target is ready, so the
// policy phase launch starts from finished actions.  lp3 = [16][32][3] floats of LDS.
__device__ __forceinline__ void policy_fin_tile(const PolicyFinishArgs& P, int r0, int rows, float* lp3, int nth) {
  const int a = P.a, NOp = 2 * a, tid = threadIdx.x;
  const unsigned long long fstep = P.scal ? (P.use_gather_step ? P.scal->gather_step : P.scal->step) : P.step_host;
  for (int e = tid; e < 16 * a; e += nth) {
    const int row = e / a, j = e - row * a, gr = r0 + row;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (gr < rows) {
      float pm[4], pl[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t at = ((size_t)(c < P.cs ? c : 0) * P.part_stride + gr) * NOp + j;
        pm[c] = P.part[at];
        pl[c] = P.part[at + a];
      }
      float mu = 0.f, lsr = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { mu += c < P.cs ? pm[c] : 0.0f; lsr += c < P.cs ? pl[c] : 0.0f; }   // slab order
      float ep;
      if (P.eps) {
        ep = P.eps[(size_t)gr * a + j];
      } else {
        float z4[4];
        philox_normal4(P.seed, fstep, P.rng_stream, gr, j >> 2, z4);
        const int q = j & 3;
        ep = q == 0 ? z4[0] : q == 1 ? z4[1] : q == 2 ? z4[2] : z4[3];
      }
      const float ls = fminf(fmaxf(lsr, LOG_SIG_MIN), LOG_SIG_MAX);
      const float sd = expf(ls);
      const float z = ep * sd + mu;
      const float act = tanhf(z);
      const float dm = mu - z;
      c0 = dm * dm / expf(2.0f * ls); c1 = ls; c2 = logf(1.0f - act * act + TANH_EPS);
      if (P.raw) { P.raw[(size_t)gr * NOp + j] = mu; P.raw[(size_t)gr * NOp + a + j] = lsr; }
      if (P.action) P.action[(size_t)gr * a + j] = act;
      if (P.eps_save) P.eps_save[(size_t)gr * a + j] = ep;
    }
    lp3[(row * 32 + j) * 3 + 0] = c0; lp3[(row * 32 + j) * 3 + 1] = c1; lp3[(row * 32 + j) * 3 + 2] = c2;
  }
  lds_barrier();
  if (P.logp && tid < 16 && r0 + tid < rows) {
    float q = 0.f, l = 0.f, jc = 0.f;
    for (int j = 0; j < a; ++j) { q += lp3[(tid * 32 + j) * 3]; l += lp3[(tid * 32 + j) * 3 + 1]; jc += lp3[(tid * 32 + j) * 3 + 2]; }
    P.logp[r0 + tid] = -0.5f * q - (l + HALF_LOG_2PI) - jc;
  }
  lds_barrier();
}

// Phase A: stage 1 = fwd{pi(s') | Q1(s,a) | Q2(s,a) | pi(s)} (tasks y = 0..3, rows drawn from the replay ring);
//          stage 2 = fwd{TQ1, TQ2 (s', a')} on the workgroups of y = 0 / 3 (prologue: finish pi(s'); next_obs as the policy task published it);
//          stage 3 = bwd{Q1, Q2 <- TD target} on the workgroups of y = 1 / 2;  y = 4: the deferred tail of the previous step.
// CT: the descriptor block is read from a `__constant__` copy (g_phase_a_tab[slot], uploaded once when the step is built) instead of the
// kernel-argument segment: see g_fwd_tab above — fields come as scalar loads where they are used instead of sitting in (and spilling out
// of) scalar registers from the kernel's entry on.  Pk is then unused.
template <int H, int ACT, int CS, bool CT = false>
__global__ __launch_bounds__(4 * H / CS) void k_sac_phase_a(const PhaseAArgs Pk, int slot) {
  const PhaseAArgs& P = CT ? g_phase_a_tab[slot] : Pk;
  constexpr int GRP = 0; constexpr bool XCH = true;
  constexpr int PH = 0, MT = 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FwdGroup* GP = nullptr;
  const int bx = blockIdx.x, cs = blockIdx.z, y = blockIdx.y;
  unsigned* tflag = P.flags + PHASE_TAIL_FLAG * 32;
  if (y == 4) {
    if (bx == 0 && cs == 0) {
      const TailLite& TL = P.f1.tail[0];
      tail_lite_run(TL, smem);
      if (threadIdx.x == 0) {
        __hip_atomic_store(&TL.scal->alpha, TL.scal->alpha, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // what the critic target of THIS step reads (stage 3), past the caches
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(tflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  if (bx * 16 >= P.f1.rows) return;   // padding tiles (grid.x is a multiple of 8): no member of such a tile takes part
  unsigned* f1 = P.flags + (3 * bx) * 32;
  unsigned* f2 = P.flags + (3 * bx + 1) * 32;
  ILSX_STAMP(P.dbg, 0);
#if defined(ILSX_STAMPS) && defined(ILSX_STAMPS_FINE)
  if (P.dbg && threadIdx.x == 0 && (y == 1 || y == 2)) P.dbg[(size_t)ILSX_WG_LINEAR * ILSX_TRACE_SLOTS + 5] = clock64();   // shader clock (s_memtime) beside the 100 MHz stamps
#endif
  xch_mark_xcd(P.flags + PHASE_MASK_WORD(bx));
  {
    const FwdArgs& A = P.f1;
    const FwdTask& T = A.t[y];
    const bool first_task = y == 0;
#define XCH_HOOK_STAGE
#define XCH_HOOK_FIN
#include "fwd_split_tile.inc"
#undef XCH_HOOK_STAGE
#undef XCH_HOOK_FIN
  }
  ILSX_STAMP(P.dbg, 1);
  unsigned* f0 = P.flags + (3 * bx + 2) * 32;
  if (y == 0) {   // the policy-on-next_obs slices: what the target critics wait for (they do not need the critics' own stage 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(f0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(f1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    xch_arrive(f1);
  }
  ILSX_STAMP(P.dbg, 2);
  if (y == 0 || y == 3) {
    {   // the target critic's weights are on their way while the tile's policy slices finish; next_obs = the rows stage 1 published
      const FwdArgs& A = P.f2;
      const FwdTask& T = A.t[y == 0 ? 0 : 1];
      const bool first_task = y == 0;
#define XCH_HOOK_STAGE ILSX_STAMP_SYNC(P.dbg, 7); xch_wait(f0, CS, P.err); ILSX_STAMP(P.dbg, 3);
#define XCH_HOOK_FIN
#define XCH_FINE_DBG P.dbg
#include "fwd_split_tile.inc"
#undef XCH_FINE_DBG
#undef XCH_HOOK_STAGE
#undef XCH_HOOK_FIN
    }
    ILSX_STAMP(P.dbg, 4);
    xch_arrive(f2);
    ILSX_STAMP(P.dbg, 5);
  } else {
    xch_wait(f1, 4 * CS, P.err);   // the other slices' layer-1 activations, the lead slice's layer 0
    if (P.fin_pi_on && y == 1 && cs == 0) {   // workgroup-uniform
      xch_wait(tflag, 1u, P.err);   // the pending tail of the previous step reads that step's log pi: it has to be through before this one's lands
      policy_fin_tile(P.fin_pi, bx * 16, P.f1.rows, smem, 4 * H / CS);
    }
    {
      const BwdArgs& A = P.b1;
      const BwdTask& T = A.t[y - 1];
#define XCH_HOOK_ACT
#define XCH_HOOK_HEAD xch_wait(f2, 2 * CS, P.err); xch_wait(tflag, 1u, P.err); ILSX_STAMP(P.dbg, 3);
#include "bwd_split_tile.inc"
#undef XCH_HOOK_ACT
#undef XCH_HOOK_HEAD
    }
    ILSX_STAMP(P.dbg, 4);
#if defined(ILSX_STAMPS) && defined(ILSX_STAMPS_FINE)
    if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)ILSX_WG_LINEAR * ILSX_TRACE_SLOTS + 6] = clock64();
#endif
  }
}

// Phase C: stage 1 = fwd{Q1, Q2 (s, a~)} with the updated critics (prologue: finish pi(s) with the second noise draw);
//          stage 2 = bwd{Q1, Q2 -> d(-min Q)/da~} on the same workgroups (y = 0 / 1);  stage 3 = bwd{pi} on workgroups of its own (y = 2);
//          y = 3: advance the replay-draw counter (nothing in this launch reads it).
template <int H, int ACT, int CS, bool CT = false>
__global__ __launch_bounds__(4 * H / CS) void k_sac_phase_c(const PhaseCArgs Pk, int slot) {
  const PhaseCArgs& P = CT ? g_phase_c_tab[slot] : Pk;
  constexpr int GRP = 0; constexpr bool XCH = true;
  constexpr int PH = 0, MT = 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FwdGroup* GP = nullptr;
  const int bx = blockIdx.x, cs = blockIdx.z, y = blockIdx.y;
  if (y == 3) {
    if (bx == 0 && cs == 0 && threadIdx.x == 0) P.f3.tail[0].scal->gather_step += 1;
    if (P.aslot && bx == 1 && cs == 0) alpha_slot_run(P.aslot_logp, P.aslot_B, P.aslot_te, P.aslot_invB, P.aslot, smem);   // workgroup-uniform
    const float tau = P.polyak_tau;
    for (int i = ((int)(cs * gridDim.x + bx) * (int)blockDim.x + (int)threadIdx.x) * 4; i < P.polyak_n; i += (int)(gridDim.x * gridDim.z * blockDim.x) * 4) {
      const float4 p = *reinterpret_cast<const float4*>(P.polyak_P + i);
      float4 t = *reinterpret_cast<const float4*>(P.polyak_T + i);
      t.x = t.x * (1.0f - tau) + p.x * tau; t.y = t.y * (1.0f - tau) + p.y * tau;
      t.z = t.z * (1.0f - tau) + p.z * tau; t.w = t.w * (1.0f - tau) + p.w * tau;
      *reinterpret_cast<float4*>(P.polyak_T + i) = t;
    }
    return;
  }
  if (bx * 16 >= P.f3.rows) return;
  unsigned* f1 = P.flags + (3 * bx) * 32;
  unsigned* f2 = P.flags + (3 * bx + 1) * 32;
  ILSX_STAMP(P.dbg, 0);
  xch_mark_xcd(P.flags + PHASE_MASK_WORD(bx));
  if (y == 2) {
    // the policy's backward has workgroups of its own: weights and saved activations (from phase A's launch) are requested at once,
    // the head gradient waits for both critics' dQ/da partials (stage 2) — and with them for the action / noise / raw head stage 1 published
    {
      const BwdArgs& A = P.b3;
      const BwdTask& T = A.t[0];
#define XCH_HOOK_ACT
#define XCH_HOOK_HEAD xch_wait(f2, 2 * CS, P.err); ILSX_STAMP(P.dbg, 4);
#include "bwd_split_tile.inc"
#undef XCH_HOOK_ACT
#undef XCH_HOOK_HEAD
    }
    ILSX_STAMP(P.dbg, 5);
    return;
  }
  {
    const FwdArgs& A = P.f3;
    const FwdTask& T = A.t[y];
    const bool first_task = y == 0;
#define XCH_HOOK_STAGE
#define XCH_HOOK_FIN
#include "fwd_split_tile.inc"
#undef XCH_HOOK_STAGE
#undef XCH_HOOK_FIN
  }
  ILSX_STAMP(P.dbg, 1);
  xch_arrive(f1);
  {
    const BwdArgs& A = P.b2;
    const BwdTask& T = A.t[y];
#define XCH_HOOK_ACT xch_wait(f1, 2 * CS, P.err); ILSX_STAMP(P.dbg, 2);
#define XCH_HOOK_HEAD
#include "bwd_split_tile.inc"
#undef XCH_HOOK_ACT
#undef XCH_HOOK_HEAD
  }
  ILSX_STAMP(P.dbg, 3);
  xch_arrive(f2);
}
#endif  // ILSX_KERNEL_IMPL

// ================================================================================================
// Weight gradients: dW[n][k] = sum_r A[r][n] * Bm[r][k], db[n] = sum_r A[r][n]  (contraction over
// the batch).  One 1024-thread workgroup per 32(n) x 64(k) output tile: wave = (n-half, row-eighth);
// each wave issues all its operand loads up front, runs its MFMAs, and the 8 row-partials are summed
// through LDS.  dynamic LDS = DW_LDS_BYTES.
enum { DW_OUT_NATURAL = 0, DW_OUT_PACK_F = 1, DW_OUT_PACK_FB = 2 };
// One weight matrix of a dW launch.  The table travels in the kernel arguments (scalar registers), so a
// workgroup finds its tile without a dependent global load.
struct DwMat {
  const float* A; const float* Bm; float* dW; float* dWb; float* db;
  int lda, NA, ldb, NB, ldw, mode;   // ldw: natural row stride, or K (PACK_F); NA rows for PACK_B
  int rows, bias_rows;               // > 0: contract over `rows` (row-stacked jobs); only the first bias_rows feed db
  int tile0, ktiles;                 // first workgroup of this matrix, number of 64-wide k tiles
  int agent;                         // grouped launches: index into DwArgs::fuses
};
#define DW_MAX_MATS 10
// Optional optimiser epilogue of the dW kernel: every workgroup owns its output tile completely (the batch
// contraction happens inside it), so Adam (+ Polyak) can be applied to that tile right there; the separate
// k_adam_polyak pass and a kernel boundary disappear.  Gbase/P/M/V/T are arena bases with identical layouts.
struct AdamFuse {
  const float* Gbase; float* P; float* M; float* V; float* T;   // T nullable (no target network)
  float b1, b2, eps, tau;
  const float* step_size; const float* bc2_sqrt;
  int on;
  float l2x2;   // gradient of an L2 penalty lambda*sum(p^2) folded in: g += l2x2 * p  (ppo.py:147-148, l2x2 = 2*lambda)
};
struct DwArgs {
  DwMat m[DW_MAX_MATS];
  int nmat, rows_all, ntiles, pad;
  AdamFuse F;
  unsigned long long* dbg;
  int xs;
  // large batches: the contraction is split over blockIdx.y row ranges; each range writes its partial gradient arena
  // slab (same layout as G) and k_dw_reduce sums the slabs in a fixed order (+ Adam): deterministic, no atomics
  int splits, rows_per_split;
  float* g_lo; float* g_hi;   // extent of the gradient arena the table's matrices live in (set by build_dw_jobs)
  float* scratch; size_t span;
  // grouped launch: one self-contained record per output tile (its matrix and its agent's optimiser) in device memory
  const struct DwTileG* gtiles;
  unsigned* zero_flags;   // non-null: workgroup 0 zeroes the PHASE_NFLAGS arrival counters of the phase kernel that follows this launch
  const int* zero_err;    //   ... or arms them satisfied (PHASE_FLAG_DEAD) when *zero_err says a wait of this window already gave up
  int tile_nh, tile_kt;   // grouped launches: the tile shape the gtiles table was built for (0 = 2 x 4); host-side only
  int strip;              // grouped launches: the table is one record per 16 x 64 output strip of k_dw_strip (one wavefront each); host-side only
};
struct DwTileG { DwMat J; AdamFuse F; };
#define DW_SPLIT_MIN_ROWS 1024
#define DW_BIG_MIN_ROWS 4096
#define DW_TILE_N 32
#define DW_TILE_K 64

#ifdef ILSX_KERNEL_IMPL
// (Measured and rejected, round 2: nontemporal stores here — no change; system-scope write-through stores — the launch got
//  1.8 us slower and the step lost 3 %.  The 4 us between this launch and the next one is not the L2 write-back of these lines.)
#define ILSX_ST(ptr, val) (*(ptr) = (val))
// Pointers read from a table in device memory (grouped launches) are GENERIC to the compiler, and every access through them a FLAT instruction:
// a 64-bit address built per access, counted on the LDS counter as well as on the vector-memory one (so an LDS wait also waits for them).  G = true
// states at the access what the pointer is — global_load / global_store, as in the single-run instances whose pointers arrive in the kernel arguments.
typedef float __attribute__((address_space(1))) ilsx_gfloat;
#ifdef ILSX_FLAT_TABLE_PTRS   // A/B build (make VAR=flat VARFLAGS=-DILSX_FLAT_TABLE_PTRS): the accesses as the compiler infers them
#define ILSX_G(G) false
#else
#define ILSX_G(G) (G)
#endif
template <bool G>
__device__ __forceinline__ float ld_f(const float* p) {
  if (ILSX_G(G)) return *(const ilsx_gfloat*)p;
  return *p;
}
template <bool G>
__device__ __forceinline__ void st_f(float* p, float v) {
  if (ILSX_G(G)) *(ilsx_gfloat*)p = v;
  else ILSX_ST(p, v);
}
struct AdamOperands { float p, m, v, t; };
template <bool G = false>
__device__ __forceinline__ AdamOperands adam_prefetch(const AdamFuse& F, size_t i0) {
  AdamOperands o;
  o.p = ld_f<G>(F.P + i0); o.m = ld_f<G>(F.M + i0); o.v = ld_f<G>(F.V + i0); o.t = F.T ? ld_f<G>(F.T + i0) : 0.0f;
  return o;
}
template <bool G = false>
__device__ __forceinline__ void adam_apply(const AdamFuse& F, float step, float bc2s, const AdamOperands& o, size_t i0,
                                           size_t i1, bool two, float g) {
  g = g + F.l2x2 * o.p;
  const float m = o.m * F.b1 + (1.0f - F.b1) * g;
  const float v = o.v * F.b2 + (1.0f - F.b2) * g * g;
  const float p = o.p - step * (m / (sqrtf(v) / bc2s + F.eps));
  st_f<G>(F.M + i0, m); st_f<G>(F.V + i0, v); st_f<G>(F.P + i0, p);
  float tg = 0.0f;
  if (F.T) { tg = o.t * (1.0f - F.tau) + p * F.tau; st_f<G>(F.T + i0, tg); }
  if (two) {  // second packing of the same matrix
    st_f<G>(F.M + i1, m); st_f<G>(F.V + i1, v); st_f<G>(F.P + i1, p);
    if (F.T) st_f<G>(F.T + i1, tg);
  }
}

// NH = 16-wide n blocks per workgroup (1 | 2), KT = 16-wide k tiles per wave (2 | 4): output tile (16 NH) x (16 KT), 8 NH waves
// (wave = row-eighth x n block).  Per-element arithmetic and summation order do not depend on NH / KT (bit-identical results); small
// tiles spread a small-batch launch over more CUs at fewer waves per SIMD (launch_bwd_dw picks them for the single-run steps).
#define DW_LDS_BYTES_OF(NH, KT) ((8 * (NH) * 4 * (KT) * 64 + 8 * (NH) * 16) * 4)
// LOW: the same tile in <= 64 registers, for EIGHT waves per SIMD (two 16-wave workgroups per CU; k_mlp_bwd_dw_low): a trip's two 128-row
// steps are requested one after the other and the output pointers + optimiser operands after the contraction instead of at entry — what
// a lone workgroup wins by having everything in flight at once, two resident workgroups win by overlapping their phases.  Same MFMA chain,
// same sums: bit-identical.
template <bool GRP, int NH, int KT, bool LOW>
__device__ __forceinline__ void dw_tile_body(const DwArgs& D, float* smem) {
  constexpr int NW = 8 * NH, NT = 512 * NH, TN = 16 * NH, TK = 16 * KT, EPT = (KT + 1) / 2;   // EPT: output elements per thread (KT = 1: half the threads finish one)
  float* part = smem;                         // [NW waves][4 KT acc regs][64 lanes]
  float* bpart = smem + NW * 4 * KT * 64;     // [NW waves][16]
  if (blockIdx.x & ((1u << D.xs) - 1u)) return;
  const int bx = blockIdx.x >> D.xs;
  if (D.zero_flags && bx == 0 && blockIdx.y == 0 && threadIdx.x < PHASE_NFLAGS)
    D.zero_flags[threadIdx.x * 32] = (D.zero_err && *D.zero_err) ? PHASE_FLAG_DEAD : 0u;
  int mi = 0;
  if (!GRP) {
#pragma unroll
    for (int i = 1; i < DW_MAX_MATS; ++i)
      if (i < D.nmat && bx >= D.m[i].tile0) mi = i;
  }
  DwTileG Rg;
  if (GRP) {
    Rg = D.gtiles[bx];   // by value at entry (see k_mlp2_fwd_split)
  } else { Rg.J = D.m[mi]; Rg.F = D.F; }
  const DwMat& J = Rg.J;
  const AdamFuse& F = Rg.F;
  const int local = bx - J.tile0;
  const int n0 = (local / J.ktiles) * TN, k0 = (local % J.ktiles) * TK;
  int rows = J.rows > 0 ? J.rows : D.rows_all, brows = J.rows > 0 ? J.bias_rows : D.rows_all;
  int r_begin = 0;
  float* out_shift = nullptr;   // split mode: outputs land in this row range's slab
  if (D.splits > 1) {
    r_begin = blockIdx.y * D.rows_per_split;
    rows = min(rows, r_begin + D.rows_per_split);
    brows = rows;
    out_shift = D.scratch + (size_t)blockIdx.y * D.span;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int wn = wave % NH, wr = wave / NH;
  const int nsub = n0 + 16 * wn;
  const bool n_ok = nsub + li < J.NA;
  int ntk = (J.NB - k0 + 15) / 16;
  if (ntk > KT) ntk = KT;
  ILSX_STAMP(D.dbg, 0);
  // ---- the two output elements this thread will finish (and their optimiser operands: requested now, or behind the contraction when LOW)
  const bool packed = J.mode != DW_OUT_NATURAL;
  float* g0p[EPT]; float* g1p[EPT]; bool live[EPT];
  AdamOperands ao[EPT];
  float ad_step = 0.f, ad_bc2s = 1.f;
  bool bias_thread = false;
  float* gbp = nullptr;
  AdamOperands aob;
#define DW_OUTPUT_ELEMS()                                                                                                          \
  {                                                                                                                                \
    if (F.on) { ad_step = ld_f<GRP>(F.step_size); ad_bc2s = ld_f<GRP>(F.bc2_sqrt); }                                                                   \
    _Pragma("unroll") for (int h = 0; h < EPT; ++h) {                                                                              \
      const int e = tid + NT * h, ee = e % (256 * NH * KT);                                                                        \
      const int on = ee / (256 * KT), rest = ee % (256 * KT), t = rest >> 8, q = rest & 255;                                       \
      /* thread <-> element of the 16 x 16 block: packed outputs in ADDRESS order (consecutive lanes = consecutive floats of the   \
         forward packing and whole 64-byte runs of the backward packing: every optimiser stream of the epilogue moves full lines;  \
         the accumulator order - lane = column, register = row - touched 16 of every 64 bytes per instruction); natural outputs    \
         keep the accumulator order */                                                                                             \
      const int v = packed ? (q >> 2) & 3 : (q >> 6) & 3;                                                                          \
      const int ol = packed ? ((q >> 4) & 3) * 16 + 4 * (q >> 6) + (q & 3) : q & 63;                                               \
      const int n = n0 + 16 * on + 4 * (ol >> 4) + v, k = k0 + 16 * t + (ol & 15);                                                 \
      live[h] = e < 256 * NH * KT && t < ntk && n < J.NA && k < J.NB;                                                              \
      g0p[h] = nullptr; g1p[h] = nullptr;                                                                                          \
      if (live[h]) {                                                                                                               \
        g0p[h] = J.mode == DW_OUT_NATURAL ? J.dW + (size_t)n * J.ldw + k : J.dW + pack_f(n, k, J.ldw);                             \
        g1p[h] = J.mode == DW_OUT_PACK_FB ? J.dWb + pack_b(n, k, J.NA) : nullptr;                                                  \
        if (out_shift) {                                                                                                           \
          g0p[h] = out_shift + (g0p[h] - D.g_lo);                                                                                  \
          if (g1p[h]) g1p[h] = out_shift + (g1p[h] - D.g_lo);                                                                      \
        }                                                                                                                          \
        if (F.on) ao[h] = adam_prefetch<GRP>(F, (size_t)(g0p[h] - F.Gbase));                                                            \
      }                                                                                                                            \
    }                                                                                                                              \
    bias_thread = J.db && k0 == 0 && tid < 16 * NH && n0 + 16 * (tid >> 4) + (tid & 15) < J.NA;                                    \
    gbp = bias_thread ? J.db + n0 + 16 * (tid >> 4) + (tid & 15) : nullptr;                                                        \
    if (gbp && out_shift) gbp = out_shift + (gbp - D.g_lo);                                                                        \
    if (bias_thread && F.on) aob = adam_prefetch<GRP>(F, (size_t)(gbp - F.Gbase));                                                      \
  }
  if (!LOW) DW_OUTPUT_ELEMS();

  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.0f;
  bool k_ok[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) k_ok[t] = k0 + 16 * t + li < J.NB;
  // two 128-row steps per trip: both steps' operands are requested before the first MFMA (a 256-row batch is one round of
  // loads per wave instead of two dependent ones); accumulation order is unchanged
  // A tile whose 16 NH x 16 KT block lies inside the matrix, on a trip whose 256 rows lie inside the batch, loads without guards off two
  // base addresses (workgroup-uniform test; guarded, each of a trip's 8 + 8 KT loads was a compare, an EXEC branch and three address
  // instructions — most of what this kernel issued beside its MFMAs); the values and the order of the MFMAs are the same
  const bool tile_in = n0 + TN <= J.NA && k0 + TK <= J.NB;
  for (int rc = r_begin + 16 * wr; rc < rows; rc += 256) {
    float a[2][4], b[2][KT][4];
    const int rbase = rc - 16 * wr;   // first row of the trip (wave-independent)
    if (tile_in && rbase + 256 <= rows && rbase + 256 <= brows) {
      const float* const pa = J.A + (size_t)(rc + 4 * g) * J.lda + nsub + li;
      const float* const pb = J.Bm + (size_t)(rc + 4 * g) * J.ldb + k0 + li;
      if (LOW) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            a[0][s] = ld_f<GRP>(pa + (size_t)(128 * u + s) * J.lda);
#pragma unroll
            for (int t = 0; t < KT; ++t) b[0][t][s] = ld_f<GRP>(pb + (size_t)(128 * u + s) * J.ldb + 16 * t);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = MFMA16(a[0][s], b[0][t][s], acc[t]);
            bsum += a[0][s];
          }
        }
        continue;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          a[u][s] = ld_f<GRP>(pa + (size_t)(128 * u + s) * J.lda);
#pragma unroll
          for (int t = 0; t < KT; ++t) b[u][t][s] = ld_f<GRP>(pb + (size_t)(128 * u + s) * J.ldb + 16 * t);
        }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int t = 0; t < KT; ++t) acc[t] = MFMA16(a[u][s], b[u][t][s], acc[t]);
          bsum += a[u][s];
        }
      continue;
    }
    if (LOW) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = rc + 128 * u + 4 * g + s;
          const bool r_ok = r < rows;
          a[0][s] = (r_ok && n_ok) ? ld_f<GRP>(J.A + (size_t)r * J.lda + nsub + li) : 0.0f;
#pragma unroll
          for (int t = 0; t < KT; ++t)
            b[0][t][s] = (r_ok && k_ok[t]) ? ld_f<GRP>(J.Bm + (size_t)r * J.ldb + k0 + 16 * t + li) : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int t = 0; t < KT; ++t)
            if (t < ntk) acc[t] = MFMA16(a[0][s], b[0][t][s], acc[t]);
          bsum += (rc + 128 * u + 4 * g + s < brows) ? a[0][s] : 0.0f;
        }
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int r = rc + 128 * u + 4 * g + s;
        const bool r_ok = r < rows;
        a[u][s] = (r_ok && n_ok) ? ld_f<GRP>(J.A + (size_t)r * J.lda + nsub + li) : 0.0f;
#pragma unroll
        for (int t = 0; t < KT; ++t)
          b[u][t][s] = (r_ok && k_ok[t]) ? ld_f<GRP>(J.Bm + (size_t)r * J.ldb + k0 + 16 * t + li) : 0.0f;
      }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < KT; ++t)
          if (t < ntk) acc[t] = MFMA16(a[u][s], b[u][t][s], acc[t]);
        bsum += (rc + 128 * u + 4 * g + s < brows) ? a[u][s] : 0.0f;
      }
  }
  ILSX_STAMP(D.dbg, 1);
  // partial tiles -> LDS
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) part[(wave * 4 * KT + t * 4 + v) * 64 + lane] = acc[t][v];
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (g == 0) bpart[wave * 16 + li] = bsum;
  if (LOW) DW_OUTPUT_ELEMS();
#undef DW_OUTPUT_ELEMS
  lds_barrier();
  ILSX_STAMP(D.dbg, 2);
  // sum the 8 row-partials; thread <-> (n-half, tile, reg, lane)
#pragma unroll
  for (int h = 0; h < EPT; ++h) {
    const int e = tid + NT * h, ee = e % (256 * NH * KT);
    const int on = ee / (256 * KT), rest = ee % (256 * KT), t = rest >> 8, q = rest & 255;
    // thread <-> element of the 16 x 16 block: packed outputs in ADDRESS order (consecutive lanes = consecutive floats of the forward
    // packing and whole 64-byte runs of the backward packing: every optimiser stream of the epilogue moves full lines; the accumulator
    // order — lane = column, register = row — touched 16 of every 64 bytes per instruction); natural outputs keep the accumulator order
    const int v = packed ? (q >> 2) & 3 : (q >> 6) & 3;
    const int ol = packed ? ((q >> 4) & 3) * 16 + 4 * (q >> 6) + (q & 3) : q & 63;
    float s = 0.0f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) s += part[((r8 * NH + on) * 4 * KT + t * 4 + v) * 64 + ol];
    if (live[h]) {
      st_f<GRP>(g0p[h], s);
      if (g1p[h] && !F.on) st_f<GRP>(g1p[h], s);   // with the optimiser fused nobody reads the gradient's second packing (get_grads and the flat views read the first): one of the hidden -> hidden matrices' 14 streams less
      if (F.on) adam_apply<GRP>(F, ad_step, ad_bc2s, ao[h], (size_t)(g0p[h] - F.Gbase), g1p[h] ? (size_t)(g1p[h] - F.Gbase) : 0,
                           g1p[h] != nullptr, s);
    }
  }
  if (bias_thread) {
    const int on = tid >> 4, ol = tid & 15;
    float s = 0.0f;
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) s += bpart[(r8 * NH + on) * 16 + ol];
    st_f<GRP>(gbp, s);
    if (F.on) adam_apply<GRP>(F, ad_step, ad_bc2s, aob, (size_t)(gbp - F.Gbase), 0, false, s);
  }
  ILSX_STAMP(D.dbg, 7);
}
template <bool GRP, int NH, int KT>
__global__ __launch_bounds__(512 * NH) void k_mlp_bwd_dw(const DwArgs D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  dw_tile_body<GRP, NH, KT, false>(D, smem);
}
template <bool GRP, int NH, int KT>
__global__ __launch_bounds__(512 * NH) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_mlp_bwd_dw_low(const DwArgs D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  dw_tile_body<GRP, NH, KT, true>(D, smem);
}

// (Measured and rejected, round 4: a 64 x 64 block shape for SAC-sized batches — k_dw_blk: eight waves = the eight row-eighths, every
//  wave covering the whole block with one 16-byte load per lane and dimension feeding 16 MFMAs per 4-row step, tile j = columns
//  n0 + 4 i + j; bit-identical to this kernel and 5x fewer load instructions per MFMA, but 19.4 us against 6.4 us for the single-run
//  launch (48 workgroups instead of 576: the launch is made of latency and of the optimiser epilogue's parallelism, not of MFMAs)
//  and 31.4 against 22.9 us for K = 8 grouped seeds (the 64 accumulator registers leave no room to request a trip's operands and the
//  optimiser streams up front at four waves per SIMD: 136 bytes of scratch and four dependent load rounds per trip).)
// ---- the weight gradients of LARGE batches (PPO's 32768-row minibatches; split mode of launch_bwd_dw).  k_mlp_bwd_dw feeds every MFMA
// with one 4-byte global load per lane and operand (64-byte runs of four different rows per instruction): at thousands of rows per wave
// it is bound by the vector-memory path, not by the MFMA pipe (26 % of the fp32 peak).  Here a 256-thread workgroup owns a 128 (n) x 128 (k)
// block of one matrix over one row range: 32-row chunks of delta[:, n-block] and activations[:, k-block] are fetched with 16-byte
// loads (whole 512-byte row segments per 32 lanes), parked in registers while the previous chunk is consumed, and staged in LDS
// (row stride 144 floats: the four rows a 16x16x4 step reads fall in different banks); each wave owns 64 x 64 outputs = 16 accumulator
// tiles and issues 16 MFMAs per 8 LDS reads.  Per row a block moves 1 KiB for 32 kFLOP.  Partial gradients go to this row range's
// slab in the arena layout (k_dw_reduce sums the slabs in a fixed order and applies Adam): deterministic, no atomics.
#define DWB_RC 32      // rows per staged chunk
#define DWB_LD 144     // LDS row stride (floats)
template <bool FULL>   // FULL: the block's 128 columns exist and rows are 16-byte aligned — one float4 per (row, column quad), no column guards
__device__ __forceinline__ void dwb_fetch(const float* __restrict__ X, int ld, int c0, int ncols, int rc, int r_end, int tid, float (&p)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q, r = rc + (idx >> 5), c = c0 + 4 * (idx & 31);
    const bool r_ok = r < r_end;
    const float* src = X + (size_t)(r_ok ? r : r_end - 1) * ld + c;
    if constexpr (FULL) {
      const float4 v = *reinterpret_cast<const float4*>(src);
      p[4 * q] = r_ok ? v.x : 0.0f; p[4 * q + 1] = r_ok ? v.y : 0.0f; p[4 * q + 2] = r_ok ? v.z : 0.0f; p[4 * q + 3] = r_ok ? v.w : 0.0f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = c + j < ncols;
        const float v = src[ok ? j : -c];   // a column past the matrix reads the row's first element instead (in bounds), and is zeroed
        p[4 * q + j] = (r_ok && ok) ? v : 0.0f;
      }
    }
  }
}
__device__ __forceinline__ void dwb_park(float* S, int tid, const float (&p)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    *reinterpret_cast<float4*>(S + (idx >> 5) * DWB_LD + 4 * (idx & 31)) = make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
  }
}
// one staged chunk: 8 steps of 4 rows; NTI x NKI accumulator tiles of this wave are live (4 x 4 in the hidden -> hidden blocks)
template <int NTI, int NKI>
__device__ __forceinline__ void dwb_chunk(const float* ap, const float* bp, f32x4 (&acc)[4][4], float (&bs)[4]) {
#pragma unroll
  for (int s = 0; s < DWB_RC / 4; ++s) {
    float a[NTI], b[NKI];
#pragma unroll
    for (int i = 0; i < NTI; ++i) a[i] = ap[4 * s * DWB_LD + 16 * i];
#pragma unroll
    for (int t = 0; t < NKI; ++t) b[t] = bp[4 * s * DWB_LD + 16 * t];
#pragma unroll
    for (int i = 0; i < NTI; ++i) {
#pragma unroll
      for (int t = 0; t < NKI; ++t) acc[i][t] = MFMA16(a[i], b[t], acc[i][t]);
      bs[i] += a[i];
    }
  }
}
__global__ __launch_bounds__(256, 2) void k_dw_big(const DwArgs D) {
  __shared__ __attribute__((aligned(16))) float As[DWB_RC * DWB_LD];
  __shared__ __attribute__((aligned(16))) float Bs[DWB_RC * DWB_LD];
  // 1-D grid, block-major: the `splits` row ranges of one block are consecutive workgroups, so the heavy blocks (hidden -> hidden: 8 k
  // tiles per wave against 1 for the first layer and the heads) are dealt round-robin over all 8 XCDs.  (Block index on grid.x with 8
  // blocks per range put every heavy block on the same four XCDs: 90 us per launch.)
  const int bx = blockIdx.x / D.splits, by = blockIdx.x - bx * D.splits;
  int mi = 0;
#pragma unroll
  for (int i = 1; i < DW_MAX_MATS; ++i)
    if (i < D.nmat && bx >= D.m[i].tile0) mi = i;
  const DwMat& J = D.m[mi];   // tile0 / ktiles in 128 x 128 blocks here (launch_bwd_dw re-tiles its copy of the table)
  const int local = bx - J.tile0;
  const int n0 = (local / J.ktiles) * 128, k0 = (local % J.ktiles) * 128;
  const int r_begin = by * D.rows_per_split, r_end = min(D.rows_all, r_begin + D.rows_per_split);
  float* const slab = D.scratch + (size_t)by * D.span;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: say so, the tile counts below steer scalar branches
  const int wn = wave & 1, wk = wave >> 1;
  const int nw0 = n0 + 64 * wn, kw0 = k0 + 64 * wk;
  const int nti = max(0, min(4, (J.NA - nw0 + 15) >> 4)), nki = max(0, min(4, (J.NB - kw0 + 15) >> 4));
  const bool fa = (J.lda & 3) == 0 && (reinterpret_cast<size_t>(J.A) & 15) == 0 && n0 + 128 <= J.NA;
  const bool fb = (J.ldb & 3) == 0 && (reinterpret_cast<size_t>(J.Bm) & 15) == 0 && k0 + 128 <= J.NB;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  float pa[16], pb[16];
  if (fa) dwb_fetch<true>(J.A, J.lda, n0, J.NA, r_begin, r_end, tid, pa); else dwb_fetch<false>(J.A, J.lda, n0, J.NA, r_begin, r_end, tid, pa);
  if (fb) dwb_fetch<true>(J.Bm, J.ldb, k0, J.NB, r_begin, r_end, tid, pb); else dwb_fetch<false>(J.Bm, J.ldb, k0, J.NB, r_begin, r_end, tid, pb);
  const float* ap = As + g * DWB_LD + 64 * wn + li;
  const float* bp = Bs + g * DWB_LD + 64 * wk + li;
  for (int rc = r_begin; rc < r_end; rc += DWB_RC) {
    lds_barrier();   // the previous chunk has been consumed
    dwb_park(As, tid, pa);
    dwb_park(Bs, tid, pb);
    lds_barrier();
    if (rc + DWB_RC < r_end) {   // the next chunk travels while this one is multiplied
      if (fa) dwb_fetch<true>(J.A, J.lda, n0, J.NA, rc + DWB_RC, r_end, tid, pa); else dwb_fetch<false>(J.A, J.lda, n0, J.NA, rc + DWB_RC, r_end, tid, pa);
      if (fb) dwb_fetch<true>(J.Bm, J.ldb, k0, J.NB, rc + DWB_RC, r_end, tid, pb); else dwb_fetch<false>(J.Bm, J.ldb, k0, J.NB, rc + DWB_RC, r_end, tid, pb);
    }
    // the shapes that occur: hidden -> hidden 4 x 4; first layer 4 x 1 (K padded to 16 .. 64: 4 x NKI); heads 1 x 4; anything else 4 x 4 on
    // the zero padding the staging wrote (correct, some idle MFMAs)
    if (nti == 4 && nki == 4) dwb_chunk<4, 4>(ap, bp, acc, bs);
    else if (nti == 4 && nki == 1) dwb_chunk<4, 1>(ap, bp, acc, bs);
    else if (nti == 1 && nki == 4) dwb_chunk<1, 4>(ap, bp, acc, bs);
    else if (nti == 0 || nki == 0) {}
    else dwb_chunk<4, 4>(ap, bp, acc, bs);
  }
  // ---- this row range's partial block -> its slab (arena layout).  D[n = 4g + v][k = li] per accumulator tile
  const size_t off_w = (size_t)(J.dW - D.g_lo), off_wb = J.mode == DW_OUT_PACK_FB ? (size_t)(J.dWb - D.g_lo) : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (i >= nti || t >= nki) continue;
      const int nb = nw0 + 16 * i + 4 * g, k = kw0 + 16 * t + li;
      if (k >= J.NB) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = nb + v;
        if (n < J.NA) slab[off_w + (J.mode == DW_OUT_NATURAL ? (size_t)n * J.ldw + k : (size_t)pack_f(n, k, J.ldw))] = acc[i][t][v];
      }
      if (J.mode == DW_OUT_PACK_FB)   // hidden -> hidden: NA is a multiple of 16, the four n of a lane are one 16-byte word of the backward packing
        *reinterpret_cast<float4*>(slab + off_wb + pack_b(nb, k, J.NA)) = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
    }
  if (J.db && k0 == 0 && wk == 0) {
    const size_t off_b = (size_t)(J.db - D.g_lo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float b = bs[i];
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      const int n = nw0 + 16 * i + li;
      if (g == 0 && n < J.NA) slab[off_b + n] = b;
    }
  }
}

// ---- the weight gradients in throughput shape (grouped launches: K co-resident runs).  ONE wavefront (a 64-thread workgroup) owns a
// 16 (n) x 64 (k) output strip completely: it runs all eight row-eighth contractions k_mlp_bwd_dw spreads over eight waves — eight
// accumulator sets, the same MFMA chain per row-eighth (rows rc + 128 u + 4 g + s, u = 0..1, s = 0..3, per 256-row trip) — and adds
// the eight partial tiles in the order that kernel's LDS reduction adds them (r8 = 0..7), so every gradient, moment and parameter
// is bit-identical to the tiled kernel's.  What goes away: the 16-wave workgroup whose barrier waits for its slowest wave, the 8-way
// reduction through LDS, and 15 of 16 descriptor reads; what stays in LDS is one wave-private 16 x 16 transpose per output block
// (accumulator order -> address order of the packed layouts: every optimiser stream of the epilogue then moves whole 16-byte words
// per lane, 1 KiB per instruction).  The second packing of a hidden -> hidden matrix is updated from ITS OWN copies of P / M / V / T
// (identical to the first packing's by construction: both are written with the same values by every kernel) instead of from values
// carried across lanes.  256 MFMAs per strip.
#define DW_STRIP_LDS_FLOATS (4 * 16 * 20)
template <bool GRP, bool WHOLE>   // WHOLE: the contraction is whole 256-row trips (host-checked), no row guards
__global__ __launch_bounds__(64, 2) void k_dw_strip(const DwArgs D) {
  __shared__ __attribute__((aligned(16))) float tile[DW_STRIP_LDS_FLOATS];   // [4 k tiles][16 n][20: 16 k + pad]
  static_assert(GRP, "the strip shape is a grouped-launch shape");
  const DwTileG Rg = D.gtiles[blockIdx.x];   // by value at entry: scalar loads (blockIdx is uniform)
  const DwMat& J = Rg.J;
  const AdamFuse& F = Rg.F;
  const int local = blockIdx.x - J.tile0;
  const int n0 = (local / J.ktiles) * 16, k0 = (local % J.ktiles) * 64;
  const int rows = J.rows > 0 ? J.rows : D.rows_all, brows = J.rows > 0 ? J.bias_rows : D.rows_all;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const bool n_ok = n0 + li < J.NA;
  int ntk = (J.NB - k0 + 15) / 16;
  if (ntk > 4) ntk = 4;
  bool k_ok[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) k_ok[t] = k0 + 16 * t + li < J.NB;
  float ad_step = 0.f, ad_bc2s = 1.f;
  if (F.on) { ad_step = *F.step_size; ad_bc2s = *F.bc2_sqrt; }
  f32x4 acc[8][4];
  float bs[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    bs[w] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[w][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = J.db != nullptr && k0 == 0;
  // operand addresses: one pointer per lane and operand column (columns past the matrix are clamped onto its last one and the value
  // dropped by a select, so that no load sits under a branch: a predicated load costs ~8 instructions of mask handling, and this kernel
  // lives on instruction issue); rows need no clamp when the batch is whole 256-row trips (the SAC batch), else the guarded form below
  // (the table's pointers come out of memory, so the compiler knows no address space for them and would emit FLAT loads, which count on
  //  both memory counters and force full waits between batches; they are global memory)
  typedef const float __attribute__((address_space(1)))* gfptr;
  const gfptr pa = (gfptr)(J.A + (size_t)(4 * g) * J.lda + (n_ok ? n0 + li : J.NA - 1));
  gfptr pb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pb[t] = (gfptr)(J.Bm + (size_t)(4 * g) * J.ldb + (k_ok[t] ? k0 + 16 * t + li : J.NB - 1));
  for (int rc0 = 0; rc0 < rows; rc0 += 256) {
#pragma unroll
    for (int w = 0; w < 8; ++w) {   // row-eighth w of this 256-row trip: rows rc0 + 16 w + {0..15} and + 128
      const int rc = rc0 + 16 * w;
      if (rc >= rows) continue;   // uniform (the tiled kernel's wave w does not take this trip either)
      float a[2][4], b[2][4][4];
      if constexpr (WHOLE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const size_t ro = (size_t)(rc + 128 * u + s4);   // uniform part of the row index (the lane's 4 g is in the pointers)
            const float av = pa[ro * J.lda];
            a[u][s4] = n_ok ? av : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) { const float bv = pb[t][ro * J.ldb]; b[u][t][s4] = k_ok[t] ? bv : 0.0f; }
          }
      } else {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int r = rc + 128 * u + 4 * g + s4;
          const bool r_ok = r < rows;
          a[u][s4] = (r_ok && n_ok) ? J.A[(size_t)r * J.lda + n0 + li] : 0.0f;
#pragma unroll
          for (int t = 0; t < 4; ++t) b[u][t][s4] = (r_ok && k_ok[t]) ? J.Bm[(size_t)r * J.ldb + k0 + 16 * t + li] : 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[w][t] = MFMA16(a[u][s4], b[u][t][s4], acc[w][t]);   // k tiles past the matrix contract zeros and are not stored
        }
      if (want_bias) {   // uniform; same order of additions as the tiled kernel (u, then s)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            if constexpr (WHOLE) bs[w] += a[u][s4];   // whole trips and no row-stacked jobs (host-checked): every row feeds the bias
            else bs[w] += (rc + 128 * u + 4 * g + s4 < brows) ? a[u][s4] : 0.0f;
          }
      }
    }
  }
  // ---- the eight partial tiles in r8 order (k_mlp_bwd_dw: s = 0; s += part[r8])
  f32x4 out[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    out[t] = (f32x4){0.f, 0.f, 0.f, 0.f};   // (0 + part[0] like the tiled kernel: a -0 partial ends as +0 there too)
#pragma unroll
    for (int w = 0; w < 8; ++w) { out[t][0] += acc[w][t][0]; out[t][1] += acc[w][t][1]; out[t][2] += acc[w][t][2]; out[t][3] += acc[w][t][3]; }
  }
  // accumulator order (lane = column k, register = row n = 4 g + v) -> LDS [t][n][k]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) tile[(t * 16 + 4 * g + v) * 20 + li] = out[t][v];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wavefront: its LDS operations execute in order, this only keeps the compiler from moving the reads up
  const int nl = lane & 15, q4 = lane >> 4;   // address order of a forward-packed / natural block: lane <-> (n = lane % 16, k = 4 (lane / 16) .. + 3)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t >= ntk) continue;   // uniform
    const int n = n0 + nl, k = k0 + 16 * t + 4 * q4;
    const float4 gv = *reinterpret_cast<const float4*>(tile + (t * 16 + nl) * 20 + 4 * q4);
    const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
    if (J.mode == DW_OUT_NATURAL) {   // row-major [NA][ldw]: heads (NB = H: k + 3 < NB whenever k < NB, H is a multiple of 16)
      if (n < J.NA && k < J.NB) {
        float* gp = J.dW + (size_t)n * J.ldw + k;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (k + c < J.NB) {
            gp[c] = gs[c];
            if (F.on) { const AdamOperands o = adam_prefetch(F, (size_t)(gp + c - F.Gbase)); adam_apply(F, ad_step, ad_bc2s, o, (size_t)(gp + c - F.Gbase), 0, false, gs[c]); }
          }
        }
      }
    } else if (n < J.NA && k < J.NB) {   // forward-packed block: 4 consecutive floats per lane (k % 4 = 0..3).  NA, NB are multiples of 16 for packed matrices
      float* gp = J.dW + pack_f(n, k, J.ldw);
      const size_t i0 = (size_t)(gp - F.Gbase);
      float4 p4, m4, v4, t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (F.on) {
        p4 = *reinterpret_cast<const float4*>(F.P + i0); m4 = *reinterpret_cast<const float4*>(F.M + i0); v4 = *reinterpret_cast<const float4*>(F.V + i0);
        if (F.T) t4 = *reinterpret_cast<const float4*>(F.T + i0);
      }
      *reinterpret_cast<float4*>(gp) = gv;
      if (F.on) {
        float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // adam_apply's expressions, in its order
          const float gg = gs[c] + F.l2x2 * pp[c];
          const float m = mm[c] * F.b1 + (1.0f - F.b1) * gg;
          const float v = vv[c] * F.b2 + (1.0f - F.b2) * gg * gg;
          const float pn = pp[c] - ad_step * (m / (sqrtf(v) / ad_bc2s + F.eps));
          mm[c] = m; vv[c] = v; pp[c] = pn;
          if (F.T) tt[c] = tt[c] * (1.0f - F.tau) + pn * F.tau;
        }
        *reinterpret_cast<float4*>(F.M + i0) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *reinterpret_cast<float4*>(F.V + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<float4*>(F.P + i0) = make_float4(pp[0], pp[1], pp[2], pp[3]);
        if (F.T) *reinterpret_cast<float4*>(F.T + i0) = make_float4(tt[0], tt[1], tt[2], tt[3]);
      }
    }
    if (J.mode == DW_OUT_PACK_FB) {   // second packing: lane <-> (k = lane % 16, n = 4 (lane / 16) .. + 3); its own P / M / V / T copies
      const int kb = k0 + 16 * t + nl, nb = n0 + 4 * q4;
      if (nb < J.NA && kb < J.NB) {
        float gb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) gb[c] = tile[(t * 16 + 4 * q4 + c) * 20 + nl];
        float* gp = J.dWb + pack_b(nb, kb, J.NA);
        const size_t i1 = (size_t)(gp - F.Gbase);
        float4 p4, m4, v4, t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (F.on) {
          p4 = *reinterpret_cast<const float4*>(F.P + i1); m4 = *reinterpret_cast<const float4*>(F.M + i1); v4 = *reinterpret_cast<const float4*>(F.V + i1);
          if (F.T) t4 = *reinterpret_cast<const float4*>(F.T + i1);
        }
        *reinterpret_cast<float4*>(gp) = make_float4(gb[0], gb[1], gb[2], gb[3]);
        if (F.on) {
          float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float gg = gb[c] + F.l2x2 * pp[c];
            const float m = mm[c] * F.b1 + (1.0f - F.b1) * gg;
            const float v = vv[c] * F.b2 + (1.0f - F.b2) * gg * gg;
            const float pn = pp[c] - ad_step * (m / (sqrtf(v) / ad_bc2s + F.eps));
            mm[c] = m; vv[c] = v; pp[c] = pn;
            if (F.T) tt[c] = tt[c] * (1.0f - F.tau) + pn * F.tau;
          }
          *reinterpret_cast<float4*>(F.M + i1) = make_float4(mm[0], mm[1], mm[2], mm[3]);
          *reinterpret_cast<float4*>(F.V + i1) = make_float4(vv[0], vv[1], vv[2], vv[3]);
          *reinterpret_cast<float4*>(F.P + i1) = make_float4(pp[0], pp[1], pp[2], pp[3]);
          if (F.T) *reinterpret_cast<float4*>(F.T + i1) = make_float4(tt[0], tt[1], tt[2], tt[3]);
        }
      }
    }
  }
  // ---- bias gradients (strips with k0 == 0): per row-eighth the sum over the four lane groups as the tiled kernel's two shuffles form it,
  //      then the eighths in r8 order
  if (want_bias) {   // uniform
    float sb = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      float x = bs[w];
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      sb += x;
    }
    if (g == 0 && n_ok) {
      float* gbp = J.db + n0 + li;
      *gbp = sb;
      if (F.on) { const AdamOperands o = adam_prefetch(F, (size_t)(gbp - F.Gbase)); adam_apply(F, ad_step, ad_bc2s, o, (size_t)(gbp - F.Gbase), 0, false, sb); }
    }
  }
}

// Sum the row-range slabs of a split weight-gradient launch in slab order, store the gradient, and apply the optimiser
// epilogue (Adam + L2 + Polyak) elementwise: the arena is a flat vector, both packings of a matrix carry identical
// gradients and moments, so they stay identical.
struct DwReduceArgs { const float* scratch; int splits; size_t span; float* g_lo; AdamFuse F; size_t off0; };
__global__ __launch_bounds__(256) void k_dw_reduce(const DwReduceArgs R) {
  // wave w of a workgroup sums slabs [w q, (w+1) q) of one run of 64 float4 (loads batched eight at a time), the four partial sums are
  // added in wave order: a fixed order for every element whatever the grid
  __shared__ float4 part[3][64];
  const AdamFuse& F = R.F;
  float step = 0.f, bc2s = 1.f;
  if (F.on) { step = *F.step_size; bc2s = *F.bc2_sqrt; }
  const size_t n4 = R.span >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = (R.splits + 3) >> 2, y0 = wave * q, y1 = min(R.splits, y0 + q);
  for (size_t base = (size_t)blockIdx.x * 64; base < n4; base += (size_t)gridDim.x * 64) {   // workgroup-uniform trip count
    const size_t i = base + lane;
    const bool ok = i < n4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float4* src = reinterpret_cast<const float4*>(R.scratch) + i;
      int y = y0;
      for (; y + 8 <= y1; y += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(y + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; y < y1; ++y) { const float4 v = src[(size_t)y * n4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    if (wave) part[wave - 1][lane] = s;
    lds_barrier();
    if (wave == 0 && ok) {
#pragma unroll
      for (int w = 0; w < 3; ++w) { const float4 v = part[w][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      reinterpret_cast<float4*>(R.g_lo)[i] = s;
      if (F.on) {
        const float gs[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const size_t j = R.off0 + 4 * i + c;
          AdamOperands o;
          o.p = F.P[j]; o.m = F.M[j]; o.v = F.V[j]; o.t = F.T ? F.T[j] : 0.0f;
          adam_apply(F, step, bc2s, o, j, 0, false, gs[c]);
        }
      }
    }
    lds_barrier();
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

