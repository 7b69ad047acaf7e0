// Can anything run beside a wave that issues back-to-back fp32 MFMAs on its SIMD?  512-thread workgroups, one per CU: waves 0-3 (one per
// SIMD) run role A, waves 4-7 role B.  Times: each role alone, both together (sum = no overlap, max = full overlap); then ONE wave
// per SIMD with other instructions interleaved between its own MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { IDLE, MFMA, VALU, LDS, GLOAD, MFMA_NOP1, MFMA_NOP3, VALU_PRIO, MIX_FMA2, MIX_FMA4, MIX_FMA6, MIX_LDS, MIX_GLOAD, NROLES };
static const char* NAMES[] = {"idle", "mfma", "valu", "lds", "gload", "mfma + s_nop 7", "mfma + 3 s_nop 7", "valu at s_setprio 3",
                              "mfma | 2 fma", "mfma | 4 fma", "mfma | 6 fma", "mfma | ds_read_b128", "mfma | global_load (1 per 16)"};
template <int ROLE>
__device__ __forceinline__ float do_role(int iters, float seed, float* lds, const int* chase, const float* gbuf) {
  float s = 0.f;
  if constexpr (ROLE == MFMA || ROLE == MFMA_NOP1 || ROLE == MFMA_NOP3 || ROLE >= MIX_FMA2) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed - threadIdx.x;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if constexpr (ROLE == MFMA_NOP1) asm volatile("s_nop 7");
        if constexpr (ROLE == MFMA_NOP3) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
        if constexpr (ROLE == MIX_FMA2 || ROLE == MIX_FMA4 || ROLE == MIX_FMA6) {
          constexpr int K = ROLE == MIX_FMA2 ? 2 : ROLE == MIX_FMA4 ? 4 : 6;
#pragma unroll
          for (int j = 0; j < K; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
        }
        if constexpr (ROLE == MIX_LDS) { const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + i * 64) & 4095)); v[i % 6] += q.x + q.w; }
        if constexpr (ROLE == MIX_GLOAD) { if (i == 0) v[0] += gbuf[(size_t)(it & 1023) * 512 + threadIdx.x]; }
        if constexpr (ROLE >= MIX_FMA2) __builtin_amdgcn_sched_barrier(0);   // keep the interleave as written
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += v[i];
  } else if constexpr (ROLE == VALU || ROLE == VALU_PRIO) {
    if constexpr (ROLE == VALU_PRIO) __builtin_amdgcn_s_setprio(3);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  } else if constexpr (ROLE == LDS) {
    for (int it = 0; it < iters * 16; ++it) {
      const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + it * 64) & 4095));
      s += q.x + q.y + q.z + q.w;
    }
  } else if constexpr (ROLE == GLOAD) {
    int p = threadIdx.x & 63;
    for (int it = 0; it < iters / 4; ++it) p = chase[p];
    s = (float)p;
  }
  return s;
}
template <int RA, int RB>
__global__ __launch_bounds__(512) void k_overlap(float* out, const int* chase, const float* gbuf, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed + i;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s;
  if (wave < 4) s = do_role<RA>(iters, seed, lds, chase, gbuf); else s = do_role<RB>(iters, seed, lds, chase, gbuf);
  if (s == 12345.678f) out[0] = s;
}
template <int RA, int RB>
static float run(float* out, const int* chase, const float* gbuf, int iters, int n_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_overlap<RA, RB><<<n_cu, 512>>>(out, chase, gbuf, iters, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k_overlap<RA, RB><<<n_cu, 512>>>(out, chase, gbuf, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}
template <int RA, int RB>
static void pair(float* out, const int* chase, const float* gbuf, int iters, int n_cu) {
  const float ta = run<RA, IDLE>(out, chase, gbuf, iters, n_cu), tb = run<IDLE, RB>(out, chase, gbuf, iters, n_cu), tab = run<RA, RB>(out, chase, gbuf, iters, n_cu);
  printf("A = %-18s alone %6.1f us | B = %-20s alone %6.1f us | together %6.1f us  (sum %.1f, max %.1f)\n", NAMES[RA], ta, NAMES[RB], tb, tab, ta + tb, ta > tb ? ta : tb);
}
template <int RA>
static void solo(float* out, const int* chase, const float* gbuf, int iters, int n_cu) {
  printf("one wave per SIMD: %-32s %6.1f us\n", NAMES[RA], run<RA, IDLE>(out, chase, gbuf, iters, n_cu));
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int n_cu = pr.multiProcessorCount, iters = 400;
  float* out; hipMalloc(&out, 4);
  float* gbuf; hipMalloc(&gbuf, 1024 * 512 * 4); hipMemset(gbuf, 0, 1024 * 512 * 4);
  int* chase; hipMalloc(&chase, 1 << 22);
  int* h = (int*)malloc(1 << 22);
  for (int i = 0; i < (1 << 20); ++i) h[i] = (int)(((long long)i * 40503 + 12345) & ((1 << 20) - 1));
  hipMemcpy(chase, h, 1 << 22, hipMemcpyHostToDevice);
  pair<MFMA, VALU>(out, chase, gbuf, iters, n_cu);
  pair<MFMA, LDS>(out, chase, gbuf, iters, n_cu);
  pair<MFMA, GLOAD>(out, chase, gbuf, iters, n_cu);
  pair<MFMA, VALU_PRIO>(out, chase, gbuf, iters, n_cu);
  pair<MFMA_NOP1, VALU>(out, chase, gbuf, iters, n_cu);
  pair<MFMA_NOP3, VALU>(out, chase, gbuf, iters, n_cu);
  pair<VALU, VALU>(out, chase, gbuf, iters, n_cu);
  pair<MFMA, MFMA>(out, chase, gbuf, iters, n_cu);
  solo<MFMA>(out, chase, gbuf, iters, n_cu);
  solo<MIX_FMA2>(out, chase, gbuf, iters, n_cu);
  solo<MIX_FMA4>(out, chase, gbuf, iters, n_cu);
  solo<MIX_FMA6>(out, chase, gbuf, iters, n_cu);
  solo<MIX_LDS>(out, chase, gbuf, iters, n_cu);
  solo<MIX_GLOAD>(out, chase, gbuf, iters, n_cu);
  return 0;
}
