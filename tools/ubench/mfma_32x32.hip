// v_mfma_f32_32x32x2_f32 beside v_mfma_f32_16x16x4_f32 ("half the MFMA instructions and half the operand fetches per FLOP"):
//   1. what each form sustains (independent accumulator chains, no memory traffic);
//   2. the cost of other instructions issued between a wave's own MFMAs, at EQUAL FLOPs (n fmas per 16x16x4 == 2n per 32x32x2);
//   3. a second wave of the same SIMD doing VALU work beside a wave of back-to-back 32x32x2 (sum = no overlap, max = full overlap);
//   4. are the two forms the same fp32 fma chain?  C[32x32] = A[32x16] B[16x32] with k ascending through either form, compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_32x32 mfma_32x32.hip && ./mfma_32x32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FORM, int NFMA, int LDSRD>   // FORM 16 / 32; NFMA fmas (and LDSRD 16-byte LDS reads) after every MFMA
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
  __syncthreads();
  float a = seed + threadIdx.x, b = seed - threadIdx.x, s = 0.f;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
  if constexpr (FORM == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFMA; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < LDSRD; ++j) { const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i + j) * 64) & 4095)); v[(i + j) & 15] += q.x + q.w; }
        if constexpr (NFMA + LDSRD > 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = (f32x16){0}; acc[i][0] = seed; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFMA; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
#pragma unroll
        for (int j = 0; j < LDSRD; ++j) { const float4 q = *reinterpret_cast<const float4*>(lds + ((threadIdx.x * 4 + (i + j) * 64) & 4095)); v[(i + j) & 15] += q.x + q.w; }
        if constexpr (NFMA + LDSRD > 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) out[0] = s;
}
template <int FORM, int NFMA, int LDSRD>
static float run(float* out, int iters, int n_cu) {   // one wave per SIMD: 256-thread workgroup per CU; 16 (form 16) / 8 (form 32) MFMAs per iteration = equal FLOPs
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mix<FORM, NFMA, LDSRD><<<n_cu, 256>>>(out, iters, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k_mix<FORM, NFMA, LDSRD><<<n_cu, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}
// two waves per SIMD: waves 0-3 back-to-back MFMAs of FORM, waves 4-7 an fma chain
template <int FORM, bool A_ON, bool B_ON>
__global__ __launch_bounds__(512) void k_pair(float* out, int iters, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float s = 0.f;
  if (wave < 4) {
    if constexpr (A_ON) {
      float a = seed + threadIdx.x, b = seed - threadIdx.x;
      if constexpr (FORM == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0];
      } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i] = (f32x16){0}; acc[i][0] = seed; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0];
      }
    }
  } else if constexpr (B_ON) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters * 16; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  if (s == 12345.678f) out[0] = s;
}
template <int FORM, bool A_ON, bool B_ON>
static float runp(float* out, int iters, int n_cu) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_pair<FORM, A_ON, B_ON><<<n_cu, 512>>>(out, iters, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k_pair<FORM, A_ON, B_ON><<<n_cu, 512>>>(out, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}
// 4. one wave: C = A B, k ascending, through either form
__global__ __launch_bounds__(64) void k_same(const float* A, const float* B, float* C16, float* C32) {   // A[32][16], B[16][32] row-major
  const int l = threadIdx.x;
  {   // 16x16x4: four 16 x 16 quadrants, 4 MFMAs each (k = 4m + (l >> 4))
    for (int qi = 0; qi < 2; ++qi)
      for (int qj = 0; qj < 2; ++qj) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < 4; ++m) {
          const int k = 4 * m + (l >> 4);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * qi + (l & 15)) * 16 + k], B[k * 32 + 16 * qj + (l & 15)], acc, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) C16[(16 * qi + 4 * (l >> 4) + r) * 32 + 16 * qj + (l & 15)] = acc[r];   // D: row = 4*(l>>4)+r, col = l&15
      }
  }
  {   // 32x32x2: 8 MFMAs (k = 2s + (l >> 5))
    f32x16 acc = {0};
    for (int s = 0; s < 8; ++s) {
      const int k = 2 * s + (l >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 16 + k], B[k * 32 + (l & 31)], acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C32[(8 * (r >> 2) + 4 * (l >> 5) + (r & 3)) * 32 + (l & 31)] = acc[r];   // D: row = 8*(r/4) + 4*(l>>5) + r%4, col = l&31
  }
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int n_cu = pr.multiProcessorCount, iters = 400;
  float* out; hipMalloc(&out, 4);
  printf("%s: %d CUs.  One wave per SIMD, %d MFMAs of 16x16x4 == %d of 32x32x2 per wave (equal FLOPs); times in us\n", pr.name, n_cu, iters * 16, iters * 8);
  printf("back-to-back              16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 0, 0>(out, iters, n_cu), run<32, 0, 0>(out, iters, n_cu));
  printf("+ 1 | 2 fma per MFMA      16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 1, 0>(out, iters, n_cu), run<32, 2, 0>(out, iters, n_cu));
  printf("+ 2 | 4 fma per MFMA      16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 2, 0>(out, iters, n_cu), run<32, 4, 0>(out, iters, n_cu));
  printf("+ 4 | 8 fma per MFMA      16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 4, 0>(out, iters, n_cu), run<32, 8, 0>(out, iters, n_cu));
  printf("+ 8 | 16 fma per MFMA     16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 8, 0>(out, iters, n_cu), run<32, 16, 0>(out, iters, n_cu));
  printf("+ 1 | 2 ds_read_b128      16x16x4 %7.1f | 32x32x2 %7.1f\n", run<16, 0, 1>(out, iters, n_cu), run<32, 0, 2>(out, iters, n_cu));
  printf("+ 1 | 1 ds_read_b128      16x16x4 %7.1f | 32x32x2 %7.1f   (32x32x2 also halves the operand fetches)\n", run<16, 0, 1>(out, iters, n_cu), run<32, 0, 1>(out, iters, n_cu));
  printf("+ 4 fma | 4 fma           16x16x4 %7.1f | 32x32x2 %7.1f   (per-MFMA overhead that does not scale with the tile)\n", run<16, 4, 0>(out, iters, n_cu), run<32, 4, 0>(out, iters, n_cu));
  for (int form : {16, 32}) {
    const float ta = form == 16 ? runp<16, true, false>(out, iters, n_cu) : runp<32, true, false>(out, iters, n_cu);
    const float tb = runp<16, false, true>(out, iters, n_cu);
    const float tab = form == 16 ? runp<16, true, true>(out, iters, n_cu) : runp<32, true, true>(out, iters, n_cu);
    printf("two waves per SIMD: A = %dx MFMA alone %6.1f | B = valu alone %6.1f | together %6.1f (sum %.1f, max %.1f)\n", form, ta, tb, tab, ta + tb, ta > tb ? ta : tb);
  }
  std::vector<float> hA(32 * 16), hB(16 * 32), c16(1024), c32(1024);
  srand(7);
  int worst = 0;
  for (int trial = 0; trial < 50; ++trial) {
    for (auto& x : hA) x = (float)rand() / RAND_MAX * 4.f - 2.f;
    for (auto& x : hB) x = (float)rand() / RAND_MAX * 4.f - 2.f;
    float *dA, *dB, *d16, *d32;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&d16, 4096); hipMalloc(&d32, 4096);
    hipMemcpy(dA, hA.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 2048, hipMemcpyHostToDevice);
    k_same<<<1, 64>>>(dA, dB, d16, d32);
    hipMemcpy(c16.data(), d16, 4096, hipMemcpyDeviceToHost); hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost);
    int diff = 0, host_diff = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        float ref = 0.f;
        for (int k = 0; k < 16; ++k) ref = fmaf(hA[i * 16 + k], hB[k * 32 + j], ref);
        diff += memcmp(&c16[i * 32 + j], &c32[i * 32 + j], 4) != 0;
        host_diff += memcmp(&c16[i * 32 + j], &ref, 4) != 0;
      }
    if (diff + host_diff > worst) worst = diff + host_diff;
    if (trial == 0) printf("same chain?  16x16x4 vs 32x32x2: %d of 1024 elements differ; 16x16x4 vs host fmaf chain (k ascending): %d differ\n", diff, host_diff);
    hipFree(dA); hipFree(dB); hipFree(d16); hipFree(d32);
  }
  printf("50 random trials: worst differing-element count %d\n", worst);
  return 0;
}
