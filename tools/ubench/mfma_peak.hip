// What v_mfma_f32_16x16x4_f32 sustains on this part: independent accumulator chains, no memory traffic, for launches of ~50 us
// (the length of the framework's large-batch kernels) up to milliseconds, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float seed) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){seed, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;   // never true: keeps the chain alive
}
template <int NACC>
static void run(int wgs_per_cu, int iters, int n_cu, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = wgs_per_cu * n_cu;
  k_mfma<NACC><<<grid, 256>>>(out, iters, 0.5f);
  hipDeviceSynchronize();
  float best = 1e30f, first = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k_mfma<NACC><<<grid, 256>>>(out, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) first = ms;
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * 4 * iters * NACC * 2048.0;   // 4 waves x MFMAs x 2*16*16*4
  printf("acc chains %2d, %d waves/SIMD, %7d MFMAs/wave: %8.1f us (first %8.1f)  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters * NACC, best * 1e3, first * 1e3,
         flop / (best * 1e-3) / 1e12);
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("%s: %d CUs, clock %d MHz\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000);
  float* out; hipMalloc(&out, 4);
  const int n_cu = pr.multiProcessorCount;
  for (int iters : {100, 400, 1600, 25600}) {
    run<16>(1, iters, n_cu, out);
    run<16>(2, iters / 2, n_cu, out);
    run<8>(4, iters / 2, n_cu, out);
  }
  run<4>(1, 6400, n_cu, out);
  run<1>(1, 25600, n_cu, out);
  return 0;
}
