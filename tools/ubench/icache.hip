// Microbenchmark (gfx950): what does straight-line code cost on a launch that starts with a cold instruction cache?
// The SAC step's kernels execute 2-4k instructions ONCE per workgroup (one wave per SIMD), ~8k lines of ISA each.  Here the same
// arithmetic (N fused multiply-adds per lane, one dependent chain of 4 accumulators) runs (a) fully unrolled (N instructions of
// code) and (b) as a loop over a 64-instruction body; 256 workgroups x 256 threads, 8 alternating launches per graph replay so
// that every launch finds another kernel's code in the instruction cache.   hipcc --offload-arch=gfx950 -O3 -o icache icache.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N, int SALT>
__global__ __launch_bounds__(256) void k_unrolled(float* out, float a, float b) {
  float x0 = threadIdx.x * 1e-3f + SALT, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    x0 = fmaf(x0, a, b + (float)(i & 7)); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int N, int SALT>
__global__ __launch_bounds__(256) void k_looped(float* out, float a, float b, int trips) {
  float x0 = threadIdx.x * 1e-3f + SALT, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
#pragma unroll 1
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x0 = fmaf(x0, a, b + (float)(i & 7)); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <class F>
static int time_graph(hipStream_t st, const char* name, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < 8; ++s) launch(s);
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipEventRecord(e0, st));
  for (int i = 0; i < 200; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipEventRecord(e1, st));
  CHK(hipStreamSynchronize(st));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %.2f us per launch\n", name, 1e3 * ms / 200 / 8);
  return 0;
}

int main() {
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float* out;
  CHK(hipMalloc(&out, 256 * 256 * 4));
  const dim3 G(256), B(256);
#define ALT(KA, KB) [&](int s) { if (s & 1) hipLaunchKernelGGL(KA, G, B, 0, st, out, 1.0001f, 1e-3f); else hipLaunchKernelGGL(KB, G, B, 0, st, out, 1.0001f, 1e-3f); }
#define ALTL(KA, KB, T) [&](int s) { if (s & 1) hipLaunchKernelGGL(KA, G, B, 0, st, out, 1.0001f, 1e-3f, T); else hipLaunchKernelGGL(KB, G, B, 0, st, out, 1.0001f, 1e-3f, T); }
  if (time_graph(st, "unrolled  1024 fma  (two kernels alternate)", ALT((k_unrolled<1024, 1>), (k_unrolled<1024, 2>)))) return 1;
  if (time_graph(st, "looped    1024 fma", ALTL((k_looped<1024, 1>), (k_looped<1024, 2>), 16))) return 1;
  if (time_graph(st, "unrolled  4096 fma", ALT((k_unrolled<4096, 1>), (k_unrolled<4096, 2>)))) return 1;
  if (time_graph(st, "looped    4096 fma", ALTL((k_looped<4096, 1>), (k_looped<4096, 2>), 64))) return 1;
  if (time_graph(st, "unrolled  8192 fma", ALT((k_unrolled<8192, 1>), (k_unrolled<8192, 2>)))) return 1;
  if (time_graph(st, "looped    8192 fma", ALTL((k_looped<8192, 1>), (k_looped<8192, 2>), 128))) return 1;
  if (time_graph(st, "unrolled  8192 fma, same kernel every launch", ALT((k_unrolled<8192, 1>), (k_unrolled<8192, 1>)))) return 1;
  return 0;
}
