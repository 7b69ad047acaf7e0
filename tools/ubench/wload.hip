// Microbenchmark: how fast can ONE workgroup (1024 threads) pull a 256 KB weight matrix out of L2 into
// registers on gfx950, for the access patterns the MLP kernels could use?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pattern 0: row-fragment (lane li -> row n0+li, 16B at col 16c+4g): 16 x 64B segments per instruction
// pattern 1: packed (1 KB contiguous per instruction)
// pattern 2: dword column pattern (bwd): W[(16c+4g+s)*256 + c0+li]
template <int PAT>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ W, float* out, unsigned long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  if (PAT == 0) {
    const float* wp = W + (size_t)(wave * 16 + li) * 256 + 4 * g;
    float4 r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = *reinterpret_cast<const float4*>(wp + 16 * c);
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += r[c].x + r[c].y + r[c].z + r[c].w;
  } else if (PAT == 1) {
    const float* wp = W + (size_t)wave * 4096 + lane * 4;
    float4 r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) r[c] = *reinterpret_cast<const float4*>(wp + 256 * c);
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += r[c].x + r[c].y + r[c].z + r[c].w;
  } else {
    const float* wp = W + (size_t)(4 * g) * 256 + wave * 16 + li;
    float r[16][4];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) r[c][s] = wp[(size_t)(16 * c + s) * 256];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += r[c][0] + r[c][1] + r[c][2] + r[c][3];
  }
  out[blockIdx.x * 1024 + tid] = acc;
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT>
int run(const char* name, const float* W, float* out, unsigned long long* cyc, int nwg) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<PAT>, dim3(nwg), dim3(1024), 0, 0, W, out, cyc);
  CHK(hipEventRecord(a));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<PAT>, dim3(nwg), dim3(1024), 0, 0, W, out, cyc);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(nwg);
  CHK(hipMemcpy(h.data(), cyc, nwg * 8, hipMemcpyDeviceToHost));
  unsigned long long mx = 0, sum = 0;
  for (auto v : h) { mx = v > mx ? v : mx; sum += v; }
  printf("%-28s nwg=%3d  launch avg %.2f us   in-kernel cycles avg %llu max %llu  (256 KB/WG)\n", name, nwg, ms * 1e3 / reps,
         sum / nwg, mx);
  return 0;
}

int main() {
  float *W, *out;
  unsigned long long* cyc;
  CHK(hipMalloc(&W, 256 * 256 * 4));
  CHK(hipMalloc(&out, 256 * 1024 * 4));
  CHK(hipMalloc(&cyc, 256 * 8));
  CHK(hipMemset(W, 0, 256 * 256 * 4));
  for (int nwg : {1, 16, 48, 128}) {
    if (run<0>("row-fragment float4 (fwd)", W, out, cyc, nwg)) return 1;
    if (run<1>("packed 1KB/instr", W, out, cyc, nwg)) return 1;
    if (run<2>("dword column (bwd)", W, out, cyc, nwg)) return 1;
  }
  return 0;
}
