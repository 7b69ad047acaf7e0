// Microbenchmark (gfx950): what does a dependent stage cost as (a) the next kernel of a captured hipGraph and (b) a grid-wide
// barrier inside ONE persistent kernel?  The SAC step is 8 dependent stages of ~256 workgroups; this decides whether folding the
// stages into one kernel pays.   hipcc --offload-arch=gfx950 -O3 -o gridsync gridsync.hip
//
// Every stage does the same small amount of "work" that forces real cross-workgroup (cross-XCD) visibility: workgroup b writes
// 1 KiB to slab[stage][b] and reads the 1 KiB workgroup (b + 37) % n wrote in the previous stage (checked at the end).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void stage_work(float* slab, int stage, int nblocks, float* acc, int extra_bytes, const float* pad) {
  const int b = blockIdx.x, t = threadIdx.x;
  float v = (float)(stage + 1);
  if (stage > 0) v += slab[((size_t)(stage - 1) * nblocks + (b + 37) % nblocks) * 256 + t];
  // optional weight-like traffic: extra_bytes per workgroup streamed from a read-only pad
  for (int i = t * 4; i < extra_bytes / 4; i += 1024) {
    const float4 w = *reinterpret_cast<const float4*>(pad + ((size_t)b * 0 + i));
    v += 1e-30f * (w.x + w.y + w.z + w.w);
  }
  slab[((size_t)stage * nblocks + b) * 256 + t] = v;
  *acc = v;
}

__global__ __launch_bounds__(256) void k_stage(float* slab, int stage, int nblocks, float* out, int extra_bytes, const float* pad) {
  float acc;
  stage_work(slab, stage, nblocks, &acc, extra_bytes, pad);
  if (stage == 7) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_empty() {}

// sense-free monotonic-counter barrier: the counter only grows, generation g is complete when ctr >= g * nblocks
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();   // release (agent scope: L2 write-back of this XCD's dirty lines)
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();   // acquire (invalidate)
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_persistent(float* slab, int nblocks, float* out, unsigned* ctr, unsigned base, int extra_bytes,
                                                    const float* pad, int rounds) {
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    for (int stage = 0; stage < 8; ++stage) {
      stage_work(slab, stage, nblocks, &acc, extra_bytes, pad);
      grid_barrier(ctr, base + (unsigned)(r * 8 + stage + 1) * nblocks);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}


// ---- stage cost vs traffic: every stage streams `rd` bytes in and `wr` bytes out in total (spread over the workgroups)
__global__ __launch_bounds__(256) void k_traffic(const float4* __restrict__ src, float4* __restrict__ dst, size_t rd16, size_t wr16, float* out) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (size_t i = tid; i < rd16; i += nth) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  for (size_t i = tid; i < wr16; i += nth) dst[i] = make_float4(acc, acc, acc, acc);
  if (acc == 12345.f) out[0] = acc;
}
int traffic_main(hipStream_t st) {
  float4 *src, *dst; float* out;
  CHK(hipMalloc(&src, 64 << 20)); CHK(hipMalloc(&dst, 64 << 20)); CHK(hipMalloc(&out, 64));
  CHK(hipMemset(src, 0, 64 << 20));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const size_t MB = 1 << 20;
  const size_t cases[][2] = {{0, 0}, {MB, 0}, {4 * MB, 0}, {16 * MB, 0}, {0, MB}, {0, 4 * MB}, {0, 16 * MB}, {4 * MB, 4 * MB}, {2 * MB, 6 * MB}};
  for (auto& c : cases) {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < 8; ++s) {   // alternate direction so that a stage reads what the previous one wrote (like dW -> fwd)
      float4* a = (s & 1) ? dst : src; float4* b = (s & 1) ? src : dst;
      hipLaunchKernelGGL(k_traffic, dim3(256), dim3(256), 0, st, a, b, c[0] / 16, c[1] / 16, out);
    }
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipStreamSynchronize(st));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("traffic: read %5.1f MB  write %5.1f MB per stage (256 wgs): %.2f us per stage\n", c[0] / 1048576.0, c[1] / 1048576.0, 1e3 * ms / 200 / 8);
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  }
  return 0;
}

int main() {
  const int nblocks_list[3] = {128, 256, 512};
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *slab, *out, *pad;
  unsigned* ctr;
  CHK(hipMalloc(&slab, (size_t)8 * 512 * 256 * 4));
  CHK(hipMalloc(&out, (size_t)512 * 256 * 4));
  CHK(hipMalloc(&pad, 1 << 20));
  CHK(hipMalloc(&ctr, 4));
  CHK(hipMemset(pad, 0, 1 << 20));
  CHK(hipMemset(ctr, 0, 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int extra : {0, 65536}) {
    for (int nb : nblocks_list) {
      // ---- (a) 8 dependent kernels per graph replay
      hipGraph_t g; hipGraphExec_t ge;
      CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < 8; ++s) hipLaunchKernelGGL(k_stage, dim3(nb), dim3(256), 0, st, slab, s, nb, out, extra, pad);
      CHK(hipStreamEndCapture(st, &g));
      CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 50; ++i) CHK(hipGraphLaunch(ge, st));
      CHK(hipStreamSynchronize(st));
      const int reps = 500;
      CHK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CHK(hipGraphLaunch(ge, st));
      CHK(hipEventRecord(e1, st));
      CHK(hipStreamSynchronize(st));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<float> h((size_t)nb * 256);
      CHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
      const double us_graph = 1e3 * ms / reps / 8;
      const float want = 36.0f;   // 1+2+...+8
      bool ok_a = true;
      for (float v : h) ok_a = ok_a && v == want;
      CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
      // ---- (b) persistent kernel, 8 barriers per round
      unsigned base = 0;
      CHK(hipMemsetAsync(ctr, 0, 4, st));
      const int rounds = 200;
      hipLaunchKernelGGL(k_persistent, dim3(nb), dim3(256), 0, st, slab, nb, out, ctr, base, extra, pad, 20);
      CHK(hipStreamSynchronize(st));
      base = 20u * 8u * nb;
      CHK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_persistent, dim3(nb), dim3(256), 0, st, slab, nb, out, ctr, base, extra, pad, rounds);
      CHK(hipEventRecord(e1, st));
      CHK(hipStreamSynchronize(st));
      CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
      bool ok_b = true;
      for (float v : h) ok_b = ok_b && v == want;
      const double us_bar = 1e3 * ms / rounds / 8;
      printf("extra=%6d B/wg  nblocks=%3d : graph stage %.2f us (%s)   persistent stage+barrier %.2f us (%s)\n", extra, nb, us_graph,
             ok_a ? "ok" : "WRONG", us_bar, ok_b ? "ok" : "WRONG");
    }
  }
  // ---- empty kernels in a graph: the floor of a dependent launch
  {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < 8; ++s) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st);
    CHK(hipStreamEndCapture(st, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 50; ++i) CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < 500; ++i) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipStreamSynchronize(st));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel in a graph: %.2f us per launch\n", 1e3 * ms / 500 / 8);
  }
  return traffic_main(st);
}
