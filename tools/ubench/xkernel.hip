// Microbenchmark (gfx950): can a CONSUMER kernel that is already running (launched on a second stream, no dependency) pick up what a
// PRODUCER kernel wrote on other XCDs, through a counter, without the kernel boundary between them — and what does that cost against the
// ordinary dependent launch?  The question behind "each weight-gradient launch as a last stage of the phase kernel before it" (DESIGN §9
// (2)): the step's four launch boundaries cost ~2.4 us each plus the next kernel's start-up.
//   producer P: 256 workgroups x 256 threads, each busy for ~D us (a dependent-FMA loop standing in for a phase kernel), then writes its
//               1 KiB slice of X (value = iteration), waits for the stores' acknowledgement, signals one agent-scope counter
//   consumer Q: 576 workgroups x 512 threads; waits for the counter to reach 256 x iteration, drops its vector L1, reads 512 floats that
//               OTHER workgroups (other XCDs) wrote, checks them against the iteration number, does a little arithmetic, writes a result
//   join    R : one tiny kernel after both (stands for the next phase kernel)
// forms:  seq      P -> Q -> R on one stream (Q without the wait): today's chain
//         conc/S   P on stream a, Q on stream b with NO dependency on P, R after both; producer stores of kind S:
//                  0 plain            (data sits dirty in the producer XCD's L2: expected to FAIL across XCDs)
//                  1 nontemporal      (__builtin_nontemporal_store)
//                  2 sc0 sc1          (system-scope write-through store)
//                  3 plain + buffer_wbl2 sc1 before the signal (write the XCD's L2 back)
//         consumer loads are plain (its XCD's L2 can only hold lines of X that were (re)fetched after its own kernel-start invalidate and that
//         the producers of the SAME XCD wrote — fresh either way) — the mismatch count says whether that reasoning holds; variant L = 1 reads
//         with agent-scope (sc1) loads instead.
//   hipcc --offload-arch=gfx950 -O3 -o xkernel xkernel.hip && ./xkernel
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NP = 256, NQ = 576, SLICE = 256;   // floats per producer workgroup

template <int S>
__global__ __launch_bounds__(256) void k_prod(float* X, unsigned* counter, int iter, int busy, float* sink) {
  const int wg = blockIdx.x, t = threadIdx.x;
  float a = (float)t * 1e-3f;
  for (int i = 0; i < busy; ++i) a = fmaf(a, 1.0000001f, 1e-7f);   // ~4 cycles per trip
  float* p = X + (size_t)wg * SLICE + t;
  const float v = (float)iter + (a > 1e30f ? 1.0f : 0.0f);
  if (S == 1) __builtin_nontemporal_store(v, p);
  else if (S == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (S == 3) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a > 1e30f) sink[0] = a;
}

template <int WAIT, int L>
__global__ __launch_bounds__(512) void k_cons(const float* X, unsigned* counter, int iter, float* out, int* mism, int* err, unsigned long long* clk) {
  const int wg = blockIdx.x, t = threadIdx.x;
  if (WAIT) {
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(NP * iter)) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18)) { *err = 1; break; }
      }
      if (wg == 0) clk[0] = wall_clock64();
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");
  }
  // 512 floats from producers spread over the whole grid (every XCD)
  const int src = (wg * 37 + t * 5) % NP;
  const float* q = X + (size_t)src * SLICE + ((t * 7 + wg) & (SLICE - 1));
  float v;
  if (L == 1) v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else v = *q;
  if (v != (float)iter) atomicAdd(mism, 1);
  float a = v;
  for (int i = 0; i < 200; ++i) a = fmaf(a, 0.999f, 0.5f);
  out[(size_t)wg * 512 + t] = a;
  if (wg == NQ - 1 && t == 0) clk[1] = wall_clock64();
}

__global__ void k_join(float* out) { if (threadIdx.x == 0 && out[0] < -1e30f) out[1] = 0.f; }

template <int S, int L>
static int run_conc(const char* name, float* X, unsigned* counter, float* out, float* sink, int* mism, int* err, unsigned long long* clk, int iters, int busy,
                    bool consumer_first) {
  hipStream_t sa, sb;
  CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t ea, eb;
  CHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  CHK(hipMemset(counter, 0, 4)); CHK(hipMemset(mism, 0, 4)); CHK(hipMemset(err, 0, 4)); CHK(hipMemset(X, 0, NP * SLICE * 4));
  CHK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= iters; ++i) {
      const int it = rep * iters + i;
      // both kernels of iteration `it` start after the join of the previous one
      CHK(hipEventRecord(ea, sa));
      CHK(hipStreamWaitEvent(sb, ea, 0));
      if (consumer_first) hipLaunchKernelGGL((k_cons<1, L>), dim3(NQ), dim3(512), 0, sb, X, counter, it, out, mism, err, clk);
      hipLaunchKernelGGL((k_prod<S>), dim3(NP), dim3(256), 0, sa, X, counter, it, busy, sink);
      if (!consumer_first) hipLaunchKernelGGL((k_cons<1, L>), dim3(NQ), dim3(512), 0, sb, X, counter, it, out, mism, err, clk);
      CHK(hipEventRecord(eb, sb));
      CHK(hipStreamWaitEvent(sa, eb, 0));
      hipLaunchKernelGGL(k_join, dim3(1), dim3(64), 0, sa, out);
    }
    CHK(hipStreamSynchronize(sa)); CHK(hipStreamSynchronize(sb));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    if (us < best) best = us;
  }
  int hm = 0, he = 0;
  CHK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
  printf("%-44s %8.2f us per iteration   mismatches %d   wait timed out %d\n", name, best, hm, he);
  // the same fork / join captured ONCE (20 iterations per graph; the consumer's target comes from a device-side iteration counter
  // is not needed here: every replay re-arms the counter with a memset node) and replayed
  {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipMemset(mism, 0, 4)); CHK(hipMemset(err, 0, 4));
    CHK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
    CHK(hipMemsetAsync(counter, 0, 4, sa));
    for (int it = 1; it <= 20; ++it) {
      CHK(hipEventRecord(ea, sa));
      CHK(hipStreamWaitEvent(sb, ea, 0));
      if (consumer_first) hipLaunchKernelGGL((k_cons<1, L>), dim3(NQ), dim3(512), 0, sb, X, counter, it, out, mism, err, clk);
      hipLaunchKernelGGL((k_prod<S>), dim3(NP), dim3(256), 0, sa, X, counter, it, busy, sink);
      if (!consumer_first) hipLaunchKernelGGL((k_cons<1, L>), dim3(NQ), dim3(512), 0, sb, X, counter, it, out, mism, err, clk);
      CHK(hipEventRecord(eb, sb));
      CHK(hipStreamWaitEvent(sa, eb, 0));
      hipLaunchKernelGGL(k_join, dim3(1), dim3(64), 0, sa, out);
    }
    CHK(hipStreamEndCapture(sa, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double bg = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters / 20; ++i) CHK(hipGraphLaunch(ge, sa));
      CHK(hipStreamSynchronize(sa));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (iters / 20 * 20);
      if (us < bg) bg = us;
    }
    CHK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost));
    printf("%-44s %8.2f us per iteration   mismatches %d   wait timed out %d\n", "    ... as a graph with parallel branches", bg, hm, he);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  hipStreamDestroy(sa); hipStreamDestroy(sb);
  return 0;
}

int main(int argc, char** argv) {
  const int iters = 300, busy = argc > 1 ? atoi(argv[1]) : 12000;   // 12000 trips x 4 cycles ~ 20 us at 2.4 GHz
  float *X, *out, *sink; unsigned* counter; int *mism, *err; unsigned long long* clk;
  CHK(hipMalloc(&X, NP * SLICE * 4)); CHK(hipMalloc(&out, (size_t)NQ * 512 * 4)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&counter, 128));
  CHK(hipMalloc(&mism, 4)); CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&clk, 64));
  // ---- sequential chain on one stream, direct launches and as a graph
  {
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHK(hipMemset(mism, 0, 4)); CHK(hipMemset(counter, 0, 4));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= iters; ++i) {
        const int it = rep * iters + i;
        hipLaunchKernelGGL((k_prod<0>), dim3(NP), dim3(256), 0, s, X, counter, it, busy, sink);
        hipLaunchKernelGGL((k_cons<0, 0>), dim3(NQ), dim3(512), 0, s, X, counter, it, out, mism, err, clk);
        hipLaunchKernelGGL(k_join, dim3(1), dim3(64), 0, s, out);
      }
      CHK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      if (us < best) best = us;
    }
    int hm = 0;
    CHK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("%-44s %8.2f us per iteration   mismatches %d\n", "seq: P -> Q -> R, direct launches", best, hm);
    // the same as a graph of 3 x 20 nodes
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipMemset(mism, 0, 4));
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 1; i <= 20; ++i) {
      hipLaunchKernelGGL((k_prod<0>), dim3(NP), dim3(256), 0, s, X, counter, 1, busy, sink);
      hipLaunchKernelGGL((k_cons<0, 0>), dim3(NQ), dim3(512), 0, s, X, counter, 1, out, mism, err, clk);
      hipLaunchKernelGGL(k_join, dim3(1), dim3(64), 0, s, out);
    }
    CHK(hipStreamEndCapture(s, &g));
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters / 20; ++i) CHK(hipGraphLaunch(ge, s));
      CHK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (iters / 20 * 20);
      if (us < best) best = us;
    }
    printf("%-44s %8.2f us per iteration\n", "seq: the same chain replayed from a graph", best);
    // producer alone, for the budget
    best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= iters; ++i) hipLaunchKernelGGL((k_prod<0>), dim3(NP), dim3(256), 0, s, X, counter, i, busy, sink);
      CHK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      if (us < best) best = us;
    }
    printf("%-44s %8.2f us per iteration\n", "P alone, back to back", best);
    hipStreamDestroy(s);
  }
  // ---- concurrent consumer
  if (run_conc<0, 0>("conc: plain stores, plain loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<1, 0>("conc: nontemporal stores, plain loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<2, 0>("conc: sc0 sc1 stores, plain loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<3, 0>("conc: plain stores + wbl2 sc1, plain loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<2, 1>("conc: sc0 sc1 stores, sc1 loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<0, 1>("conc: plain stores, sc1 loads", X, counter, out, sink, mism, err, clk, iters, busy, false)) return 1;
  if (run_conc<2, 0>("conc: sc0 sc1 stores, CONSUMER launched first", X, counter, out, sink, mism, err, clk, iters, busy, true)) return 1;
  return 0;
}
