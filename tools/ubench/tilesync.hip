// Microbenchmark (gfx950): what does a TILE-LOCAL hand-off between the workgroups of one row tile cost inside one launch?
// The SAC step's forward / backward launches are row-local: the 16 workgroups (4 tasks x 4 column slices) that serve one 16-row tile
// only ever exchange data with each other, and they all sit on one XCD (workgroups are dealt to the 8 XCDs round-robin in linear
// order, x fastest; tile = blockIdx.x).  If such a group can hand data over through its XCD's L2 with a flag instead of a kernel
// boundary, three launches (F1 F2 B1) become one.  Measured here: R rounds of { every workgroup writes 1 KiB, signals, waits for all
// MEMBERS workgroups of its tile, reads a neighbour's 1 KiB } with (a) agent-scope release / acquire atomics, (b) relaxed atomics +
// __threadfence, against (c) the same exchange as dependent kernels of a captured graph.   hipcc --offload-arch=gfx950 -O3 -o tilesync tilesync.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TILES = 16, MEMBERS = 16, NWG = TILES * MEMBERS, ROUNDS = 64;

template <int MODE>
__global__ __launch_bounds__(256) void k_tilesync(float* slab, unsigned* flags, unsigned base, float* out, unsigned long long* clk, int* err) {
  const int tile = blockIdx.x, member = blockIdx.y, wg = member * TILES + tile, t = threadIdx.x;
  unsigned* flag = flags + tile * 32;   // one 128-byte line per tile
  float acc = 0.f;
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < ROUNDS; ++r) {
    float* mine = slab + ((size_t)(r & 1) * NWG + wg) * 256 + t;
    if (MODE == 2) __hip_atomic_store(mine, acc + (float)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) __hip_atomic_store(mine, acc + (float)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *mine = acc + (float)(r + 1);
    if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores are acknowledged by L2 before anyone is told
    __syncthreads();
    if (t == 0) {
      const unsigned target = base + (unsigned)(r + 1) * MEMBERS;
      if (MODE == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
      } else if (MODE >= 4) {
        // plain data stores (acknowledged by this XCD's L2 before the barrier), flag = relaxed atomic (agent scope in 4 / 5, workgroup
        // scope = executed in this XCD's L2 in 6), no fence; the consumer drops its vector L1 (buffer_inv) and reads with plain loads
        if (MODE == 6) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        if (MODE == 8) {   // PIPELINED poll: three looks in flight, each consumed as it lands (s_waitcnt vmcnt(2)) and re-issued at once
          unsigned a, b, c, n;
          asm volatile(
              "global_load_dword %[a], %[p], off sc1\n\t"
              "s_sleep 1\n\t"
              "global_load_dword %[b], %[p], off sc1\n\t"
              "s_sleep 1\n\t"
              "global_load_dword %[c], %[p], off sc1\n\t"
              "s_mov_b32 %[n], 0\n"
              ".Lpoll_%=:\n\t"
              "s_waitcnt vmcnt(2)\n\t"
              "v_cmp_le_u32_e32 vcc, %[t], %[a]\n\t"
              "s_cbranch_vccnz .Ldone_%=\n\t"
              "global_load_dword %[a], %[p], off sc1\n\t"
              "s_waitcnt vmcnt(2)\n\t"
              "v_cmp_le_u32_e32 vcc, %[t], %[b]\n\t"
              "s_cbranch_vccnz .Ldone_%=\n\t"
              "global_load_dword %[b], %[p], off sc1\n\t"
              "s_waitcnt vmcnt(2)\n\t"
              "v_cmp_le_u32_e32 vcc, %[t], %[c]\n\t"
              "s_cbranch_vccnz .Ldone_%=\n\t"
              "global_load_dword %[c], %[p], off sc1\n\t"
              "s_add_u32 %[n], %[n], 1\n\t"
              "s_cmp_lt_u32 %[n], 0x100000\n\t"
              "s_cbranch_scc1 .Lpoll_%=\n\t"
              "s_mov_b32 %[n], -1\n"
              ".Ldone_%=:\n\t"
              "s_waitcnt vmcnt(0)"
              : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [n] "=&s"(n)
              : [p] "v"(flag), [t] "v"(target)
              : "vcc", "scc", "memory");
          if (n == 0xffffffffu) *err = 1;
        } else
        if (MODE == 7) {   // the poll on the SCALAR memory path (glc: past the scalar cache, from the L2): no vector-memory counter involved
          unsigned v;
          do {
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(flag) : "memory");
            if (++spins > (1 << 22)) { *err = 1; break; }
          } while (v < target);
        } else
        while ((MODE == 6 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                          : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { *err = 1; break; }
        }
      } else if (MODE == 2 || MODE == 3) {
        // no fences: the data went out as relaxed atomic stores of the same scope (complete before the barrier's s_waitcnt), the flag is a
        // relaxed atomic of that scope.  MODE 2 = workgroup scope (sc0: through this XCD's L2 only — valid ONLY because all members of a
        // tile sit on one XCD), MODE 3 = agent scope (sc1: coherent across XCDs)
        if (MODE == 2) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((MODE == 2 ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                          : __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { *err = 1; break; }
        }
      } else {
        __threadfence();
        atomicAdd(flag, 1u);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *err = 1; break; }
        }
        __threadfence();
      }
    }
    __syncthreads();
    const int nb = ((member + 5) % MEMBERS) * TILES + tile;
    // the neighbour's line was written by another CU: read it past this CU's vector L1 (agent-scope load)
    const float* theirs = slab + ((size_t)(r & 1) * NWG + nb) * 256 + t;
    if (MODE == 4) asm volatile("buffer_inv sc1" ::: "memory");
    if (MODE >= 5) asm volatile("buffer_inv sc0" ::: "memory");
    if (MODE >= 4) acc = *(volatile const float*)theirs;
    else
    acc = MODE == 2 ? __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                    : __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();   // nobody overwrites parity (r & 1) before the round after next: two barriers away
  }
  const unsigned long long t1 = wall_clock64();
  out[wg * 256 + t] = acc;
  if (t == 0) clk[wg] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_stage(float* slab, int r, float* out) {
  const int tile = blockIdx.x, member = blockIdx.y, wg = member * TILES + tile, t = threadIdx.x;
  float acc = (float)r;
  if (r > 0) acc += slab[((size_t)((r - 1) & 1) * NWG + ((member + 5) % MEMBERS) * TILES + tile) * 256 + t];
  slab[((size_t)(r & 1) * NWG + wg) * 256 + t] = acc;
  if (r == 7) out[wg * 256 + t] = acc;
}

int main() {
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *slab, *out; unsigned* flags; unsigned long long* clk; int* err;
  CHK(hipMalloc(&slab, (size_t)2 * NWG * 256 * 4)); CHK(hipMalloc(&out, (size_t)NWG * 256 * 4));
  CHK(hipMalloc(&flags, TILES * 128)); CHK(hipMalloc(&clk, NWG * 8)); CHK(hipMalloc(&err, 4));
  CHK(hipMemset(flags, 0, TILES * 128)); CHK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  unsigned base = 0;
  // mode 2 (workgroup-scope atomics on the DATA, no invalidate) is not run by default: its first launch can spin on a stale L1 line
  // until the bound trips (4 s)
  for (int mode = 3; mode < 9; ++mode) {
    CHK(hipMemset(err, 0, 4));
    for (int rep = 0; rep < 3; ++rep) {
      CHK(hipEventRecord(e0, st));
      if (mode == 0) hipLaunchKernelGGL(k_tilesync<0>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 1) hipLaunchKernelGGL(k_tilesync<1>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 2) hipLaunchKernelGGL(k_tilesync<2>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 3) hipLaunchKernelGGL(k_tilesync<3>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 4) hipLaunchKernelGGL(k_tilesync<4>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 5) hipLaunchKernelGGL(k_tilesync<5>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 6) hipLaunchKernelGGL(k_tilesync<6>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else if (mode == 7) hipLaunchKernelGGL(k_tilesync<7>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      else hipLaunchKernelGGL(k_tilesync<8>, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, flags, base, out, clk, err);
      CHK(hipEventRecord(e1, st));
      CHK(hipStreamSynchronize(st));
      base += ROUNDS * MEMBERS;
      float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[NWG]; int herr = 0; float ho[4];
      CHK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost));
      unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
      printf("tile-local hand-off, %s: %.2f us per round (kernel %.1f us / %d rounds; slowest workgroup %.2f us per round)  err=%d out=%g\n",
             mode == 0 ? "release/acquire atomics" : mode == 1 ? "relaxed atomics + __threadfence" : mode == 2 ? "WORKGROUP-scope relaxed atomics, no fence (same-XCD L2)" : mode == 3 ? "AGENT-scope relaxed atomics, no fence" : mode == 4 ? "plain data + agent flag + buffer_inv sc1" : mode == 5 ? "plain data + agent flag + buffer_inv sc0" : mode == 6 ? "plain data + WORKGROUP flag + buffer_inv sc0" : mode == 7 ? "plain data + agent flag polled by s_load glc + buffer_inv sc0" : "plain data + agent flag, PIPELINED poll (3 looks in flight) + buffer_inv sc0", 1e3 * ms / ROUNDS, 1e3 * ms, ROUNDS, mx * 0.01 / ROUNDS, herr, ho[0]);
    }
  }
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int s = 0; s < 8; ++s) hipLaunchKernelGGL(k_stage, dim3(TILES, MEMBERS), dim3(256), 0, st, slab, s, out);
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipStreamSynchronize(st));
  CHK(hipEventRecord(e0, st));
  for (int i = 0; i < 200; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipEventRecord(e1, st));
  CHK(hipStreamSynchronize(st));
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("same exchange as dependent kernels of a graph: %.2f us per stage\n", 1e3 * ms / 200 / 8);
  return 0;
}
