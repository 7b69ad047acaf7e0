// Where does a 3-D vec-env step spend its cycles?  Builds env3d_wave.h with E3W_PROFILE (stage timers in LDS) into a small shared
// library; tools/ubench/env3d_phases.py drives it with the Humanoid / Ant model and prints cycles per stage of one dynamics
// evaluation (20 per env step).  Not part of libilsx.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -shared -fPIC tools/ubench/env3d_phases.hip -o /tmp/libe3p.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#define E3W_PROFILE 1
#include "../../ilswiss_amd/csrc/env3d_wave.h"

__global__ __launch_bounds__(64) void k_phases(const Spatial3Dev* mp, const double* q0, const double* v0, const float* act, int n_steps, double* out,
                                               double* qv_out) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  e3w_lds* S = (e3w_lds*)smem;
  const Spatial3Dev& m = *mp;
  const int lane = threadIdx.x, env = blockIdx.x;
  for (int i = lane; i < m.nq; i += 64) S[E3WOff::Q0 + i] = q0[env * m.nq + i];
  for (int i = lane; i < m.nv; i += 64) S[E3WOff::V0 + i] = v0[env * m.nv + i];
  if (lane < 16) S[E3WOff::TOTAL + lane] = 0.0;
  E3W_SYNC();
  if (lane == 0) S[E3WOff::TOTAL + 15] = (double)__builtin_amdgcn_s_memtime();
  E3W_SYNC();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  E3WRegs regs[1];
  e3w_regs_init(regs[0], m, lane);
  e3w_regs_pin(regs[0]);
  for (int s = 0; s < n_steps; ++s) {
    double r; bool d;
    if (m.nv == 23) e3w_task_step<23>(S, m, lane, regs, act + (size_t)env * m.n_act, r, d);
    else e3w_task_step<14>(S, m, lane, regs, act + (size_t)env * m.n_act, r, d);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane < 13) out[env * 16 + lane] = S[E3WOff::TOTAL + lane];
  if (lane == 13) out[env * 16 + 13] = (double)(t1 - t0);
  for (int i = lane; i < m.nq; i += 64) qv_out[env * 64 + i] = S[E3WOff::Q0 + i];
}

extern "C" int e3p_run(const ilsx_spatial_model* sm, int n_env, int n_steps, const double* q0, const double* v0, const float* act, double* out,
                       double* q_out, float* ms) {
  Spatial3Dev m;
  if (e3_build_model(sm, m)) return -1;
  Spatial3Dev* dm; double *dq, *dv, *dout, *dqo; float* da;
  hipMalloc(&dm, sizeof m); hipMemcpy(dm, &m, sizeof m, hipMemcpyHostToDevice);
  hipMalloc(&dq, n_env * m.nq * 8); hipMemcpy(dq, q0, n_env * m.nq * 8, hipMemcpyHostToDevice);
  hipMalloc(&dv, n_env * m.nv * 8); hipMemcpy(dv, v0, n_env * m.nv * 8, hipMemcpyHostToDevice);
  hipMalloc(&da, n_env * m.n_act * 4); hipMemcpy(da, act, n_env * m.n_act * 4, hipMemcpyHostToDevice);
  hipMalloc(&dout, n_env * 16 * 8); hipMalloc(&dqo, n_env * 64 * 8);
  const size_t lds = (E3WOff::TOTAL + 16) * 8;
  hipFuncSetAttribute((const void*)k_phases, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_phases, dim3(n_env), dim3(64), lds, 0, dm, dq, dv, da, 1, dout, dqo);   // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_phases, dim3(n_env), dim3(64), lds, 0, dm, dq, dv, da, n_steps, dout, dqo);
  hipEventRecord(e1);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  hipEventElapsedTime(ms, e0, e1);
  hipMemcpy(out, dout, n_env * 16 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(q_out, dqo, n_env * 64 * 8, hipMemcpyDeviceToHost);
  hipFree(dm); hipFree(dq); hipFree(dv); hipFree(da); hipFree(dout); hipFree(dqo);
  return 0;
}
