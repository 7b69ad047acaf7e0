// host_common.h — host-side objects behind the opaque handles of include/ilsx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ilsx.h"
#include "kernels.h"

void ilsx_set_err(const char* fmt, ...);

#define ILSX_FAIL(code, ...)   \
  do {                         \
    ilsx_set_err(__VA_ARGS__); \
    return (code);             \
  } while (0)

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      ilsx_set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);    \
      return ILSX_ERR_HIP;                                                                        \
    }                                                                                             \
  } while (0)

#define ILSX_TRY(expr)        \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != ILSX_OK) return rc_; \
  } while (0)

struct ilsx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<int> phase_slots;   // constant-table slots taken by this ctx's agents (PhaseConst): released with the ctx, whose agents die with it
  uint64_t seed = 0;
  uint32_t next_rng_stream = 1;
  unsigned long long act_calls = 0, ppo_act_calls = 0;   // Philox counters of ilsx_policy_act / ilsx_ppo_policy_act (one draw per call)
  std::vector<void*> allocs;
  // small reusable staging buffer for host->device row uploads
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // optional per-kernel HIP-event timing (include/ilsx.h "kernel timing")
  // row-range slabs of split weight-gradient launches (large batches): ONE region per gradient arena (keyed by the table's g_lo).  The kernels
  // write only the live words of a slab and rely on the padding words staying zero — true as long as no table with another layout ever
  // writes the same region (PPO's value and policy tables used to share one: ADVICE r4)
  struct DwScratch { const float* key; size_t span; void* p; size_t bytes; };   // key + span name the table's gradient range
  std::vector<DwScratch> dw_scratch;
  int rt_single = 1, rt_grouped = 1;   // 16-row tiles per workgroup in the column-split kernels (ILSX_RT / ILSX_RT_GROUPED)
  int xcd_shift = 0;  // ILSX_XCD_SHIFT: confine the split-MLP / dW kernels to every 2^k-th workgroup slot (3 = one XCD)
  unsigned long long* dbg_stamps = nullptr;  // device trace buffer for ILSX_STAMP (debug): [launch][ILSX_TRACE_MAXWG][ILSX_TRACE_SLOTS]
  int dbg_launches = 0, dbg_max_launches = 0;
  unsigned long long* dbg_next() {   // slab of the next instrumented launch, or nullptr when tracing is off / full
    if (!dbg_stamps || dbg_launches >= dbg_max_launches) return nullptr;
    return dbg_stamps + (size_t)(dbg_launches++) * ILSX_TRACE_MAXWG * ILSX_TRACE_SLOTS;
  }
  void* comm = nullptr;   // ncclComm_t of a split run (ilsx_comm.hip); collectives go on `stream`
  int comm_n = 0, comm_rank = 0;
  bool prof_on = false;
  struct ProfRec { int kid; hipEvent_t a, b; };
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_free;
  double prof_ms[ILSX_K_COUNT] = {0};
  uint64_t prof_n[ILSX_K_COUNT] = {0};
  const char* prof_name[ILSX_K_COUNT] = {nullptr};   // source spelling of the kernel last launched under each slot
};

// RAII bracket around ONE kernel launch.  When profiling, the launch goes through hipExtLaunchKernelGGL, which stamps the two
// events with the dispatch's own begin / end timestamps (what rocprofv3's kernel trace reports) instead of bracketing it
// with two extra marker packets (which adds ~3 us to a ~10 us kernel).
struct ProfScope {
  ilsx_ctx* c; int kid; hipEvent_t a = nullptr, b = nullptr; bool launched = false;
  ProfScope(ilsx_ctx* ctx, int k);
  ~ProfScope();
};
#define ILSX_LAUNCH(ps, kernel, grid, block, lds, stream, ...)                                            \
  do {                                                                                                    \
    if ((ps).a) {                                                                                         \
      (ps).c->prof_name[(ps).kid] = #kernel; (ps).launched = true;                                        \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (ps).a, (ps).b, 0, __VA_ARGS__);            \
    } else {                                                                                              \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                  \
    }                                                                                                     \
  } while (0)

int comm_allreduce_sum(ilsx_ctx* c, float* buf, size_t n);   // in place, on c->stream
int ctx_alloc(ilsx_ctx* c, size_t bytes, void** out, bool zero = true);
// One allocation carved into many buffers.  Every buffer an agent touches in a step then sits in ONE virtually contiguous
// range (a handful of 2 MiB translation fragments) instead of ~60 separate hipMalloc's: a kernel that reads a dozen of them
// as its first operands otherwise pays a dozen page-table walks before its first useful byte (measured: the first operands of
// the backward launch landed 2.5 us after workgroup start, the weights — one allocation — after 0.55 us).
struct Slab {
  std::vector<std::pair<void**, size_t>> items;
  size_t bytes = 0;
  template <class T> void add(T** p, size_t count) {
    bytes = (bytes + 255) & ~(size_t)255;
    items.push_back({(void**)p, bytes});
    bytes += count * sizeof(T);
  }
  int commit(ilsx_ctx* c, void** base) {   // zero-filled
    int rc = ctx_alloc(c, bytes ? bytes : 256, base, true);
    if (rc != ILSX_OK) return rc;
    for (auto& it : items) *it.first = (char*)*base + it.second;
    return ILSX_OK;
  }
};
int ctx_free(ilsx_ctx* c, void* p);
int ctx_stage(ilsx_ctx* c, size_t bytes, void** out);

// Layout of one network: flat ABI order <-> internal (first-layer rows padded to KP, heads fused).
struct NetLayout {
  ilsx_mlp_cfg cfg;
  int KP = 0, NO = 0;
  int off_W[ILSX_MAX_HID], off_Wb[ILSX_MAX_HID], off_b[ILSX_MAX_HID], ld[ILSX_MAX_HID];
  int off_Wh = 0, off_bh = 0;
  size_t n_int = 0;   // internal floats (multiple of 4)
  size_t n_flat = 0;  // ABI floats
  int hl[ILSX_MAX_HID] = {0, 0, 0};   // logical width of hidden layer l (<= cfg.hidden: ilsx_mlp_cfg::hidden_sizes)
  int out_of(int l) const { return hl[l] > 0 ? hl[l] : cfg.hidden; }
  int in_of(int l) const { return l == 0 ? cfg.in_dim : out_of(l - 1); }   // LOGICAL inputs of layer l (flat ABI sizes)
};
int net_layout_build(const ilsx_mlp_cfg& cfg, NetLayout* L);
void net_flat_to_internal(const NetLayout& L, const float* flat, float* internal);
void net_internal_to_flat(const NetLayout& L, const float* internal, float* flat);
NetView net_view(const NetLayout& L, float* base);

struct ilsx_net {
  ilsx_ctx* ctx = nullptr;
  NetLayout lay;
  float* base = nullptr;  // device, internal layout
  bool owns = true;
  // lazily sized workspace for the standalone forward / act entry points
  float* ws_out = nullptr;
  int ws_rows = 0;
  // MlpGaussianNoisePolicy (policies.py:130-188): single head, tanh output, clipped Gaussian exploration noise
  bool noise_policy = false;
  float noise = 0.f, noise_clip = 0.f, max_act = 1.f;
  bool out_linear = false;   // output activation: identity (Mlp's default) instead of tanh (ilsx_net_set_output_linear)
};

int net_upload_flat(ilsx_ctx* ctx, const NetLayout& L, float* dev_base, const float* src, size_t n, int src_is_device);
int net_download_flat(ilsx_ctx* ctx, const NetLayout& L, const float* dev_base, float* dst, size_t n, int dst_is_device);

int launch_fwd(ilsx_ctx* ctx, const FwdArgs& A, int H, int act, int KP, int cs = 1);
int launch_bwd_dx(ilsx_ctx* ctx, const BwdArgs& A, int H, int act, int cs = 1);
int launch_policy_finish(ilsx_ctx* ctx, const PolicyFinishArgs& P);
// merged phase kernels of the single-run SAC step (kernels.h); phase_fits: every workgroup of such a launch is resident at once
bool phase_fits(ilsx_ctx* ctx, int rows, int H, int cs, int ntasks);
// The phase kernels' descriptor blocks in CONSTANT memory (kernels.h g_phase_a_tab / g_phase_c_tab, `CT` instances): one slot per agent, the
// host's copy of what the slot holds.  slot < 0: none (table full, or ILSX_PHASE_CT=0): the block travels in the kernel arguments as before.
struct PhaseConst {
  int slot = -1, device = 0;
  bool tried = false, valid_a = false, valid_c = false;
  unsigned long long key_a = 0, key_c = 0;   // what the held blocks were built from (the owner's state key: the blocks are a function of it; their
                                             // bytes cannot be compared — padding of by-value sub-records is whatever the stack held)
  PhaseAArgs a; PhaseCArgs c;
};
int phase_const_alloc(int device);            // a free slot of that device's tables, or -1
void phase_const_prepare(ilsx_ctx* ctx, PhaseConst* ct);   // first use: take a slot (once; none free = stays without)
void phase_const_free(int device, int slot);
int launch_phase_a(ilsx_ctx* ctx, const PhaseAArgs& P, int H, int act, int KPmax, int cs, PhaseConst* ct = nullptr, unsigned long long key = 0, bool upload_only = false);
int launch_phase_c(ilsx_ctx* ctx, const PhaseCArgs& P, int H, int act, int KPmax, int cs, PhaseConst* ct = nullptr, unsigned long long key = 0, bool upload_only = false);
int device_cus(ilsx_ctx* ctx);   // compute units of the context's device
// constant-memory descriptor tables of grouped launches (kernels.h g_fwd_tab / g_bwd_tab): slots per device, first fit; -1 = none free
int grp_const_alloc(int device, bool fwd, int n, int* base);
void grp_const_free(int device, bool fwd, int base);
int grp_const_upload(ilsx_ctx* ctx, bool fwd, int base, const void* host_records, size_t count);
struct ilsx_sac;
int sac_staged_batch(ilsx_sac* s, int B, float** obs, float** act, float** rew, float** done, float** nobs);   // ilsx_sac.hip
int sac_step_staged(ilsx_sac* s, ilsx_sac_stats* stats);
int sac_dims(const ilsx_sac* s, int* o, int* a);
int sac_window_begin(ilsx_sac* s, int B);   // steps on staged batches with the deferred tail + phase kernels (ilsx_sac.hip)
int sac_window_step(ilsx_sac* s);
int sac_window_end(ilsx_sac* s, bool teardown = false);   // teardown: closing on an error path (no roll-back vote, no retry); ILSX_RETRY_WINDOW: the window's phase kernels reported a broken hand-off — roll back and re-run
// internal status (never crosses the C ABI): a window of steps that ran on the merged phase kernels has to be rolled back
// (sac_snapshot_restore) and run again; the agent has switched itself to one launch per stage
#define ILSX_RETRY_WINDOW (-1000)
int sac_snapshot_take(ilsx_sac* s);         // checkpoint of scalars | parameters | gradients | Adam moments (one device-to-device copy)
int sac_snapshot_restore(ilsx_sac* s);
bool sac_window_may_use_phase(ilsx_sac* s, int B);
// column-split factor the 2-hidden-layer fast path uses for width H (1 = generic kernels)
int mlp2_split_factor(int n_hidden, int H);
int launch_bwd_dw(ilsx_ctx* ctx, const DwArgs& table, int rows, const AdamFuse* fuse = nullptr);
// appends one matrix to a dW table (computes its tile range)
int dw_table_add(DwArgs* T, const float* A, int lda, int NA, const float* Bm, int ldb, int NB, float* dW, float* dWb, int ldw,
                 float* db, int mode, int rows = 0, int bias_rows = 0);
int launch_adam(ilsx_ctx* ctx, const AdamArgs& A);
// appends the dW/db matrices of one network to a dW table
int build_dw_jobs(const NetLayout& L, float* gbase, const float* xsave, float* const* hsave,
                  float* const* dsave, const float* dhead, DwArgs* table);

// Replay ring: HBM-resident transition records + host mirror of the reference's cursors.
struct ilsx_replay {
  ilsx_ctx* ctx = nullptr;
  int64_t cap = 0;
  int o = 0, a = 0, rec = 0;  // rec = floats per record (multiple of 32)
  float* data = nullptr;      // [cap][rec]
  DevReplayState* dstate = nullptr;
  bool dstate_stale = false;   // the host cursors moved (fused rollout) since {size, top} were last written to `dstate`: replay_flush_state
  uint64_t seed = 0;
  uint32_t rng_stream = 0;
  unsigned long long sample_ctr = 0;  // host-side Philox step for standalone sample calls
  // reference cursors (simple_replay_buffer.py:61-68)
  int64_t top = 0, size = 0, cur_start = 0;
  std::deque<std::pair<int64_t, int64_t>> trajs;  // insertion-ordered (start,end)
  std::vector<uint8_t> start_flag;                // slot is a key of `trajs`
};
// bring the device copy of {size, top} up to date (enqueued on the ring's stream) before a kernel that draws from the ring reads it
int replay_flush_state(ilsx_replay* rb);
// n rows were written into the ring at [top, top+n) by a device kernel: advance the cursors.
int replay_advance_device_rows(ilsx_replay* rb, int n);
// n_paths whole episodes staged at stage[env][t][rec] enter the ring contiguously, in the order given (path mode of the fused rollout)
int replay_insert_paths(ilsx_replay* rb, const float* stage, int stage_len, const int* envs, const int* lens, const uint8_t* last_terminal,
                        int n_paths);
int replay_launch_sample(ilsx_replay* rb, int B, const int64_t* idx, const DevScalars* scal,
                         unsigned long long step_host, float* obs, float* act, float* rew, float* done,
                         float* nobs, int64_t* idx_out);
