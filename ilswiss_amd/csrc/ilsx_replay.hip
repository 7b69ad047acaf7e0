// ilsx_replay.hip — HBM-resident replay ring (rlkit/data_management/simple_replay_buffer.py:17-442,
// env_replay_buffer.py:7-49).
//
// Layout: one transition = one record [obs(o) | act(a) | rew | done | next_obs(o) | absorbing(2) | pad] of
// `rec` = round_up(2o+a+4, 32) floats, so a random-row gather touches whole 128-byte lines (Hopper:
// exactly one line per transition) instead of the 5-6 partial lines an SoA layout would cost.
// HBM-bound: algorithmic bytes per sampled row = 2*(2o+a+2)*4 (read + write), SURVEY.md §8d.
#include "host_common.h"

__global__ __launch_bounds__(256) void k_replay_add(float* __restrict__ data, int rec, long long cap, long long top,
                                                    const float* __restrict__ obs, const float* __restrict__ act,
                                                    const float* __restrict__ rew, const unsigned char* __restrict__ done,
                                                    const float* __restrict__ nobs, int n, int o, int a) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)n * rec) return;
  const int i = (int)(e / rec), c = (int)(e - (long long)i * rec);
  float v = 0.0f;
  if (c < o) v = obs[(size_t)i * o + c];
  else if (c < o + a) v = act[(size_t)i * a + (c - o)];
  else if (c == o + a) v = rew[i];
  else if (c == o + a + 1) v = done[i] ? 1.0f : 0.0f;
  else if (c < 2 * o + a + 2) v = nobs[(size_t)i * o + (c - o - a - 2)];
  long long slot = top + i;
  if (slot >= cap) slot -= cap;
  data[(size_t)slot * rec + c] = v;
}

__global__ void k_replay_set_state(DevReplayState* s, long long size, long long top) {
  s->size = size;
  s->top = top;
}

// random_batch: one thread per (row, record column); splits the record into the reference's batch keys.
__global__ __launch_bounds__(256) void k_replay_sample(const float* __restrict__ data, int rec, const DevReplayState* st,
                                                       const long long* __restrict__ idx, uint64_t seed, uint32_t stream,
                                                       const DevScalars* scal, unsigned long long step_host, int B, int o,
                                                       int a, float* __restrict__ obs, float* __restrict__ act,
                                                       float* __restrict__ rew, float* __restrict__ done,
                                                       float* __restrict__ nobs, long long* __restrict__ idx_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int used = 2 * o + a + 2;
  if (e >= B * rec) return;
  const int r = e / rec, c = e - r * rec;
  if (c >= used) return;
  long long row;
  if (idx) row = idx[r];
  else row = replay_draw(seed, scal ? scal->step : step_host, stream, (uint32_t)r, st->size);
  const float v = data[(size_t)row * rec + c];
  if (c < o) obs[(size_t)r * o + c] = v;
  else if (c < o + a) act[(size_t)r * a + (c - o)] = v;
  else if (c == o + a) rew[r] = v;
  else if (c == o + a + 1) done[r] = v;
  else nobs[(size_t)r * o + (c - o - a - 2)] = v;
  if (idx_out && c == 0) idx_out[r] = row;
}

// Hindsight relabelling gather (rlkit/data_management/relabel_replay_buffer.py:66-163): the ring's observation segment is
// [observation (d_obs) | desired_goal (dg) | achieved_goal (dg)].  Row r of the batch is record idx[r]; for the first n_relabel rows the
// desired goal (of the observation AND of the next observation, :106-113) is replaced by the ACHIEVED goal of the NEXT observation of
// record idx_rel[r] (the `future` / `final` index the host drew in the reference's RandomState order); when relabelling is on, EVERY
// row's reward is recomputed from (next achieved goal, desired goal) (:139-147) with the goal env's sparse / dense rule.  Outputs are
// what the goal-conditioned trainers consume (her/td3.py:95-99): obs_cat = observation | desired_goal, nobs_cat likewise.
// One thread per (row, output column of obs_cat).
__global__ __launch_bounds__(256) void k_her_gather(const float* __restrict__ data, int rec, const long long* __restrict__ idx,
                                                    const long long* __restrict__ idx_rel, int B, int n_relabel, int relabel_on,
                                                    int d_obs, int dg, int a, int reward_kind, float threshold,
                                                    float* __restrict__ obs_cat, float* __restrict__ act, float* __restrict__ rew,
                                                    float* __restrict__ done, float* __restrict__ nobs_cat) {
  const int W = d_obs + dg, e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * W) return;
  const int r = e / W, c = e - r * W, o = d_obs + 2 * dg;
  const float* R = data + (size_t)idx[r] * rec;
  const float* nob = R + o + a + 2;                    // next observation segment of this record
  const bool rl = relabel_on && r < n_relabel;
  const float* gsrc = rl ? data + (size_t)idx_rel[r] * rec + o + a + 2 + d_obs + dg : nullptr;   // next achieved goal of the relabel record
  if (c < d_obs) {
    obs_cat[(size_t)r * W + c] = R[c];
    nobs_cat[(size_t)r * W + c] = nob[c];
  } else {
    const int j = c - d_obs;
    obs_cat[(size_t)r * W + c] = rl ? gsrc[j] : R[d_obs + j];
    nobs_cat[(size_t)r * W + c] = rl ? gsrc[j] : nob[d_obs + j];
  }
  if (c < a) act[(size_t)r * a + c] = R[o + c];
  if (c == 0) {
    done[r] = R[o + a + 1];
    float rv = R[o + a];
    if (relabel_on) {   // compute_reward(next_achieved_goals, desired_goals): gym's GoalEnv rule — sparse -(d > threshold), dense -d
      float d2 = 0.0f;
      for (int j = 0; j < dg; ++j) {
        const float g = rl ? gsrc[j] : R[d_obs + j];
        const float df = nob[d_obs + dg + j] - g;
        d2 = fmaf(df, df, d2);
      }
      const float dist = sqrtf(d2);
      rv = reward_kind == 0 ? (dist > threshold ? -1.0f : 0.0f) : -dist;
    }
    rew[r] = rv;
  }
}

// Bandwidth form: n_batches*B whole records per launch, 16 bytes per lane.  A wavefront owns 64 consecutive output rows per trip: lane l draws
// the index of row l ONCE (the first form drew it in every lane of every 16-byte piece: 8 Philox blocks per 128-byte record, ~150 VALU
// instructions per 16 bytes moved — the kernel was issue-bound at 5.0 TB/s, not HBM-bound), the trip's 64 x REC4 pieces are then walked in
// output order (a wave's store is one contiguous 1 KiB run, its load 64 / REC4 whole records) with the row's index fetched from the lane that
// drew it.  REC4 = float4s per record: 8 (Hopper) and 16 (Walker2d, HalfCheetah) as unrolled instances — all loads of a trip in flight before
// the first store —, 0 = any width, one record at a time with the index read as a wave-uniform value.  Same draws, same output as before.
__device__ __forceinline__ long long lane_value64(long long v, int src_lane, bool wide) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src_lane, 64);
  const unsigned hi = wide ? (unsigned)__shfl((int)(unsigned)((unsigned long long)v >> 32), src_lane, 64) : 0u;   // rings of < 2^32 rows: one exchange
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <int REC4, bool NT>
__global__ __launch_bounds__(256) void k_replay_sample_many(const float4* __restrict__ data, int rec4, const DevReplayState* st,
                                                            uint64_t seed, uint32_t stream, unsigned long long step0, int B,
                                                            long long total_rows, float4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long size = st->size;
  const bool wide = size > 0xffffffffLL;   // wave-uniform
  constexpr int RPT = REC4 > 0 ? 64 : 8;   // rows per trip: 64 for the unrolled instances; wide records spread over more wavefronts (8 rows each)
  for (long long base = wave * RPT; base < total_rows; base += nwaves * RPT) {
    const long long gr = base + lane;
    long long row = 0;
    if (lane < RPT && gr < total_rows) {
      const long long batch = gr / B;
      row = replay_draw(seed, step0 + batch, stream, (uint32_t)(gr - batch * B), size);
    }
    if (REC4 > 0 && base + 64 <= total_rows) {   // (wave-uniform) a whole trip: no guards, every load in flight before the first store
      constexpr int R4 = REC4 > 0 ? REC4 : 1;
      float4 v[R4];
#pragma unroll
      for (int t = 0; t < R4; ++t) {
        const int e = t * 64 + lane;
        v[t] = data[lane_value64(row, e / R4, wide) * R4 + (e % R4)];
      }
#pragma unroll
      for (int t = 0; t < R4; ++t) {
        typedef float nf4 __attribute__((ext_vector_type(4)));
        if (NT) __builtin_nontemporal_store(*reinterpret_cast<const nf4*>(&v[t]), reinterpret_cast<nf4*>(out + base * R4 + t * 64 + lane));
        else out[base * R4 + t * 64 + lane] = v[t];
      }
    } else {                                      // any width, and the ragged last trip: four records at a time, their pieces in flight together
      const int nr = total_rows - base < RPT ? (int)(total_rows - base) : RPT;
      for (int rr = 0; rr < nr; rr += 4) {
        long long src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) src[u] = lane_value64(row, rr + u < 64 ? rr + u : 63, wide);
        for (int c0 = 0; c0 < rec4; c0 += 64) {
          const int c = c0 + lane < rec4 ? c0 + lane : rec4 - 1;   // the load is unguarded (a valid, unused piece), the store is not
          float4 w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) w[u] = data[src[u] * rec4 + c];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (rr + u < nr && c0 + lane < rec4) out[(base + rr + u) * rec4 + c] = w[u];
        }
      }
    }
  }
}

// The device copy of {size, top} is only read by kernels that DRAW from the ring (the fused gather of a train step, the sample kernels, the
// discriminator's prep).  The fused rollout advances the cursors every vec step; pushing the pair with a launch of its own each time was a
// third launch per rollout step for nothing — the rollout paths now only mark the state stale (replay_mark_state) and every consumer
// brings it up to date first (replay_flush_state, on the ring's stream, ahead of its own launches).
int replay_flush_state(ilsx_replay* rb);
static int replay_mark_state(ilsx_replay* rb) { rb->dstate_stale = true; return ILSX_OK; }
static int replay_push_state(ilsx_replay* rb) {
  rb->dstate_stale = false;
  hipLaunchKernelGGL(k_replay_set_state, dim3(1), dim3(1), 0, rb->ctx->stream, rb->dstate, (long long)rb->size,
                     (long long)rb->top);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
int replay_flush_state(ilsx_replay* rb) { return rb->dstate_stale ? replay_push_state(rb) : ILSX_OK; }

extern "C" int ilsx_replay_create(ilsx_ctx* ctx, int64_t capacity, int obs_dim, int act_dim, uint64_t seed,
                                  ilsx_replay** out) {
  if (!ctx || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_create: NULL argument");
  if (capacity < 1 || obs_dim < 1 || act_dim < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_create: bad sizes");
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_replay* rb = new ilsx_replay();
  rb->ctx = ctx;
  rb->cap = capacity;
  rb->o = obs_dim;
  rb->a = act_dim;
  rb->rec = (2 * obs_dim + act_dim + 4 + 31) / 32 * 32;   // obs | act | rew | done | next_obs | absorbing[2] | pad (Hopper: 29 -> 32 floats)
  rb->seed = seed;
  rb->rng_stream = ctx->next_rng_stream++;
  rb->start_flag.assign((size_t)capacity, 0);
  int rc = ctx_alloc(ctx, (size_t)capacity * rb->rec * sizeof(float), (void**)&rb->data, true);
  if (rc == ILSX_OK) rc = ctx_alloc(ctx, sizeof(DevReplayState), (void**)&rb->dstate, true);
  if (rc != ILSX_OK) { delete rb; return rc; }
  *out = rb;
  return ILSX_OK;
}

extern "C" int ilsx_replay_destroy(ilsx_replay* rb) {
  if (!rb) return ILSX_OK;
  ctx_free(rb->ctx, rb->data);
  ctx_free(rb->ctx, rb->dstate);
  delete rb;
  return ILSX_OK;
}

// host mirror of add_sample/_advance/terminate_episode cursor logic (simple_replay_buffer.py:78-132,228-237)
static void host_advance(ilsx_replay* rb) {
  if (rb->start_flag[rb->top]) {  // O(1) membership test; the overwritten start is normally the oldest entry
    for (auto it = rb->trajs.begin(); it != rb->trajs.end(); ++it)
      if (it->first == rb->top) { rb->trajs.erase(it); break; }
    rb->start_flag[rb->top] = 0;
  }
  rb->top = (rb->top + 1) % rb->cap;
  if (rb->size < rb->cap) rb->size++;
}
static void host_set_endpoint(ilsx_replay* rb, int64_t start, int64_t end) {
  if (rb->start_flag[start]) {
    for (auto& p : rb->trajs)
      if (p.first == start) { p.second = end; return; }  // dict assignment keeps the original position
  }
  rb->start_flag[start] = 1;
  rb->trajs.emplace_back(start, end);
}
static void host_terminate(ilsx_replay* rb) {
  if (rb->cur_start != rb->top) {
    host_set_endpoint(rb, rb->cur_start, rb->top);
    rb->cur_start = rb->top;
  }
}

extern "C" int ilsx_replay_add(ilsx_replay* rb, const float* obs, const float* act, const float* rew,
                               const uint8_t* done, const float* nobs, int n, const uint8_t* ep_end_host,
                               int data_is_device) {
  if (!rb || n < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_add: bad argument");
  if (n == 0) return ILSX_OK;
  if (!obs || !act || !rew || !done || !nobs) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_add: NULL row array");
  if (n > rb->cap) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_add: n=%d exceeds capacity %lld", n, (long long)rb->cap);
  ilsx_ctx* ctx = rb->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  const int o = rb->o, a = rb->a;
  const float *d_obs = obs, *d_act = act, *d_rew = rew, *d_nobs = nobs;
  const uint8_t* d_done = done;
  std::vector<uint8_t> done_host;
  if (!data_is_device) {
    const size_t fl = (size_t)n * (2 * o + a + 1);
    void* st = nullptr;
    ILSX_TRY(ctx_stage(ctx, fl * sizeof(float) + (size_t)n + 64, &st));
    float* f = (float*)st;
    float* s_obs = f; float* s_act = s_obs + (size_t)n * o; float* s_rew = s_act + (size_t)n * a;
    float* s_nobs = s_rew + n; uint8_t* s_done = (uint8_t*)(s_nobs + (size_t)n * o);
    HIPCHK(hipMemcpyAsync(s_obs, obs, (size_t)n * o * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s_act, act, (size_t)n * a * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s_rew, rew, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s_nobs, nobs, (size_t)n * o * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(s_done, done, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    d_obs = s_obs; d_act = s_act; d_rew = s_rew; d_nobs = s_nobs; d_done = s_done;
    done_host.assign(done, done + n);
  } else {
    done_host.resize(n);
    HIPCHK(hipMemcpyAsync(done_host.data(), done, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  const long long total = (long long)n * rb->rec;
  {
  ProfScope ps(ctx, ILSX_K_REPLAY_ADD);
  ILSX_LAUNCH(ps, k_replay_add, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, rb->data, rb->rec,
                     (long long)rb->cap, (long long)rb->top, d_obs, d_act, d_rew, d_done, d_nobs, n, o, a);
  }
  HIPCHK(hipGetLastError());
  if (!data_is_device) HIPCHK(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next call
  for (int i = 0; i < n; ++i) {
    if (done_host[i]) {  // add_sample :95-98
      const int64_t nxt = (rb->top + 1) % rb->cap;
      host_set_endpoint(rb, rb->cur_start, nxt);
      rb->cur_start = nxt;
    }
    host_advance(rb);
    if (ep_end_host && ep_end_host[i]) host_terminate(rb);
  }
  return replay_push_state(rb);
}

// Whole episodes staged by the fused rollout (path mode) enter the ring the way BaseAlgorithm._handle_vec_rollout_ending
// (base_algorithm.py:509-519) does it: ended envs in ascending order, each path's samples appended contiguously through add_sample's
// cursor logic (simple_replay_buffer.py:78-108: a terminal row closes the trajectory at top + 1), then terminate_episode (:126-132).
struct PathEntry { int env, len; long long start; };
__global__ void k_paths_copy(const float* __restrict__ stage, int stage_len, int rec, float* __restrict__ ring, long long cap,
                             const PathEntry* __restrict__ table) {
  const PathEntry e = table[blockIdx.x];
  const int r4 = rec >> 2;   // rec is a multiple of 32 floats
  const float4* src = reinterpret_cast<const float4*>(stage + (size_t)e.env * stage_len * rec);
  for (long long i = threadIdx.x; i < (long long)e.len * r4; i += blockDim.x) {
    const long long t = i / r4, c = i - t * r4;
    long long slot = e.start + t;
    if (slot >= cap) slot -= cap;
    reinterpret_cast<float4*>(ring + (size_t)slot * rec)[c] = src[i];
  }
}
int replay_insert_paths(ilsx_replay* rb, const float* stage, int stage_len, const int* envs, const int* lens, const uint8_t* last_terminal,
                        int n_paths) {
  if (n_paths <= 0) return ILSX_OK;
  ilsx_ctx* ctx = rb->ctx;
  std::vector<PathEntry> table((size_t)n_paths);
  for (int k = 0; k < n_paths; ++k) {
    if (lens[k] < 1 || lens[k] > stage_len || lens[k] >= rb->cap) ILSX_FAIL(ILSX_ERR_ARG, "path of %d samples does not fit (stage %d, capacity %lld)", lens[k], stage_len, (long long)rb->cap);
    table[k] = PathEntry{envs[k], lens[k], (long long)rb->top};
    for (int t = 0; t < lens[k]; ++t) {
      if (t == lens[k] - 1 && last_terminal[k]) {
        const int64_t nxt = (rb->top + 1) % rb->cap;
        host_set_endpoint(rb, rb->cur_start, nxt);
        rb->cur_start = nxt;
      }
      host_advance(rb);
    }
    host_terminate(rb);
  }
  void* st = nullptr;
  ILSX_TRY(ctx_stage(ctx, table.size() * sizeof(PathEntry), &st));
  HIPCHK(hipMemcpyAsync(st, table.data(), table.size() * sizeof(PathEntry), hipMemcpyHostToDevice, ctx->stream));
  long long total = 0;
  for (int k = 0; k < n_paths; ++k) total += lens[k];
  if (total <= rb->cap) {
    hipLaunchKernelGGL(k_paths_copy, dim3(n_paths), dim3(256), 0, ctx->stream, stage, stage_len, rb->rec, rb->data, (long long)rb->cap,
                       (const PathEntry*)st);
  } else {   // one flush larger than the ring: later paths overwrite earlier ones of the same flush, so the copies must be ordered
    for (int k = 0; k < n_paths; ++k)
      hipLaunchKernelGGL(k_paths_copy, dim3(1), dim3(256), 0, ctx->stream, stage, stage_len, rb->rec, rb->data, (long long)rb->cap,
                         (const PathEntry*)st + k);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));   // `table` and the staging buffer are reused by the next call
  return replay_mark_state(rb);
}

int replay_advance_device_rows(ilsx_replay* rb, int n) {
  // rows inserted by the fused rollout kernel carry no trajectory bookkeeping of their own (SAC samples
  // uniformly over rows, simple_replay_buffer.py:242); overwritten trajectory starts are still dropped.
  for (int i = 0; i < n; ++i) host_advance(rb);
  rb->cur_start = rb->top;
  return replay_mark_state(rb);
}

// _absorbing [cap, 2] (simple_replay_buffer.py:66-67,91-92) lives in the two floats behind next_obs of every record; add_sample
// without the keyword leaves [0, 0] (k_replay_add zero-fills the pad).
__global__ void k_replay_absorbing(float* data, int rec, long long cap, long long slot0, int n, int off, const float* absorbing, int set,
                                   const long long* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n) return;
  const int r = i >> 1, c = i & 1;
  long long slot = idx ? idx[r] : slot0 + r;
  if (slot >= cap) slot -= cap;
  float* p = data + (size_t)slot * rec + off + c;
  if (set) *p = absorbing[i];
  else const_cast<float*>(absorbing)[i] = *p;
}
extern "C" int ilsx_replay_set_absorbing(ilsx_replay* rb, int64_t slot0, int n, const float* absorbing_host) {
  if (!rb || n < 0 || slot0 < 0 || slot0 >= rb->cap || n > rb->cap || (!absorbing_host && n)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_set_absorbing: bad argument");
  if (n == 0) return ILSX_OK;
  ilsx_ctx* ctx = rb->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  void* st = nullptr;
  ILSX_TRY(ctx_stage(ctx, (size_t)n * 2 * sizeof(float), &st));
  HIPCHK(hipMemcpyAsync(st, absorbing_host, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_replay_absorbing, dim3((2 * n + 255) / 256), dim3(256), 0, ctx->stream, rb->data, rb->rec, (long long)rb->cap,
                     (long long)slot0, n, 2 * rb->o + rb->a + 2, (const float*)st, 1, (const long long*)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused
  return ILSX_OK;
}
extern "C" int ilsx_replay_get_absorbing(ilsx_replay* rb, const int64_t* idx, int n, float* absorbing) {
  if (!rb || !idx || !absorbing || n < 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_get_absorbing: bad argument");
  HIPCHK(hipSetDevice(rb->ctx->device));
  hipLaunchKernelGGL(k_replay_absorbing, dim3((2 * n + 255) / 256), dim3(256), 0, rb->ctx->stream, rb->data, rb->rec, (long long)rb->cap,
                     0ll, n, 2 * rb->o + rb->a + 2, (const float*)absorbing, 0, (const long long*)idx);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_replay_terminate_episode(ilsx_replay* rb) {
  if (!rb) ILSX_FAIL(ILSX_ERR_ARG, "replay is NULL");
  host_terminate(rb);
  return ILSX_OK;
}

int replay_launch_sample(ilsx_replay* rb, int B, const int64_t* idx, const DevScalars* scal,
                         unsigned long long step_host, float* obs, float* act, float* rew, float* done, float* nobs,
                         int64_t* idx_out) {
  const int total = B * rb->rec;
  ILSX_TRY(replay_flush_state(rb));
  ProfScope ps(rb->ctx, ILSX_K_REPLAY_SAMPLE);
  ILSX_LAUNCH(ps, k_replay_sample, dim3((total + 255) / 256), dim3(256), 0, rb->ctx->stream, rb->data, rb->rec,
                     rb->dstate, (const long long*)idx, rb->seed, rb->rng_stream, scal, step_host, B, rb->o, rb->a, obs,
                     act, rew, done, nobs, (long long*)idx_out);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_replay_sample(ilsx_replay* rb, int B, const int64_t* idx, float* obs, float* act, float* rew,
                                  float* done, float* nobs, int64_t* idx_out) {
  if (!rb || B < 1 || !obs || !act || !rew || !done || !nobs) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_sample: bad argument");
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_replay_sample: buffer is empty");
  HIPCHK(hipSetDevice(rb->ctx->device));
  return replay_launch_sample(rb, B, idx, nullptr, ++rb->sample_ctr, obs, act, rew, done, nobs, idx_out);
}

extern "C" int ilsx_replay_sample_many(ilsx_replay* rb, int n_batches, int B, float* out_records) {
  if (!rb || n_batches < 1 || B < 1 || !out_records) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_replay_sample_many: bad argument");
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_replay_sample_many: buffer is empty");
  HIPCHK(hipSetDevice(rb->ctx->device));
  const long long rows = (long long)n_batches * B;
  const int rec4 = rb->rec / 4;
  ILSX_TRY(replay_flush_state(rb));
  const int rpt = (rec4 == 8 || rec4 == 16) ? 64 : 8;
  long long blocks = (rows + 4 * rpt - 1) / (4 * rpt);   // four wavefronts of `rpt` rows per workgroup
  if (blocks > 256 * 16) blocks = 256 * 16;  // >> 256 CUs, grid-stride the rest
  ProfScope ps(rb->ctx, ILSX_K_REPLAY_SAMPLE_MANY);
  // non-temporal output stores (written once, read by whoever consumes the batch later: keeps the ring, not the output, in the L2 / Infinity
  // Cache) — measured 41.3 -> 39.7 us (Hopper records), 86.1 -> 77.0 us (Walker2d); ILSX_REPLAY_NT=0 for A/B
  static const bool nt = []() { const char* e = getenv("ILSX_REPLAY_NT"); return !e || atoi(e) != 0; }();
#define SM_LAUNCH(R4, N) ILSX_LAUNCH(ps, (k_replay_sample_many<R4, N>), dim3((unsigned)blocks), dim3(256), 0, rb->ctx->stream, (const float4*)rb->data, \
                                     rec4, rb->dstate, rb->seed, rb->rng_stream, rb->sample_ctr + 1, B, rows, (float4*)out_records)
  if (rec4 == 8) { if (nt) SM_LAUNCH(8, true); else SM_LAUNCH(8, false); }
  else if (rec4 == 16) { if (nt) SM_LAUNCH(16, true); else SM_LAUNCH(16, false); }
  else SM_LAUNCH(0, false);
#undef SM_LAUNCH
  rb->sample_ctr += n_batches;
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

extern "C" int ilsx_replay_record_floats(const ilsx_replay* rb, int* out) {
  if (!rb || !out) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  *out = rb->rec;
  return ILSX_OK;
}
extern "C" int ilsx_replay_size(ilsx_replay* rb, int64_t* size, int64_t* top) {
  if (!rb) ILSX_FAIL(ILSX_ERR_ARG, "replay is NULL");
  if (size) *size = rb->size;
  if (top) *top = rb->top;
  return ILSX_OK;
}
extern "C" int ilsx_replay_clear(ilsx_replay* rb) {
  if (!rb) ILSX_FAIL(ILSX_ERR_ARG, "replay is NULL");
  HIPCHK(hipSetDevice(rb->ctx->device));
  HIPCHK(hipMemsetAsync(rb->data, 0, (size_t)rb->cap * rb->rec * sizeof(float), rb->ctx->stream));  // :397-442
  rb->top = rb->size = rb->cur_start = 0;
  rb->trajs.clear();
  std::fill(rb->start_flag.begin(), rb->start_flag.end(), 0);
  return replay_push_state(rb);
}
extern "C" int ilsx_replay_traj_endpoints(ilsx_replay* rb, int64_t* starts, int64_t* ends, int max, int* n) {
  if (!rb || !n) ILSX_FAIL(ILSX_ERR_ARG, "NULL argument");
  int k = 0;
  for (auto& p : rb->trajs) {
    if (k < max && starts && ends) { starts[k] = p.first; ends[k] = p.second; }
    ++k;
  }
  *n = k;
  return ILSX_OK;
}

extern "C" int ilsx_her_gather(ilsx_replay* rb, const int64_t* idx, const int64_t* idx_relabel, int B, int n_relabel, int relabel_on,
                               int d_obs, int d_goal, int reward_kind, float threshold, float* obs_cat, float* act, float* rew,
                               float* done, float* nobs_cat) {
  if (!rb || !idx || B < 1 || !obs_cat || !act || !rew || !done || !nobs_cat) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_her_gather: bad argument");
  if (d_obs < 1 || d_goal < 1 || d_obs + 2 * d_goal != rb->o)
    ILSX_FAIL(ILSX_ERR_ARG, "ilsx_her_gather: ring observation width %d != d_obs %d + 2 * d_goal %d", rb->o, d_obs, d_goal);
  if (rb->a > d_obs + d_goal) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "ilsx_her_gather: act_dim %d > d_obs + d_goal %d", rb->a, d_obs + d_goal);
  if (relabel_on && (n_relabel < 0 || n_relabel > B || (n_relabel > 0 && !idx_relabel))) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_her_gather: n_relabel / idx_relabel");
  if (reward_kind < 0 || reward_kind > 1) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_her_gather: reward_kind must be 0 (sparse) or 1 (dense)");
  if (rb->size < 1) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_her_gather: buffer is empty");
  HIPCHK(hipSetDevice(rb->ctx->device));
  const int total = B * (d_obs + d_goal);
  ProfScope ps(rb->ctx, ILSX_K_REPLAY_SAMPLE);
  ILSX_LAUNCH(ps, k_her_gather, dim3((total + 255) / 256), dim3(256), 0, rb->ctx->stream, rb->data, rb->rec, (const long long*)idx,
              (const long long*)idx_relabel, B, n_relabel, relabel_on, d_obs, d_goal, rb->a, reward_kind, threshold, obs_cat, act, rew,
              done, nobs_cat);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}
