/* ilsx.h — C ABI of libilsx.so, the MI355X (gfx950) engine behind ILSwiss's hot path.
 *
 * The reference (Ericonaldo/ILSwiss) is pure Python and has no FFI; the boundary this library sits
 * behind is the set of duck-typed Python interfaces its training loop calls.  Each
 * entry point below names the reference interface (file:line under /root/reference) whose arithmetic
 * it replaces; ilswiss_amd/*.py binds them with ctypes and re-exposes the reference's class/method
 * names.  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - return value: 0 = OK, <0 = ilsx_status; text via ilsx_last_error() (thread-local). No C++
 *     exception crosses the ABI.
 *   - pointers are DEVICE pointers on the ctx's GPU unless the name/flag says host.
 *   - caller owns every input/output buffer; the library owns what *_create returned until *_destroy.
 *   - a ctx is bound to one HIP stream; calls are asynchronous on that stream unless they return
 *     host-visible values (stats, sizes), in which case they synchronise the stream themselves.
 *   - flat parameter layout of a network ("torch parameters() order", rlkit/torch/common/networks.py:57-83,
 *     policies.py:231-239): fc0.W[H,in] row-major | fc0.b[H] | fc1.W[H,H] | fc1.b[H] | ... |
 *     head0.W[out,H] | head0.b[out] [| head1.W[out,H] | head1.b[out]]      (y = x W^T + b).
 *   - stochastic entry points take optional explicit noise / indices (parity mode) and otherwise use
 *     counter-based Philox4x32-10 keyed by (ctx seed, stream id, step counter).
 */
#ifndef ILSX_H
#define ILSX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILSX_ABI_VERSION 1

typedef enum {
  ILSX_OK = 0,
  ILSX_ERR_ARG = -1,      /* bad argument / shape */
  ILSX_ERR_HIP = -2,      /* a HIP runtime call failed */
  ILSX_ERR_NOMEM = -3,
  ILSX_ERR_STATE = -4,    /* call not valid in the object's current state */
  ILSX_ERR_UNSUPPORTED = -5
} ilsx_status;

typedef struct ilsx_ctx ilsx_ctx;
typedef struct ilsx_net ilsx_net;
typedef struct ilsx_replay ilsx_replay;
typedef struct ilsx_sac ilsx_sac;
typedef struct ilsx_sac_group ilsx_sac_group;
typedef struct ilsx_vecenv ilsx_vecenv;
typedef struct ilsx_disc ilsx_disc;
typedef struct ilsx_ppo ilsx_ppo;
typedef struct ilsx_td3 ilsx_td3;
typedef struct ilsx_sacv ilsx_sacv;
typedef struct ilsx_bc ilsx_bc;

enum { ILSX_ACT_RELU = 0, ILSX_ACT_TANH = 1 };

/* ---------------------------------------------------------------- context */
int ilsx_abi_version(void);
const char* ilsx_last_error(void);
/* stream: an existing hipStream_t to run on (e.g. torch's current stream), or NULL to create one. */
int ilsx_ctx_create(int hip_device, void* hip_stream, uint64_t seed, ilsx_ctx** out);
int ilsx_ctx_sync(ilsx_ctx* ctx);
int ilsx_ctx_destroy(ilsx_ctx* ctx);
void* ilsx_ctx_stream(ilsx_ctx* ctx);
/* Every object that draws random numbers (vec env, replay ring, trainer, discriminator) takes the ctx's next Philox stream id when it is
   created.  Reads the cursor into *current (nullable) and, when set_to > 0, moves it: an object built on ANOTHER ctx of the same run (the eval
   env of rl_alg_params.eval_async lives on a stream of its own) can then take the id it would have had on this one, and this ctx skips it, so
   the run's other objects keep their streams.  The reference has one global torch / numpy generator per process (no counterpart). */
int ilsx_ctx_rng_stream_cursor(ilsx_ctx* ctx, uint32_t set_to, uint32_t* current);
/* Device scratch owned by the ctx (freed with it); used by the Python adapters for staging. */
int ilsx_ctx_alloc(ilsx_ctx* ctx, size_t bytes, void** out);
int ilsx_ctx_free(ilsx_ctx* ctx, void* ptr);
int ilsx_memcpy_h2d(ilsx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int ilsx_memcpy_d2h(ilsx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* syncs */

/* ---------------------------------------------------------------- split-run communicator
 * The path's ONE exchange step: when a single run is split over G GPUs (one process / ctx per GPU, every rank a full
 * replica of the parameters, B/G rows of the batch each, mean-loss gradients pre-scaled by 1/(B*G) through
 * ilsx_sac_cfg.grad_world), the flat gradient arena is all-reduced (sum) between backward and the optimiser step.  No
 * reference counterpart (run_experiment.py:57-78 runs independent processes only).  The collective is RCCL's
 * ncclAllReduce enqueued on the ctx stream; librccl.so.1 is dlopen'ed at the first call here.  Rank 0 draws the id, the
 * host program carries its 128 bytes to the other ranks (file, socket, torch.distributed — ilswiss_amd/parallel.py), then
 * every rank calls ilsx_comm_init.  With a communicator of grad_world ranks on the ctx, ilsx_sac_train_step and
 * ilsx_sac_train_from_replay run critic-backward -> all-reduce -> critic Adam+Polyak -> actor-backward -> all-reduce
 * (alpha gradient in the same message) -> actor Adam, all stream-ordered, no host synchronisation. */
#define ILSX_COMM_ID_BYTES 128
int ilsx_comm_unique_id(uint8_t* id_host /* [ILSX_COMM_ID_BYTES] */);
int ilsx_comm_init(ilsx_ctx* ctx, const uint8_t* id_host, int n_ranks, int rank);
int ilsx_comm_destroy(ilsx_ctx* ctx);
int ilsx_comm_info(const ilsx_ctx* ctx, int* n_ranks, int* rank);   /* 0 ranks = no communicator */
int ilsx_comm_allreduce_sum(ilsx_ctx* ctx, float* dev_buf, size_t n);   /* in place, on the ctx stream */

/* ---------------------------------------------------------------- kernel timing (bench.py roofline leg)
 * When enabled, every launch of a library kernel carries a start / stop hipEvent stamped with the dispatch's own
 * begin / end time (hipExtLaunchKernelGGL — the interval rocprofv3's kernel trace reports; the graph path is
 * bypassed while profiling); ilsx_prof_read synchronises and returns launches + summed ms of a slot,
 * ilsx_prof_kernel the source spelling of the kernel last launched under it (several kernels share a slot:
 * the column-split and the generic MLP kernels). */
enum { ILSX_K_MLP_FWD = 0, ILSX_K_MLP_BWD_DX = 1, ILSX_K_MLP_BWD_DW = 2, ILSX_K_ADAM = 3,
       ILSX_K_REPLAY_SAMPLE = 4, ILSX_K_REPLAY_ADD = 5, ILSX_K_REPLAY_SAMPLE_MANY = 6, ILSX_K_SAC_STATS = 7,
       ILSX_K_SAC_FINISH = 8, ILSX_K_ENV_STEP = 9, ILSX_K_POLICY_FINISH = 10, ILSX_K_DISC_BWD = 11, ILSX_K_PPO_GAE = 12,
       ILSX_K_SAC_PHASE_A = 13, ILSX_K_SAC_PHASE_C = 14,   /* the merged phase launches of the single-run SAC step */
       ILSX_K_COUNT = 16 };
int ilsx_prof_enable(ilsx_ctx* ctx, int on);
int ilsx_prof_reset(ilsx_ctx* ctx);
int ilsx_prof_read(ilsx_ctx* ctx, int kernel_id, uint64_t* launches, double* total_ms);
const char* ilsx_prof_kernel(ilsx_ctx* ctx, int kernel_id);
const char* ilsx_kernel_name(int kernel_id);
/* debugging aid (tools/step_gantt.py): while a trace buffer is set, thread 0 of EVERY workgroup of every MLP forward /
 * backward / weight-gradient launch writes its phase timestamps (100 MHz constant clock) into
 * dev_trace[launch][2048 workgroups][8 slots] (uint64; launch = order of launch since the buffer was set, at most
 * max_launches; slot 0 = workgroup start, 7 = end, 1..3 = phase boundaries).  NULL = off; launches_so_far (nullable)
 * receives the number of launches recorded into the PREVIOUS buffer. */
int ilsx_debug_set_stamp_buffer(ilsx_ctx* ctx, void* dev_trace, int max_launches, int* launches_so_far);
/* Host arithmetic only (no device needed): how a weight-gradient launch over `rows` batch rows is cut into row ranges (`big` != 0: the
 * large-batch block kernel).  Every range is non-empty and a whole number of the kernel's row steps; together they cover [0, rows).
 * The reference has no counterpart (torch.autograd sums the batch in one GEMM, e.g. ppo.py:145-153): test hook for the split plan. */
int ilsx_debug_dw_split(int rows, int big, int* splits, int* rows_per_split);

/* ---------------------------------------------------------------- networks
 * Replaces rlkit/torch/common/networks.py:23-115 (Mlp / FlattenMlp) and the heads of
 * rlkit/torch/common/policies.py:191-345 (n_heads = 2: mean | log_std). */
typedef struct {
  int32_t in_dim;      /* FlattenMlp: sum of the concatenated input dims */
  int32_t n_hidden;    /* 1..3 hidden layers */
  int32_t hidden;      /* the width the kernels run at: 64, 128 or 256 */
  int32_t out_dim;     /* per head */
  int32_t n_heads;     /* 1 (Mlp) or 2 (Gaussian policy: last_fc, last_fc_log_std) */
  int32_t act;         /* ILSX_ACT_* hidden activation */
  int32_t hidden_sizes[3];   /* networks.py:23-60 takes ANY list of widths: hidden_sizes[l] = the logical width of hidden layer l, 1..hidden
                                (0 = hidden).  A layer narrower than `hidden` is embedded in the `hidden`-wide kernels as structural zeros:
                                units past its width have zero weights and bias, hence pre-activation 0, output 0 (relu and tanh), gradient 0,
                                and Adam / Polyak / L2 never move them — the padded network IS the hidden_sizes network.  Every flat parameter /
                                gradient / optimiser-state vector that crosses this ABI has the LOGICAL sizes (torch's parameters() order). */
} ilsx_mlp_cfg;

int ilsx_net_create(ilsx_ctx* ctx, const ilsx_mlp_cfg* cfg, ilsx_net** out);
int ilsx_net_destroy(ilsx_net* net);
int ilsx_net_num_params(const ilsx_net* net, size_t* out);
/* networks.py:57-83 + pytorch_util.py:20-29 init rule: hidden W ~ U(+-1/sqrt(out_features)),
 * hidden b = b_init, heads W,b ~ U(+-init_w); host RNG (splitmix64/xoshiro) seeded by `seed`. */
int ilsx_net_init(ilsx_net* net, uint64_t seed, float init_w, float b_init);
int ilsx_net_set_params(ilsx_net* net, const float* src, size_t n, int src_is_device);
int ilsx_net_get_params(const ilsx_net* net, float* dst, size_t n, int dst_is_device);
/* Mlp.forward (networks.py:85-101): x[rows,in_dim] -> y[rows, n_heads*out_dim] (raw head outputs). */
int ilsx_mlp_forward(ilsx_net* net, const float* x, int rows, float* y);
/* ReparamTanhMultivariateGaussianPolicy.get_actions / forward (policies.py:241-307):
 * obs[n,o] -> act[n,a]; deterministic: tanh(mean); else tanh(mean + std*eps), eps[n,a] or NULL=Philox.
 * logp (nullable) [n] receives log pi(a|s) (distributions.py:74-97). */
int ilsx_policy_act(ilsx_net* pi, const float* obs, int n, int deterministic, const float* eps,
                    float* act, float* logp);
/* ReparamTanhMultivariateNormal.log_prob with pre_tanh_value=None (policies.py:329-345):
 * logp[n] of GIVEN actions act[n,a] under pi(.|obs). */
int ilsx_policy_log_prob(ilsx_net* pi, const float* obs, const float* act, int n, float* logp);
/* Marks a single-head Mlp as MlpGaussianNoisePolicy (policies.py:130-188): ilsx_policy_act / ilsx_rollout_step then
 * return max_act*tanh(out) + clip(policy_noise*N(0,1), +-policy_noise_clip) (no noise when deterministic). */
int ilsx_net_set_noise_policy(ilsx_net* pi, float policy_noise, float policy_noise_clip, float max_act);
/* Output activation of a noise policy: 0 = tanh (what td3_exp_script.py:75 passes), 1 = identity (Mlp's default, networks.py:31:
 * action = max_act * last_fc(h) + clipped noise).  Set it before ilsx_td3_create adopts the network. */
int ilsx_net_set_output_linear(ilsx_net* pi, int linear);

/* ---------------------------------------------------------------- replay buffer
 * Replaces rlkit/data_management/simple_replay_buffer.py:17-442 + env_replay_buffer.py:7-49.
 * HBM-resident ring of transition records (obs|act|rew|done|next_obs|absorbing, one 128-byte-aligned record per
 * transition).  Cursor semantics (_top/_size/_traj_endpoints) follow the reference exactly. */
int ilsx_replay_create(ilsx_ctx* ctx, int64_t capacity, int obs_dim, int act_dim, uint64_t seed,
                       ilsx_replay** out);
int ilsx_replay_destroy(ilsx_replay* rb);
/* n x add_sample (+ terminate_episode() after rows whose ep_end flag is set; simple_replay_buffer.py:78-132).
 * Row arrays are contiguous [n,dim]; `data_is_device` says where obs/act/rew/done/nobs live;
 * ep_end_host is a HOST array of n flags or NULL. */
int ilsx_replay_add(ilsx_replay* rb, const float* obs, const float* act, const float* rew,
                    const uint8_t* done, const float* nobs, int n, const uint8_t* ep_end_host,
                    int data_is_device);
int ilsx_replay_terminate_episode(ilsx_replay* rb);
/* _absorbing[cap,2] (simple_replay_buffer.py:66-67,91-92; the wrap_absorbing branch of add_path, :163-213): the flags of the n
 * slots from slot0 on (ring order) are set from HOST absorbing[n,2]; rows added without them hold [0,0].  get: device
 * idx int64[n] -> device absorbing[n,2] (the `absorbing` key of random_batch, :266-268). */
int ilsx_replay_set_absorbing(ilsx_replay* rb, int64_t slot0, int n, const float* absorbing_host);
int ilsx_replay_get_absorbing(ilsx_replay* rb, const int64_t* idx, int n, float* absorbing);
/* random_batch (simple_replay_buffer.py:239-293): idx (device int64[B]) or NULL = uniform with
 * replacement over [0,size) from Philox.  Outputs (device): obs[B,o] act[B,a] rew[B] done[B] (0/1 as
 * float, rlkit/torch/core.py:124-143) nobs[B,o]; idx_out (nullable device int64[B]) gets the rows used. */
int ilsx_replay_sample(ilsx_replay* rb, int B, const int64_t* idx, float* obs, float* act, float* rew,
                       float* done, float* nobs, int64_t* idx_out);
/* Bandwidth microbenchmark form of the same kernel: n_batches x B rows gathered in ONE launch into
 * out[n_batches*B, record] (device, record = ilsx_replay_record_floats()). */
int ilsx_replay_sample_many(ilsx_replay* rb, int n_batches, int B, float* out_records);
int ilsx_replay_record_floats(const ilsx_replay* rb, int* out);
/* Hindsight relabelling gather — HindsightReplayBuffer.random_batch (rlkit/data_management/relabel_replay_buffer.py:66-163) over a ring
 * whose observation segment is [observation (d_obs) | desired_goal (d_goal) | achieved_goal (d_goal)].  idx / idx_relabel: DEVICE int64[B]
 * (the sampled steps and their `future` / `final` relabel steps, drawn by the caller in the reference's RandomState order); the first
 * n_relabel rows get the relabel step's next achieved goal as desired goal (:106-113); relabel_on: every reward is recomputed from
 * (next achieved goal, desired goal) (:139-147) by gym's GoalEnv rule, reward_kind 0 = sparse -(dist > threshold), 1 = dense -dist.
 * Device outputs: obs_cat[B, d_obs + d_goal] = observation | desired_goal and nobs_cat likewise (what her/td3.py:95-99 and
 * her/sac.py:80-84 feed the networks), act[B, a], rew[B], done[B]. */
int ilsx_her_gather(ilsx_replay* rb, const int64_t* idx, const int64_t* idx_relabel, int B, int n_relabel, int relabel_on, int d_obs,
                    int d_goal, int reward_kind, float threshold, float* obs_cat, float* act, float* rew, float* done, float* nobs_cat);
int ilsx_replay_size(ilsx_replay* rb, int64_t* size, int64_t* top);
int ilsx_replay_clear(ilsx_replay* rb);
/* _traj_endpoints in insertion order: writes up to max pairs (start,end) to HOST arrays; *n = count. */
int ilsx_replay_traj_endpoints(ilsx_replay* rb, int64_t* starts_host, int64_t* ends_host, int max, int* n);

/* ---------------------------------------------------------------- SAC (twin Q, auto alpha)
 * Replaces rlkit/torch/algorithms/sac/sac_alpha.py:21-76 (ctor kwargs == cfg fields) and
 * :78-181 (train_step), :245-247 (_update_target_network). */
typedef struct {
  float reward_scale, discount, policy_lr, qf_lr, alpha_lr, soft_target_tau, alpha;
  int32_t train_alpha;
  float policy_mean_reg_weight, policy_std_reg_weight, beta_1;
  int32_t has_target_entropy;  /* 0: default -dim(A)/2 (sac_alpha.py:56-58) */
  float target_entropy;
  int32_t max_batch;           /* largest B any train call will use (workspace size) */
  int32_t grad_world;          /* split-run mode: number of ranks sharing one batch (1 = off); the
                                  mean-over-batch scale becomes 1/(B*grad_world) */
} ilsx_sac_cfg;

typedef struct {              /* scalars of sac_alpha.py:186-233 for the step just taken */
  float qf1_loss, qf2_loss, policy_loss, alpha_loss, alpha;
  float q1_mean, q2_mean, log_pi_mean, policy_mu_mean, policy_log_std_mean;
  double log_alpha;
  /* Std / Max / Min of {Q1 Predictions, Q2 Predictions, Log Pis, Policy mu, Policy log std} (sac_alpha.py:202-233) */
  float ext_std[5], ext_max[5], ext_min[5];
} ilsx_sac_stats;

/* Takes over the storage of pi/q1/q2 (they become views into the agent's parameter arena and stay
 * valid for ilsx_policy_act / get_params until the agent is destroyed). */
int ilsx_sac_create(ilsx_ctx* ctx, const ilsx_sac_cfg* cfg, ilsx_net* pi, ilsx_net* q1, ilsx_net* q2,
                    ilsx_sac** out);
int ilsx_sac_destroy(ilsx_sac* sac);
/* One SoftActorCritic.train_step on an explicit batch (device pointers, fp32; rew/done are [B]).
 * eps_next / eps_cur: the two N(0,1) draws [B,a] (sac_alpha.py:102,142) or NULL = Philox. */
int ilsx_sac_train_step(ilsx_sac* sac, const float* obs, const float* act, const float* rew,
                        const float* done, const float* nobs, int B, const float* eps_next,
                        const float* eps_cur, ilsx_sac_stats* stats);
/* TorchRLAlgorithm._do_training (torch_rl_algorithm.py:28-34): n_steps x (random_batch + train_step)
 * with on-device sampling; one hipGraph replay per step, no host round trip.  stats (nullable): the statistics of the
 * FIRST batch of the call — the reference fills eval_statistics on the first train_step after end_epoch
 * (sac_alpha.py:185-190) and the caller passes `stats` exactly then. */
int ilsx_sac_train_from_replay(ilsx_sac* sac, ilsx_replay* rb, int n_steps, int B, ilsx_sac_stats* stats);
/* A single run's steps inside ilsx_sac_train_from_replay / ilsx_advirl_train run on merged "phase" kernels whose workgroups hand data to
 * each other inside one launch; that needs the GPU to this process.  The reference's launcher starts every worker of a sweep on the SAME
 * GPU (run_experiment.py:57-78), so the library checkpoints parameters / optimiser state / counters at the start of every such window
 * and, if a hand-off timed out or a row tile's workgroups were placed on two XCDs, rolls the window back and re-runs it on one launch
 * per stage (where the agent then stays): the caller sees the same updates either way.  This reports what happened:
 * fallbacks = windows rolled back so far, disabled = 1 once the agent has left the phase kernels, last_window_on_phase = 1 if the most
 * recent window used them, wgs_per_cu_a / _c = resident workgroups per CU the runtime's occupancy calculator gives the two kernels
 * (what the "every waited-for workgroup is resident" check is made with).  Every output nullable. */
int ilsx_sac_phase_state(ilsx_sac* sac, int* fallbacks, int* disabled, int* last_window_on_phase, int* wgs_per_cu_a, int* wgs_per_cu_c);
/* test aid: the next window behaves as if one of its hand-offs had timed out (exercises the roll-back path on an exclusive GPU) */
int ilsx_sac_debug_break_phase(ilsx_sac* sac);
/* Split-run (multi-GPU) phases: train_step == critic_backward ; critic_update ; actor_backward ;
 * actor_update, with an all-reduce(sum) of the gradient arena between backward and update. */
int ilsx_sac_set_batch(ilsx_sac* sac, const float* obs, const float* act, const float* rew,
                       const float* done, const float* nobs, int B, const float* eps_next,
                       const float* eps_cur);
int ilsx_sac_critic_backward(ilsx_sac* sac);
int ilsx_sac_critic_update(ilsx_sac* sac);
int ilsx_sac_actor_backward(ilsx_sac* sac);
int ilsx_sac_actor_update(ilsx_sac* sac);
/* The statistics most recently asked for (a train call with `stats` / a grouped call with want_stats), as they stood right after their
 * step; after hand-driven phases: the device scalars as they are now. */
int ilsx_sac_last_stats(ilsx_sac* sac, ilsx_sac_stats* stats);
/* flat fp32 views: which = 0 pi, 1 q1, 2 q2, 3 target_q1, 4 target_q2 */
int ilsx_sac_get_params(ilsx_sac* sac, int which, float* dst, size_t n, int dst_is_device);
int ilsx_sac_set_params(ilsx_sac* sac, int which, const float* src, size_t n, int src_is_device);
/* gradients of the last backward, same layout/which as the parameters (0 pi, 1 q1, 2 q2) */
int ilsx_sac_get_grads(ilsx_sac* sac, int which, float* dst, size_t n, int dst_is_device);
/* gradient arena for the RCCL all-reduce: segment 0 = critics (q1|q2), 1 = actor (pi | alpha slot) */
int ilsx_sac_grads_ptr(ilsx_sac* sac, int segment, float** dev_ptr, size_t* n);
int ilsx_sac_get_log_alpha(ilsx_sac* sac, double* out);
int ilsx_sac_set_log_alpha(ilsx_sac* sac, double v);
/* optimiser state for snapshots (sac_alpha.py:249-273): which = 0 pi, 1 q1, 2 q2; m,v host fp32[n] */
int ilsx_sac_get_adam(ilsx_sac* sac, int which, float* m_host, float* v_host, size_t n, int64_t* t);
int ilsx_sac_set_adam(ilsx_sac* sac, int which, const float* m_host, const float* v_host, size_t n, int64_t t);
/* alpha_optimizer state (float64 scalar Adam, sac_alpha.py:74-76) + the Philox step counter */
int ilsx_sac_get_alpha_opt(ilsx_sac* sac, double* m, double* v, int64_t* t, uint64_t* rng_step);
int ilsx_sac_set_alpha_opt(ilsx_sac* sac, double m, double v, int64_t t, uint64_t rng_step);

/* Parity aids for the fused path (tests only; nothing in the training path calls them).
 * ilsx_sac_debug_batch re-derives, with the STANDALONE sample kernel (the one behind ilsx_replay_sample) and a
 * standalone Philox kernel, the batch rows and the two N(0,1) draws that gradient step number `step` of this agent
 * takes when it samples from `rb` inside ilsx_sac_train_from_replay / ilsx_sac_group_train_from_replay (`step` = the
 * agent's Philox step counter when that step runs: 0 for a fresh agent's first step; ilsx_sac_get_alpha_opt's
 * rng_step is the counter of the NEXT step).  All outputs are device pointers: obs[B,o] act[B,a] rew[B] done[B]
 * nobs[B,o] eps_next[B,a] eps_cur[B,a] idx int64[B]; every one nullable.
 * ilsx_sac_debug_last_batch copies what the LAST step actually used out of the agent's workspace: the rows its first
 * forward launch gathered and published (obs/act/rew/done/nobs) and the eps_cur its policy head consumed. */
int ilsx_sac_debug_batch(ilsx_sac* sac, ilsx_replay* rb, uint64_t step, int B, float* obs, float* act, float* rew,
                         float* done, float* nobs, float* eps_next, float* eps_cur, int64_t* idx);
int ilsx_sac_debug_last_batch(ilsx_sac* sac, int B, float* obs, float* act, float* rew, float* done, float* nobs,
                              float* eps_cur);
/* Known-answer aid (tests only): device raw[n_rows*4] = the Philox4x32-10 block of counter (row, 0, step lo, step hi ^ stream*0x9E3779B9)
 * under key (seed lo, seed hi ^ stream); device normals[n_rows*a] = the N(0,1) draws every stochastic policy epilogue of the library
 * makes for (seed, step, stream, row, dim).  Either output nullable. */
/* kind: 0 = ilsx_replay, 1 = ilsx_sac (its two streams are id, id + 1: eps_next, eps_cur), 2 = ilsx_disc.  The Philox stream id the
 * object was given at creation (ids are handed out per ctx in creation order) and the seed its draws are keyed by. */
int ilsx_debug_rng_stream(const void* object, int kind, uint32_t* stream, uint64_t* seed);
int ilsx_debug_philox(ilsx_ctx* ctx, uint64_t seed, uint64_t step, uint32_t stream, int n_rows, int a, uint32_t* raw, float* normals);

/* ---------------------------------------------------------------- adversarial-IRL discriminator
 * Replaces rlkit/torch/algorithms/adv_irl/disc_models/simple_disc_models.py:8-48 (MLPDisc, with and without batch norm),
 * AdvIRL._do_reward_training (adv_irl.py:133-216: BCE-with-logits on [expert; policy] + WGAN-GP gradient
 * penalty, the double backward derived by hand) and the reward modes of _do_policy_training (:277-298).
 * cfg fields == the YAML keys (exp_specs/gail/gail_walker.yaml:24-28,55-59). */
enum { ILSX_DISC_AIRL = 0, ILSX_DISC_GAIL = 1, ILSX_DISC_GAIL2 = 2, ILSX_DISC_FAIRL = 3 };
typedef struct {
  int32_t obs_dim, act_dim;       /* discriminator input = cat(obs, act); state_only: cat(obs, next_obs), act_dim == obs_dim */
  int32_t hid_dim, hid_act;       /* layer blocks of hid_dim (64/128/256), ILSX_ACT_* */
  int32_t use_grad_pen;
  float clamp_magnitude, disc_lr, disc_momentum, grad_pen_weight;
  int32_t max_batch;              /* disc_optim_batch_size upper bound (rows per class) */
  int32_t state_only;             /* adv_irl.py:140-162,269: discriminator input = cat(obs, next_obs); act_dim must equal obs_dim and
                                     every `act` row pointer of the entry points below carries next_obs rows */
  int32_t num_layer_blocks;       /* simple_disc_models.py:11,29-39: hidden (Linear, act) blocks, 1..3; 0 = 2.  2 runs the fused
                                     double-backward kernel, 1 and 3 the same mathematics as a chain of per-layer launches */
  int32_t use_bn;                 /* simple_disc_models.py:15,30-31,36-37 (the constructor's default): every block is Linear -> BatchNorm1d ->
                                     act.  Train mode in both forwards of a step (cross-entropy rows and penalty interpolates, each with its
                                     own batch statistics, running statistics updated by both), the penalty's double backward through the
                                     batch statistics; ilsx_disc_reward uses the running statistics (eval mode, adv_irl.py:268-274).  Any
                                     hid_dim; parameters in torch's order: per block W | b | gamma | beta, then the output layer.  A chain of
                                     simple launches (csrc/disc_bn.h), not the fused MFMA kernels of use_bn = 0 */
  int32_t grad_world;             /* split-run mode (SURVEY section 8e "Disc: same"; 0 / 1 = off): this rank holds B / G rows per class (and B / G
                                     interpolates); the cross-entropy and penalty means are over B * G rows, the gradient arena is all-reduced
                                     (sum) on the ctx's communicator before Adam; the logged statistics are this rank's rows'.  use_bn = 1 is
                                     refused (batch statistics would have to cross ranks) */
} ilsx_disc_cfg;
typedef struct { float ce_loss, grad_pen, accuracy; } ilsx_disc_stats;   /* "Disc CE Loss", "Grad Pen", "Disc Acc" */
int ilsx_disc_create(ilsx_ctx* ctx, const ilsx_disc_cfg* cfg, ilsx_disc** out);
int ilsx_disc_destroy(ilsx_disc* disc);
int ilsx_disc_num_params(const ilsx_disc* disc, size_t* out);
/* flat layout: fc0.W | fc0.b | fc1.W | fc1.b | out.W | out.b (nn.Sequential parameters() order); HOST arrays */
int ilsx_disc_set_params(ilsx_disc* disc, const float* src_host, size_t n);
int ilsx_disc_get_params(ilsx_disc* disc, float* dst_host, size_t n);
int ilsx_disc_get_grads(ilsx_disc* disc, float* dst_host, size_t n);
/* use_bn discriminators: the BatchNorm running statistics (module buffers, not parameters: snapshots carry them beside the parameters),
 * HOST arrays [num_layer_blocks * hid_dim] each, block-major */
int ilsx_disc_get_bn_stats(ilsx_disc* disc, float* running_mean_host, float* running_var_host, size_t n);
int ilsx_disc_set_bn_stats(ilsx_disc* disc, const float* running_mean_host, const float* running_var_host, size_t n);
/* one _do_reward_training step; inputs are device rows exp_obs[B,o] exp_act[B,a] pol_obs[B,o] pol_act[B,a];
 * eps (device [B], U[0,1) interpolation weights) or NULL = Philox. */
int ilsx_disc_train_step(ilsx_disc* disc, const float* exp_obs, const float* exp_act, const float* pol_obs,
                         const float* pol_act, int B, const float* eps, ilsx_disc_stats* stats);
/* reward relabelling: rew[n] (nullable) by mode + optional clip; logits[n] (nullable) = clamped D(s,a) */
int ilsx_disc_reward(ilsx_disc* disc, const float* obs, const float* act, int n, int mode, int has_min, float rew_clip_min,
                     int has_max, float rew_clip_max, float* rew, float* logits);

/* ---------------------------------------------------------------- co-resident seeds
 * The reference runs a seed sweep as independent processes (run_experiment.py:57-78).  A group steps K independent
 * SoftActorCritic agents of identical shape in lock-step on ONE GPU: every stage of the fused step is one launch whose grid
 * carries all agents' tasks ("grouped GEMMs, weights differ per seed"), so K runs cost 9 dependent launches per step
 * instead of 9K.  The agents stay ordinary ilsx_sac objects (parameters, statistics, snapshots through their own entry
 * points); results are bit-identical to stepping each agent alone with ilsx_sac_train_from_replay. */
int ilsx_sac_group_create(ilsx_ctx* ctx, ilsx_sac* const* agents, int n_agents, ilsx_sac_group** out);
int ilsx_sac_group_destroy(ilsx_sac_group* group);
/* n_steps gradient steps of every agent; agent k samples its batch from rbs[k].  want_stats: the FIRST step of the call also
 * computes every agent's statistics (sac_alpha.py:185-190; read them with ilsx_sac_last_stats).
 * The agents may live in sibling contexts of the group's ctx — contexts created on the SAME device and HIP stream
 * (ilsx_ctx_create(device, ilsx_ctx_stream(ctx), seed_k, ...)): each run then keeps the Philox key and per-object stream ids it would have in a
 * process of its own, which is what makes K grouped seeds reproduce K single-process runs (run_experiment.py --group). */
int ilsx_sac_group_train_from_replay(ilsx_sac_group* group, ilsx_replay* const* rbs, int n_steps, int B, int want_stats);

/* AdvIRL._do_training (adv_irl.py:126-131) for one train call: `loops` x { disc_updates discriminator steps ;
 * policy_updates SAC steps on rewards relabelled by the discriminator (mode / clips as ilsx_disc_reward) }, every batch
 * drawn on the device from the expert / policy replay rings (get_batch, :106-113).  Statistics (all nullable, host) are
 * those of the first discriminator / policy batch of the call; rew_stats4 = {Mean, Std, Max, Min} of its rewards. */
int ilsx_advirl_train(ilsx_disc* disc, ilsx_sac* policy_trainer, ilsx_replay* expert_rb, ilsx_replay* policy_rb, int loops,
                      int disc_updates, int policy_updates, int disc_batch, int policy_batch, int mode, int has_min,
                      float rew_clip_min, int has_max, float rew_clip_max, ilsx_disc_stats* disc_stats,
                      ilsx_sac_stats* sac_stats, float* rew_stats4);
/* policy_optim_batch_size_from_expert (adv_irl.py:239-255): the last n_from_expert rows of every policy batch are drawn from the
 * expert ring (torch.cat([policy rows, expert rows])), relabelled and trained on like the others.  0 (default) = off. */
int ilsx_advirl_set_policy_batch_from_expert(ilsx_disc* disc, int n_from_expert);

/* ---------------------------------------------------------------- TD3
 * Replaces rlkit/torch/algorithms/td3/td3.py:21-70 (ctor), :72-124 (train_step), :180-183 (soft updates).  cfg fields ==
 * the YAML keys of exp_specs/td3/td3_hopper.yaml:12-13,39-45 (policy_noise / policy_noise_clip are the noise of the
 * policy MODULE, which the target policy inherits through policy.copy(); td3.py's target_policy_noise* are unused).
 * The value list of a statistic is {Mean, Std, Max, Min} (core/eval_util.py create_stats_ordered_dict). */
typedef struct {
  float reward_scale, discount, policy_lr, qf_lr;
  int32_t policy_and_target_update_period;
  float soft_target_tau, policy_noise, policy_noise_clip, max_act;
  int32_t max_batch;
  /* her != 0: rlkit/torch/algorithms/her/td3.py (goal-conditioned TD3; the caller concatenates observation | desired_goal): target
   * action = clamp(policy_noise * N(0,1), -max_act, max_act) as that file computes it (:104-114), target value clipped to
   * [clip_return_l, clip_return_r] (:116-122; defaults -1/(1-discount), 0 are the caller's to fill), policy loss + mean(a^2) (:148-152) */
  int32_t her;
  float clip_return_l, clip_return_r;
} ilsx_td3_cfg;
typedef struct {
  float qf1_loss, qf2_loss, policy_loss;
  float q1_pred[4], q2_pred[4], q_target[4], bellman1[4], bellman2[4], policy_action[4];
} ilsx_td3_stats;
/* adopts the three networks' storage like ilsx_sac_create; pi: single-head Mlp (becomes a noise policy) */
int ilsx_td3_create(ilsx_ctx* ctx, const ilsx_td3_cfg* cfg, ilsx_net* pi, ilsx_net* qf1, ilsx_net* qf2, ilsx_td3** out);
int ilsx_td3_destroy(ilsx_td3* td3);
/* device batch rows as in ilsx_sac_train_step; eps_target: device [B,a] N(0,1) draws of the target policy's noise
 * (policies.py:182-184) or NULL = Philox; stats nullable (host). */
int ilsx_td3_train_step(ilsx_td3* td3, const float* obs, const float* act, const float* rew, const float* done,
                        const float* nobs, int B, const float* eps_target, ilsx_td3_stats* stats);
int ilsx_td3_train_from_replay(ilsx_td3* td3, ilsx_replay* rb, int n_steps, int B, ilsx_td3_stats* stats);
/* which: 0 qf1, 1 qf2, 2 policy, 3 target_qf1, 4 target_qf2, 5 target_policy; HOST flat arrays */
int ilsx_td3_get_params(ilsx_td3* td3, int which, float* dst_host, size_t n);
int ilsx_td3_set_params(ilsx_td3* td3, int which, const float* src_host, size_t n);

/* ---------------------------------------------------------------- SAC with a state-value function
 * Replaces rlkit/torch/algorithms/sac/sac.py:23-68 (ctor), :70-179 (train_step), :242-243 (soft update of V).
 * cfg fields == the YAML `sac_params` keys (exp_specs/sac/sac_hopper.yaml:36-47 with run_scripts/sac_exp_script.py). */
typedef struct {
  float reward_scale, discount, alpha, policy_lr, qf_lr, vf_lr, soft_target_tau;
  float policy_mean_reg_weight, policy_std_reg_weight, beta_1;
  int32_t max_batch;
} ilsx_sacv_cfg;
typedef struct {
  float qf1_loss, qf2_loss, vf_loss, policy_loss;
  float q1_pred[4], q2_pred[4], v_pred[4], log_pi[4], policy_mu[4], policy_log_std[4];
} ilsx_sacv_stats;
int ilsx_sacv_create(ilsx_ctx* ctx, const ilsx_sacv_cfg* cfg, ilsx_net* pi, ilsx_net* qf1, ilsx_net* qf2, ilsx_net* vf,
                     ilsx_sacv** out);
int ilsx_sacv_destroy(ilsx_sacv* sac);
/* eps: device [B,a] N(0,1) draws of the single policy sample of the step (sac.py:123) or NULL = Philox */
int ilsx_sacv_train_step(ilsx_sacv* sac, const float* obs, const float* act, const float* rew, const float* done,
                         const float* nobs, int B, const float* eps, ilsx_sacv_stats* stats);
int ilsx_sacv_train_from_replay(ilsx_sacv* sac, ilsx_replay* rb, int n_steps, int B, ilsx_sacv_stats* stats);
/* which: 0 qf1, 1 qf2, 2 vf, 3 policy, 6 target_vf; HOST flat arrays */
int ilsx_sacv_get_params(ilsx_sacv* sac, int which, float* dst_host, size_t n);
int ilsx_sacv_set_params(ilsx_sacv* sac, int which, const float* src_host, size_t n);

/* ---------------------------------------------------------------- optimiser state (snapshots / resume)
 * The optimizer.state_dict() halves of the reference's get_snapshot / load_snapshot pairs (td3.py:185-210, sac.py:245-270,
 * ppo.py:47-55 policy/value optimisers, adv_irl.py:75-77 disc_optimizer, bc.py:33-41) and what `load_params` resumes from
 * (run_scripts/sac_alpha_exp_script.py:106-108,142-146; rlkit/core/logger.py:31-49).  Adam's exp_avg / exp_avg_sq travel in
 * the flat ABI layout of the parameter block they belong to (HOST fp32 [n], n = that block's parameter count); meta.t is the
 * owning optimiser's step count, meta.rng_step the agent's Philox step counter, meta.n_train_steps the trainer's own step
 * counter where it has one (TD3's delayed-update parity, td3.py:101).  `which` numbers the TRAINABLE blocks exactly like the
 * agent's *_get_params.  SAC-alpha has its own pair (ilsx_sac_get_adam / _set_adam / _alpha_opt above). */
typedef struct { int64_t t; uint64_t rng_step; int64_t n_train_steps; } ilsx_opt_meta;
int ilsx_td3_get_opt(ilsx_td3* td3, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta);    /* 0 qf1, 1 qf2, 2 policy */
int ilsx_td3_set_opt(ilsx_td3* td3, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta);
int ilsx_sacv_get_opt(ilsx_sacv* sac, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta);  /* 0 qf1, 1 qf2, 2 vf, 3 policy */
int ilsx_sacv_set_opt(ilsx_sacv* sac, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta);
int ilsx_bc_get_opt(ilsx_bc* bc, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta);
int ilsx_bc_set_opt(ilsx_bc* bc, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta);
int ilsx_ppo_get_opt(ilsx_ppo* ppo, int which, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta);    /* 0 policy (| action_log_std), 1 value net */
int ilsx_ppo_set_opt(ilsx_ppo* ppo, int which, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta);
int ilsx_disc_get_opt(ilsx_disc* disc, float* m_host, float* v_host, size_t n, ilsx_opt_meta* meta);
int ilsx_disc_set_opt(ilsx_disc* disc, const float* m_host, const float* v_host, size_t n, const ilsx_opt_meta* meta);

/* ---------------------------------------------------------------- behaviour cloning
 * Replaces rlkit/torch/algorithms/bc/bc.py:14-41 (ctor: Adam(lr, betas=(momentum, 0.999)) over the policy) and :77-106
 * (_do_training / _do_update_step).  MLE: loss = -mean(policy.get_log_prob(obs, acts)); MSE: loss = mean over rows of
 * sum_j (policy(obs)[0] - acts)^2 with policy(obs)[0] a SAMPLED action, like the reference. */
enum { ILSX_BC_MLE = 0, ILSX_BC_MSE = 1 };
typedef struct { int32_t mode; float lr, momentum; int32_t max_batch; } ilsx_bc_cfg;
int ilsx_bc_create(ilsx_ctx* ctx, const ilsx_bc_cfg* cfg, ilsx_net* pi, ilsx_bc** out);   /* adopts pi like ilsx_sac_create */
int ilsx_bc_destroy(ilsx_bc* bc);
/* device rows obs[B,o], act[B,a]; eps (device [B,a], MSE mode's N(0,1) draws) or NULL = Philox; stat (host, nullable) =
 * "Log-Likelihood" (MLE) or "MSE" of this batch (bc.py:92-102) */
int ilsx_bc_train_step(ilsx_bc* bc, const float* obs, const float* act, int B, const float* eps, float* stat);
int ilsx_bc_train_from_replay(ilsx_bc* bc, ilsx_replay* expert_rb, int n_updates, int B, float* stat);

/* ---------------------------------------------------------------- PPO
 * Replaces rlkit/torch/algorithms/ppo/ppo.py:57-100 (calc_adv: per-trajectory GAE, zero bootstrap, per-trajectory
 * advantage standardisation) and :102-170 (train_step: update_epoch x shuffled minibatches of value MSE + L2 and
 * the clipped surrogate, grad-norm clip 20), with ReparamMultivariateGaussianPolicy (both conditioned_std settings)
 * (rlkit/torch/common/policies.py:348-478) and the tanh value net of run_scripts/ppo_exp_script.py:82-96.
 * cfg fields == the YAML keys of exp_specs/ppo/ppo_hopper.yaml:41-50 (+ net_size / num_hidden_layers :13-14). */
typedef struct {
  int32_t obs_dim, act_dim, n_hidden, hidden;
  float reward_scale, discount, clip_eps, policy_lr, value_lr, gae_tau, value_l2_reg;
  int32_t mini_batch_size, update_epoch;
  int32_t max_samples;            /* upper bound on the on-policy samples of one train call */
  int32_t use_value_clip;         /* ppo.py:24,137-143: value loss = mean(max((v-R)^2, (v_old + clamp(v-v_old, +-clip_eps) - R)^2)) */
  int32_t conditioned_std;        /* policies.py:354,368-374,401-405: log_std = clamp(last_fc_log_std(h), -20, 2), a second head of the policy net
                                   * (flat: fc.. | last_fc | last_fc_log_std) instead of the action_log_std parameter behind the mean net */
  int32_t hidden_sizes[3];        /* networks.py:23-60 takes any list of widths: logical width of hidden layer l (1..hidden; 0 = hidden), for the
                                   * policy and the value net alike (ppo_exp_script.py:79-96 builds both from one list) — ilsx_mlp_cfg::hidden_sizes */
  int32_t grad_world;             /* split-run mode (SURVEY section 8e "PPO split: same, per-minibatch grads"; 0 / 1 = off): this rank holds N / G of the
                                   * on-policy samples (its own envs' trajectories: GAE and the per-trajectory advantage standardisation stay local)
                                   * and mini_batch_size / G rows of every minibatch; mean-loss gradients are scaled 1 / (rows * G); the value arena
                                   * is all-reduced (sum) before its Adam (+ L2, applied once), the policy arena (with the action_log_std gradient)
                                   * before clip_grad_norm_ + Adam, on the communicator of the ctx (ilsx_comm_init).  Every rank must hold the
                                   * same number of samples and use the same minibatch size */
} ilsx_ppo_cfg;
int ilsx_ppo_create(ilsx_ctx* ctx, const ilsx_ppo_cfg* cfg, ilsx_ppo** out);
int ilsx_ppo_destroy(ilsx_ppo* ppo);
/* which: 0 = policy, flat = fc*.W|fc*.b|last_fc.W|last_fc.b|action_log_std[a] ; 1 = value net.  HOST arrays. */
int ilsx_ppo_num_params(const ilsx_ppo* ppo, int which, size_t* out);
int ilsx_ppo_set_params(ilsx_ppo* ppo, int which, const float* src_host, size_t n);
int ilsx_ppo_get_params(ilsx_ppo* ppo, int which, float* dst_host, size_t n);
/* calc_adv + the fixed log-probs over N = traj_offsets_host[n_traj] device rows obs[N,o] act[N,a] rew[N]; trajectory
 * t owns rows [traj_offsets_host[t], traj_offsets_host[t+1]).  bootstrap_values (device [n_traj], nullable): V of the
 * observation that follows each trajectory's last sample; NULL = 0 everywhere = the reference, which also zeroes it at
 * time-limit ends (ppo.py:74).  Device outputs [N], all nullable. */
int ilsx_ppo_gae(ilsx_ppo* ppo, const float* obs, const float* act, const float* rew, const int32_t* traj_offsets_host,
                 int n_traj, const float* bootstrap_values, float* returns, float* advantages, float* values,
                 float* fixed_log_probs);
/* values[n] = vf(obs[n,o]) (device) */
int ilsx_ppo_values(ilsx_ppo* ppo, const float* obs, int n, float* values);
/* one PPO.train_step.  perms_host [update_epoch][N] int32 row permutations (torch.randperm in the reference,
 * ppo.py:116) or NULL = drawn by the library. */
int ilsx_ppo_train(ilsx_ppo* ppo, const float* obs, const float* act, const float* rew, const int32_t* traj_offsets_host,
                   int n_traj, const float* bootstrap_values, const int32_t* perms_host);
/* debug: one library-drawn shuffle of [0,n) (keyed Feistel bijection; the NULL-perms path of ilsx_ppo_train) -> HOST */
int ilsx_ppo_debug_perm(ilsx_ppo* ppo, int n, uint32_t key, int32_t* perm_host);
/* test hook: the norm of the last minibatch's policy gradient as clip_grad_norm_ saw it (ppo.py:166; split run: of the all-reduced gradient) */
int ilsx_ppo_debug_grad_norm(ilsx_ppo* ppo, float* out);
/* get_actions (policies.py:392-417): act[n,a] = mean + exp(log_std)*eps (eps device [n,a] or NULL = Philox), or the
 * mean when deterministic; logp[n] nullable. */
int ilsx_ppo_policy_act(ilsx_ppo* ppo, const float* obs, int n, int deterministic, const float* eps, float* act,
                        float* logp);

/* ---------------------------------------------------------------- vectorised env stepper
 * Replaces the reference's vec-env path: rlkit/envs/vecenvs.py:158-257 (BaseVectorEnv.reset/step),
 * rlkit/envs/worker/subproc.py:59-113 (process + pipe per env), rlkit/envs/wrappers.py:342-352
 * (NormalizedBoxEnv action map + clip) and the MuJoCo step under gym's HopperEnv / Walker2dEnv, whose
 * reward / termination / reset rules are restated in rlkit/envs/mujoco/hopper.py:11-40, walker2d.py:11-36
 * (HalfCheetah: gym's HalfCheetahEnv — forward velocity - 0.1*|a|^2, no termination; rlkit/envs/envs_dict.py:7).
 * The dynamics model is a planar articulated-body engine (body 0: slide-x, slide-z, hinge; every other body
 * one hinge; capsule-vs-floor soft contacts + soft joint limits solved by PGS; RK4).  Model constants come
 * from the caller (ilswiss_amd/envs/models.py); physics parity with MuJoCo is UNPINNED. */
#define ILSX_ENV_MAX_BODY 8
#define ILSX_ENV_MAX_GEOM 8
enum { ILSX_TASK_HOPPER = 0, ILSX_TASK_WALKER2D = 1, ILSX_TASK_HALFCHEETAH = 2 /* never terminates */ };
typedef struct {
  int32_t task, n_body, n_geom, frame_skip, pgs_iters, pad0;
  int32_t parent[ILSX_ENV_MAX_BODY], limited[ILSX_ENV_MAX_BODY], geom_body[ILSX_ENV_MAX_GEOM];
  double anchor[ILSX_ENV_MAX_BODY][2];   /* hinge position in the parent's frame */
  double com[ILSX_ENV_MAX_BODY][2];      /* centre of mass in the body frame */
  double mass[ILSX_ENV_MAX_BODY], inertia[ILSX_ENV_MAX_BODY], jsign[ILSX_ENV_MAX_BODY];
  double armature[ILSX_ENV_MAX_BODY], damping[ILSX_ENV_MAX_BODY], range[ILSX_ENV_MAX_BODY][2], gear[ILSX_ENV_MAX_BODY];
  double geom_p1[ILSX_ENV_MAX_GEOM][2], geom_p2[ILSX_ENV_MAX_GEOM][2], geom_radius[ILSX_ENV_MAX_GEOM],
      geom_friction[ILSX_ENV_MAX_GEOM];  /* capsule end points (body frame), radius, sliding friction */
  double timestep, gravity, reset_noise, contact_margin;
  double contact_solref[2], contact_solimp[3], limit_solref[2], limit_solimp[3];
  double ctrl_cost, alive_bonus, z_min, z_max, ang_max, state_max;
  double init_qpos[ILSX_ENV_MAX_BODY + 2];
  double stiffness[ILSX_ENV_MAX_BODY];   /* joint spring towards 0 (half_cheetah.xml) */
  double reset_noise_vel_std;            /* > 0: reset qvel ~ N(0, std) instead of U(+-reset_noise) (HalfCheetahEnv.reset_model) */
  double qvel_clip;                      /* > 0: observations clip qvel to +-qvel_clip (Hopper / Walker2d: 10); <= 0: none */
  int32_t max_rows, pad1;                /* constraint rows per env the solver keeps (0 = 8 for 4 bodies, 12 otherwise; <= 16) */
} ilsx_planar_model;

int ilsx_vecenv_create(ilsx_ctx* ctx, const ilsx_planar_model* model, int n_env, uint64_t seed, ilsx_vecenv** out);
/* 3-D models — Ant-v2 and Humanoid-v2 (rlkit/envs/envs_dict.py:6,9; reward / termination / observation / reset rules of
 * rlkit/envs/mujoco/ant.py:11-43 and humanoid.py:24-73, Humanoid with the full 376-dim observation of gym's -v2: qpos[2:] | qvel |
 * cinert | cvel | qfrc_actuator | cfrc_ext).  Link 0 is the root body on a free joint (qpos = position + unit quaternion w x y z,
 * qvel = world linear + BODY-frame angular velocity); every other link hangs off its parent by one hinge (a MuJoCo body with k
 * hinges is a chain of k links, the first k-1 massless).  Constants come from the caller (ilswiss_amd/envs/models3d.py); the
 * engine is stated in oracle/spatial_env.py; physics parity with MuJoCo is UNPINNED.  The handle is an ordinary
 * ilsx_vecenv: reset / step / rollout / evaluation entry points are the ones below. */
#define ILSX_ENV3_MAX_LINK 20
#define ILSX_ENV3_MAX_CONTACT 32
#define ILSX_ENV3_MAX_BODY 16
enum { ILSX_TASK_ANT = 3, ILSX_TASK_HUMANOID = 4 };
typedef struct {
  int32_t task, n_link, n_act, n_contact, n_body, frame_skip, pgs_iters, max_rows;
  int32_t parent[ILSX_ENV3_MAX_LINK], limited[ILSX_ENV3_MAX_LINK], act_link[ILSX_ENV3_MAX_LINK];   /* act_link[k]: link whose hinge actuator k drives */
  int32_t contact_link[ILSX_ENV3_MAX_CONTACT], body_link[ILSX_ENV3_MAX_BODY];   /* body_link[b]: link carrying MuJoCo body b+1 (observation rows) */
  double anchor[ILSX_ENV3_MAX_LINK][3];   /* hinge anchor in the parent link's frame */
  double axis[ILSX_ENV3_MAX_LINK][3];     /* unit hinge axis, link frame */
  double quat0[ILSX_ENV3_MAX_LINK][4];    /* fixed rotation parent -> link at q = 0 (w x y z) */
  double com[ILSX_ENV3_MAX_LINK][3], mass[ILSX_ENV3_MAX_LINK], inertia[ILSX_ENV3_MAX_LINK][6];   /* about the COM, link frame: xx yy zz xy xz yz */
  double armature[ILSX_ENV3_MAX_LINK], damping[ILSX_ENV3_MAX_LINK], stiffness[ILSX_ENV3_MAX_LINK], range[ILSX_ENV3_MAX_LINK][2],
      gear[ILSX_ENV3_MAX_LINK];
  double contact_pos[ILSX_ENV3_MAX_CONTACT][3], contact_radius[ILSX_ENV3_MAX_CONTACT], contact_friction[ILSX_ENV3_MAX_CONTACT];
  double timestep, gravity, reset_noise, reset_noise_vel_std, contact_margin, ctrl_range;
  double contact_solref[2], contact_solimp[3], limit_solref[2], limit_solimp[3];
  double ctrl_cost, alive_bonus, vel_weight, z_min, z_max;
  double init_qpos[ILSX_ENV3_MAX_LINK + 6];
} ilsx_spatial_model;
int ilsx_vecenv_create_spatial(ilsx_ctx* ctx, const ilsx_spatial_model* model, int n_env, uint64_t seed, ilsx_vecenv** out);
/* Path mode of the fused rollout (ilsx_rollout_step with a replay ring): 0 (default) = every transition enters the ring when it
 * happens; 1 = the reference's order (rlkit/core/base_algorithm.py:509-519, simple_replay_buffer.py:78-132): an episode's samples are
 * staged in HBM and enter the ring contiguously when the episode ends (ended envs in ascending order), the trajectory is registered in
 * _traj_endpoints, unfinished episodes are not sampleable.  Costs one 4-byte-per-env read-back per rollout step. */
int ilsx_vecenv_set_path_mode(ilsx_vecenv* env, int on);
/* sizes of the simulator state rows of ilsx_vecenv_get_state / _set_state (planar: nq == nv; 3-D: nq == nv + 1) */
int ilsx_vecenv_state_dims(const ilsx_vecenv* env, int* nq, int* nv);
int ilsx_vecenv_destroy(ilsx_vecenv* env);
int ilsx_vecenv_dims(const ilsx_vecenv* env, int* obs_dim, int* act_dim, int* n_dof, int* n_env);
/* BaseVectorEnv.reset(id) (vecenvs.py:158-181): ids_host = NULL resets all; obs (device, nullable) [n_ids,o]. */
int ilsx_vecenv_reset(ilsx_vecenv* env, const int32_t* ids_host, int n_ids, float* obs);
/* BaseVectorEnv.step(action, id) (vecenvs.py:183-257), sync mode: act[n_ids,a] -> obs[n_ids,o], rew[n_ids],
 * done[n_ids] (all device; outputs nullable).  No auto-reset (the caller resets finished ids, like the reference). */
int ilsx_vecenv_step(ilsx_vecenv* env, const float* act, const int32_t* ids_host, int n_ids, float* obs, float* rew,
                     uint8_t* done);
/* simulator state, HOST float64 qpos [n_env, nq], qvel [n_env, nv] (tests / snapshots) */
int ilsx_vecenv_get_state(ilsx_vecenv* env, double* qpos_host, double* qvel_host);
int ilsx_vecenv_set_state(ilsx_vecenv* env, const double* qpos_host, const double* qvel_host);
int ilsx_vecenv_cur_obs(ilsx_vecenv* env, float** dev_ptr);  /* [n_env,o] current (normalised if norm_obs) observations (device) */
/* ScaledEnv / MinmaxEnv (rlkit/envs/wrappers.py:53-203): every observation the env hands out or records becomes
 * (raw - shift) / scale; shift / scale are HOST float64 [obs_dim], scale already including the wrappers' EPS
 * (std + EPS, or max - min + EPS).  Resets all envs. */
int ilsx_vecenv_set_obs_affine(ilsx_vecenv* env, const double* shift_host, const double* scale_host);
/* Batched terminal predicates of rlkit/envs/terminals.py:14-117 (TerminalFunc.is_terminal(obs, act, next_obs), the functions
 * MBPO's FakeEnv and model rollouts label transitions with): done[n] = f(next_obs[n,o]) on the device, NaN / inf
 * semantics and the Hopper upper-bound-only state check (:61) as in the reference. */
enum { ILSX_TERM_INVERTED_PENDULUM = 0, ILSX_TERM_INVERTED_DOUBLE_PENDULUM = 1, ILSX_TERM_HOPPER = 2, ILSX_TERM_WALKER2D = 3,
       ILSX_TERM_HALFCHEETAH = 4, ILSX_TERM_HUMANOID = 5, ILSX_TERM_ANT = 6 };
int ilsx_is_terminal(ilsx_ctx* ctx, int kind, const float* next_obs, int n, int obs_dim, uint8_t* done);
/* Running observation statistics of BaseVectorEnv (vecenvs.py:104-113,299-327; RunningMeanStd normalizer.py:128-152):
 * norm_obs: reset/step/cur_obs/rollouts return clip((obs-mean)/sqrt(var+eps), +-10); update_obs_rms: every batch of
 * observations returned by reset/step updates (mean, var, count) first.  Statistics are float64; get/set use HOST arrays
 * [obs_dim] (an eval env shares the training env's statistics by set after get, ppo_exp_script.py:68-75). */
int ilsx_vecenv_obs_norm(ilsx_vecenv* env, int norm_obs, int update_obs_rms);
int ilsx_vecenv_get_obs_rms(ilsx_vecenv* env, double* mean_host, double* var_host, double* count);
int ilsx_vecenv_set_obs_rms(ilsx_vecenv* env, const double* mean_host, const double* var_host, double count);
/* PPO's sampling phase for ALL envs on the device (torch_rl_algorithm.py:30-32 over base_algorithm.py:183-277): T vec
 * steps with ppo's Gaussian policy, auto-reset on done / max_path_length, into env-major device buffers (sample (env,t) at
 * row env*T+t): obs[n_env*T,o] as the policy saw it, act[n_env*T,a], rew[n_env*T], ends[n_env*T] (1 = the episode ended
 * after this sample); last_values[n_env] (nullable) = vf(observation after the last step). */
int ilsx_ppo_rollout(ilsx_ppo* ppo, ilsx_vecenv* env, int T, int max_path_length, float* obs, float* act, float* rew,
                     uint8_t* ends, float* last_values);
/* One iteration of BaseAlgorithm's sampling loop for ALL envs on the device (base_algorithm.py:183-277):
 * actions (policy, or env.action_space.sample() when random_actions) -> physics -> one transition record per
 * env written straight into the replay ring (nullable) -> auto-reset on done or max_path_length.  no_terminal: the
 * stored terminal flag is forced to 0 (base_algorithm.py:195-196,208-210; the adv-IRL configs). */
int ilsx_rollout_step(ilsx_vecenv* env, ilsx_net* pi, ilsx_replay* rb, int max_path_length, int random_actions,
                      int deterministic, int no_terminal);
/* Evaluation on the device — VecPathSampler.obtain_samples / rollout (samplers/vec_sampler.py:5-93,124-142): reset all envs,
 * then every env plays ONE episode with `pi` (ilsx_net: tanh-Gaussian or noise policy) or `ppo`'s Gaussian policy; envs that
 * end (terminal or max_path_length) are frozen, not reset.  stats_host[18] (nullable, float64) accumulates what
 * eval_util.get_generic_path_information (:15-80) needs: {paths, steps, returns sum/sumsq/max/min, lengths
 * sum/sumsq/max/min, rewards sum/sumsq/max/min, actions sum/sumsq/max/min}; reset_stats = 0 keeps accumulating over calls. */
#define ILSX_EVAL_NSTATS 18
int ilsx_eval_rollout(ilsx_vecenv* env, ilsx_net* pi, ilsx_ppo* ppo, int max_path_length, int deterministic, int reset_stats,
                      double* stats_host);
/* The same for K runs at once (one small eval env per run, the lock-step loop of co-resident seeds): every run rolls whole rounds until ITS
 * statistics hold >= num_steps steps (VecPathSampler.obtain_samples, vec_sampler.py:126-146); runs of one shape step as one launch per stage, and
 * a run leaves a round at the check at which its own ilsx_eval_rollout would, so its counters and statistics are those of evaluating it alone.
 * stats_host: [n_runs][18] doubles, ilsx_eval_rollout's layout.  ILSX_ERR_UNSUPPORTED: the runs cannot share launches (evaluate them one by one). */
int ilsx_eval_rollouts_lockstep(ilsx_vecenv* const* envs, ilsx_net* const* pis, int n_runs, int max_path_length, int deterministic,
                                int64_t num_steps, double* stats_host);
/* ilsx_rollout_step in two halves, for a host that advances several runs side by side, each on its own ctx / stream (the lock-step loop of
 * co-resident seeds): _begin only enqueues the step; _end does the part that needs the host — with path mode on it waits for the step and
 * inserts the episodes that ended in it, otherwise nothing.  begin(run 0..K-1) ; end(run 0..K-1) overlaps the K runs' launches on the GPU. */
int ilsx_rollout_step_begin(ilsx_vecenv* env, ilsx_net* pi, ilsx_replay* rb, int max_path_length, int random_actions,
                            int deterministic, int no_terminal);
int ilsx_rollout_step_end(ilsx_vecenv* env);
/* n_steps lock-step iterations of n_runs runs: per iteration begin(run 0..K-1) ; end(run 0..K-1).  Run k acts at random while its ring holds
 * fewer than min_steps_before_training[k] samples (base_algorithm.py:186-188).  One call per stretch between two train triggers. */
int ilsx_rollout_steps_lockstep(ilsx_vecenv* const* envs, ilsx_net* const* pis, ilsx_replay* const* rbs, int n_runs, int n_steps,
                                int max_path_length, const int64_t* min_steps_before_training, int deterministic, int no_terminal);
/* DAgger's sampling iteration (dagger/dagger.py:45-71): the envs are driven by `pi`, the action stored in the replay record is
 * `expert`'s action for the observation acted on. */
int ilsx_rollout_step_relabel(ilsx_vecenv* env, ilsx_net* pi, ilsx_net* expert, int expert_deterministic, ilsx_replay* rb,
                              int max_path_length, int no_terminal);
/* finished episodes and the sum of their returns since the last reset of the counters */
int ilsx_rollout_stats(ilsx_vecenv* env, double* episodes, double* return_sum, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ILSX_H */
