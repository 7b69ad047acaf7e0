// ilsx_core.hip — context, networks and kernel launch helpers of libilsx.so (gfx950 only).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <random>

#define ILSX_KERNEL_IMPL 1  // the shared __global__ kernels of kernels.h are emitted by this TU only
#include <mutex>
#include "host_common.h"

static thread_local std::string g_err = "";

void ilsx_set_err(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

static int kernels_init_once();
extern "C" int ilsx_abi_version(void) { return ILSX_ABI_VERSION; }
extern "C" const char* ilsx_last_error(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------ ctx
int ctx_alloc(ilsx_ctx* c, size_t bytes, void** out, bool zero) {
  if (bytes == 0) bytes = 16;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) ILSX_FAIL(ILSX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  if (zero) HIPCHK(hipMemsetAsync(p, 0, bytes, c->stream));
  c->allocs.push_back(p);
  *out = p;
  return ILSX_OK;
}
int ctx_free(ilsx_ctx* c, void* p) {
  if (!p) return ILSX_OK;
  auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
  if (it == c->allocs.end()) ILSX_FAIL(ILSX_ERR_ARG, "ctx_free: pointer not owned by this ctx");
  c->allocs.erase(it);
  HIPCHK(hipStreamSynchronize(c->stream));
  // row-range slabs of split weight-gradient launches are keyed by the gradient range they belong to (launch_bwd_dw): when the allocation
  // holding that range goes, so do they — hipMalloc hands the address to the next trainer, whose table (same span, another live / padding
  // layout) must not inherit this one's nonzero words, and a destroyed trainer's slabs should not stay allocated until the ctx goes
  hipDeviceptr_t base = nullptr; size_t size = 0;
  if (!c->dw_scratch.empty() && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess) {
    for (size_t i = 0; i < c->dw_scratch.size();) {
      const char* k = (const char*)c->dw_scratch[i].key;
      if (k >= (const char*)base && k < (const char*)base + size) {
        void* sp = c->dw_scratch[i].p;
        c->dw_scratch.erase(c->dw_scratch.begin() + i);
        if (sp) {
          auto js = std::find(c->allocs.begin(), c->allocs.end(), sp);
          if (js != c->allocs.end()) c->allocs.erase(js);
          HIPCHK(hipFree(sp));
        }
      } else ++i;
    }
  }
  HIPCHK(hipFree(p));
  return ILSX_OK;
}
int ctx_stage(ilsx_ctx* c, size_t bytes, void** out) {
  if (bytes > c->stage_bytes) {
    if (c->stage) ILSX_TRY(ctx_free(c, c->stage));
    c->stage = nullptr;
    size_t nb = std::max(bytes, (size_t)1 << 20);
    ILSX_TRY(ctx_alloc(c, nb, &c->stage, false));
    c->stage_bytes = nb;
  }
  *out = c->stage;
  return ILSX_OK;
}

extern "C" int ilsx_ctx_create(int hip_device, void* hip_stream, uint64_t seed, ilsx_ctx** out) {
  if (!out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ctx_create: out is NULL");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (hip_device < 0 || hip_device >= ndev)
    ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ctx_create: device %d out of range (%d visible)", hip_device, ndev);
  HIPCHK(hipSetDevice(hip_device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, hip_device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "libilsx is built for gfx950 (MI355X) only; device reports %s", prop.gcnArchName);
  ILSX_TRY(kernels_init_once());
  ilsx_ctx* c = new ilsx_ctx();
  c->device = hip_device;
  c->seed = seed;
  if (const char* e = getenv("ILSX_XCD_SHIFT")) c->xcd_shift = atoi(e);
  if (const char* e = getenv("ILSX_RT")) c->rt_single = std::max(1, atoi(e));
  if (const char* e = getenv("ILSX_RT_GROUPED")) c->rt_grouped = std::max(1, atoi(e));
  if (hip_stream) {
    c->stream = (hipStream_t)hip_stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      ILSX_FAIL(ILSX_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  *out = c;
  return ILSX_OK;
}
extern "C" int ilsx_ctx_sync(ilsx_ctx* c) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  return ILSX_OK;
}
extern "C" int ilsx_ctx_rng_stream_cursor(ilsx_ctx* c, uint32_t set_to, uint32_t* current) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  if (current) *current = c->next_rng_stream;
  if (set_to > 0) c->next_rng_stream = set_to;
  return ILSX_OK;
}
extern "C" int ilsx_ctx_destroy(ilsx_ctx* c) {
  if (!c) return ILSX_OK;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  ilsx_comm_destroy(c);
  for (auto& r : c->prof_pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (hipEvent_t e : c->prof_free) hipEventDestroy(e);
  for (int slot : c->phase_slots) phase_const_free(c->device, slot);
  c->phase_slots.clear();
  for (void* p : c->allocs) hipFree(p);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return ILSX_OK;
}
extern "C" void* ilsx_ctx_stream(ilsx_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int ilsx_ctx_alloc(ilsx_ctx* c, size_t bytes, void** out) {
  if (!c || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_ctx_alloc: NULL argument");
  HIPCHK(hipSetDevice(c->device));
  return ctx_alloc(c, bytes, out, true);
}
extern "C" int ilsx_ctx_free(ilsx_ctx* c, void* p) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  return ctx_free(c, p);
}
extern "C" int ilsx_memcpy_h2d(ilsx_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (!dst && bytes) || (!src && bytes)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_memcpy_h2d: NULL argument");
  // synchronous w.r.t. the host buffer, ordered on the ctx stream
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ILSX_OK;
}
extern "C" int ilsx_memcpy_d2h(ilsx_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!c || (!dst && bytes) || (!src && bytes)) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_memcpy_d2h: NULL argument");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return ILSX_OK;
}

// ------------------------------------------------------------------------------------------ profiling
static hipEvent_t prof_event(ilsx_ctx* c) {
  if (!c->prof_free.empty()) { hipEvent_t e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
ProfScope::ProfScope(ilsx_ctx* ctx, int k) : c(ctx), kid(k) {
  if (!c->prof_on) return;
  a = prof_event(c); b = prof_event(c);
}
ProfScope::~ProfScope() {
  if (!a) return;
  if (launched) c->prof_pending.push_back({kid, a, b});
  else { c->prof_free.push_back(a); c->prof_free.push_back(b); }
}
static int prof_collect(ilsx_ctx* c) {
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& r : c->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.kid] += ms; c->prof_n[r.kid] += 1; }
    c->prof_free.push_back(r.a); c->prof_free.push_back(r.b);
  }
  c->prof_pending.clear();
  return ILSX_OK;
}
extern "C" int ilsx_prof_enable(ilsx_ctx* c, int on) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  ILSX_TRY(prof_collect(c));
  c->prof_on = on != 0;
  return ILSX_OK;
}
extern "C" int ilsx_prof_reset(ilsx_ctx* c) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
  ILSX_TRY(prof_collect(c));
  for (int i = 0; i < ILSX_K_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_name[i] = nullptr; }
  return ILSX_OK;
}
extern "C" int ilsx_prof_read(ilsx_ctx* c, int kid, uint64_t* launches, double* total_ms) {
  if (!c || kid < 0 || kid >= ILSX_K_COUNT) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_prof_read: bad argument");
  ILSX_TRY(prof_collect(c));
  if (launches) *launches = c->prof_n[kid];
  if (total_ms) *total_ms = c->prof_ms[kid];
  return ILSX_OK;
}
extern "C" const char* ilsx_prof_kernel(ilsx_ctx* c, int kid) {
  if (!c || kid < 0 || kid >= ILSX_K_COUNT || !c->prof_name[kid]) return "";
  return c->prof_name[kid];
}
extern "C" const char* ilsx_kernel_name(int kid) {
  static const char* names[ILSX_K_COUNT] = {"k_mlp_fwd", "k_mlp_bwd_dx", "k_mlp_bwd_dw", "k_adam_polyak",
      "k_replay_sample", "k_replay_add", "k_replay_sample_many", "k_sac_stats", "k_sac_finish", "k_env_step",
      "k_policy_finish", "k_disc_bwd", "k_ppo_gae", "k_sac_phase_a", "k_sac_phase_c", ""};
  return (kid >= 0 && kid < ILSX_K_COUNT) ? names[kid] : "";
}

// ------------------------------------------------------------------------------------------ layout
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

int net_layout_build(const ilsx_mlp_cfg& cfg, NetLayout* L) {
  if (cfg.n_hidden < 1 || cfg.n_hidden > ILSX_MAX_HID)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "n_hidden=%d not in 1..%d", cfg.n_hidden, ILSX_MAX_HID);
  if (cfg.hidden != 64 && cfg.hidden != 128 && cfg.hidden != 256)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "hidden=%d: supported widths are 64, 128, 256", cfg.hidden);
  if (cfg.in_dim < 1 || cfg.in_dim > 1024) ILSX_FAIL(ILSX_ERR_ARG, "in_dim=%d out of range", cfg.in_dim);
  if (cfg.n_heads < 1 || cfg.n_heads > 2 || cfg.out_dim < 1 || cfg.n_heads * cfg.out_dim > ILSX_MAX_NO)
    ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "heads: n_heads=%d out_dim=%d (n_heads*out_dim must be <= %d)", cfg.n_heads,
              cfg.out_dim, ILSX_MAX_NO);
  if (cfg.act != ILSX_ACT_RELU && cfg.act != ILSX_ACT_TANH) ILSX_FAIL(ILSX_ERR_ARG, "act=%d unknown", cfg.act);
  L->cfg = cfg;
  for (int l = 0; l < ILSX_MAX_HID; ++l) {
    const int hs = l < cfg.n_hidden ? cfg.hidden_sizes[l] : 0;
    if (hs < 0 || hs > cfg.hidden) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "hidden_sizes[%d]=%d: 1..hidden=%d (0 = hidden)", l, hs, cfg.hidden);
    L->hl[l] = hs > 0 ? hs : (l < cfg.n_hidden ? cfg.hidden : 0);
    L->cfg.hidden_sizes[l] = l < cfg.n_hidden ? L->hl[l] : 0;   // canonical form: two layouts of the same network compare equal
  }
  L->KP = round_up(cfg.in_dim, 16);
  L->NO = cfg.n_heads * cfg.out_dim;
  const int H = cfg.hidden;
  size_t off = 0, nflat = 0;
  for (int l = 0; l < cfg.n_hidden; ++l) {
    L->ld[l] = l == 0 ? L->KP : H;
    L->off_W[l] = (int)off;
    off += (size_t)H * L->ld[l];
    L->off_Wb[l] = 0;
    if (l > 0) {  // second, backward-packed copy of every hidden->hidden matrix
      L->off_Wb[l] = (int)off;
      off += (size_t)H * H;
    }
    L->off_b[l] = (int)off;
    off += H;
    nflat += (size_t)L->out_of(l) * L->in_of(l) + L->out_of(l);
  }
  L->off_Wh = (int)off;
  off += (size_t)L->NO * H;
  L->off_bh = (int)off;
  off += round_up(L->NO, 4);
  nflat += (size_t)L->NO * L->out_of(cfg.n_hidden - 1) + L->NO;
  L->n_int = off;
  L->n_flat = nflat;
  return ILSX_OK;
}

void net_flat_to_internal(const NetLayout& L, const float* flat, float* in) {
  std::fill(in, in + L.n_int, 0.0f);
  const int H = L.cfg.hidden;
  size_t f = 0;
  for (int l = 0; l < L.cfg.n_hidden; ++l) {   // logical rows / columns only: everything else stays the structural zero of the fill above
    const int ind = L.in_of(l), outd = L.out_of(l);
    for (int n = 0; n < outd; ++n)
      for (int k = 0; k < ind; ++k) {
        const float v = flat[f + (size_t)n * ind + k];
        in[L.off_W[l] + pack_f(n, k, L.ld[l])] = v;
        if (l > 0) in[L.off_Wb[l] + pack_b(n, k, H)] = v;
      }
    f += (size_t)outd * ind;
    for (int n = 0; n < outd; ++n) in[L.off_b[l] + n] = flat[f + n];
    f += outd;
  }
  const int od = L.cfg.out_dim, hlast = L.out_of(L.cfg.n_hidden - 1);
  for (int h = 0; h < L.cfg.n_heads; ++h) {
    for (int j = 0; j < od; ++j)
      for (int k = 0; k < hlast; ++k) in[L.off_Wh + (size_t)(h * od + j) * H + k] = flat[f + (size_t)j * hlast + k];
    f += (size_t)od * hlast;
    for (int j = 0; j < od; ++j) in[L.off_bh + h * od + j] = flat[f + j];
    f += od;
  }
}

void net_internal_to_flat(const NetLayout& L, const float* in, float* flat) {
  const int H = L.cfg.hidden;
  size_t f = 0;
  for (int l = 0; l < L.cfg.n_hidden; ++l) {
    const int ind = L.in_of(l), outd = L.out_of(l);
    for (int n = 0; n < outd; ++n)
      for (int k = 0; k < ind; ++k) flat[f + (size_t)n * ind + k] = in[L.off_W[l] + pack_f(n, k, L.ld[l])];
    f += (size_t)outd * ind;
    for (int n = 0; n < outd; ++n) flat[f + n] = in[L.off_b[l] + n];
    f += outd;
  }
  const int od = L.cfg.out_dim, hlast = L.out_of(L.cfg.n_hidden - 1);
  for (int h = 0; h < L.cfg.n_heads; ++h) {
    for (int j = 0; j < od; ++j)
      for (int k = 0; k < hlast; ++k) flat[f + (size_t)j * hlast + k] = in[L.off_Wh + (size_t)(h * od + j) * H + k];
    f += (size_t)od * hlast;
    for (int j = 0; j < od; ++j) flat[f + j] = in[L.off_bh + h * od + j];
    f += od;
  }
}

NetView net_view(const NetLayout& L, float* base) {
  NetView v;
  v.base = base;
  for (int l = 0; l < ILSX_MAX_HID; ++l) {
    v.off_W[l] = l < L.cfg.n_hidden ? L.off_W[l] : 0;
    v.off_Wb[l] = l < L.cfg.n_hidden ? L.off_Wb[l] : 0;
    v.off_b[l] = l < L.cfg.n_hidden ? L.off_b[l] : 0;
    v.ld[l] = l < L.cfg.n_hidden ? L.ld[l] : 0;
  }
  v.off_Wh = L.off_Wh;
  v.off_bh = L.off_bh;
  v.nhid = L.cfg.n_hidden;
  v.H = L.cfg.hidden;
  v.in_dim = L.cfg.in_dim;
  v.KP = L.KP;
  v.NO = L.NO;
  return v;
}

// ------------------------------------------------------------------------------------------ launches
static size_t fwd_lds_bytes(int H, int KP) {
  return sizeof(float) * (16 * (KP + ILSX_LDS_PAD) + 2 * 16 * (H + ILSX_LDS_PAD) + 16 * ILSX_MAX_NO);
}
static size_t bwd_lds_bytes(int H) { return sizeof(float) * (2 * 16 * (H + ILSX_LDS_PAD) + 16 * ILSX_MAX_NO); }

// Raise the dynamic-LDS cap of every forward instantiation once (never inside a stream capture).
static int kernels_init_once() {
  static bool done = false;
  if (done) return ILSX_OK;
  const int cap = 160 * 1024;
#define SET_FWD(HH, AA) HIPCHK(hipFuncSetAttribute((const void*)k_mlp_fwd<HH, AA>, hipFuncAttributeMaxDynamicSharedMemorySize, cap))
  SET_FWD(64, ACT_RELU); SET_FWD(128, ACT_RELU); SET_FWD(256, ACT_RELU);
  SET_FWD(64, ACT_TANH); SET_FWD(128, ACT_TANH); SET_FWD(256, ACT_TANH);
#undef SET_FWD
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<false, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<true, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw_low<true, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<true, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<true, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<true, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<false, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp_bwd_dw<false, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
#define SET_G(HH, AA, CC, GG) HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<HH, AA, CC, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, cap))
  SET_G(256, ACT_RELU, 4, 0); SET_G(256, ACT_RELU, 4, 1); SET_G(256, ACT_RELU, 4, 2); SET_G(256, ACT_RELU, 4, 3); SET_G(256, ACT_TANH, 4, 0); SET_G(256, ACT_TANH, 4, 1); SET_G(256, ACT_TANH, 4, 2); SET_G(256, ACT_TANH, 4, 3);
#define SET_PH(AA, GG) HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<256, AA, 4, GG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<256, AA, 4, GG, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap))
  SET_PH(ACT_RELU, 0); SET_PH(ACT_RELU, 1); SET_PH(ACT_RELU, 2); SET_PH(ACT_TANH, 0); SET_PH(ACT_TANH, 1); SET_PH(ACT_TANH, 2);
#undef SET_PH
  SET_G(128, ACT_RELU, 2, 0); SET_G(128, ACT_RELU, 2, 1); SET_G(128, ACT_RELU, 2, 2); SET_G(128, ACT_RELU, 2, 3); SET_G(128, ACT_TANH, 2, 0); SET_G(128, ACT_TANH, 2, 1); SET_G(128, ACT_TANH, 2, 2); SET_G(128, ACT_TANH, 2, 3);
#undef SET_G
#define SET_MT(AA, MM) HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<256, AA, 4, true, 0, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<256, AA, 4, true, 1, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_fwd_split<256, AA, 4, true, 2, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_mlp2_bwd_split<256, AA, 4, true, MM>, hipFuncAttributeMaxDynamicSharedMemorySize, cap))
  SET_MT(ACT_RELU, 2); SET_MT(ACT_RELU, 4); SET_MT(ACT_TANH, 2); SET_MT(ACT_TANH, 4);
#undef SET_MT
#define SET_PHASE(HH, AA, CC) HIPCHK(hipFuncSetAttribute((const void*)k_sac_phase_a<HH, AA, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_sac_phase_a<HH, AA, CC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_sac_phase_c<HH, AA, CC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap)); \
  HIPCHK(hipFuncSetAttribute((const void*)k_sac_phase_c<HH, AA, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, cap))
  SET_PHASE(256, ACT_RELU, 4); SET_PHASE(256, ACT_TANH, 4); SET_PHASE(128, ACT_RELU, 2); SET_PHASE(128, ACT_TANH, 2);
#undef SET_PHASE
  done = true;
  return ILSX_OK;
}

#define DISPATCH_H_ACT(H, act, CALL)                                   \
  do {                                                                 \
    if (act == ILSX_ACT_RELU) {                                        \
      if (H == 64) { CALL(64, ACT_RELU); }                             \
      else if (H == 128) { CALL(128, ACT_RELU); }                      \
      else { CALL(256, ACT_RELU); }                                    \
    } else {                                                           \
      if (H == 64) { CALL(64, ACT_TANH); }                             \
      else if (H == 128) { CALL(128, ACT_TANH); }                      \
      else { CALL(256, ACT_TANH); }                                    \
    }                                                                  \
  } while (0)

extern "C" int ilsx_debug_set_stamp_buffer(ilsx_ctx* c, void* dev_trace, int max_launches, int* launches_so_far) {
  if (!c) ILSX_FAIL(ILSX_ERR_ARG, "ctx is NULL");
#ifndef ILSX_STAMPS
  if (dev_trace) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "this libilsx was built without stamps: make -C ilswiss_amd/csrc STAMPS=1 and load libilsx_stamps.so (ILSX_LIB)");
#endif
  if (launches_so_far) *launches_so_far = c->dbg_launches;
  c->dbg_stamps = (unsigned long long*)dev_trace;
  c->dbg_max_launches = dev_trace ? max_launches : 0;
  c->dbg_launches = 0;
  return ILSX_OK;
}

// ---- constant-memory descriptor tables of grouped launches (kernels.h g_fwd_tab / g_bwd_tab): slot allocation, first fit, per device
struct ConstSlots { std::vector<std::pair<int, int>> used; };   // (base, count), sorted by base
static ConstSlots g_slots[2][16];   // [forward | backward][device]
int grp_const_alloc(int device, bool fwd, int n, int* base) {
  if (device < 0 || device >= 16 || n < 1 || n > GRP_CONST_SLOTS) return -1;
  auto& u = g_slots[fwd ? 0 : 1][device].used;
  int at = 0;
  size_t i = 0;
  for (; i < u.size(); ++i) { if (u[i].first - at >= n) break; at = u[i].first + u[i].second; }
  if (at + n > GRP_CONST_SLOTS) return -1;
  u.insert(u.begin() + i, {at, n});
  *base = at;
  return 0;
}
void grp_const_free(int device, bool fwd, int base) {
  if (device < 0 || device >= 16) return;
  auto& u = g_slots[fwd ? 0 : 1][device].used;
  for (size_t i = 0; i < u.size(); ++i) if (u[i].first == base) { u.erase(u.begin() + i); return; }
}
int grp_const_upload(ilsx_ctx* ctx, bool fwd, int base, const void* host, size_t count) {   // synchronous w.r.t. the host table
  const size_t rec = fwd ? sizeof(FwdTaskG) : sizeof(BwdTask);
  if (fwd) HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_fwd_tab), host, count * rec, (size_t)base * rec, hipMemcpyHostToDevice, ctx->stream));
  else HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_bwd_tab), host, count * rec, (size_t)base * rec, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ILSX_OK;
}

int mlp2_split_factor(int n_hidden, int H) {
  if (n_hidden != 2) return 1;
  return H == 256 ? 4 : H == 128 ? 2 : 1;
}

static size_t fwd_split_lds_bytes(int H, int KP, int cs) {
  return sizeof(float) * (16 * (KP + ILSX_LDS_PAD) + 16 * (H + ILSX_LDS_PAD) + 16 * (H / cs + ILSX_LDS_PAD) +
                          4 * (4 * H / cs / 64) * 4 * 64);
}
static size_t bwd_split_lds_bytes(int H, int cs, int mt = 1) {
  return sizeof(float) * mt * (16 * (H + ILSX_LDS_PAD) + 16 * (H / cs + ILSX_LDS_PAD) + 16 * ILSX_MAX_NO);
}
// macro tiles (fwd_split_tile.inc, MT > 1): mt row tiles per workgroup; ph = the forward's phase (0 whole, 1 layer 0 only, 2 layer 1 + heads);
// not_max = the widest head of the launch in 16-output tiles, a_max = the widest finished policy (log-prob staging shares the partial-tile region)
static size_t fwd_split_lds_bytes_mt(int H, int KP, int cs, int mt, int ph, int not_max, int a_max) {
  const size_t mr = 16 * (size_t)mt, nwv = 4 * H / cs / 64;
  size_t fl = mr * (H / cs + ILSX_LDS_PAD);                       // hs
  if (ph != 2) fl += mr * (KP + ILSX_LDS_PAD);                    // xs
  if (ph != 1) fl += mr * (H + ILSX_LDS_PAD);                     // h0
  const size_t part = ph == 1 ? 0 : (size_t)mt * std::max(not_max, 1) * nwv * 256, lp3 = ph == 2 ? 0 : mr * std::max(a_max, 1) * 3;
  fl += std::max(part, lp3);
  return sizeof(float) * fl;
}
// the grouped macro-tile launches run on a 1-D grid (GrpSwizzle, kernels.h)
static int grp_swizzle_fill(GrpSwizzle* S, int ntasks, int agents, int rows, int mt) {
  if (agents < 1 || ntasks % agents) ILSX_FAIL(ILSX_ERR_ARG, "grouped launch: %d tasks do not divide over %d agents", ntasks, agents);
  S->agents = agents; S->tpa = ntasks / agents; S->tiles = (rows + 16 * mt - 1) / (16 * mt);
  S->np8 = (agents * S->tiles + 7) / 8;
  return ILSX_OK;
}

int launch_fwd(ilsx_ctx* ctx, const FwdArgs& A0, int H, int act, int KPmax, int cs) {
  if (A0.rows <= 0) return ILSX_OK;
  FwdArgs A = A0;
  A.dbg = ctx->dbg_next();
  ProfScope ps(ctx, ILSX_K_MLP_FWD);
  if (cs > 1 && A.tasks && A.mt > 1) {   // grouped launch on macro tiles: 1-D grid, (agent, tile) -> XCD fixed across the lock-step's launches
    if (!(H == 256 && cs == 4) || (A.mt != 2 && A.mt != 4)) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "macro-tile forward: H=%d cs=%d mt=%d", H, cs, A.mt);
    A.xs = 0; A.rt = 1; A.ctab = 0;
    ILSX_TRY(grp_swizzle_fill(&A.swz, A.ntasks, A.swz.agents, A.rows, A.mt));
    const unsigned nwork = 8u * A.swz.np8 * A.swz.tpa * cs;
    if (A.tail_mode && (!A.tail || A.tail_n < 1)) ILSX_FAIL(ILSX_ERR_ARG, "deferred tail: bad record table");
    const dim3 grid(nwork + (A.tail_mode ? A.tail_n : 0)), gridw(nwork), block(4 * H / cs);
#define MT_CALL(AA, PP, MM, GR, AR) do { const size_t lds = fwd_split_lds_bytes_mt(H, KPmax, cs, MM, PP, A.mt_not, A.mt_a); \
      if (lds > 160 * 1024) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "forward macro tile needs %zu B of LDS (> 160 KiB)", lds); \
      ILSX_LAUNCH(ps, (k_mlp2_fwd_split<256, AA, 4, true, PP, MM>), GR, block, lds, ctx->stream, AR); } while (0)
#define MT_CALL_PH(AA, MM) do { if (A.l0_split) { FwdArgs A2 = A; A2.tail_mode = 0; A2.tail = nullptr; A2.tail_n = 0; A2.dbg = ctx->dbg_next(); \
      MT_CALL(AA, 1, MM, grid, A); MT_CALL(AA, 2, MM, gridw, A2); } else { MT_CALL(AA, 0, MM, grid, A); } } while (0)
    if (act == ILSX_ACT_RELU) { if (A.mt == 2) MT_CALL_PH(ACT_RELU, 2); else MT_CALL_PH(ACT_RELU, 4); }
    else { if (A.mt == 2) MT_CALL_PH(ACT_TANH, 2); else MT_CALL_PH(ACT_TANH, 4); }
#undef MT_CALL_PH
#undef MT_CALL
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  if (cs > 1) {
    const size_t lds = fwd_split_lds_bytes(H, KPmax, cs);
    if (lds > 160 * 1024) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "forward tile needs %zu B of LDS (> 160 KiB)", lds);
    A.xs = ctx->xcd_shift;
    if (A.rt <= 0) A.rt = (A.tasks && ctx->rt_grouped > 0) ? ctx->rt_grouped : ctx->rt_single;
    const int tiles = (A.rows + 15) / 16;
    // XCD-aware tile mapping: workgroups are dealt to the 8 XCDs round-robin in linear order (x fastest), so with grid.x a
    // multiple of 8 row tile t runs on XCD t % 8 in EVERY column-split launch, whatever its task / slice count — the
    // activations a forward launch leaves in an XCD's L2 are read back by the backward launch's workgroups of the same
    // tile from that L2.  (Measured: one extra x column in the two forward launches costs the backward launches that
    // consume their activations 2.9 us each.)  Padding workgroups exit at once.
    dim3 grid(((((tiles + A.rt - 1) / A.rt) + 7) & ~7) << A.xs, A.ntasks, cs), block(4 * H / cs);
    if (A.tail_mode) { if (!A.tail || A.tail_n < 1 || A.tail_n > (int)(grid.x * grid.z)) ILSX_FAIL(ILSX_ERR_ARG, "deferred tail: bad record table"); grid.y += 1; }
    if (A.late < 0) A.late = (long)tiles * A.ntasks * cs > 2L * device_cus(ctx);   // more workgroups than two per CU: the 4-waves-per-SIMD shape (kernels.h GRP == 3)
    if (H == 256 && cs == 4 && A.l0_split) {   // wide inputs: layer 0 (column-split) in its own launch, then layer 1 + heads from hsave[0]
      FwdArgs A2 = A;
      A2.tail_mode = 0; A2.tail = nullptr; A2.tail_n = 0; A2.dbg = ctx->dbg_next();
      dim3 grid2(grid.x, A.ntasks, cs);
#define L0_CALL(AA, GG) do { ILSX_LAUNCH(ps, (k_mlp2_fwd_split<256, AA, 4, GG, 1>), grid, block, lds, ctx->stream, A); \
                             ILSX_LAUNCH(ps, (k_mlp2_fwd_split<256, AA, 4, GG, 2>), grid2, block, lds, ctx->stream, A2); } while (0)
      const int gg = A.tasks ? (A.ctab ? 2 : 1) : 0;
      if (act == ILSX_ACT_RELU) { if (gg == 2) L0_CALL(ACT_RELU, 2); else if (gg == 1) L0_CALL(ACT_RELU, 1); else L0_CALL(ACT_RELU, 0); }
      else { if (gg == 2) L0_CALL(ACT_TANH, 2); else if (gg == 1) L0_CALL(ACT_TANH, 1); else L0_CALL(ACT_TANH, 0); }
#undef L0_CALL
    } else if (H == 256 && cs == 4) {
#define FWD_GRP_CALL(HH, AA, CC) do { if (A.tasks && A.ctab && A.late) ILSX_LAUNCH(ps, (k_mlp2_fwd_split<HH, AA, CC, 3>), grid, block, lds, ctx->stream, A); \
        else if (A.tasks && A.ctab) ILSX_LAUNCH(ps, (k_mlp2_fwd_split<HH, AA, CC, 2>), grid, block, lds, ctx->stream, A); \
        else if (A.tasks) ILSX_LAUNCH(ps, (k_mlp2_fwd_split<HH, AA, CC, 1>), grid, block, lds, ctx->stream, A); \
        else ILSX_LAUNCH(ps, (k_mlp2_fwd_split<HH, AA, CC, 0>), grid, block, lds, ctx->stream, A); } while (0)
      if (act == ILSX_ACT_RELU) FWD_GRP_CALL(256, ACT_RELU, 4); else FWD_GRP_CALL(256, ACT_TANH, 4);
    } else if (H == 128 && cs == 2) {
      if (act == ILSX_ACT_RELU) FWD_GRP_CALL(128, ACT_RELU, 2); else FWD_GRP_CALL(128, ACT_TANH, 2);
#undef FWD_GRP_CALL
    } else {
      ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "no column-split forward kernel for H=%d cs=%d", H, cs);
    }
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  const size_t lds = fwd_lds_bytes(H, KPmax);
  if (lds > 160 * 1024) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "forward tile needs %zu B of LDS (> 160 KiB)", lds);
  dim3 grid((A.rows + 15) / 16, A.ntasks), block(4 * H);
#define CALL_FWD(HH, AA) ILSX_LAUNCH(ps, (k_mlp_fwd<HH, AA>), grid, block, lds, ctx->stream, A)
  DISPATCH_H_ACT(H, act, CALL_FWD);
#undef CALL_FWD
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// ---- merged phase kernels of the single-run SAC step (kernels.h: k_sac_phase_a / _c)
// Every workgroup of a phase launch must be resident at once (they wait for each other): at most one workgroup per CU.
int device_cus(ilsx_ctx* ctx) {
  static int n_cu = 0;
  if (!n_cu) {
    hipDeviceProp_t pr;
    n_cu = hipGetDeviceProperties(&pr, ctx->device) == hipSuccess ? pr.multiProcessorCount : 1;
  }
  return n_cu;
}
static size_t phase_lds_bytes(int H, int KP, int cs) { return std::max(fwd_split_lds_bytes(H, KP, cs), bwd_split_lds_bytes(H, cs)); }
// workgroups of a phase kernel one CU holds at once (registers, LDS): what the runtime's occupancy calculator says for the launch shape
void phase_wgs_per_cu(int H, int cs, int* occ_a, int* occ_c) {
  static int oa[2] = {-1, -1}, oc[2] = {-1, -1};
  const int i = H == 256 ? 0 : 1;
  if (oa[i] < 0) {
    int a = 0, c = 0;
    const size_t lds = phase_lds_bytes(H, 32, cs);
    hipError_t e1, e2;
    if (H == 256) { e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void*)k_sac_phase_a<256, ACT_RELU, 4>, 4 * H / cs, lds);
                    e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, (const void*)k_sac_phase_c<256, ACT_RELU, 4>, 4 * H / cs, lds); }
    else { e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void*)k_sac_phase_a<128, ACT_RELU, 2>, 4 * H / cs, lds);
           e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, (const void*)k_sac_phase_c<128, ACT_RELU, 2>, 4 * H / cs, lds); }
    oa[i] = e1 == hipSuccess ? std::max(a, 1) : 1;   // a launch that runs at all holds one workgroup per CU
    oc[i] = e2 == hipSuccess ? std::max(c, 1) : 1;
  }
  *occ_a = oa[i]; *occ_c = oc[i];
}
bool phase_fits(ilsx_ctx* ctx, int rows, int H, int cs, int ntasks) {
  const int n_cu = device_cus(ctx);
  if (!((H == 256 && cs == 4) || (H == 128 && cs == 2))) return false;
  const int tiles = (rows + 15) / 16, work = tiles * ntasks * cs;
  // (padding tiles and the idle part of the bookkeeping rows exit at once.)  Two conditions: the launch is the latency shape the phase kernels
  // are built for — at most one working workgroup per CU — and EVERY workgroup another one waits for is resident at once by the device's own
  // occupancy figure, not by assumption: phase A = its four task rows plus the bookkeeping row's tail workgroup, whose flag the critics'
  // backward waits for (+1); phase C = its three working rows (nobody waits for its bookkeeping row, which may run after them)
  int occ_a = 1, occ_c = 1;
  phase_wgs_per_cu(H, cs, &occ_a, &occ_c);
  return tiles <= PHASE_MAX_TILES && work <= n_cu && work + 1 <= occ_a * n_cu && tiles * 3 * cs <= occ_c * n_cu;
}

// ---- the phase kernels' descriptor blocks in constant memory (kernels.h g_phase_a_tab / g_phase_c_tab)
static std::vector<bool> g_phase_slots[16];   // [device][slot] in use
static std::mutex g_phase_slots_mu;           // agents are built and destroyed from more than one host thread (grouped runs, evaluation threads)
int phase_const_alloc(int device) {
  static const bool off = []() { const char* e = getenv("ILSX_PHASE_CT"); return e && atoi(e) == 0; }();
  if (off || device < 0 || device >= 16) return -1;
  std::lock_guard<std::mutex> lock(g_phase_slots_mu);
  auto& u = g_phase_slots[device];
  if (u.empty()) u.assign(PHASE_CONST_SLOTS, false);
  for (int i = 0; i < PHASE_CONST_SLOTS; ++i) if (!u[i]) { u[i] = true; return i; }
  return -1;
}
void phase_const_free(int device, int slot) {
  if (device < 0 || device >= 16 || slot < 0 || slot >= PHASE_CONST_SLOTS) return;
  std::lock_guard<std::mutex> lock(g_phase_slots_mu);
  if (!g_phase_slots[device].empty()) g_phase_slots[device][slot] = false;
}
void phase_const_prepare(ilsx_ctx* ctx, PhaseConst* ct) {
  if (!ct || ct->tried) return;
  ct->tried = true;
  ct->device = ctx->device;
  ct->slot = phase_const_alloc(ctx->device);
  if (ct->slot >= 0) ctx->phase_slots.push_back(ct->slot);
}
// Does the slot hold `blk`?  A block that differs from the host's copy is uploaded when `stream` is not capturing (synchronised first: no launch
// that reads the slot may be in flight; then the copy itself is waited for).  During a capture nothing can be uploaded — a copy would become a
// node of the graph, replayed with every step, and side-stream work invalidates a thread-local capture on this runtime — so the owner uploads
// BEFORE it begins the capture (ilsx_sac.hip sac_phase_const_prime builds the blocks in a dry pass); a block that still differs travels as arguments.
template <class Blk>
static bool phase_const_ready(ilsx_ctx* ctx, PhaseConst* ct, const Blk& blk, unsigned long long key, Blk& held, bool& valid, unsigned long long& held_key,
                              const void* symbol) {
  phase_const_prepare(ctx, ct);
  if (!ct || ct->slot < 0 || blk.dbg || !key) return false;   // (measurement builds stamp through a per-launch pointer: arguments)
  if (valid && held_key == key) return true;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
  valid = false;
  memcpy(&held, &blk, sizeof blk);
  if (hipMemcpyToSymbolAsync(symbol, &held, sizeof blk, (size_t)ct->slot * sizeof blk, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
  valid = true; held_key = key;
  return true;
}

int launch_phase_a(ilsx_ctx* ctx, const PhaseAArgs& P0, int H, int act, int KPmax, int cs, PhaseConst* ct, unsigned long long key, bool upload_only) {
  PhaseAArgs P = P0;
  P.f1.xs = P.f2.xs = P.b1.xs = 0; P.f1.rt = P.f2.rt = 1;
  P.f1.dbg = P.f2.dbg = nullptr; P.b1.dbg = nullptr;
  P.dbg = upload_only ? nullptr : ctx->dbg_next();   // (a dry pass takes no slab of the trace buffer)
  if (P.b1.ga_parts < 1) P.b1.ga_parts = 1;
  const int tiles = (P.f1.rows + 15) / 16;
  dim3 grid((tiles + 7) & ~7, 5, cs), block(4 * H / cs);
  const size_t lds = phase_lds_bytes(H, KPmax, cs);
  const bool use_ct = ct && phase_const_ready(ctx, ct, P, key, ct->a, ct->valid_a, ct->key_a, HIP_SYMBOL(g_phase_a_tab));
  if (upload_only) return ILSX_OK;
  const int slot = use_ct ? ct->slot : 0;
  static const bool trace = getenv("ILSX_PHASE_CT_TRACE") != nullptr;
  if (trace) { hipStreamCaptureStatus cap = hipStreamCaptureStatusNone; hipStreamIsCapturing(ctx->stream, &cap);
               fprintf(stderr, "launch_phase_a: %s, slot %d, capturing %d\n", use_ct ? "constant table" : "kernel arguments", ct ? ct->slot : -9, (int)cap); }
  ProfScope ps(ctx, ILSX_K_SAC_PHASE_A);
#define PHASE_A(HH, AA, CC) do { if (use_ct) ILSX_LAUNCH(ps, (k_sac_phase_a<HH, AA, CC, true>), grid, block, lds, ctx->stream, P, slot); \
                                 else ILSX_LAUNCH(ps, (k_sac_phase_a<HH, AA, CC>), grid, block, lds, ctx->stream, P, slot); } while (0)
  if (H == 256 && act == ILSX_ACT_RELU) PHASE_A(256, ACT_RELU, 4);
  else if (H == 256) PHASE_A(256, ACT_TANH, 4);
  else if (act == ILSX_ACT_RELU) PHASE_A(128, ACT_RELU, 2);
  else PHASE_A(128, ACT_TANH, 2);
#undef PHASE_A
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int launch_phase_c(ilsx_ctx* ctx, const PhaseCArgs& P0, int H, int act, int KPmax, int cs, PhaseConst* ct, unsigned long long key, bool upload_only) {
  PhaseCArgs P = P0;
  P.f3.xs = P.b2.xs = P.b3.xs = 0; P.f3.rt = 1;
  P.f3.dbg = nullptr; P.b2.dbg = P.b3.dbg = nullptr;
  P.dbg = upload_only ? nullptr : ctx->dbg_next();
  if (P.b2.ga_parts < 1) P.b2.ga_parts = 1;
  if (P.b3.ga_parts < 1) P.b3.ga_parts = 1;
  const int tiles = (P.f3.rows + 15) / 16;
  dim3 grid((tiles + 7) & ~7, 4, cs), block(4 * H / cs);   // y: Q1, Q2, the policy's backward, the bookkeeping row
  const size_t lds = phase_lds_bytes(H, KPmax, cs);
  const bool use_ct = ct && phase_const_ready(ctx, ct, P, key, ct->c, ct->valid_c, ct->key_c, HIP_SYMBOL(g_phase_c_tab));
  if (upload_only) return ILSX_OK;
  const int slot = use_ct ? ct->slot : 0;
  ProfScope ps(ctx, ILSX_K_SAC_PHASE_C);
#define PHASE_C(HH, AA, CC) do { if (use_ct) ILSX_LAUNCH(ps, (k_sac_phase_c<HH, AA, CC, true>), grid, block, lds, ctx->stream, P, slot); \
                                 else ILSX_LAUNCH(ps, (k_sac_phase_c<HH, AA, CC>), grid, block, lds, ctx->stream, P, slot); } while (0)
  if (H == 256 && act == ILSX_ACT_RELU) PHASE_C(256, ACT_RELU, 4);
  else if (H == 256) PHASE_C(256, ACT_TANH, 4);
  else if (act == ILSX_ACT_RELU) PHASE_C(128, ACT_RELU, 2);
  else PHASE_C(128, ACT_TANH, 2);
#undef PHASE_C
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int launch_bwd_dx(ilsx_ctx* ctx, const BwdArgs& A0, int H, int act, int cs) {
  if (A0.rows <= 0) return ILSX_OK;
  BwdArgs A = A0;
  A.dbg = ctx->dbg_next();
  if (A.ga_parts < 1) A.ga_parts = 1;
  ProfScope ps(ctx, ILSX_K_MLP_BWD_DX);
  if (cs > 1 && A.tasks && A.mt > 1) {   // grouped launch on macro tiles (see launch_fwd)
    if (!(H == 256 && cs == 4) || (A.mt != 2 && A.mt != 4)) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "macro-tile backward: H=%d cs=%d mt=%d", H, cs, A.mt);
    A.xs = 0; A.ctab = 0;
    ILSX_TRY(grp_swizzle_fill(&A.swz, A.ntasks, A.swz.agents, A.rows, A.mt));
    const dim3 grid(8u * A.swz.np8 * A.swz.tpa * cs), block(4 * H / cs);
    const size_t lds = bwd_split_lds_bytes(H, cs, A.mt);
    if (lds > 160 * 1024) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "backward macro tile needs %zu B of LDS (> 160 KiB)", lds);
    if (act == ILSX_ACT_RELU) { if (A.mt == 2) ILSX_LAUNCH(ps, (k_mlp2_bwd_split<256, ACT_RELU, 4, true, 2>), grid, block, lds, ctx->stream, A); else ILSX_LAUNCH(ps, (k_mlp2_bwd_split<256, ACT_RELU, 4, true, 4>), grid, block, lds, ctx->stream, A); }
    else { if (A.mt == 2) ILSX_LAUNCH(ps, (k_mlp2_bwd_split<256, ACT_TANH, 4, true, 2>), grid, block, lds, ctx->stream, A); else ILSX_LAUNCH(ps, (k_mlp2_bwd_split<256, ACT_TANH, 4, true, 4>), grid, block, lds, ctx->stream, A); }
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  if (cs > 1) {
    const size_t lds = bwd_split_lds_bytes(H, cs);
    A.xs = ctx->xcd_shift;
    dim3 grid(((((A.rows + 15) / 16) + 7) & ~7) << A.xs, A.ntasks, cs), block(4 * H / cs);   // same tile -> XCD mapping as launch_fwd
    if (A.late < 0) A.late = (long)((A.rows + 15) / 16) * A.ntasks * cs > 2L * device_cus(ctx);   // as in launch_fwd
    if (H == 256 && cs == 4) {
#define BWD_GRP_CALL(HH, AA, CC) do { if (A.tasks && A.ctab && A.late) ILSX_LAUNCH(ps, (k_mlp2_bwd_split<HH, AA, CC, 3>), grid, block, lds, ctx->stream, A); \
        else if (A.tasks && A.ctab) ILSX_LAUNCH(ps, (k_mlp2_bwd_split<HH, AA, CC, 2>), grid, block, lds, ctx->stream, A); \
        else if (A.tasks) ILSX_LAUNCH(ps, (k_mlp2_bwd_split<HH, AA, CC, 1>), grid, block, lds, ctx->stream, A); \
        else ILSX_LAUNCH(ps, (k_mlp2_bwd_split<HH, AA, CC, 0>), grid, block, lds, ctx->stream, A); } while (0)
      if (act == ILSX_ACT_RELU) BWD_GRP_CALL(256, ACT_RELU, 4); else BWD_GRP_CALL(256, ACT_TANH, 4);
    } else if (H == 128 && cs == 2) {
      if (act == ILSX_ACT_RELU) BWD_GRP_CALL(128, ACT_RELU, 2); else BWD_GRP_CALL(128, ACT_TANH, 2);
#undef BWD_GRP_CALL
    } else {
      ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "no column-split backward kernel for H=%d cs=%d", H, cs);
    }
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  const size_t lds = bwd_lds_bytes(H);
  dim3 grid((A.rows + 15) / 16, A.ntasks), block(4 * H);
#define CALL_BWD(HH, AA) ILSX_LAUNCH(ps, (k_mlp_bwd_dx<HH, AA>), grid, block, lds, ctx->stream, A)
  DISPATCH_H_ACT(H, act, CALL_BWD);
#undef CALL_BWD
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int launch_policy_finish(ilsx_ctx* ctx, const PolicyFinishArgs& P) {
  if (P.rows <= 0) return ILSX_OK;
  ProfScope ps(ctx, ILSX_K_POLICY_FINISH);
  ILSX_LAUNCH(ps, k_policy_finish, dim3((P.rows + 63) / 64), dim3(64), 0, ctx->stream, P);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

// Row ranges of a large-batch weight-gradient launch: `splits` ranges of `rows_per_split` rows (a whole number of the kernel's row steps:
// 128 for the tile kernel, DWB_RC for k_dw_big), none of them empty, together covering [0, rows).
void dw_split_plan(int rows, bool big, int* splits, int* rows_per_split) {
  const int unit = big ? DWB_RC : 128, cap = big ? 64 : 32;
  int s = rows / 512;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  const int rps = ((rows + s - 1) / s + unit - 1) / unit * unit;
  *rows_per_split = rps;
  *splits = (rows + rps - 1) / rps;   // rounding a range up to whole steps can leave trailing ranges empty: they are dropped
}
extern "C" int ilsx_debug_dw_split(int rows, int big, int* splits, int* rows_per_split) {
  if (rows < 1 || !splits || !rows_per_split) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_debug_dw_split: bad argument");
  dw_split_plan(rows, big != 0, splits, rows_per_split);
  return ILSX_OK;
}

int launch_bwd_dw(ilsx_ctx* ctx, const DwArgs& table, int rows, const AdamFuse* fuse) {
  if (table.ntiles <= 0 || rows <= 0) return ILSX_OK;
  DwArgs D = table;
  D.rows_all = rows;
  memset(&D.F, 0, sizeof D.F);
  if (fuse) D.F = *fuse;
  D.dbg = ctx->dbg_next();
  D.xs = ctx->xcd_shift;
  D.splits = 1; D.rows_per_split = rows; D.scratch = nullptr; D.span = 0;
  bool stacked = false;
  for (int i = 0; i < D.nmat && !D.gtiles; ++i) stacked = stacked || D.m[i].rows > 0;
  if (rows >= DW_SPLIT_MIN_ROWS && D.g_lo && !stacked) {   // large batch: split the contraction over row ranges
    // >= DW_BIG_MIN_ROWS rows: the LDS-staged 128 x 128 block kernel (k_dw_big), row ranges of >= 512 rows, up to 64 of them (its heavy
    // blocks — the hidden -> hidden matrix — are 4 per range: 256 of them fill the chip).  ILSX_DW_BIG = 0 keeps the tile kernel (A/B, tests)
    static const int big_env = []() { const char* e = getenv("ILSX_DW_BIG"); return e ? atoi(e) : 1; }();
    const bool big = big_env && rows >= DW_BIG_MIN_ROWS;
    int splits, rps;
    dw_split_plan(rows, big, &splits, &rps);
    const size_t span = (size_t)(D.g_hi - D.g_lo);
    const size_t need = (size_t)splits * span * sizeof(float);
    ilsx_ctx::DwScratch* reg = nullptr;
    for (auto& r : ctx->dw_scratch) if (r.key == D.g_lo && r.span == span) reg = &r;
    if (!reg) { ctx->dw_scratch.push_back({D.g_lo, span, nullptr, 0}); reg = &ctx->dw_scratch.back(); }
    if (reg->bytes < need) {
      if (reg->p) ILSX_TRY(ctx_free(ctx, reg->p));
      reg->p = nullptr; reg->bytes = 0;
      ILSX_TRY(ctx_alloc(ctx, need, &reg->p, true));   // zeroed: padding words are never written, and no other table writes this region
      reg->bytes = need;
    }
    D.splits = splits; D.rows_per_split = rps;
    D.scratch = (float*)reg->p; D.span = span; D.xs = 0;
    const AdamFuse keep = D.F;
    D.F.on = 0;
    if (big) {
      D.ntiles = 0;
      for (int i = 0; i < D.nmat; ++i) {   // the table again in 128 x 128 blocks
        D.m[i].ktiles = (D.m[i].NB + 127) / 128;
        D.m[i].tile0 = D.ntiles;
        D.ntiles += ((D.m[i].NA + 127) / 128) * D.m[i].ktiles;
      }
      ProfScope ps(ctx, ILSX_K_MLP_BWD_DW);
      ILSX_LAUNCH(ps, k_dw_big, dim3(D.ntiles * splits), dim3(256), 0, ctx->stream, D);
    } else {
      ProfScope ps(ctx, ILSX_K_MLP_BWD_DW);
      ILSX_LAUNCH(ps, (k_mlp_bwd_dw<false, 2, 4>), dim3(D.ntiles, splits), dim3(1024), DW_LDS_BYTES_OF(2, 4), ctx->stream, D);
    }
    DwReduceArgs R;
    R.scratch = D.scratch; R.splits = splits; R.span = span; R.g_lo = D.g_lo; R.F = keep;
    R.off0 = keep.on ? (size_t)(D.g_lo - keep.Gbase) : 0;
    int blocks = (int)((span / 4 + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    ProfScope ps(ctx, ILSX_K_ADAM);
    ILSX_LAUNCH(ps, k_dw_reduce, dim3(blocks), dim3(256), 0, ctx->stream, R);
    HIPCHK(hipGetLastError());
    return ILSX_OK;
  }
  ProfScope ps(ctx, ILSX_K_MLP_BWD_DW);
  if (D.gtiles && D.strip) {
    bool whole = (rows & 255) == 0;   // whole 256-row trips, and every matrix contracts over the batch rows (grouped tables hold no row-stacked jobs: group_build)
    if (whole) ILSX_LAUNCH(ps, (k_dw_strip<true, true>), dim3(D.ntiles), dim3(64), 0, ctx->stream, D);
    else ILSX_LAUNCH(ps, (k_dw_strip<true, false>), dim3(D.ntiles), dim3(64), 0, ctx->stream, D);
  } else if (D.gtiles) {
    const int gnh = D.tile_nh ? D.tile_nh : 2, gkt = D.tile_kt ? D.tile_kt : 4;
    // two 16-wave workgroups per CU (the <= 64-register instance) once the launch has more tiles than CUs: ILSX_DW_GRP_LOW = 0 / 1 pins it
    const char* low_e = getenv("ILSX_DW_GRP_LOW");   // read per launch: tests switch it
    const int low_env = low_e ? atoi(low_e) : -1;
    const bool low = low_env >= 0 ? low_env != 0 : D.ntiles > device_cus(ctx);
    if (gnh == 2 && gkt == 4 && low) ILSX_LAUNCH(ps, (k_mlp_bwd_dw_low<true, 2, 4>), dim3(D.ntiles << D.xs), dim3(1024), DW_LDS_BYTES_OF(2, 4), ctx->stream, D);
    else if (gnh == 2 && gkt == 4) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<true, 2, 4>), dim3(D.ntiles << D.xs), dim3(1024), DW_LDS_BYTES_OF(2, 4), ctx->stream, D);
    else if (gnh == 1 && gkt == 2) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<true, 1, 2>), dim3(D.ntiles << D.xs), dim3(512), DW_LDS_BYTES_OF(1, 2), ctx->stream, D);
    else if (gnh == 1 && gkt == 1) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<true, 1, 1>), dim3(D.ntiles << D.xs), dim3(512), DW_LDS_BYTES_OF(1, 1), ctx->stream, D);
    else if (gnh == 1 && gkt == 4) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<true, 1, 4>), dim3(D.ntiles << D.xs), dim3(512), DW_LDS_BYTES_OF(1, 4), ctx->stream, D);
    else ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "grouped dW tile %d x %d", gnh, gkt);
  } else {
    // small batches (one 256-row trip per wave): the table's 32 x 64 tiles are a few dozen 16-wave workgroups, each CU's 4 waves per
    // SIMD then share one MFMA pipe and one issue port while most of the chip idles; smaller tiles give the same waves (same
    // arithmetic, same order) to more CUs
    const char* shape_env = getenv("ILSX_DW_TILE");   // "NH KT" digits (24, 12, 11); unset / 0 = by size.  Read per launch: tests switch it
    const int shape = shape_env ? atoi(shape_env) : 0;
    int nh = 2, kt = 4;
    auto retile = [&](int nh_, int kt_) {
      D.ntiles = 0;
      for (int i = 0; i < D.nmat; ++i) {
        D.m[i].ktiles = (D.m[i].NB + 16 * kt_ - 1) / (16 * kt_);
        D.m[i].tile0 = D.ntiles;
        D.ntiles += ((D.m[i].NA + 16 * nh_ - 1) / (16 * nh_)) * D.m[i].ktiles;
      }
      nh = nh_; kt = kt_;
    };
    {   // (unsplit launches only get here: one to a few 256-row trips per wave, or row-stacked jobs)
      if (shape) {
        retile(shape / 10, shape % 10);
      } else {   // the smallest tiles while the launch stays within ~2-3 workgroups per CU (measured on the SAC step: 16 x 16 tiles at
                 // 576 workgroups beat 16 x 32 at 304; both beat 32 x 64 at 88)
        const int n_cu = device_cus(ctx);
        retile(1, 1);
        if (D.ntiles > 3 * n_cu) retile(1, 2);
        if (D.ntiles > 3 * n_cu) retile(2, 4);
      }
    }
    if (nh == 2 && kt == 4) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<false, 2, 4>), dim3(D.ntiles << D.xs), dim3(1024), DW_LDS_BYTES_OF(2, 4), ctx->stream, D);
    else if (nh == 1 && kt == 2) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<false, 1, 2>), dim3(D.ntiles << D.xs), dim3(512), DW_LDS_BYTES_OF(1, 2), ctx->stream, D);
    else if (nh == 1 && kt == 1) ILSX_LAUNCH(ps, (k_mlp_bwd_dw<false, 1, 1>), dim3(D.ntiles << D.xs), dim3(512), DW_LDS_BYTES_OF(1, 1), ctx->stream, D);
    else ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "ILSX_DW_TILE=%d: use 24, 12 or 11", shape);
  }
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int dw_table_add(DwArgs* T, const float* A, int lda, int NA, const float* Bm, int ldb, int NB, float* dW, float* dWb, int ldw,
                 float* db, int mode, int rows, int bias_rows) {
  if (T->nmat >= DW_MAX_MATS) ILSX_FAIL(ILSX_ERR_UNSUPPORTED, "dW table full (%d matrices)", DW_MAX_MATS);
  DwMat& m = T->m[T->nmat++];
  m.A = A; m.Bm = Bm; m.dW = dW; m.dWb = dWb; m.db = db;
  m.lda = lda; m.NA = NA; m.ldb = ldb; m.NB = NB; m.ldw = ldw; m.mode = mode; m.rows = rows; m.bias_rows = bias_rows;
  m.ktiles = (NB + DW_TILE_K - 1) / DW_TILE_K;
  m.tile0 = T->ntiles;
  T->ntiles += ((NA + DW_TILE_N - 1) / DW_TILE_N) * m.ktiles;
  return ILSX_OK;
}

int launch_adam(ilsx_ctx* ctx, const AdamArgs& A) {
  const int n4 = A.n / 4;
  int blocks = (n4 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  ProfScope ps(ctx, ILSX_K_ADAM);
  ILSX_LAUNCH(ps, k_adam_polyak, dim3(blocks), dim3(256), 0, ctx->stream, A);
  HIPCHK(hipGetLastError());
  return ILSX_OK;
}

int build_dw_jobs(const NetLayout& L, float* gbase, const float* xsave, float* const* hsave,
                  float* const* dsave, const float* dhead, DwArgs* table) {
  const int H = L.cfg.hidden;
  if (!table->g_lo || gbase < table->g_lo) table->g_lo = gbase;
  if (!table->g_hi || gbase + L.n_int > table->g_hi) table->g_hi = gbase + L.n_int;
  for (int l = 0; l < L.cfg.n_hidden; ++l) {
    const float* Bm = l == 0 ? xsave : hsave[l - 1];
    const int ldb = l == 0 ? L.KP : H;
    ILSX_TRY(dw_table_add(table, dsave[l], H, H, Bm, ldb, ldb, gbase + L.off_W[l], l > 0 ? gbase + L.off_Wb[l] : nullptr,
                          L.ld[l], gbase + L.off_b[l], l > 0 ? DW_OUT_PACK_FB : DW_OUT_PACK_F));
  }
  return dw_table_add(table, dhead, L.NO, L.NO, hsave[L.cfg.n_hidden - 1], H, H, gbase + L.off_Wh, nullptr, H,
                      gbase + L.off_bh, DW_OUT_NATURAL);
}

// ------------------------------------------------------------------------------------------ nets
extern "C" int ilsx_net_create(ilsx_ctx* ctx, const ilsx_mlp_cfg* cfg, ilsx_net** out) {
  if (!ctx || !cfg || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_create: NULL argument");
  HIPCHK(hipSetDevice(ctx->device));
  ilsx_net* n = new ilsx_net();
  n->ctx = ctx;
  int rc = net_layout_build(*cfg, &n->lay);
  if (rc != ILSX_OK) { delete n; return rc; }
  rc = ctx_alloc(ctx, n->lay.n_int * sizeof(float), (void**)&n->base, true);
  if (rc != ILSX_OK) { delete n; return rc; }
  *out = n;
  return ILSX_OK;
}
extern "C" int ilsx_net_destroy(ilsx_net* n) {
  if (!n) return ILSX_OK;
  if (n->owns && n->base) ctx_free(n->ctx, n->base);
  if (n->ws_out) ctx_free(n->ctx, n->ws_out);
  delete n;
  return ILSX_OK;
}
extern "C" int ilsx_net_num_params(const ilsx_net* n, size_t* out) {
  if (!n || !out) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_num_params: NULL argument");
  *out = n->lay.n_flat;
  return ILSX_OK;
}

int net_upload_flat(ilsx_ctx* ctx, const NetLayout& L, float* dev_base, const float* src, size_t n, int src_is_device) {
  if (n != L.n_flat) ILSX_FAIL(ILSX_ERR_ARG, "parameter count %zu != expected %zu", n, L.n_flat);
  std::vector<float> flat(L.n_flat), in(L.n_int);
  if (src_is_device) {
    HIPCHK(hipMemcpyAsync(flat.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    memcpy(flat.data(), src, n * sizeof(float));
  }
  net_flat_to_internal(L, flat.data(), in.data());
  HIPCHK(hipMemcpyAsync(dev_base, in.data(), L.n_int * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ILSX_OK;
}
int net_download_flat(ilsx_ctx* ctx, const NetLayout& L, const float* dev_base, float* dst, size_t n, int dst_is_device) {
  if (n != L.n_flat) ILSX_FAIL(ILSX_ERR_ARG, "parameter count %zu != expected %zu", n, L.n_flat);
  std::vector<float> flat(L.n_flat), in(L.n_int);
  HIPCHK(hipMemcpyAsync(in.data(), dev_base, L.n_int * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  net_internal_to_flat(L, in.data(), flat.data());
  if (dst_is_device) {
    HIPCHK(hipMemcpyAsync(dst, flat.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else {
    memcpy(dst, flat.data(), n * sizeof(float));
  }
  return ILSX_OK;
}

extern "C" int ilsx_net_set_params(ilsx_net* n, const float* src, size_t cnt, int src_is_device) {
  if (!n || !src) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_set_params: NULL argument");
  HIPCHK(hipSetDevice(n->ctx->device));
  return net_upload_flat(n->ctx, n->lay, n->base, src, cnt, src_is_device);
}
extern "C" int ilsx_net_get_params(const ilsx_net* n, float* dst, size_t cnt, int dst_is_device) {
  if (!n || !dst) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_get_params: NULL argument");
  HIPCHK(hipSetDevice(n->ctx->device));
  return net_download_flat(n->ctx, n->lay, n->base, dst, cnt, dst_is_device);
}

extern "C" int ilsx_net_init(ilsx_net* n, uint64_t seed, float init_w, float b_init) {
  if (!n) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_init: NULL net");
  // networks.py:57-83, pytorch_util.py:20-29: bound = 1/sqrt(weight.size(0)) = 1/sqrt(OUT features)
  const NetLayout& L = n->lay;
  std::mt19937_64 gen(seed);
  std::vector<float> flat(L.n_flat);
  size_t f = 0;
  for (int l = 0; l < L.cfg.n_hidden; ++l) {
    const int outd = L.out_of(l);
    const float bound = 1.0f / std::sqrt((float)outd);   // fanin_init: 1 / sqrt(weight.size(0)) — the layer's LOGICAL width
    std::uniform_real_distribution<float> u(-bound, bound);
    const size_t nw = (size_t)outd * L.in_of(l);
    for (size_t i = 0; i < nw; ++i) flat[f + i] = u(gen);
    f += nw;
    for (int i = 0; i < outd; ++i) flat[f + i] = b_init;
    f += outd;
  }
  std::uniform_real_distribution<float> uh(-init_w, init_w);
  for (; f < L.n_flat; ++f) flat[f] = uh(gen);
  return ilsx_net_set_params(n, flat.data(), flat.size(), 0);
}

extern "C" int ilsx_mlp_forward(ilsx_net* n, const float* x, int rows, float* y) {
  if (!n || !x || !y || rows < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_mlp_forward: bad argument");
  HIPCHK(hipSetDevice(n->ctx->device));
  FwdArgs A;
  memset(&A, 0, sizeof A);
  FwdTask& t = A.t[0];
  t.net = net_view(n->lay, n->base);
  t.x0 = x; t.d0 = n->lay.cfg.in_dim; t.s0 = n->lay.cfg.in_dim;
  t.out = y;
  t.head = HEAD_RAW;
  A.rows = rows; A.ntasks = 1; A.seed = n->ctx->seed;
  return launch_fwd(n->ctx, A, n->lay.cfg.hidden, n->lay.cfg.act, n->lay.KP);
}

extern "C" int ilsx_policy_act(ilsx_net* pi, const float* obs, int nrows, int deterministic, const float* eps,
                               float* act, float* logp) {
  if (!pi || !obs || !act || nrows < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_policy_act: bad argument");
  if (pi->lay.cfg.n_heads != 2 && !pi->noise_policy)
    ILSX_FAIL(ILSX_ERR_ARG, "ilsx_policy_act: network has %d heads (need mean|log_std, or a noise policy)", pi->lay.cfg.n_heads);
  HIPCHK(hipSetDevice(pi->ctx->device));
  FwdArgs A;
  memset(&A, 0, sizeof A);
  FwdTask& t = A.t[0];
  t.net = net_view(pi->lay, pi->base);
  t.x0 = obs; t.d0 = pi->lay.cfg.in_dim; t.s0 = pi->lay.cfg.in_dim;
  t.head = deterministic ? HEAD_TANH_DET : HEAD_TANH_SAMPLE;
  if (pi->noise_policy) {
    if (logp) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_policy_act: a noise policy has no log-probability");
    t.head = pi->out_linear ? HEAD_DET_LIN_NOISE : HEAD_DET_TANH_NOISE;
    t.noise = deterministic ? 0.0f : pi->noise; t.noise_clip = pi->noise_clip; t.max_act = pi->max_act;
  }
  t.eps = eps;
  t.action = act;
  t.logp = logp;
  t.rng_stream = 0x41435400u;  // 'ACT'
  A.rows = nrows; A.ntasks = 1; A.seed = pi->ctx->seed;
  A.scal = nullptr;
  A.step_host = ++pi->ctx->act_calls;   // per ctx, not per process: a run keeps its noise stream whatever else lives in the process (grouped runs)
  return launch_fwd(pi->ctx, A, pi->lay.cfg.hidden, pi->lay.cfg.act, pi->lay.KP);
}

extern "C" int ilsx_net_set_noise_policy(ilsx_net* pi, float policy_noise, float policy_noise_clip, float max_act) {
  if (!pi) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_set_noise_policy: NULL network");
  if (pi->lay.cfg.n_heads != 1) ILSX_FAIL(ILSX_ERR_ARG, "a noise policy is a single-head Mlp (policies.py:130-188)");
  pi->noise_policy = true; pi->noise = policy_noise; pi->noise_clip = policy_noise_clip; pi->max_act = max_act;
  return ILSX_OK;
}

extern "C" int ilsx_net_set_output_linear(ilsx_net* pi, int linear) {
  if (!pi) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_net_set_output_linear: NULL network");
  if (!pi->noise_policy) ILSX_FAIL(ILSX_ERR_STATE, "ilsx_net_set_output_linear: not a noise policy (ilsx_net_set_noise_policy first)");
  pi->out_linear = linear != 0;
  return ILSX_OK;
}

extern "C" int ilsx_policy_log_prob(ilsx_net* pi, const float* obs, const float* act, int nrows, float* logp) {
  if (!pi || !obs || !act || !logp || nrows < 0) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_policy_log_prob: bad argument");
  if (pi->lay.cfg.n_heads != 2) ILSX_FAIL(ILSX_ERR_ARG, "ilsx_policy_log_prob: network is not a Gaussian policy");
  HIPCHK(hipSetDevice(pi->ctx->device));
  FwdArgs A;
  memset(&A, 0, sizeof A);
  FwdTask& t = A.t[0];
  t.net = net_view(pi->lay, pi->base);
  t.x0 = obs; t.d0 = pi->lay.cfg.in_dim; t.s0 = pi->lay.cfg.in_dim;
  t.head = HEAD_TANH_LOGP_OF_ACT;
  t.act_in = act;
  t.logp = logp;
  A.rows = nrows; A.ntasks = 1; A.seed = pi->ctx->seed;
  return launch_fwd(pi->ctx, A, pi->lay.cfg.hidden, pi->lay.cfg.act, pi->lay.KP);
}
