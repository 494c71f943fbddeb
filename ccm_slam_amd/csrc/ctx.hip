// ctx.hip — context, error reporting, device memory wrappers, event-based kernel timing.
#include "common.h"
#include <cstring>

static std::mutex g_err_mu;
static std::string g_last_err;

bool ccm_dbg(const char* topic) {
  static const std::string list = [] { const char* e = std::getenv("CCM_DBG"); return std::string(e ? e : ""); }();
  if (list.empty()) return false;
  if (list == "all" || list == "1") return true;
  const std::string t(topic);
  size_t at = 0;
  while (at <= list.size()) {
    size_t end = list.find(',', at);
    if (end == std::string::npos) end = list.size();
    if (list.compare(at, end - at, t) == 0) return true;
    at = end + 1;
  }
  return false;
}

int ccm_set_error(ccm_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  std::lock_guard<std::mutex> lk(g_err_mu);
  g_last_err = msg;
  return code;
}

// ---- co-residency lease (common.h: ccm_coresident_scope) --------------------------------------
namespace {
constexpr int kLeaseDevices = 64, kLeaseRing = 64;
struct DevLease {
  std::mutex mu;
  int n_ctx = 0;                      // live contexts of this process on the device
  hipEvent_t ring[kLeaseRing] = {};   // created on first use; an event is re-recorded only kLeaseRing launches later (a wait captures the record it was issued behind)
  int head = -1;                      // ring[head]: recorded behind the last bracketed kernel
  hipStream_t last_stream = nullptr;  // the stream that kernel went to (the same stream needs no wait: stream order)
  int64_t launches = 0, chained = 0, aborts = 0;
};
DevLease g_lease[kLeaseDevices];
}  // namespace

ccm_coresident_scope::ccm_coresident_scope(ccm_ctx* c) : ctx(c) {
  DevLease& L = g_lease[c->device % kLeaseDevices];
  L.mu.lock();
  L.launches++;
  if (L.n_ctx > 1 && L.head >= 0 && L.last_stream != c->stream) {
    if (hipStreamWaitEvent(c->stream, L.ring[L.head], 0) == hipSuccess) { chained = true; L.chained++; }
    else (void)hipGetLastError();
  }
}
ccm_coresident_scope::~ccm_coresident_scope() {
  DevLease& L = g_lease[ctx->device % kLeaseDevices];
  if (L.n_ctx > 1) {
    const int nxt = (L.head + 1) % kLeaseRing;
    if (!L.ring[nxt] && hipEventCreateWithFlags(&L.ring[nxt], hipEventDisableTiming) != hipSuccess) { L.ring[nxt] = nullptr; (void)hipGetLastError(); }
    if (L.ring[nxt] && hipEventRecord(L.ring[nxt], ctx->stream) == hipSuccess) { L.head = nxt; L.last_stream = ctx->stream; }
    else (void)hipGetLastError();
  }
  L.mu.unlock();
}
void ccm_coresident_note_abort(ccm_ctx* ctx) {
  DevLease& L = g_lease[ctx->device % kLeaseDevices];
  std::lock_guard<std::mutex> lk(L.mu);
  L.aborts++;
}
// a context joins / leaves its device's lease.  The second context of a process waits once for the device: a bracketed kernel the first context launched
// while it was alone has no event behind it.
static void lease_join(ccm_ctx* c) {
  DevLease& L = g_lease[c->device % kLeaseDevices];
  bool second;
  { std::lock_guard<std::mutex> lk(L.mu); second = ++L.n_ctx == 2; }
  // (round 6, advisor) the drain runs OUTSIDE the lease mutex: from the increment above every bracketed launch of the other context records its event, so all the drain
  // has to cover is what that context launched while it was alone — and a thread that launches meanwhile no longer blocks on the mutex for a whole device drain.  The
  // ring is left as it is: an event recorded in between must not be forgotten, and waiting for an old, completed one costs nothing.
  if (second) (void)hipDeviceSynchronize();
}
static void lease_leave(ccm_ctx* c) {
  DevLease& L = g_lease[c->device % kLeaseDevices];
  std::lock_guard<std::mutex> lk(L.mu);
  L.n_ctx--;
  if (L.last_stream == c->stream) { L.last_stream = nullptr; L.head = -1; }   // (the stream was synchronised by the caller: nothing of it is in flight, and its handle may be reused)
}

extern "C" int ccm_coresidency_stats(int device_id, int64_t* launches, int64_t* chained, int64_t* aborted, int* contexts) {
  if (device_id < 0) return CCM_E_ARG;
  DevLease& L = g_lease[device_id % kLeaseDevices];   // (the same rule as the lease itself: lease_join)
  std::lock_guard<std::mutex> lk(L.mu);
  if (launches) *launches = L.launches;
  if (chained) *chained = L.chained;
  if (aborted) *aborted = L.aborts;
  if (contexts) *contexts = L.n_ctx;
  return CCM_OK;
}

extern "C" const char* ccm_version(void) { return "ccm_hip 0.1 gfx950"; }

extern "C" int ccm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char* ccm_last_error(const ccm_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  // copy under the lock: the returned pointer must stay valid after another thread replaces g_last_err
  static thread_local std::string tl_copy;
  std::lock_guard<std::mutex> lk(g_err_mu);
  tl_copy = g_last_err;
  return tl_copy.c_str();
}

extern "C" int ccm_ctx_create(int device_id, ccm_ctx** out) {
  if (!out) return ccm_set_error(nullptr, CCM_E_ARG, "ccm_ctx_create: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return ccm_set_error(nullptr, CCM_E_NOGPU,
                         "ccm_ctx_create: no HIP device visible (this library has no CPU fallback)");
  if (device_id < 0 || device_id >= n)
    return ccm_set_error(nullptr, CCM_E_ARG, "ccm_ctx_create: device_id out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
    return ccm_set_error(nullptr, CCM_E_HIP, "ccm_ctx_create: hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return ccm_set_error(nullptr, CCM_E_NOGPU,
                         std::string("ccm_ctx_create: device is ") + prop.gcnArchName +
                             ", kernels are built for gfx950 only");
  ccm_ctx* c = new ccm_ctx();
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return ccm_set_error(nullptr, CCM_E_HIP, "ccm_ctx_create: stream creation failed");
  }
  lease_join(c);
  *out = c;
  return CCM_OK;
}

extern "C" void ccm_ctx_destroy(ccm_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  lease_leave(ctx);
  ccm_comm_destroy(ctx);
  for (auto& s : ctx->prof)
    for (auto& p : s.pending) { ctx->ev_pool.push_back(p.first); ctx->ev_pool.push_back(p.second); }
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  if (ctx->d_scratch) hipFree(ctx->d_scratch);
  if (ctx->d_io) hipFree(ctx->d_io);
  if (ctx->h_pin) hipHostFree(ctx->h_pin);
  for (void* b : ctx->rb_free) hipHostFree(b);
  ctx->rb_free.clear();
  for (auto& kv : ctx->pool_free) hipFree(kv.second);
  hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int ccm_ctx_sync(ccm_ctx* ctx) {
  if (!ctx) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

extern "C" int ccm_dev_alloc(ccm_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  CCM_HIP_CHECK(ctx, hipMalloc(dptr, bytes ? bytes : 16));
  return CCM_OK;
}
extern "C" int ccm_dev_free(ccm_ctx* ctx, void* dptr) {
  if (!ctx) return CCM_E_ARG;
  if (dptr) CCM_HIP_CHECK(ctx, hipFree(dptr));
  return CCM_OK;
}
extern "C" int ccm_memcpy_h2d(ccm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}
extern "C" int ccm_memcpy_d2h(ccm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return CCM_E_ARG;
  CCM_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  CCM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CCM_OK;
}

int ccm_scratch(ccm_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->d_scratch_bytes) {
    if (ctx->d_scratch) { hipStreamSynchronize(ctx->stream); hipFree(ctx->d_scratch); ctx->d_scratch = nullptr; }
    size_t nb = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&ctx->d_scratch, nb);
    if (e != hipSuccess) { ctx->d_scratch_bytes = 0; return ccm_set_error(ctx, CCM_E_HIP, "scratch hipMalloc failed"); }
    ctx->d_scratch_bytes = nb;
  }
  *out = ctx->d_scratch;
  return CCM_OK;
}

int ccm_io_scratch(ccm_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->d_io_bytes) {
    if (ctx->d_io) { hipStreamSynchronize(ctx->stream); hipFree(ctx->d_io); ctx->d_io = nullptr; }
    size_t nb = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&ctx->d_io, nb);
    if (e != hipSuccess) { ctx->d_io_bytes = 0; return ccm_set_error(ctx, CCM_E_HIP, "io scratch hipMalloc failed"); }
    ctx->d_io_bytes = nb;
  }
  *out = ctx->d_io;
  return CCM_OK;
}

// ---- profiling -----------------------------------------------------------------------
static hipEvent_t take_event(ccm_ctx* ctx) {
  if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}

ccm_prof_scope::ccm_prof_scope(ccm_ctx* c, int k, hipStream_t launch_stream) : ctx(c), cls(k), st(launch_stream ? launch_stream : (c ? c->stream : nullptr)) {
  on = c && (c->prof_class == -1 || c->prof_class == k);
  if (!on) return;
  e0 = take_event(ctx); e1 = take_event(ctx);
  hipEventRecord(e0, st);   // on the stream the bracketed kernels are launched on (the ORB batch path alternates between two)
}
ccm_prof_scope::~ccm_prof_scope() {
  if (!on) return;
  hipEventRecord(e1, st);
  ctx->prof[cls].pending.emplace_back(e0, e1);
}

extern "C" int ccm_prof_enable(ccm_ctx* ctx, int kernel_class) {
  if (!ctx || kernel_class < -2 || kernel_class >= CCM_K_COUNT) return CCM_E_ARG;
  ctx->prof_class = kernel_class;
  return CCM_OK;
}

static void drain(ccm_ctx* ctx, ccm_prof_slot& s) {
  for (auto& p : s.pending) {
    hipEventSynchronize(p.second);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { s.total_ms += ms; s.launches++; }
    ctx->ev_pool.push_back(p.first); ctx->ev_pool.push_back(p.second);
  }
  s.pending.clear();
}

extern "C" int ccm_prof_reset(ccm_ctx* ctx) {
  if (!ctx) return CCM_E_ARG;
  hipStreamSynchronize(ctx->stream);
  for (auto& s : ctx->prof) { drain(ctx, s); s.launches = 0; s.total_ms = 0.0; }
  return CCM_OK;
}

extern "C" int ccm_prof_read(ccm_ctx* ctx, int kernel_class, int64_t* launches, double* total_ms) {
  if (!ctx || kernel_class < 0 || kernel_class >= CCM_K_COUNT) return CCM_E_ARG;
  hipStreamSynchronize(ctx->stream);
  drain(ctx, ctx->prof[kernel_class]);
  if (launches) *launches = ctx->prof[kernel_class].launches;
  if (total_ms) *total_ms = ctx->prof[kernel_class].total_ms;
  return CCM_OK;
}

int ccm_pin_scratch(ccm_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->h_pin_bytes) {
    if (ctx->h_pin) { hipStreamSynchronize(ctx->stream); hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; }
    const size_t nb = ccm_align256(bytes * 2);
    hipError_t e = hipHostMalloc(&ctx->h_pin, nb, hipHostMallocCoherent);   // (the pose optimisation polls a ticket its kernel writes into this block: fine-grained, explicitly)
    if (e != hipSuccess) { ctx->h_pin_bytes = 0; return ccm_set_error(ctx, CCM_E_HIP, "pinned scratch hipHostMalloc failed"); }
    ctx->h_pin_bytes = nb;
  }
  *out = ctx->h_pin;
  return CCM_OK;
}

// buckets: 4 KiB, then {1, 1.5} x 2^k — at most 50 % slack, few distinct sizes
static size_t pool_bucket(size_t bytes) {
  for (size_t b = 4096;; b *= 2) {
    if (b >= bytes) return b;
    if (b + b / 2 >= bytes) return b + b / 2;
  }
}

int ccm_pool_get(ccm_ctx* ctx, size_t bytes, void** out, size_t* actual) {
  const size_t b = pool_bucket(bytes);
  auto it = ctx->pool_free.find(b);
  if (it != ctx->pool_free.end()) { *out = it->second; ctx->pool_free.erase(it); ctx->pool_bytes -= b; *actual = b; return CCM_OK; }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, b);
  if (e != hipSuccess) {   // give the cache back to the driver and retry once
    for (auto& kv : ctx->pool_free) hipFree(kv.second);
    ctx->pool_free.clear(); ctx->pool_bytes = 0;
    e = hipMalloc(&p, b);
  }
  if (e != hipSuccess) return ccm_set_error(ctx, CCM_E_HIP, std::string("pooled hipMalloc failed: ") + hipGetErrorString(e));
  *out = p; *actual = b;
  return CCM_OK;
}

void ccm_pool_put(ccm_ctx* ctx, void* p, size_t actual) {
  constexpr size_t kKeep = (size_t)16 << 30;   // retain at most 16 GiB of idle blocks (288 GB of HBM per GPU)
  if (!p) return;
  if (ctx->pool_bytes + actual > kKeep) { hipFree(p); return; }
  ctx->pool_free.emplace(actual, p);
  ctx->pool_bytes += actual;
}
