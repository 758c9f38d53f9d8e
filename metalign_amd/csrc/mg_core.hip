// mg_core.hip — lifecycle, error text, raw device memory, per-kernel event timing.
#include <cstring>

#include "mg_internal.h"

namespace mg {

static thread_local char g_err[512] = "";

Context& ctx() {
  static Context c;
  return c;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- test / diagnostic knobs ----
static const char* const kDebugKeys[] = {
    "k3_hashed", "k3_flush_tiles", "k3_grid", "lds_pad", "force_list", "distinct_hint_ppm", "resident_scan", "no_fused", "resident_ablate",
    "flush_order", "no_avx2", "gzip_threads", "pgzip_chunk", "pgzip_thp", "pgzip_timing", "stream_thin", "stream_threads", "inflate_trace", "inflate_loose_find", "shares_threads", "kc_wg_per_cu", "kc_ablate", "kc_stagger", "kc_gate_extra", "inflate_dev_max_bytes"};
static int64_t g_debug[sizeof(kDebugKeys) / sizeof(kDebugKeys[0])] = {};
static int debug_index(const char* key) {
  for (size_t i = 0; i < sizeof(kDebugKeys) / sizeof(kDebugKeys[0]); ++i)
    if (strcmp(kDebugKeys[i], key) == 0) return (int)i;
  return -1;
}
int64_t dbg(const char* key) {
  const int i = debug_index(key);
  return i < 0 ? 0 : g_debug[i];
}

static uint64_t size_class(uint64_t n) {
  if (n < 256) return 256;
  // classes: powers of two and the midpoints between them (<= 33 % slack)
  uint64_t p = 256;
  while (p < n) {
    if (p + p / 2 >= n) return p + p / 2;
    p <<= 1;
  }
  return p;
}

static void seal_open_batch();
static void release_completed_batches();

void* pool_alloc(uint64_t bytes, uint64_t* got) {
  Context& c = ctx();
  const uint64_t cls = size_class(bytes);
  if (!c.fence_sealed.empty()) release_completed_batches();
  auto& fl = c.pool[cls];
  if (fl.empty() && !c.fence_open.empty()) {  // (lazily: one set of events per batch of frees, not per free)
    seal_open_batch();
    release_completed_batches();
  }
  if (!fl.empty()) {
    void* p = fl.back();
    fl.pop_back();
    *got = cls;
    return p;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {
    pool_release_all();  // give cached blocks back and retry once
    e = hipMalloc(&p, cls);
  }
  if (e != hipSuccess) {
    fail(MG_ERR_NOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)cls, hipGetErrorString(e));
    return nullptr;
  }
  *got = cls;
  return p;
}

// With the library's side streams in use (stage A x2, stage C) a block freed by the host may still be read or written
// by a kernel queued on ANOTHER stream than the one that will reuse it — the free list knows nothing about streams, and
// a handle destructor (an exception in the middle of a pipelined run, a __del__) does not synchronise.  Such blocks
// are fenced: frees are collected into a batch; when an allocation finds its free list empty, the open batch is
// SEALED — one event recorded on every stream of the library — and sealed batches whose events have all completed
// hand their blocks to the free lists.  With only the main stream in use nothing is fenced (stream order is enough).
static hipEvent_t fence_event() {
  Context& c = ctx();
  if (!c.fence_events.empty()) { hipEvent_t e = c.fence_events.back(); c.fence_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

static void seal_open_batch() {
  Context& c = ctx();
  if (c.fence_open.empty()) return;
  Context::FenceBatch b;
  b.blocks.swap(c.fence_open);
  hipStream_t streams[6] = {c.stream, c.stream_a, c.stream_a2, c.stream_c, c.stream_inf, c.stream_r};
  for (hipStream_t st : streams) {
    if (!st) continue;
    hipEvent_t e = fence_event();
    if (!e || hipEventRecord(e, st) != hipSuccess) {  // cannot fence: wait for everything instead
      (void)hipDeviceSynchronize();
      if (e) c.fence_events.push_back(e);
      continue;
    }
    b.events.push_back(e);
  }
  c.fence_sealed.push_back(std::move(b));
}

static void release_completed_batches() {
  Context& c = ctx();
  for (size_t i = 0; i < c.fence_sealed.size();) {
    Context::FenceBatch& b = c.fence_sealed[i];
    bool done = true;
    for (hipEvent_t e : b.events)
      if (hipEventQuery(e) != hipSuccess) { done = false; break; }
    if (!done) { ++i; continue; }
    for (auto& blk : b.blocks) c.pool[blk.second].push_back(blk.first);
    for (hipEvent_t e : b.events) c.fence_events.push_back(e);
    c.fence_sealed.erase(c.fence_sealed.begin() + (long)i);
  }
}

void pool_free(void* p, uint64_t bytes) {
  Context& c = ctx();
  if (!c.ready) { (void)hipFree(p); return; }
  if (c.a_side || c.c_side || c.inf_side) {  // side streams in use: reusable once they have passed
    c.fence_open.emplace_back(p, bytes);
    // Sealed promptly, not only when an allocation finds its free list empty: otherwise the open batch collects the
    // frees of as many passes as the free lists last, every drained list then costs fresh hipMallocs until the batch
    // completes, and the memory held grows with the square root of the number of passes (tools/soak.py: 4.0 -> 8.7 GB
    // over 6000 passes of configs[1]).
    if (c.fence_open.size() >= 32) seal_open_batch();
  } else {
    c.pool[bytes].push_back(p);
  }
}

void pool_release_all() {
  Context& c = ctx();
  if (!c.fence_open.empty() || !c.fence_sealed.empty()) (void)hipDeviceSynchronize();
  for (auto& blk : c.fence_open) (void)hipFree(blk.first);
  c.fence_open.clear();
  for (auto& b : c.fence_sealed) {
    for (auto& blk : b.blocks) (void)hipFree(blk.first);
    for (hipEvent_t e : b.events) c.fence_events.push_back(e);
  }
  c.fence_sealed.clear();
  for (auto& kv : c.pool)
    for (void* p : kv.second) (void)hipFree(p);
  c.pool.clear();
}

uint64_t* host_words() { return ctx().pinned; }

void* scratch(const char* name, uint64_t bytes) {
  Context& c = ctx();
  DevBuf*& b = c.scratch[std::string(c.scratch_prefix) + name];
  if (!b) b = new DevBuf();
  if (b->bytes < bytes || !b->p) {
    // growing frees the old block to the pool: with more than one stream in use, whatever still runs on it has to
    // finish first (rare: sizes settle after the first batch)
    if (b->p && (c.a_side || c.c_side || c.inf_side)) (void)hipDeviceSynchronize();
    uint64_t want = bytes + bytes / 4 + 256;
    if (b->alloc(want) != MG_OK) return nullptr;
  }
  return b->p;
}

void scratch_release_all() {
  Context& c = ctx();
  for (auto& kv : c.scratch) delete kv.second;
  c.scratch.clear();
}

static hipEvent_t prof_event() {
  Context& c = ctx();
  if (!c.prof_pool.empty()) {
    hipEvent_t e = c.prof_pool.back();
    c.prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(const char* n, hipStream_t st) : name(n), on(ctx().prof_on), stream(st ? st : ctx().stream) {
  if (on && ctx().prof_only[0] && strcmp(ctx().prof_only, n) != 0) on = false;
  if (!on) return;
  a = prof_event();
  b = prof_event();
  if (!a || !b) { on = false; return; }
  (void)hipEventRecord(a, stream);
}

ProfScope::~ProfScope() {
  if (!on) return;
  Context& c = ctx();
  (void)hipEventRecord(b, stream);
  c.prof_pending.push_back({name, a, b});
}

void prof_collect() {
  Context& c = ctx();
  for (auto& p : c.prof_pending) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ProfEntry& e = c.prof[p.name];
      e.launches += 1;
      e.total_ms += ms;
    }
    c.prof_pool.push_back(p.a);
    c.prof_pool.push_back(p.b);
  }
  c.prof_pending.clear();
}

}  // namespace mg

using mg::ctx;
using mg::fail;

extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }

int mg_debug_set(const char* key, int64_t value) {
  if (!key) {  // every knob back to "never set"
    memset(mg::g_debug, 0, sizeof(mg::g_debug));
    return MG_OK;
  }
  const int i = mg::debug_index(key);
  if (i < 0) return fail(MG_ERR_ARG, "mg_debug_set: no knob named %s", key);
  mg::g_debug[i] = value;
  return MG_OK;
}

int64_t mg_debug_get(const char* key) { return key ? mg::dbg(key) : 0; }

int mg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int init_common(int device, void* stream) {
  mg::Context& c = ctx();
  if (c.ready) mg_shutdown();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(MG_ERR_HIP, "no HIP device visible (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(MG_ERR_ARG, "device %d out of range [0,%d)", device, n);
  MG_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  MG_HIP(hipGetDeviceProperties(&prop, device));
  c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (stream) {
    c.stream = reinterpret_cast<hipStream_t>(stream);
    c.own_stream = false;
  } else {
    MG_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    c.own_stream = true;
  }
  MG_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.pinned), 64 * sizeof(uint64_t), hipHostMallocDefault));
  MG_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.pend_pinned), mg::Context::kPendSlots * 8 * sizeof(uint64_t), hipHostMallocDefault));
  c.device = device;
  c.ready = true;
  return MG_OK;
}

int mg_init(int device) { return init_common(device, nullptr); }
int mg_init_on_stream(int device, void* hip_stream) {
  // The legacy default stream has the handle 0: accepting it here would silently give the library a stream of its
  // own, unordered with the caller's work (torch's default stream is exactly that).
  if (!hip_stream) return fail(MG_ERR_ARG, "mg_init_on_stream needs an explicit stream (not the default stream, handle 0)");
  return init_common(device, hip_stream);
}

void mg_shutdown(void) {
  mg::Context& c = ctx();
  if (!c.ready) return;
  (void)hipStreamSynchronize(c.stream);
  mg::stream_release_all();
  mg::inflate_release_all();
  mg::scratch_release_all();
  mg::pool_release_all();
  if (c.pinned) (void)hipHostFree(c.pinned);
  if (c.pend_pinned) (void)hipHostFree(c.pend_pinned);
  if (c.stream_c) { (void)hipStreamSynchronize(c.stream_c); (void)hipStreamDestroy(c.stream_c); }
  if (c.stream_inf) { (void)hipStreamSynchronize(c.stream_inf); (void)hipStreamDestroy(c.stream_inf); }
  if (c.stream_r) { (void)hipStreamSynchronize(c.stream_r); (void)hipStreamDestroy(c.stream_r); }
  if (c.stream_a) { (void)hipStreamSynchronize(c.stream_a); (void)hipStreamDestroy(c.stream_a); }
  if (c.stream_a2) { (void)hipStreamSynchronize(c.stream_a2); (void)hipStreamDestroy(c.stream_a2); }
  for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : c.fence_events) (void)hipEventDestroy(e);
  if (c.ev_c) (void)hipEventDestroy(c.ev_c);
  mg::prof_collect();
  for (hipEvent_t e : c.prof_pool) (void)hipEventDestroy(e);
  if (c.own_stream && c.stream) (void)hipStreamDestroy(c.stream);
  c = mg::Context();
}

const char* mg_last_error(void) { return mg::g_err; }

int mg_mem_info(uint64_t* free_bytes, uint64_t* total_bytes, uint64_t* pooled_bytes) {
  MG_REQUIRE_READY();
  size_t f = 0, t = 0;
  MG_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  if (pooled_bytes) {
    mg::Context& c = mg::ctx();
    uint64_t held = 0;
    for (auto& kv : c.pool) held += kv.first * kv.second.size();
    for (auto& blk : c.fence_open) held += blk.second;
    for (auto& b : c.fence_sealed)
      for (auto& blk : b.blocks) held += blk.second;
    *pooled_bytes = held;
  }
  return MG_OK;
}

int mg_mem_trim(void) {
  MG_REQUIRE_READY();
  MG_HIP(hipDeviceSynchronize());  // nothing is in flight: the grow-only scratch buffers can go as well
  mg::scratch_release_all();
  mg::ctx().k3_priv_ptr = nullptr;  // (stage C's private bins are scratch: their all-zero invariant goes with them)
  mg::ctx().k3_priv_nb = 0;
  mg::ctx().k3_flags = nullptr;
  mg::ctx().k3_nflags = 0;
  mg::pool_release_all();
  return MG_OK;
}

int mg_device_name(char* buf, int cap) {
  MG_REQUIRE_READY();
  hipDeviceProp_t prop;
  MG_HIP(hipGetDeviceProperties(&prop, ctx().device));
  snprintf(buf, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return MG_OK;
}

int mg_dev_malloc(void** d_ptr, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (!d_ptr) return fail(MG_ERR_ARG, "null out pointer");
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(MG_ERR_NOMEM, "hipMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
  return MG_OK;
}

int mg_dev_free(void* d_ptr) {
  MG_REQUIRE_READY();
  if (d_ptr) MG_HIP(hipFree(d_ptr));
  return MG_OK;
}

int mg_memcpy_h2d(void* d_dst, const void* h_src, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (bytes == 0) return MG_OK;
  MG_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx().stream));
  MG_HIP(hipStreamSynchronize(ctx().stream));
  return MG_OK;
}

int mg_memcpy_d2h(void* h_dst, const void* d_src, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (bytes == 0) return MG_OK;
  if (ctx().c_side) MG_HIP(hipStreamSynchronize(ctx().stream_c));  // a synchronous read sees stage C's results too
  MG_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  MG_HIP(hipStreamSynchronize(ctx().stream));
  return MG_OK;
}

int mg_host_alloc(void** h_ptr, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (!h_ptr) return fail(MG_ERR_ARG, "null out pointer");
  hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) return fail(MG_ERR_NOMEM, "hipHostMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
  return MG_OK;
}

int mg_host_free(void* h_ptr) {
  MG_REQUIRE_READY();
  if (h_ptr) MG_HIP(hipHostFree(h_ptr));
  return MG_OK;
}

int mg_memcpy_h2d_async(void* d_dst, const void* h_pinned_src, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (bytes == 0) return MG_OK;
  MG_HIP(hipMemcpyAsync(d_dst, h_pinned_src, bytes, hipMemcpyHostToDevice, ctx().stream));
  return MG_OK;
}

int mg_memcpy_d2h_async(void* h_pinned_dst, const void* d_src, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (bytes == 0) return MG_OK;
  MG_HIP(hipMemcpyAsync(h_pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx().stream));
  return MG_OK;
}

int mg_dev_memset(void* d_ptr, int byte_value, uint64_t bytes) {
  MG_REQUIRE_READY();
  if (bytes == 0) return MG_OK;
  MG_HIP(hipMemsetAsync(d_ptr, byte_value, bytes, ctx().stream));
  return MG_OK;
}

int mg_sync(void) {
  MG_REQUIRE_READY();
  if (ctx().c_side) MG_HIP(hipStreamSynchronize(ctx().stream_c));
  MG_HIP(hipStreamSynchronize(ctx().stream));
  return MG_OK;
}

int mg_stage_c_side_stream(int on) {
  MG_REQUIRE_READY();
  mg::Context& c = ctx();
  if (on && !c.stream_c) {
    MG_HIP(hipStreamCreateWithFlags(&c.stream_c, hipStreamNonBlocking));
    MG_HIP(hipEventCreateWithFlags(&c.ev_c, hipEventDisableTiming));
  }
  if (!on && c.c_side) MG_HIP(hipStreamSynchronize(c.stream_c));
  c.c_side = on != 0;
  return MG_OK;
}

int mg_stage_a_side_stream(int on) {
  MG_REQUIRE_READY();
  mg::Context& c = ctx();
  if (on < 0 || on > 4) return fail(MG_ERR_ARG, "stage-A stream selector %d outside [0,4]", on);
  // 3 / 4: streams 1 / 2 at the LOWEST priority the device has.  A single shard's passes (stage A of consecutive passes on the two
  // streams in turn, no collective anywhere) run 5 % faster with them — the short kernels of stage B and C on the other streams
  // get their wavefronts ahead of the next pass's persistent counting kernel: 2.33 -> 2.22 ms per pass at configs[2]; with the
  // exchange's collectives on the main stream it is the other way round (2.51 -> 2.76 ms), so a job asks for what it runs.
  const bool want_low = on >= 3;
  if (on >= 3) on -= 2;
  if (on && want_low != c.a_low) {  // the other kind: the streams are made anew (rare: once per job)
    for (hipStream_t* s : {&c.stream_a, &c.stream_a2})
      if (*s) { MG_HIP(hipStreamSynchronize(*s)); MG_HIP(hipStreamDestroy(*s)); *s = nullptr; }
    c.a_low = want_low;
  }
  {
    int lo = 0, hi = 0;  // (numerically hi <= lo: hi is the greatest priority)
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int w = 1; w <= 2; ++w) {
      hipStream_t* s = w == 1 ? &c.stream_a : &c.stream_a2;
      if (on != w || *s) continue;
      if (c.a_low) MG_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo));
      else MG_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    }
  }
  if (!on && c.a_side) {
    if (c.stream_a) MG_HIP(hipStreamSynchronize(c.stream_a));
    if (c.stream_a2) MG_HIP(hipStreamSynchronize(c.stream_a2));
  }
  c.a_side = on;
  return MG_OK;
}

int mg_event_create(void** ev) {
  MG_REQUIRE_READY();
  if (!ev) return fail(MG_ERR_ARG, "null out pointer");
  hipEvent_t e = nullptr;
  MG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *ev = e;
  return MG_OK;
}

int mg_event_record(void* ev) {
  MG_REQUIRE_READY();
  if (!ev) return fail(MG_ERR_ARG, "null event");
  MG_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), ctx().stream));
  return MG_OK;
}

int mg_event_synchronize(void* ev) {
  MG_REQUIRE_READY();
  if (!ev) return fail(MG_ERR_ARG, "null event");
  MG_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev)));
  return MG_OK;
}

int mg_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(ev));
  return MG_OK;
}

int mg_stage_a_workgroups_per_cu(int n) {
  MG_REQUIRE_READY();
  if (n < 0 || n > 8) return fail(MG_ERR_ARG, "workgroups per CU %d outside [0,8]", n);
  ctx().a_side_wg_per_cu = (unsigned)n;
  return MG_OK;
}

int mg_stage_c_join(void) {
  MG_REQUIRE_READY();
  mg::Context& c = ctx();
  if (!c.c_side) return MG_OK;
  MG_HIP(hipEventRecord(c.ev_c, c.stream_c));
  MG_HIP(hipStreamWaitEvent(c.stream, c.ev_c, 0));
  return MG_OK;
}

int mg_prof_enable(int on) {
  MG_REQUIRE_READY();
  if (on) {  // event creation costs tens of microseconds apiece: not inside somebody's timed region
    mg::Context& c = ctx();
    while (c.prof_pool.size() < 512) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) break;
      c.prof_pool.push_back(e);
    }
  }
  ctx().prof_on = on != 0;
  ctx().prof_only[0] = 0;
  return MG_OK;
}

int mg_prof_only(const char* kernel) {
  MG_REQUIRE_READY();
  snprintf(ctx().prof_only, sizeof(ctx().prof_only), "%s", kernel ? kernel : "");
  return MG_OK;
}

int mg_prof_reset(void) {
  MG_REQUIRE_READY();
  mg::prof_collect();
  ctx().prof.clear();
  return MG_OK;
}

int mg_prof_get(const char* kernel, uint64_t* launches, double* total_ms) {
  MG_REQUIRE_READY();
  mg::prof_collect();
  auto it = ctx().prof.find(kernel ? kernel : "");
  if (launches) *launches = it == ctx().prof.end() ? 0 : it->second.launches;
  if (total_ms) *total_ms = it == ctx().prof.end() ? 0.0 : it->second.total_ms;
  return MG_OK;
}

}  // extern "C"
