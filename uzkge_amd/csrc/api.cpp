// extern "C" surface of libuzkge_gpu.so (declared in include/uzkge_gpu.h).
// Thin: argument checks, staging of host buffers, error mapping; all compute is in the .hip files.
// There is deliberately no CPU fallback: without a gfx950 device every compute call fails with
// UZK_ERR_DEVICE.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>

#include "ctx.hpp"
#include "host_ec64.hpp"
#include "host_math.hpp"

namespace uzk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// Every extern "C" entry point is a function-try-block that ends here: no C++ exception (std::bad_alloc from a host-side vector,
// map or shared_ptr; std::system_error from a mutex) crosses the C ABI into a caller that cannot unwind it (Rust with
// panic = "abort", ctypes).  Locks taken by the entry point are released by the unwinding; a prover that was mid-round must be
// destroyed by the caller.
int on_exception(const char* fn) noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { set_error("%s: out of host memory", fn); }
    catch (const std::exception& e) { set_error("%s: %s", fn, e.what()); }
    catch (...) { set_error("%s: C++ exception of unknown type", fn); }
    return UZK_ERR_DEVICE;
}

// Process-wide state: the device this process is bound to, the SRS registry (bases are read-only and shared by
// every context), the contexts.  Lock order: a context's own mutex first, then Shared::mu (held briefly).
struct Shared {
    std::mutex mu;
    bool bound = false;
    int device = -1;                       // the process's default device (uzk_init); contexts of uzk_ctx_create_on name their own
    std::map<int, int> num_cus;            // per device that a context has been made ready on

    std::map<uint64_t, Ctx::Srs> srs;
    uint64_t next_handle = 1;
    std::map<uint64_t, Ctx*> contexts;
    uint64_t next_ctx = 1;                 // never reused: a stale handle cannot name a later context
    std::atomic<uint64_t> epoch{1};        // bumped whenever a context is destroyed (uzk_ctx_destroy, uzk_shutdown)
    std::map<const void*, size_t> pinned;  // uzk_host_alloc blocks (base -> bytes): uploads from them are asynchronous
};
static Shared& shared() {
    static Shared s;
    return s;
}
static Ctx& default_ctx() {
    static Ctx c;
    return c;
}
// The calling thread's context is remembered by HANDLE; the pointer is only a cache, valid while no context has been
// destroyed since it was looked up (Shared::epoch).  A worker thread that outlives uzk_shutdown / uzk_ctx_destroy of its
// context therefore falls back to the default context instead of dereferencing a freed Ctx.
static thread_local uint64_t t_handle = 0;          // 0: the default context
static thread_local Ctx* t_cached = nullptr;
static thread_local uint64_t t_epoch = 0;
static thread_local Ctx* t_scope = nullptr;         // CtxScope: a library-internal context while a shared round runs on this thread

CtxScope::CtxScope(Ctx* c) : prev(t_scope) { t_scope = c; }
CtxScope::~CtxScope() { t_scope = prev; }

Ctx& ctx() {
    if (t_scope) return *t_scope;
    if (t_handle == 0) return default_ctx();
    Shared& s = shared();
    if (t_cached && t_epoch == s.epoch.load(std::memory_order_acquire)) return *t_cached;
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.contexts.find(t_handle);
    if (it == s.contexts.end()) { t_handle = 0; t_cached = nullptr; return default_ctx(); }
    t_cached = it->second;
    t_epoch = s.epoch.load(std::memory_order_acquire);
    return *t_cached;
}
std::mutex& ctx_mutex() { return ctx().mu; }

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return UZK_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + (bytes >> 3);   // 12.5 % head-room against regrowth
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();        // clear the runtime's sticky last error
        p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    cap = want;
    return UZK_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

hipEvent_t Ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void Ctx::prof_begin(const char* name) {
    ProfEntry pe;
    pe.name = name;
    pe.e0 = get_event();
    pe.e1 = get_event();
    (void)hipEventRecord(pe.e0, cur_stream ? cur_stream : stream);
    prof_pending.push_back(pe);
}
void Ctx::prof_end() {
    if (!prof_pending.empty()) (void)hipEventRecord(prof_pending.back().e1, cur_stream ? cur_stream : stream);
}
int Ctx::prof_collect() {
    if (prof_pending.empty()) return UZK_OK;
    UZK_HIP(hipStreamSynchronize(stream));
    if (stream2) UZK_HIP(hipStreamSynchronize(stream2));
    for (auto& pe : prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
            auto& t = prof_totals[pe.name];
            t.first += ms;
            t.second += 1;
        }
        event_pool.push_back(pe.e0);
        event_pool.push_back(pe.e1);
    }
    prof_pending.clear();
    return UZK_OK;
}

// The device this process was last bound to by uzk_init: a lazy re-initialisation after uzk_shutdown
// returns to it instead of silently moving to device 0.
static int g_last_device = 0;

// compute units of `device` (Shared::mu held by the caller); the device becomes one uzk_shutdown synchronises
static int device_cus_locked(Shared& s, int device, int* cus) {
    auto it = s.num_cus.find(device);
    if (it == s.num_cus.end()) {
        int v = 0;
        UZK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device));
        it = s.num_cus.emplace(device, v > 0 ? v : 256).first;
    }
    *cus = it->second;
    return UZK_OK;
}
// binds the process's default device (Shared::mu held by the caller)
static int bind_device_locked(Shared& s, int device) {
    UZK_HIP(hipSetDevice(device));
    int cus = 0;
    UZK_TRY(device_cus_locked(s, device, &cus));
    s.device = device;
    s.bound = true;
    g_last_device = device;
    return UZK_OK;
}

// Makes the calling thread's context usable: process bound to a device, HIP's per-thread current device set to it,
// the context's stream created.  Called with the context's mutex held.
int require_ready() {
    Shared& s = shared();
    Ctx& c = ctx();
    int device;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (!s.bound) {
            // lazy init (device 0, or the one uzk_init bound before a shutdown) so a plain library user need not call uzk_init
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
                set_error("no HIP device visible: the MI355X backend has no CPU fallback");
                return UZK_ERR_DEVICE;
            }
            UZK_TRY(bind_device_locked(s, g_last_device < n ? g_last_device : 0));
        }
        device = c.want_device >= 0 ? c.want_device : s.device;
        UZK_TRY(device_cus_locked(s, device, &c.num_cus));
    }
    // HIP's current device is per thread: a prover thread that never called uzk_init would otherwise
    // allocate and launch on device 0 while the streams and the SRS live on the context's device
    UZK_HIP(hipSetDevice(device));
    if (!c.ready) {
        UZK_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        c.device = device;
        c.ready = true;
    }
    return UZK_OK;
}

void devices_synchronize() {
    Shared& s = shared();
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        for (auto& kv : s.num_cus) devs.push_back(kv.first);
    }
    for (int d : devs) { (void)hipSetDevice(d); (void)hipDeviceSynchronize(); }
}

// frees everything a context owns (its mutex held, or no other user left)
static void ctx_release(Ctx& c) {
    if (!c.ready) return;
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    ntt_free_plans(c);
    msm_free(c);
    poly_free(c);
    c.ntt_scratch[0].release(); c.ntt_scratch[1].release(); c.ntt_io.release(); c.msm_scalars.release();
    if (c.msm_tail_host) { (void)hipHostFree(c.msm_tail_host); c.msm_tail_host = nullptr; c.msm_tail_cap = 0; }
    for (auto& pe : c.prof_pending) { (void)hipEventDestroy(pe.e0); (void)hipEventDestroy(pe.e1); }
    c.prof_pending.clear();
    for (auto e : c.event_pool) (void)hipEventDestroy(e);
    c.event_pool.clear();
    (void)hipStreamDestroy(c.stream);
    c.stream = nullptr;
    c.cur_stream = nullptr;
    c.ready = false;
    c.device = -1;
}

static const Fp* as_fp(const uint64_t* p) { return reinterpret_cast<const Fp*>(p); }

static void copy_tuning(const Ctx& from, Ctx& to);
int ctx_init_internal(Ctx& c, int device) {
    c.want_device = device;
    {
        std::lock_guard<std::mutex> lk(default_ctx().mu);
        copy_tuning(default_ctx(), c);
    }
    CtxScope scope(&c);
    std::lock_guard<std::mutex> lk(c.mu);
    return require_ready();
}
void ctx_release_internal(Ctx& c) {
    std::lock_guard<std::mutex> lk(c.mu);
    ctx_release(c);
}

}  // namespace uzk

using namespace uzk;

#define API_LOCK std::lock_guard<std::mutex> _lk(ctx_mutex())

extern "C" {

#ifndef UZK_SRC_HASH
#define UZK_SRC_HASH "unstamped"
#endif
const char* uzk_version(void) { return "uzkge-amd 0.3 (gfx950) src:" UZK_SRC_HASH; }
const char* uzk_last_error(void) { return g_err; }

int uzk_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} catch (...) { return uzk::on_exception("uzk_device_count"); }

int uzk_init(int device) try {
    API_LOCK;
    Shared& s = shared();
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (s.bound && s.device != device) {
            set_error("uzk_init(%d): already bound to device %d (one process per GPU)", device, s.device);
            return UZK_ERR_PARAMETER;
        }
        if (!s.bound) {
            int n = 0;
            if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
                set_error("no HIP device visible: the MI355X backend has no CPU fallback");
                return UZK_ERR_DEVICE;
            }
            if (device < 0 || device >= n) {
                set_error("uzk_init(%d): %d device(s) visible", device, n);
                return UZK_ERR_PARAMETER;
            }
            UZK_TRY(bind_device_locked(s, device));
        }
    }
    return require_ready();
} catch (...) { return uzk::on_exception("uzk_init"); }

// Frees every context (the default one and those of uzk_ctx_create), every SRS and table, and unbinds the device.
// No other thread may be inside the library.  Threads that had made a destroyed context current fall back to the
// default context on their next call (the handle no longer resolves).  Memory from uzk_dev_alloc / uzk_host_alloc
// belongs to the caller and is not touched.
int uzk_shutdown(void) try {
    Shared& s = shared();
    coalesce_release_all();                // shared provers, their workspaces and internal contexts
    sharded_release_all();                 // sharded SRSs: their chunks' contexts and registry entries
    prover_release_all();                  // circuits and provers own device memory (takes Shared::mu itself)
    std::lock_guard<std::mutex> lk(s.mu);
    if (!s.bound) return UZK_OK;
    ctx_release(default_ctx());
    for (auto& kv : s.contexts) { ctx_release(*kv.second); delete kv.second; }
    s.contexts.clear();
    s.epoch.fetch_add(1, std::memory_order_acq_rel);
    t_handle = 0;
    t_cached = nullptr;
    for (auto& kv : s.srs) {
        (void)hipSetDevice(kv.second.device);
        if (kv.second.owned && kv.second.d_points) (void)hipFree(kv.second.d_points);
        if (kv.second.d_table) (void)hipFree(kv.second.d_table);
    }
    s.srs.clear();
    s.num_cus.clear();
    s.bound = false;
    s.device = -1;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_shutdown"); }

/* ---- contexts ------------------------------------------------------------------------------- */
// the experiment switches and the forced window width travel with the creator: a tool that tunes the default context
// and then starts prover threads measures what it configured
namespace uzk {
static void copy_tuning(const Ctx& from, Ctx& to) {
    to.msm_window_bits = from.msm_window_bits;
    to.tune_no_precompute = from.tune_no_precompute; to.tune_ntt_tile = from.tune_ntt_tile; to.tune_ntt_two_pass = from.tune_ntt_two_pass; to.tune_small = from.tune_small;
    to.tune_chunk_log = from.tune_chunk_log; to.tune_stream_log = from.tune_stream_log; to.tune_stream_min_log = from.tune_stream_min_log;
    to.tune_seg_sort = from.tune_seg_sort; to.tune_arith29 = from.tune_arith29;
}
}  // namespace uzk
int uzk_ctx_create(uint64_t* ctx_out) try {
    if (!ctx_out) { set_error("uzk_ctx_create: null pointer"); return UZK_ERR_PARAMETER; }
    Ctx* c = new Ctx();
    {
        API_LOCK;                                      // binds the device through the caller's current context
        int rc = require_ready();
        if (rc != UZK_OK) { delete c; return rc; }
        copy_tuning(ctx(), *c);
    }
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    const uint64_t h = s.next_ctx++;
    s.contexts[h] = c;
    *ctx_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_create"); }
// A context on a named device: one process can then drive several GPUs -- a pool of prover threads, each with a context (and its
// circuits and provers) on its own device, or the point chunks of one MSM (uzk_msm_g1_sharded).  uzk_init keeps its meaning: the
// device of the default context and of uzk_ctx_create.
int uzk_ctx_create_on(int device, uint64_t* ctx_out) try {
    if (!ctx_out) { set_error("uzk_ctx_create_on: null pointer"); return UZK_ERR_PARAMETER; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible: the MI355X backend has no CPU fallback"); return UZK_ERR_DEVICE; }
    if (device < 0 || device >= n) { set_error("uzk_ctx_create_on(%d): %d device(s) visible", device, n); return UZK_ERR_PARAMETER; }
    Ctx* c = new Ctx();
    {
        API_LOCK;
        copy_tuning(ctx(), *c);
    }
    c->want_device = device;
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    const uint64_t h = s.next_ctx++;
    s.contexts[h] = c;
    *ctx_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_create_on"); }
int uzk_ctx_device(uint64_t handle, int* device_out) try {
    if (!device_out) { set_error("uzk_ctx_device: null pointer"); return UZK_ERR_PARAMETER; }
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    Ctx* c = nullptr;
    if (handle == 0) c = &default_ctx();
    else {
        auto it = s.contexts.find(handle);
        if (it == s.contexts.end()) { set_error("uzk_ctx_device: unknown context %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        c = it->second;
    }
    *device_out = c->ready ? c->device : (c->want_device >= 0 ? c->want_device : (s.bound ? s.device : g_last_device));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_device"); }
int uzk_ctx_set_current(uint64_t handle) try {
    if (handle == 0) { t_handle = 0; t_cached = nullptr; return UZK_OK; }
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.contexts.find(handle);
    if (it == s.contexts.end()) { set_error("uzk_ctx_set_current: unknown context %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
    t_handle = handle;
    t_cached = it->second;
    t_epoch = s.epoch.load(std::memory_order_acquire);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_set_current"); }
int uzk_ctx_destroy(uint64_t handle) try {
    Shared& s = shared();
    Ctx* c = nullptr;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.contexts.find(handle);
        if (it == s.contexts.end()) { set_error("uzk_ctx_destroy: unknown context %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        c = it->second;
        s.contexts.erase(it);
        s.epoch.fetch_add(1, std::memory_order_acq_rel);   // every thread's cached pointer is looked up again
    }
    if (t_handle == handle) { t_handle = 0; t_cached = nullptr; }
    { std::lock_guard<std::mutex> lk(c->mu); ctx_release(*c); }
    delete c;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_destroy"); }

int uzk_ctx_current(uint64_t* ctx_out) try {
    if (!ctx_out) { set_error("uzk_ctx_current: null pointer"); return UZK_ERR_PARAMETER; }
    (void)ctx();                                         // a handle whose context is gone resolves to the default context here
    *ctx_out = t_handle;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_current"); }
// The calling thread's current context waits (on the device: no host synchronisation) for everything queued so far on
// `other`.  Two contexts in one prover thread are two lanes with their own stream and workspaces: independent steps of a
// proof -- the coset FFTs of the wire polynomials and the commit of the same round, the two openings -- run side by side,
// and this is the edge between them.
int uzk_ctx_wait(uint64_t other) try {
    Ctx* cur = &ctx();
    Ctx* oth = nullptr;
    if (other == 0) oth = &default_ctx();
    else {
        Shared& s = shared();
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.contexts.find(other);
        if (it == s.contexts.end()) { set_error("uzk_ctx_wait: unknown context %llu", (unsigned long long)other); return UZK_ERR_PARAMETER; }
        oth = it->second;
    }
    if (oth == cur) return UZK_OK;
    std::unique_lock<std::mutex> l1(cur->mu, std::defer_lock), l2(oth->mu, std::defer_lock);
    std::lock(l1, l2);                                   // both locks, in whatever order avoids a deadlock with a thread waiting the other way
    UZK_TRY(require_ready());
    if (!oth->ready) return UZK_OK;                      // nothing was ever queued there
    hipEvent_t ev = oth->get_event();
    struct Return { Ctx* c; hipEvent_t e; ~Return() { c->event_pool.push_back(e); } } back{oth, ev};   // on every path, errors included
    UZK_HIP(hipEventRecord(ev, oth->stream));
    UZK_HIP(hipStreamWaitEvent(cur->stream, ev, 0));     // the wait has captured this recording; a later re-record does not move it
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ctx_wait"); }

/* ---- device memory ----------------------------------------------------------------------------
 * What a host language needs to keep data resident between the *_device entry points without linking the HIP runtime
 * itself.  Copies and fills are ordered on the calling context's stream, i.e. with that context's kernels. */
int uzk_dev_alloc(size_t bytes, void** d_out) try {
    API_LOCK;
    if (!d_out) { set_error("uzk_dev_alloc: null pointer"); return UZK_ERR_PARAMETER; }
    *d_out = nullptr;
    UZK_TRY(require_ready());
    if (bytes == 0) return UZK_OK;
    hipError_t e = hipMalloc(d_out, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();        // the runtime keeps a failed call as its sticky "last error": the next launch check must not see it
        *d_out = nullptr;
        set_error("uzk_dev_alloc(%zu): %s", bytes, hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_dev_alloc"); }
// Waits for the calling context's stream first: work queued on it may still use the block.
int uzk_dev_free(void* d_ptr) try {
    API_LOCK;
    if (!d_ptr) return UZK_OK;
    UZK_TRY(require_ready());
    UZK_HIP(hipStreamSynchronize(ctx().stream));
    UZK_HIP(hipFree(d_ptr));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_dev_free"); }
int uzk_host_alloc(size_t bytes, void** h_out) try {
    API_LOCK;
    if (!h_out) { set_error("uzk_host_alloc: null pointer"); return UZK_ERR_PARAMETER; }
    *h_out = nullptr;
    UZK_TRY(require_ready());
    if (bytes == 0) return UZK_OK;
    hipError_t e = hipHostMalloc(h_out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *h_out = nullptr; set_error("uzk_host_alloc(%zu): %s", bytes, hipGetErrorString(e)); return UZK_ERR_DEVICE; }
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    s.pinned[*h_out] = bytes;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_host_alloc"); }
int uzk_host_free(void* h_ptr) try {
    API_LOCK;
    if (!h_ptr) return UZK_OK;
    UZK_TRY(require_ready());
    Shared& s = shared();
    size_t bytes = 0;
    {
        // the entry goes BEFORE the memory does: once hipHostFree has run, another context's uzk_host_alloc may be handed the same
        // address and register it -- an erase after the free would then remove the new block's entry
        std::lock_guard<std::mutex> lk(s.mu);
        auto it = s.pinned.find(h_ptr);
        if (it == s.pinned.end()) { set_error("uzk_host_free: %p is not a block of uzk_host_alloc", h_ptr); return UZK_ERR_PARAMETER; }
        bytes = it->second;
        s.pinned.erase(it);
    }
    hipError_t e = hipStreamSynchronize(ctx().stream);
    if (e == hipSuccess) e = hipHostFree(h_ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(s.mu);
        s.pinned[h_ptr] = bytes;            // still allocated: still pinned, still freeable
        set_error("uzk_host_free: %s", hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_host_free"); }
}  // extern "C"
namespace uzk {
// true when [p, p + bytes) lies inside a block of uzk_host_alloc
bool is_pinned_block(const void* p, size_t bytes) {
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.pinned.upper_bound(p);
    if (it == s.pinned.begin()) return false;
    --it;
    const char* base = static_cast<const char*>(it->first);
    const char* q = static_cast<const char*>(p);
    return q >= base && q + bytes <= base + it->second;
}
}  // namespace uzk
extern "C" {
static int dev_copy_common(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, int kind, const char* who) {
    if (kind != UZK_COPY_H2D && kind != UZK_COPY_D2H && kind != UZK_COPY_D2D) { set_error("%s: kind must be UZK_COPY_H2D / D2H / D2D", who); return UZK_ERR_PARAMETER; }
    if (width == 0 || rows == 0) return UZK_OK;
    if (!dst || !src) { set_error("%s: null pointer", who); return UZK_ERR_PARAMETER; }
    if (rows > 1 && (dpitch < width || spitch < width)) { set_error("%s: a pitch is smaller than the row width", who); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    const hipMemcpyKind k = kind == UZK_COPY_H2D ? hipMemcpyHostToDevice : kind == UZK_COPY_D2H ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (rows == 1 || (dpitch == width && spitch == width)) UZK_HIP(hipMemcpyAsync(dst, src, width * rows, k, c.stream));
    else UZK_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, k, c.stream));
    // D2H: the data is in `dst` on return.  H2D: `src` may be reused on return -- unless it is pinned memory of
    // uzk_host_alloc, whose uploads stay asynchronous (the caller keeps it unchanged until the next synchronising call).
    const size_t span = rows > 1 ? (rows - 1) * spitch + width : width;
    if (kind == UZK_COPY_D2H || (kind == UZK_COPY_H2D && !is_pinned_block(src, span))) UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}
int uzk_dev_copy(void* dst, const void* src, size_t bytes, int kind) try {
    API_LOCK;
    return dev_copy_common(dst, bytes, src, bytes, bytes, 1, kind, "uzk_dev_copy");
} catch (...) { return uzk::on_exception("uzk_dev_copy"); }
int uzk_dev_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows, int kind) try {
    API_LOCK;
    return dev_copy_common(dst, dst_pitch, src, src_pitch, width, rows, kind, "uzk_dev_copy2d");
} catch (...) { return uzk::on_exception("uzk_dev_copy2d"); }
int uzk_dev_memset(void* d_dst, int byte, size_t bytes) try {
    API_LOCK;
    if (bytes == 0) return UZK_OK;
    if (!d_dst) { set_error("uzk_dev_memset: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    UZK_HIP(hipMemsetAsync(d_dst, byte, bytes, ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_dev_memset"); }
int uzk_dev_memset2d(void* d_dst, size_t pitch, int byte, size_t width, size_t rows) try {
    API_LOCK;
    if (width == 0 || rows == 0) return UZK_OK;
    if (!d_dst) { set_error("uzk_dev_memset2d: null pointer"); return UZK_ERR_PARAMETER; }
    if (rows > 1 && pitch < width) { set_error("uzk_dev_memset2d: pitch smaller than the row width"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    UZK_HIP(hipMemset2DAsync(d_dst, pitch, byte, width, rows, ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_dev_memset2d"); }

/* ---- SRS (process-wide registry: the bases are read-only and shared by every context) ------------ */
}  // extern "C"
namespace uzk {
uint64_t srs_insert(const Ctx::Srs& e) {
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    const uint64_t h = s.next_handle++;
    s.srs[h] = e;
    return h;
}
// copy of the registry entry (the pointers stay valid until uzk_srs_release / uzk_srs_precompute on that handle)
bool srs_lookup(uint64_t handle, Ctx::Srs* out) {
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.srs.find(handle);
    if (it == s.srs.end()) return false;
    *out = it->second;
    return true;
}
// removes the entry and hands it to the caller, who frees what it owns
bool srs_erase(uint64_t handle, Ctx::Srs* out) {
    Shared& s = shared();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.srs.find(handle);
    if (it == s.srs.end()) return false;
    *out = it->second;
    s.srs.erase(it);
    return true;
}
}  // namespace uzk
extern "C" {

int uzk_srs_register(const uzk_g1_affine* points, size_t n, uint64_t* handle_out) try {
    API_LOCK;
    if (!handle_out || (n > 0 && !points)) { set_error("uzk_srs_register: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Ctx::Srs s;
    s.n = n;
    s.owned = true;
    s.device = c.device;
    if (n > 0) {
        UZK_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_points), n * sizeof(Affine)));
        UZK_HIP(hipMemcpyAsync(s.d_points, points, n * sizeof(Affine), hipMemcpyHostToDevice, c.stream));
        UZK_HIP(hipStreamSynchronize(c.stream));
    }
    *handle_out = srs_insert(s);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_register"); }

int uzk_srs_register_device(const void* d_points, size_t n, uint64_t* handle_out) try {
    API_LOCK;
    if (!handle_out || (n > 0 && !d_points)) { set_error("uzk_srs_register_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx::Srs s;
    s.n = n;
    s.owned = false;
    s.device = ctx().device;
    s.d_points = const_cast<Affine*>(static_cast<const Affine*>(d_points));
    *handle_out = srs_insert(s);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_register_device"); }

// The caller makes sure no context still runs an MSM over this handle.
int uzk_srs_release(uint64_t handle) try {
    API_LOCK;
    Ctx& c = ctx();
    Shared& sh = shared();
    Ctx::Srs e;
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        auto it = sh.srs.find(handle);
        if (it == sh.srs.end()) { set_error("uzk_srs_release: unknown handle %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        e = it->second;
        sh.srs.erase(it);
    }
    if (c.ready) (void)hipStreamSynchronize(c.stream);
    (void)hipSetDevice(e.device);
    if (e.owned && e.d_points) (void)hipFree(e.d_points);
    if (e.d_table) (void)hipFree(e.d_table);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_release"); }

// Build the table before contexts start using the handle concurrently (the entry is replaced, not versioned).
int uzk_srs_precompute(uint64_t handle, int window_bits) try {
    API_LOCK;
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Ctx::Srs s;
    if (!srs_lookup(handle, &s)) { set_error("uzk_srs_precompute: unknown handle"); return UZK_ERR_PARAMETER; }
    if (window_bits != 0 && (window_bits < 4 || window_bits > 24)) {
        set_error("uzk_srs_precompute: window bits must be 0 (auto) or 4..24");
        return UZK_ERR_PARAMETER;
    }
    if (s.n == 0) return UZK_OK;
    if (s.device != c.device) { set_error("uzk_srs_precompute: the SRS lives on device %d, the calling context on device %d", s.device, c.device); return UZK_ERR_PARAMETER; }
    const int cb = msm_precompute_window_bits(s.n, window_bits);
    if (s.d_table && s.pre_c == cb) return UZK_OK;
    Affine* old_table = s.d_table;
    Affine* table = nullptr;
    uint32_t W = 0;
    UZK_TRY(msm_build_table(c, s.d_points, s.n, cb, &table, &W));
    {
        Shared& sh = shared();
        std::lock_guard<std::mutex> lk(sh.mu);
        auto it = sh.srs.find(handle);
        if (it == sh.srs.end()) { (void)hipFree(table); set_error("uzk_srs_precompute: handle released meanwhile"); return UZK_ERR_PARAMETER; }
        it->second.d_table = table;
        it->second.pre_c = cb;
        it->second.pre_W = W;
    }
    if (old_table) { (void)hipStreamSynchronize(c.stream); (void)hipFree(old_table); }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_precompute"); }

int uzk_srs_len(uint64_t handle, size_t* n_out) try {
    Ctx::Srs s;
    if (!n_out || !srs_lookup(handle, &s)) { set_error("uzk_srs_len: unknown handle"); return UZK_ERR_PARAMETER; }
    *n_out = s.n;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_len"); }

/* ---- MSM ---------------------------------------------------------------------------------- */
static int msm_checked(uint64_t srs_handle, size_t offset, size_t n, Ctx::Srs* srs_out) {
    if (!srs_lookup(srs_handle, srs_out)) { set_error("msm: unknown SRS handle %llu", (unsigned long long)srs_handle); return UZK_ERR_PARAMETER; }
    if (offset > srs_out->n || n > srs_out->n - offset) {
        // KZG commit: degree + 1 > SRS length (kzg_poly_commitment.rs:283-285)
        set_error("msm: offset %zu + n %zu exceeds SRS length %zu", offset, n, srs_out->n);
        return UZK_ERR_DEGREE;
    }
    if (srs_out->device != ctx().device) { set_error("msm: the SRS lives on device %d, the calling context on device %d", srs_out->device, ctx().device); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}
}  // extern "C"
namespace uzk {
// general mode unless the handle carries a window table
int msm_dispatch_view(const Ctx::Srs& s, size_t offset, const ScalarView& sv, size_t n, uint32_t batch, Jac* out) {
    Ctx& c = ctx();
    if (s.d_table && !c.tune_no_precompute)
        return msm_run(c, s.d_table, sv, n, batch, out, s.pre_c, (uint32_t)s.n, (uint32_t)offset);
    return msm_run(c, s.d_points + offset, sv, n, batch, out, 0, 0, 0);
}
}  // namespace uzk
extern "C" {
static int msm_dispatch_one(const Ctx::Srs& s, size_t offset, const Fp* d_scalars, size_t n, uint32_t batch, Jac* out) {
    Ctx& c = ctx();
    // one vector of more than 2^24 points over plain bases: chunks of 2^24 into one bucket set (msm_run_chunked)
    if (batch == 1 && n > ((size_t)1 << 24) && !(s.d_table && !c.tune_no_precompute) && c.tune_chunk_log >= 25)
        return msm_run_chunked(c, s.d_points + offset, d_scalars, n, out);
    return msm_dispatch_view(s, offset, ScalarView::dense(d_scalars, n), n, batch, out);
}
// The sort indexes (point, window) pairs with 31 bits, so one pass handles at most 2^26 points per
// scalar vector (2^26 * 16 windows); longer inputs are cut into point chunks whose partial sums are
// folded on the host -- the same decomposition the multi-GPU path uses across ranks.
static int msm_dispatch(const Ctx::Srs& s, size_t offset, const Fp* d_scalars, size_t n, uint32_t batch, Jac* out) {
    const size_t kChunk = (size_t)1 << ctx().tune_chunk_log;
    if (n <= kChunk || batch != 1) return msm_dispatch_one(s, offset, d_scalars, n, batch, out);
    XYZZ acc = xyzz_inf();
    for (size_t lo = 0; lo < n; lo += kChunk) {
        const size_t len = std::min(kChunk, n - lo);
        Jac part;
        UZK_TRY(msm_dispatch_one(s, offset + lo, d_scalars + lo, len, 1, &part));
        XYZZ q = xyzz_from_jac(part);
        xyzz_add(acc, q);
    }
    *out = xyzz_to_jac(acc);
    return UZK_OK;
}

int uzk_msm_g1_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t n, uzk_g1_jac* out) try {
    API_LOCK;
    if (!out || (n > 0 && !d_scalars_mont)) { set_error("uzk_msm_g1_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx::Srs srs;
    UZK_TRY(msm_checked(srs_handle, offset, n, &srs));
    Jac r;
    UZK_TRY(msm_dispatch(srs, offset, static_cast<const Fp*>(d_scalars_mont), n, 1, &r));
    std::memcpy(out, &r, sizeof r);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_g1_device"); }

int uzk_msm_g1(uint64_t srs_handle, size_t offset, const uint64_t* scalars_mont, size_t n, uzk_g1_jac* out) try {
    API_LOCK;
    if (!out || (n > 0 && !scalars_mont)) { set_error("uzk_msm_g1: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Ctx::Srs srs;
    UZK_TRY(msm_checked(srs_handle, offset, n, &srs));
    Jac r = jac_inf();
    const bool table = srs.d_table && !c.tune_no_precompute;
    if (n >= ((size_t)1 << c.tune_stream_min_log) && n <= ((size_t)1 << c.tune_chunk_log) && !table && c.tune_stream_log >= 0) {
        // large general-mode MSM: the scalars arrive in point chunks under the previous chunk's accumulation (msm.hip)
        UZK_TRY(msm_run_streamed(c, srs.d_points + offset, as_fp(scalars_mont), n, &r));
    } else if (n > 0) {
        UZK_TRY(c.msm_scalars.reserve(n * sizeof(Fp)));
        UZK_HIP(hipMemcpyAsync(c.msm_scalars.p, scalars_mont, n * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
        UZK_TRY(msm_dispatch(srs, offset, c.msm_scalars.as<Fp>(), n, 1, &r));
    }
    std::memcpy(out, &r, sizeof r);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_g1"); }

int uzk_msm_g1_batch_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t n, uint32_t batch,
                            uzk_g1_jac* out) try {
    API_LOCK;
    if (batch > 0 && (!out || (n > 0 && !d_scalars_mont))) { set_error("uzk_msm_g1_batch_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx::Srs srs;
    UZK_TRY(msm_checked(srs_handle, offset, n, &srs));
    std::vector<Jac> r(batch);
    UZK_TRY(msm_dispatch(srs, offset, static_cast<const Fp*>(d_scalars_mont), n, batch, r.data()));
    if (batch) std::memcpy(out, r.data(), (size_t)batch * sizeof(Jac));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_g1_batch_device"); }

int uzk_msm_g1_batch(uint64_t srs_handle, size_t offset, const uint64_t* scalars_mont, size_t n, uint32_t batch,
                     uzk_g1_jac* out) try {
    API_LOCK;
    if (batch > 0 && (!out || (n > 0 && !scalars_mont))) { set_error("uzk_msm_g1_batch: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Ctx::Srs srs;
    UZK_TRY(msm_checked(srs_handle, offset, n, &srs));
    std::vector<Jac> r(batch, jac_inf());
    if (n > 0 && batch > 0) {
        const size_t bytes = (size_t)n * batch * sizeof(Fp);
        UZK_TRY(c.msm_scalars.reserve(bytes));
        UZK_HIP(hipMemcpyAsync(c.msm_scalars.p, scalars_mont, bytes, hipMemcpyHostToDevice, c.stream));
        UZK_TRY(msm_dispatch(srs, offset, c.msm_scalars.as<Fp>(), n, batch, r.data()));
    }
    if (batch) std::memcpy(out, r.data(), (size_t)batch * sizeof(Jac));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_g1_batch"); }

// out[b] = sum_{i<n} d_scalars[b*stride + i] * SRS[offset + i] + sum_{j<tail_n} tail[b*tail_n + j] * SRS[offset + n + j]
int uzk_msm_g1_batch_tail_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t stride, size_t n, uint32_t batch,
                                 const void* tail_scalars_mont, uint32_t tail_n, int tail_on_device, uzk_g1_jac* out) try {
    API_LOCK;
    if (batch > 0 && (!out || (n > 0 && !d_scalars_mont) || (tail_n > 0 && !tail_scalars_mont))) { set_error("uzk_msm_g1_batch_tail_device: null pointer"); return UZK_ERR_PARAMETER; }
    if (batch > 1 && stride < n) { set_error("uzk_msm_g1_batch_tail_device: stride %zu smaller than n %zu", stride, n); return UZK_ERR_PARAMETER; }
    if (tail_n > 4096) { set_error("uzk_msm_g1_batch_tail_device: tail_n %u exceeds 4096", tail_n); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    const size_t total = n + tail_n;
    Ctx::Srs srs;
    UZK_TRY(msm_checked(srs_handle, offset, total, &srs));
    if (total > ((size_t)1 << c.tune_chunk_log)) { set_error("uzk_msm_g1_batch_tail_device: n + tail_n = %zu exceeds one sort pass (2^%d points)", total, c.tune_chunk_log); return UZK_ERR_PARAMETER; }
    if (batch == 0) return UZK_OK;
    ScalarView sv;
    sv.main = static_cast<const Fp*>(d_scalars_mont);
    sv.stride = stride;
    sv.n_main = (uint32_t)n;
    sv.tail_n = tail_n;
    if (tail_n && tail_on_device) sv.tail = static_cast<const Fp*>(tail_scalars_mont);
    else if (tail_n) {
        // a few dozen elements: copied into pinned, device-visible memory the digit kernel reads directly; the call returns
        // after the window sums have arrived, so one slot per context is enough
        const size_t bytes = (size_t)batch * tail_n * sizeof(Fp);
        if (c.msm_tail_cap < bytes) {
            if (c.msm_tail_host) { UZK_HIP(hipStreamSynchronize(c.stream)); (void)hipHostFree(c.msm_tail_host); c.msm_tail_host = nullptr; c.msm_tail_cap = 0; }
            const size_t cap = std::max<size_t>(bytes, 1 << 14);
            UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.msm_tail_host), cap, hipHostMallocDefault));
            c.msm_tail_cap = cap;
        }
        std::memcpy(c.msm_tail_host, tail_scalars_mont, bytes);
        sv.tail = c.msm_tail_host;
    }
    std::vector<Jac> r(batch);
    UZK_TRY(msm_dispatch_view(srs, offset, sv, total, batch, r.data()));
    std::memcpy(out, r.data(), (size_t)batch * sizeof(Jac));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_g1_batch_tail_device"); }

int uzk_msm_g1_raw(const uzk_g1_affine* points, const uint64_t* scalars_mont, size_t n, uzk_g1_jac* out) try {
    if (!out || (n > 0 && (!points || !scalars_mont))) { set_error("uzk_msm_g1_raw: null pointer"); return UZK_ERR_PARAMETER; }
    uint64_t h = 0;
    UZK_TRY(uzk_srs_register(points, n, &h));
    int rc = uzk_msm_g1(h, 0, scalars_mont, n, out);
    (void)uzk_srs_release(h);
    return rc;
} catch (...) { return uzk::on_exception("uzk_msm_g1_raw"); }

int uzk_g1_fold(const uzk_g1_jac* partials, size_t count, uzk_g1_jac* out) try {
    if (!out || (count > 0 && !partials)) { set_error("uzk_g1_fold: null pointer"); return UZK_ERR_PARAMETER; }
    h64::J acc = h64::j_inf();
    for (size_t i = 0; i < count; ++i) {
        Jac j;
        std::memcpy(&j, &partials[i], sizeof j);
        acc = h64::j_add(acc, h64::j_from(j));
    }
    const Jac r = h64::j_to(acc);
    std::memcpy(out, &r, sizeof r);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_g1_fold"); }

int uzk_g1_to_affine(const uzk_g1_jac* p, uzk_g1_affine* out) try {
    if (!p || !out) { set_error("uzk_g1_to_affine: null pointer"); return UZK_ERR_PARAMETER; }
    Jac j;
    std::memcpy(&j, p, sizeof j);
    Affine a = jac_to_affine_host(j);
    std::memcpy(out, &a, sizeof a);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_g1_to_affine"); }

/* ---- NTT ---------------------------------------------------------------------------------- */
int uzk_domain_supported(uint64_t n) try { return domain_supported(n) ? 1 : 0; } catch (...) { return uzk::on_exception("uzk_domain_supported"); }

int uzk_domain_group_gen(uint64_t n, uint64_t out_mont[4]) try {
    if (!out_mont) { set_error("uzk_domain_group_gen: null pointer"); return UZK_ERR_PARAMETER; }
    if (!domain_exists(n)) { set_error("no evaluation domain of size %llu", (unsigned long long)n); return UZK_ERR_FFT; }
    Fp w = fr_root_of_unity(n);
    std::memcpy(out_mont, &w, sizeof w);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_domain_group_gen"); }

static int ntt_device_common(const void* d_in, void* d_out, uint64_t n, uint32_t batch, int inverse,
                             const uint64_t* coset_shift_mont, int sync) {
    if (!domain_supported(n)) {
        set_error("no evaluation domain of size %llu (need 2^k, k <= %d, or 3 * 2^k, k <= %d)", (unsigned long long)n, UZK_NTT_MAX_LOG2, UZK_NTT_MAX_LOG2_MIXED);
        return UZK_ERR_FFT;
    }
    if (!d_in || !d_out) { set_error("uzk_ntt_fr*_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    UZK_TRY(ntt_run(c, static_cast<const Fp*>(d_in), static_cast<Fp*>(d_out), n, inverse != 0,
                    coset_shift_mont ? as_fp(coset_shift_mont) : nullptr, batch));
    if (sync) UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}
static int ntt_host_common(uint64_t* data, uint64_t n, uint32_t batch, int inverse, const uint64_t* coset_shift_mont) {
    if (!domain_supported(n)) {
        set_error("no evaluation domain of size %llu (need 2^k, k <= %d, or 3 * 2^k, k <= %d)", (unsigned long long)n, UZK_NTT_MAX_LOG2, UZK_NTT_MAX_LOG2_MIXED);
        return UZK_ERR_FFT;
    }
    if (!data) { set_error("uzk_ntt_fr*: null pointer"); return UZK_ERR_PARAMETER; }
    if (batch == 0) return UZK_OK;
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    const size_t bytes = (size_t)n * batch * sizeof(Fp);
    UZK_TRY(c.ntt_io.reserve(bytes));
    UZK_HIP(hipMemcpyAsync(c.ntt_io.p, data, bytes, hipMemcpyHostToDevice, c.stream));
    UZK_TRY(ntt_run(c, c.ntt_io.as<Fp>(), c.ntt_io.as<Fp>(), n, inverse != 0,
                    coset_shift_mont ? as_fp(coset_shift_mont) : nullptr, batch));
    UZK_HIP(hipMemcpyAsync(data, c.ntt_io.p, bytes, hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}

int uzk_ntt_fr_device(const void* d_in, void* d_out, uint64_t n, int inverse, const uint64_t* coset_shift_mont, int sync) try {
    API_LOCK;
    return ntt_device_common(d_in, d_out, n, 1, inverse, coset_shift_mont, sync);
} catch (...) { return uzk::on_exception("uzk_ntt_fr_device"); }
int uzk_ntt_fr_batch_device(const void* d_in, void* d_out, uint64_t n, uint32_t batch, int inverse,
                            const uint64_t* coset_shift_mont, int sync) try {
    API_LOCK;
    return ntt_device_common(d_in, d_out, n, batch, inverse, coset_shift_mont, sync);
} catch (...) { return uzk::on_exception("uzk_ntt_fr_batch_device"); }
int uzk_ntt_fr_batch_strided_device(const void* d_in, uint64_t in_stride, void* d_out, uint64_t out_stride, uint64_t n, uint32_t batch,
                                    int inverse, const uint64_t* coset_shift_mont, int sync) try {
    API_LOCK;
    if (!domain_supported(n)) {
        set_error("no evaluation domain of size %llu (need 2^k, k <= %d, or 3 * 2^k, k <= %d)", (unsigned long long)n, UZK_NTT_MAX_LOG2, UZK_NTT_MAX_LOG2_MIXED);
        return UZK_ERR_FFT;
    }
    if (!d_in || !d_out) { set_error("uzk_ntt_fr_batch_strided_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    UZK_TRY(ntt_run(c, static_cast<const Fp*>(d_in), static_cast<Fp*>(d_out), n, inverse != 0,
                    coset_shift_mont ? as_fp(coset_shift_mont) : nullptr, batch, in_stride, out_stride));
    if (sync) UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_ntt_fr_batch_strided_device"); }
int uzk_ntt_fr(uint64_t* data, uint64_t n, int inverse, const uint64_t* coset_shift_mont) try {
    API_LOCK;
    return ntt_host_common(data, n, 1, inverse, coset_shift_mont);
} catch (...) { return uzk::on_exception("uzk_ntt_fr"); }
int uzk_ntt_fr_batch(uint64_t* data, uint64_t n, uint32_t batch, int inverse, const uint64_t* coset_shift_mont) try {
    API_LOCK;
    return ntt_host_common(data, n, batch, inverse, coset_shift_mont);
} catch (...) { return uzk::on_exception("uzk_ntt_fr_batch"); }

/* ---- polynomial helpers next to the hot path ------------------------------------------------ */
int uzk_poly_eval_batch(const uint64_t* coefs, uint64_t n, uint32_t batch, const uint64_t* x_mont, uint64_t* out) try {
    API_LOCK;
    if (!x_mont || (batch > 0 && (!out || (n > 0 && !coefs)))) { set_error("uzk_poly_eval_batch: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_eval_batch_host(ctx(), as_fp(coefs), n, batch, *as_fp(x_mont), reinterpret_cast<Fp*>(out));
} catch (...) { return uzk::on_exception("uzk_poly_eval_batch"); }
int uzk_poly_eval_batch_device(const void* d_coefs, uint64_t n, uint32_t batch, const uint64_t* x_mont, uint64_t* out) try {
    API_LOCK;
    if (!x_mont || (batch > 0 && (!out || (n > 0 && !d_coefs)))) { set_error("uzk_poly_eval_batch_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_eval_batch(ctx(), static_cast<const Fp*>(d_coefs), n, batch, *as_fp(x_mont), reinterpret_cast<Fp*>(out));
} catch (...) { return uzk::on_exception("uzk_poly_eval_batch_device"); }
int uzk_z_poly_device(const void* d_w, const uint32_t* d_perm, const void* d_group, const uint64_t* k, const uint64_t* beta_mont,
                      const uint64_t* gamma_mont, uint32_t n, uint32_t n_wires, void* d_z) try {
    API_LOCK;
    if (n > 0 && (!d_w || !d_perm || !d_group || !k || !beta_mont || !gamma_mont || !d_z)) { set_error("uzk_z_poly_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return z_poly_device(ctx(), static_cast<const Fp*>(d_w), d_perm, static_cast<const Fp*>(d_group), as_fp(k), *as_fp(beta_mont),
                         *as_fp(gamma_mont), n, n_wires, static_cast<Fp*>(d_z));
} catch (...) { return uzk::on_exception("uzk_z_poly_device"); }

int uzk_open_quotient_device(const void* d_polys, uint64_t n, uint32_t batch, const uint64_t* z_mont,
                             const uint64_t* alpha_mont, void* d_q, uint64_t* evals_out) try {
    API_LOCK;
    if (!d_polys || !z_mont || !alpha_mont || !d_q || !evals_out) { set_error("uzk_open_quotient_device: null pointer"); return UZK_ERR_PARAMETER; }
    if (d_q == d_polys) { set_error("uzk_open_quotient_device: output aliases the input"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return open_quotient_run(ctx(), static_cast<const Fp*>(d_polys), n, batch, *as_fp(z_mont), *as_fp(alpha_mont),
                             static_cast<Fp*>(d_q), reinterpret_cast<Fp*>(evals_out));
} catch (...) { return uzk::on_exception("uzk_open_quotient_device"); }

int uzk_open_quotient(const uint64_t* polys, uint64_t n, uint32_t batch, const uint64_t* z_mont, const uint64_t* alpha_mont,
                      uint64_t* q_out, uint64_t* evals_out) try {
    API_LOCK;
    if (!polys || !z_mont || !alpha_mont || !q_out || !evals_out) { set_error("uzk_open_quotient: null pointer"); return UZK_ERR_PARAMETER; }
    if (n == 0 || batch == 0) { set_error("uzk_open_quotient: need batch > 0 and n > 0"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    const size_t in_bytes = (size_t)n * batch * sizeof(Fp);
    UZK_TRY(c.poly_io.reserve(in_bytes + (size_t)n * sizeof(Fp)));
    Fp* d_p = c.poly_io.as<Fp>();
    Fp* d_q = d_p + (size_t)n * batch;
    UZK_HIP(hipMemcpyAsync(d_p, polys, in_bytes, hipMemcpyHostToDevice, c.stream));
    UZK_TRY(open_quotient_run(c, d_p, n, batch, *as_fp(z_mont), *as_fp(alpha_mont), d_q, reinterpret_cast<Fp*>(evals_out)));
    UZK_HIP(hipMemcpyAsync(q_out, d_q, (size_t)n * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_open_quotient"); }

int uzk_fold_blinds_device(const void* d_coefs, uint64_t len, uint64_t n_fold, void* d_out, uint64_t* blinds_out) try {
    API_LOCK;
    if (!d_coefs || !d_out || (len > n_fold && !blinds_out)) { set_error("uzk_fold_blinds_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return fold_blinds_run(ctx(), static_cast<const Fp*>(d_coefs), len, n_fold, static_cast<Fp*>(d_out), reinterpret_cast<Fp*>(blinds_out));
} catch (...) { return uzk::on_exception("uzk_fold_blinds_device"); }

int uzk_poly_lincomb_device(const void* const* d_polys, const uint64_t* lens, const uint64_t* scalars_mont, uint32_t count,
                            void* d_out, uint64_t out_len) try {
    API_LOCK;
    if (!d_polys || !lens || !scalars_mont || (out_len > 0 && !d_out)) { set_error("uzk_poly_lincomb_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_lincomb_run(ctx(), d_polys, lens, as_fp(scalars_mont), count, static_cast<Fp*>(d_out), out_len);
} catch (...) { return uzk::on_exception("uzk_poly_lincomb_device"); }

int uzk_hide_polynomial_device(void* d_coefs, uint64_t len, const uint64_t* blinds_mont, uint32_t hiding_degree, uint64_t zeroing_degree) try {
    API_LOCK;
    if (!d_coefs || (hiding_degree > 0 && !blinds_mont)) { set_error("uzk_hide_polynomial_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_hide_run(ctx(), static_cast<Fp*>(d_coefs), len, as_fp(blinds_mont), hiding_degree, zeroing_degree);
} catch (...) { return uzk::on_exception("uzk_hide_polynomial_device"); }

int uzk_hide_polynomial_batch_device(void* d_coefs, uint64_t stride, uint64_t len_in, uint32_t count, const uint64_t* blinds_mont,
                                     uint32_t hiding_degree, uint64_t zeroing_degree) try {
    API_LOCK;
    if (count > 0 && hiding_degree > 0 && (!d_coefs || !blinds_mont)) { set_error("uzk_hide_polynomial_batch_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_hide_batch_run(ctx(), static_cast<Fp*>(d_coefs), stride, len_in, count, as_fp(blinds_mont), hiding_degree, zeroing_degree);
} catch (...) { return uzk::on_exception("uzk_hide_polynomial_batch_device"); }

int uzk_fold_blinds_batch_device(const void* d_polys, uint64_t in_stride, const uint64_t* lens, uint64_t n_fold, uint32_t batch, void* d_out,
                                 uint64_t out_stride, void* d_tail, uint32_t tail_n, uint64_t* blinds_out) try {
    API_LOCK;
    if (batch > 0 && (!d_polys || !lens || !d_out || (tail_n > 0 && !d_tail))) { set_error("uzk_fold_blinds_batch_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return fold_blinds_batch_run(ctx(), static_cast<const Fp*>(d_polys), in_stride, lens, n_fold, batch, static_cast<Fp*>(d_out), out_stride,
                                 static_cast<Fp*>(d_tail), tail_n, reinterpret_cast<Fp*>(blinds_out));
} catch (...) { return uzk::on_exception("uzk_fold_blinds_batch_device"); }

int uzk_poly_trimmed_len_device(const void* d_polys, uint64_t stride, const uint64_t* lens, uint32_t batch, uint64_t* out_lens, int sync) try {
    API_LOCK;
    if (batch > 0 && (!d_polys || !lens || !out_lens)) { set_error("uzk_poly_trimmed_len_device: null pointer"); return UZK_ERR_PARAMETER; }
    if (!sync && batch > 0 && !is_pinned_block(out_lens, batch * sizeof(uint64_t))) {
        set_error("uzk_poly_trimmed_len_device: an asynchronous call needs out_lens in uzk_host_alloc memory");
        return UZK_ERR_PARAMETER;
    }
    UZK_TRY(require_ready());
    return poly_trimmed_len_run(ctx(), static_cast<const Fp*>(d_polys), stride, lens, batch, out_lens, sync != 0);
} catch (...) { return uzk::on_exception("uzk_poly_trimmed_len_device"); }

int uzk_split_t_device(const void* d_t, uint64_t t_len, uint64_t chunk, uint32_t n_chunks, const uint64_t* rands_mont, void* d_chunks,
                       uint64_t chunk_stride, uint64_t* lens_out) try {
    API_LOCK;
    if (!d_t || !rands_mont || !d_chunks) { set_error("uzk_split_t_device: null pointer"); return UZK_ERR_PARAMETER; }
    if (d_t == d_chunks) { set_error("uzk_split_t_device: output aliases the input"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return split_t_run(ctx(), static_cast<const Fp*>(d_t), t_len, chunk, n_chunks, as_fp(rands_mont), static_cast<Fp*>(d_chunks), chunk_stride, lens_out);
} catch (...) { return uzk::on_exception("uzk_split_t_device"); }

int uzk_poly_eval_ptrs_device(const void* const* d_polys, const uint64_t* lens, const uint32_t* point_idx, uint32_t count,
                              const uint64_t* points_mont, uint32_t n_points, uint64_t* out) try {
    API_LOCK;
    if (count > 0 && (!d_polys || !lens || !point_idx || !points_mont || !out)) { set_error("uzk_poly_eval_ptrs_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return poly_eval_ptrs(ctx(), d_polys, lens, point_idx, count, as_fp(points_mont), n_points, reinterpret_cast<Fp*>(out));
} catch (...) { return uzk::on_exception("uzk_poly_eval_ptrs_device"); }

int uzk_open_quotient_ptrs_device(const void* const* d_polys, const uint64_t* lens, uint32_t count, const uint64_t* z_mont,
                                  const uint64_t* alpha_mont, void* d_q, uint64_t q_cap, uint64_t* evals_out) try {
    API_LOCK;
    if (!d_polys || !lens || !z_mont || !alpha_mont || !d_q) { set_error("uzk_open_quotient_ptrs_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return open_quotient_ptrs(ctx(), d_polys, lens, count, *as_fp(z_mont), *as_fp(alpha_mont), static_cast<Fp*>(d_q), q_cap,
                              reinterpret_cast<Fp*>(evals_out));
} catch (...) { return uzk::on_exception("uzk_open_quotient_ptrs_device"); }

int uzk_t_quotient_device(const uzk_quotient_args* args, void* d_out, int sync) try {
    API_LOCK;
    if (!args || !d_out) { set_error("uzk_t_quotient_device: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    static_assert(sizeof(uzk_quotient_args) == 8 + 56 * sizeof(void*) + (3 + 5 + 3 + 16) * 32, "ABI layout");
    UZK_TRY(t_quotient_run(ctx(), args, static_cast<Fp*>(d_out)));
    if (sync) UZK_HIP(hipStreamSynchronize(ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_t_quotient_device"); }

int uzk_z_poly(const uint64_t* w, const uint32_t* perm, const uint64_t* group, const uint64_t* k, const uint64_t* beta_mont,
               const uint64_t* gamma_mont, uint32_t n, uint32_t n_wires, uint64_t* z_out) try {
    API_LOCK;
    if (!w || !perm || !group || !k || !beta_mont || !gamma_mont || !z_out) { set_error("uzk_z_poly: null pointer"); return UZK_ERR_PARAMETER; }
    if (n_wires == 0 || n_wires > 8 || (uint64_t)n * n_wires >= (1ull << 32)) { set_error("uzk_z_poly: bad shape"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return z_poly_run(ctx(), as_fp(w), perm, as_fp(group), as_fp(k), *as_fp(beta_mont), *as_fp(gamma_mont), n, n_wires,
                      reinterpret_cast<Fp*>(z_out));
} catch (...) { return uzk::on_exception("uzk_z_poly"); }

/* ---- synthetic workloads ------------------------------------------------------------------ */
int uzk_synth_points_arith(void* d_points, size_t n, const uint64_t* seed_scalar_mont) try {
    API_LOCK;
    if ((n > 0 && !d_points) || !seed_scalar_mont) { set_error("uzk_synth_points_arith: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return synth_points_arith(ctx(), static_cast<Affine*>(d_points), n, *as_fp(seed_scalar_mont));
} catch (...) { return uzk::on_exception("uzk_synth_points_arith"); }
int uzk_synth_points_random(void* d_points, size_t n, uint64_t seed) try {
    API_LOCK;
    if (n > 0 && !d_points) { set_error("uzk_synth_points_random: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    UZK_TRY(synth_points_random(ctx(), static_cast<Affine*>(d_points), n, seed));
    UZK_HIP(hipStreamSynchronize(ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_synth_points_random"); }
int uzk_synth_scalars(void* d_scalars, size_t n, uint64_t seed) try {
    API_LOCK;
    if (n > 0 && !d_scalars) { set_error("uzk_synth_scalars: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    UZK_TRY(synth_scalars(ctx(), static_cast<Fp*>(d_scalars), n, seed));
    UZK_HIP(hipStreamSynchronize(ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_synth_scalars"); }

int uzk_synth_scalars_mix(void* d_scalars, size_t n, uint64_t seed) try {
    API_LOCK;
    if (n > 0 && !d_scalars) { set_error("uzk_synth_scalars_mix: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    UZK_TRY(synth_scalars_mix(ctx(), static_cast<Fp*>(d_scalars), n, seed));
    UZK_HIP(hipStreamSynchronize(ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_synth_scalars_mix"); }

/* ---- element-wise field arithmetic on host arrays (product: the host mirrors' helper) ----------- */
int uzk_field_op_device(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) try {
    API_LOCK;
    if (n > 0 && (!a || !b || !out)) { set_error("uzk_field_op_device: null pointer"); return UZK_ERR_PARAMETER; }
    const bool known = op == 0 || op == 1 || op == 2 || op == 4 || op == 5 || op == 6 || op == 7;
    if (field < 0 || field > 1 || !known) { set_error("uzk_field_op_device: bad field/op (mul, add, sub, sqr, neg, from_mont, to_mont)"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return field_op_device(ctx(), field, op, as_fp(a), as_fp(b), reinterpret_cast<Fp*>(out), n);
} catch (...) { return uzk::on_exception("uzk_field_op_device"); }

/* ---- test hooks (include/uzkge_gpu_test.h): known-answer entry points of the arithmetic cores ---- */
int uzk_test_field_kat(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) try {
    API_LOCK;
    if (n > 0 && (!a || !b || !out)) { set_error("uzk_test_field_kat: null pointer"); return UZK_ERR_PARAMETER; }
    if (field < 0 || field > 1 || op < 0 || op > 33) { set_error("uzk_test_field_kat: bad field/op"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return field_op_device(ctx(), field, op, as_fp(a), as_fp(b), reinterpret_cast<Fp*>(out), n);
} catch (...) { return uzk::on_exception("uzk_test_field_kat"); }
int uzk_test_g1_kat(int op, const uzk_g1_affine* a, const uzk_g1_affine* b, uzk_g1_jac* out, size_t n) try {
    API_LOCK;
    if (n > 0 && (!a || !b || !out)) { set_error("uzk_test_g1_kat: null pointer"); return UZK_ERR_PARAMETER; }
    if (op < 0 || op > 21) { set_error("uzk_test_g1_kat: bad op"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return g1_op_device(ctx(), op, reinterpret_cast<const Affine*>(a), reinterpret_cast<const Affine*>(b),
                        reinterpret_cast<Jac*>(out), n);
} catch (...) { return uzk::on_exception("uzk_test_g1_kat"); }

/* ---- measurement -------------------------------------------------------------------------- */
int uzk_profile_enable(int on) try {
    API_LOCK;
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    UZK_TRY(c.prof_collect());
    c.prof_on = on != 0;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_profile_enable"); }
int uzk_profile_reset(void) try {
    API_LOCK;
    Ctx& c = ctx();
    if (c.ready) UZK_TRY(c.prof_collect());
    c.prof_totals.clear();
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_profile_reset"); }
int uzk_profile_get(const char* name, double* total_ms, uint64_t* launches) try {
    API_LOCK;
    Ctx& c = ctx();
    if (!name) { set_error("uzk_profile_get: null name"); return UZK_ERR_PARAMETER; }
    if (c.ready) UZK_TRY(c.prof_collect());
    auto it = c.prof_totals.find(name);
    if (total_ms) *total_ms = (it == c.prof_totals.end()) ? 0.0 : it->second.first;
    if (launches) *launches = (it == c.prof_totals.end()) ? 0 : it->second.second;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_profile_get"); }
int uzk_profile_dump(char* buf, size_t cap) try {
    API_LOCK;
    Ctx& c = ctx();
    if (!buf || cap == 0) { set_error("uzk_profile_dump: null buffer"); return UZK_ERR_PARAMETER; }
    if (c.ready) UZK_TRY(c.prof_collect());
    size_t off = 0;
    buf[0] = 0;
    for (auto& kv : c.prof_totals) {
        int w = snprintf(buf + off, cap - off, "%s %llu %.6f\n", kv.first.c_str(),
                         (unsigned long long)kv.second.second, kv.second.first);
        if (w < 0 || (size_t)w >= cap - off) break;
        off += (size_t)w;
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_profile_dump"); }
int uzk_sync(void) try {
    API_LOCK;
    UZK_TRY(require_ready());
    UZK_HIP(hipStreamSynchronize(ctx().stream));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_sync"); }
void* uzk_stream(void) {
    API_LOCK;
    if (require_ready() != UZK_OK) return nullptr;
    return ctx().stream;
}
int uzk_msm_set_window_bits(int c) try {
    API_LOCK;
    if (c != 0 && (c < 4 || c > 22)) { set_error("window bits must be 0 (auto) or 4..22"); return UZK_ERR_PARAMETER; }
    ctx().msm_window_bits = c;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_set_window_bits"); }

int uzk_msm_plan_info(size_t n, int* window_bits, int* windows) try {
    API_LOCK;
    if (!window_bits || !windows || n == 0) { set_error("uzk_msm_plan_info: bad arguments"); return UZK_ERR_PARAMETER; }
    msm_plan_info(ctx(), n, window_bits, windows);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_msm_plan_info"); }

int uzk_tune(const char* key, int value) try {
    API_LOCK;
    if (!key) { set_error("uzk_tune: null key"); return UZK_ERR_PARAMETER; }
    Ctx& c = ctx();
    if (!std::strcmp(key, "msm_no_precompute")) c.tune_no_precompute = value;
    else if (!std::strcmp(key, "msm_stream_log")) c.tune_stream_log = (value >= -1 && value <= 26) ? value : 0;
    else if (!std::strcmp(key, "msm_stream_min_log")) c.tune_stream_min_log = (value >= 4 && value <= 26) ? value : 22;
    else if (!std::strcmp(key, "msm_chunk_log")) c.tune_chunk_log = (value >= 8 && value <= 26) ? value : 26;
    else if (!std::strcmp(key, "msm_small")) c.tune_small = value ? 1 : 0;
    else if (!std::strcmp(key, "msm_seg_sort")) c.tune_seg_sort = value;
    else if (!std::strcmp(key, "ntt_tile")) c.tune_ntt_tile = (value == 1024 || value == 2048) ? value : 0;
    else if (!std::strcmp(key, "ntt_two_pass")) c.tune_ntt_two_pass = value ? 1 : 0;
    else if (!std::strcmp(key, "arith29")) c.tune_arith29 = value & 7;
    else { set_error("uzk_tune: unknown key %s", key); return UZK_ERR_PARAMETER; }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_tune"); }

}  // extern "C"
