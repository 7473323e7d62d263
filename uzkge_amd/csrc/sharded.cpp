// One MSM over the GPUs of a node from ONE process (include/uzkge_gpu.h, "sharded SRS"): the bases are cut into contiguous point
// chunks (chunk i = [i n / N, (i + 1) n / N): uzkge_amd/sharded.py chunk_bounds, the split of north_star and SURVEY.md 8e), chunk
// i lives in the HBM of devices[i] under a context of its own, and a call runs the chunks' MSMs side by side -- one host thread
// per chunk, each uploading only its part of the scalars -- and folds the N 96-byte partial sums on the host (uzk_g1_fold's
// arithmetic).  Nothing moves between devices.  The process-per-GPU form of the same split, with the partial sums exchanged by
// RCCL, is uzkge_amd/sharded.py (bench.py --gpus N); this is the form a single Rust process links.
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <shared_mutex>
#include <thread>

#include "ctx.hpp"
#include "host_ec64.hpp"
#include "host_math.hpp"

namespace uzk {
namespace {

// One host thread per chunk, alive as long as the sharded SRS (round 6: a call used to create and join N - 1 std::threads -- 30 to
// 60 us each on this host, every call, and a cold thread's first HIP call on a device pays that runtime's per-thread set-up again).
// Calls on one handle from several threads queue per chunk: the chunk has ONE context (one stream, one set of workspaces) anyway.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    bool stop = false;
    void post(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!th.joinable()) th = std::thread([this] { loop(); });
            jobs.push_back(std::move(f));
        }
        cv.notify_one();
    }
    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                f = std::move(jobs.front());
                jobs.pop_front();
            }
            f();
        }
    }
    void join() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};
struct Chunk {
    int device = 0;
    size_t lo = 0, hi = 0;
    Ctx ctx;                       // internal: this chunk's stream and workspaces on its device
    uint64_t srs = 0;              // its bases in the process-wide registry (they live on `device`)
    Worker worker;                 // runs this chunk's part of every call (chunk 0 runs on the caller's thread)
};
// counts the chunks of one call that are still out
struct Latch {
    std::mutex mu;
    std::condition_variable cv;
    size_t left;
    explicit Latch(size_t n) : left(n) {}
    void done() { std::lock_guard<std::mutex> lk(mu); if (--left == 0) cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return left == 0; }); }
};
struct ShardedSrs {
    size_t n = 0;
    std::vector<std::unique_ptr<Chunk>> chunks;
    std::shared_mutex use;         // calls hold it shared; a release waits for the calls in flight (their threads still hold the object)
};
struct Registry {
    std::mutex mu;
    std::map<uint64_t, std::shared_ptr<ShardedSrs>> all;
    uint64_t next = 1;
};
Registry& reg() {
    static Registry* r = new Registry();
    return *r;
}
constexpr uint64_t kShardedBit = 1ull << 61;

void release(ShardedSrs& s) {
    std::unique_lock<std::shared_mutex> ul(s.use);
    for (auto& c : s.chunks) {
        c->worker.join();          // (no call is in flight: the lock above is exclusive)
        if (c->srs) {
            CtxScope scope(&c->ctx);
            (void)uzk_srs_release(c->srs);
            c->srs = 0;
        }
        ctx_release_internal(c->ctx);
    }
    s.chunks.clear();
}

}  // namespace

void sharded_release_all() {
    Registry& r = reg();
    std::map<uint64_t, std::shared_ptr<ShardedSrs>> all;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        all.swap(r.all);
    }
    for (auto& kv : all) release(*kv.second);
}

}  // namespace uzk

using namespace uzk;

extern "C" {

int uzk_srs_register_sharded(const uzk_g1_affine* points, size_t n, const int* devices, uint32_t n_devices, int window_bits, uint64_t* handle_out) try {
    if (!handle_out || !devices || (n > 0 && !points)) { set_error("uzk_srs_register_sharded: null pointer"); return UZK_ERR_PARAMETER; }
    if (n_devices == 0 || n_devices > 64) { set_error("uzk_srs_register_sharded: 1 .. 64 chunks"); return UZK_ERR_PARAMETER; }
    if (window_bits != -1 && window_bits != 0 && (window_bits < 4 || window_bits > 24)) { set_error("uzk_srs_register_sharded: window_bits must be -1 (no table), 0 (automatic) or 4 .. 24"); return UZK_ERR_PARAMETER; }
    const int visible = uzk_device_count();
    if (visible <= 0) { set_error("no HIP device visible: the MI355X backend has no CPU fallback"); return UZK_ERR_DEVICE; }
    for (uint32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= visible) { set_error("uzk_srs_register_sharded: devices[%u] = %d, %d device(s) visible", i, devices[i], visible); return UZK_ERR_PARAMETER; }
    auto s = std::make_shared<ShardedSrs>();
    s->n = n;
    for (uint32_t i = 0; i < n_devices; ++i) {
        auto c = std::make_unique<Chunk>();
        c->device = devices[i];
        c->lo = (size_t)((unsigned __int128)i * n / n_devices);
        c->hi = (size_t)((unsigned __int128)(i + 1) * n / n_devices);
        int rc = ctx_init_internal(c->ctx, c->device);
        // A chunk's scalars always come from the host: stream them under the chunk's own accumulation from 2^19 points on (a single call
        // starts at 2^22, where its first sub-chunk is a quarter of the vector; a chunk IS a fraction of one already).  Round 6, two and
        // eight virtual shards of a 2^22-point MSM on one GPU: 7.7-8.3 -> 7.0-7.1 ms and 8.2-9.5 -> 7.2 ms.
        c->ctx.tune_stream_min_log = std::min(c->ctx.tune_stream_min_log, 19);
        if (rc == UZK_OK) {
            CtxScope scope(&c->ctx);
            rc = uzk_srs_register(points + c->lo, c->hi - c->lo, &c->srs);
            if (rc == UZK_OK && window_bits >= 0 && c->hi > c->lo) rc = uzk_srs_precompute(c->srs, window_bits);
        }
        s->chunks.push_back(std::move(c));
        if (rc != UZK_OK) { release(*s); return rc; }
    }
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    const uint64_t h = kShardedBit | r.next++;
    r.all[h] = s;
    *handle_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_register_sharded"); }

int uzk_srs_release_sharded(uint64_t handle) try {
    std::shared_ptr<ShardedSrs> s;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.all.find(handle);
        if (it == r.all.end()) { set_error("uzk_srs_release_sharded: unknown handle %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        s = it->second;
        r.all.erase(it);
    }
    release(*s);
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_release_sharded"); }

int uzk_msm_g1_sharded(uint64_t handle, const uint64_t* scalars_mont, size_t n, uzk_g1_jac* partials_out, uzk_g1_jac* out) try {
    if (!out || (n > 0 && !scalars_mont)) { set_error("uzk_msm_g1_sharded: null pointer"); return UZK_ERR_PARAMETER; }
    std::shared_ptr<ShardedSrs> s;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.all.find(handle);
        if (it == r.all.end()) { set_error("uzk_msm_g1_sharded: unknown handle %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        s = it->second;
    }
    std::shared_lock<std::shared_mutex> in_use(s->use);
    if (s->chunks.empty()) { set_error("uzk_msm_g1_sharded: the handle was released"); return UZK_ERR_PARAMETER; }
    // KZG commit: degree + 1 > SRS length (kzg_poly_commitment.rs:283-285)
    if (n > s->n) { set_error("msm: n %zu exceeds SRS length %zu", n, s->n); return UZK_ERR_DEGREE; }
    const size_t N = s->chunks.size();
    std::vector<uzk_g1_jac> part(N);
    std::vector<int> rc(N, UZK_OK);
    std::vector<std::string> msg(N);
    auto work = [&](size_t i) {
        Chunk& c = *s->chunks[i];
        std::memset(&part[i], 0, sizeof part[i]);                 // z = 0: infinity, what an empty chunk contributes
        const size_t lo = std::min(c.lo, n), hi = std::min(c.hi, n);   // a short vector ends inside (or before) this chunk
        if (hi == lo) return;
        CtxScope scope(&c.ctx);                                   // this thread's calls go to the chunk's context and device
        rc[i] = uzk_msm_g1(c.srs, 0, scalars_mont + 4 * lo, hi - lo, &part[i]);
        if (rc[i] != UZK_OK) msg[i] = uzk_last_error();
    };
    Latch out_standing(N - 1);
    for (size_t i = 1; i < N; ++i) s->chunks[i]->worker.post([&, i] { work(i); out_standing.done(); });
    work(0);
    out_standing.wait();
    for (size_t i = 0; i < N; ++i)
        if (rc[i] != UZK_OK) { set_error("uzk_msm_g1_sharded: chunk %zu (device %d): %s", i, s->chunks[i]->device, msg[i].c_str()); return rc[i]; }
    if (partials_out) std::memcpy(partials_out, part.data(), N * sizeof(uzk_g1_jac));
    return uzk_g1_fold(part.data(), N, out);
} catch (...) { return uzk::on_exception("uzk_msm_g1_sharded"); }

int uzk_srs_sharded_info(uint64_t handle, size_t* n_out, uint32_t* n_chunks_out, int* devices_out, size_t* bounds_out) try {
    std::shared_ptr<ShardedSrs> s;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.all.find(handle);
        if (it == r.all.end()) { set_error("uzk_srs_sharded_info: unknown handle %llu", (unsigned long long)handle); return UZK_ERR_PARAMETER; }
        s = it->second;
    }
    std::shared_lock<std::shared_mutex> in_use(s->use);
    if (n_out) *n_out = s->n;
    if (n_chunks_out) *n_chunks_out = (uint32_t)s->chunks.size();
    for (size_t i = 0; i < s->chunks.size(); ++i) {
        if (devices_out) devices_out[i] = s->chunks[i]->device;
        if (bounds_out) { bounds_out[2 * i] = s->chunks[i]->lo; bounds_out[2 * i + 1] = s->chunks[i]->hi; }
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_srs_sharded_info"); }

}  // extern "C"
