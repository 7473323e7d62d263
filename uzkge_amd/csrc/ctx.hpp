// Process-wide backend context: one HIP device, one stream, grow-only HBM workspaces, per-kernel
// hipEvent profiling.  One process per GPU (the multi-GPU MSM runs one process per device and
// exchanges 96-byte partial sums through the host language's RCCL binding).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/uzkge_gpu.h"
#include "../../include/uzkge_gpu_test.h"
#include "ec.hpp"

namespace uzk {

void set_error(const char* fmt, ...);
int on_exception(const char* fn) noexcept;   // the catch (...) of every extern "C" entry point: message + UZK_ERR_DEVICE

#define UZK_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            (void)hipGetLastError(); /* the runtime keeps it as its sticky last error: clear it */ \
            ::uzk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return UZK_ERR_DEVICE;                                                         \
        }                                                                                  \
    } while (0)

#define UZK_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != UZK_OK) return _rc; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);   // grow-only; contents NOT preserved on growth
    void release();
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Small per-round arguments of the prover's lane kernels (challenges, blinds, lengths, pointer lists): the host writes them into
// pinned memory, ONE copy per round puts them into HBM, the kernels read them with scalar loads.  A round ends with a
// synchronisation (its commitments or evaluations are results), so the next round may overwrite the block.
struct ArgArena {
    char* h = nullptr;
    char* d = nullptr;
    size_t cap = 0, used = 0, uploaded = 0;
    int init(size_t bytes);
    void release();
    void reset() { used = 0; uploaded = 0; }
    // `count` objects of T: host pointer to fill, device pointer for the kernels (nullptr when the block is full: callers size it)
    template <class T> T* push(size_t count, const T** dev) {
        const size_t at = (used + 15) & ~(size_t)15, bytes = count * sizeof(T);
        if (at + bytes > cap) { *dev = nullptr; return nullptr; }
        used = at + bytes;
        *dev = reinterpret_cast<const T*>(d + at);
        return reinterpret_cast<T*>(h + at);
    }
    int upload(hipStream_t s);   // what was pushed since the last upload
};

struct ProfEntry {
    std::string name;
    hipEvent_t e0, e1;
};

struct NttPlan;   // ntt.hip
struct MsmWork;   // msm.hip

// One context = one stream, one set of grow-only workspaces, one lock.  The default context serves callers that
// never ask for another; uzk_ctx_create / uzk_ctx_set_current give every prover thread its own, so independent
// proofs overlap on the GPU (at n = 2^14 a single proof leaves most of the chip idle).  The device binding and the
// SRS registry are process-wide (Shared, api.cpp).
struct Ctx {
    std::mutex mu;
    bool ready = false;
    int want_device = -1;              // uzk_ctx_create_on: the device this context lives on; -1: the process's default device (uzk_init)
    int device = -1;                   // the device it was made ready on
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // second pipeline instance of large MSMs
    hipStream_t cur_stream = nullptr;  // stream the profiling events are recorded on
    int num_cus = 256;
    // profiling
    bool prof_on = false;
    std::vector<ProfEntry> prof_pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<double, uint64_t>> prof_totals;
    // NTT plans keyed by (n << 1 | inverse)
    std::map<uint64_t, NttPlan*> ntt_plans;
    DevBuf ntt_scratch[2];
    DevBuf ntt_io;            // staging for the host-pointer API
    // MSM
    MsmWork* msm = nullptr;
    DevBuf msm_scalars;       // staging for host scalars
    Fp* msm_tail_host = nullptr;   // pinned: the tail scalars of the current call (uzk_msm_g1_batch_tail_device)
    size_t msm_tail_cap = 0;
    int msm_window_bits = 0;  // 0 = auto
    // uzk_tune: switches with a production use, and three that let the tests reach every pipeline at small sizes
    int tune_no_precompute = 0;    // 1: ignore window tables (uzk_srs_precompute) -- the general pipeline over the plain bases
    int tune_ntt_two_pass = 1;     // 2^17 .. 2^21 as two passes of 9 .. 11 bits (0: the three passes of 5 .. 8 bits -- tests reach both)
    int tune_ntt_tile = 0;         // elements per workgroup of an NTT pass: 2048 (512 threads), 1024 (256 threads), 0 = by size
    int tune_small = 1;            // 1: n <= 2^15 takes the one-workgroup-per-slot pipeline (msm_small_*); 0: the general pipeline at every size
    int tune_chunk_log = 26;       // point-chunk size of one sort pass (tests lower it to reach the chunk loop at small n)
    int tune_stream_log = 0;       // log2 of the point chunk of a streamed host-scalar MSM (0 = 21); -1: never stream (upload, then one MSM)
    int tune_stream_min_log = 22;  // host-scalar MSMs of at least 2^this points are streamed (tests lower it)
    // bit 0: the quotient kernel, bit 1: evaluations / linear combinations, bit 2: the bucket-side group additions of the MSM on the lazy
    // 29-bit arithmetic (lz29.hpp) instead of 8 x 32-bit Montgomery words; same bytes either way.  Default: all on (UZK_ARITH29 in the
    // environment changes the default of every context of the process: the A/B runs of tests/cpp/prover_rounds); uzk_tune("arith29", mask).
    int tune_arith29 = default_arith29();
    static int default_arith29() {
        static const int v = [] { const char* e = std::getenv("UZK_ARITH29"); return e ? std::atoi(e) : 7; }();
        return v;
    }
    int tune_seg_sort = 1;         // 1: last pass of a packed two-pass sort = one workgroup per segment (msm_radix_segment_kernel); 0: the generic kernels; 10 + k: instantiation k whatever the size (tests)
    // poly.hip workspaces (grow-only)
    DevBuf poly_tmp, poly_tmp2, poly_io, zpoly_tmp, open_tmp, poly_args;
    DevBuf poly_cnt;                 // per-polynomial arrival counters of poly_eval_small (zero between calls); trimmed-length results
    uint32_t trim_flip = 0;          // which of the two trimmed-length result sets the next call uses
    void* poly_host = nullptr;       // pinned, device-visible: small results written by the kernels themselves
    size_t poly_host_cap = 0;
    // ntt.hip: decimated sub-vectors of the small 3 * 2^k path; three-level power tables of arbitrary
    // elements (coset shifts, mixed-radix roots) cached by VALUE, least recently used entry replaced
    DevBuf ntt_sub;
    struct PowCache { Fp key; bool valid = false; uint64_t stamp = 0; DevBuf buf; };
    PowCache pow_cache[8];
    uint64_t pow_stamp = 0;
    std::vector<void*> ntt_fused;   // NttFused* (ntt.hip): tables of the fused coset / radix-3 transforms
    // an entry of the process-wide SRS registry
    struct Srs {
        Affine* d_points = nullptr;
        size_t n = 0;
        bool owned = false;
        int device = 0;
        // optional window table (uzk_srs_precompute): table[j*n + i] = 2^(pre_c*j) * P_i
        Affine* d_table = nullptr;
        int pre_c = 0;
        uint32_t pre_W = 0;
    };

    hipEvent_t get_event();
    void prof_begin(const char* name);
    void prof_end();
    int prof_collect();   // sync + fold pending events into totals
};

// The scalars of an MSM call: vector b, element i is main[b * stride + i] for i < n_main and
// tail[b * tail_n + (i - n_main)] beyond (n = n_main + tail_n).  The tail carries the blind factors of a commit
// (apply_blind_factors, kzg_poly_commitment.rs:299-313) so that they ride in the commit's own MSM; it may live in
// pinned host memory (a few dozen elements, read once by the digit kernel).
struct ScalarView {
    const Fp* main = nullptr;
    uint64_t stride = 0;
    uint32_t n_main = 0;
    const Fp* tail = nullptr;
    uint32_t tail_n = 0;
    // Vectors in groups: vector b lives at main[(b / group) * group_stride + (b % group) * stride].  The prover's batched round 1
    // commits 8 of the 10 evaluation vectors of every proof of a lockstep batch (prover.cpp); one group = everything else.
    uint32_t group = 0xffffffffu;
    uint64_t group_stride = 0;
    static ScalarView dense(const Fp* p, size_t n) { ScalarView v; v.main = p; v.stride = n; v.n_main = (uint32_t)n; return v; }
    __host__ __device__ const Fp& at(uint32_t b, uint32_t i) const {
        return i < n_main ? main[(uint64_t)(b / group) * group_stride + (uint64_t)(b % group) * stride + i]
                          : tail[(uint64_t)b * tail_n + (i - n_main)];
    }
};

Ctx& ctx();
std::mutex& ctx_mutex();
int require_ready();
// While one lives, ctx() on this thread is `c`: a round that several callers share runs on a library-internal context
// (coalesce.cpp), whatever context the thread that happens to run it has made current.
struct CtxScope {
    Ctx* prev;
    explicit CtxScope(Ctx* c);
    ~CtxScope();
    CtxScope(const CtxScope&) = delete;
    CtxScope& operator=(const CtxScope&) = delete;
};
int ctx_init_internal(Ctx& c, int device);     // a context no handle names, ready on `device`, tuned like the default context
void ctx_release_internal(Ctx& c);
void devices_synchronize();                    // every device a context has been made ready on
void coalesce_release_all();                   // coalesce.cpp (uzk_shutdown)
void sharded_release_all();                    // sharded.cpp (uzk_shutdown)
// api.cpp: the process-wide SRS registry and the pinned-block table, for prover.cpp
uint64_t srs_insert(const Ctx::Srs& e);
bool srs_lookup(uint64_t handle, Ctx::Srs* out);
bool srs_erase(uint64_t handle, Ctx::Srs* out);
bool is_pinned_block(const void* p, size_t bytes);
int msm_dispatch_view(const Ctx::Srs& s, size_t offset, const ScalarView& sv, size_t n, uint32_t batch, Jac* out);
// prover.cpp: frees every circuit and prover (uzk_shutdown)
void prover_release_all();

// RAII kernel-launch bracket:  { KernelScope ks(c, "name"); kernel<<<...>>>(...); }
// host-side section timer feeding the same profile table (entries named host_*)
struct HostScope {
    Ctx& c;
    const char* name;
    std::chrono::steady_clock::time_point t0;
    HostScope(Ctx& c_, const char* n) : c(c_), name(n), t0(std::chrono::steady_clock::now()) {}
    ~HostScope() {
        if (!c.prof_on) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        auto& e = c.prof_totals[name];
        e.first += ms; e.second += 1;
    }
};

struct KernelScope {
    Ctx& c;
    KernelScope(Ctx& c_, const char* name) : c(c_) { if (c.prof_on) c.prof_begin(name); }
    ~KernelScope() { if (c.prof_on) c.prof_end(); }
};

// entry points implemented in the .hip files
int ntt_run(Ctx& c, const Fp* d_in, Fp* d_out, uint64_t n, bool inverse, const Fp* coset_shift_host, uint32_t batch,
            uint64_t in_stride = 0, uint64_t out_stride = 0, uint64_t in_len = 0);
void ntt_free_plans(Ctx& c);
void msm_plan_info(Ctx& c, size_t n, int* window_bits, int* windows);
int msm_run(Ctx& c, const Affine* points, const ScalarView& scalars, size_t n, uint32_t batch, Jac* out_host, int pre_c,
            uint32_t pre_stride, uint32_t pre_off);
int msm_run_streamed(Ctx& c, const Affine* points, const Fp* scalars_host, size_t n, Jac* out_host);
int msm_run_chunked(Ctx& c, const Affine* points, const Fp* d_scalars, size_t n, Jac* out_host);
int msm_build_table(Ctx& c, const Affine* d_points, size_t n, int cb, Affine** table_out, uint32_t* W_out);
int msm_precompute_window_bits(size_t n, int forced);
void msm_free(Ctx& c);
int synth_points_arith(Ctx& c, Affine* d_points, size_t n, const Fp& seed_scalar_mont);
int synth_points_random(Ctx& c, Affine* d_points, size_t n, uint64_t seed);
int synth_scalars(Ctx& c, Fp* d_scalars, size_t n, uint64_t seed);
int synth_scalars_mix(Ctx& c, Fp* d_scalars, size_t n, uint64_t seed);
int poly_eval_batch(Ctx& c, const Fp* d_coefs, uint64_t n, uint32_t batch, const Fp& x, Fp* out_host);
int poly_eval_batch_host(Ctx& c, const Fp* coefs_host, uint64_t n, uint32_t batch, const Fp& x, Fp* out_host);
int open_quotient_run(Ctx& c, const Fp* d_polys, uint64_t n, uint32_t batch, const Fp& z, const Fp& alpha, Fp* d_q,
                      Fp* evals_host);
int poly_eval_ptrs(Ctx& c, const void* const* d_polys, const uint64_t* lens, const uint32_t* point_idx, uint32_t count,
                   const Fp* points_host, uint32_t n_points, Fp* out_host);
int open_quotient_ptrs(Ctx& c, const void* const* d_polys, const uint64_t* lens, uint32_t count, const Fp& z, const Fp& alpha, Fp* d_q,
                       uint64_t q_cap, Fp* evals_host);
int poly_hide_batch_run(Ctx& c, Fp* d_coefs, uint64_t stride, uint64_t len_in, uint32_t count, const Fp* blinds_host, uint32_t hd,
                        uint64_t zeroing_degree);
int fold_blinds_batch_run(Ctx& c, const Fp* d_polys, uint64_t in_stride, const uint64_t* lens_host, uint64_t N, uint32_t batch, Fp* d_out,
                          uint64_t out_stride, Fp* d_tail, uint32_t tail_n, Fp* blinds_host);
int split_t_run(Ctx& c, const Fp* d_t, uint64_t t_len, uint64_t chunk, uint32_t n_chunks, const Fp* rands_host, Fp* d_chunks,
                uint64_t chunk_stride, uint64_t* lens_out);
int poly_trimmed_len_run(Ctx& c, const Fp* d_polys, uint64_t stride, const uint64_t* lens_host, uint32_t batch, uint64_t* out_host, bool sync);
int fold_blinds_run(Ctx& c, const Fp* d_coefs, uint64_t len, uint64_t N, Fp* d_out, Fp* blinds_host);
int poly_lincomb_run(Ctx& c, const void* const* d_polys, const uint64_t* lens, const Fp* scalars_host, uint32_t count, Fp* d_out,
                     uint64_t out_len);
int poly_hide_run(Ctx& c, Fp* d_coefs, uint64_t len, const Fp* blinds_host, uint32_t hiding_degree, uint64_t zeroing_degree);
int poly_scatter_run(Ctx& c, Fp* d_dst, uint64_t dst_stride, const uint32_t* idx_pinned, const Fp* val_pinned, uint32_t count, uint32_t batch);
struct QuotientDev;
int t_quotient_run(Ctx& c, const void* args_c_abi, Fp* d_out);
int z_poly_device(Ctx& c, const Fp* d_w, const uint32_t* d_perm, const Fp* d_group, const Fp* k_host, const Fp& beta,
                  const Fp& gamma, uint32_t n, uint32_t n_wires, Fp* d_z);
int z_poly_lanes(Ctx& c, ArgArena& args, const Fp* d_w, uint64_t w_lane_stride, const uint32_t* d_perm, const Fp* d_group, const Fp* k_host,
                 const Fp* d_bg, uint32_t n, uint32_t n_wires, uint32_t lanes, Fp* d_z, uint64_t z_lane_stride, uint8_t* lane_ok);
int t_quotient_lanes(Ctx& c, const void* args_c_abi, uint64_t own_stride, const void* d_lanes, uint32_t lanes, Fp* d_out, uint64_t out_stride);
void quotient_lane(const Fp& alpha, const Fp& beta, const Fp& gamma, const Fp* k, void* lane_out);
size_t quotient_lane_bytes();
// rounds.hip: the lane kernels of the prover's rounds (device-resident argument entries are filled through the *_fill helpers)
int hide_lanes(Ctx& c, Fp* d_coefs, uint64_t lane_stride, uint64_t slot_stride, uint32_t n, uint32_t slots, uint32_t lanes, const Fp* d_blinds);
int poly_eval_lanes(Ctx& c, const void* d_polys, uint32_t count, uint64_t max_len, const Fp* d_points, uint32_t lanes, uint32_t* d_counters, Fp* out_host_pinned);
void eval_poly_fill(void* host_entry, const void* p, uint64_t lane_stride, uint64_t len, uint32_t pt);
size_t eval_poly_bytes();
int poly_lincomb_lanes(Ctx& c, const void* d_polys, uint32_t count, const uint32_t* d_lens, const Fp* d_scalars, uint32_t lanes, Fp* d_out, uint64_t out_stride,
                       uint64_t out_len);
void lin_poly_fill(void* host_entry, const void* p, uint64_t lane_stride);
size_t lin_poly_bytes();
int open_div_lanes(Ctx& c, const Fp* d_h, uint64_t h_stride, uint64_t n, const void* d_pows, uint32_t count, Fp* d_q, uint64_t q_stride, uint64_t q_cap);
void div_pows_fill(void* host_entry, const Fp& z, int per);
size_t div_pows_bytes();
int open_div_per(uint64_t n);
int split_t_lanes(Ctx& c, const Fp* d_t, uint64_t t_stride, const uint32_t* d_t_lens, uint64_t chunk, uint32_t n_chunks, const Fp* d_rands, uint32_t lanes,
                  Fp* d_chunks, uint64_t chunk_stride);
int fold_blinds_lanes(Ctx& c, const Fp* d_polys, uint64_t in_stride, const uint32_t* d_lens, uint64_t N, uint32_t count, Fp* d_out, uint64_t out_stride, Fp* d_tail,
                      uint32_t tail_n);
int trimmed_len_lanes(Ctx& c, const Fp* d_polys, uint64_t stride, uint64_t cap, uint32_t count, uint64_t* d_sets, uint32_t count_max, uint32_t* flip,
                      uint64_t* out_host_pinned);
int z_poly_run(Ctx& c, const Fp* w_host, const uint32_t* perm_host, const Fp* group_host, const Fp* k_host,
               const Fp& beta, const Fp& gamma, uint32_t n, uint32_t n_wires, Fp* z_host);
void poly_free(Ctx& c);
int field_op_device(Ctx& c, int field, int op, const Fp* a, const Fp* b, Fp* out, size_t n);
int g1_op_device(Ctx& c, int op, const Affine* a, const Affine* b, Jac* out, size_t n);

}  // namespace uzk
