// Circuits and the five prover rounds behind the C ABI (include/uzkge_gpu.h, "circuits and the five prover rounds").
//
// prover_with_lagrange (uzkge/src/plonk/prover.rs:88-394) keeps per-circuit data in PlonkProverParams (indexer.rs:76-138) and one
// proof's polynomials in Vec<Fr>s between its Fiat-Shamir rounds.  Here the first is a Circuit (HBM-resident tables with
// copy-on-write replacement, because refresh_prover_params_public_key -- shuffle/src/gen_params/params.rs:57-129 -- swaps twelve
// of them once per game), the second a Prover (the buffers of `batch` proofs advancing in lockstep), and uzk_prove_round1..5 do
// what the reference does between two transcript draws.  Everything here is host-side sequencing of the kernels in ntt.hip,
// msm.hip and poly.hip on the calling context's stream; the transcript, the prng and r_poly's O(1) scalars stay with the caller.
#include <algorithm>
#include <cstring>
#include <memory>

#include "ctx.hpp"
#include "host_math.hpp"

namespace uzk {
namespace {

constexpr uint32_t kSlots = UZK_CIRCUIT_SLOTS;
constexpr uint32_t kWires = 5, kWsel = 3, kProofSlots = 10;       // slots of a proof's own polynomials: w0..4, w_sel0..2, pi, z
constexpr uint32_t kTail = 6;                                      // blinds || -blinds, three slots each (apply_blind_factors)
constexpr uint32_t kMaxBatch = 64;
constexpr int kBatchWindowBits = 15;                             // window width of the lockstep-batch commit table (tools/rounds_window_sweep.sh)

struct DevBlock {                                                  // device memory that any thread may drop the last reference to
    void* p = nullptr;
    size_t bytes = 0;
    int device = 0;
    ~DevBlock() {
        if (!p) return;
        (void)hipSetDevice(device);
        (void)hipFree(p);                                          // waits for the device: nothing still reads the block
    }
};
static int dev_block(size_t bytes, std::shared_ptr<DevBlock>* out) {
    auto b = std::make_shared<DevBlock>();
    b->device = bound_device();
    hipError_t e = hipMalloc(&b->p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        b->p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    b->bytes = bytes;
    *out = b;
    return UZK_OK;
}

// One slot of a circuit: coefficient form (n elements allocated, `len` meaningful) and coset evaluations (6n), both inside a
// block shared with the other slots that were installed by the same call.
struct SlotRef {
    std::shared_ptr<DevBlock> blk;
    const Fp* poly = nullptr;
    const Fp* coset = nullptr;
    uint64_t len = 0;
};
struct TableSet { SlotRef s[kSlots]; };

struct Circuit {
    uint32_t n = 0, m = 0, shuffle = 0, n_slots = 0;
    uint64_t srs = 0;                                              // registry handle: lagrange[0..n) || pcs[0..3) || pcs[n..n+3)
    uint64_t srs_batch = 0;                                        // the same bases under a second handle with the wide-window table (0: none)
    Fp k[kWires], anemoi_g, anemoi_g_inv, edwards_a, group_gen, k1_inv, z_h_inv[6];
    std::shared_ptr<DevBlock> fixed;                               // permutation (5n u32) | group (n Fp)
    const uint32_t* d_perm = nullptr;
    const Fp* d_group = nullptr;
    std::mutex mu;                                                 // guards `tables`
    std::shared_ptr<const TableSet> tables;
};

struct Prover {
    uint32_t n = 0, m = 0, cs = 0, B = 0;
    std::mutex mu;
    std::shared_ptr<DevBlock> blk;
    Fp *d_evals = nullptr, *d_coefs = nullptr, *d_coset = nullptr, *d_tq = nullptr, *d_t = nullptr, *d_chunks = nullptr, *d_fold = nullptr,
       *d_tail = nullptr, *d_q = nullptr, *d_r = nullptr;
    // pinned: measured trimmed lengths (t: B, quotients: 2B) and the public-input list of round 1
    uint64_t* h_lens = nullptr;
    void* h_pi = nullptr;
    size_t h_pi_cap = 0;
    // the proof in flight
    int round = 0;                                                 // rounds completed
    std::shared_ptr<Circuit> circuit;
    std::shared_ptr<const TableSet> snap;
    uint32_t n_first = 0, np = 0;                                  // committed in round 1 (5 or 8); slots in use (7 or 10)
    uint32_t hiding[kWires + kWsel] = {};
    std::vector<Fp> beta, gamma, zeta, zeta_omega;
    std::vector<uint64_t> chunk_lens;                              // B x 5
    ~Prover() {
        if (h_lens) (void)hipHostFree(h_lens);
        if (h_pi) (void)hipHostFree(h_pi);
    }
    uint32_t sl_pi() const { return n_first; }
    uint32_t sl_z() const { return n_first + 1; }
    Fp* evals(uint32_t b, uint32_t slot) const { return d_evals + ((uint64_t)b * kProofSlots + slot) * n; }
    Fp* coefs(uint32_t b, uint32_t slot) const { return d_coefs + ((uint64_t)b * kProofSlots + slot) * m; }
    Fp* coset(uint32_t b, uint32_t slot) const { return d_coset + ((uint64_t)b * kProofSlots + slot) * m; }
};

struct Registry {
    std::mutex mu;
    std::map<uint64_t, std::shared_ptr<Circuit>> circuits;
    std::map<uint64_t, std::shared_ptr<Prover>> provers;
    uint64_t next = 1;
};
static Registry& reg() {
    static Registry r;
    return r;
}
static std::shared_ptr<Circuit> find_circuit(uint64_t h) {
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.circuits.find(h);
    return it == r.circuits.end() ? nullptr : it->second;
}
static std::shared_ptr<Prover> find_prover(uint64_t h) {
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.provers.find(h);
    return it == r.provers.end() ? nullptr : it->second;
}

static const Fp* as_fp(const uint64_t* p) { return reinterpret_cast<const Fp*>(p); }
static Fp fp_of(const uint64_t* w) { Fp f; std::memcpy(&f, w, sizeof f); return f; }
static bool fp_eq(const Fp& a, const Fp& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// largest power of two <= degree, as the reference's loops compute it (pcs.rs:139-145, helpers.rs:1367-1373); 0 for degree 0
static uint64_t max_power_of_2(uint64_t degree) {
    uint64_t p = 1;
    if (degree == 0) return 0;
    while (p * 2 <= degree) p *= 2;
    return p;
}

// coefficient forms [count][n] (zero padded) -> coset evaluations [count][6n]: zero-padded copy and ONE batched coset FFT
// (indexer.rs:316-470: `coset_fft_with_domain(&domain_m, &k[1])` per table)
static int derive_cosets(Ctx& c, uint32_t n, const Fp& k1, const Fp* polys, Fp* cosets, uint32_t count) {
    const uint64_t m = 6ull * n;
    UZK_HIP(hipMemsetAsync(cosets, 0, (size_t)count * m * sizeof(Fp), c.stream));
    UZK_HIP(hipMemcpy2DAsync(cosets, (size_t)m * sizeof(Fp), polys, (size_t)n * sizeof(Fp), (size_t)n * sizeof(Fp), count, hipMemcpyDeviceToDevice, c.stream));
    return ntt_run(c, cosets, cosets, m, false, &k1, count);
}

// Installs `count` polynomials whose coefficient forms already sit in blk (polys area: [count][n], zero padded) as slots
// first .. first + count of a new table set.
static int derive_and_install(Ctx& c, Circuit& cir, std::shared_ptr<DevBlock> blk, uint32_t first, uint32_t count, const uint64_t* lens) {
    const uint32_t n = cir.n, m = cir.m;
    Fp* polys = static_cast<Fp*>(blk->p);
    Fp* cosets = polys + (uint64_t)count * n;
    UZK_TRY(derive_cosets(c, n, cir.k[1], polys, cosets, count));
    UZK_HIP(hipStreamSynchronize(c.stream));                       // other contexts' streams read these tables
    auto next = std::make_shared<TableSet>();
    std::lock_guard<std::mutex> lk(cir.mu);
    if (cir.tables) *next = *cir.tables;
    for (uint32_t i = 0; i < count; ++i) {
        SlotRef& s = next->s[first + i];
        s.blk = blk;
        s.poly = polys + (uint64_t)i * n;
        s.coset = cosets + (uint64_t)i * m;
        s.len = lens[i];
    }
    cir.tables = next;
    return UZK_OK;
}

// `count` evaluation vectors (host) -> coefficient forms in d_polys ([count][n]: batched iFFT), their trimmed lengths
// (FpPolynomial::from_coefs after ifft_with_domain, field_polynomial.rs:594-597) and, optionally, the Lagrange commitments of the
// evaluations (the commit closure's Lagrange branch, indexer.rs:284-299) over `srs`.
static int polys_from_evals(Ctx& c, const Ctx::Srs* srs, uint32_t n, uint32_t count, const uint64_t* evals, Fp* d_polys, std::vector<uint64_t>& lens, Jac* cms) {
    UZK_TRY(c.poly_io.reserve((size_t)count * n * sizeof(Fp)));
    Fp* d_evals = c.poly_io.as<Fp>();
    UZK_HIP(hipMemcpyAsync(d_evals, evals, (size_t)count * n * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    UZK_TRY(ntt_run(c, d_evals, d_polys, n, true, nullptr, count));
    lens.assign(count, 0);
    std::vector<uint64_t> cap(16, n);
    for (uint32_t i0 = 0; i0 < count; i0 += 16)
        UZK_TRY(poly_trimmed_len_run(c, d_polys + (uint64_t)i0 * n, n, cap.data(), std::min<uint32_t>(16, count - i0), lens.data() + i0, true));
    if (cms) UZK_TRY(msm_dispatch_view(*srs, 0, ScalarView::dense(d_evals, n), n, count, cms));
    return UZK_OK;
}

static bool slot_range_ok(const Circuit& cir, uint32_t first, uint32_t count) {
    return count > 0 && first < cir.n_slots && count <= cir.n_slots - first;
}

// uploads `count` coefficient forms (host) into a fresh block and installs them
static int upload_and_install(Ctx& c, Circuit& cir, uint32_t first, uint32_t count, const uint64_t* const* polys, const uint64_t* lens) {
    const uint32_t n = cir.n;
    std::shared_ptr<DevBlock> blk;
    UZK_TRY(dev_block((size_t)count * ((size_t)n + cir.m) * sizeof(Fp), &blk));
    Fp* d = static_cast<Fp*>(blk->p);
    UZK_HIP(hipMemsetAsync(d, 0, (size_t)count * n * sizeof(Fp), c.stream));
    std::vector<uint64_t> l(count);
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t slot = first + i;
        if (slot == UZK_CS_COSET_QUOTIENT && !polys[i]) {
            // coset_quotient[i] = k[1] * g_m^i (indexer.rs:278-282) = the coset evaluations of the polynomial X
            const Fp one = Fr::one();
            UZK_HIP(hipMemcpyAsync(d + (uint64_t)i * n + 1, &one, sizeof(Fp), hipMemcpyHostToDevice, c.stream));
            UZK_HIP(hipStreamSynchronize(c.stream));               // `one` is a stack variable
            l[i] = 2;
            continue;
        }
        l[i] = lens[i];
        if (l[i] > n) { set_error("circuit: slot %u has %llu coefficients, n = %u", slot, (unsigned long long)l[i], n); return UZK_ERR_PARAMETER; }
        if (l[i] && !polys[i]) { set_error("circuit: slot %u is null", slot); return UZK_ERR_PARAMETER; }
        if (l[i]) UZK_HIP(hipMemcpyAsync(d + (uint64_t)i * n, polys[i], (size_t)l[i] * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    }
    return derive_and_install(c, cir, blk, first, count, l.data());
}

static void release_circuit_srs(Circuit& cir) {
    for (uint64_t* h : {&cir.srs_batch, &cir.srs}) {               // the second handle adopted the first one's points: it goes first
        Ctx::Srs e;
        if (*h && srs_erase(*h, &e)) {
            if (e.owned && e.d_points) (void)hipFree(e.d_points);
            if (e.d_table) (void)hipFree(e.d_table);
        }
        *h = 0;
    }
}

// commit = lagrange_pcs.commit(evals) + apply_blind_factors (prover.rs:132-142) as ONE batched MSM with tail scalars.
// Two window tables serve two regimes (measured): a single proof's commits are chains of dependent additions on
// a mostly idle chip -- narrow windows, one workgroup per (vector, window), shortest chain; the 8 x B / 5 x B / 2 x B vectors of
// a lockstep batch fill the chip, and there the NUMBER of additions counts: 15-bit windows over one shared bucket set per
// vector take 17 additions per scalar instead of 32.
static int commit(Circuit& cir, const ScalarView& sv, uint32_t batch, uint32_t lockstep, Jac* out) {
    Ctx::Srs srs;
    if (!srs_lookup(lockstep >= 2 && cir.srs_batch ? cir.srs_batch : cir.srs, &srs)) { set_error("prover: the circuit's commit bases are gone"); return UZK_ERR_PARAMETER; }
    return msm_dispatch_view(srs, 0, sv, (size_t)sv.n_main + sv.tail_n, batch, out);
}

// pinned, device-visible copy of a round's tail scalars; the commit that reads it synchronises before the round returns
static int stage_tail(Ctx& c, const std::vector<Fp>& tail, const Fp** out) {
    const size_t bytes = tail.size() * sizeof(Fp);
    if (c.msm_tail_cap < bytes) {
        if (c.msm_tail_host) { UZK_HIP(hipStreamSynchronize(c.stream)); (void)hipHostFree(c.msm_tail_host); c.msm_tail_host = nullptr; c.msm_tail_cap = 0; }
        const size_t cap = std::max<size_t>(bytes, 1 << 14);
        UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.msm_tail_host), cap, hipHostMallocDefault));
        c.msm_tail_cap = cap;
    }
    std::memcpy(c.msm_tail_host, tail.data(), bytes);
    *out = c.msm_tail_host;
    return UZK_OK;
}
static void put_tail(std::vector<Fp>& tail, size_t vec, const Fp* blinds, uint32_t hd) {
    for (uint32_t j = 0; j < 3; ++j) {
        const Fp b = j < hd ? blinds[j] : Fr::zero();
        tail[vec * kTail + j] = b;
        tail[vec * kTail + 3 + j] = Fr::neg(b);
    }
}

static int need_round(const Prover& p, int done, const char* who) {
    if (p.round != done || !p.circuit) { set_error("%s: the prover has completed %d round(s) of its proof, this call needs %d", who, p.round, done); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}
// a failed round ends the proof: tables are released, the next call must be round 1
static int fail(Prover& p, int rc) {
    p.round = 0;
    p.snap.reset();
    p.circuit.reset();
    return rc;
}
#define ROUND_TRY(expr)                       \
    do {                                      \
        int _rc = (expr);                     \
        if (_rc != UZK_OK) return fail(p, _rc); \
    } while (0)
#define ROUND_HIP(expr)                                                                     \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (void)hipGetLastError();                                                        \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return fail(p, UZK_ERR_DEVICE);                                                 \
        }                                                                                   \
    } while (0)

}  // namespace

void prover_release_all() {
    Registry& r = reg();
    std::map<uint64_t, std::shared_ptr<Circuit>> cs;
    std::map<uint64_t, std::shared_ptr<Prover>> ps;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        cs.swap(r.circuits);
        ps.swap(r.provers);
    }
    const int dev = bound_device();
    if (dev >= 0) { (void)hipSetDevice(dev); (void)hipDeviceSynchronize(); }
    ps.clear();
    for (auto& kv : cs) release_circuit_srs(*kv.second);
    cs.clear();
}

}  // namespace uzk

using namespace uzk;
#define API_LOCK std::lock_guard<std::mutex> _lk(ctx_mutex())

extern "C" {

int uzk_circuit_create(const uzk_circuit_desc* desc, uint64_t* circuit_out) try {
    if (!desc || !circuit_out) { set_error("uzk_circuit_create: null pointer"); return UZK_ERR_PARAMETER; }
    const uint32_t n = desc->n;
    if (n < 16 || n > (1u << 20) || (n & (n - 1))) { set_error("uzk_circuit_create: n must be a power of two in 16 .. 2^20 (n = %u)", n); return UZK_ERR_PARAMETER; }
    if (!desc->lagrange_bases || !desc->blind_bases || !desc->permutation) { set_error("uzk_circuit_create: null pointer"); return UZK_ERR_PARAMETER; }
    const Fp omega = fr_root_of_unity(n);
    if (!fp_eq(omega, fp_of(desc->group_gen))) {
        set_error("uzk_circuit_create: the caller's group_gen of the size-%u domain is not the library's (uzk_domain_group_gen)", n);
        return UZK_ERR_FFT;
    }
    for (uint64_t i = 0; i < (uint64_t)kWires * n; ++i)
        if (desc->permutation[i] >= kWires * n) { set_error("uzk_circuit_create: permutation[%llu] out of range", (unsigned long long)i); return UZK_ERR_PARAMETER; }
    auto cir = std::make_shared<Circuit>();
    cir->n = n; cir->m = 6 * n; cir->shuffle = desc->shuffle ? 1 : 0;
    cir->n_slots = cir->shuffle ? kSlots : (uint32_t)UZK_CS_QPK;
    for (uint32_t j = 0; j < kWires; ++j) cir->k[j] = fp_of(desc->k[j]);
    cir->anemoi_g = fp_of(desc->anemoi_g); cir->anemoi_g_inv = fp_of(desc->anemoi_g_inv); cir->edwards_a = fp_of(desc->edwards_a);
    cir->group_gen = omega;
    if (Fr::is_zero(cir->k[1])) { set_error("uzk_circuit_create: k[1] is zero"); return UZK_ERR_PARAMETER; }
    cir->k1_inv = fr_inv(cir->k[1]);
    {   // 1 / Z_H on the coset: 1 / (k1^n g_m^(n i) - 1), i < 6 (helpers.rs:242-252)
        const Fp gm_n = f_pow_u64<Fr>(fr_root_of_unity(cir->m), n);
        Fp mult = f_pow_u64<Fr>(cir->k[1], n);
        for (int i = 0; i < 6; ++i) {
            const Fp d = Fr::sub(mult, Fr::one());
            if (Fr::is_zero(d)) { set_error("uzk_circuit_create: k[1] lies in the evaluation domain"); return UZK_ERR_PARAMETER; }
            cir->z_h_inv[i] = fr_inv(d);
            mult = Fr::mul(mult, gm_n);
        }
    }
    // commit bases: the Lagrange SRS followed by the six monomial powers apply_blind_factors touches; registered (and the window
    // table built) through the public entry points, which take the context lock themselves
    {
        std::vector<uzk_g1_affine> bases((size_t)n + kTail);
        std::memcpy(bases.data(), desc->lagrange_bases, (size_t)n * sizeof(uzk_g1_affine));
        std::memcpy(bases.data() + n, desc->blind_bases, kTail * sizeof(uzk_g1_affine));
        UZK_TRY(uzk_srs_register(bases.data(), bases.size(), &cir->srs));
    }
    if (desc->precompute) {
        if (desc->precompute != 1 && (desc->precompute < 4 || desc->precompute > 24)) { (void)uzk_srs_release(cir->srs); set_error("uzk_circuit_create: precompute must be 0, 1 (automatic) or a window width 4 .. 24"); return UZK_ERR_PARAMETER; }
        int rc = uzk_srs_precompute(cir->srs, desc->precompute == 1 ? 0 : (int)desc->precompute);
        if (rc == UZK_OK && desc->precompute == 1) {
            // automatic: a second table for lockstep batches over the same resident bases (adopted, not copied)
            Ctx::Srs first;
            if (!srs_lookup(cir->srs, &first)) rc = UZK_ERR_PARAMETER;
            if (rc == UZK_OK) rc = uzk_srs_register_device(first.d_points, first.n, &cir->srs_batch);
            // window width of the lockstep table: 15 bits at n = 2^14 (swept), one bit per halving below, never under the 8 bits
            // of the one-workgroup-per-window pipeline
            int lg = 0;
            while ((2u << lg) <= n) ++lg;
            if (rc == UZK_OK) rc = uzk_srs_precompute(cir->srs_batch, std::max(8, std::min(kBatchWindowBits, lg + 1)));
        }
        if (rc != UZK_OK) { release_circuit_srs(*cir); return rc; }
    }
    int rc;
    {
        API_LOCK;
        rc = require_ready();
        Ctx& c = ctx();
        auto body = [&]() -> int {
            UZK_TRY(dev_block((size_t)kWires * n * sizeof(uint32_t) + (size_t)n * sizeof(Fp), &cir->fixed));
            Fp* d_group = static_cast<Fp*>(cir->fixed->p);
            uint32_t* d_perm = reinterpret_cast<uint32_t*>(d_group + n);
            cir->d_group = d_group; cir->d_perm = d_perm;
            UZK_HIP(hipMemcpyAsync(d_perm, desc->permutation, (size_t)kWires * n * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
            // group[i] = omega^i (prover_params.group): the forward NTT of X
            const Fp one = Fr::one();
            UZK_HIP(hipMemsetAsync(d_group, 0, (size_t)n * sizeof(Fp), c.stream));
            UZK_HIP(hipMemcpyAsync(d_group + 1, &one, sizeof(Fp), hipMemcpyHostToDevice, c.stream));
            UZK_HIP(hipStreamSynchronize(c.stream));
            UZK_TRY(ntt_run(c, d_group, d_group, n, false, nullptr, 1));
            return upload_and_install(c, *cir, 0, cir->n_slots, desc->polys, desc->poly_lens);
        };
        if (rc == UZK_OK) rc = body();
    }
    if (rc != UZK_OK) { release_circuit_srs(*cir); return rc; }
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    const uint64_t h = r.next++;
    r.circuits[h] = cir;
    *circuit_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_create"); }

int uzk_circuit_update_tables(uint64_t circuit, uint32_t first_slot, uint32_t count, const uint64_t* const* polys, const uint64_t* lens) try {
    API_LOCK;
    auto cir = find_circuit(circuit);
    if (!cir) { set_error("uzk_circuit_update_tables: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    if (!polys || !lens) { set_error("uzk_circuit_update_tables: null pointer"); return UZK_ERR_PARAMETER; }
    if (!slot_range_ok(*cir, first_slot, count)) { set_error("uzk_circuit_update_tables: slots %u .. %u of %u", first_slot, first_slot + count, cir->n_slots); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    return upload_and_install(ctx(), *cir, first_slot, count, polys, lens);
} catch (...) { return uzk::on_exception("uzk_circuit_update_tables"); }

int uzk_circuit_refresh_tables(uint64_t circuit, uint32_t first_slot, uint32_t count, const uint64_t* evals, uint64_t* polys_out,
                               uint64_t* lens_out, uint64_t* coset_out, uzk_g1_jac* commitments_out) try {
    API_LOCK;
    auto cir = find_circuit(circuit);
    if (!cir) { set_error("uzk_circuit_refresh_tables: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    if (!evals) { set_error("uzk_circuit_refresh_tables: null pointer"); return UZK_ERR_PARAMETER; }
    if (!slot_range_ok(*cir, first_slot, count)) { set_error("uzk_circuit_refresh_tables: slots %u .. %u of %u", first_slot, first_slot + count, cir->n_slots); return UZK_ERR_PARAMETER; }
    if (first_slot <= UZK_CS_COSET_QUOTIENT && UZK_CS_COSET_QUOTIENT < first_slot + count) { set_error("uzk_circuit_refresh_tables: slot 20 (coset_quotient) has no evaluation form"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    const uint32_t n = cir->n, m = cir->m;
    std::shared_ptr<DevBlock> blk;
    UZK_TRY(dev_block((size_t)count * ((size_t)n + m) * sizeof(Fp), &blk));
    Fp* d_polys = static_cast<Fp*>(blk->p);
    Ctx::Srs srs;
    if (!srs_lookup(count >= 2 && cir->srs_batch ? cir->srs_batch : cir->srs, &srs)) { set_error("uzk_circuit_refresh_tables: the circuit's commit bases are gone"); return UZK_ERR_PARAMETER; }
    std::vector<uint64_t> lens;
    std::vector<Jac> cm(commitments_out ? count : 0);
    UZK_TRY(polys_from_evals(c, &srs, n, count, evals, d_polys, lens, commitments_out ? cm.data() : nullptr));
    if (commitments_out) std::memcpy(commitments_out, cm.data(), (size_t)count * sizeof(Jac));
    UZK_TRY(derive_and_install(c, *cir, blk, first_slot, count, lens.data()));
    if (polys_out) UZK_HIP(hipMemcpyAsync(polys_out, d_polys, (size_t)count * n * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    if (coset_out) UZK_HIP(hipMemcpyAsync(coset_out, d_polys + (uint64_t)count * n, (size_t)count * m * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    if (polys_out || coset_out) UZK_HIP(hipStreamSynchronize(c.stream));
    if (lens_out) std::memcpy(lens_out, lens.data(), (size_t)count * sizeof(uint64_t));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_refresh_tables"); }

int uzk_preprocess_tables(uint64_t lagrange_srs, uint32_t n, uint32_t count, const uint64_t* evals, const uint64_t* k1_mont, uint64_t* polys_out,
                          uint64_t* lens_out, uint64_t* coset_out, uzk_g1_jac* commitments_out) try {
    API_LOCK;
    if (!evals || (coset_out && !k1_mont)) { set_error("uzk_preprocess_tables: null pointer"); return UZK_ERR_PARAMETER; }
    if (n < 2 || n > (1u << 24) || (n & (n - 1)) || count == 0 || count > 4096) { set_error("uzk_preprocess_tables: n must be a power of two in 2 .. 2^24, 1 <= count <= 4096"); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Ctx::Srs srs;
    if (commitments_out) {
        if (!srs_lookup(lagrange_srs, &srs)) { set_error("uzk_preprocess_tables: unknown SRS handle %llu", (unsigned long long)lagrange_srs); return UZK_ERR_PARAMETER; }
        if (srs.n < n) { set_error("uzk_preprocess_tables: the SRS holds %zu bases, n = %u", srs.n, n); return UZK_ERR_DEGREE; }
    }
    const uint64_t m = 6ull * n;
    std::shared_ptr<DevBlock> blk;
    UZK_TRY(dev_block((size_t)count * ((size_t)n + (coset_out ? m : 0)) * sizeof(Fp), &blk));
    Fp* d_polys = static_cast<Fp*>(blk->p);
    std::vector<uint64_t> lens;
    std::vector<Jac> cm(commitments_out ? count : 0);
    UZK_TRY(polys_from_evals(c, &srs, n, count, evals, d_polys, lens, commitments_out ? cm.data() : nullptr));
    if (commitments_out) std::memcpy(commitments_out, cm.data(), (size_t)count * sizeof(Jac));
    if (coset_out) {
        Fp* d_cosets = d_polys + (uint64_t)count * n;
        UZK_TRY(derive_cosets(c, n, fp_of(k1_mont), d_polys, d_cosets, count));
        UZK_HIP(hipMemcpyAsync(coset_out, d_cosets, (size_t)count * m * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    }
    if (polys_out) UZK_HIP(hipMemcpyAsync(polys_out, d_polys, (size_t)count * n * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    if (lens_out) std::memcpy(lens_out, lens.data(), (size_t)count * sizeof(uint64_t));
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_preprocess_tables"); }

int uzk_circuit_table(uint64_t circuit, uint32_t slot, int which, const void** d_out, uint64_t* len_out) try {
    auto cir = find_circuit(circuit);
    if (!cir || !d_out || slot >= cir->n_slots || (which != 0 && which != 1)) { set_error("uzk_circuit_table: bad arguments"); return UZK_ERR_PARAMETER; }
    std::lock_guard<std::mutex> lk(cir->mu);
    const SlotRef& s = cir->tables->s[slot];
    *d_out = which ? static_cast<const void*>(s.coset) : static_cast<const void*>(s.poly);
    if (len_out) *len_out = which ? cir->m : s.len;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_table"); }

int uzk_circuit_release(uint64_t circuit) try {
    API_LOCK;
    std::shared_ptr<Circuit> cir;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.circuits.find(circuit);
        if (it == r.circuits.end()) { set_error("uzk_circuit_release: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
        cir = it->second;
        r.circuits.erase(it);
    }
    Ctx& c = ctx();
    if (c.ready) { (void)hipSetDevice(c.device); (void)hipStreamSynchronize(c.stream); }
    release_circuit_srs(*cir);
    return UZK_OK;                                                 // tables and the fixed block go with the last reference
} catch (...) { return uzk::on_exception("uzk_circuit_release"); }

int uzk_prover_create(uint32_t n, uint32_t batch, uint64_t* prover_out) try {
    API_LOCK;
    if (!prover_out) { set_error("uzk_prover_create: null pointer"); return UZK_ERR_PARAMETER; }
    if (n < 16 || n > (1u << 20) || (n & (n - 1)) || batch == 0 || batch > kMaxBatch) {
        set_error("uzk_prover_create: n must be a power of two in 16 .. 2^20 and 1 <= batch <= %u", kMaxBatch);
        return UZK_ERR_PARAMETER;
    }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    auto p = std::make_shared<Prover>();
    p->n = n; p->m = 6 * n; p->cs = n + 8; p->B = batch;
    const uint64_t m = p->m, cs = p->cs, B = batch;
    const uint64_t elems = B * ((uint64_t)kProofSlots * n + 2ull * kProofSlots * m + 2 * m + 5 * cs + 5ull * n + 5 * kTail + 2 * cs + cs);
    UZK_TRY(dev_block(elems * sizeof(Fp), &p->blk));
    Fp* q = static_cast<Fp*>(p->blk->p);
    p->d_evals = q; q += B * kProofSlots * n;
    p->d_coefs = q; q += B * kProofSlots * m;
    p->d_coset = q; q += B * kProofSlots * m;
    p->d_tq = q; q += B * m;
    p->d_t = q; q += B * m;
    p->d_chunks = q; q += B * 5 * cs;
    p->d_fold = q; q += B * 5 * n;
    p->d_tail = q; q += B * 5 * kTail;
    p->d_q = q; q += B * 2 * cs;
    p->d_r = q;
    // coefficient slots are zero beyond what a round writes (the coset FFTs read all 6n elements); the evaluation slots must
    // hold field elements from the start (a lockstep batch transforms slot z before round 2 has filled it)
    UZK_HIP(hipMemsetAsync(p->blk->p, 0, elems * sizeof(Fp), c.stream));
    UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_lens), 3 * B * sizeof(uint64_t), hipHostMallocDefault));
    UZK_HIP(hipStreamSynchronize(c.stream));
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    const uint64_t h = r.next++;
    r.provers[h] = p;
    *prover_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prover_create"); }

int uzk_prover_destroy(uint64_t prover) try {
    API_LOCK;
    std::shared_ptr<Prover> p;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.provers.find(prover);
        if (it == r.provers.end()) { set_error("uzk_prover_destroy: unknown prover %llu", (unsigned long long)prover); return UZK_ERR_PARAMETER; }
        p = it->second;
        r.provers.erase(it);
    }
    Ctx& c = ctx();
    if (c.ready) { (void)hipSetDevice(c.device); (void)hipStreamSynchronize(c.stream); }
    std::lock_guard<std::mutex> lk(p->mu);                         // a round in flight on another thread finishes first
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prover_destroy"); }

int uzk_prover_buffer(uint64_t prover, int which, void** d_out, uint64_t* elems_out) try {
    auto pp = find_prover(prover);
    if (!pp || !d_out || !elems_out) { set_error("uzk_prover_buffer: bad arguments"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    const uint64_t n = p.n, m = p.m, cs = p.cs;
    switch (which) {
        case 0: *d_out = p.d_evals; *elems_out = kProofSlots * n; break;
        case 1: *d_out = p.d_coefs; *elems_out = kProofSlots * m; break;
        case 2: *d_out = p.d_coset; *elems_out = kProofSlots * m; break;
        case 3: *d_out = p.d_tq; *elems_out = m; break;
        case 4: *d_out = p.d_t; *elems_out = m; break;
        case 5: *d_out = p.d_chunks; *elems_out = 5 * cs; break;
        case 6: *d_out = p.d_fold; *elems_out = 5 * n; break;
        case 7: *d_out = p.d_tail; *elems_out = 5 * kTail; break;
        case 8: *d_out = p.d_q; *elems_out = 2 * cs; break;
        case 9: *d_out = p.d_r; *elems_out = cs; break;
        default: set_error("uzk_prover_buffer: which = %d", which); return UZK_ERR_PARAMETER;
    }
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prover_buffer"); }

/* ---- round 1 (prover.rs:151-192) ----------------------------------------------------------------------------------------- */
int uzk_prove_round1(uint64_t prover, uint64_t circuit, const void* witness, const void* wsel, int inputs_on_device,
                     const uint32_t* pi_index, const uint64_t* pi_value, uint32_t pi_count, const uint32_t* hiding,
                     const uint64_t* blinds, uzk_g1_jac* cm_out) try {
    API_LOCK;
    auto pp = find_prover(prover);
    auto cir = find_circuit(circuit);
    if (!pp || !cir) { set_error("uzk_prove_round1: unknown prover or circuit"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    if (!witness || !hiding || !blinds || !cm_out || (pi_count && (!pi_index || !pi_value))) { set_error("uzk_prove_round1: null pointer"); return UZK_ERR_PARAMETER; }
    if (cir->n != p.n) { set_error("uzk_prove_round1: the prover was made for n = %u, the circuit has n = %u", p.n, cir->n); return UZK_ERR_PARAMETER; }
    if (cir->shuffle && !wsel) { set_error("uzk_prove_round1: a shuffle circuit needs the wire selectors"); return UZK_ERR_PARAMETER; }
    const uint32_t n = p.n, m = p.m, B = p.B;
    const uint32_t n_first = wsel ? kWires + kWsel : kWires;
    for (uint32_t i = 0; i < n_first; ++i)
        if (hiding[i] > 3) { set_error("uzk_prove_round1: hiding degree %u (polynomial %u) exceeds 3", hiding[i], i); return UZK_ERR_PARAMETER; }
    {   // round 3 needs every chunk of t to fold onto n coefficients with <= 3 blinds: 5n + 8 <= 5n - 2 + sum of the wires' degrees <= 5n + 11
        uint32_t sum = 0;
        for (uint32_t i = 0; i < kWires; ++i) sum += hiding[i];
        if (sum < 10 || sum > 13) { set_error("uzk_prove_round1: the wires' hiding degrees sum to %u; the device flow covers 10 .. 13 (TurboCS: 13)", sum); return UZK_ERR_PARAMETER; }
    }
    for (uint32_t j = 0; j < pi_count; ++j)
        if (pi_index[j] >= n) { set_error("uzk_prove_round1: public input %u sits at constraint %u of %u", j, pi_index[j], n); return UZK_ERR_PARAMETER; }
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    // a new proof: take the circuit's current tables for all five rounds
    p.round = 0;
    p.circuit = cir;
    { std::lock_guard<std::mutex> lk(cir->mu); p.snap = cir->tables; }
    p.n_first = n_first; p.np = n_first + 2;
    std::memcpy(p.hiding, hiding, n_first * sizeof(uint32_t));
    const hipMemcpyKind kind = inputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    bool pageable = !inputs_on_device;
    // witness [B][5n] (and selectors [B][3n]) into the evaluation slots [B][10][n]
    ROUND_HIP(hipMemcpy2DAsync(p.d_evals, (size_t)kProofSlots * n * sizeof(Fp), witness, (size_t)kWires * n * sizeof(Fp), (size_t)kWires * n * sizeof(Fp), B, kind, c.stream));
    if (wsel) ROUND_HIP(hipMemcpy2DAsync(p.d_evals + (uint64_t)kWires * n, (size_t)kProofSlots * n * sizeof(Fp), wsel, (size_t)kWsel * n * sizeof(Fp), (size_t)kWsel * n * sizeof(Fp), B, kind, c.stream));
    if (pageable && is_pinned_block(witness, (size_t)B * kWires * n * sizeof(Fp)) && (!wsel || is_pinned_block(wsel, (size_t)B * kWsel * n * sizeof(Fp)))) pageable = false;
    if (pageable) ROUND_HIP(hipStreamSynchronize(c.stream));       // the caller may reuse ordinary host memory on return
    // PI evaluations (pi_poly, helpers.rs:111-131): zero, then the online values at their constraint indices
    ROUND_HIP(hipMemset2DAsync(p.evals(0, p.sl_pi()), (size_t)kProofSlots * n * sizeof(Fp), 0, (size_t)n * sizeof(Fp), B, c.stream));
    if (pi_count) {
        const size_t need = (size_t)pi_count * sizeof(uint32_t) + 16 + (size_t)B * pi_count * sizeof(Fp);
        if (p.h_pi_cap < need) {
            if (p.h_pi) { ROUND_HIP(hipStreamSynchronize(c.stream)); (void)hipHostFree(p.h_pi); p.h_pi = nullptr; p.h_pi_cap = 0; }
            ROUND_HIP(hipHostMalloc(&p.h_pi, need + (need >> 1), hipHostMallocDefault));
            p.h_pi_cap = need + (need >> 1);
        }
        // find_position takes the FIRST position whose index matches: keep first occurrences only
        Fp* h_val = static_cast<Fp*>(p.h_pi);
        uint32_t kept = 0;
        std::vector<uint32_t> keep;
        keep.reserve(pi_count);
        {
            std::vector<uint8_t> seen(n, 0);
            for (uint32_t j = 0; j < pi_count; ++j)
                if (!seen[pi_index[j]]) { seen[pi_index[j]] = 1; keep.push_back(j); }
        }
        kept = (uint32_t)keep.size();
        uint32_t* h_idx = reinterpret_cast<uint32_t*>(h_val + (size_t)B * kept);
        for (uint32_t b = 0; b < B; ++b)
            for (uint32_t j = 0; j < kept; ++j) h_val[(size_t)b * kept + j] = as_fp(pi_value)[(size_t)b * pi_count + keep[j]];
        for (uint32_t j = 0; j < kept; ++j) h_idx[j] = pi_index[keep[j]];
        ROUND_TRY(poly_scatter_run(c, p.evals(0, p.sl_pi()), (uint64_t)kProofSlots * n, h_idx, h_val, kept, B));
    }
    // iFFT(n) of every proof's evaluation vectors straight into their 6n-slots.  One proof: the np - 1 vectors that exist; a
    // lockstep batch: all np slots of every proof in one strided batch (slot z is transformed again in round 2)
    if (wsel) {
        if (B == 1) ROUND_TRY(ntt_run(c, p.d_evals, p.d_coefs, n, true, nullptr, kProofSlots - 1, n, m));
        else ROUND_TRY(ntt_run(c, p.d_evals, p.d_coefs, n, true, nullptr, B * kProofSlots, n, m));
    } else {
        // without wire selectors the slots in use are w0..4, pi, z: the PI vector lives at slot 5
        for (uint32_t b = 0; b < B; ++b) ROUND_TRY(ntt_run(c, p.evals(b, 0), p.coefs(b, 0), n, true, nullptr, kWires + 1, n, m));
    }
    // hide_polynomial (helpers.rs:139-158): three blind slots per polynomial, unused ones zero
    std::vector<Fp> tail((size_t)B * n_first * kTail);
    for (uint32_t b = 0; b < B; ++b) {
        const Fp* bl = as_fp(blinds) + (size_t)b * n_first * 3;
        std::vector<Fp> hb((size_t)n_first * 3);
        for (uint32_t i = 0; i < n_first; ++i) {
            for (uint32_t j = 0; j < 3; ++j) hb[i * 3 + j] = j < hiding[i] ? bl[i * 3 + j] : Fr::zero();
            put_tail(tail, (size_t)b * n_first + i, bl + i * 3, hiding[i]);
        }
        ROUND_TRY(poly_hide_batch_run(c, p.coefs(b, 0), m, n, n_first, hb.data(), 3, n));
    }
    ScalarView sv;
    sv.main = p.d_evals; sv.stride = n; sv.n_main = n; sv.tail_n = kTail;
    sv.group = n_first; sv.group_stride = (uint64_t)kProofSlots * n;
    ROUND_TRY(stage_tail(c, tail, &sv.tail));
    std::vector<Jac> cm((size_t)B * n_first);
    ROUND_TRY(commit(*cir, sv, B * n_first, B, cm.data()));
    std::memcpy(cm_out, cm.data(), cm.size() * sizeof(Jac));
    p.round = 1;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prove_round1"); }

/* ---- round 2 (prover.rs:194-209) ----------------------------------------------------------------------------------------- */
int uzk_prove_round2(uint64_t prover, const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinds_z, uzk_g1_jac* cm_z_out) try {
    API_LOCK;
    auto pp = find_prover(prover);
    if (!pp) { set_error("uzk_prove_round2: unknown prover"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    if (!beta || !gamma || !blinds_z || !cm_z_out) { set_error("uzk_prove_round2: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(need_round(p, 1, "uzk_prove_round2"));
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Circuit& cir = *p.circuit;
    const uint32_t n = p.n, m = p.m, B = p.B, z = p.sl_z();
    p.beta.assign(as_fp(beta), as_fp(beta) + B);
    p.gamma.assign(as_fp(gamma), as_fp(gamma) + B);
    for (uint32_t b = 0; b < B; ++b)
        ROUND_TRY(z_poly_device(c, p.evals(b, 0), cir.d_perm, cir.d_group, cir.k, p.beta[b], p.gamma[b], n, kWires, p.evals(b, z)));
    ROUND_TRY(ntt_run(c, p.evals(0, z), p.coefs(0, z), n, true, nullptr, B, (uint64_t)kProofSlots * n, (uint64_t)kProofSlots * m));
    std::vector<Fp> tail((size_t)B * kTail);
    for (uint32_t b = 0; b < B; ++b) {
        ROUND_TRY(poly_hide_batch_run(c, p.coefs(b, z), m, n, 1, as_fp(blinds_z) + (size_t)b * 3, 3, n));
        put_tail(tail, b, as_fp(blinds_z) + (size_t)b * 3, 3);
    }
    ScalarView sv;
    sv.main = p.evals(0, z); sv.stride = (uint64_t)kProofSlots * n; sv.n_main = n; sv.tail_n = kTail;
    ROUND_TRY(stage_tail(c, tail, &sv.tail));
    std::vector<Jac> cm(B);
    ROUND_TRY(commit(cir, sv, B, B, cm.data()));
    std::memcpy(cm_z_out, cm.data(), cm.size() * sizeof(Jac));
    p.round = 2;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prove_round2"); }

/* ---- round 3 (prover.rs:211-239) ----------------------------------------------------------------------------------------- */
int uzk_prove_round3(uint64_t prover, const uint64_t* alpha, const uint64_t* t_rands, uzk_g1_jac* cm_t_out) try {
    API_LOCK;
    auto pp = find_prover(prover);
    if (!pp) { set_error("uzk_prove_round3: unknown prover"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    if (!alpha || !t_rands || !cm_t_out) { set_error("uzk_prove_round3: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(need_round(p, 2, "uzk_prove_round3"));
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, m = p.m, B = p.B, cs = p.cs;
    // t_poly (helpers.rs:223-678): coset FFTs of the proof's polynomials over the 6n domain, the quotient kernel against the
    // circuit's coset tables, the inverse coset transform
    if (p.np == kProofSlots) ROUND_TRY(ntt_run(c, p.d_coefs, p.d_coset, m, false, &cir.k[1], B * kProofSlots));
    else for (uint32_t b = 0; b < B; ++b) ROUND_TRY(ntt_run(c, p.coefs(b, 0), p.coset(b, 0), m, false, &cir.k[1], p.np));
    for (uint32_t b = 0; b < B; ++b) {
        uzk_quotient_args qa;
        std::memset(&qa, 0, sizeof qa);
        qa.n = n; qa.factor = 6;
        for (uint32_t i = 0; i < kWires; ++i) qa.vec[UZK_TQ_W + i] = p.coset(b, i);
        if (cir.shuffle) for (uint32_t i = 0; i < kWsel; ++i) qa.vec[UZK_TQ_WSEL + i] = p.coset(b, kWires + i);
        qa.vec[UZK_TQ_PI] = p.coset(b, p.sl_pi());
        qa.vec[UZK_TQ_Z] = p.coset(b, p.sl_z());
        for (uint32_t s = 0; s < cir.n_slots; ++s) qa.vec[UZK_TQ_Q + s] = tab.s[s].coset;
        std::memcpy(qa.alpha, as_fp(alpha) + b, 32); std::memcpy(qa.beta, &p.beta[b], 32); std::memcpy(qa.gamma, &p.gamma[b], 32);
        std::memcpy(qa.k, cir.k, sizeof cir.k);
        std::memcpy(qa.anemoi_g, &cir.anemoi_g, 32); std::memcpy(qa.anemoi_g_inv, &cir.anemoi_g_inv, 32); std::memcpy(qa.edwards_a, &cir.edwards_a, 32);
        std::memcpy(qa.z_h_inv, cir.z_h_inv, sizeof cir.z_h_inv);
        ROUND_TRY(t_quotient_run(c, &qa, p.d_tq + (uint64_t)b * m));
    }
    ROUND_TRY(ntt_run(c, p.d_tq, p.d_t, m, true, &cir.k1_inv, B));
    // FpPolynomial::from_coefs trims t (helpers.rs:673-677) and its coefs.len() drives the split (helpers.rs:1333).  A satisfied
    // circuit gives deg t = deg z + sum_j deg w_j - n, i.e. 5n - 2 + sum_j hiding_j coefficients: go on with that while the
    // device measures the trimmed lengths into pinned memory; the commit below synchronises, then compare -- and redo the
    // split with the measured length where they differ.
    uint64_t t_expected = 5ull * n - 2;
    for (uint32_t i = 0; i < kWires; ++i) t_expected += p.hiding[i];
    {
        // (uzk_tune("prover_t_cap", 1): the synthetic, unsatisfied circuits of the timing and parity chains -- t is taken as its
        // first t_expected coefficients, as tests/chain_oracle.py does)
        std::vector<uint64_t> cap(16, c.tune_prover_t_cap ? t_expected : (uint64_t)m);
        for (uint32_t b0 = 0; b0 < B; b0 += 16)
            ROUND_TRY(poly_trimmed_len_run(c, p.d_t + (uint64_t)b0 * m, m, cap.data(), std::min<uint32_t>(16, B - b0), p.h_lens + b0, false));
    }
    p.chunk_lens.assign((size_t)B * 5, 0);
    std::vector<Jac> cm((size_t)B * 5);
    auto split_and_commit = [&](const std::vector<uint64_t>& t_len) -> int {
        for (uint32_t b = 0; b < B; ++b) {
            // every chunk must fold onto n coefficients with at most three blinds: n <= coefs.len() <= n + 3
            if (t_len[b] > 5ull * (n + 2) + 1 || t_len[b] < 4ull * (n + 2) + n) {
                set_error("uzk_prove_round3: proof %u: t has %llu coefficients, a satisfied circuit gives %llu (the witness does not satisfy the "
                          "circuit; the reference's apply_blind_factors indexes past its SRS here)", b, (unsigned long long)t_len[b], (unsigned long long)t_expected);
                return UZK_ERR_COMMITMENT;
            }
            uint64_t* cl = p.chunk_lens.data() + (size_t)b * 5;
            UZK_TRY(split_t_run(c, p.d_t + (uint64_t)b * m, t_len[b], n + 2, 5, as_fp(t_rands) + (size_t)b * 5, p.d_chunks + (uint64_t)b * 5 * cs, cs, cl));
            UZK_TRY(fold_blinds_batch_run(c, p.d_chunks + (uint64_t)b * 5 * cs, cs, cl, n, 5, p.d_fold + (uint64_t)b * 5 * n, n, p.d_tail + (uint64_t)b * 5 * kTail, kTail, nullptr));
        }
        UZK_TRY(ntt_run(c, p.d_fold, p.d_fold, n, false, nullptr, B * 5));
        ScalarView sv;
        sv.main = p.d_fold; sv.stride = n; sv.n_main = n; sv.tail = p.d_tail; sv.tail_n = kTail;
        return commit(cir, sv, B * 5, B, cm.data());
    };
    std::vector<uint64_t> t_len(B, t_expected);
    ROUND_TRY(split_and_commit(t_len));
    bool differ = false;
    for (uint32_t b = 0; b < B; ++b) if (p.h_lens[b] != t_len[b]) { t_len[b] = p.h_lens[b]; differ = true; }
    if (differ) ROUND_TRY(split_and_commit(t_len));
    std::memcpy(cm_t_out, cm.data(), cm.size() * sizeof(Jac));
    p.round = 3;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prove_round3"); }

/* ---- round 4 (prover.rs:241-273) ----------------------------------------------------------------------------------------- */
int uzk_prove_round4(uint64_t prover, const uint64_t* zeta, uint64_t* evals_out) try {
    API_LOCK;
    auto pp = find_prover(prover);
    if (!pp) { set_error("uzk_prove_round4: unknown prover"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    if (!zeta || !evals_out) { set_error("uzk_prove_round4: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(need_round(p, 3, "uzk_prove_round4"));
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, B = p.B;
    p.zeta.assign(as_fp(zeta), as_fp(zeta) + B);
    p.zeta_omega.resize(B);
    const uint32_t per = cir.shuffle ? 19 : 15;
    for (uint32_t b = 0; b < B; ++b) {
        p.zeta_omega[b] = Fr::mul(cir.group_gen, p.zeta[b]);
        const void* ptr[19];
        uint64_t len[19];
        uint32_t pt[19], k = 0;
        auto own = [&](uint32_t slot, uint64_t l, uint32_t point) { ptr[k] = p.coefs(b, slot); len[k] = l; pt[k] = point; ++k; };
        auto cirp = [&](uint32_t slot) { ptr[k] = tab.s[slot].poly; len[k] = tab.s[slot].len; pt[k] = 0; ++k; };
        for (uint32_t i = 0; i < kWires; ++i) own(i, n + p.hiding[i], 0);
        for (uint32_t i = 0; i < kWires - 1; ++i) cirp(UZK_CS_S + i);
        cirp(UZK_CS_QPRK + 2);
        cirp(UZK_CS_QPRK + 3);
        own(p.sl_z(), n + 3, 1);
        for (uint32_t i = 0; i < 3; ++i) own(i, n + p.hiding[i], 1);
        if (cir.shuffle) {
            cirp(UZK_CS_QECC);
            for (uint32_t i = 0; i < kWsel; ++i) own(kWires + i, n + p.hiding[kWires + i], 0);
        }
        // a zero polynomial (len 0) evaluates to zero without a launch slot of its own: poly_eval_ptrs handles it
        const Fp points[2] = {p.zeta[b], p.zeta_omega[b]};
        ROUND_TRY(poly_eval_ptrs(c, ptr, len, pt, k, points, 2, reinterpret_cast<Fp*>(evals_out) + (size_t)b * per));
    }
    p.round = 4;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prove_round4"); }

/* ---- round 5 (prover.rs:296-372) ----------------------------------------------------------------------------------------- */
int uzk_prove_round5(uint64_t prover, const uint64_t* r_scalars, const uint64_t* alpha_zeta, const uint64_t* alpha_zeta_omega,
                     uzk_g1_jac* openings_out) try {
    API_LOCK;
    auto pp = find_prover(prover);
    if (!pp) { set_error("uzk_prove_round5: unknown prover"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    if (!r_scalars || !alpha_zeta || !alpha_zeta_omega || !openings_out) { set_error("uzk_prove_round5: null pointer"); return UZK_ERR_PARAMETER; }
    UZK_TRY(need_round(p, 4, "uzk_prove_round5"));
    UZK_TRY(require_ready());
    Ctx& c = ctx();
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, B = p.B, cs = p.cs;
    const uint32_t n_r = cir.shuffle ? 43 : 19;
    for (uint32_t b = 0; b < B; ++b) {
        // r(X) (helpers.rs:1030-1080): q (9), z, the last s, qb, q_prk1, q_prk2, [q_pk (12), q_g (12)], the t chunks (5)
        const void* ptr[43];
        uint64_t len[43];
        uint32_t k = 0;
        auto cirp = [&](uint32_t slot) { ptr[k] = tab.s[slot].poly; len[k] = tab.s[slot].len; ++k; };
        for (uint32_t i = 0; i < 9; ++i) cirp(UZK_CS_Q + i);
        ptr[k] = p.coefs(b, p.sl_z()); len[k] = n + 3; ++k;
        cirp(UZK_CS_S + 4); cirp(UZK_CS_QB); cirp(UZK_CS_QPRK); cirp(UZK_CS_QPRK + 1);
        if (cir.shuffle) {
            for (uint32_t i = 0; i < 12; ++i) cirp(UZK_CS_QPK + i);
            for (uint32_t i = 0; i < 12; ++i) cirp(UZK_CS_QG + i);
        }
        for (uint32_t i = 0; i < 5; ++i) { ptr[k] = p.d_chunks + ((uint64_t)b * 5 + i) * cs; len[k] = p.chunk_lens[(size_t)b * 5 + i]; ++k; }
        Fp* d_r = p.d_r + (uint64_t)b * cs;
        ROUND_TRY(poly_lincomb_run(c, ptr, len, as_fp(r_scalars) + (size_t)b * n_r, k, d_r, n + 3));
        // polys_to_open at zeta (prover.rs:329-347): w (5), s (4), q_prk3, q_prk4, [q_ecc, w_sel (3)], r; at zeta omega: z, w0..2
        k = 0;
        for (uint32_t i = 0; i < kWires; ++i) { ptr[k] = p.coefs(b, i); len[k] = n + p.hiding[i]; ++k; }
        for (uint32_t i = 0; i < kWires - 1; ++i) cirp(UZK_CS_S + i);
        cirp(UZK_CS_QPRK + 2); cirp(UZK_CS_QPRK + 3);
        if (cir.shuffle) {
            cirp(UZK_CS_QECC);
            for (uint32_t i = 0; i < kWsel; ++i) { ptr[k] = p.coefs(b, kWires + i); len[k] = n + p.hiding[kWires + i]; ++k; }
        }
        ptr[k] = d_r; len[k] = n + 3; ++k;
        Fp* d_q = p.d_q + (uint64_t)b * 2 * cs;
        ROUND_TRY(open_quotient_ptrs(c, ptr, len, k, p.zeta[b], as_fp(alpha_zeta)[b], d_q, cs, nullptr));
        k = 0;
        ptr[k] = p.coefs(b, p.sl_z()); len[k] = n + 3; ++k;
        for (uint32_t i = 0; i < 3; ++i) { ptr[k] = p.coefs(b, i); len[k] = n + p.hiding[i]; ++k; }
        ROUND_TRY(open_quotient_ptrs(c, ptr, len, k, p.zeta_omega[b], as_fp(alpha_zeta_omega)[b], d_q + cs, cs, nullptr));
    }
    // degree = q.degree() (pcs.rs:138) = the trimmed length minus one; both openings hold a polynomial of n + 3 coefficients
    // (z and r always have n + 3), so q has n + 2: max_power_of_2 = n, two blinds.  Expected lengths first, the device's
    // measurement checked after the commit, as for t.
    {
        std::vector<uint64_t> cap(16, n + 3);
        for (uint32_t v0 = 0; v0 < 2 * B; v0 += 16)
            ROUND_TRY(poly_trimmed_len_run(c, p.d_q + (uint64_t)v0 * cs, cs, cap.data(), std::min<uint32_t>(16, 2 * B - v0), p.h_lens + B + v0, false));
    }
    std::vector<Jac> cm((size_t)B * 2);
    auto fold_and_commit = [&](const std::vector<uint64_t>& q_len) -> int {
        for (uint32_t v = 0; v < 2 * B; ++v) {
            if (q_len[v] < 1 || max_power_of_2(q_len[v] - 1) != n || q_len[v] > (uint64_t)n + 3) {
                set_error("uzk_prove_round5: opening quotient %u has %llu coefficients; the device flow covers degree n .. n + 2", v, (unsigned long long)q_len[v]);
                return UZK_ERR_COMMITMENT;
            }
        }
        for (uint32_t b = 0; b < B; ++b)
            UZK_TRY(fold_blinds_batch_run(c, p.d_q + (uint64_t)b * 2 * cs, cs, q_len.data() + (size_t)b * 2, n, 2, p.d_fold + (uint64_t)b * 2 * n, n,
                                          p.d_tail + (uint64_t)b * 2 * kTail, kTail, nullptr));
        UZK_TRY(ntt_run(c, p.d_fold, p.d_fold, n, false, nullptr, B * 2));
        ScalarView sv;
        sv.main = p.d_fold; sv.stride = n; sv.n_main = n; sv.tail = p.d_tail; sv.tail_n = kTail;
        return commit(cir, sv, B * 2, B, cm.data());
    };
    std::vector<uint64_t> q_len((size_t)B * 2, (uint64_t)n + 2);
    ROUND_TRY(fold_and_commit(q_len));
    bool differ = false;
    for (uint32_t v = 0; v < 2 * B; ++v) if (p.h_lens[B + v] != q_len[v]) { q_len[v] = p.h_lens[B + v]; differ = true; }
    if (differ) ROUND_TRY(fold_and_commit(q_len));
    std::memcpy(openings_out, cm.data(), cm.size() * sizeof(Jac));
    // the proof is complete: its tables may go (a table set replaced meanwhile is freed here)
    p.round = 0;
    p.snap.reset();
    p.circuit.reset();
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_prove_round5"); }

}  // extern "C"
