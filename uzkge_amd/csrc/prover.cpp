// Circuits and the five prover rounds behind the C ABI (include/uzkge_gpu.h, "circuits and the five prover rounds").
//
// prover_with_lagrange (uzkge/src/plonk/prover.rs:88-394) keeps per-circuit data in PlonkProverParams (indexer.rs:76-138) and one
// proof's polynomials in Vec<Fr>s between its Fiat-Shamir rounds.  Here the first is a Circuit (HBM-resident tables with
// copy-on-write replacement, because refresh_prover_params_public_key -- shuffle/src/gen_params/params.rs:57-129 -- swaps twelve
// of them once per game), the second a Prover (the buffers of up to `batch` proofs advancing in lockstep), and
// round1_lanes .. round5_lanes do what the reference does between two transcript draws for every lane at once: each step of a
// round is ONE launch over the lanes (rounds.hip, the strided NTT / MSM batches).  Two ways lead here: uzk_prove_round1..5 on a
// prover of `batch` proofs (the caller holds the batch), and coalesce.cpp, which gathers the concurrent calls of provers of one
// proof into lanes.  The transcript, the prng and r_poly's O(1) scalars stay with the caller.
#include <algorithm>
#include <cstring>

#include "host_math.hpp"
#include "prover.hpp"

namespace uzk {

constexpr int kBatchWindowBits = 15;                             // window width of the lockstep-batch commit table (profiles/r04_rounds_window_sweep*.txt)

DevBlock::~DevBlock() {
    if (!p) return;
    (void)hipSetDevice(device);
    (void)hipFree(p);                                              // waits for the device: nothing still reads the block
}
int dev_block(size_t bytes, std::shared_ptr<DevBlock>* out) {
    auto b = std::make_shared<DevBlock>();
    b->device = ctx().device;
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
CommitBases::~CommitBases() {
    (void)hipSetDevice(device);
    if (wide.d_table) (void)hipFree(wide.d_table);
    if (plain.d_table) (void)hipFree(plain.d_table);
    if (plain.d_points) (void)hipFree(plain.d_points);             // `wide` names the same points
}
Prover::~Prover() {
    (void)hipSetDevice(device);
    if (h_lens) (void)hipHostFree(h_lens);
    if (h_evals) (void)hipHostFree(h_evals);
    if (h_pi) (void)hipHostFree(h_pi);
    args.release();
}

namespace {

struct Registry {
    std::mutex mu;
    std::map<uint64_t, std::shared_ptr<Circuit>> circuits;
    std::map<uint64_t, std::shared_ptr<Prover>> provers;
    uint64_t next = 1;
};
Registry& reg() {
    static Registry r;
    return r;
}
std::shared_ptr<Prover> find_prover(uint64_t h) {
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.provers.find(h);
    return it == r.provers.end() ? nullptr : it->second;
}

const Fp* as_fp(const uint64_t* p) { return reinterpret_cast<const Fp*>(p); }
Fp fp_of(const uint64_t* w) { Fp f; std::memcpy(&f, w, sizeof f); return f; }
bool fp_eq(const Fp& a, const Fp& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// largest power of two <= degree, as the reference's loops compute it (pcs.rs:139-145, helpers.rs:1367-1373); 0 for degree 0
uint64_t max_power_of_2(uint64_t degree) {
    uint64_t p = 1;
    if (degree == 0) return 0;
    while (p * 2 <= degree) p *= 2;
    return p;
}

// coefficient forms [count][n] (zero padded) -> coset evaluations [count][6n]: zero-padded copy and ONE batched coset FFT
// (indexer.rs:316-470: `coset_fft_with_domain(&domain_m, &k[1])` per table)
int derive_cosets(Ctx& c, uint32_t n, const Fp& k1, const Fp* polys, Fp* cosets, uint32_t count) {
    const uint64_t m = 6ull * n;
    UZK_HIP(hipMemsetAsync(cosets, 0, (size_t)count * m * sizeof(Fp), c.stream));
    UZK_HIP(hipMemcpy2DAsync(cosets, (size_t)m * sizeof(Fp), polys, (size_t)n * sizeof(Fp), (size_t)n * sizeof(Fp), count, hipMemcpyDeviceToDevice, c.stream));
    return ntt_run(c, cosets, cosets, m, false, &k1, count, 0, 0, /*in_len: n coefficients, zeros beyond*/ n);
}

// Installs `count` polynomials whose coefficient forms already sit in blk (polys area: [count][n], zero padded) as slots
// first .. first + count of a new table set.
int derive_and_install(Ctx& c, Circuit& cir, std::shared_ptr<DevBlock> blk, uint32_t first, uint32_t count, const uint64_t* lens) {
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
int polys_from_evals(Ctx& c, const Ctx::Srs* srs, uint32_t n, uint32_t count, const uint64_t* evals, Fp* d_polys, std::vector<uint64_t>& lens, Jac* cms) {
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

bool slot_range_ok(const Circuit& cir, uint32_t first, uint32_t count) {
    return count > 0 && first < cir.n_slots && count <= cir.n_slots - first;
}

// uploads `count` coefficient forms (host) into a fresh block and installs them
int upload_and_install(Ctx& c, Circuit& cir, uint32_t first, uint32_t count, const uint64_t* const* polys, const uint64_t* lens) {
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

// commit = lagrange_pcs.commit(evals) + apply_blind_factors (prover.rs:132-142) as ONE batched MSM with tail scalars.
// Two window tables serve two regimes (measured): a single proof's commits are chains of dependent additions on
// a mostly idle chip -- narrow windows, one workgroup per (vector, window), shortest chain; the 8 x B / 5 x B / 2 x B vectors of
// a lockstep batch fill the chip, and there the NUMBER of additions counts: 15-bit windows over one shared bucket set per
// vector take 17 additions per scalar instead of 32.
int commit(const Circuit& cir, const ScalarView& sv, uint32_t batch, uint32_t lockstep, Jac* out) {
    const CommitBases& cb = *cir.bases;
    // (round 6 re-check: the wide table for ONE proof's commits of >= 8 / 5 / 2 / 1 vectors: 2.01 / 2.10 / 2.25 / 2.43 ms per proof against 1.94)
    return msm_dispatch_view(lockstep >= 2 && cb.wide.d_table ? cb.wide : cb.plain, 0, sv, (size_t)sv.n_main + sv.tail_n, batch, out);
}

// pinned, device-visible copy of a round's tail scalars; the commit that reads it synchronises before the round returns
int stage_tail(Ctx& c, const std::vector<Fp>& tail, const Fp** out) {
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
void put_tail(std::vector<Fp>& tail, size_t vec, const Fp* blinds, uint32_t hd) {
    for (uint32_t j = 0; j < 3; ++j) {
        const Fp b = j < hd ? blinds[j] : Fr::zero();
        tail[vec * kTail + j] = b;
        tail[vec * kTail + 3 + j] = Fr::neg(b);
    }
}

int need_round(const Prover& p, int done, const char* who) {
    if (p.round != done || !p.circuit) { set_error("%s: the prover has completed %d round(s) of its proof, this call needs %d", who, p.round, done); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}
int need_owner(const Prover& p, const Ctx& c, const char* who) {
    if (p.owner != &c) { set_error("%s: the proof began on another context (a proof's rounds are ordered on ONE context's stream)", who); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}
// a failed round ends the proof: tables are released, the next call must be round 1
int fail(Prover& p, int rc) {
    prover_end_proof(p);
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

// A lane whose own data is at fault: with a status array the lane dies alone, without one the call fails as a whole.
// (expands inside the round functions: `p`, `st`)
#define LANE_FAIL(b, code, ...)                                       \
    do {                                                              \
        set_error(__VA_ARGS__);                                       \
        if (!st) return fail(p, (code));                              \
        if (st[b].rc == UZK_OK) { st[b].rc = (code); st[b].msg = uzk_last_error(); } \
        p.dead[b] = 1;                                                \
    } while (0)

template <class T>
T* push_or_fail(ArgArena& a, size_t count, const T** dev) {
    T* h = a.push<T>(count, dev);
    if (!h) set_error("prover: the round's argument block is full (%zu of %zu bytes used)", a.used, a.cap);
    return h;
}
#define ARG_PUSH(var, dvar, T, count)                       \
    const T* dvar = nullptr;                                \
    T* var = push_or_fail<T>(p.args, (count), &dvar);       \
    if (!var) return fail(p, UZK_ERR_PARAMETER)

bool contiguous(const void* const* ptrs, uint32_t k, size_t bytes) {
    for (uint32_t b = 1; b < k; ++b)
        if (static_cast<const char*>(ptrs[b]) != static_cast<const char*>(ptrs[0]) + (size_t)b * bytes) return false;
    return true;
}

}  // namespace

std::shared_ptr<Circuit> find_circuit(uint64_t h) {
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.circuits.find(h);
    return it == r.circuits.end() ? nullptr : it->second;
}
uint32_t evals_per_proof(const Circuit& cir) { return cir.shuffle ? 19 : 15; }
uint32_t r_scalars_per_proof(const Circuit& cir) { return cir.shuffle ? 43 : 19; }

void prover_end_proof(Prover& p) {
    p.round = 0;
    p.k = 0;
    p.owner = nullptr;
    p.snap.reset();
    p.circuit.reset();
}

void prover_release_all() {
    Registry& r = reg();
    std::map<uint64_t, std::shared_ptr<Circuit>> cs;
    std::map<uint64_t, std::shared_ptr<Prover>> ps;
    {
        std::lock_guard<std::mutex> lk(r.mu);
        cs.swap(r.circuits);
        ps.swap(r.provers);
    }
    devices_synchronize();
    ps.clear();
    cs.clear();
}

int prover_alloc(Ctx& c, uint32_t n, uint32_t B, std::shared_ptr<Prover>* out) {
    auto p = std::make_shared<Prover>();
    p->n = n; p->m = 6 * n; p->cs = n + 8; p->B = B; p->device = c.device;
    const uint64_t m = p->m, cs = p->cs;
    const uint64_t elems = B * ((uint64_t)kProofSlots * n + 2ull * kProofSlots * m + 2 * m + 5 * cs + 5ull * n + 5 * kTail + 2 * cs + cs + 2 * cs);
    const size_t words = (size_t)B * 19 * sizeof(uint32_t) + 2 * 2 * (size_t)B * sizeof(uint64_t);
    UZK_TRY(dev_block(elems * sizeof(Fp) + words + 64, &p->blk));
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
    p->d_r = q; q += B * cs;
    p->d_h = q; q += B * 2 * cs;
    p->d_trim = reinterpret_cast<uint64_t*>(q);
    p->d_counters = reinterpret_cast<uint32_t*>(p->d_trim + 2 * 2 * (size_t)B);
    // coefficient slots are zero beyond what a round writes (the coset FFTs read all 6n elements); the evaluation slots must
    // hold field elements from the start (a lockstep batch transforms slot z before round 2 has filled it); counters and
    // trimmed-length sets start at zero
    UZK_HIP(hipMemsetAsync(p->blk->p, 0, p->blk->bytes, c.stream));
    UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_lens), 3 * (size_t)B * sizeof(uint64_t), hipHostMallocDefault));
    UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_evals), (size_t)B * 19 * sizeof(Fp), hipHostMallocDefault));
    UZK_TRY(p->args.init(8192 + 4096 * (size_t)B));
    UZK_HIP(hipStreamSynchronize(c.stream));
    *out = p;
    return UZK_OK;
}

int prover_move_lane(Ctx& c, const Prover& src, uint32_t from, Prover& dst, uint32_t to) {
    if (src.n != dst.n || from >= src.B || to >= dst.B) { set_error("prover_move_lane: bad lanes"); return UZK_ERR_PARAMETER; }
    const uint64_t n = src.n, m = src.m, cs = src.cs;
    struct Part { const Fp* s; Fp* d; uint64_t per; };
    const Part parts[] = {
        {src.d_evals, dst.d_evals, kProofSlots * n}, {src.d_coefs, dst.d_coefs, kProofSlots * m}, {src.d_coset, dst.d_coset, kProofSlots * m},
        {src.d_t, dst.d_t, m}, {src.d_chunks, dst.d_chunks, 5 * cs}, {src.d_r, dst.d_r, cs},
    };
    for (const Part& pt : parts)
        UZK_HIP(hipMemcpyAsync(pt.d + (uint64_t)to * pt.per, pt.s + (uint64_t)from * pt.per, pt.per * sizeof(Fp), hipMemcpyDeviceToDevice, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    // the host side of the proof in flight
    dst.round = src.round; dst.k = std::max(dst.k, to + 1);
    dst.circuit = src.circuit; dst.snap = src.snap; dst.n_first = src.n_first; dst.np = src.np;
    std::memcpy(dst.hiding, src.hiding, sizeof dst.hiding);
    auto put = [&](std::vector<Fp>& d, const std::vector<Fp>& s) { if (from < s.size()) { if (d.size() <= to) d.resize(to + 1); d[to] = s[from]; } };
    put(dst.beta, src.beta); put(dst.gamma, src.gamma); put(dst.zeta, src.zeta); put(dst.zeta_omega, src.zeta_omega);
    if (src.chunk_lens.size() >= (size_t)(from + 1) * 5) {
        if (dst.chunk_lens.size() < (size_t)(to + 1) * 5) dst.chunk_lens.resize((size_t)(to + 1) * 5);
        std::copy_n(src.chunk_lens.begin() + (size_t)from * 5, 5, dst.chunk_lens.begin() + (size_t)to * 5);
    }
    if (dst.dead.size() <= to) dst.dead.resize(to + 1, 0);
    dst.dead[to] = 0;
    return UZK_OK;
}

int round1_check(const Circuit& cir, uint32_t n, bool has_wsel, const uint32_t* pi_index, uint32_t pi_count, const uint32_t* hiding) {
    if (cir.n != n) { set_error("uzk_prove_round1: the prover was made for n = %u, the circuit has n = %u", n, cir.n); return UZK_ERR_PARAMETER; }
    if (cir.shuffle && !has_wsel) { set_error("uzk_prove_round1: a shuffle circuit needs the wire selectors"); return UZK_ERR_PARAMETER; }
    const uint32_t n_first = has_wsel ? kWires + kWsel : kWires;
    for (uint32_t i = 0; i < n_first; ++i)
        if (hiding[i] > 3) { set_error("uzk_prove_round1: hiding degree %u (polynomial %u) exceeds 3", hiding[i], i); return UZK_ERR_PARAMETER; }
    // round 3 needs every chunk of t to fold onto n coefficients with <= 3 blinds: 5n + 8 <= 5n - 2 + sum of the wires' degrees <= 5n + 11
    uint32_t sum = 0;
    for (uint32_t i = 0; i < kWires; ++i) sum += hiding[i];
    if (sum < 10 || sum > 13) { set_error("uzk_prove_round1: the wires' hiding degrees sum to %u; the device flow covers 10 .. 13 (TurboCS: 13)", sum); return UZK_ERR_PARAMETER; }
    for (uint32_t j = 0; j < pi_count; ++j)
        if (pi_index[j] >= n) { set_error("uzk_prove_round1: public input %u sits at constraint %u of %u", j, pi_index[j], n); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}

/* ---- round 1 (prover.rs:151-192) ----------------------------------------------------------------------------------------- */
int round1_lanes(Ctx& c, Prover& p, const std::shared_ptr<Circuit>& cir, uint32_t k, const Lane1* L, int inputs_on_device, const uint32_t* pi_index,
                 uint32_t pi_count, const uint32_t* hiding, LaneStatus* st) {
    (void)st;                                                      // nothing in round 1 is one lane's fault
    const uint32_t n = p.n, m = p.m;
    const bool has_wsel = L[0].wsel != nullptr;
    const uint32_t n_first = has_wsel ? kWires + kWsel : kWires;
    if (k == 0 || k > p.B || cir->device != p.device || c.device != p.device) { set_error("uzk_prove_round1: prover, circuit and context are not on one device"); return UZK_ERR_PARAMETER; }
    // a new proof: take the circuit's current tables for all five rounds
    p.round = 0;
    p.k = k;
    p.owner = &c;
    p.circuit = cir;
    { std::lock_guard<std::mutex> lk(cir->mu); p.snap = cir->tables; }
    p.n_first = n_first; p.np = n_first + 2;
    std::memcpy(p.hiding, hiding, n_first * sizeof(uint32_t));
    p.dead.assign(k, 0);
    p.args.reset();
    // hide_polynomial's blinds (helpers.rs:139-158): three slots per polynomial, unused ones zero; the PI polynomial rides in the
    // same launch with zero blinds, so its slots [n, n + 3) never keep what an earlier proof of this prover wrote there
    const uint32_t slots = n_first + 1;
    ARG_PUSH(hb, d_hb, Fp, (size_t)k * slots * 3);
    std::vector<Fp> tail((size_t)k * n_first * kTail);
    for (uint32_t b = 0; b < k; ++b) {
        for (uint32_t i = 0; i < n_first; ++i) {
            for (uint32_t j = 0; j < 3; ++j) hb[((size_t)b * slots + i) * 3 + j] = j < hiding[i] ? L[b].blinds[i * 3 + j] : Fr::zero();
            put_tail(tail, (size_t)b * n_first + i, L[b].blinds + i * 3, hiding[i]);
        }
        for (uint32_t j = 0; j < 3; ++j) hb[((size_t)b * slots + n_first) * 3 + j] = Fr::zero();
    }
    ROUND_TRY(p.args.upload(c.stream));
    // witness [5n] (and selectors [3n]) of every lane into its evaluation slots [10][n]
    const hipMemcpyKind kind = inputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // (Measured and not kept, round 6: ONE kernel reading every lane's pinned witness over PCIe and zeroing the PI slot, instead of these
    // 2 k copies + a memset -- 10 % SLOWER, 1460 against 1626 proofs/s lockstep8: a kernel that waits on PCIe holds a compute queue
    // for the 0.7 ms the copy engines spend beside the other three launch sequences.  profiles/EXPERIMENTS.md, round 6.)
    {
        std::vector<const void*> w(k), s(k);
        for (uint32_t b = 0; b < k; ++b) { w[b] = L[b].witness; s[b] = L[b].wsel; }
        const size_t wb = (size_t)kWires * n * sizeof(Fp), sb = (size_t)kWsel * n * sizeof(Fp), lane = (size_t)kProofSlots * n * sizeof(Fp);
        if (contiguous(w.data(), k, wb)) ROUND_HIP(hipMemcpy2DAsync(p.d_evals, lane, w[0], wb, wb, k, kind, c.stream));
        else for (uint32_t b = 0; b < k; ++b) ROUND_HIP(hipMemcpyAsync(p.evals(b, 0), w[b], wb, kind, c.stream));
        if (has_wsel) {
            if (contiguous(s.data(), k, sb)) ROUND_HIP(hipMemcpy2DAsync(p.evals(0, kWires), lane, s[0], sb, sb, k, kind, c.stream));
            else for (uint32_t b = 0; b < k; ++b) ROUND_HIP(hipMemcpyAsync(p.evals(b, kWires), s[b], sb, kind, c.stream));
        }
        bool pageable = false;
        if (!inputs_on_device)
            for (uint32_t b = 0; b < k && !pageable; ++b) pageable = !is_pinned_block(w[b], wb) || (has_wsel && !is_pinned_block(s[b], sb));
        if (pageable) ROUND_HIP(hipStreamSynchronize(c.stream));   // the caller may reuse ordinary host memory on return
    }
    // PI evaluations (pi_poly, helpers.rs:111-131): zero, then the online values at their constraint indices
    ROUND_HIP(hipMemset2DAsync(p.evals(0, p.sl_pi()), (size_t)kProofSlots * n * sizeof(Fp), 0, (size_t)n * sizeof(Fp), k, c.stream));
    if (pi_count) {
        const size_t need = (size_t)pi_count * sizeof(uint32_t) + 16 + (size_t)k * pi_count * sizeof(Fp);
        if (p.h_pi_cap < need) {
            if (p.h_pi) { ROUND_HIP(hipStreamSynchronize(c.stream)); (void)hipHostFree(p.h_pi); p.h_pi = nullptr; p.h_pi_cap = 0; }
            ROUND_HIP(hipHostMalloc(&p.h_pi, need + (need >> 1), hipHostMallocDefault));
            p.h_pi_cap = need + (need >> 1);
        }
        // find_position takes the FIRST position whose index matches: keep first occurrences only
        Fp* h_val = static_cast<Fp*>(p.h_pi);
        std::vector<uint32_t> keep;
        keep.reserve(pi_count);
        {
            std::vector<uint8_t> seen(n, 0);
            for (uint32_t j = 0; j < pi_count; ++j)
                if (!seen[pi_index[j]]) { seen[pi_index[j]] = 1; keep.push_back(j); }
        }
        const uint32_t kept = (uint32_t)keep.size();
        uint32_t* h_idx = reinterpret_cast<uint32_t*>(h_val + (size_t)k * kept);
        for (uint32_t b = 0; b < k; ++b)
            for (uint32_t j = 0; j < kept; ++j) h_val[(size_t)b * kept + j] = L[b].pi_value[keep[j]];
        for (uint32_t j = 0; j < kept; ++j) h_idx[j] = pi_index[keep[j]];
        ROUND_TRY(poly_scatter_run(c, p.evals(0, p.sl_pi()), (uint64_t)kProofSlots * n, h_idx, h_val, kept, k));
    }
    // iFFT(n) of every proof's evaluation vectors straight into their 6n-slots.  One proof: the np - 1 vectors that exist; a
    // lockstep batch: all ten slots of every lane in one strided batch (slot z is transformed again in round 2, unused slots
    // hold zeros)
    if (k == 1) ROUND_TRY(ntt_run(c, p.d_evals, p.d_coefs, n, true, nullptr, p.np - 1, n, m));
    else ROUND_TRY(ntt_run(c, p.d_evals, p.d_coefs, n, true, nullptr, k * kProofSlots, n, m));
    ROUND_TRY(hide_lanes(c, p.d_coefs, (uint64_t)kProofSlots * m, m, n, slots, k, d_hb));
    ScalarView sv;
    sv.main = p.d_evals; sv.stride = n; sv.n_main = n; sv.tail_n = kTail;
    sv.group = n_first; sv.group_stride = (uint64_t)kProofSlots * n;
    ROUND_TRY(stage_tail(c, tail, &sv.tail));
    std::vector<Jac> cm((size_t)k * n_first);
    ROUND_TRY(commit(*cir, sv, k * n_first, k, cm.data()));
    for (uint32_t b = 0; b < k; ++b) std::memcpy(L[b].cm_out, cm.data() + (size_t)b * n_first, n_first * sizeof(Jac));
    p.round = 1;
    return UZK_OK;
}

/* ---- round 2 (prover.rs:194-209) ----------------------------------------------------------------------------------------- */
int round2_lanes(Ctx& c, Prover& p, const Lane2* L, LaneStatus* st) {
    Circuit& cir = *p.circuit;
    const uint32_t n = p.n, m = p.m, k = p.k, z = p.sl_z();
    p.args.reset();
    p.beta.resize(k); p.gamma.resize(k);
    ARG_PUSH(bg, d_bg, Fp, (size_t)k * 2);
    ARG_PUSH(bz, d_bz, Fp, (size_t)k * 3);
    std::vector<Fp> tail((size_t)k * kTail);
    for (uint32_t b = 0; b < k; ++b) {
        p.beta[b] = *L[b].beta; p.gamma[b] = *L[b].gamma;
        bg[2 * b] = p.beta[b]; bg[2 * b + 1] = p.gamma[b];
        for (uint32_t j = 0; j < 3; ++j) bz[b * 3 + j] = L[b].blinds_z[j];
        put_tail(tail, b, L[b].blinds_z, 3);
    }
    ROUND_TRY(p.args.upload(c.stream));
    std::vector<uint8_t> ok(k, 1);
    ROUND_TRY(z_poly_lanes(c, p.args, p.d_evals, (uint64_t)kProofSlots * n, cir.d_perm, cir.d_group, cir.k, d_bg, n, kWires, k, p.evals(0, z), (uint64_t)kProofSlots * n, ok.data()));
    for (uint32_t b = 0; b < k; ++b)
        if (!ok[b] && !p.dead[b]) LANE_FAIL(b, UZK_ERR_PARAMETER, "uzk_prove_round2: proof %u: a permutation denominator is zero", b);
    ROUND_TRY(ntt_run(c, p.evals(0, z), p.coefs(0, z), n, true, nullptr, k, (uint64_t)kProofSlots * n, (uint64_t)kProofSlots * m));
    ROUND_TRY(hide_lanes(c, p.coefs(0, z), (uint64_t)kProofSlots * m, m, n, 1, k, d_bz));
    ScalarView sv;
    sv.main = p.evals(0, z); sv.stride = (uint64_t)kProofSlots * n; sv.n_main = n; sv.tail_n = kTail;
    ROUND_TRY(stage_tail(c, tail, &sv.tail));
    std::vector<Jac> cm(k);
    ROUND_TRY(commit(cir, sv, k, k, cm.data()));
    for (uint32_t b = 0; b < k; ++b) if (!p.dead[b]) std::memcpy(L[b].cm_z_out, &cm[b], sizeof(Jac));
    p.round = 2;
    return UZK_OK;
}

/* ---- round 3 (prover.rs:211-239) ----------------------------------------------------------------------------------------- */
int round3_lanes(Ctx& c, Prover& p, const Lane3* L, LaneStatus* st) {
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, m = p.m, k = p.k, cs = p.cs;
    p.args.reset();
    const size_t lane_bytes = quotient_lane_bytes();
    ARG_PUSH(ql, d_ql, char, (size_t)k * lane_bytes);
    ARG_PUSH(rands, d_rands, Fp, (size_t)k * 5);
    for (uint32_t b = 0; b < k; ++b) {
        quotient_lane(*L[b].alpha, p.beta[b], p.gamma[b], cir.k, ql + (size_t)b * lane_bytes);
        for (uint32_t i = 0; i < 5; ++i) rands[b * 5 + i] = L[b].t_rands[i];
    }
    ROUND_TRY(p.args.upload(c.stream));
    // t_poly (helpers.rs:223-678): coset FFTs of the proof's polynomials over the 6n domain, the quotient kernel against the
    // circuit's coset tables, the inverse coset transform
    // One launch sequence for every lane: the lanes' slots lie a fixed stride apart.  A circuit without wire selectors uses seven of
    // a lane's ten slots; several lanes then transform the three idle ones along (zeros since the allocation, or what another circuit's
    // proof left there: never read) rather than launch once per lane -- 30 % more points in ONE launch against k launches.
    // Every slot holds at most n + 3 coefficients (n from the iFFT, <= 3 blinds from hide_polynomial; zeros beyond since the
    // allocation): the transform is told so and neither reads nor combines the two zero thirds of its 6n-point inputs (ntt_run in_len).
    const uint64_t coef_len = (uint64_t)n + 8;
    if (p.np == kProofSlots || k > 1) ROUND_TRY(ntt_run(c, p.d_coefs, p.d_coset, m, false, &cir.k[1], k * kProofSlots, 0, 0, coef_len));
    else ROUND_TRY(ntt_run(c, p.coefs(0, 0), p.coset(0, 0), m, false, &cir.k[1], p.np, 0, 0, coef_len));
    {
        uzk_quotient_args qa;
        std::memset(&qa, 0, sizeof qa);
        qa.n = n; qa.factor = 6;
        for (uint32_t i = 0; i < kWires; ++i) qa.vec[UZK_TQ_W + i] = p.coset(0, i);
        if (cir.shuffle) for (uint32_t i = 0; i < kWsel; ++i) qa.vec[UZK_TQ_WSEL + i] = p.coset(0, kWires + i);
        qa.vec[UZK_TQ_PI] = p.coset(0, p.sl_pi());
        qa.vec[UZK_TQ_Z] = p.coset(0, p.sl_z());
        for (uint32_t s = 0; s < cir.n_slots; ++s) qa.vec[UZK_TQ_Q + s] = tab.s[s].coset;
        std::memcpy(qa.anemoi_g, &cir.anemoi_g, 32); std::memcpy(qa.anemoi_g_inv, &cir.anemoi_g_inv, 32); std::memcpy(qa.edwards_a, &cir.edwards_a, 32);
        std::memcpy(qa.z_h_inv, cir.z_h_inv, sizeof cir.z_h_inv);
        ROUND_TRY(t_quotient_lanes(c, &qa, (uint64_t)kProofSlots * m, d_ql, k, p.d_tq, m));
    }
    ROUND_TRY(ntt_run(c, p.d_tq, p.d_t, m, true, &cir.k1_inv, k));
    // FpPolynomial::from_coefs trims t (helpers.rs:673-677) and its coefs.len() drives the split (helpers.rs:1333).  A satisfied
    // circuit gives deg t = deg z + sum_j deg w_j - n, i.e. 5n - 2 + sum_j hiding_j coefficients: go on with that while the
    // device measures the trimmed lengths into pinned memory; the commit below synchronises, then compare -- and redo the
    // split with the measured length where they differ.
    uint64_t t_expected = 5ull * n - 2;
    for (uint32_t i = 0; i < kWires; ++i) t_expected += p.hiding[i];
    // (uzk_test_circuit_truncate_t: the synthetic, unsatisfied circuits of the timing and parity chains -- t is taken as its
    // first t_expected coefficients, as tests/chain_oracle.py does)
    ROUND_TRY(trimmed_len_lanes(c, p.d_t, m, cir.truncate_t ? t_expected : (uint64_t)m, k, p.d_trim, 2 * p.B, &p.trim_flip, p.h_lens));
    p.chunk_lens.assign((size_t)k * 5, 0);
    std::vector<Jac> cm((size_t)k * 5);
    auto split_and_commit = [&](const std::vector<uint64_t>& t_len) -> int {
        const uint32_t* d_tl = nullptr;
        uint32_t* tl = push_or_fail<uint32_t>(p.args, k, &d_tl);
        const uint32_t* d_cl = nullptr;
        uint32_t* cl = push_or_fail<uint32_t>(p.args, (size_t)k * 5, &d_cl);
        if (!tl || !cl) return UZK_ERR_PARAMETER;
        for (uint32_t b = 0; b < k; ++b) {
            tl[b] = (uint32_t)t_len[b];
            // split_t_and_commit (helpers.rs:1335-1363): chunks of n + 2, each but the last one coefficient longer (its blind)
            const uint64_t chunk = (uint64_t)n + 2, last_start = 4 * chunk;
            for (uint32_t i = 0; i < 4; ++i) p.chunk_lens[(size_t)b * 5 + i] = chunk + 1;
            p.chunk_lens[(size_t)b * 5 + 4] = t_len[b] > last_start ? t_len[b] - last_start : 1;
            for (uint32_t i = 0; i < 5; ++i) cl[b * 5 + i] = (uint32_t)p.chunk_lens[(size_t)b * 5 + i];
        }
        UZK_TRY(p.args.upload(c.stream));
        UZK_TRY(split_t_lanes(c, p.d_t, m, d_tl, (uint64_t)n + 2, 5, d_rands, k, p.d_chunks, cs));
        UZK_TRY(fold_blinds_lanes(c, p.d_chunks, cs, d_cl, n, 5 * k, p.d_fold, n, p.d_tail, kTail));
        UZK_TRY(ntt_run(c, p.d_fold, p.d_fold, n, false, nullptr, k * 5));
        ScalarView sv;
        sv.main = p.d_fold; sv.stride = n; sv.n_main = n; sv.tail = p.d_tail; sv.tail_n = kTail;
        return commit(cir, sv, k * 5, k, cm.data());
    };
    std::vector<uint64_t> t_len(k, t_expected);
    ROUND_TRY(split_and_commit(t_len));
    bool differ = false;
    for (uint32_t b = 0; b < k; ++b) {
        if (p.dead[b]) continue;
        const uint64_t got = p.h_lens[b];
        // every chunk must fold onto n coefficients with at most three blinds: n <= coefs.len() <= n + 3
        if (got > 5ull * (n + 2) + 1 || got < 4ull * (n + 2) + n) {
            LANE_FAIL(b, UZK_ERR_COMMITMENT,
                      "uzk_prove_round3: proof %u: t has %llu coefficients, a satisfied circuit gives %llu (the witness does not satisfy the "
                      "circuit; the reference's apply_blind_factors indexes past its SRS here)", b, (unsigned long long)got, (unsigned long long)t_expected);
            continue;
        }
        if (got != t_len[b]) { t_len[b] = got; differ = true; }
    }
    if (differ) ROUND_TRY(split_and_commit(t_len));
    for (uint32_t b = 0; b < k; ++b) if (!p.dead[b]) std::memcpy(L[b].cm_t_out, cm.data() + (size_t)b * 5, 5 * sizeof(Jac));
    p.round = 3;
    return UZK_OK;
}

/* ---- round 4 (prover.rs:241-273) ----------------------------------------------------------------------------------------- */
int round4_lanes(Ctx& c, Prover& p, const Lane4* L, LaneStatus* st) {
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, m = p.m, k = p.k;
    const uint32_t per = evals_per_proof(cir);
    for (uint32_t b = 0; b < k; ++b)
        if (!p.dead[b] && L[b].evals_cap < per) LANE_FAIL(b, UZK_ERR_PARAMETER, "uzk_prove_round4: evals_out holds %zu elements, a proof of this circuit gives %u", L[b].evals_cap, per);
    p.args.reset();
    p.zeta.resize(k); p.zeta_omega.resize(k);
    ARG_PUSH(pts, d_pts, Fp, (size_t)k * 2);
    for (uint32_t b = 0; b < k; ++b) {
        p.zeta[b] = *L[b].zeta;
        p.zeta_omega[b] = Fr::mul(cir.group_gen, p.zeta[b]);
        pts[2 * b] = p.zeta[b]; pts[2 * b + 1] = p.zeta_omega[b];
    }
    const size_t eb = eval_poly_bytes();
    ARG_PUSH(ep, d_ep, char, (size_t)per * eb);
    uint32_t cnt = 0;
    uint64_t max_len = 1;
    const uint64_t own_stride = (uint64_t)kProofSlots * m;
    auto own = [&](uint32_t slot, uint64_t l, uint32_t point) { eval_poly_fill(ep + (size_t)cnt++ * eb, p.coefs(0, slot), own_stride, l, point); max_len = std::max(max_len, l); };
    auto cirp = [&](uint32_t slot) { eval_poly_fill(ep + (size_t)cnt++ * eb, tab.s[slot].poly, 0, tab.s[slot].len, 0); max_len = std::max(max_len, tab.s[slot].len); };
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
    ROUND_TRY(p.args.upload(c.stream));
    ROUND_TRY(poly_eval_lanes(c, d_ep, per, max_len, d_pts, k, p.d_counters, p.h_evals));
    ROUND_HIP(hipStreamSynchronize(c.stream));
    for (uint32_t b = 0; b < k; ++b) if (!p.dead[b]) std::memcpy(L[b].evals_out, p.h_evals + (size_t)b * per, per * sizeof(Fp));
    p.round = 4;
    return UZK_OK;
}

/* ---- round 5 (prover.rs:296-372) ----------------------------------------------------------------------------------------- */
int round5_lanes(Ctx& c, Prover& p, const Lane5* L, LaneStatus* st) {
    Circuit& cir = *p.circuit;
    const TableSet& tab = *p.snap;
    const uint32_t n = p.n, m = p.m, k = p.k, cs = p.cs;
    const uint32_t n_r = r_scalars_per_proof(cir);
    const uint64_t own_stride = (uint64_t)kProofSlots * m;
    for (uint32_t b = 0; b < k; ++b)
        if (!p.dead[b] && L[b].r_count != n_r) LANE_FAIL(b, UZK_ERR_PARAMETER, "uzk_prove_round5: %zu r_poly scalars given, a proof of this circuit takes %u", L[b].r_count, n_r);
    // a dead lane goes on with a live neighbour's scalars (its own may be too few to read); nothing of it is kept
    std::vector<uint32_t> src(k);
    {
        uint32_t live = k;
        for (uint32_t b = 0; b < k; ++b) if (!p.dead[b]) { live = b; break; }
        if (live == k) { prover_end_proof(p); return UZK_OK; }     // every lane has failed: nothing left to prove
        for (uint32_t b = 0; b < k; ++b) src[b] = p.dead[b] ? live : b;
    }
    p.args.reset();
    const size_t lb = lin_poly_bytes();
    // r(X) (helpers.rs:1030-1080): q (9), z, the last s, qb, q_prk1, q_prk2, [q_pk (12), q_g (12)], the t chunks (5)
    ARG_PUSH(rp, d_rp, char, (size_t)n_r * lb);
    ARG_PUSH(rl, d_rl, uint32_t, (size_t)k * n_r);
    ARG_PUSH(rs, d_rs, Fp, (size_t)k * n_r);
    {
        uint32_t j = 0;
        std::vector<uint32_t> shared_len(n_r, 0);
        auto cirp = [&](uint32_t slot) { lin_poly_fill(rp + (size_t)j * lb, tab.s[slot].poly, 0); shared_len[j] = (uint32_t)tab.s[slot].len; ++j; };
        for (uint32_t i = 0; i < 9; ++i) cirp(UZK_CS_Q + i);
        const uint32_t j_z = j;
        lin_poly_fill(rp + (size_t)j * lb, p.coefs(0, p.sl_z()), own_stride); shared_len[j] = n + 3; ++j;
        cirp(UZK_CS_S + 4); cirp(UZK_CS_QB); cirp(UZK_CS_QPRK); cirp(UZK_CS_QPRK + 1);
        if (cir.shuffle) {
            for (uint32_t i = 0; i < 12; ++i) cirp(UZK_CS_QPK + i);
            for (uint32_t i = 0; i < 12; ++i) cirp(UZK_CS_QG + i);
        }
        const uint32_t j_chunks = j;
        for (uint32_t i = 0; i < 5; ++i) { lin_poly_fill(rp + (size_t)j * lb, p.d_chunks + (uint64_t)i * cs, 5ull * cs); ++j; }
        (void)j_z;
        for (uint32_t b = 0; b < k; ++b) {
            for (uint32_t t = 0; t < n_r; ++t) { rl[(size_t)b * n_r + t] = shared_len[t]; rs[(size_t)b * n_r + t] = L[src[b]].r_scalars[t]; }
            for (uint32_t i = 0; i < 5; ++i) rl[(size_t)b * n_r + j_chunks + i] = (uint32_t)p.chunk_lens[(size_t)b * 5 + i];
        }
    }
    // polys_to_open at zeta (prover.rs:329-347): w (5), s (4), q_prk3, q_prk4, [q_ecc, w_sel (3)], r; at zeta omega: z, w0..2.
    // h = sum_k alpha^k p_k; the constant sum_k alpha^k p_k(z) batch_prove subtracts first (pcs.rs:124-130) only changes h_0, i.e.
    // the remainder -- never the quotient.
    const uint32_t c0 = cir.shuffle ? 16 : 12, c1 = 4;
    ARG_PUSH(p0, d_p0, char, (size_t)c0 * lb);
    ARG_PUSH(l0, d_l0, uint32_t, (size_t)k * c0);
    ARG_PUSH(s0, d_s0, Fp, (size_t)k * c0);
    ARG_PUSH(p1, d_p1, char, (size_t)c1 * lb);
    ARG_PUSH(l1, d_l1, uint32_t, (size_t)k * c1);
    ARG_PUSH(s1, d_s1, Fp, (size_t)k * c1);
    {
        uint32_t j = 0;
        std::vector<uint32_t> len0(c0, 0);
        auto own0 = [&](uint32_t slot, uint32_t l) { lin_poly_fill(p0 + (size_t)j * lb, p.coefs(0, slot), own_stride); len0[j] = l; ++j; };
        auto cir0 = [&](uint32_t slot) { lin_poly_fill(p0 + (size_t)j * lb, tab.s[slot].poly, 0); len0[j] = (uint32_t)tab.s[slot].len; ++j; };
        for (uint32_t i = 0; i < kWires; ++i) own0(i, n + p.hiding[i]);
        for (uint32_t i = 0; i < kWires - 1; ++i) cir0(UZK_CS_S + i);
        cir0(UZK_CS_QPRK + 2); cir0(UZK_CS_QPRK + 3);
        if (cir.shuffle) {
            cir0(UZK_CS_QECC);
            for (uint32_t i = 0; i < kWsel; ++i) own0(kWires + i, n + p.hiding[kWires + i]);
        }
        lin_poly_fill(p0 + (size_t)j * lb, p.d_r, cs); len0[j] = n + 3; ++j;
        lin_poly_fill(p1, p.coefs(0, p.sl_z()), own_stride);
        for (uint32_t i = 0; i < 3; ++i) lin_poly_fill(p1 + (size_t)(1 + i) * lb, p.coefs(0, i), own_stride);
        for (uint32_t b = 0; b < k; ++b) {
            Fp a = Fr::one();
            for (uint32_t t = 0; t < c0; ++t) { l0[(size_t)b * c0 + t] = len0[t]; s0[(size_t)b * c0 + t] = a; a = Fr::mul(a, *L[b].alpha_zeta); }
            a = Fr::one();
            l1[(size_t)b * c1] = n + 3;
            for (uint32_t i = 0; i < 3; ++i) l1[(size_t)b * c1 + 1 + i] = n + p.hiding[i];
            for (uint32_t t = 0; t < c1; ++t) { s1[(size_t)b * c1 + t] = a; a = Fr::mul(a, *L[b].alpha_zeta_omega); }
        }
    }
    const uint64_t hlen = (uint64_t)n + 3;                         // z and r always have n + 3 coefficients: the longest of both lists
    const size_t db = div_pows_bytes();
    ARG_PUSH(dp, d_dp, char, (size_t)2 * k * db);
    for (uint32_t b = 0; b < k; ++b) {
        div_pows_fill(dp + (size_t)(2 * b) * db, p.zeta[b], open_div_per(hlen));
        div_pows_fill(dp + (size_t)(2 * b + 1) * db, p.zeta_omega[b], open_div_per(hlen));
    }
    ROUND_TRY(p.args.upload(c.stream));
    ROUND_TRY(poly_lincomb_lanes(c, d_rp, n_r, d_rl, d_rs, k, p.d_r, cs, hlen));
    // opening v = 2 b + w (w = 0: at zeta, 1: at zeta omega): h_v at d_h + v cs, q_v at d_q + v cs
    ROUND_TRY(poly_lincomb_lanes(c, d_p0, c0, d_l0, d_s0, k, p.d_h, 2ull * cs, hlen));
    ROUND_TRY(poly_lincomb_lanes(c, d_p1, c1, d_l1, d_s1, k, p.d_h + cs, 2ull * cs, hlen));
    ROUND_TRY(open_div_lanes(c, p.d_h, cs, hlen, d_dp, 2 * k, p.d_q, cs, cs));
    // degree = q.degree() (pcs.rs:138) = the trimmed length minus one; both openings hold a polynomial of n + 3 coefficients
    // (z and r always have n + 3), so q has n + 2: max_power_of_2 = n, two blinds.  Expected lengths first, the device's
    // measurement checked after the commit, as for t.
    ROUND_TRY(trimmed_len_lanes(c, p.d_q, cs, hlen, 2 * k, p.d_trim, 2 * p.B, &p.trim_flip, p.h_lens + p.B));
    std::vector<Jac> cm((size_t)k * 2);
    auto fold_and_commit = [&](const std::vector<uint64_t>& q_len) -> int {
        const uint32_t* d_ql = nullptr;
        uint32_t* ql = push_or_fail<uint32_t>(p.args, (size_t)2 * k, &d_ql);
        if (!ql) return UZK_ERR_PARAMETER;
        for (uint32_t v = 0; v < 2 * k; ++v) ql[v] = (uint32_t)q_len[v];
        UZK_TRY(p.args.upload(c.stream));
        UZK_TRY(fold_blinds_lanes(c, p.d_q, cs, d_ql, n, 2 * k, p.d_fold, n, p.d_tail, kTail));
        UZK_TRY(ntt_run(c, p.d_fold, p.d_fold, n, false, nullptr, k * 2));
        ScalarView sv;
        sv.main = p.d_fold; sv.stride = n; sv.n_main = n; sv.tail = p.d_tail; sv.tail_n = kTail;
        return commit(cir, sv, k * 2, k, cm.data());
    };
    std::vector<uint64_t> q_len((size_t)k * 2, (uint64_t)n + 2);
    ROUND_TRY(fold_and_commit(q_len));
    bool differ = false;
    for (uint32_t b = 0; b < k; ++b) {
        if (p.dead[b]) continue;
        for (uint32_t w = 0; w < 2; ++w) {
            const uint32_t v = 2 * b + w;
            const uint64_t got = p.h_lens[p.B + v];
            if (got < 1 || max_power_of_2(got - 1) != n || got > (uint64_t)n + 3) {
                LANE_FAIL(b, UZK_ERR_COMMITMENT, "uzk_prove_round5: opening quotient %u has %llu coefficients; the device flow covers degree n .. n + 2", v, (unsigned long long)got);
                break;
            }
            if (got != q_len[v]) { q_len[v] = got; differ = true; }
        }
    }
    if (differ) ROUND_TRY(fold_and_commit(q_len));
    for (uint32_t b = 0; b < k; ++b) if (!p.dead[b]) std::memcpy(L[b].openings_out, cm.data() + (size_t)b * 2, 2 * sizeof(Jac));
    // the proof is complete: its tables may go (a table set replaced meanwhile is freed here)
    prover_end_proof(p);
    return UZK_OK;
}

}  // namespace uzk

using namespace uzk;
#define API_LOCK std::lock_guard<std::mutex> _lk(ctx_mutex())

namespace {
// builds the commit bases of a circuit on the calling context's device: points, the narrow table and (automatic mode) the wide one
int make_bases(Ctx& c, const uzk_circuit_desc* desc, std::shared_ptr<CommitBases>* out) {
    const uint32_t n = desc->n;
    auto cb = std::make_shared<CommitBases>();
    cb->device = c.device;
    std::vector<uzk_g1_affine> bases((size_t)n + kTail);
    std::memcpy(bases.data(), desc->lagrange_bases, (size_t)n * sizeof(uzk_g1_affine));
    std::memcpy(bases.data() + n, desc->blind_bases, kTail * sizeof(uzk_g1_affine));
    Ctx::Srs& s = cb->plain;
    s.n = bases.size(); s.owned = true; s.device = c.device;
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&s.d_points), s.n * sizeof(Affine)));
    UZK_HIP(hipMemcpyAsync(s.d_points, bases.data(), s.n * sizeof(Affine), hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    cb->wide = s; cb->wide.owned = false;
    if (desc->precompute) {
        const int cbits = msm_precompute_window_bits(s.n, desc->precompute == 1 ? 0 : (int)desc->precompute);
        UZK_TRY(msm_build_table(c, s.d_points, s.n, cbits, &s.d_table, &s.pre_W));
        s.pre_c = cbits;
        if (desc->precompute == 1) {
            // automatic: a second table for lockstep batches over the same resident bases.  Window width: 15 bits at n = 2^14
            // (swept), one bit per halving below, never under the 8 bits of the one-workgroup-per-window pipeline
            int lg = 0;
            while ((2u << lg) <= n) ++lg;
            const int wbits = msm_precompute_window_bits(s.n, std::max(8, std::min(kBatchWindowBits, lg + 1)));
            UZK_TRY(msm_build_table(c, s.d_points, s.n, wbits, &cb->wide.d_table, &cb->wide.pre_W));
            cb->wide.pre_c = wbits;
        }
    }
    *out = cb;
    return UZK_OK;
}
}  // namespace

extern "C" {

int uzk_circuit_create(const uzk_circuit_desc* desc, uint64_t* circuit_out) try {
    if (!desc || !circuit_out) { set_error("uzk_circuit_create: null pointer"); return UZK_ERR_PARAMETER; }
    const uint32_t n = desc->n;
    if (n < 16 || n > (1u << 20) || (n & (n - 1))) { set_error("uzk_circuit_create: n must be a power of two in 16 .. 2^20 (n = %u)", n); return UZK_ERR_PARAMETER; }
    if (!desc->lagrange_bases || !desc->blind_bases || !desc->permutation) { set_error("uzk_circuit_create: null pointer"); return UZK_ERR_PARAMETER; }
    if (desc->precompute && desc->precompute != 1 && (desc->precompute < 4 || desc->precompute > 24)) { set_error("uzk_circuit_create: precompute must be 0, 1 (automatic) or a window width 4 .. 24"); return UZK_ERR_PARAMETER; }
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
    {
        API_LOCK;
        UZK_TRY(require_ready());
        Ctx& c = ctx();
        cir->device = c.device;
        // commit bases: the Lagrange SRS followed by the six monomial powers apply_blind_factors touches, and their window tables
        UZK_TRY(make_bases(c, desc, &cir->bases));
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
        UZK_TRY(upload_and_install(c, *cir, 0, cir->n_slots, desc->polys, desc->poly_lens));
    }
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    const uint64_t h = r.next++;
    r.circuits[h] = cir;
    *circuit_out = h;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_create"); }

// a circuit lives on the device of the context that created it; its tables are installed and refreshed from contexts of that device
static int circuit_here(const Circuit& cir, const char* who) {
    UZK_TRY(require_ready());
    if (ctx().device != cir.device) { set_error("%s: the circuit lives on device %d, the calling context on device %d", who, cir.device, ctx().device); return UZK_ERR_PARAMETER; }
    return UZK_OK;
}

int uzk_circuit_update_tables(uint64_t circuit, uint32_t first_slot, uint32_t count, const uint64_t* const* polys, const uint64_t* lens) try {
    API_LOCK;
    auto cir = find_circuit(circuit);
    if (!cir) { set_error("uzk_circuit_update_tables: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    if (!polys || !lens) { set_error("uzk_circuit_update_tables: null pointer"); return UZK_ERR_PARAMETER; }
    if (!slot_range_ok(*cir, first_slot, count)) { set_error("uzk_circuit_update_tables: slots %u .. %u of %u", first_slot, first_slot + count, cir->n_slots); return UZK_ERR_PARAMETER; }
    UZK_TRY(circuit_here(*cir, "uzk_circuit_update_tables"));
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
    UZK_TRY(circuit_here(*cir, "uzk_circuit_refresh_tables"));
    Ctx& c = ctx();
    const uint32_t n = cir->n, m = cir->m;
    std::shared_ptr<DevBlock> blk;
    UZK_TRY(dev_block((size_t)count * ((size_t)n + m) * sizeof(Fp), &blk));
    Fp* d_polys = static_cast<Fp*>(blk->p);
    const Ctx::Srs& srs = count >= 2 && cir->bases->wide.d_table ? cir->bases->wide : cir->bases->plain;
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
        if (srs.device != c.device) { set_error("uzk_preprocess_tables: the SRS lives on device %d, the calling context on device %d", srs.device, c.device); return UZK_ERR_PARAMETER; }
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
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.circuits.find(circuit);
    if (it == r.circuits.end()) { set_error("uzk_circuit_release: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    r.circuits.erase(it);                                          // tables, commit bases and the fixed block go with the last reference: a proof in flight finishes
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_release"); }

int uzk_test_circuit_truncate_t(uint64_t circuit, int on) try {
    auto cir = find_circuit(circuit);
    if (!cir) { set_error("uzk_test_circuit_truncate_t: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    cir->truncate_t = on ? 1 : 0;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_test_circuit_truncate_t"); }

int uzk_circuit_info(uint64_t circuit, uint32_t* n_out, uint32_t* evals_per_proof_out, uint32_t* r_scalars_per_proof_out, int* device_out) try {
    auto cir = find_circuit(circuit);
    if (!cir) { set_error("uzk_circuit_info: unknown circuit %llu", (unsigned long long)circuit); return UZK_ERR_PARAMETER; }
    if (n_out) *n_out = cir->n;
    if (evals_per_proof_out) *evals_per_proof_out = evals_per_proof(*cir);
    if (r_scalars_per_proof_out) *r_scalars_per_proof_out = r_scalars_per_proof(*cir);
    if (device_out) *device_out = cir->device;
    return UZK_OK;
} catch (...) { return uzk::on_exception("uzk_circuit_info"); }

}  // extern "C"

/* ---- provers that hold their own lanes (batch >= 1, not shared): the explicit lockstep API -------------------------------------- */
namespace uzk {

int explicit_prover_create(uint32_t n, uint32_t batch, uint64_t* prover_out) {
    API_LOCK;
    UZK_TRY(require_ready());
    std::shared_ptr<Prover> p;
    UZK_TRY(prover_alloc(ctx(), n, batch, &p));
    Registry& r = reg();
    std::lock_guard<std::mutex> lk(r.mu);
    const uint64_t h = r.next++;
    r.provers[h] = p;
    *prover_out = h;
    return UZK_OK;
}
bool explicit_prover_known(uint64_t prover) { return find_prover(prover) != nullptr; }
int explicit_prover_destroy(uint64_t prover) {
    std::shared_ptr<Prover> p;
    {
        Registry& r = reg();
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.provers.find(prover);
        if (it == r.provers.end()) { set_error("uzk_prover_destroy: unknown prover %llu", (unsigned long long)prover); return UZK_ERR_PARAMETER; }
        p = it->second;
        r.provers.erase(it);
    }
    std::lock_guard<std::mutex> lk(p->mu);                         // a round in flight on another thread finishes first (rounds end synchronised)
    return UZK_OK;
}
int explicit_prover_buffer(uint64_t prover, int which, void** d_out, uint64_t* elems_out) {
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
}

// per-proof arrays [B][...] of the explicit API -> one Lane per proof
int explicit_round1(uint64_t prover, uint64_t circuit, const void* witness, const void* wsel, int inputs_on_device, const uint32_t* pi_index, const uint64_t* pi_value,
                    uint32_t pi_count, const uint32_t* hiding, const uint64_t* blinds, uzk_g1_jac* cm_out) {
    API_LOCK;
    auto pp = find_prover(prover);
    auto cir = find_circuit(circuit);
    if (!pp || !cir) { set_error("uzk_prove_round1: unknown prover or circuit"); return UZK_ERR_PARAMETER; }
    Prover& p = *pp;
    std::lock_guard<std::mutex> plk(p.mu);
    UZK_TRY(round1_check(*cir, p.n, wsel != nullptr, pi_index, pi_count, hiding));
    UZK_TRY(require_ready());
    const uint32_t n = p.n, B = p.B, n_first = wsel ? kWires + kWsel : kWires;
    std::vector<Lane1> L(B);
    for (uint32_t b = 0; b < B; ++b) {
        L[b].witness = static_cast<const char*>(witness) + (size_t)b * kWires * n * sizeof(Fp);
        L[b].wsel = wsel ? static_cast<const char*>(wsel) + (size_t)b * kWsel * n * sizeof(Fp) : nullptr;
        L[b].pi_value = as_fp(pi_value) + (size_t)b * pi_count;
        L[b].blinds = as_fp(blinds) + (size_t)b * n_first * 3;
        L[b].cm_out = reinterpret_cast<Jac*>(cm_out) + (size_t)b * n_first;
    }
    return round1_lanes(ctx(), p, cir, B, L.data(), inputs_on_device, pi_index, pi_count, hiding, nullptr);
}
#define EXPLICIT_ENTER(who, done)                                                        \
    API_LOCK;                                                                            \
    auto pp = find_prover(prover);                                                       \
    if (!pp) { set_error(who ": unknown prover"); return UZK_ERR_PARAMETER; }            \
    Prover& p = *pp;                                                                     \
    std::lock_guard<std::mutex> plk(p.mu);                                               \
    UZK_TRY(need_round(p, done, who));                                                   \
    UZK_TRY(require_ready());                                                            \
    UZK_TRY(need_owner(p, ctx(), who));                                                  \
    const uint32_t B = p.k

int explicit_round2(uint64_t prover, const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinds_z, uzk_g1_jac* cm_z_out) {
    EXPLICIT_ENTER("uzk_prove_round2", 1);
    std::vector<Lane2> L(B);
    for (uint32_t b = 0; b < B; ++b) L[b] = Lane2{as_fp(beta) + b, as_fp(gamma) + b, as_fp(blinds_z) + (size_t)b * 3, reinterpret_cast<Jac*>(cm_z_out) + b};
    return round2_lanes(ctx(), p, L.data(), nullptr);
}
int explicit_round3(uint64_t prover, const uint64_t* alpha, const uint64_t* t_rands, uzk_g1_jac* cm_t_out) {
    EXPLICIT_ENTER("uzk_prove_round3", 2);
    std::vector<Lane3> L(B);
    for (uint32_t b = 0; b < B; ++b) L[b] = Lane3{as_fp(alpha) + b, as_fp(t_rands) + (size_t)b * 5, reinterpret_cast<Jac*>(cm_t_out) + (size_t)b * 5};
    return round3_lanes(ctx(), p, L.data(), nullptr);
}
int explicit_round4(uint64_t prover, const uint64_t* zeta, uint64_t* evals_out, size_t evals_cap) {
    EXPLICIT_ENTER("uzk_prove_round4", 3);
    const uint32_t per = evals_per_proof(*p.circuit);
    if (evals_cap < (size_t)B * per) { set_error("uzk_prove_round4: evals_out holds %zu elements, %u proofs of this circuit give %u each", evals_cap, B, per); return UZK_ERR_PARAMETER; }
    std::vector<Lane4> L(B);
    for (uint32_t b = 0; b < B; ++b) L[b] = Lane4{as_fp(zeta) + b, reinterpret_cast<Fp*>(evals_out) + (size_t)b * per, per};
    return round4_lanes(ctx(), p, L.data(), nullptr);
}
int explicit_round5(uint64_t prover, const uint64_t* r_scalars, size_t r_count, const uint64_t* alpha_zeta, const uint64_t* alpha_zeta_omega, uzk_g1_jac* openings_out) {
    EXPLICIT_ENTER("uzk_prove_round5", 4);
    const uint32_t n_r = r_scalars_per_proof(*p.circuit);
    if (r_count != (size_t)B * n_r) { set_error("uzk_prove_round5: %zu r_poly scalars given, %u proofs of this circuit take %u each", r_count, B, n_r); return UZK_ERR_PARAMETER; }
    std::vector<Lane5> L(B);
    for (uint32_t b = 0; b < B; ++b) L[b] = Lane5{as_fp(r_scalars) + (size_t)b * n_r, n_r, as_fp(alpha_zeta) + b, as_fp(alpha_zeta_omega) + b, reinterpret_cast<Jac*>(openings_out) + (size_t)b * 2};
    return round5_lanes(ctx(), p, L.data(), nullptr);
}

}  // namespace uzk
