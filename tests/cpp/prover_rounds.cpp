// prover_rounds -- the hot-path work of one PlonK proof issued from compiled host code through the C ABI ONLY
// (include/uzkge_gpu.h; device buffers come from uzk_dev_alloc / uzk_dev_copy*): what a GPU-resident
// `prover_with_lagrange` (uzkge/src/plonk/prover.rs:88-394) issues, in its order and with its call mix, with no interpreter
// between the calls.  Built with plain g++ -- no hip_runtime.h, no -lamdhip64 on the link line -- exactly what a Rust host
// has (rust/uzkge-glue/gpu_prover.rs mirrors this file call for call; tools/prover_chain.py is the same chain in Python).
//
//   setup     46 per-circuit polynomials -> coset evaluations over the 6n domain (the indexer's loop)        indexer.rs:316-470
//   round 1   iFFT(n) x9 (5 wires, 3 wire selectors, pi) into 6n-slots, hide, 8 commits with blinds       prover.rs:151-192
//   round 2   z_poly, iFFT(n), hide, commit                                                               prover.rs:199-209
//   round 3   coset FFT(6n) x10, quotient kernel, coset iFFT(6n); split_t_and_commit (t of 5n + 11 coefficients, chunk n + 2):
//             split, fold, FFT(n), 5 commits with blinds                                    helpers.rs:223-678, 1323-1408
//   round 4   15 evaluations at zeta, 4 at zeta * omega: one launch                                       prover.rs:246-273
//   round 5   r(X) = sum of 43 scalar_k p_k; batch_prove of 16 polynomials at zeta and of 4 at zeta * omega:
//             quotient, fold, FFT(n), commit with blinds                                    helpers.rs:681-1090, pcs.rs:107-168
//
// Every commit is lagrange_pcs.commit(evals) + apply_blind_factors (prover.rs:132-142) as ONE batched MSM with tail scalars:
// bases = lagrange[0..n) || srs[0..3) || srs[n..n+3), tail = b || -b (uzk_msm_g1_batch_tail_device).  Challenges, blinds and
// r_poly's scalars are inputs (transcript / rng / O(1) formulas are out of scope).  Inputs and outputs are raw little-endian
// files in a directory written / read by tests/test_gpu_cpp_mirror.py, which holds the outputs to tests/golden/vectors_v3.npz.
//
// usage: prover_rounds <dir> [reps] [threads]
//   reps > 0: also time `reps` chains and print ms per chain (witness resident; then once more with the 9 n witness elements
//   uploaded from pinned host memory at the start of every chain);
//   threads > 1: after the checked single chain, `threads` host threads -- each with its own context (uzk_ctx_create:
//   own stream, workspaces and lock) and its own device buffers, all sharing the one registered SRS -- run `reps` chains
//   each at the same time: proofs per second of one GPU serving several provers.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/uzkge_gpu.h"

#define CK(x) do { int rc_ = (x); if (rc_ != UZK_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, uzk_last_error()); std::exit(1); } } while (0)

struct Fr { uint64_t l[4]; };
static const uint64_t R_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t FR_ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};   // R mod r

static Fr fr_neg(const Fr& a) {                          // Montgomery form is linear: plain modular negation
    Fr r{};
    if (!(a.l[0] | a.l[1] | a.l[2] | a.l[3])) return r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) { unsigned __int128 t = (unsigned __int128)R_MOD[i] - a.l[i] - (uint64_t)br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
    return r;
}

static std::string g_dir;
template <class T> static std::vector<T> rd(const char* name) {
    std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "missing input %s\n", name); std::exit(2); }
    const size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), bytes);
    return v;
}
static void wr(const char* name, const void* p, size_t bytes) {
    std::ofstream f(g_dir + "/out_" + name + ".bin", std::ios::binary);
    f.write(static_cast<const char*>(p), bytes);
}
template <class T> static T* dmalloc(size_t count) { void* p = nullptr; CK(uzk_dev_alloc(count * sizeof(T), &p)); return static_cast<T*>(p); }
template <class T> static void upload(T* d, const std::vector<T>& h) { CK(uzk_dev_copy(d, h.data(), h.size() * sizeof(T), UZK_COPY_H2D)); }

// slots of the 46 per-circuit polynomials: q (9), s (5), l1, qb, q_prk (4), coset_quotient, q_pk (12), q_g (12), q_ecc
enum { T_Q = 0, T_S = 9, T_L1 = 14, T_QB = 15, T_QPRK = 16, T_CQ = 20, T_QPK = 21, T_QG = 33, T_QECC = 45, N_TABLES = 46 };

typedef std::vector<std::pair<const void*, uint64_t>> PolyList;     // device address and coefficient count
static void split_list(const PolyList& v, std::vector<const void*>& p, std::vector<uint64_t>& l) {
    p.clear(); l.clear();
    for (auto& e : v) { p.push_back(e.first); l.push_back(e.second); }
}

// One prover: reads the inputs, makes them resident, runs the chain (once for the outputs, then timed).  `gate`: workers
// of a multi-threaded run meet there after their warm-up so that the timed loops overlap.
static void worker(uint64_t srs, int reps, bool write_outputs, bool own_context, std::atomic<int>* gate, int gate_n, double* ms_out,
                   double* ms_upload_out, std::vector<uint64_t>* digest_out) {
    uint64_t ctx = 0;
    if (own_context) { CK(uzk_ctx_create(&ctx)); CK(uzk_ctx_set_current(ctx)); }     // everything below is ordered on this context's stream

    // ---- inputs
    const auto meta = rd<uint64_t>("meta");                       // n, shuffle, precompute
    const size_t n = meta[0], m = 6 * n, cs = n + 8;
    const uint64_t t_len = 5 * n + 11;     // deg t = 5n + 10 (ChainInputs.t_len)
    const bool shuffle = meta[1] != 0;
    const auto evals9 = rd<Fr>("evals9");                         // w0..w4, wsel0..2, pi  (9 n)
    const auto perm = rd<uint32_t>("perm");
    const auto table_polys = rd<Fr>("table_polys");               // 46 n, coefficient form
    const auto k = rd<Fr>("k");                                   // 5
    const auto sc = rd<Fr>("scalars");   // beta gamma alpha zeta alpha_open alpha_open2 anemoi_g anemoi_g_inv edwards_a k1_inv zeta_omega
    const Fr beta = sc[0], gamma = sc[1], alpha = sc[2], zeta = sc[3], alpha_open = sc[4], alpha_open2 = sc[5], anemoi_g = sc[6],
             anemoi_g_inv = sc[7], edwards_a = sc[8], k1_inv = sc[9], zeta_omega = sc[10];
    const auto z_h_inv = rd<Fr>("z_h_inv");                       // 6
    // 8 x 3 (w0..4, wsel0..2; the hiding degrees are 3,3,3,2,2 and 2,2,2 -- constraint_system/turbo/mod.rs:366-373, prover.rs:186 --
    // an unused third slot holds a zero blind, which changes nothing), 3 for z
    const auto blinds8 = rd<Fr>("blinds8"), blinds_z = rd<Fr>("blinds_z");
    const auto t_rands = rd<Fr>("t_rands"), r_scalars = rd<Fr>("r_scalars");         // 5, 43

    // ---- device residency (the SRS is registered once, by main)
    Fr* d_evals = dmalloc<Fr>(9 * n);   upload(d_evals, evals9);
    uint32_t* d_perm = dmalloc<uint32_t>(5 * n); upload(d_perm, perm);
    Fr *d_coefs = dmalloc<Fr>(10 * m), *d_coset = dmalloc<Fr>(10 * m), *d_tq = dmalloc<Fr>(m), *d_t = dmalloc<Fr>(m), *d_z = dmalloc<Fr>(n),
       *d_chunks = dmalloc<Fr>(5 * cs), *d_fold = dmalloc<Fr>(5 * n), *d_tail = dmalloc<Fr>(5 * 6), *d_q = dmalloc<Fr>(2 * cs), *d_r = dmalloc<Fr>(cs),
       *d_group = dmalloc<Fr>(n), *d_tpolys = dmalloc<Fr>(N_TABLES * n), *d_tables = dmalloc<Fr>(N_TABLES * m);
    CK(uzk_dev_memset(d_coefs, 0, 10 * m * sizeof(Fr)));          // coefficient slots: 6n each, zero beyond n + 3
    // setup, once per circuit (the indexer's loop): coefficient polynomials -> zero-padded 6n-slots -> one batched coset FFT
    upload(d_tpolys, table_polys);
    CK(uzk_dev_memset(d_tables, 0, N_TABLES * m * sizeof(Fr)));
    CK(uzk_dev_copy2d(d_tables, m * sizeof(Fr), d_tpolys, n * sizeof(Fr), n * sizeof(Fr), N_TABLES, UZK_COPY_D2D));
    CK(uzk_ntt_fr_batch_device(d_tables, d_tables, m, N_TABLES, 0, k[1].l, 1));
    {   // group[i] = omega^i: forward NTT of X
        std::vector<Fr> x(n);
        std::memset(x.data(), 0, n * sizeof(Fr));
        std::memcpy(x[1].l, FR_ONE, 32);
        upload(d_group, x);
        CK(uzk_ntt_fr_device(d_group, d_group, n, 0, nullptr, 1));
    }
    // Pinned result words of the asynchronous trimmed-length checks: [0] t, [1..2] the two opening quotients.
    uint64_t* h_lens = nullptr;
    { void* p = nullptr; CK(uzk_host_alloc(4 * sizeof(uint64_t), &p)); h_lens = static_cast<uint64_t*>(p); }
    Fr* h_evals = nullptr;                                        // pinned copy of the witness for the upload-inclusive timing
    { void* p = nullptr; CK(uzk_host_alloc(9 * n * sizeof(Fr), &p)); h_evals = static_cast<Fr*>(p); std::memcpy(h_evals, evals9.data(), 9 * n * sizeof(Fr)); }

    // tail scalars of a commit: blinds || -blinds, three slots each (apply_blind_factors, kzg_poly_commitment.rs:299-313)
    auto tails = [](const Fr* blinds, uint32_t count, uint32_t hd) {
        std::vector<Fr> t((size_t)count * 6, Fr{});
        for (uint32_t i = 0; i < count; ++i)
            for (uint32_t j = 0; j < hd; ++j) { t[i * 6 + j] = blinds[i * hd + j]; t[i * 6 + 3 + j] = fr_neg(blinds[i * hd + j]); }
        return t;
    };
    const auto tail8 = tails(blinds8.data(), 8, 3), tail_z = tails(blinds_z.data(), 1, 3);

    // the polynomials of rounds 4 and 5 by address and length (they never move)
    auto coef = [&](int slot) { return std::make_pair((const void*)(d_coefs + slot * m), (uint64_t)(n + 3)); };
    auto tpoly = [&](int slot) { return std::make_pair((const void*)(d_tpolys + slot * n), (uint64_t)n); };
    PolyList ev_polys, r_polys, open_zeta, open_zo;
    std::vector<uint32_t> ev_point;
    for (int i = 0; i < 5; ++i) { ev_polys.push_back(coef(i)); ev_point.push_back(0); }
    for (int i = 0; i < 4; ++i) { ev_polys.push_back(tpoly(T_S + i)); ev_point.push_back(0); }
    ev_polys.push_back(tpoly(T_QPRK + 2)); ev_point.push_back(0);
    ev_polys.push_back(tpoly(T_QPRK + 3)); ev_point.push_back(0);
    ev_polys.push_back(coef(9)); ev_point.push_back(1);
    for (int i = 0; i < 3; ++i) { ev_polys.push_back(coef(i)); ev_point.push_back(1); }
    if (shuffle) {
        ev_polys.push_back(tpoly(T_QECC)); ev_point.push_back(0);
        for (int i = 0; i < 3; ++i) { ev_polys.push_back(coef(5 + i)); ev_point.push_back(0); }
    }
    for (int i = 0; i < 9; ++i) r_polys.push_back(tpoly(T_Q + i));
    r_polys.push_back(coef(9)); r_polys.push_back(tpoly(T_S + 4)); r_polys.push_back(tpoly(T_QB));
    r_polys.push_back(tpoly(T_QPRK)); r_polys.push_back(tpoly(T_QPRK + 1));
    if (shuffle) {
        for (int i = 0; i < 12; ++i) r_polys.push_back(tpoly(T_QPK + i));
        for (int i = 0; i < 12; ++i) r_polys.push_back(tpoly(T_QG + i));
    }
    const size_t r_chunks_at = r_polys.size();                    // the five t chunks follow; their lengths come from split_t
    for (int i = 0; i < 5; ++i) r_polys.push_back({(const void*)(d_chunks + i * cs), 0});
    for (int i = 0; i < 5; ++i) open_zeta.push_back(coef(i));
    for (int i = 0; i < 4; ++i) open_zeta.push_back(tpoly(T_S + i));
    open_zeta.push_back(tpoly(T_QPRK + 2)); open_zeta.push_back(tpoly(T_QPRK + 3));
    if (shuffle) { open_zeta.push_back(tpoly(T_QECC)); for (int i = 0; i < 3; ++i) open_zeta.push_back(coef(5 + i)); }
    open_zeta.push_back({(const void*)d_r, (uint64_t)(n + 3)});
    open_zo = {coef(9), coef(0), coef(1), coef(2)};
    std::vector<const void*> ev_p, r_p, oz_p, ozo_p;
    std::vector<uint64_t> ev_l, r_l, oz_l, ozo_l;
    split_list(ev_polys, ev_p, ev_l); split_list(r_polys, r_p, r_l); split_list(open_zeta, oz_p, oz_l); split_list(open_zo, ozo_p, ozo_l);
    const Fr points[2] = {zeta, zeta_omega};

    uzk_g1_jac cm_w_wsel[8], cm_z[1], cm_t[5], cm_q[2];
    std::vector<Fr> evals(ev_p.size()), t_blinds(5 * 3), q_blinds(2 * 3);
    uint64_t chunk_lens[5];
    void* tq_ptrs[UZK_TQ_NVEC];
    bool want_blinds = false, upload_witness = false;
    int redone = 0;                                               // rounds redone because a measured length differed from the expected one

    auto chain = [&]() {
        if (upload_witness) CK(uzk_dev_copy(d_evals, h_evals, 9 * n * sizeof(Fr), UZK_COPY_H2D));      // pinned: asynchronous
        // ---- round 1: iFFT straight into the 6n-slots, hide, commit
        CK(uzk_ntt_fr_batch_strided_device(d_evals, n, d_coefs, m, n, 9, 1, nullptr, 0));
        CK(uzk_hide_polynomial_batch_device(d_coefs, m, n, 8, blinds8[0].l, 3, n));
        CK(uzk_msm_g1_batch_tail_device(srs, 0, d_evals, n, n, 8, tail8.data(), 6, 0, cm_w_wsel));
        // ---- round 2
        CK(uzk_z_poly_device(d_evals, d_perm, d_group, k[0].l, beta.l, gamma.l, (uint32_t)n, 5, d_z));
        CK(uzk_ntt_fr_batch_strided_device(d_z, n, d_coefs + 9 * m, m, n, 1, 1, nullptr, 0));
        CK(uzk_hide_polynomial_batch_device(d_coefs + 9 * m, m, n, 1, blinds_z[0].l, 3, n));
        CK(uzk_msm_g1_batch_tail_device(srs, 0, d_z, n, n, 1, tail_z.data(), 6, 0, cm_z));
        // ---- round 3.  (Measured and not done: a second context as a side lane -- uzk_ctx_wait is the edge -- for the nine coset
        // FFTs that need only round 1's coefficients, under round 1's commit: that commit is eight vectors wide and fills the
        // issue slots itself, the chain gains nothing (2.08 ms either way) and four provers sharing the GPU lose 18 %
        // (857 -> 702 proofs/s); for the opening at zeta * omega beside r(X) and the opening at zeta: 2.097 -> 2.082 ms, not worth
        // a second stream per prover.)
        CK(uzk_ntt_fr_batch_device(d_coefs, d_coset, m, 10, 0, k[1].l, 0));
        uzk_quotient_args qa;
        std::memset(&qa, 0, sizeof qa);
        qa.n = (uint32_t)n; qa.factor = 6;
        for (int i = 0; i < 5; ++i) qa.vec[UZK_TQ_W + i] = d_coset + i * m;
        for (int i = 0; i < 3; ++i) qa.vec[UZK_TQ_WSEL + i] = shuffle ? d_coset + (5 + i) * m : nullptr;
        qa.vec[UZK_TQ_PI] = d_coset + 8 * m; qa.vec[UZK_TQ_Z] = d_coset + 9 * m;
        for (int i = 0; i < 21; ++i) qa.vec[UZK_TQ_Q + i] = d_tables + i * m;
        for (int i = 0; i < 25; ++i) qa.vec[UZK_TQ_QPK + i] = shuffle ? d_tables + (21 + i) * m : nullptr;
        for (int i = 0; i < UZK_TQ_NVEC; ++i) tq_ptrs[i] = const_cast<void*>(qa.vec[i]);
        std::memcpy(qa.alpha, alpha.l, 32); std::memcpy(qa.beta, beta.l, 32); std::memcpy(qa.gamma, gamma.l, 32);
        std::memcpy(qa.k, k.data(), 5 * 32);
        std::memcpy(qa.anemoi_g, anemoi_g.l, 32); std::memcpy(qa.anemoi_g_inv, anemoi_g_inv.l, 32); std::memcpy(qa.edwards_a, edwards_a.l, 32);
        std::memcpy(qa.z_h_inv, z_h_inv.data(), 6 * 32);
        CK(uzk_t_quotient_device(&qa, d_tq, 0));
        CK(uzk_ntt_fr_device(d_tq, d_t, m, 1, k1_inv.l, 0));
        // split_t_and_commit (helpers.rs:1323-1408) with the reference's argument n + 2: every chunk's degree (= coefs.len())
        // has max_power_of_2 = n, so all five fold onto n coefficients
        // from_coefs trims t (field_polynomial.rs:86-90) and its coefs.len() drives the split (helpers.rs:1333).  A well-formed proof
        // has deg t = 5n + 10, so the chain goes on with t_len = 5n + 11 while the device measures the real trimmed length into
        // pinned memory; the commit below synchronises, then the two are compared -- and the split is redone with the measured
        // length in the (never observed) case that they differ.
        CK(uzk_poly_trimmed_len_device(d_t, m, &t_len, 1, h_lens, 0));
        auto split_and_commit = [&](uint64_t len) {
            // split_t_and_commit (helpers.rs:1323-1408) with the reference's argument n + 2: every chunk's degree (= coefs.len())
            // has max_power_of_2 = n, so all five fold onto n coefficients
            CK(uzk_split_t_device(d_t, len, n + 2, 5, t_rands[0].l, d_chunks, cs, chunk_lens));
            CK(uzk_fold_blinds_batch_device(d_chunks, cs, chunk_lens, n, 5, d_fold, n, d_tail, 6, want_blinds ? t_blinds[0].l : nullptr));
            CK(uzk_ntt_fr_batch_device(d_fold, d_fold, n, 5, 0, nullptr, 0));
            CK(uzk_msm_g1_batch_tail_device(srs, 0, d_fold, n, n, 5, d_tail, 6, 1, cm_t));
        };
        split_and_commit(t_len);
        if (h_lens[0] != t_len) { ++redone; split_and_commit(h_lens[0]); }
        // ---- round 4: the evaluations of prover.rs:246-273 in one launch
        CK(uzk_poly_eval_ptrs_device(ev_p.data(), ev_l.data(), ev_point.data(), (uint32_t)ev_p.size(), points[0].l, 2, evals[0].l));
        // ---- round 5
        for (int i = 0; i < 5; ++i) r_l[r_chunks_at + i] = chunk_lens[i];
        CK(uzk_poly_lincomb_device(r_p.data(), r_l.data(), r_scalars[0].l, (uint32_t)r_p.size(), d_r, n + 3));
        CK(uzk_open_quotient_ptrs_device(oz_p.data(), oz_l.data(), (uint32_t)oz_p.size(), zeta.l, alpha_open.l, d_q, cs, nullptr));
        CK(uzk_open_quotient_ptrs_device(ozo_p.data(), ozo_l.data(), (uint32_t)ozo_p.size(), zeta_omega.l, alpha_open2.l, d_q + cs, cs, nullptr));
        // degree = q.degree() (pcs.rs:138) = trimmed length - 1 = n + 1 for polynomials of n + 3 coefficients: max_power_of_2 = n,
        // two blinds (pcs.rs:137-156).  Same pattern as for t: go on with the expected lengths, let the device measure, compare
        // after the commit's synchronisation.
        const uint64_t q_caps[2] = {n + 3, n + 3};
        CK(uzk_poly_trimmed_len_device(d_q, cs, q_caps, 2, h_lens + 1, 0));
        auto fold_and_commit = [&](const uint64_t* q_lens) {
            CK(uzk_fold_blinds_batch_device(d_q, cs, q_lens, n, 2, d_fold, n, d_tail, 6, want_blinds ? q_blinds[0].l : nullptr));
            CK(uzk_ntt_fr_batch_device(d_fold, d_fold, n, 2, 0, nullptr, 0));
            CK(uzk_msm_g1_batch_tail_device(srs, 0, d_fold, n, n, 2, d_tail, 6, 1, cm_q));
        };
        const uint64_t q_expected[2] = {n + 2, n + 2};
        fold_and_commit(q_expected);
        if (h_lens[1] != q_expected[0] || h_lens[2] != q_expected[1]) { ++redone; fold_and_commit(h_lens + 1); }
    };

    want_blinds = write_outputs;
    chain();
    CK(uzk_sync());
    want_blinds = false;
    if (write_outputs) {
        wr("cm_w_wsel", cm_w_wsel, sizeof cm_w_wsel); wr("cm_z", cm_z, sizeof cm_z); wr("cm_t", cm_t, sizeof cm_t); wr("cm_q", cm_q, sizeof cm_q);
        wr("evals", evals.data(), evals.size() * 32);
        wr("t_blinds", t_blinds.data(), t_blinds.size() * 32);
        wr("q_blinds", q_blinds.data(), q_blinds.size() * 32);
        auto dump = [&](const char* name, const Fr* d, size_t count) {
            std::vector<Fr> h(count);
            CK(uzk_dev_copy(h.data(), d, count * sizeof(Fr), UZK_COPY_D2H));
            wr(name, h.data(), count * sizeof(Fr));
        };
        dump("coefs", d_coefs, 10 * m); dump("coset_evals", d_coset, 10 * m); dump("t_quotient", d_tq, m); dump("t", d_t, m);
        dump("z_evals", d_z, n); dump("r", d_r, n + 3); dump("chunks", d_chunks, 5 * cs); dump("quotients", d_q, 2 * cs); dump("tables", d_tables, N_TABLES * m);
        { uint64_t nulls = 0; for (int i = 0; i < UZK_TQ_NVEC; ++i) nulls += tq_ptrs[i] == nullptr; wr("tq_null_slots", &nulls, 8); }
    }

    if (reps > 0) {
        for (int pass = 0; pass < 2; ++pass) {
            upload_witness = pass == 1;
            for (int r = 0; r < 3; ++r) chain();        // settle workspaces, plans and clocks before timing
            CK(uzk_sync());
            if (gate && pass == 0) { gate->fetch_add(1); while (gate->load() < gate_n) std::this_thread::yield(); }
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) chain();
            CK(uzk_sync());
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
            if (pass == 0) *ms_out = ms; else if (ms_upload_out) *ms_upload_out = ms;
            if (gate) break;                            // the multi-threaded run times the resident form only
        }
    }
    if (digest_out) {      // what this prover produced last: affine commitments and the evaluations (compared across threads)
        auto put_points = [&](const uzk_g1_jac* j, int count) {
            for (int i = 0; i < count; ++i) { uzk_g1_affine a; CK(uzk_g1_to_affine(&j[i], &a)); digest_out->insert(digest_out->end(), a.x, a.x + 4); digest_out->insert(digest_out->end(), a.y, a.y + 4); }
        };
        put_points(cm_w_wsel, 8); put_points(cm_z, 1); put_points(cm_t, 5); put_points(cm_q, 2);
        for (const Fr& f : evals) digest_out->insert(digest_out->end(), f.l, f.l + 4);
    }
    for (void* p : {(void*)d_evals, (void*)d_perm, (void*)d_coefs, (void*)d_coset, (void*)d_tq, (void*)d_t, (void*)d_z, (void*)d_chunks, (void*)d_fold,
                    (void*)d_tail, (void*)d_q, (void*)d_r, (void*)d_group, (void*)d_tpolys, (void*)d_tables}) CK(uzk_dev_free(p));
    CK(uzk_host_free(h_evals)); CK(uzk_host_free(h_lens));
    if (redone) std::fprintf(stderr, "note: %d round tail(s) redone with measured polynomial lengths\n", redone);
    if (own_context) { CK(uzk_ctx_set_current(0)); CK(uzk_ctx_destroy(ctx)); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: prover_rounds <dir> [reps] [threads]\n"); return 2; }
    g_dir = argv[1];
    const int reps = argc > 2 ? std::atoi(argv[2]) : 0;
    const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
    setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite*/ 0);   // one stream per prover thread; before the process's first HIP call
    CK(uzk_init(0));
    if (const char* t = std::getenv("UZK_TUNE")) {        // "key=value[,key=value]": experiment switches for A/B runs (contexts inherit them)
        std::string all(t);
        size_t pos = 0;
        while (pos < all.size()) {
            const size_t end = all.find(',', pos) == std::string::npos ? all.size() : all.find(',', pos);
            const std::string kv = all.substr(pos, end - pos);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) CK(uzk_tune(kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1)));
            pos = end + 1;
        }
    }
    const auto meta = rd<uint64_t>("meta");
    const auto bases = rd<uzk_g1_affine>("bases");                // n + 6 points
    uint64_t srs = 0;
    CK(uzk_srs_register(bases.data(), bases.size(), &srs));
    if (meta.size() > 2 && meta[2]) CK(uzk_srs_precompute(srs, 0));
    double ms = 0, ms_up = 0;
    std::vector<uint64_t> digest0;
    worker(srs, reps, true, false, nullptr, 0, &ms, &ms_up, &digest0);
    if (reps > 0) std::printf("{\"ms_per_chain\": %.4f, \"ms_per_chain_with_witness_upload\": %.4f, \"reps\": %d, \"n\": %llu}\n", ms, ms_up, reps, (unsigned long long)meta[0]);
    if (threads > 1 && reps > 0) {
        std::atomic<int> gate{0};
        std::vector<double> per(threads, 0.0);
        std::vector<std::vector<uint64_t>> digests(threads);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, srs, reps, false, true, &gate, threads, &per[t], nullptr, &digests[t]);
        for (auto& th : pool) th.join();
        double worst = 0;
        bool agree = true;
        for (int t = 0; t < threads; ++t) { worst = per[t] > worst ? per[t] : worst; agree = agree && digests[t] == digest0 && !digest0.empty(); }
        std::printf("{\"threads\": %d, \"ms_per_chain_slowest_thread\": %.4f, \"proofs_per_s\": %.1f, \"single_thread_proofs_per_s\": %.1f, "
                    "\"threads_agree_with_single\": %s}\n", threads, worst, threads * 1e3 / worst, 1e3 / ms, agree ? "true" : "false");
        if (!agree) { std::printf("FAILED: a thread's commitments / evaluations differ from the single-threaded chain\n"); return 1; }
    }
    std::printf("OK\n");
    CK(uzk_srs_release(srs));
    return 0;
}
