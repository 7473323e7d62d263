// prover_rounds -- the hot-path work of one PlonK proof issued from compiled host code through the C ABI ONLY
// (include/uzkge_gpu.h; device buffers come from uzk_dev_alloc / uzk_dev_copy*): what a GPU-resident
// `prover_with_lagrange` (uzkge/src/plonk/prover.rs:88-394) would issue, in its order, with no interpreter between the
// calls.  Built with plain g++ -- no hip_runtime.h, no -lamdhip64 on the link line -- exactly what a Rust host has
// (rust/uzkge-glue/gpu_prover.rs mirrors this file call for call).
//
//   round 1   iFFT(n) x9 (5 wires, 3 wire selectors, pi), hide_polynomial, 8 commits       prover.rs:151-192
//   round 2   z_poly, iFFT(n), hide, commit                                                 prover.rs:199-209
//   round 3   coset FFT(6n) x10, quotient kernel, coset iFFT(6n); split t: fold, FFT(n), commit     helpers.rs:223-678, 1323-1408
//   round 4   evaluations at zeta and zeta * omega                                         prover.rs:246-273
//   round 5   r(X) = sum scalar_k p_k, two batch_prove openings                             helpers.rs:1030, pcs.rs:107-168
//
// Every commit is MSM(lagrange SRS, evaluations) + blind factors (prover.rs:132-149); the blinds ride in the same MSM:
// bases = lagrange[0..n) || srs[0..3) || srs[n..n+3), scalars = evals || b || -b.  Challenges and blinds are inputs
// (transcript / rng are out of scope).  Inputs and outputs are raw little-endian files in a directory written / read by
// tests/test_gpu_cpp_mirror.py, which compares the outputs with the oracle chain (tests/chain_oracle.py).
//
// usage: prover_rounds <dir> [reps] [threads]
//   reps > 0: also time `reps` chains and print ms per chain;
//   threads > 1: after the checked single chain, `threads` host threads -- each with its own context (uzk_ctx_create:
//   own stream, workspaces and lock) and its own device buffers, all sharing the one registered SRS -- run `reps` chains
//   each at the same time: proofs per second of one GPU serving several provers.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/uzkge_gpu.h"

#define CK(x) do { int rc_ = (x); if (rc_ != UZK_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, uzk_last_error()); std::exit(1); } } while (0)

struct Fr { uint64_t l[4]; };
static const uint64_t R_MOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static Fr fr_sub(const Fr& a, const Fr& b) {            // Montgomery form is linear: plain modular subtraction
    Fr r; unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) { unsigned __int128 t = (unsigned __int128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
    if (br) { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)r.l[i] + R_MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static Fr fr_neg(const Fr& a) { Fr z{}; return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) ? fr_sub(z, a) : z; }

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

// One prover: reads the inputs, makes them resident, runs the chain (once for the outputs, then timed).  `gate`: workers
// of a multi-threaded run meet there after their warm-up so that the timed loops overlap.
static void worker(uint64_t srs, int reps, bool write_outputs, bool own_context, std::atomic<int>* gate, int gate_n, double* ms_out,
                   std::vector<uint64_t>* digest_out) {
    uint64_t ctx = 0;
    if (own_context) { CK(uzk_ctx_create(&ctx)); CK(uzk_ctx_set_current(ctx)); }     // copies below are ordered on this context's stream

    // ---- inputs
    const auto meta = rd<uint64_t>("meta");                       // n, shuffle
    const size_t n = meta[0], m = 6 * n;
    const bool shuffle = meta[1] != 0;
    const auto evals9 = rd<Fr>("evals9");                         // w0..w4, wsel0..2, pi  (9 n)
    const auto perm = rd<uint32_t>("perm");
    const auto tables = rd<Fr>("tables");                         // 46 m
    const auto k = rd<Fr>("k");                                   // 5
    const auto sc = rd<Fr>("scalars");     // beta gamma alpha zeta alpha_open anemoi_g anemoi_g_inv edwards_a k1_inv zeta_omega
    const Fr beta = sc[0], gamma = sc[1], alpha = sc[2], zeta = sc[3], alpha_open = sc[4], anemoi_g = sc[5], anemoi_g_inv = sc[6],
             edwards_a = sc[7], k1_inv = sc[8], zeta_omega = sc[9];
    const auto z_h_inv = rd<Fr>("z_h_inv");                       // 6
    const auto blinds_w = rd<Fr>("blinds_w"), blinds_wsel = rd<Fr>("blinds_wsel"), blinds_z = rd<Fr>("blinds_z");   // 5x2, 3x2, 3
    const auto t_rands = rd<Fr>("t_rands"), r_scalars = rd<Fr>("r_scalars");                                       // 5, 12

    // ---- device residency (the SRS is registered once, by main)
    Fr* d_evals = dmalloc<Fr>(9 * n);   upload(d_evals, evals9);
    uint32_t* d_perm = dmalloc<uint32_t>(5 * n); upload(d_perm, perm);
    Fr* d_tables = dmalloc<Fr>(46 * m); upload(d_tables, tables);
    Fr *d_coefs = dmalloc<Fr>(10 * m), *d_tmp = dmalloc<Fr>(10 * n), *d_coset = dmalloc<Fr>(10 * m), *d_tq = dmalloc<Fr>(m), *d_t = dmalloc<Fr>(m),
       *d_z = dmalloc<Fr>(n), *d_sc = dmalloc<Fr>(8 * (n + 6)), *d_chunks = dmalloc<Fr>(5 * (n + 8)), *d_fold = dmalloc<Fr>(5 * n),
       *d_q = dmalloc<Fr>(2 * (n + 8)), *d_r = dmalloc<Fr>(n + 8), *d_open = dmalloc<Fr>(16 * (n + 8)), *d_group = dmalloc<Fr>(n);
    CK(uzk_dev_memset(d_coefs, 0, 10 * m * sizeof(Fr)));          // coefficient slots: 6n each, zero beyond n + 3
    {   // group[i] = omega^i: forward NTT of X
        std::vector<Fr> x(n);
        std::memset(x.data(), 0, n * sizeof(Fr));
        const uint64_t one[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};   // R mod r
        std::memcpy(x[1].l, one, 32);
        upload(d_tmp, x);
        CK(uzk_ntt_fr_device(d_tmp, d_group, n, 0, nullptr, 1));
    }

    // commit `count` evaluation vectors (device, stride n) with their blinds: one batched MSM over n + 6 bases.  The
    // blind tails go through pinned host memory (one slot per commit of the chain), so nothing waits for the upload.
    Fr* h_tails = nullptr;                                        // pinned (uzk_host_alloc): uploads from it are asynchronous
    { void* p = nullptr; CK(uzk_host_alloc((4 * 8 * 6 + 16) * sizeof(Fr), &p)); h_tails = static_cast<Fr*>(p); }
    Fr* h_fix = h_tails + 4 * 8 * 6;                              // 16 more pinned elements for split_t's head / rand patches
    int commit_no = 0;
    auto commit = [&](const Fr* d_ev, uint32_t count, const std::vector<std::vector<Fr>>& blinds, uzk_g1_jac* out) {
        CK(uzk_dev_copy2d(d_sc, (n + 6) * sizeof(Fr), d_ev, n * sizeof(Fr), n * sizeof(Fr), count, UZK_COPY_D2D));
        Fr* tail = h_tails + (commit_no++ % 4) * 8 * 6;
        std::memset(tail, 0, count * 6 * sizeof(Fr));
        for (uint32_t i = 0; i < count; ++i)
            for (size_t j = 0; j < blinds[i].size(); ++j) { tail[i * 6 + j] = blinds[i][j]; tail[i * 6 + 3 + j] = fr_neg(blinds[i][j]); }
        CK(uzk_dev_copy2d(d_sc + n, (n + 6) * sizeof(Fr), tail, 6 * sizeof(Fr), 6 * sizeof(Fr), count, UZK_COPY_H2D));
        CK(uzk_msm_g1_batch_device(srs, 0, d_sc, n + 6, count, out));      // returns after the window sums have arrived
    };

    uzk_g1_jac cm_w_wsel[8], cm_z[1], cm_t[5], cm_q[2];
    std::vector<Fr> evals_zeta(10), z_eval_zo(1), open_ev_zeta(16), open_ev_zo(1);
    std::vector<std::vector<Fr>> t_blinds(5), q_blinds(2);
    void* tq_ptrs[UZK_TQ_NVEC];

    auto chain = [&]() {
        // ---- round 1
        CK(uzk_dev_memset2d(d_coefs + n, m * sizeof(Fr), 0, 8 * sizeof(Fr), 10));          // the slots the blinds are added into
        CK(uzk_ntt_fr_batch_device(d_evals, d_tmp, n, 9, 1, nullptr, 0));
        CK(uzk_dev_copy2d(d_coefs, m * sizeof(Fr), d_tmp, n * sizeof(Fr), n * sizeof(Fr), 9, UZK_COPY_D2D));
        std::vector<std::vector<Fr>> bl8(8);
        for (int i = 0; i < 5; ++i) { bl8[i] = {blinds_w[2 * i], blinds_w[2 * i + 1]}; CK(uzk_hide_polynomial_device(d_coefs + i * m, m, bl8[i][0].l, 2, n)); }
        for (int i = 0; i < 3; ++i) { bl8[5 + i] = {blinds_wsel[2 * i], blinds_wsel[2 * i + 1]}; CK(uzk_hide_polynomial_device(d_coefs + (5 + i) * m, m, bl8[5 + i][0].l, 2, n)); }
        commit(d_evals, 8, bl8, cm_w_wsel);
        // ---- round 2
        CK(uzk_z_poly_device(d_evals, d_perm, d_group, k[0].l, beta.l, gamma.l, (uint32_t)n, 5, d_z));
        CK(uzk_ntt_fr_device(d_z, d_tmp, n, 1, nullptr, 0));
        CK(uzk_dev_copy(d_coefs + 9 * m, d_tmp, n * sizeof(Fr), UZK_COPY_D2D));
        CK(uzk_hide_polynomial_device(d_coefs + 9 * m, m, blinds_z[0].l, 3, n));
        commit(d_z, 1, {{blinds_z[0], blinds_z[1], blinds_z[2]}}, cm_z);
        // ---- round 3
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
        // split t (taken as 5n + 2 coefficients): chunk i gets + rand_i X^n and - rand_(i-1)  (helpers.rs:1353-1363)
        CK(uzk_dev_memset(d_chunks, 0, 5 * (n + 8) * sizeof(Fr)));
        CK(uzk_dev_copy2d(d_chunks, (n + 8) * sizeof(Fr), d_t, n * sizeof(Fr), n * sizeof(Fr), 5, UZK_COPY_D2D));
        CK(uzk_dev_copy(d_chunks + 4 * (n + 8) + n, d_t + 5 * n, 2 * sizeof(Fr), UZK_COPY_D2D));
        Fr heads[5];
        CK(uzk_dev_copy2d(heads, sizeof(Fr), d_t, n * sizeof(Fr), sizeof(Fr), 5, UZK_COPY_D2H));      // synchronises
        Fr prev{};
        for (int i = 0; i < 5; ++i) {
            h_fix[i] = fr_sub(heads[i], prev);                                // coefs[0] -= rand_(i-1)
            CK(uzk_dev_copy(d_chunks + i * (n + 8), &h_fix[i], sizeof(Fr), UZK_COPY_H2D));
            if (i < 4) { h_fix[8 + i] = t_rands[i]; CK(uzk_dev_copy(d_chunks + i * (n + 8) + n, &h_fix[8 + i], sizeof(Fr), UZK_COPY_H2D)); }   // coefs[n] += rand_i
            prev = t_rands[i];
        }
        for (int i = 0; i < 5; ++i) {
            const size_t len = i < 4 ? n + 1 : n + 2;
            t_blinds[i].assign(len - n, Fr{});
            CK(uzk_fold_blinds_device(d_chunks + i * (n + 8), len, n, d_fold + i * n, t_blinds[i][0].l));
        }
        CK(uzk_ntt_fr_batch_device(d_fold, d_fold, n, 5, 0, nullptr, 0));
        commit(d_fold, 5, t_blinds, cm_t);
        // ---- round 4
        CK(uzk_poly_eval_batch_device(d_coefs, m, 10, zeta.l, evals_zeta[0].l));
        CK(uzk_poly_eval_batch_device(d_coefs + 9 * m, m, 1, zeta_omega.l, z_eval_zo[0].l));
        // ---- round 5
        const void* polys[12];
        uint64_t lens[12];
        polys[0] = d_coefs + 9 * m; lens[0] = n + 3;
        for (int i = 0; i < 5; ++i) { polys[1 + i] = d_chunks + i * (n + 8); lens[1 + i] = n + 2; }
        for (int i = 0; i < 6; ++i) { polys[6 + i] = d_coefs + i * m; lens[6 + i] = n + 3; }
        CK(uzk_poly_lincomb_device(polys, lens, r_scalars[0].l, 12, d_r, n + 3));
        CK(uzk_dev_memset(d_open, 0, 16 * (n + 8) * sizeof(Fr)));
        CK(uzk_dev_copy2d(d_open, (n + 8) * sizeof(Fr), d_coefs, m * sizeof(Fr), (n + 3) * sizeof(Fr), 10, UZK_COPY_D2D));
        CK(uzk_dev_copy(d_open + 10 * (n + 8), d_chunks, 5 * (n + 8) * sizeof(Fr), UZK_COPY_D2D));
        CK(uzk_dev_copy(d_open + 15 * (n + 8), d_r, (n + 3) * sizeof(Fr), UZK_COPY_D2D));
        CK(uzk_open_quotient_device(d_open, n + 8, 16, zeta.l, alpha_open.l, d_q, open_ev_zeta[0].l));
        CK(uzk_open_quotient_device(d_open + 9 * (n + 8), n + 8, 1, zeta_omega.l, alpha_open.l, d_q + (n + 8), open_ev_zo[0].l));
        for (int j = 0; j < 2; ++j) {      // q has degree n + 1: max_power_of_2 = n, two blinds (pcs.rs:137-156)
            q_blinds[j].assign(2, Fr{});
            CK(uzk_fold_blinds_device(d_q + j * (n + 8), n + 2, n, d_fold + j * n, q_blinds[j][0].l));
        }
        CK(uzk_ntt_fr_batch_device(d_fold, d_fold, n, 2, 0, nullptr, 0));
        commit(d_fold, 2, q_blinds, cm_q);
    };

    chain();
    CK(uzk_sync());
    if (write_outputs) {
    // ---- outputs
    wr("cm_w_wsel", cm_w_wsel, sizeof cm_w_wsel); wr("cm_z", cm_z, sizeof cm_z); wr("cm_t", cm_t, sizeof cm_t); wr("cm_q", cm_q, sizeof cm_q);
    wr("evals_zeta", evals_zeta.data(), 10 * 32); wr("z_eval_zeta_omega", z_eval_zo.data(), 32);
    wr("open_evals_zeta", open_ev_zeta.data(), 16 * 32); wr("open_evals_zeta_omega", open_ev_zo.data(), 32);
    { std::vector<Fr> flat; for (auto& v : t_blinds) flat.insert(flat.end(), v.begin(), v.end()); wr("t_blinds", flat.data(), flat.size() * 32); }
    { std::vector<Fr> flat; for (auto& v : q_blinds) flat.insert(flat.end(), v.begin(), v.end()); wr("q_blinds", flat.data(), flat.size() * 32); }
    auto dump = [&](const char* name, const Fr* d, size_t count) {
        std::vector<Fr> h(count);
        CK(uzk_dev_copy(h.data(), d, count * sizeof(Fr), UZK_COPY_D2H));
        wr(name, h.data(), count * sizeof(Fr));
    };
    dump("coefs", d_coefs, 10 * m); dump("coset_evals", d_coset, 10 * m); dump("t_quotient", d_tq, m); dump("t", d_t, m);
    dump("z_evals", d_z, n); dump("r", d_r, n + 3);
    { uint64_t nulls = 0; for (int i = 0; i < UZK_TQ_NVEC; ++i) nulls += tq_ptrs[i] == nullptr; wr("tq_null_slots", &nulls, 8); }
    }

    if (reps > 0) {
        for (int r = 0; r < 3; ++r) chain();        // settle workspaces, plans and clocks before timing
        CK(uzk_sync());
        if (gate) { gate->fetch_add(1); while (gate->load() < gate_n) std::this_thread::yield(); }
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) chain();
        CK(uzk_sync());
        *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    if (digest_out) {      // what this prover produced last: affine commitments and the evaluations (compared across threads)
        auto put_points = [&](const uzk_g1_jac* j, int count) {
            for (int i = 0; i < count; ++i) { uzk_g1_affine a; CK(uzk_g1_to_affine(&j[i], &a)); digest_out->insert(digest_out->end(), a.x, a.x + 4); digest_out->insert(digest_out->end(), a.y, a.y + 4); }
        };
        put_points(cm_w_wsel, 8); put_points(cm_z, 1); put_points(cm_t, 5); put_points(cm_q, 2);
        for (const auto* v : {&evals_zeta, &z_eval_zo, &open_ev_zeta, &open_ev_zo}) for (const Fr& f : *v) digest_out->insert(digest_out->end(), f.l, f.l + 4);
    }
    for (void* p : {(void*)d_evals, (void*)d_perm, (void*)d_tables, (void*)d_coefs, (void*)d_tmp, (void*)d_coset, (void*)d_tq, (void*)d_t, (void*)d_z,
                    (void*)d_sc, (void*)d_chunks, (void*)d_fold, (void*)d_q, (void*)d_r, (void*)d_open, (void*)d_group}) CK(uzk_dev_free(p));
    CK(uzk_host_free(h_tails));
    if (own_context) { CK(uzk_ctx_set_current(0)); CK(uzk_ctx_destroy(ctx)); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: prover_rounds <dir> [reps] [threads]\n"); return 2; }
    g_dir = argv[1];
    const int reps = argc > 2 ? std::atoi(argv[2]) : 0;
    const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
    setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite*/ 0);   // one stream per prover thread; before the process's first HIP call
    CK(uzk_init(0));
    const auto meta = rd<uint64_t>("meta");
    const auto bases = rd<uzk_g1_affine>("bases");                // n + 6 points
    uint64_t srs = 0;
    CK(uzk_srs_register(bases.data(), bases.size(), &srs));
    if (meta.size() > 2 && meta[2]) CK(uzk_srs_precompute(srs, 0));
    double ms = 0;
    std::vector<uint64_t> digest0;
    worker(srs, reps, true, false, nullptr, 0, &ms, &digest0);
    if (reps > 0) std::printf("{\"ms_per_chain\": %.4f, \"reps\": %d, \"n\": %llu}\n", ms, reps, (unsigned long long)meta[0]);
    if (threads > 1 && reps > 0) {
        std::atomic<int> gate{0};
        std::vector<double> per(threads, 0.0);
        std::vector<std::vector<uint64_t>> digests(threads);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, srs, reps, false, true, &gate, threads, &per[t], &digests[t]);
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
