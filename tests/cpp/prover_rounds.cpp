// prover_rounds -- the hot-path work of one PlonK proof driven from compiled host code through the C ABI ONLY
// (include/uzkge_gpu.h): a circuit made resident with uzk_circuit_create, a prover workspace from uzk_prover_create, and the
// five Fiat-Shamir rounds of `prover_with_lagrange` (uzkge/src/plonk/prover.rs:88-394) as uzk_prove_round1..5 -- the same
// entry points tools/prover_chain.py (Python) and rust/uzkge-glue/gpu_prover.rs (Rust) drive, so there is ONE implementation
// of the round sequence, inside the library, and three thin callers.  Built with plain g++ -- no hip_runtime.h, no
// -lamdhip64 on the link line -- exactly what a Rust host has.
//
//   setup     uzk_circuit_create: 46 per-circuit polynomials -> coset evaluations over the 6n domain        indexer.rs:316-470
//   round 1   iFFT(n) x9 (5 wires, 3 wire selectors, pi) into 6n-slots, hide, 8 commits with blinds       prover.rs:151-192
//   round 2   z_poly, iFFT(n), hide, commit                                                               prover.rs:199-209
//   round 3   coset FFT(6n) x10, quotient kernel, coset iFFT(6n); split_t_and_commit (t of 5n + 11 coefficients, chunk n + 2):
//             split, fold, FFT(n), 5 commits with blinds                                    helpers.rs:223-678, 1323-1408
//   round 4   15 evaluations at zeta, 4 at zeta * omega: one launch                                       prover.rs:246-273
//   round 5   r(X) = sum of 43 scalar_k p_k; batch_prove of 16 polynomials at zeta and of 4 at zeta * omega:
//             quotient, fold, FFT(n), commit with blinds                                    helpers.rs:681-1090, pcs.rs:107-168
//
// Challenges, blinds and r_poly's scalars are inputs (transcript / rng / O(1) formulas are the caller's).  Inputs and outputs
// are raw little-endian files in a directory written / read by tests/test_gpu_cpp_mirror.py, which holds the outputs to
// tests/golden/vectors_v3.npz.
//
// usage: prover_rounds <dir> [reps] [threads] [batch]
//   reps > 0: also time chains of `reps` proofs (after >= 0.5 s of warm-up: the clock ramp) in five blocks and print the
//   MEDIAN ms per proof -- witness resident in HBM, then uploaded from pinned host memory at the start of every proof;
//   threads > 1: after the checked single chain, `threads` host threads -- each with its own context (uzk_ctx_create: own
//   stream, workspaces and lock) and its own prover, all sharing the one circuit -- prove at the same time: proofs per
//   second of one GPU serving several provers;
//   batch > 1: every such thread proves `batch` witnesses in lockstep (uzk_prover_create(n, batch): commits of 8 x batch
//   vectors, transforms of 10 x batch).
#include <algorithm>
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
template <class T> static std::vector<T> times(const std::vector<T>& v, uint32_t B) {      // the same proof B times over
    std::vector<T> r;
    r.reserve(v.size() * B);
    for (uint32_t b = 0; b < B; ++b) r.insert(r.end(), v.begin(), v.end());
    return r;
}

struct Inputs {
    uint64_t n = 0;
    bool shuffle = true, satisfiable = false;
    uint32_t precompute = 1;                            // uzk_circuit_desc.precompute: 0 none, 1 automatic, 4 .. 24 a window width
    std::vector<Fr> witness, wsel, pi_value, blinds8, blinds_z, t_rands, r_scalars, k, sc;
    std::vector<uint32_t> pi_index, perm;
    std::vector<Fr> table_polys;
    std::vector<uzk_g1_affine> bases;
    uint32_t hiding[8] = {3, 3, 3, 2, 2, 2, 2, 2};      // constraint_system/turbo/mod.rs:366-373, prover.rs:186
    void load() {
        const auto meta = rd<uint64_t>("meta");                   // n, shuffle, precompute
        n = meta[0]; shuffle = meta[1] != 0; precompute = meta.size() > 2 ? (uint32_t)meta[2] : 0; satisfiable = meta.size() > 3 && meta[3];
        const auto evals9 = rd<Fr>("evals9");                     // w0..w4, wsel0..2, pi  (9 n)
        witness.assign(evals9.begin(), evals9.begin() + 5 * n);
        wsel.assign(evals9.begin() + 5 * n, evals9.begin() + 8 * n);
        for (uint64_t i = 0; i < n; ++i) {                        // the public inputs: the non-zero entries of the pi evaluations
            const Fr& v = evals9[8 * n + i];
            if (v.l[0] | v.l[1] | v.l[2] | v.l[3]) { pi_index.push_back((uint32_t)i); pi_value.push_back(v); }
        }
        perm = rd<uint32_t>("perm");
        table_polys = rd<Fr>("table_polys");                      // 46 n, coefficient form
        k = rd<Fr>("k");
        sc = rd<Fr>("scalars");   // beta gamma alpha zeta alpha_open alpha_open2 anemoi_g anemoi_g_inv edwards_a k1_inv zeta_omega
        blinds8 = rd<Fr>("blinds8"); blinds_z = rd<Fr>("blinds_z"); t_rands = rd<Fr>("t_rands"); r_scalars = rd<Fr>("r_scalars");
        r_scalars.resize(shuffle ? 43 : 19);                      // without the shuffle polynomials r has 19 terms: the first 19 scalars
        bases = rd<uzk_g1_affine>("bases");                       // lagrange (n) || pcs[0..3) || pcs[n..n+3)
    }
};

static uint64_t make_circuit(const Inputs& in) {
    uzk_circuit_desc d;
    std::memset(&d, 0, sizeof d);
    d.n = (uint32_t)in.n; d.shuffle = in.shuffle; d.precompute = in.precompute;
    d.lagrange_bases = in.bases.data(); d.blind_bases = in.bases.data() + in.n; d.permutation = in.perm.data();
    std::memcpy(d.k, in.k.data(), 5 * 32);
    std::memcpy(d.anemoi_g, in.sc[6].l, 32); std::memcpy(d.anemoi_g_inv, in.sc[7].l, 32); std::memcpy(d.edwards_a, in.sc[8].l, 32);
    CK(uzk_domain_group_gen(in.n, d.group_gen));
    // the frozen vectors treat all 46 slots as arbitrary polynomials, slot 20 (coset_quotient) included
    for (int s = 0; s < UZK_CIRCUIT_SLOTS; ++s) { d.polys[s] = in.table_polys[(size_t)s * in.n].l; d.poly_lens[s] = in.n; }
    uint64_t h = 0;
    CK(uzk_circuit_create(&d, &h));
    return h;
}

// One prover thread: its own prover (and, for the concurrent runs, its own context), the shared circuit.
static void worker(const Inputs& in, uint64_t circuit, uint32_t B, int reps, bool write_outputs, bool own_context, std::atomic<int>* gate, int gate_n,
                   double* ms_out, double* ms_upload_out, std::vector<uint64_t>* digest_out) {
    uint64_t ctx = 0;
    if (own_context) { CK(uzk_ctx_create(&ctx)); CK(uzk_ctx_set_current(ctx)); }     // everything below is ordered on this context's stream
    const size_t n = in.n, m = 6 * n, cs = n + 8;
    const uint32_t per_ev = in.shuffle ? 19 : 15;
    uint64_t prover = 0;
    CK(uzk_prover_create((uint32_t)n, B, &prover));
    // per-proof inputs, the same proof B times over
    const auto witness = times(in.witness, B), wsel = times(in.wsel, B), pi_value = times(in.pi_value, B), blinds8 = times(in.blinds8, B),
               blinds_z = times(in.blinds_z, B), t_rands = times(in.t_rands, B), r_scalars = times(in.r_scalars, B);
    auto rep1 = [&](int idx) { return times(std::vector<Fr>{in.sc[idx]}, B); };
    const auto beta = rep1(0), gamma = rep1(1), alpha = rep1(2), zeta = rep1(3), alpha_open = rep1(4), alpha_open2 = rep1(5);
    // the witness resident in HBM / in pinned host memory (the upload-inclusive timing)
    Fr *d_wit = nullptr, *h_wit = nullptr;
    { void* p = nullptr; CK(uzk_dev_alloc(B * 8 * n * sizeof(Fr), &p)); d_wit = static_cast<Fr*>(p); }
    { void* p = nullptr; CK(uzk_host_alloc(B * 8 * n * sizeof(Fr), &p)); h_wit = static_cast<Fr*>(p); }
    std::memcpy(h_wit, witness.data(), B * 5 * n * sizeof(Fr));
    std::memcpy(h_wit + B * 5 * n, wsel.data(), B * 3 * n * sizeof(Fr));
    CK(uzk_dev_copy(d_wit, h_wit, B * 8 * n * sizeof(Fr), UZK_COPY_H2D));
    CK(uzk_sync());

    std::vector<uzk_g1_jac> cm_w_wsel(8 * B), cm_z(B), cm_t(5 * B), cm_q(2 * B);
    std::vector<Fr> evals((size_t)per_ev * B), t_blinds, q_blinds;
    int source = 0;                                               // 0: ordinary host memory, 1: device, 2: pinned host memory
    bool want_blinds = false;
    auto tail_blinds = [&](uint32_t count) {                      // the fold's blinds, read back from the tail buffer: [count][6] -> [count][3]
        void* d = nullptr; uint64_t elems = 0;
        CK(uzk_prover_buffer(prover, 7, &d, &elems));
        std::vector<Fr> t((size_t)count * 6), r((size_t)count * 3);
        CK(uzk_dev_copy(t.data(), d, t.size() * sizeof(Fr), UZK_COPY_D2H));
        for (uint32_t i = 0; i < count; ++i) for (int j = 0; j < 3; ++j) r[i * 3 + j] = t[i * 6 + j];
        return r;
    };
    auto chain = [&]() {
        const void *w = witness.data(), *s = wsel.data();
        if (source == 1) { w = d_wit; s = d_wit + B * 5 * n; }
        if (source == 2) { w = h_wit; s = h_wit + B * 5 * n; }
        CK(uzk_prove_round1(prover, circuit, w, s, source == 1, in.pi_index.data(), pi_value[0].l, (uint32_t)in.pi_index.size(), in.hiding,
                            blinds8[0].l, cm_w_wsel.data()));
        CK(uzk_prove_round2(prover, beta[0].l, gamma[0].l, blinds_z[0].l, cm_z.data()));
        CK(uzk_prove_round3(prover, alpha[0].l, t_rands[0].l, cm_t.data()));
        if (want_blinds) t_blinds = tail_blinds(5);
        CK(uzk_prove_round4(prover, zeta[0].l, evals[0].l));
        CK(uzk_prove_round5(prover, r_scalars[0].l, alpha_open[0].l, alpha_open2[0].l, cm_q.data()));
        if (want_blinds) q_blinds = tail_blinds(2);
    };

    want_blinds = write_outputs;
    chain();
    CK(uzk_sync());
    want_blinds = false;
    if (write_outputs) {
        wr("cm_w_wsel", cm_w_wsel.data(), 8 * sizeof(uzk_g1_jac)); wr("cm_z", cm_z.data(), sizeof(uzk_g1_jac));
        wr("cm_t", cm_t.data(), 5 * sizeof(uzk_g1_jac)); wr("cm_q", cm_q.data(), 2 * sizeof(uzk_g1_jac));
        wr("evals", evals.data(), (size_t)per_ev * 32);
        wr("t_blinds", t_blinds.data(), t_blinds.size() * 32);
        wr("q_blinds", q_blinds.data(), q_blinds.size() * 32);
        auto dump = [&](const char* name, int which, size_t count, size_t offset = 0) {
            void* d = nullptr; uint64_t elems = 0;
            CK(uzk_prover_buffer(prover, which, &d, &elems));
            std::vector<Fr> h(count);
            CK(uzk_dev_copy(h.data(), static_cast<Fr*>(d) + offset, count * sizeof(Fr), UZK_COPY_D2H));
            wr(name, h.data(), count * sizeof(Fr));
        };
        dump("coefs", 1, 10 * m); dump("coset_evals", 2, 10 * m); dump("t_quotient", 3, m); dump("t", 4, m);
        dump("z_evals", 0, n, 9 * n); dump("r", 9, n + 3); dump("chunks", 5, 5 * cs); dump("quotients", 8, 2 * cs);
        {   // the circuit's coset tables (46, or 21 without the shuffle feature)
            const int n_slots = in.shuffle ? UZK_CIRCUIT_SLOTS : UZK_CS_QPK;
            std::vector<Fr> h((size_t)n_slots * m);
            for (int sl = 0; sl < n_slots; ++sl) {
                const void* d = nullptr; uint64_t len = 0;
                CK(uzk_circuit_table(circuit, sl, 1, &d, &len));
                CK(uzk_dev_copy(h.data() + (size_t)sl * m, d, m * sizeof(Fr), UZK_COPY_D2H));
            }
            wr("tables", h.data(), h.size() * sizeof(Fr));
        }
    }

    if (reps > 0) {
        for (int pass = 0; pass < 2; ++pass) {
            source = pass == 0 ? 1 : 2;
            // warm up for at least half a second: workspaces, plans -- and the clock ramp, which a handful of 2 ms chains does not cover
            const auto w0 = std::chrono::steady_clock::now();
            do { chain(); } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.5);
            CK(uzk_sync());
            if (gate && pass == 0) { gate->fetch_add(1); while (gate->load() < gate_n) std::this_thread::yield(); }
            std::vector<double> blocks;
            for (int blk = 0; blk < 5; ++blk) {
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r) chain();
                CK(uzk_sync());
                blocks.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps / B);
            }
            std::sort(blocks.begin(), blocks.end());
            const double ms = blocks[blocks.size() / 2];          // median of the five blocks, per proof
            if (pass == 0) *ms_out = ms; else if (ms_upload_out) *ms_upload_out = ms;
            if (gate) break;                                      // the concurrent run times the resident form only
        }
    }
    if (digest_out) {      // what this prover produced last: affine commitments and the evaluations (compared across threads and batch lanes)
        auto put_points = [&](const uzk_g1_jac* j, int count) {
            for (int i = 0; i < count; ++i) { uzk_g1_affine a; CK(uzk_g1_to_affine(&j[i], &a)); digest_out->insert(digest_out->end(), a.x, a.x + 4); digest_out->insert(digest_out->end(), a.y, a.y + 4); }
        };
        for (uint32_t b = 0; b < B; ++b) {                        // every lane of a lockstep batch proved the same witness: one digest each
            put_points(cm_w_wsel.data() + 8 * b, 8); put_points(cm_z.data() + b, 1); put_points(cm_t.data() + 5 * b, 5); put_points(cm_q.data() + 2 * b, 2);
            for (uint32_t e = 0; e < per_ev; ++e) digest_out->insert(digest_out->end(), evals[(size_t)b * per_ev + e].l, evals[(size_t)b * per_ev + e].l + 4);
        }
    }
    CK(uzk_dev_free(d_wit)); CK(uzk_host_free(h_wit));
    CK(uzk_prover_destroy(prover));
    if (own_context) { CK(uzk_ctx_set_current(0)); CK(uzk_ctx_destroy(ctx)); }
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: prover_rounds <dir> [reps] [threads] [batch]\n"); return 2; }
    g_dir = argv[1];
    const int reps = argc > 2 ? std::atoi(argv[2]) : 0;
    const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
    const int batch = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
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
    Inputs in;
    in.load();
    // The chain's circuit is synthetic (random polynomials, tools/prover_chain.py ChainInputs): no witness satisfies it, t fills all
    // 6n coefficients, and round 3 would refuse it as the reference aborts on it.  The timing / parity chain reads t as its first
    // 5n + 11 coefficients (tests/chain_oracle.py does the same); meta[3] != 0 (a satisfiable circuit) leaves the real check on.
    if (!in.satisfiable) CK(uzk_tune("prover_t_cap", 1));
    const uint64_t circuit = make_circuit(in);
    double ms = 0, ms_up = 0;
    std::vector<uint64_t> digest0;
    worker(in, circuit, 1, reps, true, false, nullptr, 0, &ms, &ms_up, &digest0);
    if (reps > 0) std::printf("{\"ms_per_chain\": %.4f, \"ms_per_chain_with_witness_upload\": %.4f, \"reps\": %d, \"blocks\": 5, \"n\": %llu}\n", ms, ms_up, reps, (unsigned long long)in.n);
    if ((threads > 1 || batch > 1) && reps > 0) {
        std::atomic<int> gate{0};
        std::vector<double> per(threads, 0.0);
        std::vector<std::vector<uint64_t>> digests(threads);
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, std::cref(in), circuit, (uint32_t)batch, reps, false, true, &gate, threads, &per[t], nullptr, &digests[t]);
        for (auto& th : pool) th.join();
        double worst = 0;
        bool agree = !digest0.empty();
        for (int t = 0; t < threads; ++t) {
            worst = per[t] > worst ? per[t] : worst;
            agree = agree && digests[t].size() == digest0.size() * batch;
            for (int b = 0; agree && b < batch; ++b) agree = std::equal(digest0.begin(), digest0.end(), digests[t].begin() + (size_t)b * digest0.size());
        }
        // per[t] is ms per PROOF of thread t (a lockstep batch counts `batch` proofs per chain)
        std::printf("{\"threads\": %d, \"batch\": %d, \"ms_per_proof_slowest_thread\": %.4f, \"proofs_per_s\": %.1f, \"single_thread_proofs_per_s\": %.1f, "
                    "\"threads_agree_with_single\": %s}\n", threads, batch, worst, threads * 1e3 / worst, 1e3 / ms, agree ? "true" : "false");
        if (!agree) { std::printf("FAILED: a thread's commitments / evaluations differ from the single-threaded chain\n"); return 1; }
    }
    std::printf("OK\n");
    CK(uzk_circuit_release(circuit));
    return 0;
}
