// prover_rounds -- the hot-path work of PlonK proofs driven from compiled host code through the C ABI ONLY
// (include/uzkge_gpu.h): a circuit made resident with uzk_circuit_create, prover workspaces from uzk_prover_create, and the
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
// usage: prover_rounds <dir> [reps] [threads] [lanes] [mode] [skew]
//   reps > 0: also time chains of `reps` proofs of the files' witness on a prover that owns its lane (after >= 0.5 s of warm-up:
//   the clock ramp) in five blocks and print the MEDIAN ms per proof -- witness resident in HBM, then uploaded from pinned host
//   memory at the start of every proof.
//   threads > 1 or lanes > 1: the throughput section.  EVERY proof of it has its own witness, public inputs, blinds, challenges
//   and r_poly scalars (seeded per thread and lane), the witness is uploaded from pinned host memory at the start of every
//   proof, and every thread's last proofs are compared, commitment by commitment and evaluation by evaluation, with the proof a
//   single-threaded prover of one proof makes of the same inputs.  mode:
//     shared    (default when lanes <= 1) the reference's call pattern: `threads` host threads, each calling round 1..5 on its
//               own prover of ONE proof from the default context -- what rust/uzkge-glue does; the library runs callers that
//               stand at the same round together (uzk_coalesce_config(lanes or 8, ...))
//     lockstep  (default when lanes > 1) each thread holds `lanes` witnesses and a prover of that many proofs
//               (uzk_prover_create(n, lanes)) on its own context
//     private   each thread a prover that owns its lane on its own context: no sharing, the streams overlap
//   UZK_THINK_US=<us> (environment): every thread pauses a random time in [0, 2 x us) between two proofs -- a host that spends
//   time on its own work (witness generation) between proofs; the rate then counts that time too.
//   skew = 1: witness columns drawn from the classes real witnesses consist of (50 % zero, 20 % one,
//   10 % minus one, 10 % < 2^16, 10 % uniform) instead of uniform field elements.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <shared_mutex>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/uzkge_gpu.h"
#include "../../include/uzkge_gpu_test.h"      // uzk_test_circuit_truncate_t: the synthetic circuits of the timing chains

#define CK(x) do { int rc_ = (x); if (rc_ != UZK_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, uzk_last_error()); std::exit(1); } } while (0)

struct Fr { uint64_t l[4]; };

// BN254 Fr: the modulus and the Montgomery forms of 1 and -1 (R = 2^256)
static const Fr kMod = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const Fr kOne = {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}};
static bool geq(const Fr& a, const Fr& b) { for (int i = 3; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; return true; }
static Fr sub_raw(const Fr& a, const Fr& b) { Fr r; unsigned __int128 br = 0; for (int i = 0; i < 4; ++i) { unsigned __int128 t = (unsigned __int128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; } return r; }
static Fr add_mod(const Fr& a, const Fr& b) { Fr r; unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; } return geq(r, kMod) ? sub_raw(r, kMod) : r; }
static Fr mont_small(uint32_t v) {           // v * R mod r by double-and-add over Montgomery(1)
    Fr acc = {{0, 0, 0, 0}};
    for (int bit = 31; bit >= 0; --bit) { acc = add_mod(acc, acc); if ((v >> bit) & 1) acc = add_mod(acc, kOne); }
    return acc;
}
struct Rng {                                  // SplitMix64
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    Fr fr() { Fr r = {{next(), next(), next(), next() >> 4}}; return r; }        // < 2^252 < r: a valid element
};

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

// one proof's inputs
struct Proof {
    std::vector<Fr> witness, wsel, pi_value, blinds8, blinds_z, t_rands, r_scalars;
    Fr beta, gamma, alpha, zeta, alpha_open, alpha_open2;
};
struct Inputs {
    uint64_t n = 0;
    bool shuffle = true, satisfiable = false;
    uint32_t precompute = 1;                            // uzk_circuit_desc.precompute: 0 none, 1 automatic, 4 .. 24 a window width
    Proof file;                                         // the proof of the input files
    std::vector<Fr> k, sc;
    std::vector<uint32_t> pi_index, perm;
    std::vector<Fr> table_polys;
    std::vector<uzk_g1_affine> bases;
    uint32_t hiding[8] = {3, 3, 3, 2, 2, 2, 2, 2};      // constraint_system/turbo/mod.rs:366-373, prover.rs:186
    uint32_t n_r() const { return shuffle ? 43 : 19; }
    uint32_t n_ev() const { return shuffle ? 19 : 15; }
    void load() {
        const auto meta = rd<uint64_t>("meta");                   // n, shuffle, precompute
        n = meta[0]; shuffle = meta[1] != 0; precompute = meta.size() > 2 ? (uint32_t)meta[2] : 0; satisfiable = meta.size() > 3 && meta[3];
        const auto evals9 = rd<Fr>("evals9");                     // w0..w4, wsel0..2, pi  (9 n)
        file.witness.assign(evals9.begin(), evals9.begin() + 5 * n);
        file.wsel.assign(evals9.begin() + 5 * n, evals9.begin() + 8 * n);
        for (uint64_t i = 0; i < n; ++i) {                        // the public inputs: the non-zero entries of the pi evaluations
            const Fr& v = evals9[8 * n + i];
            if (v.l[0] | v.l[1] | v.l[2] | v.l[3]) { pi_index.push_back((uint32_t)i); file.pi_value.push_back(v); }
        }
        perm = rd<uint32_t>("perm");
        table_polys = rd<Fr>("table_polys");                      // 46 n, coefficient form
        k = rd<Fr>("k");
        sc = rd<Fr>("scalars");   // beta gamma alpha zeta alpha_open alpha_open2 anemoi_g anemoi_g_inv edwards_a k1_inv zeta_omega
        file.beta = sc[0]; file.gamma = sc[1]; file.alpha = sc[2]; file.zeta = sc[3]; file.alpha_open = sc[4]; file.alpha_open2 = sc[5];
        file.blinds8 = rd<Fr>("blinds8"); file.blinds_z = rd<Fr>("blinds_z"); file.t_rands = rd<Fr>("t_rands"); file.r_scalars = rd<Fr>("r_scalars");
        file.r_scalars.resize(n_r());                             // without the shuffle polynomials r has 19 terms: the first 19 scalars
        bases = rd<uzk_g1_affine>("bases");                       // lagrange (n) || pcs[0..3) || pcs[n..n+3)
    }
    // a proof of its own: witness, public inputs, blinds, challenges and r_poly scalars from the seed
    Proof make(uint64_t seed, bool skew) const {
        Rng g{seed * 0x2545f4914f6cdd1dull + 12345};
        Proof p;
        const Fr zero = {{0, 0, 0, 0}}, minus_one = sub_raw(kMod, kOne);
        auto column = [&](size_t count, std::vector<Fr>& out) {
            out.resize(count);
            for (size_t i = 0; i < count; ++i) {
                if (!skew) { out[i] = g.fr(); continue; }
                const uint64_t c = g.next() % 10;
                out[i] = c < 5 ? zero : c < 7 ? kOne : c == 7 ? minus_one : c == 8 ? mont_small((uint32_t)(g.next() & 0xffff)) : g.fr();
            }
        };
        column(5 * n, p.witness);
        column(3 * n, p.wsel);
        p.pi_value.resize(pi_index.size());
        for (auto& v : p.pi_value) v = g.fr();
        p.blinds8.resize(24);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 3; ++j) p.blinds8[i * 3 + j] = j < (int)hiding[i] ? g.fr() : zero;
        p.blinds_z.resize(3); for (auto& v : p.blinds_z) v = g.fr();
        p.t_rands.resize(5); for (auto& v : p.t_rands) v = g.fr();
        p.r_scalars.resize(n_r()); for (auto& v : p.r_scalars) v = g.fr();
        p.beta = g.fr(); p.gamma = g.fr(); p.alpha = g.fr(); p.zeta = g.fr(); p.alpha_open = g.fr(); p.alpha_open2 = g.fr();
        return p;
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
    // The chain's circuit is synthetic (random polynomials, tools/prover_chain.py ChainInputs): no witness satisfies it, t fills all
    // 6n coefficients, and round 3 would refuse it as the reference aborts on it.  The timing / parity chain reads t as its first
    // 5n + 11 coefficients (tests/chain_oracle.py does the same); meta[3] != 0 (a satisfiable circuit) leaves the real check on.
    if (!in.satisfiable) CK(uzk_test_circuit_truncate_t(h, 1));
    return h;
}

// what a prover produced: affine commitments and the evaluations of every lane, in lane order
static void digest_of(const Inputs& in, uint32_t B, const std::vector<uzk_g1_jac>& cm1, const std::vector<uzk_g1_jac>& cm_z, const std::vector<uzk_g1_jac>& cm_t,
                      const std::vector<uzk_g1_jac>& cm_q, const std::vector<Fr>& evals, std::vector<uint64_t>* out) {
    out->clear();
    auto put_points = [&](const uzk_g1_jac* j, int count) {
        for (int i = 0; i < count; ++i) { uzk_g1_affine a; CK(uzk_g1_to_affine(&j[i], &a)); out->insert(out->end(), a.x, a.x + 4); out->insert(out->end(), a.y, a.y + 4); }
    };
    const uint32_t per_ev = in.n_ev();
    for (uint32_t b = 0; b < B; ++b) {
        put_points(cm1.data() + 8 * b, 8); put_points(cm_z.data() + b, 1); put_points(cm_t.data() + 5 * b, 5); put_points(cm_q.data() + 2 * b, 2);
        for (uint32_t e = 0; e < per_ev; ++e) out->insert(out->end(), evals[(size_t)b * per_ev + e].l, evals[(size_t)b * per_ev + e].l + 4);
    }
}

// A prover of B proofs and their inputs laid out as the round calls take them ([B][...]); the witness also in pinned host
// memory and, optionally, resident in HBM.
static std::shared_mutex& tables_guard() { static std::shared_mutex m; return m; }

struct Bench {
    const Inputs& in;
    uint32_t B;
    uint64_t prover = 0, circuit = 0;
    Fr *h_wit = nullptr, *d_wit = nullptr;
    std::vector<Fr> pi_value, blinds8, blinds_z, t_rands, r_scalars, beta, gamma, alpha, zeta, alpha_open, alpha_open2;
    std::vector<uzk_g1_jac> cm1, cm_z, cm_t, cm_q;
    std::vector<Fr> evals;
    Bench(const Inputs& in_, uint64_t circuit_, const std::vector<Proof>& proofs, bool private_prover, bool resident) : in(in_), B((uint32_t)proofs.size()), circuit(circuit_) {
        const size_t n = in.n;
        if (private_prover) CK(uzk_prover_create_private((uint32_t)n, B, &prover));
        else CK(uzk_prover_create((uint32_t)n, B, &prover));
        { void* p = nullptr; CK(uzk_host_alloc(B * 8 * n * sizeof(Fr), &p)); h_wit = static_cast<Fr*>(p); }
        for (uint32_t b = 0; b < B; ++b) {
            const Proof& p = proofs[b];
            std::memcpy(h_wit + b * 5 * n, p.witness.data(), 5 * n * sizeof(Fr));
            std::memcpy(h_wit + B * 5 * n + b * 3 * n, p.wsel.data(), 3 * n * sizeof(Fr));
            auto app = [](std::vector<Fr>& d, const std::vector<Fr>& s) { d.insert(d.end(), s.begin(), s.end()); };
            app(pi_value, p.pi_value); app(blinds8, p.blinds8); app(blinds_z, p.blinds_z); app(t_rands, p.t_rands); app(r_scalars, p.r_scalars);
            beta.push_back(p.beta); gamma.push_back(p.gamma); alpha.push_back(p.alpha); zeta.push_back(p.zeta);
            alpha_open.push_back(p.alpha_open); alpha_open2.push_back(p.alpha_open2);
        }
        if (resident) {
            void* p = nullptr;
            CK(uzk_dev_alloc(B * 8 * n * sizeof(Fr), &p));
            d_wit = static_cast<Fr*>(p);
            CK(uzk_dev_copy(d_wit, h_wit, B * 8 * n * sizeof(Fr), UZK_COPY_H2D));
            CK(uzk_sync());
        }
        cm1.resize(8 * B); cm_z.resize(B); cm_t.resize(5 * B); cm_q.resize(2 * B); evals.resize((size_t)in.n_ev() * B);
    }
    ~Bench() {
        if (d_wit) CK(uzk_dev_free(d_wit));
        CK(uzk_host_free(h_wit));
        CK(uzk_prover_destroy(prover));
    }
    // source: 1 = the witness resident in HBM, 2 = uploaded from pinned host memory
    void chain(int source, void (*after_round3)(Bench&) = nullptr, void (*after_round5)(Bench&) = nullptr) {
        const size_t n = in.n;
        const Fr* w = source == 1 ? d_wit : h_wit;
        {
            // the glue's guard (rust/uzkge-glue/gpu_prover.rs `rounds`): round 1 runs under the SHARED side of the lock that protects the
            // circuit's public-key tables, so that every thread can stand in round 1 -- where cohorts form -- at the same time
            std::shared_lock<std::shared_mutex> tables(tables_guard());
            CK(uzk_prove_round1(prover, circuit, w, w + B * 5 * n, source == 1, in.pi_index.data(), pi_value.empty() ? nullptr : pi_value[0].l, (uint32_t)in.pi_index.size(),
                                in.hiding, blinds8[0].l, cm1.data()));
        }
        CK(uzk_prove_round2(prover, beta[0].l, gamma[0].l, blinds_z[0].l, cm_z.data()));
        CK(uzk_prove_round3(prover, alpha[0].l, t_rands[0].l, cm_t.data()));
        if (after_round3) after_round3(*this);
        CK(uzk_prove_round4(prover, zeta[0].l, evals[0].l, evals.size()));
        CK(uzk_prove_round5(prover, r_scalars[0].l, r_scalars.size(), alpha_open[0].l, alpha_open2[0].l, cm_q.data()));
        if (after_round5) after_round5(*this);
    }
    std::vector<uint64_t> digest() { std::vector<uint64_t> d; digest_of(in, B, cm1, cm_z, cm_t, cm_q, evals, &d); return d; }
};

// the fold's blinds, read back from the tail buffer: [count][6] -> [count][3]
static std::vector<Fr> tail_blinds(uint64_t prover, uint32_t count) {
    void* d = nullptr; uint64_t elems = 0;
    CK(uzk_prover_buffer(prover, 7, &d, &elems));
    std::vector<Fr> t((size_t)count * 6), r((size_t)count * 3);
    CK(uzk_dev_copy(t.data(), d, t.size() * sizeof(Fr), UZK_COPY_D2H));
    for (uint32_t i = 0; i < count; ++i) for (int j = 0; j < 3; ++j) r[i * 3 + j] = t[i * 6 + j];
    return r;
}
static std::vector<Fr> g_t_blinds, g_q_blinds;

// The checked chain of the input files on a prover that owns its lane: outputs and intermediates written for the Python test.
static double single_chain(const Inputs& in, uint64_t circuit, int reps, double* ms_upload) {
    Bench bn(in, circuit, {in.file}, /*private*/ true, /*resident*/ true);
    const size_t n = in.n, m = 6 * n, cs = n + 8;
    bn.chain(2, [](Bench& x) { g_t_blinds = tail_blinds(x.prover, 5); }, [](Bench& x) { g_q_blinds = tail_blinds(x.prover, 2); });
    CK(uzk_sync());
    wr("cm_w_wsel", bn.cm1.data(), 8 * sizeof(uzk_g1_jac)); wr("cm_z", bn.cm_z.data(), sizeof(uzk_g1_jac));
    wr("cm_t", bn.cm_t.data(), 5 * sizeof(uzk_g1_jac)); wr("cm_q", bn.cm_q.data(), 2 * sizeof(uzk_g1_jac));
    wr("evals", bn.evals.data(), (size_t)in.n_ev() * 32);
    wr("t_blinds", g_t_blinds.data(), g_t_blinds.size() * 32);
    wr("q_blinds", g_q_blinds.data(), g_q_blinds.size() * 32);
    auto dump = [&](const char* name, int which, size_t count, size_t offset = 0) {
        void* d = nullptr; uint64_t elems = 0;
        CK(uzk_prover_buffer(bn.prover, which, &d, &elems));
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
    double ms_resident = 0;
    for (int pass = 0; pass < 2 && reps > 0; ++pass) {
        const int source = pass == 0 ? 1 : 2;
        // warm up for at least half a second: workspaces, plans -- and the clock ramp, which a handful of 2 ms chains does not cover
        const auto w0 = std::chrono::steady_clock::now();
        do { bn.chain(source); } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.5);
        CK(uzk_sync());
        std::vector<double> blocks;
        for (int blk = 0; blk < 5; ++blk) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) bn.chain(source);
            CK(uzk_sync());
            blocks.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps);
        }
        std::sort(blocks.begin(), blocks.end());
        (pass == 0 ? ms_resident : *ms_upload) = blocks[blocks.size() / 2];
    }
    return ms_resident;
}

enum Mode { kShared, kLockstep, kPrivate };

struct ThreadResult {
    double seconds = 0;
    bool agree = false;
};

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: prover_rounds <dir> [reps] [threads] [lanes] [shared|lockstep|private] [skew]\n"); return 2; }
    g_dir = argv[1];
    const int reps = argc > 2 ? std::atoi(argv[2]) : 0;
    const int threads = argc > 3 ? std::max(1, std::atoi(argv[3])) : 1;
    const int lanes = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
    Mode mode = lanes > 1 ? kLockstep : kShared;
    if (argc > 5) mode = !std::strcmp(argv[5], "shared") ? kShared : !std::strcmp(argv[5], "private") ? kPrivate : kLockstep;
    const bool skew = argc > 6 && std::atoi(argv[6]) != 0;
    setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite*/ 0);   // several streams at work; before the process's first HIP call
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
    const uint64_t circuit = make_circuit(in);
    double ms_up = 0;
    // UZK_ROUNDS_NO_SINGLE=1: skip the single-proof latency phase (counter passes that should see the throughput phase only)
    const bool no_single = std::getenv("UZK_ROUNDS_NO_SINGLE") && std::atoi(std::getenv("UZK_ROUNDS_NO_SINGLE")) != 0;
    const double ms = no_single ? 0.0 : single_chain(in, circuit, reps, &ms_up);
    if (reps > 0) std::printf("{\"ms_per_chain\": %.4f, \"ms_per_chain_with_witness_upload\": %.4f, \"reps\": %d, \"blocks\": 5, \"n\": %llu}\n", ms, ms_up, reps, (unsigned long long)in.n);

    if ((threads > 1 || lanes > 1) && reps > 0) {
        const uint32_t B = mode == kLockstep ? (uint32_t)lanes : 1;
        if (mode == kShared) {
            const char *gw = std::getenv("UZK_GATHER_US"), *sw = std::getenv("UZK_STRAGGLER_US");      // 0 = the library's defaults
            const char* gr = std::getenv("UZK_GROUPS");
            CK(uzk_coalesce_config(lanes > 1 ? (uint32_t)lanes : 8, gw ? (uint32_t)std::atoi(gw) : 0, sw ? (uint32_t)std::atoi(sw) : 0, gr ? (uint32_t)std::atoi(gr) : 0));
        }
        // every thread's proofs and what a single-threaded prover of one proof makes of each of them
        std::vector<std::vector<Proof>> proofs(threads);
        std::vector<std::vector<uint64_t>> want(threads);
        for (int t = 0; t < threads; ++t) {
            for (uint32_t b = 0; b < B; ++b) proofs[t].push_back(in.make(1000 + (uint64_t)t * 64 + b, skew));
            for (uint32_t b = 0; b < B; ++b) {
                Bench one(in, circuit, {proofs[t][b]}, true, false);
                one.chain(2);
                const auto d = one.digest();
                want[t].insert(want[t].end(), d.begin(), d.end());
            }
        }
        std::atomic<int> gate{0}, done{0};
        std::vector<ThreadResult> res(threads);
        std::chrono::steady_clock::time_point t_start, t_end;
        std::atomic<bool> go{false};
        auto worker = [&](int t) {
            uint64_t ctx = 0;
            if (mode != kShared) { CK(uzk_ctx_create(&ctx)); CK(uzk_ctx_set_current(ctx)); }      // shared: the default context, as a host that knows nothing of contexts
            {
                Bench bn(in, circuit, proofs[t], mode != kShared, false);
                const auto w0 = std::chrono::steady_clock::now();
                do { bn.chain(2); } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.5);
                gate.fetch_add(1);
                while (!go.load()) std::this_thread::yield();
                const auto t0 = std::chrono::steady_clock::now();
                // UZK_THINK_US: the host's own work between two proofs (witness generation), uniform in [0, 2 x that) per thread
                static const long think_us = std::getenv("UZK_THINK_US") ? std::atol(std::getenv("UZK_THINK_US")) : 0;
                Rng pause{0x7715ull + (uint64_t)t};
                for (int r = 0; r < reps * 5; ++r) {
                    bn.chain(2);
                    if (think_us > 0) std::this_thread::sleep_for(std::chrono::microseconds((long)(pause.next() % (uint64_t)(2 * think_us))));
                }
                res[t].seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                done.fetch_add(1);
                // keep the others company until all are through (a thread that stops early would change what the rest measure), then
                // three checked proofs while everyone is still proving
                while (done.load() < threads) bn.chain(2);
                bool ok = true;
                for (int r = 0; r < 3; ++r) { bn.chain(2); ok = ok && bn.digest() == want[t]; }
                res[t].agree = ok;
            }
            if (mode != kShared) { CK(uzk_ctx_set_current(0)); CK(uzk_ctx_destroy(ctx)); }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        while (gate.load() < threads) std::this_thread::yield();
        if (mode == kShared) CK(uzk_coalesce_stats(nullptr));          // the statistics below cover the timed and the checked proofs, not the warm-up
        t_start = std::chrono::steady_clock::now();
        go.store(true);
        while (done.load() < threads) std::this_thread::sleep_for(std::chrono::microseconds(200));
        t_end = std::chrono::steady_clock::now();
        for (auto& th : pool) th.join();
        // all threads run the same number of proofs; the job is over when the last one is through
        const double wall = std::chrono::duration<double>(t_end - t_start).count();
        double slowest = 0;
        bool agree = true;
        for (int t = 0; t < threads; ++t) { slowest = std::max(slowest, res[t].seconds); agree = agree && res[t].agree; }
        const double total = (double)threads * B * reps * 5;
        uint64_t cs[16] = {};
        if (mode == kShared) CK(uzk_coalesce_stats(cs));
        std::printf("{\"mode\": \"%s\", \"threads\": %d, \"lanes\": %d, \"proofs_per_s\": %.1f, \"ms_per_proof_slowest_thread\": %.4f, \"single_thread_proofs_per_s\": %.1f, "
                    "\"witness\": \"%s, its own per proof, uploaded from pinned host memory every proof\", \"proofs_timed\": %.0f, \"threads_agree_with_single\": %s, "
                    "\"proofs_per_shared_round\": %.2f, \"widest_shared_round\": %llu, \"moved_out\": %llu, \"host_gap_us_between_shared_rounds\": %.1f, \"gather_us_per_group\": %.1f, \"groups_by_size\": [%llu, %llu, %llu, %llu, %llu, %llu, %llu, %llu]}\n",
                    mode == kShared ? "shared" : mode == kLockstep ? "lockstep" : "private", threads, lanes, total / wall, slowest * 1e3 / (B * reps * 5), 1e3 / ms,
                    skew ? "skewed classes" : "uniform", total, agree ? "true" : "false", cs[0] ? (double)cs[1] / (double)cs[0] : 0.0, (unsigned long long)cs[2],
                    (unsigned long long)cs[3], cs[6] ? (double)cs[5] / (double)cs[6] : 0.0, cs[4] ? (double)cs[7] / (double)cs[4] : 0.0, (unsigned long long)cs[8], (unsigned long long)cs[9], (unsigned long long)cs[10], (unsigned long long)cs[11],
                    (unsigned long long)cs[12], (unsigned long long)cs[13], (unsigned long long)cs[14], (unsigned long long)cs[15]);
        if (!agree) { std::printf("FAILED: a thread's commitments / evaluations differ from the single-threaded proof of the same inputs\n"); return 1; }
    }
    std::printf("OK\n");
    CK(uzk_circuit_release(circuit));
    return 0;
}
