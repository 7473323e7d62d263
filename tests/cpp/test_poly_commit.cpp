// C++ restatement of the reference's unit tests for the boundary, against include/uzkge_poly_commit.hpp:
//   test_commit                  uzkge/src/poly_commit/kzg_poly_commitment.rs:526-548
//   test_homomorphic_poly_com..  :483-514 (commit(p)+commit(q) == commit(p+q))
//   test_fft                     uzkge/src/poly_commit/field_polynomial.rs:648-719
// The naive side of each check uses the CPU oracle (test infrastructure).
#include <cstdio>
#include <fstream>
#include <iterator>
#include <memory>
#include <random>
#include "../../include/uzkge_poly_commit.hpp"

extern "C" {
void oracle_msm_naive(const uint64_t*, const uint64_t*, size_t, uint64_t*);
void oracle_g1_to_affine(const uint64_t*, uint64_t*);
void oracle_g1_add(const uint64_t*, const uint64_t*, uint64_t*);
void oracle_fr_add(const uint64_t*, const uint64_t*, uint64_t*);
void oracle_poly_eval(const uint64_t*, uint64_t, const uint64_t*, uint64_t*);
void oracle_root_of_unity(uint64_t, uint64_t*);
void oracle_fr_mul(const uint64_t*, const uint64_t*, uint64_t*);
void oracle_fr_to_mont(const uint64_t*, uint64_t*);
}
using namespace uzkge;
static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static Fr rand_fr(std::mt19937_64& g) { Fr r; for (auto& w : r.l) w = g(); r.l[3] &= 0x0FFFFFFFFFFFFFFFull; return r; }
static Fr fr_from_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; Fr r; oracle_fr_to_mont(c, r.l); return r; }
static bool same_point(const uzk_g1_jac& a, const uzk_g1_jac& b) {
    uint64_t x[8], y[8];
    oracle_g1_to_affine(reinterpret_cast<const uint64_t*>(&a), x);
    oracle_g1_to_affine(reinterpret_cast<const uint64_t*>(&b), y);
    return std::memcmp(x, y, 64) == 0;
}
static bool check_fft(const FpPolynomial& p, uint64_t n, const std::vector<Fr>& fft) {
    Fr root, omega = fr_from_u64(1);
    oracle_root_of_unity(n, root.l);
    for (uint64_t i = 0; i < n; ++i) {
        Fr ev;
        oracle_poly_eval(reinterpret_cast<const uint64_t*>(p.coefs.data()), p.coefs.size(), omega.l, ev.l);
        if (!(ev == fft[i])) return false;
        oracle_fr_mul(omega.l, root.l, omega.l);
    }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: test_poly_commit <srs-padding.bin>\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    check(uzk_init(0));
    auto pcs = KZGCommitmentSchemeBN254::from_unchecked_bytes(blob);
    EXPECT(pcs.max_degree() == 2059);
    std::mt19937_64 g(7);

    // test_commit: commit == naive sum of coef_i * SRS_i
    std::vector<Fr> c = {fr_from_u64(1), fr_from_u64(2), fr_from_u64(3)};
    for (int i = 0; i < 13; ++i) c.push_back(rand_fr(g));
    auto poly = FpPolynomial::from_coefs(c);
    uzk_g1_jac naive;
    oracle_msm_naive(reinterpret_cast<const uint64_t*>(pcs.public_parameter_group_1.data()),
                     reinterpret_cast<const uint64_t*>(poly.coefs.data()), poly.coefs.size(), reinterpret_cast<uint64_t*>(&naive));
    EXPECT(same_point(pcs.commit(poly), naive));

    // homomorphism
    std::vector<Fr> a, b, s;
    for (int i = 0; i < 20; ++i) { a.push_back(rand_fr(g)); b.push_back(rand_fr(g)); Fr t; oracle_fr_add(a[i].l, b[i].l, t.l); s.push_back(t); }
    auto ca = pcs.commit(FpPolynomial::from_coefs(a)), cb = pcs.commit(FpPolynomial::from_coefs(b)), cs = pcs.commit(FpPolynomial::from_coefs(s));
    uzk_g1_jac sum;
    oracle_g1_add(reinterpret_cast<const uint64_t*>(&ca), reinterpret_cast<const uint64_t*>(&cb), reinterpret_cast<uint64_t*>(&sum));
    EXPECT(same_point(cs, sum));

    // DegreeError and the zero polynomial
    bool threw = false;
    try { std::vector<Fr> big(2061, fr_from_u64(1)); (void)pcs.commit(FpPolynomial::from_coefs(big)); }
    catch (const UzkgeException& e) { threw = e.kind == UzkgeError::DegreeError; }
    EXPECT(threw);
    auto zero = FpPolynomial::from_coefs({Fr{}, Fr{}, Fr{}});
    EXPECT(zero.coefs.size() == 1);
    auto cz = pcs.commit(zero);
    EXPECT((cz.z[0] | cz.z[1] | cz.z[2] | cz.z[3]) == 0);

    // test_fft: the literal sequence
    const Fr one = fr_from_u64(1), zr{};
    { auto p = FpPolynomial::from_coefs({one}); EXPECT(check_fft(p, 1, *p.fft(1))); }
    { auto p = FpPolynomial::from_coefs({one, one}); EXPECT(check_fft(p, 2, *p.fft(2))); }
    { auto p = FpPolynomial::from_coefs({one, zr}); EXPECT(check_fft(p, 2, *p.fft(2))); }
    { auto p = FpPolynomial::from_coefs({zr, one}); EXPECT(check_fft(p, 2, *p.fft(2))); }
    {
        auto p = FpPolynomial::from_coefs({zr, one, one});
        auto fft = *p.fft(3);
        EXPECT(check_fft(p, 3, fft));
        EXPECT(FpPolynomial::ifft_with_domain(*FpPolynomial::quotient_evaluation_domain(3), fft) == p);
    }
    for (uint64_t n : {16ull, 32ull, 3ull, 48ull}) {
        std::vector<Fr> v;
        for (uint64_t i = 0; i < n; ++i) v.push_back(rand_fr(g));
        auto p = FpPolynomial::from_coefs(v);
        EXPECT(FpPolynomial::ifft_with_domain(n, p.fft_with_domain(n)) == p);
    }
    // an impossible domain is an FFTError, not a crash
    threw = false;
    try { std::vector<uint64_t> x(20, 0); check(uzk_ntt_fr(x.data(), 5, 0, nullptr)); }
    catch (const UzkgeException& e) { threw = e.kind == UzkgeError::FFTError; }
    EXPECT(threw);

    // Lagrange path == monomial commit, pinned on the reference's parameter files (pcs.rs:137-166,
    // gen_params/mod.rs:151-183): q has non-zero coefficients only where the monomial SRS holds real powers.
    if (argc >= 3) {
        std::ifstream fl(argv[2], std::ios::binary);
        std::vector<uint8_t> lblob((std::istreambuf_iterator<char>(fl)), std::istreambuf_iterator<char>());
        auto lag = KZGCommitmentSchemeBN254::from_unchecked_bytes(lblob);
        const size_t N = lag.max_degree() + 1;
        std::unique_ptr<KZGCommitmentSchemeBN254> mono(load_srs_params(blob, N));
        EXPECT(mono->max_degree() + 1 == N + 3);
        std::vector<Fr> q(N + 3);
        for (size_t i = 0; i < 2051; ++i) q[i] = rand_fr(g);
        for (size_t i = N; i < N + 3; ++i) q[i] = rand_fr(g);
        auto direct = mono->commit(FpPolynomial::from_coefs(q));
        auto folded = commit_folded_lagrange(*mono, lag, q, N + 2);
        EXPECT(same_point(direct, folded));
        // the prover's closure: Lagrange branch iff the sizes match, both branches commit to the same polynomial
        std::vector<Fr> low(q.begin(), q.begin() + 2051);
        auto p_low = FpPolynomial::from_coefs(low);
        auto evals = *p_low.fft(N);
        ProverCommit with(*mono, &lag, N), without(*mono, nullptr, N), wrong(*mono, &lag, N / 2);
        EXPECT(with.lagrange_pcs == &lag && without.lagrange_pcs == nullptr && wrong.lagrange_pcs == nullptr);
        EXPECT(same_point(with(evals, p_low, {}), without(evals, p_low, {})));

        // preprocessing loop (indexer.rs:316-470, params.rs:88-121): batched tables == the single calls, on both branches
        std::vector<std::vector<Fr>> tables;
        for (int t = 0; t < 3; ++t) {
            std::vector<Fr> c(2051);
            for (auto& x : c) x = rand_fr(g);
            if (t == 2) std::fill(c.begin(), c.end(), Fr{});                 // an all-zero table (unused selector)
            tables.push_back(*FpPolynomial::from_coefs(c).fft(N));
        }
        const Fr k1 = fr_from_u64(7);
        const size_t M = 2 * N;
        auto pre_l = preprocess_tables(with, tables, M, k1);
        auto pre_m = preprocess_tables(without, tables, M, k1);
        EXPECT(pre_l.cms.size() == 3 && pre_m.cms.size() == 3);
        for (int t = 0; t < 3; ++t) {
            auto coefs = FpPolynomial::ifft_with_domain(N, tables[t]);
            std::vector<Fr> padded(coefs.coefs);
            padded.resize(N, Fr{});
            EXPECT(pre_l.coefs[t] == padded && pre_m.coefs[t] == padded);
            EXPECT(pre_l.coset_evals[t] == coefs.coset_fft_with_domain(M, k1) && pre_m.coset_evals[t] == pre_l.coset_evals[t]);
            EXPECT(same_point(pre_l.cms[t], with(tables[t], coefs, {})));
            EXPECT(same_point(pre_m.cms[t], pre_l.cms[t]));                     // Lagrange commit == monomial commit (reference SRS files)
        }
    }

    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
