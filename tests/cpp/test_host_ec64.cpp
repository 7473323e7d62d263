// CPU check of uzkge_amd/csrc/host_ec64.hpp (the 4 x 64-bit host arithmetic of the MSM's final Horner step) against the
// portable host code of fp256.hpp / ec.hpp, which the rest of the test-suite pins on the oracle.
// build: g++ -O2 -std=c++17 -I uzkge_amd/csrc tests/cpp/test_host_ec64.cpp -o tests/cpp/test_host_ec64
#include <cstdio>
#include <random>
#include <vector>
#include "host_ec64.hpp"
using namespace uzk;
static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static Fp rand_fq(std::mt19937_64& g) {
    Fp r;
    for (auto& w : r.v) w = (uint32_t)g();
    r.v[7] &= 0x1FFFFFFFu;             // < 2^253 < M
    return r;
}
static bool same_point(const Jac& a, const Jac& b) {
    const bool ia = Fq::is_zero(a.z), ib = Fq::is_zero(b.z);
    if (ia || ib) return ia && ib;
    const Fp za2 = Fq::sqr(a.z), zb2 = Fq::sqr(b.z);
    return Fq::eq(Fq::mul(a.x, zb2), Fq::mul(b.x, za2)) &&
           Fq::eq(Fq::mul(a.y, Fq::mul(zb2, b.z)), Fq::mul(b.y, Fq::mul(za2, a.z)));
}
static XYZZ scalar_mul(const XYZZ& p, uint64_t k) {
    XYZZ acc = xyzz_inf();
    for (int bit = 63; bit >= 0; --bit) {
        acc = xyzz_dbl(acc);
        if ((k >> bit) & 1) xyzz_add(acc, p);
    }
    return acc;
}
static Jac horner_ref(const std::vector<XYZZ>& s, int c) {
    XYZZ total = xyzz_inf();
    for (int w = (int)s.size() - 1; w >= 0; --w) {
        if (w != (int)s.size() - 1) for (int d = 0; d < c; ++d) total = xyzz_dbl(total);
        xyzz_add(total, s[w]);
    }
    return xyzz_to_jac(total);
}

int main() {
    std::mt19937_64 g(20261004);
    for (int t = 0; t < 2000; ++t) {
        const Fp a = rand_fq(g), b = rand_fq(g);
        const h64::F x = h64::from_fp(a), y = h64::from_fp(b);
        EXPECT(Fq::eq(h64::to_fp(h64::mul(x, y)), Fq::mul(a, b)));
        EXPECT(Fq::eq(h64::to_fp(h64::mul_portable(x, y)), Fq::mul(a, b)));
        EXPECT(Fq::eq(h64::to_fp(h64::add(x, y)), Fq::add(a, b)));
        EXPECT(Fq::eq(h64::to_fp(h64::sub(x, y)), Fq::sub(a, b)));
        EXPECT(Fq::eq(h64::to_fp(h64::sub(y, x)), Fq::sub(b, a)));
    }
    {   // edge values: 0, 1, M - 1
        Fp z = Fq::zero(), one = Fq::one(), m1 = Fq::neg(Fq::one());
        for (const Fp& a : {z, one, m1}) for (const Fp& b : {z, one, m1}) {
            EXPECT(Fq::eq(h64::to_fp(h64::mul(h64::from_fp(a), h64::from_fp(b))), Fq::mul(a, b)));
            EXPECT(Fq::eq(h64::to_fp(h64::add(h64::from_fp(a), h64::from_fp(b))), Fq::add(a, b)));
            EXPECT(Fq::eq(h64::to_fp(h64::sub(h64::from_fp(a), h64::from_fp(b))), Fq::sub(a, b)));
        }
    }
#if defined(__x86_64__) && !defined(UZK_HOST_NO_ADX)
    std::printf("mulx/adx product: %s\n", h64::cpu_has_adx() ? "in use" : "CPU lacks BMI2/ADX, portable product in use");
    if (h64::cpu_has_adx()) {           // values near the modulus: the largest intermediate sums
        Fp m1 = Fq::neg(Fq::one()), m2 = Fq::neg(Fq::dbl(Fq::one()));
        for (const Fp& a : {m1, m2}) for (const Fp& b : {m1, m2})
            EXPECT(Fq::eq(h64::to_fp(h64::mul_adx(h64::from_fp(a), h64::from_fp(b))), Fq::mul(a, b)));
        for (int t = 0; t < 20000; ++t) {
            Fp a = rand_fq(g), b = rand_fq(g);
            if (t & 1) a = Fq::neg(a);
            EXPECT(Fq::eq(h64::to_fp(h64::mul_adx(h64::from_fp(a), h64::from_fp(b))), Fq::mul(a, b)));
        }
    }
#endif
    Affine G;
    G.x = Fq::one();
    G.y = Fq::dbl(Fq::one());                                   // (1, 2)
    const XYZZ g1 = xyzz_from_affine(G);
    // doubling and addition against ec.hpp
    for (int t = 0; t < 50; ++t) {
        const XYZZ p = scalar_mul(g1, g() | 1), q = scalar_mul(g1, g() | 1);
        const h64::J jp = h64::j_from_xyzz(h64::x4_from(p)), jq = h64::j_from_xyzz(h64::x4_from(q));
        EXPECT(same_point(h64::j_to(jp), xyzz_to_jac(p)));
        EXPECT(same_point(h64::j_to(h64::j_dbl(jp)), xyzz_to_jac(xyzz_dbl(p))));
        XYZZ s = p;
        xyzz_add(s, q);
        EXPECT(same_point(h64::j_to(h64::j_add(jp, jq)), xyzz_to_jac(s)));
        EXPECT(same_point(h64::j_to(h64::j_add(jp, jp)), xyzz_to_jac(xyzz_dbl(p))));       // P + P through the doubling branch
        XYZZ n = p;
        n.y = Fq::neg(n.y);
        EXPECT(Fq::is_zero(h64::j_to(h64::j_add(jp, h64::j_from_xyzz(h64::x4_from(n)))).z));   // P + (-P) = infinity
        EXPECT(same_point(h64::j_to(h64::j_add(h64::j_inf(), jq)), xyzz_to_jac(q)));
        EXPECT(same_point(h64::j_to(h64::j_add(jq, h64::j_inf())), xyzz_to_jac(q)));
        EXPECT(Fq::is_zero(h64::j_to(h64::j_dbl(h64::j_inf())).z));
    }
    // Horner: every window size the pipelines use, with empty windows, equal neighbours and an all-empty vector
    for (int c : {0, 5, 7, 8, 15, 16, 17}) {
        const int W = c ? (254 + c - 1) / c + 1 : 32;
        std::vector<XYZZ> s(W);
        for (int w = 0; w < W; ++w) s[w] = (g() % 5 == 0) ? xyzz_inf() : scalar_mul(g1, g());
        if (W > 3) s[2] = s[1];
        EXPECT(same_point(h64::horner((uint32_t)W, c, [&](uint32_t w) -> const XYZZ& { return s[w]; }), horner_ref(s, c)));
        for (auto& x : s) x = xyzz_inf();
        EXPECT(Fq::is_zero(h64::horner((uint32_t)W, c, [&](uint32_t w) -> const XYZZ& { return s[w]; }).z));
    }
    // horner_split (class-sum reduction): window w arrives as hi(w), lo(w) with value 2^s hi + lo; the split chain must equal
    // the plain Horner over the combined window sums, for every width the general pipeline picks and a pre-mode single window
    for (int c : {15, 16, 17, 18, 19}) {
        const int s8 = 8;
        const int W = (254 + c - 1) / c + 1;
        std::vector<XYZZ> hi(W), lo(W), comb(W);
        for (int w = 0; w < W; ++w) {
            hi[w] = (g() % 6 == 0) ? xyzz_inf() : scalar_mul(g1, g());
            lo[w] = (g() % 6 == 0) ? xyzz_inf() : scalar_mul(g1, g());
            XYZZ t = hi[w];
            for (int d = 0; d < s8; ++d) t = xyzz_dbl(t);
            xyzz_add(t, lo[w]);
            comb[w] = t;
        }
        if (W > 3) { hi[2] = lo[2]; XYZZ t = hi[2]; for (int d = 0; d < s8; ++d) t = xyzz_dbl(t); xyzz_add(t, lo[2]); comb[2] = t; }
        EXPECT(same_point(h64::horner_split((uint32_t)W, c, s8, [&](uint32_t w) -> const XYZZ& { return hi[w]; },
                                            [&](uint32_t w) -> const XYZZ& { return lo[w]; }), horner_ref(comb, c)));
        // one logical window (window-table mode): 2^s hi + lo
        EXPECT(same_point(h64::horner_split(1u, c, s8, [&](uint32_t) -> const XYZZ& { return hi[0]; },
                                            [&](uint32_t) -> const XYZZ& { return lo[0]; }), xyzz_to_jac(comb[0])));
    }
    std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
