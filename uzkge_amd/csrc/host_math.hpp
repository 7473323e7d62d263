// Host-side helpers built on fp256.hpp / ec.hpp: exponentiation, inversion, domain generators,
// Jacobian folding.  Only O(1)..O(log n) work happens here (plan set-up, the final window
// combination of an MSM, folding <= 8 partial sums); the data-parallel work is in the .hip files.
#pragma once
#include "ec.hpp"
#include "../../include/uzkge_gpu.h"

namespace uzk {

template <class F>
inline Fp f_pow_u256(const Fp& a, const uint32_t e[8]) {
    Fp acc = F::one(), base = a;
    for (int i = 0; i < 256; ++i) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = F::mul(acc, base);
        base = F::sqr(base);
    }
    return acc;
}
template <class F>
inline Fp f_pow_u64(const Fp& a, uint64_t e) {
    Fp acc = F::one(), base = a;
    while (e) {
        if (e & 1) acc = F::mul(acc, base);
        base = F::sqr(base);
        e >>= 1;
    }
    return acc;
}
template <class F, class C>
inline Fp f_inv(const Fp& a) {   // Fermat: a^(M-2)
    uint32_t e[8];
    uint64_t br = 2;
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)C::M[i] - br;
        e[i] = (uint32_t)t; br = (t >> 32) & 1;
    }
    return f_pow_u256<F>(a, e);
}
inline Fp fr_inv(const Fp& a) { return f_inv<Fr, FrCfg>(a); }
inline Fp fq_inv(const Fp& a) { return f_inv<Fq, FqCfg>(a); }
inline Fp fr_from_u64(uint64_t v) {
    Fp t = Fr::zero();
    t.v[0] = (uint32_t)v; t.v[1] = (uint32_t)(v >> 32);
    return Fr::to_mont(t);
}
// 1 if Fr has a size-n evaluation domain: n = 2^k (k <= 28) or 3 * 2^k
// (FpPolynomial::{evaluation_domain, quotient_evaluation_domain}, field_polynomial.rs:554-567).
inline bool domain_exists(uint64_t n) {
    if (n == 0) return false;
    uint64_t m = (n % 3 == 0) ? n / 3 : n;
    return (m & (m - 1)) == 0 && m <= (1ull << 28);
}
// 1 if the library TRANSFORMS over the size-n domain: bounded by the largest sizes the parity suite checks against the oracle
// (UZK_NTT_MAX_LOG2 / UZK_NTT_MAX_LOG2_MIXED, include/uzkge_gpu.h; tests/test_gpu_ntt_large.py).
inline bool domain_supported(uint64_t n) {
    if (n == 0) return false;
    const bool mixed = n % 3 == 0;
    const uint64_t m = mixed ? n / 3 : n;
    return (m & (m - 1)) == 0 && m <= (1ull << (mixed ? UZK_NTT_MAX_LOG2_MIXED : UZK_NTT_MAX_LOG2));
}
// group_gen of the size-n domain: 5^((r-1)/n), Montgomery form.
inline Fp fr_root_of_unity(uint64_t n) {
    // q = (r - 1) / n by long division over 32-bit limbs
    uint32_t e[8], q[8];
    uint64_t br = 1;
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)FrCfg::M[i] - br;
        e[i] = (uint32_t)t; br = (t >> 32) & 1;
    }
    unsigned __int128 rem = 0;
    for (int i = 7; i >= 0; --i) {
        unsigned __int128 cur = (rem << 32) | e[i];
        q[i] = (uint32_t)(cur / n);
        rem = cur % n;
    }
    return f_pow_u256<Fr>(fr_from_u64(5), q);
}

inline Jac jac_inf() {
    Jac r; r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero();
    return r;
}
inline Affine xyzz_to_affine_host(const XYZZ& p) {
    Affine r;
    if (xyzz_is_inf(p)) { r.x = Fq::zero(); r.y = Fq::zero(); return r; }
    Fp zi = fq_inv(p.zzz);                          // 1/ZZZ
    Fp zz_inv = Fq::sqr(Fq::mul(zi, p.zz));         // ZZ^3 = ZZZ^2  =>  1/ZZ = (ZZ/ZZZ)^2
    r.x = Fq::mul(p.x, zz_inv);
    r.y = Fq::mul(p.y, zi);
    return r;
}
inline Affine jac_to_affine_host(const Jac& p) {
    Affine r;
    if (Fq::is_zero(p.z)) { r.x = Fq::zero(); r.y = Fq::zero(); return r; }
    Fp zi = fq_inv(p.z);
    Fp zi2 = Fq::sqr(zi);
    r.x = Fq::mul(p.x, zi2);
    r.y = Fq::mul(p.y, Fq::mul(zi2, zi));
    return r;
}

}  // namespace uzk
