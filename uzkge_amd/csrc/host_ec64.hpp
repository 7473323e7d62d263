// Host-side BN254 G1 arithmetic on 4 x 64-bit limbs: the Horner combination of the window sums at the end of every
// MSM (msm_horner_host, 254 dependent doublings per commitment -- 20 % of a 2^14-point commit's wall time when done
// with the portable 8 x 32-bit host code of fp256.hpp / ec.hpp).  Same Montgomery form, same bytes as Fp
// (8 x u32 LE == 4 x u64 LE), so values move between the two by memcpy.
//
// Doublings run in Jacobian coordinates (EFD dbl-2009-l, a = 0: 2M + 5S against 6M + 3S for XYZZ), window sums arrive
// as XYZZ and are mapped with two products ((X ZZ, Y ZZZ, ZZ) is a Jacobian representative); add-2007-bl adds them.
// Host code only: nothing here is compiled for the device.
#pragma once
#include <cstring>
#if defined(__x86_64__)
#include <cpuid.h>
#endif

#include "ec.hpp"

namespace uzk {
namespace h64 {

typedef unsigned __int128 u128;

struct F {
    uint64_t l[4];
};

constexpr uint64_t limb64(const uint32_t (&w)[8], int i) { return (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32); }
constexpr uint64_t M0 = limb64(FqCfg::M, 0), M1 = limb64(FqCfg::M, 1), M2 = limb64(FqCfg::M, 2), M3 = limb64(FqCfg::M, 3);
// -M^-1 mod 2^64 from the 32-bit constant by one Newton step: x' = x (2 + M x)
constexpr uint64_t INV64 = (uint64_t)FqCfg::INV * (2 + M0 * (uint64_t)FqCfg::INV);
static_assert((uint64_t)(M0 * INV64) == ~(uint64_t)0, "INV64 = -M^-1 mod 2^64");

inline F from_fp(const Fp& a) { F r; std::memcpy(&r, &a, sizeof r); return r; }
inline Fp to_fp(const F& a) { Fp r; std::memcpy(&r, &a, sizeof r); return r; }
inline F zero() { return F{{0, 0, 0, 0}}; }
inline bool is_zero(const F& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

// r = a - M if a >= M else a   (a < 2M < 2^255)
inline F reduce_once(const F& a) {
    u128 t = (u128)a.l[0] - M0;
    const uint64_t d0 = (uint64_t)t;
    t = (u128)a.l[1] - M1 - (uint64_t)((t >> 64) & 1);
    const uint64_t d1 = (uint64_t)t;
    t = (u128)a.l[2] - M2 - (uint64_t)((t >> 64) & 1);
    const uint64_t d2 = (uint64_t)t;
    t = (u128)a.l[3] - M3 - (uint64_t)((t >> 64) & 1);
    const uint64_t d3 = (uint64_t)t;
    const bool borrow = ((t >> 64) & 1) != 0;
    return borrow ? a : F{{d0, d1, d2, d3}};
}
inline F add(const F& a, const F& b) {          // M < 2^254: no carry out of the top limb
    u128 c = (u128)a.l[0] + b.l[0];
    F s;
    s.l[0] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[1] + b.l[1]; s.l[1] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[2] + b.l[2]; s.l[2] = (uint64_t)c; c >>= 64;
    c += (u128)a.l[3] + b.l[3]; s.l[3] = (uint64_t)c;
    return reduce_once(s);
}
inline F sub(const F& a, const F& b) {
    u128 t = (u128)a.l[0] - b.l[0];
    F d;
    d.l[0] = (uint64_t)t;
    t = (u128)a.l[1] - b.l[1] - (uint64_t)((t >> 64) & 1); d.l[1] = (uint64_t)t;
    t = (u128)a.l[2] - b.l[2] - (uint64_t)((t >> 64) & 1); d.l[2] = (uint64_t)t;
    t = (u128)a.l[3] - b.l[3] - (uint64_t)((t >> 64) & 1); d.l[3] = (uint64_t)t;
    if ((t >> 64) & 1) {
        u128 c = (u128)d.l[0] + M0; d.l[0] = (uint64_t)c; c >>= 64;
        c += (u128)d.l[1] + M1; d.l[1] = (uint64_t)c; c >>= 64;
        c += (u128)d.l[2] + M2; d.l[2] = (uint64_t)c; c >>= 64;
        c += (u128)d.l[3] + M3; d.l[3] = (uint64_t)c;
    }
    return d;
}
inline F dbl(const F& a) { return add(a, a); }
// Montgomery product a b 2^-256 mod M (CIOS over 64-bit words), portable form
inline F mul_portable(const F& a, const F& b) {
    constexpr uint64_t m[4] = {M0, M1, M2, M3};
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t q = t[0] * INV64;
        c = (u128)q * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    return reduce_once(F{{t[0], t[1], t[2], t[3]}});
}
#if defined(__x86_64__) && !defined(UZK_HOST_NO_ADX)
// The same product as one mulx / adcx / adox block (generated: host_mul_adx.inc), taken when the CPU has BMI2 and ADX.
#include "host_mul_adx.inc"
__attribute__((target("bmi2,adx"))) inline F mul_adx(const F& a, const F& b) {
    static const uint64_t q[4] = {M0, M1, M2, M3};
    static const uint64_t inv = INV64;
    uint64_t z0, z1, z2, z3, z4, lo, hi;
    asm(UZK_HOST_MUL_ADX_ASM
        : [z0] "=&r"(z0), [z1] "=&r"(z1), [z2] "=&r"(z2), [z3] "=&r"(z3), [z4] "=&r"(z4), [lo] "=&r"(lo), [hi] "=&r"(hi)
        : [a] "r"(a.l), [b] "r"(b.l), [q] "r"(q), [inv] "m"(inv), "m"(a), "m"(b), "m"(q)
        : "rax", "rdx", "cc");
    return reduce_once(F{{UZK_HOST_MUL_ADX_RESULT}});
}
inline bool cpu_has_adx() {
    static const bool v = [] {
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
        return ((b >> 8) & 1) != 0 && ((b >> 19) & 1) != 0;          // CPUID.(EAX=7,ECX=0):EBX bit 8 = BMI2, bit 19 = ADX
    }();
    return v;
}
inline F mul(const F& a, const F& b) { return cpu_has_adx() ? mul_adx(a, b) : mul_portable(a, b); }
#else
inline F mul(const F& a, const F& b) { return mul_portable(a, b); }
#endif
inline F sqr(const F& a) { return mul(a, a); }

struct J {          // Jacobian (X/Z^2, Y/Z^3); infinity <=> Z == 0
    F x, y, z;
};
struct X4 {         // XYZZ (X/ZZ, Y/ZZZ); infinity <=> ZZ == 0
    F x, y, zz, zzz;
};
inline J j_inf() { J r; r.x = from_fp(Fq::one()); r.y = r.x; r.z = zero(); return r; }
inline X4 x4_from(const XYZZ& p) { return X4{from_fp(p.x), from_fp(p.y), from_fp(p.zz), from_fp(p.zzz)}; }
inline J j_from(const Jac& p) { return J{from_fp(p.x), from_fp(p.y), from_fp(p.z)}; }
inline Jac j_to(const J& p) {
    Jac r;
    if (is_zero(p.z)) { r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero(); return r; }
    r.x = to_fp(p.x); r.y = to_fp(p.y); r.z = to_fp(p.z);
    return r;
}
// (X ZZ, Y ZZZ, ZZ) represents the same point: X ZZ / ZZ^2 = X / ZZ, Y ZZZ / ZZ^3 = Y / ZZZ (ZZ^3 = ZZZ^2)
inline J j_from_xyzz(const X4& p) {
    if (is_zero(p.zz)) return j_inf();
    return J{mul(p.x, p.zz), mul(p.y, p.zzz), p.zz};
}
// dbl-2009-l (a = 0).  Infinity stays infinity (Z3 = 2 Y Z = 0); no point of G1 has Y = 0 (odd prime order).
inline J j_dbl(const J& p) {
    const F A = sqr(p.x), B = sqr(p.y), C = sqr(B);
    const F D = dbl(sub(sub(sqr(add(p.x, B)), A), C));
    const F E = add(dbl(A), A), Fv = sqr(E);
    J r;
    r.x = sub(Fv, dbl(D));
    r.y = sub(mul(E, sub(D, r.x)), dbl(dbl(dbl(C))));
    r.z = dbl(mul(p.y, p.z));
    return r;
}
// add-2007-bl with the special cases by branches: either operand at infinity, P == Q (doubling), P == -Q (infinity)
inline J j_add(const J& p, const J& q) {
    if (is_zero(p.z)) return q;
    if (is_zero(q.z)) return p;
    const F Z1Z1 = sqr(p.z), Z2Z2 = sqr(q.z);
    const F U1 = mul(p.x, Z2Z2), U2 = mul(q.x, Z1Z1);
    const F S1 = mul(mul(p.y, q.z), Z2Z2), S2 = mul(mul(q.y, p.z), Z1Z1);
    const F H = sub(U2, U1), Rh = sub(S2, S1);
    if (is_zero(H)) return is_zero(Rh) ? j_dbl(p) : j_inf();
    const F I = sqr(dbl(H)), Jv = mul(H, I), r = dbl(Rh), V = mul(U1, I);
    J o;
    o.x = sub(sub(sqr(r), Jv), dbl(V));
    o.y = sub(mul(r, sub(V, o.x)), dbl(mul(S1, Jv)));
    o.z = mul(sub(sub(sqr(add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return o;
}

// sum_w 2^(c w) S_w by Horner (c doublings per step); c = 0: the plain sum.  `get(w)` returns window sum w.
template <class Get>
inline Jac horner(uint32_t windows, int c, const Get& get) {
    J total = j_inf();
    for (int w = (int)windows - 1; w >= 0; --w) {
        if (w != (int)windows - 1) for (int d = 0; d < c; ++d) total = j_dbl(total);
        total = j_add(total, j_from_xyzz(x4_from(get((uint32_t)w))));
    }
    return j_to(total);
}

// The same sum when every window arrives as two class sums (msm_class_sums_kernel): window w = 2^s * hi(w) + lo(w).
// sum_w 2^(c w) (2^s hi_w + lo_w) by one Horner chain with the doublings split c - s | s: no more doublings than above.
template <class GetHi, class GetLo>
inline Jac horner_split(uint32_t windows, int c, int s, const GetHi& hi, const GetLo& lo) {
    J total = j_inf();
    for (int w = (int)windows - 1; w >= 0; --w) {
        if (w != (int)windows - 1) for (int d = 0; d < c - s; ++d) total = j_dbl(total);
        total = j_add(total, j_from_xyzz(x4_from(hi((uint32_t)w))));
        for (int d = 0; d < s; ++d) total = j_dbl(total);
        total = j_add(total, j_from_xyzz(x4_from(lo((uint32_t)w))));
    }
    return j_to(total);
}

}  // namespace h64
}  // namespace uzk
