// Bucket accumulator on the 29-bit-limb representation (fp29.hpp): the mixed addition of the
// accumulation loop (same XYZZ formulas as ec.hpp's xyzz_madd, EFD madd-2008-s) arranged so that
//   * the SRS points are used as they arrive (8 x 32-bit words, 2^256-form, canonical): no converted
//     copy of the SRS exists.  The accumulator keeps x = X * 2^261, y = Y * 2^261 and
//     nzz = -ZZ * 2^266, z3 = +-ZZZ * 2^266; with these exponents every product lands in the right
//     form by itself (a * b * 2^-261 per product): mul(p.x 2^256, -ZZ 2^266) = -U2 2^261, ...
//   * -P = X - U2 is formed as a limb-wise ADDITION (ZZ is kept negated: -ZZ' = -ZZ * PP stays
//     negative by itself), so its limbs stay < 2^30 and it can be squared without a carry pass;
//   * Y3 = R (Q - X3) - Y PPP is ONE dual product R*D + Y*(-PPP) with a single reduction; -PPP comes
//     for free from -P, and the sign it leaves on ZZZ' = ZZZ * (-PPP) is tracked in one bit and undone
//     by flipping the sign of the next point's y.
// Per addition: 6 products, 2 squarings, 1 dual product (1467 multiply-adds), two carry passes.
// Doubling / cancellation (P = 0) are detected on PP = P^2 (normalized: 0 or M); such a task is
// redone from its first point by the canonical code of ec.hpp in a second, normally empty, kernel.
#pragma once
#include "ec.hpp"
#include "fp29.hpp"

namespace uzk {

#if defined(__HIP_DEVICE_COMPILE__)

struct Acc29 {
    L29 x, y, nzz, z3;  // x, y: normalized, value < 7M / 2M; nzz = -ZZ, z3 = +-ZZZ: products (normalized, < 1.1M)
    bool inf;
    bool zneg;          // true ZZZ = -z3
};

__device__ __forceinline__ Acc29 acc29_inf() {
    Acc29 a;
    a.x = Fq29::zero(); a.y = Fq29::zero(); a.nzz = Fq29::zero(); a.z3 = Fq29::zero();
    a.inf = true;
    a.zneg = false;
    return a;
}
// affine point (canonical, 2^256-form; ZZ = ZZZ = 1) -> accumulator forms
__device__ __forceinline__ void acc29_set(Acc29& a, const Fp& x, const Fp& y) {
    // x 2^261 = 32 x_wire by re-limbing, then one reduction to < 2 M each: 2 x 65 instructions where two products by 2^266 took 410
    a.x = Fq29::reduce(Fq29::from_fp_x32(x));
    a.y = Fq29::reduce(Fq29::from_fp_x32(y));
    a.nzz = Fq29::constant(Fq29Cfg::NR266);
    a.z3 = Fq29::constant(Fq29Cfg::R266);
    a.inf = false;
    a.zneg = false;
}
// Adds p (negated if `negate`) into a.  Returns false -- leaving `a` unusable -- when the addition
// degenerates (p = +-acc: P = 0): the caller hands the whole task to the canonical code
// (msm_accumulate_exc_kernel).  Keeping that branch out of this loop is what keeps the kernel at
// 110 VGPRs = 4 waves per SIMD (with it inline: 164 and 3).
__device__ __forceinline__ bool acc29_madd(Acc29& a, const Affine& p_in, bool negate) {
    using F = Fq29;
    if (Fq::is_zero(p_in.y)) return true;          // infinity is (0, 0); no point of G1 has y = 0 (odd prime order)
    if (a.inf) {
        acc29_set(a, p_in.x, negate ? Fq::neg(p_in.y) : p_in.y);
        return true;
    }
    const Fp py_eff = (negate != a.zneg) ? Fq::neg(p_in.y) : p_in.y;          // sign of z3 folded into y
    const L29 nPd = F::add(F::mul(F::from_fp(p_in.x), a.nzz), a.x);           // -P = X - U2: limbs < 2^30, value < 8.2M
    const L29 PP = F::sqr(nPd);                                               // normalized, < 1.4M
    {
        const uint32_t t = PP.l[0];
        if ((t == 0 || t == Fq29Cfg::M[0]) && F::is_zero_mod_small(PP)) return false;   // same x: double or cancel
    }
    const L29 S2 = F::mul(F::from_fp(py_eff), a.z3);
    const L29 Rd = F::norm(F::sub_off(S2, a.y, Fq29Cfg::OFF2T1));             // R: normalized, value < 3.1M
    a.nzz = F::mul(a.nzz, PP);
    const L29 Q = F::mul(a.x, PP);
    const L29 nPPP = F::mul(nPd, PP);                                         // -PPP
    a.z3 = F::mul(a.z3, nPPP);
    a.zneg = !a.zneg;
    const L29 RR = F::sqr(Rd);
    // X3 = RR - PPP - 2Q (+ 4M)
    L29 x3;
#pragma unroll
    for (int i = 0; i < 9; ++i) x3.l[i] = RR.l[i] + nPPP.l[i] + Fq29Cfg::OFF4[i] - 2 * Q.l[i];
    x3 = F::norm(x3);                                                         // value < 6.2M
    const L29 D = F::sub_off(Q, x3, Fq29Cfg::OFF8T1);                         // Q - X3: limbs < 1.5 * 2^30, value < 9.1M
    a.y = F::mul2(Rd, D, a.y, nPPP);                                          // Y3 = R D - Y PPP: normalized, < 1.2M
    a.x = x3;
    return true;
}
// -> canonical XYZZ in 2^256-form
__device__ __forceinline__ XYZZ acc29_to_xyzz(const Acc29& a) {
    using F = Fq29;
    if (a.inf) return xyzz_inf();
    XYZZ r;
    // exact divisions by 2^5 (x, y: 2^261-form, values < 7 M / 2 M) and 2^10 (nzz, z3: 2^266-form, < 1.1 M) instead of four products by
    // constants and four canon() (round 6): (v + k M) / 2^S < 2 M, one conditional subtraction each
    r.x = Fq::canon(F::template to_fp_div<5>(a.x));
    r.y = Fq::canon(F::template to_fp_div<5>(a.y));
    r.zz = Fq::neg(Fq::canon(F::template to_fp_div<10>(a.nzz)));
    r.zzz = Fq::canon(F::template to_fp_div<10>(a.z3));
    if (a.zneg) r.zzz = Fq::neg(r.zzz);
    return r;
}

#elif defined(__HIPCC__)
struct Acc29;
#endif

}  // namespace uzk
