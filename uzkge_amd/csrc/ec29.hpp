// Bucket accumulator on the 29-bit-limb representation (fp29.hpp): the mixed addition of the
// accumulation loop (same XYZZ formulas as ec.hpp's xyzz_madd, EFD madd-2008-s) arranged so that
//   * the SRS points are used as they arrive (8 x 32-bit words, 2^256-form, canonical): no converted
//     copy of the SRS exists;
//   * the accumulator keeps  nx = -X * 2^261,  ny = -Y * 2^261  (normalized limbs, value < 6M) and
//     zz = ZZ * 2^266, zzz = ZZZ * 2^266  (products: normalized, < 1.3M).  With these exponents
//     every product below lands in the right form by itself: mul(p.x 2^256, ZZ 2^266) = U2 2^261, ...
//     (a * b * 2^-261 per product), and keeping X, Y negated turns the two differences
//     P = U2 - X and R = S2 - Y into plain limb-wise additions whose limbs stay < 2^30 -- small
//     enough to be squared without a carry pass.
// Per addition: 8 products + 2 squarings, two carry passes, about 90 limb additions.
// Doubling / cancellation (P = 0) are detected on PP = P^2 (normalized: 0 or M) and handled by the
// canonical code of ec.hpp.
#pragma once
#include "ec.hpp"
#include "fp29.hpp"

namespace uzk {

#if defined(__HIP_DEVICE_COMPILE__)

struct Acc29 {
    L29 nx, ny, zz, zzz;
    bool inf;
};

__device__ __forceinline__ Acc29 acc29_inf() {
    Acc29 a;
    a.nx = Fq29::zero(); a.ny = Fq29::zero(); a.zz = Fq29::zero(); a.zzz = Fq29::zero();
    a.inf = true;
    return a;
}
// canonical XYZZ in 2^256-form -> accumulator forms (ZZ = ZZZ = 1 for an affine point)
__device__ __forceinline__ void acc29_set(Acc29& a, const Fp& x, const Fp& y, const Fp* zz, const Fp* zzz) {
    a.nx = Fq29::to_261(Fq29::from_fp(Fq::neg(x)));
    a.ny = Fq29::to_261(Fq29::from_fp(Fq::neg(y)));
    if (zz == nullptr) {
        a.zz = Fq29::constant(Fq29Cfg::R266);
        a.zzz = a.zz;
    } else {
        a.zz = Fq29::mul(Fq29::from_fp(*zz), Fq29::constant(Fq29Cfg::R271));
        a.zzz = Fq29::mul(Fq29::from_fp(*zzz), Fq29::constant(Fq29Cfg::R271));
    }
    a.inf = false;
}
__device__ __forceinline__ void acc29_madd(Acc29& a, const Affine& p_in, bool negate) {
    using F = Fq29;
    if (affine_is_inf(p_in)) return;
    Affine p = p_in;
    if (negate) p.y = Fq::neg(p.y);
    if (a.inf) { acc29_set(a, p.x, p.y, nullptr, nullptr); return; }
    // (statement order keeps few temporaries alive: at most four besides the accumulator)
    const L29 Pd = F::add(F::mul(F::from_fp(p.x), a.zz), a.nx);       // limbs < 2^30, value < 7M
    const L29 PP = F::sqr(Pd);                                        // normalized, < 1.3M
    {
        const uint32_t t = PP.l[0];
        if ((t == 0 || t == Fq29Cfg::M[0]) && F::is_zero_mod_small(PP)) {   // same x: double or cancel
            const L29 Rd = F::add(F::mul(F::from_fp(p.y), a.zzz), a.ny);
            if (F::all_zero(F::canon(Rd))) {
                const XYZZ d = xyzz_dbl_affine(p);
                acc29_set(a, d.x, d.y, &d.zz, &d.zzz);
            } else {
                a.inf = true;
            }
            return;
        }
    }
    a.zz = F::mul(a.zz, PP);
    const L29 nQ = F::mul(a.nx, PP);                                  // -Q
    const L29 PPP = F::mul(Pd, PP);
    const L29 Rd = F::add(F::mul(F::from_fp(p.y), a.zzz), a.ny);
    a.zzz = F::mul(a.zzz, PPP);
    const L29 nB = F::mul(a.ny, PPP);                                 // -Y * PPP
    const L29 RR = F::sqr(Rd);
    // -X3 = PPP + 2Q - RR = PPP - 2 nQ - RR (+ 4M)
    L29 nx;
#pragma unroll
    for (int i = 0; i < 9; ++i) nx.l[i] = PPP.l[i] + Fq29Cfg::OFF4T3[i] - 2 * nQ.l[i] - RR.l[i];
    nx = F::norm(nx);                                                 // value < 5.1M
    a.nx = nx;
    const L29 D = F::sub_off(nx, nQ, Fq29Cfg::OFF2T1);                // Q - X3: limbs < 1.45 * 2^30, value < 7.1M
    const L29 A = F::mul(Rd, D);
    // -Y3 = Y PPP - R (Q - X3) = -nB - A (+ 4M)
    L29 ny;
#pragma unroll
    for (int i = 0; i < 9; ++i) ny.l[i] = Fq29Cfg::OFF4[i] - nB.l[i] - A.l[i];
    a.ny = F::norm(ny);                                               // value < 4M
}
// -> canonical XYZZ in 2^256-form
__device__ __forceinline__ XYZZ acc29_to_xyzz(const Acc29& a) {
    using F = Fq29;
    if (a.inf) return xyzz_inf();
    XYZZ r;
    r.x = Fq::neg(F::to_fp(F::canon(F::to_256(a.nx))));
    r.y = Fq::neg(F::to_fp(F::canon(F::to_256(a.ny))));
    r.zz = F::to_fp(F::canon(F::mul(a.zz, F::constant(Fq29Cfg::R251))));
    r.zzz = F::to_fp(F::canon(F::mul(a.zzz, F::constant(Fq29Cfg::R251))));
    return r;
}

#elif defined(__HIPCC__)
struct Acc29;
#endif

}  // namespace uzk
