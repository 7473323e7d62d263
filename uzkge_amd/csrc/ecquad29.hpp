// The quad group operations of ecquad.hpp on the 29-bit-limb representation (fp29.hpp): a lone wave's product costs
// 427 ns in 9 x 29-bit limbs against 566 ns in the 8 x 32-bit assembly (tools/microbench/mul_latency.hip), and the lazy
// additions / subtractions between the product stages need no carry chains -- about a fifth off every dependent step of the
// latency-bound reductions.
//
// X29: all four coordinates in 2^261-form (value * 2^261 mod M); zz, zzz normalized (products), x, y after ONE parallel
// carry step (norm1: limbs < 2^29 + 8, the top limb holds the rest -- within every consumer's limb contract, and no serial
// carry chain on the critical path); values  x < 5.2 M, y < 5.3 M, zz, zzz < 1.05 M  (checked line by line below).
// Infinity <=> every limb of zz is zero (a valid point has ZZ != 0 mod M, and a product never returns the all-zero limbs
// for a non-zero residue's ... it returns them only for the value 0).
#pragma once
#include "ec.hpp"
#include "fp29.hpp"

namespace uzk {

struct alignas(16) X29 {
    L29 x, y, zz, zzz;
};

#if defined(__HIP_DEVICE_COMPILE__)

__device__ __forceinline__ bool x29_is_inf(const X29& p) { return Fq29::all_zero(p.zz); }
__device__ __forceinline__ X29 x29_inf() {
    X29 r;
    r.x = Fq29::zero(); r.y = Fq29::zero(); r.zz = Fq29::zero(); r.zzz = Fq29::zero();
    return r;
}

template <int S>
__device__ __forceinline__ L29 quad_bcast29(const L29& v) {
    L29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        r.l[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[k], S * 0x55, 0xf, 0xf, false);
        // keep the move a move: when the compiler's DPP combine folded it into the following v_sub(rev)_u32_dpp of a lazy
        // subtraction, lanes 0, 2, 3 of the quad got wrong low limbs of Y3 on gfx950 (found with the KAT ops 11-13)
        asm volatile("" : "+v"(r.l[k]));
    }
    return r;
}
__device__ __forceinline__ L29 quad_sel29(uint32_t q, const L29& a0, const L29& a1, const L29& a2, const L29& a3) {
    L29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t lo = (q & 1) ? a1.l[k] : a0.l[k], hi = (q & 1) ? a3.l[k] : a2.l[k];
        r.l[k] = (q & 2) ? hi : lo;
    }
    return r;
}

// canonical XYZZ (2^256-form) -> X29: lane q converts coordinate q (one product), four broadcasts
__device__ __forceinline__ X29 x29_from_xyzz_quad(const XYZZ& p, uint32_t q) {
    Fp c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t lo = (q & 1) ? p.y.v[k] : p.x.v[k], hi = (q & 1) ? p.zzz.v[k] : p.zz.v[k];
        c.v[k] = (q & 2) ? hi : lo;
    }
    const L29 r = Fq29::to_261(Fq29::from_fp(c));                     // normalized, < 1.01 M; 0 stays all-zero limbs
    X29 o;
    o.x = quad_bcast29<0>(r); o.y = quad_bcast29<1>(r); o.zz = quad_bcast29<2>(r); o.zzz = quad_bcast29<3>(r);
    return o;
}
// X29 -> coordinate q of the canonical XYZZ in lane q (x | y | zz | zzz)
__device__ __forceinline__ Fp x29_coord_to_fp(const X29& p, uint32_t q) {
    const L29 c = quad_sel29(q, p.x, p.y, p.zz, p.zzz);
    return Fq29::to_fp(Fq29::canon(Fq29::to_256(c)));                // value < 5.3 M * M / 169 M -> < 1.04 M before canon
}

// 2 * a (dbl-2008-s-1), three product stages:
//   stage 1: V = U^2 (U = 2Y) | X2 = X^2
//   stage 2: W = U V | S = X V | MM = M^2 (M = 3 X2) | ZZ3 = V ZZ              X3 = MM - 2S
//   stage 3: M (S - X3) | W Y | ZZZ3 = W ZZZ                                     Y3 = M (S - X3) - W Y
__device__ __forceinline__ void x29_dbl_quad(X29& a, uint32_t q) {
    using F = Fq29;
    if (x29_is_inf(a)) return;
    const L29 U = F::add(a.y, a.y);                                            // limbs < 2^30, value < 10.6 M
    const L29 r1 = F::sqr(quad_sel29(q, U, a.x, U, a.x));                      // V | X2: limb products < 2^60; V < 1.67 M, X2 < 1.17 M
    const L29 V = quad_bcast29<0>(r1), X2 = quad_bcast29<1>(r1);
    const L29 M3 = F::norm1(F::add(F::add(X2, X2), X2));                        // normalized, value < 3.6 M
    const L29 r2 = F::mul(quad_sel29(q, U, a.x, M3, V), quad_sel29(q, V, V, M3, a.zz));   // W | S | MM | ZZ3: all < 1.11 M
    const L29 W = quad_bcast29<0>(r2), S = quad_bcast29<1>(r2), MM = quad_bcast29<2>(r2);
    const L29 X3 = F::norm1(F::template sub<4>(MM, F::add(S, S)));              // MM - 2S + 4M: normalized, value < 5.2 M
    const L29 D = F::sub_off(S, X3, Fq29Cfg::OFF8T1);                          // S - X3 + 8M: limbs < 1.5 * 2^30, value < 9.2 M
    const L29 r3 = F::mul(quad_sel29(q, M3, W, W, W), quad_sel29(q, D, a.y, a.zzz, a.zzz));   // T1 | T2 | ZZZ3: all < 1.2 M
    a.x = X3;
    a.y = F::norm1(F::template sub<4>(quad_bcast29<0>(r3), quad_bcast29<1>(r3)));      // T1 - T2 + 4M: normalized, value < 5.3 M
    a.zz = quad_bcast29<3>(r2);
    a.zzz = quad_bcast29<2>(r3);
}

// acc += p (add-2008-s), four product stages; every lane of the quad passes the same operands and leaves with the same acc.
//   stage 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1        P = U2 - U1, R = S2 - S1
//   stage 2: PP = P^2    | RR = R^2    | ZZ1 ZZ2      | ZZZ1 ZZZ2
//   stage 3: PPP = P PP  | Q = U1 PP   | ZZ3 = (ZZ1 ZZ2) PP | -                X3 = RR - PPP - 2Q
//   stage 4: R (Q - X3)  | S1 PPP      | -            | ZZZ3 = (ZZZ1 ZZZ2) PPP   Y3 = R (Q - X3) - S1 PPP
__device__ __forceinline__ void x29_add_quad(X29& acc, const X29& p, uint32_t q) {
    using F = Fq29;
    if (x29_is_inf(p)) return;
    if (x29_is_inf(acc)) { acc = p; return; }
    const L29 r1 = F::mul(quad_sel29(q, acc.x, p.x, acc.y, p.y), quad_sel29(q, p.zz, acc.zz, p.zzz, acc.zzz));   // all < 1.04 M
    const L29 U1 = quad_bcast29<0>(r1), U2 = quad_bcast29<1>(r1), S1 = quad_bcast29<2>(r1), S2 = quad_bcast29<3>(r1);
    const L29 Pd = F::norm1(F::template sub<4>(U2, U1));                        // U2 - U1 + 4M: normalized, value < 5.1 M
    const L29 Rd = F::norm1(F::template sub<4>(S2, S1));
    const L29 r2 = F::mul(quad_sel29(q, Pd, Rd, acc.zz, acc.zzz), quad_sel29(q, Pd, Rd, p.zz, p.zzz));           // all < 1.16 M
    const L29 PP = quad_bcast29<0>(r2), RR = quad_bcast29<1>(r2);
    {
        const uint32_t t = PP.l[0];
        if ((t == 0 || t == Fq29Cfg::M[0]) && F::is_zero_mod_small(PP)) {      // P = 0: same x (uniform over the quad)
            const uint32_t u = RR.l[0];
            if ((u == 0 || u == Fq29Cfg::M[0]) && F::is_zero_mod_small(RR)) x29_dbl_quad(acc, q);
            else acc = x29_inf();
            return;
        }
    }
    const L29 r3 = F::mul(quad_sel29(q, Pd, U1, r2, r2), PP);                  // PPP | Q | ZZ3 | unused: all < 1.04 M
    const L29 PPP = quad_bcast29<0>(r3), Q = quad_bcast29<1>(r3);
    L29 X3;
#pragma unroll
    for (int i = 0; i < 9; ++i) X3.l[i] = RR.l[i] + Fq29Cfg::OFF4T3[i] - PPP.l[i] - 2 * Q.l[i];   // RR - PPP - 2Q + 4M
    X3 = F::norm1(X3);                                                          // normalized, value < 5.2 M
    const L29 D = F::sub_off(Q, X3, Fq29Cfg::OFF8T1);                          // Q - X3 + 8M: limbs < 1.5 * 2^30, value < 9.1 M
    const L29 r4 = F::mul(quad_sel29(q, Rd, S1, r2, r2), quad_sel29(q, D, PPP, PPP, PPP));        // T1 | T2 | unused | ZZZ3: < 1.28 M
    acc.x = X3;
    acc.y = F::norm1(F::template sub<4>(quad_bcast29<0>(r4), quad_bcast29<1>(r4)));                // T1 - T2 + 4M: value < 5.3 M
    acc.zz = quad_bcast29<2>(r3);
    acc.zzz = quad_bcast29<3>(r4);
}

#elif defined(__HIPCC__)
// host pass of a .hip file: the kernels' bodies are parsed, never run
__device__ bool x29_is_inf(const X29& p);
__device__ X29 x29_inf();
__device__ X29 x29_from_xyzz_quad(const XYZZ& p, uint32_t q);
__device__ Fp x29_coord_to_fp(const X29& p, uint32_t q);
__device__ void x29_dbl_quad(X29& a, uint32_t q);
__device__ void x29_add_quad(X29& acc, const X29& p, uint32_t q);
#endif

}  // namespace uzk
