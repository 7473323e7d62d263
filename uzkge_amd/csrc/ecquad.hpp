// Group addition computed by the four lanes of a quad (used by the latency-bound folds and scans of small MSMs).
#pragma once
#include "ec.hpp"

namespace uzk {

// The folds and scans of a small problem are chains of DEPENDENT full additions on a nearly idle chip, so
// what counts is the latency of one addition, 14 products back to back in one lane.  Here the four lanes
// of a quad hold identical copies of both operands and each computes one product per stage,
//   stage 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1        P = U2 - U1, R = S2 - S1
//   stage 2: PP = P^2    | RR = R^2    | ZZ1 ZZ2      | ZZZ1 ZZZ2
//   stage 3: PPP = P PP  | Q = U1 PP   | ZZ3 = (ZZ1 ZZ2) PP | -                X3 = RR - PPP - 2Q
//   stage 4: R (Q - X3)  | S1 PPP      | -            | ZZZ3 = (ZZZ1 ZZZ2) PPP   Y3 = R (Q - X3) - S1 PPP
// exchanging results with DPP quad_perm moves (one v_mov per word, no LDS): four product latencies
// instead of fourteen.  Same formulas and the same canonical arithmetic as xyzz_add (add-2008-s).
#if defined(__HIP_DEVICE_COMPILE__)
template <int S>
__device__ __forceinline__ Fp quad_bcast(const Fp& v) {
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        r.v[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.v[k], S * 0x55, 0xf, 0xf, false);
        // keep the move a move (same fence as quad_bcast29, ecquad29.hpp): the compiler's DPP combine may fold a broadcast
        // into the v_sub / v_add of the modular subtraction that follows; with bound_ctrl = 0 and these masks the folded
        // form was observed to leave lanes 0, 2, 3 of a quad with wrong low limbs on gfx950 in the 29-bit code.  The 8 x 32
        // form feeds carry chains (v_sub_co / v_subb_co), which today's combine does not fold -- the fence makes that
        // independent of instruction selection; KAT ops 5-7 (tests/test_gpu_field_kat.py) guard it either way.
        asm volatile("" : "+v"(r.v[k]));
    }
    return r;
}
__device__ __forceinline__ Fp quad_sel(uint32_t q, const Fp& a0, const Fp& a1, const Fp& a2, const Fp& a3) {
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t lo = (q & 1) ? a1.v[k] : a0.v[k], hi = (q & 1) ? a3.v[k] : a2.v[k];
        r.v[k] = (q & 2) ? hi : lo;
    }
    return r;
}
// acc += p; every lane of the quad passes the same acc and p and leaves with the same acc.  q = lane & 3.
__device__ __forceinline__ void xyzz_add_quad(XYZZ& acc, const XYZZ& p, uint32_t q) {
    if (xyzz_is_inf(p)) return;
    if (xyzz_is_inf(acc)) { acc = p; return; }
    const Fp r1 = Fq::mul(quad_sel(q, acc.x, p.x, acc.y, p.y), quad_sel(q, p.zz, acc.zz, p.zzz, acc.zzz));
    const Fp U1 = quad_bcast<0>(r1), U2 = quad_bcast<1>(r1), S1 = quad_bcast<2>(r1), S2 = quad_bcast<3>(r1);
    const Fp Pd = Fq::sub(U2, U1), Rd = Fq::sub(S2, S1);
    if (Fq::is_zero(Pd)) {                             // same x: doubling or cancellation (uniform over the quad)
        if (Fq::is_zero(Rd)) acc = xyzz_dbl(acc);
        else acc = xyzz_inf();
        return;
    }
    const Fp r2 = Fq::mul(quad_sel(q, Pd, Rd, acc.zz, acc.zzz), quad_sel(q, Pd, Rd, p.zz, p.zzz));
    const Fp PP = quad_bcast<0>(r2), RR = quad_bcast<1>(r2);
    const Fp r3 = Fq::mul(quad_sel(q, Pd, U1, r2, PP), PP);                  // PPP | Q | ZZ3 | unused
    const Fp PPP = quad_bcast<0>(r3), Q = quad_bcast<1>(r3);
    const Fp X3 = Fq::sub(Fq::sub(RR, PPP), Fq::dbl(Q));
    const Fp r4 = Fq::mul(quad_sel(q, Rd, S1, PPP, r2), quad_sel(q, Fq::sub(Q, X3), PPP, PPP, PPP));   // T1 | T2 | unused | ZZZ3
    acc.x = X3;
    acc.y = Fq::sub(quad_bcast<0>(r4), quad_bcast<1>(r4));
    acc.zz = quad_bcast<2>(r3);
    acc.zzz = quad_bcast<3>(r4);
}
// 2 * a by the four lanes of a quad (dbl-2008-s-1), three product stages instead of nine products:
//   stage 1: V = U^2 (U = 2Y) | X2 = X^2
//   stage 2: W = U V | S = X V | MM = M^2 (M = 3 X2) | ZZ3 = V ZZ              X3 = MM - 2S
//   stage 3: M (S - X3) | W Y | ZZZ3 = W ZZZ                                     Y3 = M (S - X3) - W Y
__device__ __forceinline__ void xyzz_dbl_quad(XYZZ& a, uint32_t q) {
    if (xyzz_is_inf(a)) return;
    const Fp U = Fq::dbl(a.y);
    const Fp r1 = Fq::mul(quad_sel(q, U, a.x, U, a.x), quad_sel(q, U, a.x, U, a.x));            // V | X2 | (V) | (X2)
    const Fp V = quad_bcast<0>(r1), X2 = quad_bcast<1>(r1);
    const Fp M = Fq::add(Fq::dbl(X2), X2);
    const Fp r2 = Fq::mul(quad_sel(q, U, a.x, M, V), quad_sel(q, V, V, M, a.zz));                // W | S | MM | ZZ3
    const Fp W = quad_bcast<0>(r2), S = quad_bcast<1>(r2), MM = quad_bcast<2>(r2);
    const Fp X3 = Fq::sub(MM, Fq::dbl(S));
    const Fp r3 = Fq::mul(quad_sel(q, M, W, W, W), quad_sel(q, Fq::sub(S, X3), a.y, a.zzz, a.zzz));   // T1 | T2 | ZZZ3 | (ZZZ3)
    a.x = X3;
    a.y = Fq::sub(quad_bcast<0>(r3), quad_bcast<1>(r3));
    a.zz = quad_bcast<3>(r2);
    a.zzz = quad_bcast<2>(r3);
}
#else
__device__ void xyzz_add_quad(XYZZ& acc, const XYZZ& p, uint32_t q);
__device__ void xyzz_dbl_quad(XYZZ& a, uint32_t q);
#endif


}  // namespace uzk
