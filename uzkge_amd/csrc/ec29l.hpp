// Full XYZZ additions of the MSM's bucket-side kernels on the lazy 29-bit arithmetic with bounds in the types (lz29.hpp; round 6).
//
// The bucket arrays keep the wire's form (canonical 8 x 32-bit words, Montgomery radix 2^256): a coordinate enters the 2^261-form
// by re-limbing alone (LzOps::ld: 32 a = x 2^261, value < 32 M, normalized limbs) -- no conversion product per loaded point, which
// is what made 29-bit limbs lose wherever every addition loads a fresh operand (ecquad29.hpp pays one product per load).  A point
// leaves through p29_store: one product + canon per coordinate, paid once per OUTPUT, not per addition.
//
// P29: x, y, zz, zzz in 2^261-form, limbs normalized (K = 1), values < 32 M -- the bound of a freshly loaded point, which every
// result of the formulas below stays under, so loaded points and sums are one type.  Infinity <=> every limb of zz is zero: a load
// of the wire's zz = 0 gives exactly that, a product of non-zero residues never does, and cancellation sets it explicitly.
#pragma once
#include "ec.hpp"
#include "lz29.hpp"

namespace uzk {

struct P29 {
    L29 x, y, zz, zzz;
};

#if defined(__HIP_DEVICE_COMPILE__)

namespace p29 {
using Z = LzOps<Fq29>;
using Co = Lz<Fq29, 1, 32>;                                        // a coordinate
__device__ __forceinline__ Co co(const L29& v) { Co r; r.v = v; return r; }
// value < 3 M, normalized: is it 0 mod M?  (first a one-limb filter: the low limb of 0, M, 2M)
template <int V>
__device__ __forceinline__ bool is_zero(const Lz<Fq29, 1, V>& a) {
    static_assert(V <= 3, "is_zero_mod_small covers 0, M, 2M");
    const uint32_t t = a.v.l[0];
    return (t == 0 || t == Fq29Cfg::M[0] || t == Fq29Cfg::M2[0]) && Fq29::is_zero_mod_small(a.v);
}
}  // namespace p29

__device__ __forceinline__ bool p29_is_inf(const P29& p) { return Fq29::all_zero(p.zz); }
__device__ __forceinline__ P29 p29_inf() {
    P29 r;
    r.x = Fq29::zero(); r.y = Fq29::zero(); r.zz = Fq29::zero(); r.zzz = Fq29::zero();
    return r;
}
__device__ __forceinline__ P29 p29_load(const XYZZ& p) {
    using namespace p29;
    P29 r;
    r.x = Z::ld(p.x).v; r.y = Z::ld(p.y).v; r.zz = Z::ld(p.zz).v; r.zzz = Z::ld(p.zzz).v;
    return r;
}
// -> canonical wire form (one product and one canon per coordinate)
__device__ __forceinline__ XYZZ p29_store(const P29& p) {
    using namespace p29;
    if (p29_is_inf(p)) return xyzz_inf();
    XYZZ r;
    r.x = Z::to_wire(co(p.x)); r.y = Z::to_wire(co(p.y)); r.zz = Z::to_wire(co(p.zz)); r.zzz = Z::to_wire(co(p.zzz));
    return r;
}

// 2 a (dbl-2008-s-1); a != infinity
__device__ __forceinline__ void p29_dbl(P29& a) {
    using namespace p29;
    const Co X = co(a.x), Y = co(a.y), ZZ = co(a.zz), ZZZ = co(a.zzz);
    const auto U = Z::add(Y, Y);                                   // (2, 64)
    const auto V = Z::sqr(U);
    const auto W = Z::mul(U, V);
    const auto S = Z::mul(X, V);
    const auto X2 = Z::sqr(X);
    const auto M3 = Z::norm(Z::add(Z::add(X2, X2), X2));
    const auto X3 = Z::norm(Z::sub(Z::sqr(M3), Z::add(S, S)));
    const auto D = Z::norm(Z::sub(S, X3));
    const auto Y3 = Z::mul2(M3, D, W, Z::sub(Z::zero(), Y));       // M (S - X3) - W Y
    a.x = Z::template relax<1, 32>(X3).v;
    a.y = Z::template relax<1, 32>(Y3).v;
    a.zz = Z::template relax<1, 32>(Z::mul(V, ZZ)).v;
    a.zzz = Z::template relax<1, 32>(Z::mul(W, ZZZ)).v;
}

// acc += q (add-2008-s), complete: either may be infinity, equal points double, opposite points cancel.
// 10 products + 2 squarings + 1 dual product, four parallel carry steps.
__device__ __forceinline__ void p29_add(P29& acc, const P29& q) {
    using namespace p29;
    if (p29_is_inf(q)) return;
    if (p29_is_inf(acc)) { acc = q; return; }
    const Co X1 = co(acc.x), Y1 = co(acc.y), ZZ1 = co(acc.zz), ZZZ1 = co(acc.zzz);
    const Co X2 = co(q.x), Y2 = co(q.y), ZZ2 = co(q.zz), ZZZ2 = co(q.zzz);
    const auto U1 = Z::mul(X1, ZZ2), U2 = Z::mul(X2, ZZ1), S1 = Z::mul(Y1, ZZZ2), S2 = Z::mul(Y2, ZZZ1);     // < 8 M
    const auto P = Z::norm(Z::sub(U2, U1)), R = Z::norm(Z::sub(S2, S1));                                         // < 17 M
    const auto PP = Z::sqr(P);                                                                                   // < 3 M
    if (is_zero(PP)) {                                             // same x: the points are equal or opposite
        if (is_zero(Z::sqr(R))) p29_dbl(acc);
        else acc = p29_inf();
        return;
    }
    const auto RR = Z::sqr(R);
    const auto ZZ12 = Z::mul(ZZ1, ZZ2), ZZZ12 = Z::mul(ZZZ1, ZZZ2);
    const auto PPP = Z::mul(P, PP), Q = Z::mul(U1, PP);
    const auto X3 = Z::norm(Z::sub(RR, Z::add(PPP, Z::add(Q, Q))));                                              // RR - PPP - 2 Q
    const auto D = Z::norm(Z::sub(Q, X3));
    const auto Y3 = Z::mul2(R, D, S1, Z::sub(Z::zero(), PPP));                                                   // R (Q - X3) - S1 PPP
    acc.x = Z::template relax<1, 32>(X3).v;
    acc.y = Z::template relax<1, 32>(Y3).v;
    acc.zz = Z::template relax<1, 32>(Z::mul(ZZ12, PP)).v;
    acc.zzz = Z::template relax<1, 32>(Z::mul(ZZZ12, PPP)).v;
}

// ---- the same by the four lanes of a quad (ecquad.hpp's stage scheme): every lane passes the same operands and leaves with the
// same sum.  A stage's product has lane-dependent operands; its result carries the loosest of the four bounds, and the broadcast
// of ONE lane's result restates that lane's own bound (qbc<lane, V>: V is written as the prod_v of that lane's operands).
namespace p29 {
template <int S, int V, int K0, int V0>
__device__ __forceinline__ Lz<Fq29, 1, V> qbc(const Lz<Fq29, K0, V0>& r) {
    static_assert(K0 == 1 && V <= V0, "a broadcast restates one lane's bound; it cannot exceed the stage's");
    Lz<Fq29, 1, V> o;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        o.v.l[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.v.l[k], S * 0x55, 0xf, 0xf, false);
        asm volatile("" : "+v"(o.v.l[k]));                         // keep the move a move (ecquad29.hpp quad_bcast29)
    }
    return o;
}
constexpr int max4(int a, int b, int c, int d) { return (a > b ? a : b) > (c > d ? c : d) ? (a > b ? a : b) : (c > d ? c : d); }
template <int K0, int V0, int K1, int V1, int K2, int V2, int K3, int V3>
__device__ __forceinline__ Lz<Fq29, max4(K0, K1, K2, K3), max4(V0, V1, V2, V3)> qsel(uint32_t q, const Lz<Fq29, K0, V0>& a0, const Lz<Fq29, K1, V1>& a1,
                                                                                      const Lz<Fq29, K2, V2>& a2, const Lz<Fq29, K3, V3>& a3) {
    Lz<Fq29, max4(K0, K1, K2, K3), max4(V0, V1, V2, V3)> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const uint32_t lo = (q & 1) ? a1.v.l[k] : a0.v.l[k], hi = (q & 1) ? a3.v.l[k] : a2.v.l[k];
        r.v.l[k] = (q & 2) ? hi : lo;
    }
    return r;
}
}  // namespace p29

// 2 a by a quad, three product stages
__device__ __forceinline__ void p29_dbl_quad(P29& a, uint32_t q) {
    using namespace p29;
    const Co X = co(a.x), Y = co(a.y), ZZ = co(a.zz), ZZZ = co(a.zzz);
    const auto U = Z::add(Y, Y);                                                          // (2, 64)
    const auto r1 = Z::sqr(qsel(q, U, Z::template relax<2, 64>(X), U, Z::template relax<2, 64>(X)));      // V | X2
    const auto V = qbc<0, Z::prod_v(64, 64)>(r1);
    const auto X2 = qbc<1, Z::prod_v(32, 32)>(r1);
    const auto M3 = Z::norm(Z::add(Z::add(X2, X2), X2));                                  // (1, 24)
    const auto r2 = Z::mul(qsel(q, U, Z::template relax<2, 64>(X), Z::template relax<2, 64>(M3), Z::template relax<2, 64>(V)),
                           qsel(q, V, V, Z::template relax<1, 26>(M3), Z::template relax<1, 32>(ZZ)));            // W | S | MM | ZZ3
    const auto W = qbc<0, Z::prod_v(64, 26)>(r2);
    const auto S = qbc<1, Z::prod_v(32, 26)>(r2);
    const auto MM = qbc<2, Z::prod_v(24, 24)>(r2);
    const auto ZZ3 = qbc<3, Z::prod_v(26, 32)>(r2);
    const auto X3 = Z::norm(Z::sub(MM, Z::add(S, S)));
    const auto D = Z::norm(Z::sub(S, X3));
    const auto r3 = Z::mul(qsel(q, Z::template relax<1, 24>(M3), Z::template relax<1, 24>(W), Z::template relax<1, 24>(W), Z::template relax<1, 24>(W)),
                           qsel(q, Z::template relax<1, 32>(D), Y, ZZZ, ZZZ));                                   // T1 | T2 | ZZZ3
    static_assert(decltype(W)::val_v <= 24 && decltype(D)::val_v <= 32, "stage 3 operand bounds");
    const auto T1 = qbc<0, Z::prod_v(24, 32)>(r3), T2 = qbc<1, Z::prod_v(24, 32)>(r3);
    a.x = Z::template relax<1, 32>(X3).v;
    a.y = Z::template relax<1, 32>(Z::norm(Z::sub(T1, T2))).v;
    a.zz = Z::template relax<1, 32>(ZZ3).v;
    a.zzz = Z::template relax<1, 32>(qbc<2, Z::prod_v(24, 32)>(r3)).v;
}

// acc += p by a quad, four product stages
__device__ __forceinline__ void p29_add_quad(P29& acc, const P29& p, uint32_t q) {
    using namespace p29;
    if (p29_is_inf(p)) return;
    if (p29_is_inf(acc)) { acc = p; return; }
    const Co X1 = co(acc.x), Y1 = co(acc.y), ZZ1 = co(acc.zz), ZZZ1 = co(acc.zzz);
    const Co X2 = co(p.x), Y2 = co(p.y), ZZ2 = co(p.zz), ZZZ2 = co(p.zzz);
    const auto r1 = Z::mul(qsel(q, X1, X2, Y1, Y2), qsel(q, ZZ2, ZZ1, ZZZ2, ZZZ1));       // U1 | U2 | S1 | S2: all < 8 M
    constexpr int VU = Z::prod_v(32, 32);
    const auto U1 = qbc<0, VU>(r1), U2 = qbc<1, VU>(r1), S1 = qbc<2, VU>(r1), S2 = qbc<3, VU>(r1);
    const auto P = Z::norm(Z::sub(U2, U1)), R = Z::norm(Z::sub(S2, S1));                  // < 17 M
    constexpr int VP = decltype(P)::val_v;
    const auto r2 = Z::mul(qsel(q, Z::template relax<1, 32>(P), Z::template relax<1, 32>(R), ZZ1, ZZZ1),
                           qsel(q, Z::template relax<1, 32>(P), Z::template relax<1, 32>(R), ZZ2, ZZZ2));        // PP | RR | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const auto PP = qbc<0, Z::prod_v(VP, VP)>(r2), RR = qbc<1, Z::prod_v(VP, VP)>(r2);
    const auto ZZ12 = qbc<2, VU>(r2), ZZZ12 = qbc<3, VU>(r2);
    if (is_zero(PP)) {                                             // same x (uniform over the quad)
        if (is_zero(RR)) p29_dbl_quad(acc, q);
        else acc = p29_inf();
        return;
    }
    constexpr int VPP = decltype(PP)::val_v;
    const auto r3 = Z::mul(qsel(q, P, Z::template relax<1, VP>(U1), Z::template relax<1, VP>(ZZ12), Z::template relax<1, VP>(ZZ12)), PP);   // PPP | Q | ZZ3 | -
    const auto PPP = qbc<0, Z::prod_v(VP, VPP)>(r3);
    const auto Q = qbc<1, Z::prod_v(VU, VPP)>(r3);
    const auto ZZ3 = qbc<2, Z::prod_v(VU, VPP)>(r3);
    const auto X3 = Z::norm(Z::sub(RR, Z::add(PPP, Z::add(Q, Q))));
    const auto D = Z::norm(Z::sub(Q, X3));
    constexpr int VD = decltype(D)::val_v, VPPP = decltype(PPP)::val_v;
    static_assert(VD <= VP && VPPP <= VP, "stage 4 operand bounds");
    const auto r4 = Z::mul(qsel(q, R, Z::template relax<1, VP>(S1), Z::template relax<1, VP>(ZZZ12), Z::template relax<1, VP>(ZZZ12)),
                           qsel(q, Z::template relax<1, VP>(D), Z::template relax<1, VP>(PPP), Z::template relax<1, VP>(PPP), Z::template relax<1, VP>(PPP)));   // T1 | T2 | - | ZZZ3
    const auto T1 = qbc<0, Z::prod_v(VP, VD)>(r4);
    const auto T2 = qbc<1, Z::prod_v(VU, VPPP)>(r4);
    acc.x = Z::template relax<1, 32>(X3).v;
    acc.y = Z::template relax<1, 32>(Z::norm(Z::sub(T1, T2))).v;
    acc.zz = Z::template relax<1, 32>(ZZ3).v;
    acc.zzz = Z::template relax<1, 32>(qbc<3, Z::prod_v(VU, VPPP)>(r4)).v;
}
// wire -> P29 in every lane of the quad (re-limbing only); P29 -> coordinate q of the wire form in lane q (x | y | zz | zzz)
__device__ __forceinline__ Fp p29_coord_to_fp(const P29& p, uint32_t q) {
    using namespace p29;
    return Z::to_wire(qsel(q, co(p.x), co(p.y), co(p.zz), co(p.zzz)));
}

#elif defined(__HIPCC__)
// host pass of a .hip file: the kernels' bodies are parsed, never run
__device__ bool p29_is_inf(const P29& p);
__device__ P29 p29_inf();
__device__ P29 p29_load(const XYZZ& p);
__device__ XYZZ p29_store(const P29& p);
__device__ void p29_dbl(P29& a);
__device__ void p29_add(P29& acc, const P29& q);
__device__ void p29_dbl_quad(P29& a, uint32_t q);
__device__ void p29_add_quad(P29& acc, const P29& p, uint32_t q);
__device__ Fp p29_coord_to_fp(const P29& p, uint32_t q);
#endif

}  // namespace uzk
