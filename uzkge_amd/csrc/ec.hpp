// BN254 G1 group law (y^2 = x^3 + 3 over Fq) for host and device.
//
// Accumulators use XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): the mixed addition
// affine + XYZZ costs 8M + 2S with no inversion, the cheapest complete-by-branches formula for
// the Pippenger bucket loop (EFD "madd-2008-s", "add-2008-s", "dbl-2008-s-1").
// Infinity: affine (0,0) (never on the curve since 0 != 0 + 3) -- the wire format of
// include/uzkge_gpu.h; XYZZ / Jacobian infinity <=> ZZ == 0 / Z == 0.
// Every branch the bucket loop can meet is handled: acc == inf, P == inf, P == acc (doubling),
// P == -acc (-> inf); arkworks' `msm` has the same total behaviour
// (reference call site: uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290).
#pragma once
#include "fp256.hpp"

namespace uzk {

struct Affine {
    Fp x, y;
};
struct XYZZ {
    Fp x, y, zz, zzz;
};
struct Jac {
    Fp x, y, z;
};

UZK_HD bool affine_is_inf(const Affine& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
UZK_HD bool xyzz_is_inf(const XYZZ& p) { return Fq::is_zero(p.zz); }

UZK_HD XYZZ xyzz_inf() {
    XYZZ r;
    r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
    return r;
}
UZK_HD XYZZ xyzz_from_affine(const Affine& p) {
    XYZZ r;
    if (affine_is_inf(p)) return xyzz_inf();
    r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one();
    return r;
}
// 2*P for affine P (P != inf): dbl-2008-s-1 with ZZ = ZZZ = 1 ("mdbl-2008-s-1")
UZK_HD XYZZ xyzz_dbl_affine(const Affine& p) {
    XYZZ r;
    Fp U = Fq::dbl(p.y);
    Fp V = Fq::sqr(U);
    Fp W = Fq::mul(U, V);
    Fp S = Fq::mul(p.x, V);
    Fp X2 = Fq::sqr(p.x);
    Fp M = Fq::add(Fq::dbl(X2), X2);
    r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
    r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, p.y));
    r.zz = V;
    r.zzz = W;
    return r;
}
UZK_HD XYZZ xyzz_dbl(const XYZZ& p) {
    if (xyzz_is_inf(p)) return p;
    XYZZ r;
    Fp U = Fq::dbl(p.y);
    Fp V = Fq::sqr(U);
    Fp W = Fq::mul(U, V);
    Fp S = Fq::mul(p.x, V);
    Fp X2 = Fq::sqr(p.x);
    Fp M = Fq::add(Fq::dbl(X2), X2);
    r.x = Fq::sub(Fq::sqr(M), Fq::dbl(S));
    r.y = Fq::sub(Fq::mul(M, Fq::sub(S, r.x)), Fq::mul(W, p.y));
    r.zz = Fq::mul(V, p.zz);
    r.zzz = Fq::mul(W, p.zzz);
    return r;
}
// acc += (negate ? -p : p), p affine.  8M + 2S on the common path.
UZK_HD void xyzz_madd(XYZZ& acc, const Affine& p_in, bool negate) {
    if (affine_is_inf(p_in)) return;
    Affine p = p_in;
    if (negate) p.y = Fq::neg(p.y);
    if (xyzz_is_inf(acc)) { acc = xyzz_from_affine(p); return; }
    Fp U2 = Fq::mul(p.x, acc.zz);
    Fp S2 = Fq::mul(p.y, acc.zzz);
    Fp Pd = Fq::sub(U2, acc.x);
    Fp Rd = Fq::sub(S2, acc.y);
    if (Fq::is_zero(Pd)) {
        if (Fq::is_zero(Rd)) acc = xyzz_dbl_affine(p);
        else acc = xyzz_inf();
        return;
    }
    Fp PP = Fq::sqr(Pd);
    Fp PPP = Fq::mul(Pd, PP);
    Fp Q = Fq::mul(acc.x, PP);
    Fp X3 = Fq::sub(Fq::sub(Fq::sqr(Rd), PPP), Fq::dbl(Q));
    Fp Y3 = Fq::sub(Fq::mul(Rd, Fq::sub(Q, X3)), Fq::mul(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul(acc.zz, PP);
    acc.zzz = Fq::mul(acc.zzz, PPP);
}
#if defined(__HIP_DEVICE_COMPILE__)
// The bucket loop's mixed addition with the accumulator kept in the relaxed domain [0, 2M): same
// formulas, ten products without their final conditional subtraction.  p is canonical (an SRS
// point); the rare doubling / cancellation branches go through the canonical code.
__device__ __forceinline__ void xyzz_canon(XYZZ& a) {
    a.x = Fq::canon(a.x); a.y = Fq::canon(a.y); a.zz = Fq::canon(a.zz); a.zzz = Fq::canon(a.zzz);
}
__device__ __forceinline__ void xyzz_madd_rx(XYZZ& acc, const Affine& p_in, bool negate) {
    if (affine_is_inf(p_in)) return;
    Affine p = p_in;
    if (negate) p.y = Fq::neg(p.y);
    if (Fq::is_zero_rx(acc.zz)) { acc = xyzz_from_affine(p); return; }
    Fp U2 = Fq::mul_rx(p.x, acc.zz);
    Fp S2 = Fq::mul_rx(p.y, acc.zzz);
    Fp Pd = Fq::sub_rx(U2, acc.x);
    Fp Rd = Fq::sub_rx(S2, acc.y);
    if (Fq::is_zero_rx(Pd)) {
        if (Fq::is_zero_rx(Rd)) acc = xyzz_dbl_affine(p);
        else acc = xyzz_inf();
        return;
    }
    Fp PP = Fq::mul_rx(Pd, Pd);
    Fp PPP = Fq::mul_rx(Pd, PP);
    Fp Q = Fq::mul_rx(acc.x, PP);
    Fp X3 = Fq::sub_rx(Fq::sub_rx(Fq::mul_rx(Rd, Rd), PPP), Fq::dbl_rx(Q));
    Fp Y3 = Fq::sub_rx(Fq::mul_rx(Rd, Fq::sub_rx(Q, X3)), Fq::mul_rx(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul_rx(acc.zz, PP);
    acc.zzz = Fq::mul_rx(acc.zzz, PPP);
}
#elif defined(__HIPCC__)
// host pass of a HIP translation unit: kernels that call these are only parsed, never emitted
__device__ void xyzz_canon(XYZZ& a);
__device__ void xyzz_madd_rx(XYZZ& acc, const Affine& p_in, bool negate);
#endif

// acc += q, both XYZZ.  12M + 2S.
UZK_HD void xyzz_add(XYZZ& acc, const XYZZ& q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) { acc = q; return; }
    Fp U1 = Fq::mul(acc.x, q.zz);
    Fp U2 = Fq::mul(q.x, acc.zz);
    Fp S1 = Fq::mul(acc.y, q.zzz);
    Fp S2 = Fq::mul(q.y, acc.zzz);
    Fp Pd = Fq::sub(U2, U1);
    Fp Rd = Fq::sub(S2, S1);
    if (Fq::is_zero(Pd)) {
        if (Fq::is_zero(Rd)) acc = xyzz_dbl(acc);
        else acc = xyzz_inf();
        return;
    }
    Fp PP = Fq::sqr(Pd);
    Fp PPP = Fq::mul(Pd, PP);
    Fp Q = Fq::mul(U1, PP);
    Fp X3 = Fq::sub(Fq::sub(Fq::sqr(Rd), PPP), Fq::dbl(Q));
    Fp Y3 = Fq::sub(Fq::mul(Rd, Fq::sub(Q, X3)), Fq::mul(S1, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = Fq::mul(Fq::mul(acc.zz, q.zz), PP);
    acc.zzz = Fq::mul(Fq::mul(acc.zzz, q.zzz), PPP);
}
// XYZZ -> Jacobian without inversion: (X', Y', Z') = (X*ZZ, Y*ZZZ, ZZ) satisfies
// X'/Z'^2 = X/ZZ and Y'/Z'^3 = Y*ZZZ/ZZ^3 = Y/ZZZ  (since ZZ^3 = ZZZ^2).
UZK_HD Jac xyzz_to_jac(const XYZZ& p) {
    Jac r;
    if (xyzz_is_inf(p)) { r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero(); return r; }
    r.x = Fq::mul(p.x, p.zz);
    r.y = Fq::mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}
UZK_HD XYZZ xyzz_from_jac(const Jac& p) {
    XYZZ r;
    if (Fq::is_zero(p.z)) return xyzz_inf();
    r.x = p.x; r.y = p.y;
    r.zz = Fq::sqr(p.z);
    r.zzz = Fq::mul(r.zz, p.z);
    return r;
}

}  // namespace uzk
