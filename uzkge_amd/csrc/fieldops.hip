// Element-wise known-answer entry points: run the DEVICE field / group primitives on host arrays so
// tests can compare them word-for-word with the oracle (Fq/Fr mont-mul/add/sub KATs, G1
// add / double / mixed-add including P+P, P+(-P) and infinity; SURVEY.md 8c "golden vectors").
#include "ctx.hpp"
#include "ecquad.hpp"
#include "ecquad29.hpp"
#include "ec29l.hpp"
#include "fp29.hpp"
#include "lz29.hpp"

namespace uzk {

template <class F, class F29>
__global__ __launch_bounds__(256) void field_op_kernel(int op, const Fp* __restrict__ a, const Fp* __restrict__ b,
                                                       Fp* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp x = a[i], y = b[i], r;
    switch (op) {
        case 0: r = F::mul(x, y); break;
        case 1: r = F::add(x, y); break;
        case 2: r = F::sub(x, y); break;
        case 3: r = F::mul_portable(x, y); break;
        case 4: r = F::sqr(x); break;
        case 5: r = F::neg(x); break;
        case 6: r = F::from_mont(x); break;
        case 8: r = F::add_portable(x, y); break;
        case 9: r = F::sub_portable(x, y); break;
        // ---- the 9 x 29-bit-limb representation (fp29.hpp), results mapped back to canonical words ----
        case 10: r = F29::to_fp(F29::canon(F29::mul(F29::from_fp(x), F29::to_261(F29::from_fp(y))))); break;   // x*y (2^256-form)
        case 11: r = F29::to_fp(F29::canon(F29::add(F29::from_fp(x), F29::from_fp(y)))); break;
        case 12: r = F29::to_fp(F29::canon(F29::template sub<4>(F29::from_fp(x), F29::from_fp(y)))); break;
        case 13: r = F29::to_fp(F29::canon(F29::template sub<12>(F29::from_fp(x), F29::from_fp(y)))); break;
        case 14: {   // lazy chain at the documented limb bounds: ((x - y + 12M) * (x + y)) with un-normalized operands
            L29 a = F29::from_fp(x), b = F29::from_fp(y);
            L29 d = F29::template sub<12>(a, b);                 // limbs < 2^31
            L29 sum = F29::add(a, b);                            // limbs < 2^30
            L29 p = F29::mul(F29::norm(d), sum);                 // normalized x lazy sum
            L29 q = F29::mul(d, F29::to_261(F29::from_fp(y)));   // lazy difference x normalized
            L29 t = F29::template sub<4>(F29::add(p, p), q);     // 2p - q, lazy
            r = F29::to_fp(F29::canon(F29::to_256(t)));
        } break;
        case 15: r = F29::to_fp(F29::canon(F29::to_256(F29::to_261(F29::from_fp(x))))); break;   // 256 -> 261 -> 256
        case 16: { L29 t = F29::add(F29::from_fp(x), F29::from_fp(y)); r = F29::to_fp(F29::canon(F29::sqr(t))); } break;      // dedicated squaring, lazy operand
        case 17: { L29 t = F29::add(F29::from_fp(x), F29::from_fp(y)); r = F29::to_fp(F29::canon(F29::mul(t, t))); } break;   // ... against the general product
        case 18: {   // x - 3y through the three-subtrahend offset (the -X3 step of ec29.hpp)
            L29 a = F29::from_fp(x), b = F29::from_fp(y), t;
            for (int k = 0; k < 9; ++k) t.l[k] = a.l[k] + F29::Cfg::OFF4T3[k] - 2 * b.l[k] - b.l[k];
            r = F29::to_fp(F29::canon(t));
        } break;
        case 19: r = F29::to_fp(F29::canon(F29::sub_off(F29::from_fp(x), F29::from_fp(y), F29::Cfg::OFF2T1))); break;
        case 20: {   // dual product with one reduction: x*y + (x + y)*x (2^256-form), second pair lazy
            L29 a = F29::from_fp(x), b = F29::from_fp(y);
            r = F29::to_fp(F29::canon(F29::mul2(a, F29::to_261(b), F29::add(a, b), F29::to_261(a))));
        } break;
        // 21..23: the plain C++ products (ops 10, 16, 20 run the generated assembly chains)
        case 21: r = F29::to_fp(F29::canon(F29::mul_cpp(F29::from_fp(x), F29::to_261(F29::from_fp(y))))); break;
        case 22: { L29 t = F29::add(F29::from_fp(x), F29::from_fp(y)); r = F29::to_fp(F29::canon(F29::sqr_cpp(t))); } break;
        case 23: {
            L29 a = F29::from_fp(x), b = F29::from_fp(y);
            r = F29::to_fp(F29::canon(F29::mul2_cpp(a, F29::to_261(b), F29::add(a, b), F29::to_261(a))));
        } break;
        // the constant-operand product (fp29.hpp mulc): y canonical, x any 256-bit value; out = x * y mod M as PLAIN integers
        case 24: { const L29 w = F29::from_fp(y); r = F29::to_fp(F29::canon(F29::mulc(F29::from_fp(x), w, F29::wq_of(w)))); } break;
        // ... with a lazy first operand near the limb bound: 2 (x + 4M) as limb-wise sums (limbs < 2^31.7, value < 19M)
        case 25: {
            const L29 w = F29::from_fp(y);
            L29 t = F29::add(F29::from_fp(x), F29::constant(F29::Cfg::OFF4));
            t = F29::add(t, t);
            r = F29::to_fp(F29::canon(F29::mulc(t, w, F29::wq_of(w))));
        } break;
        // wq_of(y) itself, low 256 bits (floor(y 2^261 / M) mod 2^256)
        case 26: r = F29::to_fp(F29::wq_of(F29::from_fp(y))); break;
        // reduce3 of the lazy sum x + y + 4M (x, y any 256-bit values: limbs < 2^31.4, value < 14.6M for Fr and Fq), RAW: the caller
        // checks the residue and the bound value < 3M
        case 27: r = F29::to_fp(F29::reduce3(F29::add(F29::add(F29::from_fp(x), F29::from_fp(y)), F29::constant(F29::Cfg::OFF4)))); break;
        // ---- the typed lazy arithmetic (lz29.hpp) and its two free conversions; x, y canonical wire elements -----------------
        // 28: re-limb at bit offset -5 (32 x = the 2^261-form), back by exact division by 32: the identity
        case 28: r = F::canon(F29::template to_fp_div<5>(F29::from_fp_x32(x))); break;
        // 29 / 30 / 31: x y, x - y, x y + y y through ld -> typed operation -> to_wire: the wire-form results of ops 0 / 2 and 0 + 4
        case 29: { using Z = LzOps<F29>; r = Z::to_wire(Z::mul(Z::ld(x), Z::ld(y))); } break;
        case 30: { using Z = LzOps<F29>; r = Z::to_wire(Z::norm(Z::sub(Z::ld(x), Z::ld(y)))); } break;
        case 31: { using Z = LzOps<F29>; r = Z::to_wire(Z::mul2(Z::ld(x), Z::ld(y), Z::ld(y), Z::ld(y))); } break;
        // 32: a value in 2^266-form leaves by exact division by 2^10: (x in 2^261-form) * 2^266 / 2^261 = the 2^266-form of x; -> x
        case 32: { const L29 v = F29::mul(F29::from_fp_x32(x), F29::constant(F29::Cfg::R266)); r = F::canon(F29::template to_fp_div<10>(v)); } break;
        // 33: a long lazy chain at the limb bounds: ((x + y) + (x - y)) - ((y - x) - x)  =  3 x - y  ... all typed, one carry step where the types ask
        case 33: {
            using Z = LzOps<F29>;
            const auto a1 = Z::ld(x), b1 = Z::ld(y);
            const auto s1 = Z::add(Z::add(a1, b1), Z::sub(a1, b1));                 // 2 x (+ 33 M), limbs < 6 units
            const auto s2 = Z::norm(Z::sub(Z::norm(Z::sub(b1, a1)), a1));           // y - 2 x (+ offsets)
            r = Z::to_wire(Z::mul(Z::norm(Z::sub(Z::norm(s1), s2)), Z::one()));     // (4 x - y) * 1, through a product: value back below 16 M
        } break;
        default: r = F::to_mont(x); break;
    }
    out[i] = r;
}

__global__ __launch_bounds__(256) void g1_op_kernel(int op, const Affine* __restrict__ a, const Affine* __restrict__ b,
                                                    Jac* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p = a[i], q = b[i];
    XYZZ acc = xyzz_from_affine(p);
    switch (op) {
        case 0: xyzz_madd(acc, q, false); break;                       // mixed add
        case 1: { XYZZ t = xyzz_from_affine(q); xyzz_add(acc, t); } break;   // full add
        case 2: acc = xyzz_dbl(acc); break;                            // double
        case 3: xyzz_madd(acc, q, true); break;                        // mixed subtract
        default: {                                                      // (p + q) + (p + q) through non-trivial ZZ
            xyzz_madd(acc, q, false);
            XYZZ t = acc;
            xyzz_add(acc, t);
        } break;
    }
    out[i] = xyzz_to_jac(acc);
}

// ops 5..7: the quad addition (ecquad.hpp), four lanes per element
__global__ __launch_bounds__(256) void g1_quad_op_kernel(int op, const Affine* __restrict__ a, const Affine* __restrict__ b,
                                                         Jac* __restrict__ out, size_t n) {
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = gt >> 2;
    const uint32_t q = (uint32_t)(gt & 3);
    if (i >= n) return;
    const Affine p = a[i], r = b[i];
    XYZZ acc = xyzz_from_affine(p);
    if (op == 5) {                                   // p + r, trivial ZZ on both sides
        const XYZZ t = xyzz_from_affine(r);
        xyzz_add_quad(acc, t, q);
    } else if (op == 6) {                            // (p + r) + (p + r): the doubling branch through non-trivial ZZ
        xyzz_madd(acc, r, false);
        const XYZZ t = acc;
        xyzz_add_quad(acc, t, q);
    } else {                                         // (p + r) + (p - r) = 2p: general case, non-trivial ZZ on both sides
        XYZZ t = acc;
        xyzz_madd(acc, r, false);
        xyzz_madd(t, r, true);
        xyzz_add_quad(acc, t, q);
    }
    if (q == 0) out[i] = xyzz_to_jac(acc);
}

// ops 8..13: the same on the 29-bit-limb representation (ecquad29.hpp): 8 a + b, 9 2(a + b) through the addition's
// doubling branch, 10 (a + b) + (a - b), 11 4(a + b) by two quad doublings, 12 2(a + b) by one, 13 4a
__global__ __launch_bounds__(256) void g1_quad29_op_kernel(int op, const Affine* __restrict__ a, const Affine* __restrict__ b,
                                                           Jac* __restrict__ out, size_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = gt >> 2;
    const uint32_t q = (uint32_t)(gt & 3);
    if (i >= n) return;
    const Affine p = a[i], r = b[i];
    XYZZ s = xyzz_from_affine(p), t = xyzz_from_affine(r);
    X29 acc;
    if (op == 8) {
        acc = x29_from_xyzz_quad(s, q);
        x29_add_quad(acc, x29_from_xyzz_quad(t, q), q);
    } else if (op == 9) {
        xyzz_madd(s, r, false);
        acc = x29_from_xyzz_quad(s, q);
        const X29 same = acc;
        x29_add_quad(acc, same, q);
    } else if (op == 10) {
        XYZZ u = s;
        xyzz_madd(s, r, false);
        xyzz_madd(u, r, true);
        acc = x29_from_xyzz_quad(s, q);
        x29_add_quad(acc, x29_from_xyzz_quad(u, q), q);
    } else if (op == 11) {
        acc = x29_from_xyzz_quad(s, q);
        x29_add_quad(acc, x29_from_xyzz_quad(t, q), q);
        x29_dbl_quad(acc, q);
        x29_dbl_quad(acc, q);
    } else if (op == 12) {                           // 2(a + b): one doubling of an addition's result
        acc = x29_from_xyzz_quad(s, q);
        x29_add_quad(acc, x29_from_xyzz_quad(t, q), q);
        x29_dbl_quad(acc, q);
    } else {                                         // 4a: doubling of a doubling's result
        acc = x29_from_xyzz_quad(s, q);
        x29_dbl_quad(acc, q);
        x29_dbl_quad(acc, q);
    }
    const Fp mine = x29_coord_to_fp(acc, q);
    XYZZ o;
    o.x = quad_bcast<0>(mine); o.y = quad_bcast<1>(mine); o.zz = quad_bcast<2>(mine); o.zzz = quad_bcast<3>(mine);
    if (q == 0) out[i] = xyzz_to_jac(o);
#endif
}

int field_op_device(Ctx& c, int field, int op, const Fp* a, const Fp* b, Fp* out, size_t n) {
    if (n == 0) return UZK_OK;
    Fp *da = nullptr, *db = nullptr, *dout = nullptr;
    const size_t bytes = n * sizeof(Fp);
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&da), bytes));
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&db), bytes));
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&dout), bytes));
    UZK_HIP(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, c.stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (field == 0) hipLaunchKernelGGL((field_op_kernel<Fq, Fq29>), dim3(grid), dim3(256), 0, c.stream, op, da, db, dout, n);
    else hipLaunchKernelGGL((field_op_kernel<Fr, Fr29>), dim3(grid), dim3(256), 0, c.stream, op, da, db, dout, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    UZK_HIP(e);
    return UZK_OK;
}

// ops 14..21: the additions of ec29l.hpp (lazy 29-bit limbs, operands re-limbed from the wire form, bounds in the types).
// one lane per element: 14 a + b, 15 2(a + b) through the addition's doubling branch, 16 (a + b) + (a - b) (non-trivial ZZ on both
// sides), 17 (a + b) - (a + b) = infinity (cancellation) ... then + a; four lanes per element: 18..21 the same four by quads.
__global__ __launch_bounds__(256) void g1_p29_op_kernel(int op, const Affine* __restrict__ a, const Affine* __restrict__ b,
                                                        Jac* __restrict__ out, size_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
    const bool quad = op >= 18;
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i = quad ? gt >> 2 : gt;
    const uint32_t q = (uint32_t)(gt & 3);
    if (i >= n) return;
    const Affine p = a[i], r = b[i];
    XYZZ s = xyzz_from_affine(p), t = xyzz_from_affine(r);
    const int f = quad ? op - 18 : op - 14;
    auto add = [&](P29& x, const P29& y) { if (quad) p29_add_quad(x, y, q); else p29_add(x, y); };
    P29 acc;
    if (f == 0) {
        acc = p29_load(s);
        add(acc, p29_load(t));
    } else if (f == 1) {
        xyzz_madd(s, r, false);
        acc = p29_load(s);
        const P29 same = acc;
        add(acc, same);
    } else if (f == 2) {
        XYZZ u = s;
        xyzz_madd(s, r, false);
        xyzz_madd(u, r, true);
        acc = p29_load(s);
        add(acc, p29_load(u));
    } else {
        XYZZ u = s;
        xyzz_madd(u, r, false);                          // a + b
        XYZZ v = u;
        v.y = Fq::neg(v.y);                              // -(a + b), same ZZ / ZZZ
        acc = p29_load(u);
        add(acc, p29_load(v));                           // infinity
        add(acc, p29_load(s));                           // + a
        P29 w = p29_load(t);
        add(w, p29_inf());                               // b + infinity
        add(acc, w);                                     // a + b
    }
    if (quad) {
        const Fp c = p29_is_inf(acc) ? Fq::zero() : p29_coord_to_fp(acc, q);
        __shared__ Fp sh[64][4];
        sh[threadIdx.x >> 2][q] = c;
        __syncthreads();
        if (q == 0) {
            XYZZ o;
            o.x = sh[threadIdx.x >> 2][0]; o.y = sh[threadIdx.x >> 2][1]; o.zz = sh[threadIdx.x >> 2][2]; o.zzz = sh[threadIdx.x >> 2][3];
            out[i] = xyzz_to_jac(o);
        }
    } else {
        out[i] = xyzz_to_jac(p29_store(acc));
    }
#endif
}

int g1_op_device(Ctx& c, int op, const Affine* a, const Affine* b, Jac* out, size_t n) {
    if (n == 0) return UZK_OK;
    Affine *da = nullptr, *db = nullptr;
    Jac* dout = nullptr;
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&da), n * sizeof(Affine)));
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&db), n * sizeof(Affine)));
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&dout), n * sizeof(Jac)));
    UZK_HIP(hipMemcpyAsync(da, a, n * sizeof(Affine), hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipMemcpyAsync(db, b, n * sizeof(Affine), hipMemcpyHostToDevice, c.stream));
    if (op >= 18) hipLaunchKernelGGL(g1_p29_op_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, c.stream, op, da, db, dout, n);
    else if (op >= 14) hipLaunchKernelGGL(g1_p29_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, op, da, db, dout, n);
    else if (op >= 8) hipLaunchKernelGGL(g1_quad29_op_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, c.stream, op, da, db, dout, n);
    else if (op >= 5) hipLaunchKernelGGL(g1_quad_op_kernel, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, c.stream, op, da, db, dout, n);
    else hipLaunchKernelGGL(g1_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, op, da, db, dout, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, n * sizeof(Jac), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    UZK_HIP(e);
    return UZK_OK;
}

}  // namespace uzk
