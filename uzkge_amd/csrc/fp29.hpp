// "L29": the device-only representation used inside the two hot loops (bucket accumulation, NTT
// passes).  A field element is 9 limbs of 29 bits, value = sum l_i 2^(29 i); limbs may temporarily
// hold more than 29 bits (lazy carries) and values more than M (lazy reduction).
//
// Why (measured, tools/microbench/l29_rate.hip): with 29-bit limbs a column of the product scanning
// Montgomery product -- up to 9 a_i*b_j plus 9 m_i*M_j partial products -- fits a 64-bit accumulator,
// so every partial product is ONE v_mad_u64_u32 with no carry bookkeeping: 162 MADs instead of
// 128 MADs + 128 v_addc through VCC.  900 vs 1131 cycles per wave-product at 4 waves/SIMD (-20 %; the
// three products are generated assembly column chains, mul29_gfx950.inc, with the C++ forms kept as
// *_cpp), and additions/subtractions become 9 independent 32-bit adds with no reduction at all.
//
// Contract of the operations (B = 2^29):
//   mul(a, b)   limbs: max(a_i) * max(b_j) < 2^60.6 (e.g. both < 2^30.3, or < B and < 2^31.6);
//               values: a < Va*M, b < Vb*M  ->  result limbs < B (normalized), value < M*(1 + Va*Vb/169).
//               Montgomery radix is 2^261: mul(a, b) = a*b*2^-261 mod M (up to the lazy multiple of M).
//   add(a, b)   limb-wise, no carry, no reduction (caller keeps limbs < 2^32 and values in range).
//   sub(a, b, OFF)  a - b + OFF limb-wise, OFF = k*M with limbs pre-borrowed so that no limb underflows:
//               OFF4 / OFF12 (low limbs >= 2^30 - 2: b normalized or a sum of two), OFF4T3 (three
//               normalized subtrahends), OFF2T1 / OFF8T1 (one).  The caller keeps b's VALUE <= k*M.
//   sqr(a), mul2(a, b, c, d)  a*a and a*b + c*d with one reduction; same contracts (sum of the limb
//               bound products < 2^60.6).
//   norm(a)     carry propagation: limbs < B again, value unchanged.
//   reduce(a)   value < 16 M  ->  normalized, value < 2M.
//   canon(a)    value < 16 M  ->  the unique representative in [0, M), normalized.
// The external 4 x u64 Montgomery form has radix 2^256.  NTT data stays in 2^256-form (it is only ever
// multiplied by twiddles, which the plan stores in 2^261-form); MSM bases are used as they arrive
// (2^256-form): the accumulator keeps X, Y in 2^261-form and ZZ, ZZZ in 2^266-form so that every
// product lands in the right form (ec29.hpp), and the bucket sums are mapped back once per task.
#pragma once
#include "fp256.hpp"

namespace uzk {

#include "fp29_consts.inc"

struct L29 {
    uint32_t l[9];
};

#if defined(__HIPCC__)

#include "mul29_gfx950.inc"   // l29_mul_asm / l29_sqr_asm / l29_mul2_asm: the column MADs as single asm chains

template <class C>
struct Field29 {
    using Cfg = C;
    static constexpr uint32_t MASK = (1u << 29) - 1;

    __device__ __forceinline__ static L29 zero() { L29 r; for (int i = 0; i < 9; ++i) r.l[i] = 0; return r; }
    __device__ __forceinline__ static L29 constant(const uint32_t (&c)[9]) { L29 r; for (int i = 0; i < 9; ++i) r.l[i] = c[i]; return r; }

    // 8 x 32-bit words -> 9 x 29-bit limbs (pure repacking, value unchanged)
    __device__ __forceinline__ static L29 from_fp(const Fp& a) {
        L29 r;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int bit = 29 * k, w = bit >> 5, s = bit & 31;
            uint32_t v = a.v[w] >> s;
            if (s > 3) v |= a.v[w + 1] << (32 - s);     // the limb straddles two words
            r.l[k] = v & MASK;
        }
        r.l[8] = a.v[7] >> 8;
        return r;
    }
    // 8 x 32-bit words -> 9 limbs of the value TIMES 32 (re-limbing at bit offset -5): a canonical wire element a = x 2^256 enters
    // the 2^261-form as 32 a = x 2^261 exactly, at the cost of from_fp -- no conversion product.  Normalized, value < 32 * (value of a).
    __device__ __forceinline__ static L29 from_fp_x32(const Fp& a) {
        L29 r;
        r.l[0] = (a.v[0] << 5) & MASK;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const int bit = 29 * k - 5, w = bit >> 5, s = bit & 31;
            uint32_t v = a.v[w] >> s;
            if (s > 3) v |= a.v[w + 1] << (32 - s);     // the limb straddles two words (w + 1 <= 7 for k <= 7)
            r.l[k] = v & MASK;
        }
        r.l[8] = a.v[7] >> 3;
        return r;
    }
    // The way back: value / 2^S mod M as 8 x 32-bit words, by EXACT division -- add the multiple k M (k < 2^S) that clears the low
    // S bits, shift.  S = 5 maps the 2^261-form to the wire's 2^256-form, S = 10 the 2^266-form; ~90 instructions against a product
    // by a constant plus canon (~300).  a: limbs < 2^32, value < V M with (V + 2^S) / 2^S <= 2: the result is < 2 M (and < 2^256);
    // the caller's canon (one conditional subtraction) makes it canonical.
    template <int S>
    __device__ __forceinline__ static Fp to_fp_div(const L29& a) {
        static_assert(S >= 1 && S <= 10, "k M must stay below 2^10 M");
        const uint32_t k = (a.l[0] * C::INV) & ((1u << S) - 1);        // a + k M = 0 mod 2^S  (INV = -M^-1 mod 2^29)
        uint32_t t[9];
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc += (uint64_t)a.l[i] + (uint64_t)k * C::M[i];
            t[i] = i < 8 ? ((uint32_t)acc & MASK) : (uint32_t)acc;
            acc >>= 29;
        }
        Fp r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int bit = 32 * j + S, i = bit / 29, o = bit - 29 * i;
            uint64_t w = (uint64_t)t[i] >> o;
            if (i + 1 < 9) w |= (uint64_t)t[i + 1] << (29 - o);
            if (58 - o < 32 && i + 2 < 9) w |= (uint64_t)t[i + 2] << (58 - o);
            r.v[j] = (uint32_t)w;
        }
        return r;
    }
    // normalized limbs, value < 2^256 -> 8 x 32-bit words
    __device__ __forceinline__ static Fp to_fp(const L29& a) {
        Fp r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int bit = 32 * j, k = bit / 29, o = bit - 29 * k;
            uint64_t t = (uint64_t)a.l[k] >> o;
            t |= (uint64_t)a.l[k + 1] << (29 - o);
            if (58 - o < 32 && k + 2 < 9) t |= (uint64_t)a.l[k + 2] << (58 - o);
            r.v[j] = (uint32_t)t;
        }
        return r;
    }

    __device__ __forceinline__ static L29 add(const L29& a, const L29& b) {
        L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    // a - b + OFF (OFF = 4M, 8M or 12M with borrowed limbs)
    template <int K>
    __device__ __forceinline__ static L29 sub(const L29& a, const L29& b) {
        static_assert(K == 4 || K == 8 || K == 12, "offsets 4M, 8M and 12M are provided");
        L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i] + (K == 4 ? C::OFF4[i] : K == 8 ? C::OFF8[i] : C::OFF12[i]);
        return r;
    }
    __device__ __forceinline__ static L29 norm(const L29& a) {
        L29 r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t t = a.l[i] + c;
            r.l[i] = t & MASK;
            c = t >> 29;
        }
        r.l[8] = a.l[8] + c;
        return r;
    }
    // One parallel carry step: every limb keeps its low 29 bits and receives its lower neighbour's excess.  Limbs < 2^32 in,
    // limbs < 2^29 + 2^3 out (top limb: the rest), value unchanged -- enough for the products' limb contract, and unlike
    // norm() not a serial chain through the eight limbs (what a lone wave pays for in the latency-bound quad operations).
    __device__ __forceinline__ static L29 norm1(const L29& a) {
        L29 r;
        r.l[0] = a.l[0] & MASK;
#pragma unroll
        for (int i = 1; i < 8; ++i) r.l[i] = (a.l[i] & MASK) + (a.l[i - 1] >> 29);
        r.l[8] = a.l[8] + (a.l[7] >> 29);
        return r;
    }
    // Montgomery product, radix 2^261 (contract in the header comment)
    __device__ __forceinline__ static L29 mul(const L29& a, const L29& b) { return l29_mul_asm<C>(a, b); }
    __device__ __forceinline__ static L29 sqr(const L29& a) { return l29_sqr_asm<C>(a); }
    __device__ __forceinline__ static L29 mul2(const L29& a, const L29& b, const L29& c, const L29& d) { return l29_mul2_asm<C>(a, b, c, d); }
    // x * w mod M for a PRECOMPUTED w (a twiddle) and its companion wq = wq_of(w): the plain product, no Montgomery factor,
    // no quotient digits (mul29_gfx950.inc: 143 MADs + 35 shifts / masks against 162 + 43).  x: value < 2^261, limbs < 2^31.5;
    // w, wq normalized.  Result normalized, value < 3M.
    __device__ __forceinline__ static L29 mulc(const L29& x, const L29& w, const L29& wq) { return l29_mulc_asm<C>(x, w, wq); }
    // the same with a wave-uniform w / wq held in scalar registers (made uniform by uniform())
    __device__ __forceinline__ static L29 mulcs(const L29& x, const L29& w, const L29& wq) { return l29_mulcs_asm<C>(x, w, wq); }
    __device__ __forceinline__ static L29 uniform(const L29& a) {
        L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = __builtin_amdgcn_readfirstlane(a.l[i]);
        return r;
    }
    // a * b mod 2^261 (the low nine columns), normalized operands
    __device__ __forceinline__ static L29 mullo(const L29& a, const L29& b) {
        uint64_t acc = 0;
        L29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
            r.l[k] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        return r;
    }
    // floor(w 2^261 / M) for a canonical plain value w: w 2^261 - rho is a multiple of M (rho = w 2^261 mod M), so the quotient
    // is that difference times M^-1 modulo 2^261 -- one Montgomery product, one canon, one low-half product.
    __device__ __forceinline__ static L29 wq_of(const L29& w) { return mullo(canon(mul(w, constant(C::R522))), constant(C::NEGINV261)); }
    // the same three in plain C++ (reference for the generated assembly; KAT ops compare them)
    __device__ __forceinline__ static L29 mul_cpp(const L29& a, const L29& b) {
        uint64_t acc = 0;
        uint32_t m[9];
        L29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            m[k] = ((uint32_t)acc * C::INV) & MASK;
            acc += (uint64_t)m[k] * C::M[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // a*a*2^-261: the 36 cross products are taken once against the doubled operand (45 + 81 MADs
    // instead of 162).  Same contract as mul(a, a); limbs of a must also be < 2^31.
    __device__ __forceinline__ static L29 sqr_cpp(const L29& a) {
        uint32_t d[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
        uint64_t acc = 0;
        uint32_t m[9];
        L29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            m[k] = ((uint32_t)acc * C::INV) & MASK;
            acc += (uint64_t)m[k] * C::M[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; 2 * i < k; ++i) acc += (uint64_t)a.l[i] * d[k - i];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // (a*b + c*d) * 2^-261 with ONE reduction: max(a_i) max(b_j) + max(c_i) max(d_j) < 2^60.6;
    // result normalized, value < M (1 + (Va Vb + Vc Vd) / 169).
    __device__ __forceinline__ static L29 mul2_cpp(const L29& a, const L29& b, const L29& c, const L29& d) {
        uint64_t acc = 0;
        uint32_t m[9];
        L29 r;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = 0; i <= k; ++i) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
            for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            m[k] = ((uint32_t)acc * C::INV) & MASK;
            acc += (uint64_t)m[k] * C::M[0];
            acc >>= 29;
        }
#pragma unroll
        for (int k = 9; k < 17; ++k) {
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
            for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C::M[k - i];
            r.l[k - 9] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        r.l[8] = (uint32_t)acc;
        return r;
    }
    // a - b + OFF for any of the offset constants (the caller matches OFF to b's limb / value bounds)
    __device__ __forceinline__ static L29 sub_off(const L29& a, const L29& b, const uint32_t (&off)[9]) {
        L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i] + off[i];
        return r;
    }
    __device__ __forceinline__ static bool all_zero(const L29& a) {
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) z |= a.l[i];
        return z == 0;
    }

    // normalized a, value < 3M: is it 0 mod M?
    __device__ __forceinline__ static bool is_zero_mod_small(const L29& a) {
        uint32_t z = 0, m1 = 0, m2 = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { z |= a.l[i]; m1 |= a.l[i] ^ C::M[i]; m2 |= a.l[i] ^ C::M2[i]; }
        return z == 0 || m1 == 0 || m2 == 0;
    }
    // a: limbs < 2^32 - 2^3, value < 16 M  ->  normalized, value < 2M, same residue.
    // r = a - q*M with q = floor(value / M) or one less, estimated from the top limb.
    __device__ __forceinline__ static L29 reduce(const L29& a_in) {
        const L29 a = norm(a_in);
        // value >> 232 = l[8] (the normalized low limbs contribute < 1)
        const uint32_t q = (uint32_t)(((uint64_t)a.l[8] * C::MU) >> 32);
        L29 r;
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t t = (int64_t)a.l[i] - (int64_t)((uint64_t)q * C::M[i]) + c;
            r.l[i] = (uint32_t)t & MASK;
            c = t >> 29;                       // arithmetic shift: floor division
        }
        r.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)((uint64_t)q * C::M[8]) + c);
        return r;
    }
    // a: limbs < 2^31.5, value < 16 M  ->  normalized, value < 3M, same residue: a + q (2^261 - M) with the multiple of 2^261
    // dropped at the top limb.  q comes from the RAW top limb (the unpropagated carries of the low limbs add < 8 to it, nothing
    // against M >> 232 = 2^21.8): q in {Q - 2, Q - 1, Q} for Q = floor(value / M).  One chain of 18 MADs with the carry pass
    // folded in -- 37 instructions against reduce()'s ~65 (norm() first, then a signed chain).
    __device__ __forceinline__ static L29 reduce3(const L29& a) {
        const uint32_t q = __umulhi(a.l[8], C::MU);
        uint64_t acc = 0;
        L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            asm("v_mad_u64_u32 %0, vcc, %1, 1, %0\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0" : "+v"(acc) : "v"(a.l[i]), "v"(q), "s"(C::MC[i]) : "vcc");
            r.l[i] = (uint32_t)acc & MASK;
            acc >>= 29;
        }
        return r;
    }
    // a as for reduce()  ->  the unique representative in [0, M), normalized.
    __device__ __forceinline__ static L29 canon(const L29& a_in) {
        L29 r = reduce(a_in);
        L29 d;
        int32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int32_t t = (int32_t)r.l[i] - (int32_t)C::M[i] + br;
            d.l[i] = (uint32_t)t & MASK;
            br = t >> 29;
        }
        const int32_t top = (int32_t)r.l[8] - (int32_t)C::M[8] + br;
        d.l[8] = (uint32_t)top;
        const bool ge = top >= 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = ge ? d.l[i] : r.l[i];
        return r;
    }
    // map between the external 2^256-form and the internal 2^261-form (one product each)
    __device__ __forceinline__ static L29 to_261(const L29& a256) { return mul(a256, constant(C::R266)); }
    __device__ __forceinline__ static L29 to_256(const L29& a261) { return mul(a261, constant(C::R256)); }
};

using Fq29 = Field29<Fq29Cfg>;
using Fr29 = Field29<Fr29Cfg>;

#endif   // __HIPCC__

}  // namespace uzk
