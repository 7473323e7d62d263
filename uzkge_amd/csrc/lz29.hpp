// Lazy 29-bit-limb arithmetic with the bounds in the TYPE (round 6).  fp29.hpp states every operation's contract in prose and the
// two hot loops (ec29.hpp, ntt.hip) check theirs line by line in comments; the prover's polynomial kernels (the quotient: ~150
// products over ~60 loaded values, 18 terms) are too long for that, so here an element carries its bounds as template
// parameters and every operation static_asserts its contract:
//
//     Lz<F, K, V>     every limb BUT THE TOP ONE < K * (2^29 + 2^6)  (K = 1: normalized, or the output of one parallel carry step; the
//                     top limb holds the rest of the value, < V * 2^29 / 169 + K * 2^29: the products' checks account for it),
//                     value < V * M; the value is x * 2^261 mod M (2^261-form) unless stated otherwise.
//
// How values get here for free: a canonical element of the wire format (8 x 32-bit words, Montgomery radix 2^256: a = x 2^256)
// is re-limbed at bit offset -5, which yields 32 a = x 2^261 exactly -- the 2^261-form of x with value < 32 M and normalized limbs,
// at the cost of the plain re-limbing (ld).  Products shrink the bound again (value < M (1 + Va Vb / 169)).  Leaving: one
// product with a constant in plain 2^256-form (e.g. the quotient's 1 / Z_H factor, ldp) lands in 2^256-form; canon() gives the
// canonical words.  Results are exact field elements: the same bytes as the 8 x 32-bit code's.
#pragma once
#include "fp29.hpp"

namespace uzk {

#if defined(__HIPCC__)

// T * M with every low limb raised by B * 2^29, borrowed from the limb above: a - b + OFF never underflows limb-wise when
// b's limbs are < B * 2^29 (+ slack) and b's value < T * M ... (T - 1) * M suffices for the top limb, see sub().
template <class C, int B, int T>
struct Off29 {
    struct Arr { uint32_t l[9]; };
    static constexpr Arr make() {
        int64_t t[9] = {};
        uint64_t carry = 0;
        for (int i = 0; i < 9; ++i) {
            const uint64_t p = (uint64_t)C::M[i] * (uint64_t)T + carry;
            t[i] = i < 8 ? (int64_t)(p & ((1u << 29) - 1)) : (int64_t)p;
            carry = p >> 29;
        }
        for (int i = 0; i < 8; ++i) { t[i] += (int64_t)B << 29; t[i + 1] -= B; }
        Arr r{};
        for (int i = 0; i < 9; ++i) r.l[i] = (uint32_t)t[i];
        return r;
    }
    static constexpr bool ok() {
        int64_t t[9] = {};
        uint64_t carry = 0;
        for (int i = 0; i < 9; ++i) {
            const uint64_t p = (uint64_t)C::M[i] * (uint64_t)T + carry;
            t[i] = i < 8 ? (int64_t)(p & ((1u << 29) - 1)) : (int64_t)p;
            carry = p >> 29;
        }
        for (int i = 0; i < 8; ++i) { t[i] += (int64_t)B << 29; t[i + 1] -= B; }
        for (int i = 0; i < 8; ++i) if (t[i] < ((int64_t)B << 29) - B || t[i] >= ((int64_t)(B + 1) << 29)) return false;
        return t[8] >= 0 && t[8] < (1ll << 31);
    }
    static constexpr Arr value = make();
    static_assert(ok(), "offset constant out of range");
};

// the 8 x 32-bit configuration of the same field (for the final conditional subtraction on wire words)
template <class C> struct WireCfg;
template <> struct WireCfg<Fq29Cfg> { using type = FqCfg; };
template <> struct WireCfg<Fr29Cfg> { using type = FrCfg; };

template <class F, int K, int V>
struct Lz {
    L29 v;
    static constexpr int limb_k = K, val_v = V;
    static_assert(K >= 1 && K <= 7, "limbs must stay below 2^32");
};

template <class F>
struct LzOps {
    using C = typename F::Cfg;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    template <int K, int V> using E = Lz<F, K, V>;

    // canonical wire element a = x 2^256  ->  x 2^261 = 32 a, re-limbed: normalized, value < 32 M
    __device__ __forceinline__ static E<1, 32> ld(const Fp& a) { E<1, 32> r; r.v = F::from_fp_x32(a); return r; }
    // the same words taken as they are: a constant c in 2^256-form; mul(x in 2^261-form, ldp(c)) is x c in 2^256-form
    __device__ __forceinline__ static E<1, 1> ldp(const Fp& a) { E<1, 1> r; r.v = F::from_fp(a); return r; }
    __device__ __forceinline__ static E<1, 1> one() { E<1, 1> r; r.v = F::constant(C::ONE261); return r; }
    __device__ __forceinline__ static E<1, 1> zero() { E<1, 1> r; r.v = F::zero(); return r; }

    template <int Ka, int Va, int Kb, int Vb>
    __device__ __forceinline__ static E<Ka + Kb, Va + Vb> add(const E<Ka, Va>& a, const E<Kb, Vb>& b) {
        E<Ka + Kb, Va + Vb> r;
        r.v = F::add(a.v, b.v);
        return r;
    }
    // a - b + (Vb + 1) M, limb-wise with pre-borrowed offset limbs in [(Kb + 1) 2^29 - .., (Kb + 2) 2^29)
    template <int Ka, int Va, int Kb, int Vb>
    __device__ __forceinline__ static E<Ka + Kb + 2, Va + Vb + 1> sub(const E<Ka, Va>& a, const E<Kb, Vb>& b) {
        using O = Off29<C, Kb + 1, Vb + 1>;
        E<Ka + Kb + 2, Va + Vb + 1> r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v.l[i] = a.v.l[i] - b.v.l[i] + O::value.l[i];
        return r;
    }
    // a looser statement of the same element (loop-carried values keep one type)
    template <int K2, int V2, int K, int V>
    __device__ __forceinline__ static E<K2, V2> relax(const E<K, V>& a) {
        static_assert(K <= K2 && V <= V2, "relax() only loosens bounds");
        E<K2, V2> r;
        r.v = a.v;
        return r;
    }
    // UNCHECKED restatement of the bounds, for values whose bound rests on an argument the types cannot see (a running sum over a
    // run-time number of terms that the host has limited): every use states that argument in a comment
    template <int K2, int V2, int K, int V>
    __device__ __forceinline__ static E<K2, V2> assume(const E<K, V>& a) {
        E<K2, V2> r;
        r.v = a.v;
        return r;
    }
    // one parallel carry step: limbs < 2^32 in, < 2^29 + 2^3 out (the top limb takes the rest)
    template <int K, int V>
    __device__ __forceinline__ static E<1, V> norm(const E<K, V>& a) {
        E<1, V> r;
        r.v = F::norm1(a.v);
        return r;
    }
    static constexpr int prod_v(int va, int vb) { return 1 + (va * vb + 168) / 169; }
    // The TOP limb holds whatever the value has above 2^232: < V M / 2^232 = V * 2^29 / 169.9, i.e. kt(V) units of 2^29 -- more than K
    // for large V.  A column of a product holds at most ONE term with a's top limb, ONE with b's, seven others and nine reduction
    // terms (< 2^58 each): the 64-bit accumulator holds it when  kt(Va) Kb + kt(Vb) Ka + 7 Ka Kb + 9  <  64  (units of 2^58).
    static constexpr int kt(int v) { return 1 + v / 169; }
    static constexpr bool cols_fit(int ka, int va, int kb, int vb) {
        const int ta = kt(va) > ka ? kt(va) : ka, tb = kt(vb) > kb ? kt(vb) : kb;
        return ta * kb + tb * ka + 7 * ka * kb + 9 < 64;
    }
    template <int Ka, int Va, int Kb, int Vb>
    __device__ __forceinline__ static E<1, prod_v(Va, Vb)> mul(const E<Ka, Va>& a, const E<Kb, Vb>& b) {
        static_assert(Ka * Kb <= 6, "product of the limb bounds exceeds 2^60.6: carry one operand first (norm)");
        static_assert(cols_fit(Ka, Va, Kb, Vb), "a column of the product can overflow 64 bits (top limbs of large values)");
        static_assert(Va * Vb < 169 * 512, "value bound of a product");
        E<1, prod_v(Va, Vb)> r;
        r.v = F::mul(a.v, b.v);
        return r;
    }
    template <int K, int V>
    __device__ __forceinline__ static E<1, prod_v(V, V)> sqr(const E<K, V>& a) {
        static_assert(K <= 2, "squaring doubles the limbs: they must be below 2^30");
        // (a column of the squaring: <= 4 doubled cross terms + one square; with the top limb in one of them)
        static_assert(2 * (kt(V) > K ? kt(V) : K) * K + 6 * K * K + (kt(V) > K ? kt(V) : K) * (kt(V) > K ? kt(V) : K) + 9 < 64, "a column of the squaring can overflow 64 bits");
        E<1, prod_v(V, V)> r;
        r.v = F::sqr(a.v);
        return r;
    }
    // a b + c d with one reduction
    template <int Ka, int Va, int Kb, int Vb, int Kc, int Vc, int Kd, int Vd>
    __device__ __forceinline__ static E<1, 1 + (Va * Vb + Vc * Vd + 168) / 169> mul2(const E<Ka, Va>& a, const E<Kb, Vb>& b, const E<Kc, Vc>& c, const E<Kd, Vd>& d) {
        static_assert(Ka * Kb + Kc * Kd <= 6, "dual product: limb bounds");
        static_assert((kt(Va) > Ka ? kt(Va) : Ka) * Kb + (kt(Vb) > Kb ? kt(Vb) : Kb) * Ka + 7 * Ka * Kb +
                      (kt(Vc) > Kc ? kt(Vc) : Kc) * Kd + (kt(Vd) > Kd ? kt(Vd) : Kd) * Kc + 7 * Kc * Kd + 9 < 64, "a column of the dual product can overflow 64 bits");
        E<1, 1 + (Va * Vb + Vc * Vd + 168) / 169> r;
        r.v = F::mul2(a.v, b.v, c.v, d.v);
        return r;
    }
    // the canonical 8 x 32-bit words of a value < 16 M (whatever form it is in)
    template <int K, int V>
    __device__ __forceinline__ static Fp canon(const E<K, V>& a) {
        static_assert(V <= 16, "canon() takes values below 16 M");
        return F::to_fp(F::canon(a.v));
    }
    // x 2^261 -> the canonical wire words of x 2^256: exact division by 32 and one conditional subtraction when the value is small
    // enough for that (V <= 33: (V M + 31 M) / 32 < 2 M), otherwise one product by 2^256 and canon
    template <int K, int V>
    __device__ __forceinline__ static Fp to_wire(const E<K, V>& a) {
        static_assert(K <= 6, "limb bound");
        if constexpr (V <= 33) {
            return Field<typename WireCfg<C>::type>::canon(F::template to_fp_div<5>(a.v));
        } else {
            static_assert(V <= 169 * 14, "value bound of to_wire");
            return F::to_fp(F::canon(F::to_256(a.v)));
        }
    }
};

#endif   // __HIPCC__

}  // namespace uzk
