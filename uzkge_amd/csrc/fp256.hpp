// 256-bit Montgomery field arithmetic for BN254 (Fq and Fr), shared by host and device code.
//
// Representation: 8 x 32-bit little-endian limbs, Montgomery form with R = 2^256 -- the same
// bytes as the 4 x u64 LE limbs of ark-ff's Fp<MontBackend<_,4>,4> that cross the C ABI
// (include/uzkge_gpu.h), so loads/stores need no conversion.
//
// gfx950 notes (tools/microbench/int_rates.hip, measured): v_mad_u64_u32 issues at half the
// v_add_u32 rate (~4 vs ~2 cycles per wave64), v_fma_f64 at the same half rate, so 32-bit-limb
// integer MADs beat any FP64-splitting scheme; a MAD whose 65th bit is consumed is ~5x slower,
// so every product-accumulate below is arranged to provably fit 64 bits.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UZK_HD __host__ __device__ __forceinline__
#else
#define UZK_HD inline
#endif

namespace uzk {

struct alignas(16) Fp {
    uint32_t v[8];
};

struct FqCfg {
    static constexpr uint32_t M[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                       0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t INV = 0xe4866389u;   // -M^-1 mod 2^32
    static constexpr uint32_t M2[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u,
                                       0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};   // 2M
};
struct FrCfg {
    static constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                       0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr uint32_t M2[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                                       0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};   // 2M
};

#if defined(__HIP_DEVICE_COMPILE__)
template <class C>
__device__ __forceinline__ Fp mont_mul_fips(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ Fp fp_add_asm(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ Fp fp_sub_asm(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ Fp mont_mul_fips_relaxed(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ Fp fp_add_2m_asm(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ Fp fp_sub_2m_asm(const Fp& a, const Fp& b);
template <class C>
__device__ __forceinline__ void fp_reduce_once_asm(Fp& r);
#endif

template <class C>
struct Field {
    UZK_HD static Fp zero() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = 0; return r; }
    UZK_HD static Fp one() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = C::R1[i]; return r; }
    UZK_HD static Fp modulus() { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = C::M[i]; return r; }
    UZK_HD static bool is_zero(const Fp& a) {
        uint32_t o = 0;
        for (int i = 0; i < 8; ++i) o |= a.v[i];
        return o == 0;
    }
    UZK_HD static bool eq(const Fp& a, const Fp& b) {
        uint32_t o = 0;
        for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }
    // r = a - M if a >= M else a   (a < 2M)
    UZK_HD static Fp reduce_once(const Fp& a) {
        Fp d; uint64_t br = 0;
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] - C::M[i] - br;
            d.v[i] = (uint32_t)t; br = (t >> 32) & 1;
        }
        Fp r;
        for (int i = 0; i < 8; ++i) r.v[i] = br ? a.v[i] : d.v[i];
        return r;
    }
    UZK_HD static Fp add(const Fp& a, const Fp& b) {   // moduli < 2^254: no carry out of limb 7
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UZK_PORTABLE_MUL)
        return fp_add_asm<C>(a, b);
#else
        return add_portable(a, b);
#endif
    }
    UZK_HD static Fp add_portable(const Fp& a, const Fp& b) {
        Fp s; uint64_t c = 0;
        for (int i = 0; i < 8; ++i) { c += (uint64_t)a.v[i] + b.v[i]; s.v[i] = (uint32_t)c; c >>= 32; }
        return reduce_once(s);
    }
    UZK_HD static Fp sub(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UZK_PORTABLE_MUL)
        return fp_sub_asm<C>(a, b);
#else
        return sub_portable(a, b);
#endif
    }
    UZK_HD static Fp sub_portable(const Fp& a, const Fp& b) {
        Fp d; uint64_t br = 0;
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
            d.v[i] = (uint32_t)t; br = (t >> 32) & 1;
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)br;
        uint64_t c = 0;
        for (int i = 0; i < 8; ++i) { c += (uint64_t)d.v[i] + (C::M[i] & mask); d.v[i] = (uint32_t)c; c >>= 32; }
        return d;
    }
    UZK_HD static Fp neg(const Fp& a) {
        Fp d; uint64_t br = 0;
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)C::M[i] - a.v[i] - br;
            d.v[i] = (uint32_t)t; br = (t >> 32) & 1;
        }
        bool z = is_zero(a);
        for (int i = 0; i < 8; ++i) d.v[i] = z ? 0u : d.v[i];
        return d;
    }
    UZK_HD static Fp dbl(const Fp& a) { return add(a, a); }

    // Montgomery product a*b*2^-256 mod M.  CIOS over 32-bit words; every step is
    // x*y + z + w with 32-bit z,w, which never exceeds 2^64-1 (no 65th bit anywhere).
    UZK_HD static Fp mul(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UZK_PORTABLE_MUL)
        return mont_mul_fips<C>(a, b);     // inline-asm FIPS, mont_mul_gfx950.inc
#else
        return mul_portable(a, b);
#endif
    }
    // Portable CIOS (host code, and the device cross-check of the assembly version).
    UZK_HD static Fp mul_portable(const Fp& a, const Fp& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
        return mul_host64(a, b);           // host: 4 x 64-bit words, ~4x the 32-bit rate
#else
        return mul_cios32(a, b);
#endif
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    // Host-only CIOS over 64-bit words (same bytes: 8 x u32 LE == 4 x u64 LE).
    static inline Fp mul_host64(const Fp& a, const Fp& b) {
        typedef unsigned __int128 u128;
        uint64_t x[4], y[4], m[4];
        for (int i = 0; i < 4; ++i) {
            x[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
            y[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
            m[i] = (uint64_t)C::M[2 * i] | ((uint64_t)C::M[2 * i + 1] << 32);
        }
        // -M^-1 mod 2^64 from the 32-bit constant by one Newton step: inv64 = inv32 * (2 + M0 * inv32)
        const uint64_t i32 = C::INV;
        const uint64_t inv = i32 * (2 + m[0] * i32);
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)x[j] * y[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t q = t[0] * inv;
            c = (u128)q * m[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; ++j) { c += (u128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fp r;
        for (int i = 0; i < 4; ++i) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
        return reduce_once(r);
    }
#endif
    UZK_HD static Fp mul_cios32(const Fp& a, const Fp& b) {
        uint32_t t[9];
        for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
                t[j] = (uint32_t)c; c >>= 32;
            }
            // M < 2^254 keeps t[8] + c within 32 bits (top word of t never exceeds 2*M's top word)
            uint32_t t8 = t[8] + (uint32_t)c;
            uint32_t m = t[0] * C::INV;
            c = (uint64_t)m * C::M[0] + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                c = (uint64_t)m * C::M[j] + t[j] + c;
                t[j - 1] = (uint32_t)c; c >>= 32;
            }
            c += t8;
            t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
        }
        Fp r;
        for (int i = 0; i < 8; ++i) r.v[i] = t[i];
        return reduce_once(r);     // result < 2M and M < 2^254 so t[8] == 0 here
    }
    UZK_HD static Fp sqr(const Fp& a) { return mul(a, a); }
#if defined(__HIP_DEVICE_COMPILE__)
    // Relaxed domain [0, 2M) (device hot loops only): the product skips its final conditional
    // subtraction -- inputs < 2M give (ab + mM)/2^256 < 1.76 M -- and add/sub wrap at 2M.
    __device__ __forceinline__ static Fp mul_rx(const Fp& a, const Fp& b) { return mont_mul_fips_relaxed<C>(a, b); }
    __device__ __forceinline__ static Fp sub_rx(const Fp& a, const Fp& b) { return fp_sub_2m_asm<C>(a, b); }
    __device__ __forceinline__ static Fp add_rx(const Fp& a, const Fp& b) { return fp_add_2m_asm<C>(a, b); }
    __device__ __forceinline__ static Fp dbl_rx(const Fp& a) { return fp_add_2m_asm<C>(a, a); }
    __device__ __forceinline__ static bool is_zero_rx(const Fp& a) {      // a in [0, 2M): zero mod M <=> a in {0, M}
        uint32_t z = 0, m = 0;
        for (int i = 0; i < 8; ++i) { z |= a.v[i]; m |= a.v[i] ^ C::M[i]; }
        return z == 0 || m == 0;
    }
    __device__ __forceinline__ static Fp canon(const Fp& a) { Fp r = a; fp_reduce_once_asm<C>(r); return r; }
#elif defined(__HIPCC__)
    // host pass of a HIP translation unit: device code that calls these is parsed, never emitted
    __device__ static Fp mul_rx(const Fp& a, const Fp& b);
    __device__ static Fp sub_rx(const Fp& a, const Fp& b);
    __device__ static Fp add_rx(const Fp& a, const Fp& b);
    __device__ static Fp dbl_rx(const Fp& a);
    __device__ static bool is_zero_rx(const Fp& a);
    __device__ static Fp canon(const Fp& a);
#endif
    UZK_HD static Fp from_mont(const Fp& a) {
        Fp o = zero(); o.v[0] = 1;
        return mul(a, o);
    }
    UZK_HD static Fp to_mont(const Fp& a) {
        Fp r2; for (int i = 0; i < 8; ++i) r2.v[i] = C::R2[i];
        return mul(a, r2);
    }
};

using Fq = Field<FqCfg>;
using Fr = Field<FrCfg>;

#if defined(__HIP_DEVICE_COMPILE__)
#include "mont_mul_gfx950.inc"
#endif

}  // namespace uzk
