// Synthetic workloads generated on device (bench.py / tests): nothing is uploaded, so the 2^24
// and 2^26 configurations of BASELINE.json need no multi-GiB host arrays.
//   synth_scalars        uniform Fr elements from a SplitMix64 counter stream
//   synth_points_random  G1 points by try-and-increment on x (cofactor 1: every curve point is in G1)
//   synth_points_arith   P_i = (i+1) * Q: known discrete logs => closed-form MSM check at any size
#include "ctx.hpp"
#include "host_math.hpp"

namespace uzk {

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// 254 pseudo-random bits, reduced once below the modulus: a valid (Montgomery) representative
template <class F>
__host__ __device__ inline Fp random_fe(uint64_t seed, uint64_t ctr) {
    Fp r;
    for (int k = 0; k < 4; ++k) {
        uint64_t v = splitmix64(seed ^ splitmix64(ctr * 4 + k));
        r.v[2 * k] = (uint32_t)v;
        r.v[2 * k + 1] = (uint32_t)(v >> 32);
    }
    r.v[7] &= 0x3FFFFFFFu;
    return F::reduce_once(r);
}

__global__ __launch_bounds__(256) void synth_scalars_kernel(Fp* __restrict__ out, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = random_fe<Fr>(seed, i);
}

// The illustrative "prover-like" scalar mix of BASELINE.md section 4 / SURVEY.md 8d set (B): by a hash of the
// index 50 % zero, 20 % one, 10 % r - 1, 10 % below 2^16, 10 % uniform (the value classes the witness
// vectors of the real prover hold; the proportions are a placeholder).
__global__ __launch_bounds__(256) void synth_scalars_mix_kernel(Fp* __restrict__ out, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = splitmix64(seed ^ 0x6D69785F736574ull ^ splitmix64(i));
    const uint32_t cls = (uint32_t)(((h >> 32) * 100) >> 32);
    Fp r;
    if (cls < 50) r = Fr::zero();
    else if (cls < 70) r = Fr::one();
    else if (cls < 80) r = Fr::neg(Fr::one());
    else if (cls < 90) { Fp v = Fr::zero(); v.v[0] = (uint32_t)(h & 0xFFFF); r = Fr::to_mont(v); }
    else r = random_fe<Fr>(seed, i);
    out[i] = r;
}

// a^((p+1)/4) for the BN254 base field (p = 3 mod 4): square root candidate
__device__ inline Fp fq_sqrt_candidate(const Fp& a) {
    // (p+1)/4, little-endian 32-bit words
    const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u,
                           0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fp acc = Fq::one();
    for (int i = 253; i >= 0; --i) {
        acc = Fq::sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = Fq::mul(acc, a);
    }
    return acc;
}

__global__ __launch_bounds__(256) void synth_points_random_kernel(Affine* __restrict__ out, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fp three = Fq::add(Fq::one(), Fq::add(Fq::one(), Fq::one()));
    Affine p;
    for (uint64_t attempt = 0;; ++attempt) {
        Fp x = random_fe<Fq>(seed ^ 0xA5A5A5A5DEADBEEFull, i * 64 + attempt);
        Fp rhs = Fq::add(Fq::mul(Fq::sqr(x), x), three);
        Fp y = fq_sqrt_candidate(rhs);
        if (Fq::eq(Fq::sqr(y), rhs) && !(Fq::is_zero(x) && Fq::is_zero(y))) {
            if (splitmix64(seed + i) & 1) y = Fq::neg(y);
            p.x = x; p.y = y;
            break;
        }
        if (attempt >= 63) { p.x = Fq::zero(); p.y = Fq::zero(); break; }   // probability 2^-64
    }
    out[i] = p;
}

// Thread t produces points (t*L + 1 .. t*L + L) * Q in Jacobian-like form (X*ZZ, Y*ZZZ, ZZ),
// then normalises its run with one inversion (Montgomery's trick).
constexpr int kArithRun = 64;

__device__ inline Fp fq_inv_device(const Fp& a) {
    // a^(p-2)
    const uint32_t e[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                           0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fp acc = Fq::one();
    for (int i = 253; i >= 0; --i) {
        acc = Fq::sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = Fq::mul(acc, a);
    }
    return acc;
}

__global__ __launch_bounds__(64) void synth_points_arith_kernel(Affine* __restrict__ out, Fp* __restrict__ tmp_z,
                                                                 Fp* __restrict__ tmp_p, uint64_t n, Affine q) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * kArithRun;
    if (lo >= n) return;
    const uint64_t hi = (lo + kArithRun < n) ? lo + kArithRun : n;
    // acc = (lo + 1) * Q
    XYZZ acc = xyzz_inf();
    const uint64_t k = lo + 1;
    for (int bit = 63 - __clzll(k); bit >= 0; --bit) {
        acc = xyzz_dbl(acc);
        if ((k >> bit) & 1) xyzz_madd(acc, q, false);
    }
    Fp prod = Fq::one();
    for (uint64_t j = lo; j < hi; ++j) {
        // (i+1)*Q is never infinity for i+1 < r
        Affine xy;
        xy.x = Fq::mul(acc.x, acc.zz);
        xy.y = Fq::mul(acc.y, acc.zzz);
        out[j] = xy;
        tmp_z[j] = acc.zz;
        tmp_p[j] = prod;              // product of z's before j
        prod = Fq::mul(prod, acc.zz);
        xyzz_madd(acc, q, false);
    }
    Fp inv = fq_inv_device(prod);
    for (uint64_t j = hi; j-- > lo;) {
        Fp zi = Fq::mul(inv, tmp_p[j]);      // 1 / z_j
        inv = Fq::mul(inv, tmp_z[j]);
        Fp zi2 = Fq::sqr(zi);
        Affine xy = out[j];
        xy.x = Fq::mul(xy.x, zi2);
        xy.y = Fq::mul(xy.y, Fq::mul(zi2, zi));
        out[j] = xy;
    }
}

int synth_scalars(Ctx& c, Fp* d_scalars, size_t n, uint64_t seed) {
    if (n == 0) return UZK_OK;
    KernelScope ks(c, "synth_scalars");
    hipLaunchKernelGGL(synth_scalars_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_scalars,
                       (uint64_t)n, seed);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

int synth_scalars_mix(Ctx& c, Fp* d_scalars, size_t n, uint64_t seed) {
    if (n == 0) return UZK_OK;
    KernelScope ks(c, "synth_scalars_mix");
    hipLaunchKernelGGL(synth_scalars_mix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_scalars,
                       (uint64_t)n, seed);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

int synth_points_random(Ctx& c, Affine* d_points, size_t n, uint64_t seed) {
    if (n == 0) return UZK_OK;
    KernelScope ks(c, "synth_points_random");
    hipLaunchKernelGGL(synth_points_random_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream,
                       d_points, (uint64_t)n, seed);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

int synth_points_arith(Ctx& c, Affine* d_points, size_t n, const Fp& seed_scalar_mont) {
    if (n == 0) return UZK_OK;
    // Q = k * G on the host (one scalar multiplication)
    Fp k = Fr::from_mont(seed_scalar_mont);
    Affine g;
    g.x = Fq::one();
    g.y = Fq::add(Fq::one(), Fq::one());
    XYZZ acc = xyzz_inf();
    for (int bit = 255; bit >= 0; --bit) {
        acc = xyzz_dbl(acc);
        if ((k.v[bit >> 5] >> (bit & 31)) & 1) xyzz_madd(acc, g, false);
    }
    if (xyzz_is_inf(acc)) { set_error("synth_points_arith: seed scalar is zero"); return UZK_ERR_PARAMETER; }
    Affine q = xyzz_to_affine_host(acc);
    Fp *tz = nullptr, *tp = nullptr;
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&tz), n * sizeof(Fp)));
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&tp), n * sizeof(Fp)));
    const uint64_t threads = (n + kArithRun - 1) / kArithRun;
    {
        KernelScope ks(c, "synth_points_arith");
        hipLaunchKernelGGL(synth_points_arith_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, c.stream,
                           d_points, tz, tp, (uint64_t)n, q);
    }
    hipError_t e = hipStreamSynchronize(c.stream);
    (void)hipFree(tz);
    (void)hipFree(tp);
    UZK_HIP(e);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

}  // namespace uzk
