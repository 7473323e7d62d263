// Prototype: 9 x 29-bit-limb Montgomery product (R = 2^261) with carry-free 64-bit column sums, in
// plain C++ (every MAC is one v_mad_u64_u32, no carry handling), against the 8 x 32-bit assembly
// FIPS product.  Question: is the extra 27 % of MACs cheaper than 128 v_addc + VCC chains?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.hpp"
using namespace uzk;
__device__ __constant__ uint32_t kM29[9];
__device__ __forceinline__ L29 mul29(const L29& a, const L29& b, const uint32_t (&M)[9], uint32_t inv) {
    constexpr uint32_t MASK = (1u << 29) - 1;
    uint64_t acc = 0;
    uint32_t m[9];
    L29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * M[k - i];
        m[k] = ((uint32_t)acc * inv) & MASK;
        acc += (uint64_t)m[k] * M[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * M[k - i];
        r.l[k - 9] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ void madv(uint64_t& acc, uint32_t a, uint32_t b) {
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mads(uint64_t& acc, uint32_t a, uint32_t b) {
    uint64_t c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "s"(b));
}
__device__ __forceinline__ L29 mul29a(const L29& a, const L29& b, const uint32_t (&M)[9], uint32_t inv) {
    constexpr uint32_t MASK = (1u << 29) - 1;
    uint64_t acc = 0;
    uint32_t m[9];
    L29 r;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) madv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) mads(acc, m[i], M[k - i]);
        m[k] = ((uint32_t)acc * inv) & MASK;
        mads(acc, m[k], M[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) madv(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 8; i < 9; ++i) mads(acc, m[i], M[k - i]);
        r.l[k - 9] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
constexpr int ITERS = 512;
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* io, uint32_t inv) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M[9] = {0x187cfd47u, 0x10460b6cu, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu};
    if constexpr (MODE == 0) {
        L29 x, y;
        for (int i = 0; i < 9; ++i) { x.l[i] = io[t * 9 + i] & 0x1fffffff; y.l[i] = (io[t * 9 + i] >> 3) & 0x1fffffff; }
        for (int i = 0; i < ITERS; ++i) x = mul29(x, y, M, inv);
        for (int i = 0; i < 9; ++i) io[t * 9 + i] = x.l[i];
    } else if constexpr (MODE == 2) {
        L29 x, y;
        for (int i = 0; i < 9; ++i) { x.l[i] = io[t * 9 + i] & 0x1fffffff; y.l[i] = (io[t * 9 + i] >> 3) & 0x1fffffff; }
        for (int i = 0; i < ITERS; ++i) x = Fq29::mul(x, y);
        for (int i = 0; i < 9; ++i) io[t * 9 + i] = x.l[i];
    } else if constexpr (MODE == 3) {
        // the constant-operand product (fp29.hpp mulc): 143 MADs, no quotient digits
        L29 x, y, yq;
        for (int i = 0; i < 9; ++i) { x.l[i] = io[t * 9 + i] & 0x1fffffff; y.l[i] = (io[t * 9 + i] >> 3) & 0x1fffffff; yq.l[i] = (io[t * 9 + i] >> 2) & 0x1fffffff; }
        for (int i = 0; i < ITERS; ++i) x = Fq29::mulc(x, y, yq);
        for (int i = 0; i < 9; ++i) io[t * 9 + i] = x.l[i];
    } else {
        Fp x, y;
        for (int i = 0; i < 8; ++i) { x.v[i] = io[t * 9 + i]; y.v[i] = io[t * 9 + i] >> 3; }
        x.v[7] &= 0x0fffffff; y.v[7] &= 0x0fffffff;
        for (int i = 0; i < ITERS; ++i) x = Fq::mul_rx(x, y);
        for (int i = 0; i < 8; ++i) io[t * 9 + i] = x.v[i];
    }
}
template <int MODE>
void run(const char* name, int wps) {
    int blocks = 256 * wps;
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 9 * 4); hipMemset(d, 0x5a, (size_t)blocks * 256 * 9 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0x24866389u & 0x1fffffff); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 0x04866389u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    double ops = (double)blocks * 256 * ITERS;
    printf("%-10s waves/SIMD=%d  %.3f ms  %.3e op/s  %.0f cycles(@2.4GHz)/wave-op/SIMD\n", name, wps, best, ops / (best * 1e-3), best * 1e-3 * 2.4e9 / ((double)wps * ITERS));
    hipFree(d);
}
int main() { for (int w : {1, 2, 3, 4, 8}) { run<2>("l29_montgomery", w); run<3>("l29_mulc_shoup", w); run<1>("fips32_rx", w); } }
