// Latency of ONE wave's dependent chain of modular products -- the situation of the quad additions (ecquad.hpp): folds and
// scans on a nearly idle chip.  8 x 32-bit FIPS assembly (Fq::mul) against the 9 x 29-bit product (F9<FqC29>::mul).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../uzkge_amd/csrc mul_latency.hip -o mul_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp256.hpp"
#include "fp29.hpp"
using namespace uzk;
using Q9 = Fq29;
__global__ void chain32(Fp* io, int iters) {
    Fp x = io[threadIdx.x], y = io[64 + threadIdx.x];
    for (int i = 0; i < iters; ++i) x = Fq::mul(x, y);
    io[threadIdx.x] = x;
}
__global__ void chain29(Fp* io, int iters) {
    L29 x = Q9::from_fp(io[threadIdx.x]), y = Q9::from_fp(io[64 + threadIdx.x]);
    for (int i = 0; i < iters; ++i) x = Q9::mul(x, y);
    io[threadIdx.x] = Q9::to_fp(Q9::canon(x));
}
__global__ void chain29sq(Fp* io, int iters) {
    L29 x = Q9::from_fp(io[threadIdx.x]);
    for (int i = 0; i < iters; ++i) x = Q9::sqr(x);
    io[threadIdx.x] = Q9::to_fp(Q9::canon(x));
}
template <typename K> static float run(K k, Fp* d, int iters, int waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / iters;   // ns per product
}
int main() {
    Fp* d; hipMalloc(&d, 1024 * sizeof(Fp)); hipMemset(d, 1, 1024 * sizeof(Fp));
    const int it = 20000;
    for (int waves : {1, 4, 8, 16})
        std::printf("waves/WG %2d: 8x32 FIPS %.0f ns/product, 9x29 mul %.0f ns, 9x29 sqr %.0f ns\n", waves,
                    run(chain32, d, it, waves), run(chain29, d, it, waves), run(chain29sq, d, it, waves));
    return 0;
}
