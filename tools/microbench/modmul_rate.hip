// Throughput of the 256-bit Montgomery product on gfx950: assembly FIPS vs portable CIOS,
// by waves per SIMD and independent chains per lane.  Build:
//   hipcc --offload-arch=gfx950 -O3 -I uzkge_amd/csrc -o tools/microbench/modmul_rate tools/microbench/modmul_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp256.hpp"
using namespace uzk;
constexpr int ITERS = 512;
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void k(Fp* io) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fp x[ILP], y = io[t];
#pragma unroll
    for (int j = 0; j < ILP; ++j) { x[j] = y; x[j].v[0] ^= j; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            if constexpr (MODE == 0) x[j] = Fq::mul(x[j], y);
            else if constexpr (MODE == 1) x[j] = Fq::mul_portable(x[j], y);
            else if constexpr (MODE == 2) x[j] = Fq::add(x[j], y);
            else x[j] = Fq::sub(x[j], y);
        }
    }
    Fp r = x[0];
#pragma unroll
    for (int j = 1; j < ILP; ++j) r = Fq::add(r, x[j]);
    io[t] = r;
}
template <int MODE, int ILP>
void run(const char* name, int wps) {
    int blocks = 256 * wps;
    Fp* d; hipMalloc(&d, (size_t)blocks * 256 * sizeof(Fp)); hipMemset(d, 0x11, (size_t)blocks * 256 * sizeof(Fp));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, ILP><<<blocks, 256>>>(d); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); k<MODE, ILP><<<blocks, 256>>>(d); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    double ops = (double)blocks * 256 * ITERS * ILP;
    double per_simd_wave_ops = (double)wps * ITERS * ILP;
    printf("%-14s ILP=%d waves/SIMD=%d  %.3f ms  %.3e op/s  %.0f cycles(@2.4GHz)/wave-op/SIMD\n", name, ILP, wps, best, ops / (best * 1e-3), best * 1e-3 * 2.4e9 / per_simd_wave_ops);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0, 1>("mul_asm", w); run<0, 2>("mul_asm", w); run<1, 1>("mul_portable", w);
        run<2, 1>("add", w); run<3, 1>("sub", w);
    }
}
