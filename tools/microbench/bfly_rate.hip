// Cost of one radix-4 NTT butterfly with its three twiddle products, per representation
// (8 x 32-bit relaxed Montgomery with the assembly product vs 9 x 29-bit lazy-carry limbs),
// isolated from memory: data in registers, twiddles from a 256-entry L1-resident table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fp29.hpp"
using namespace uzk;
using F9 = Fr29;
constexpr int ITERS = 128;

__device__ __forceinline__ void bf2_rx(Fp& a, Fp& b) { Fp s = Fr::add_rx(a, b); b = Fr::sub_rx(a, b); a = s; }
__device__ __forceinline__ void bf2_l(L29& a, L29& b) { L29 s = F9::add(a, b); b = F9::sub<4>(a, b); a = s; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(uint32_t* io, const Fp* tw, const L29* tw29) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    if constexpr (MODE == 0) {            // fp256 butterfly
        Fp x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) x[j].v[i] = io[(t * 4 + j) * 9 + i] & (i == 7 ? 0x0fffffffu : ~0u);
        const Fp w4 = tw[64];
        for (int it = 0; it < ITERS; ++it) {
            bf2_rx(x[0], x[2]); bf2_rx(x[1], x[3]); x[3] = Fr::mul_rx(x[3], w4);
            bf2_rx(x[0], x[1]); bf2_rx(x[2], x[3]);
            Fp tt = x[1]; x[1] = x[2]; x[2] = tt;
            _Pragma("unroll") for (int s = 1; s < 4; ++s) x[s] = Fr::mul_rx(x[s], tw[(lane * s + it) & 255]);
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) io[(t * 4 + j) * 9 + i] = x[j].v[i];
    } else if constexpr (MODE == 1 || MODE == 2) {   // L29 butterfly, twiddles repacked from Fp (1) or stored as L29 (2)
        L29 x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) x[j].l[i] = io[(t * 4 + j) * 9 + i] & (i == 8 ? 0x3fffffu : 0x1fffffffu);
        const L29 w4 = F9::from_fp(tw[64]);
        for (int it = 0; it < ITERS; ++it) {
            bf2_l(x[0], x[2]); bf2_l(x[1], x[3]); x[3] = F9::mul(x[3], w4);
            bf2_l(x[0], x[1]); bf2_l(x[2], x[3]);
            L29 tt = x[1]; x[1] = x[2]; x[2] = tt;
            x[0] = F9::reduce(x[0]);
            _Pragma("unroll") for (int s = 1; s < 4; ++s) {
                const int e = (lane * s + it) & 255;
                x[s] = F9::mul(x[s], MODE == 1 ? F9::from_fp(tw[e]) : tw29[e]);
            }
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) io[(t * 4 + j) * 9 + i] = x[j].l[i];
    } else if constexpr (MODE == 3) {     // 4 independent L29 product chains (Fr constants)
        L29 x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) x[j].l[i] = io[(t * 4 + j) * 9 + i] & (i == 8 ? 0x3fffffu : 0x1fffffffu);
        const L29 w4 = F9::from_fp(tw[64]);
        for (int it = 0; it < ITERS; ++it)
            _Pragma("unroll") for (int s = 0; s < 4; ++s) x[s] = F9::mul(x[s], w4);
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) io[(t * 4 + j) * 9 + i] = x[j].l[i];
    } else if constexpr (MODE == 4) {     // 4 independent fp256 product chains
        Fp x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) x[j].v[i] = io[(t * 4 + j) * 9 + i] & (i == 7 ? 0x0fffffffu : ~0u);
        const Fp w4 = tw[64];
        for (int it = 0; it < ITERS; ++it)
            _Pragma("unroll") for (int s = 0; s < 4; ++s) x[s] = Fr::mul_rx(x[s], w4);
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) io[(t * 4 + j) * 9 + i] = x[j].v[i];
    } else if constexpr (MODE == 5) {     // L29 add/sub/reduce only
        L29 x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) x[j].l[i] = io[(t * 4 + j) * 9 + i] & (i == 8 ? 0x3fffffu : 0x1fffffffu);
        for (int it = 0; it < ITERS; ++it) {
            bf2_l(x[0], x[2]); bf2_l(x[1], x[3]);
            _Pragma("unroll") for (int s = 0; s < 4; ++s) x[s] = F9::reduce(x[s]);
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 9; ++i) io[(t * 4 + j) * 9 + i] = x[j].l[i];
    } else if constexpr (MODE == 6) {     // fp256 add/sub only (4 per iteration)
        Fp x[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) x[j].v[i] = io[(t * 4 + j) * 9 + i] & (i == 7 ? 0x0fffffffu : ~0u);
        for (int it = 0; it < ITERS; ++it) { bf2_rx(x[0], x[2]); bf2_rx(x[1], x[3]); }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) io[(t * 4 + j) * 9 + i] = x[j].v[i];
    }
}
template <int MODE>
void run(const char* name, const Fp* tw, const L29* tw29) {
    const int wps = 4, blocks = 256 * wps;
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 36 * 4); hipMemset(d, 0x15, (size_t)blocks * 256 * 36 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, tw, tw29); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, tw, tw29); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-28s %.3f ms   %.0f cycles(@2.4GHz) per wave-iteration per SIMD\n", name, best, best * 1e-3 * 2.4e9 / ((double)wps * ITERS));
    hipFree(d);
}
int main() {
    Fp* tw; L29* tw29;
    hipMalloc(&tw, 256 * sizeof(Fp)); hipMemset(tw, 0x11, 256 * sizeof(Fp));
    hipMalloc(&tw29, 256 * sizeof(L29)); hipMemset(tw29, 0x11, 256 * sizeof(L29));
    run<0>("fp256 radix4+3tw", tw, tw29);
    run<1>("l29 radix4+3tw (tw Fp)", tw, tw29);
    run<2>("l29 radix4+3tw (tw L29)", tw, tw29);
    run<3>("l29 4 products", tw, tw29);
    run<4>("fp256 4 products", tw, tw29);
    run<5>("l29 4 add/sub + 4 reduce", tw, tw29);
    run<6>("fp256 4 add/sub", tw, tw29);
}
