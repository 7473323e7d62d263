// Latency of dependent group additions for ONE wave (and for 2 / 4 waves per SIMD): lane-serial xyzz_add against the
// four-lane xyzz_add_quad / xyzz_dbl_quad of ecquad.hpp.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../uzkge_amd/csrc quad_latency.hip -o quad_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "ecquad.hpp"
#include "ecquad29.hpp"
using namespace uzk;
__global__ void chain_lane(XYZZ* io, int iters) {
    XYZZ acc = io[0], p = io[1];
    for (int i = 0; i < iters; ++i) xyzz_add(acc, p);
    if (threadIdx.x == 0) io[2] = acc;
}
__global__ void chain_quad(XYZZ* io, int iters) {
    XYZZ acc = io[0], p = io[1];
    const uint32_t q = threadIdx.x & 3;
    for (int i = 0; i < iters; ++i) xyzz_add_quad(acc, p, q);
    if (threadIdx.x == 0) io[3] = acc;
}
__global__ void chain_dblq(XYZZ* io, int iters) {
    XYZZ acc = io[0];
    const uint32_t q = threadIdx.x & 3;
    for (int i = 0; i < iters; ++i) xyzz_dbl_quad(acc, q);
    if (threadIdx.x == 0) io[4] = acc;
}
__global__ void chain_quad29(XYZZ* io, int iters) {
    const uint32_t q = threadIdx.x & 3;
    X29 acc = x29_from_xyzz_quad(io[0], q);
    const X29 p = x29_from_xyzz_quad(io[1], q);
    for (int i = 0; i < iters; ++i) x29_add_quad(acc, p, q);
    reinterpret_cast<Fp*>(&io[5])[q] = x29_coord_to_fp(acc, q);
}
__global__ void chain_dblq29(XYZZ* io, int iters) {
    const uint32_t q = threadIdx.x & 3;
    X29 acc = x29_from_xyzz_quad(io[0], q);
    for (int i = 0; i < iters; ++i) x29_dbl_quad(acc, q);
    reinterpret_cast<Fp*>(&io[6])[q] = x29_coord_to_fp(acc, q);
}
template <typename K> static float run(K k, XYZZ* d, int iters, int waves) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;   // us per operation
}
int main() {
    XYZZ h[8] = {};
    Affine G; G.x = Fq::one(); G.y = Fq::dbl(Fq::one());
    XYZZ g1 = xyzz_from_affine(G);
    h[0] = xyzz_dbl(g1);          // 2G
    h[1] = g1;                    // G: acc runs through 3G, 4G, ...
    XYZZ* d; hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const int it = 2000;
    for (int waves : {1, 4, 8, 16})
        std::printf("waves/WG %2d: lane add %.2f us, quad add %.2f us, quad dbl %.2f us | 29-bit limbs: quad add %.2f us, quad dbl %.2f us\n", waves,
                    run(chain_lane, d, it, waves), run(chain_quad, d, it, waves), run(chain_dblq, d, it, waves), run(chain_quad29, d, it, waves),
                    run(chain_dblq29, d, it, waves));
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("lane and quad chains agree: %s; 29-bit quad chain agrees: %s\n", memcmp(&h[2], &h[3], sizeof(XYZZ)) == 0 ? "yes" : "NO",
                memcmp(&h[2], &h[5], sizeof(XYZZ)) == 0 ? "yes" : "NO");
    return 0;
}
