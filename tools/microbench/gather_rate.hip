// Random 64-byte gather throughput vs table size (is msm_accumulate's point gather near a memory limit?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) P64 { uint4 a, b, c, d; };
__global__ __launch_bounds__(256) void gather64(const P64* __restrict__ tab, uint64_t mask, uint4* __restrict__ out, int per_lane) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < per_lane; ++i) {
        s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
        P64 p = tab[s & mask];
        acc.x ^= p.a.x ^ p.b.y ^ p.c.z ^ p.d.w; acc.y += p.a.y + p.d.x;
    }
    out[t] = acc;
}
int main() {
    uint4* out; hipMalloc(&out, (size_t)(1u << 22) * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lg : {20, 22, 24, 26, 28}) {          // entries: 64 MiB .. 16 GiB
        uint64_t entries = 1ull << lg;
        P64* tab; if (hipMalloc(&tab, entries * 64) != hipSuccess) { printf("alloc fail\n"); return 1; }
        hipMemset(tab, 1, entries * 64);
        for (int threads_lg : {18, 20}) {
            const int threads = 1 << threads_lg, per_lane = (1 << 26) / threads;
            gather64<<<threads / 256, 256>>>(tab, entries - 1, out, per_lane); hipDeviceSynchronize();
            hipEventRecord(e0); gather64<<<threads / 256, 256>>>(tab, entries - 1, out, per_lane); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("table %6.0f MiB  threads 2^%d: %.3f ms  %.2f G gathers/s  %.2f TB/s\n", entries * 64.0 / (1 << 20), threads_lg, ms, (1 << 26) / (ms * 1e-3) / 1e9, (double)(1ull << 32) / (ms * 1e-3) / 1e12);
        }
        hipFree(tab);
    }
}
