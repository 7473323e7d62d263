// Calibration of rocprofv3 FETCH_SIZE for the MSM access pattern (one 64-byte affine point per lane
// at a random index of a table much larger than the Infinity Cache), as MI355X_MICROARCH.md
// section HBM asks for access widths other than a coalesced 16 B/lane stream.
// Known bytes: GATHERS * 64.  Run under: rocprofv3 --pmc FETCH_SIZE --output-format csv -- ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct alignas(16) P64 { uint4 a, b, c, d; };
__global__ __launch_bounds__(256) void gather64(const P64* __restrict__ tab, uint32_t mask, uint4* __restrict__ out, int per_lane) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = t * 0x9E3779B97F4A7C15ull + 12345;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int i = 0; i < per_lane; ++i) {
        s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 27; s *= 0x94D049BB133111EBull; s ^= s >> 31;
        P64 p = tab[(uint32_t)s & mask];
        acc.x ^= p.a.x ^ p.b.y ^ p.c.z ^ p.d.w; acc.y += p.a.y + p.d.x;
    }
    out[t] = acc;
}
__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t & 0xFFFF] = in[t];
}
int main() {
    const uint32_t entries = 1u << 24;              // 1 GiB table
    P64* tab; hipMalloc(&tab, (size_t)entries * 64); hipMemset(tab, 1, (size_t)entries * 64);
    uint4* out; hipMalloc(&out, (size_t)(1u << 22) * 16);
    const int threads = 1 << 20, per_lane = 64;     // 2^26 gathers = 4 GiB of 64-byte reads
    gather64<<<threads / 256, 256>>>(tab, entries - 1, out, per_lane);
    hipDeviceSynchronize();
    stream16<<<(entries * 4) / 256, 256>>>(reinterpret_cast<const uint4*>(tab), out, (size_t)entries * 4);   // 1 GiB stream
    hipDeviceSynchronize();
    printf("gather64: %llu bytes expected; stream16: %llu bytes expected\n", (unsigned long long)threads * per_lane * 64ull, (unsigned long long)entries * 64ull);
    return 0;
}
