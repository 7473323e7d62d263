// Instruction-rate microbenchmark for gfx950: which multiply primitive should 256-bit
// Montgomery arithmetic be built on?  Each kernel runs a long chain of independent
// (ILP=8) operations per lane; we report wave-instructions/cycle/SIMD-equivalent numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[ILP];
    uint32_t a = seed ^ (tid * 2654435761u), b = (seed + tid) | 1u;
    double da = 1.0 + (double)(tid & 1023) * 1e-9, db = 0.999999 + (double)(seed & 7) * 1e-9;
    double dacc[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) { acc[k] = tid + k; dacc[k] = 1.0 + k; }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if constexpr (OP == 0) {          // v_mad_u64_u32
                acc[k] = (uint64_t)(uint32_t)acc[k] * b + acc[k];
            } else if constexpr (OP == 1) {   // v_mul_lo_u32
                uint32_t x = (uint32_t)acc[k]; x = x * b; acc[k] = x;
            } else if constexpr (OP == 2) {   // v_mul_hi_u32
                uint32_t x = (uint32_t)acc[k]; x = __umulhi(x, b) + a; acc[k] = x;
            } else if constexpr (OP == 3) {   // v_mad_u32_u24
                uint32_t x = (uint32_t)acc[k]; x = ((x & 0xffffffu) * (b & 0xffffffu)) + a; acc[k] = x;
            } else if constexpr (OP == 4) {   // v_add_co/v_addc 64-bit add
                acc[k] = acc[k] + (((uint64_t)a << 32) | b);
                asm volatile("" : "+v"(acc[k]));
            } else if constexpr (OP == 5) {   // v_fma_f64
                dacc[k] = __builtin_fma(dacc[k], db, da);
            } else if constexpr (OP == 6) {   // v_add_u32 (32-bit)
                uint32_t x = (uint32_t)acc[k]; x = x + b; asm volatile("" : "+v"(x)); acc[k] = x;
            } else if constexpr (OP == 7) {   // v_mul_hi_u32_u24
                uint32_t x = (uint32_t)acc[k]; x = (uint32_t)(((uint64_t)(x & 0xffffffu) * (b & 0xffffffu)) >> 32) + a; acc[k] = x;
            } else if constexpr (OP == 8) {   // v_mul_f64
                dacc[k] = dacc[k] * db;
            } else if constexpr (OP == 10) {  // v_lshrrev_b64 (the column shift of the 29-bit-limb product)
                asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(acc[k]));
            } else if constexpr (OP == 11) {  // v_lshl_add_u64 (64-bit add joining two column chains)
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[k]) : "v"(acc[(k + 1) % ILP]));
            } else if constexpr (OP == 12) {  // v_alignbit_b32
                uint32_t x = (uint32_t)acc[k]; asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x) : "v"(b)); acc[k] = x;
            } else if constexpr (OP == 13) {  // v_and_b32
                uint32_t x = (uint32_t)acc[k]; asm volatile("v_and_b32 %0, %1, %0" : "+v"(x) : "v"(b)); acc[k] = x;
            } else if constexpr (OP == 9) {   // v_mad_u64_u32 with carry-out consumed
                unsigned __int128 t = (unsigned __int128)((uint64_t)(uint32_t)acc[k] * b) + acc[k];
                acc[k] = (uint64_t)t + (uint64_t)(t >> 64);
            }
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) r ^= acc[k] ^ (uint64_t)__double_as_longlong(dacc[k]);
    out[tid] = r;
}

template <int OP>
int run(const char* name, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves/block=1 wave/SIMD) x waves_per_simd
    uint64_t* d; CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    rate_kernel<OP><<<blocks, 256>>>(d, 12345);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        rate_kernel<OP><<<blocks, 256>>>(d, 12345 + r);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double lane_ops = (double)blocks * 256 * ITERS * ILP;
    double wave_instr_per_simd = (double)waves_per_simd * ITERS * ILP; // per SIMD
    double cyc_at_2p4 = best * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f Tops/s  ~%.2f cycles/wave-instr/SIMD (@2.4GHz)\n",
           name, waves_per_simd, best, lane_ops / (best * 1e-3) / 1e12, cyc_at_2p4 / wave_instr_per_simd);
    CHECK(hipFree(d));
    return 0;
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mad_u64_u32", w);
        run<9>("v_mad_u64_u32+carry", w);
        run<1>("v_mul_lo_u32", w);
        run<2>("v_mul_hi_u32(+add)", w);
        run<3>("v_mad_u32_u24", w);
        run<7>("v_mul_hi_u32_u24(+add)", w);
        run<4>("add64 (add_co+addc)", w);
        run<6>("v_add_u32", w);
        run<5>("v_fma_f64", w);
        run<8>("v_mul_f64", w);
        run<10>("v_lshrrev_b64", w);
        run<11>("v_lshl_add_u64", w);
        run<12>("v_alignbit_b32", w);
        run<13>("v_and_b32", w);
    }
    return 0;
}
