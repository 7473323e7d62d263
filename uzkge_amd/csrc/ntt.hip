// BN254 Fr NTT for gfx950: natural order in and out, omega_n = 5^((r-1)/n).
// Replaces EvaluationDomain::{fft,ifft} as called from
//   uzkge/src/poly_commit/field_polynomial.rs:583-597 (fft_with_domain / ifft_with_domain)
// and the coset wrappers :589-591 / :601-607; domains 2^k and 3*2^k (:554-567).
//
// Design (MI355X-first, see DESIGN.md "NTT"):
//  * n = 2^k <= 2048: one workgroup, radix-2 DIT in LDS (latency path for tiny transforms).
//  * n = 2^k >= 4096: P = ceil(k/8) Stockham autosort passes of radix R = 2^b (b in 5..8).
//    A pass needs no transposition kernel: pass p reads x[i + r*N/R] (coalesced over i) and
//    writes y[m'*S*R + sigma*S + s] (S = product of earlier radices, i = m'*S + s), so after the
//    last pass the data is in natural order.  One workgroup (512 threads, 8 waves) owns a tile of
//    T = 2048/R adjacent columns: every global access is a run of T*32 B >= 256 B.
//    Inside the tile each of the 512 threads keeps 4 elements in VGPRs (four waves per SIMD fit)
//    and runs radix-4/2 register butterflies; the sub-pass exchanges go through 64 KiB of LDS
//    (two 16-byte planes per element, conflict-free ds_read/write_b128).  Inter-pass twiddles omega_N^(S*m'*sigma) come
//    from HBM/L2-resident tables laid out in write order (pass 0's table is N entries and is
//    streamed, coalesced, exactly once; the 1/n of the inverse transform is folded into it).
//  * The kernel is bound by 256-bit modular multiplies (about 11 per element at 2^22), not by
//    HBM: no MFMA, no GEMM reshaping (north star).
//  * n = 3 * 2^k: decimation in time by 3 -> three radix-2 transforms + one combine kernel.
#include <cstring>

#include "ctx.hpp"
#include "fp29.hpp"
#include "host_math.hpp"

namespace uzk {

// ---------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------
struct NttPlan {
    uint64_t n = 0;          // power of two
    int log_n = 0;
    bool inverse = false;
    bool scaled = false;     // inverse only: fold 1/n in
    int npass = 0;
    int bits[4] = {0, 0, 0, 0};
    Fp* d_tw256 = nullptr;        // omega_256^e (direction-specific), e < 256   (n >= 4096)
    Fp* d_tw2048c = nullptr;      // two-pass plans (2^17 .. 2^21): omega_2048^e as pairs for their passes of 9 .. 11 bits
    Fp* d_tw256c = nullptr;       // the same 256 twiddles as pairs (w, floor(w 2^261 / M)) for the constant-operand product: two
                                  // sets of limb planes, the second 256 * 36 bytes behind the first (uzk_tune("ntt_mulc"))
    Fp* d_tw_pass[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t tw_count[4] = {0, 0, 0, 0};   // entries per pass table
    Fp* d_small_tw = nullptr;     // omega_n^e, e < n/2                          (n <= 2048)
    Fp scale;                     // 1/n (Montgomery) when scaled, else one
    // three-level power tables of omega (device): A[j]=w^j, B[j]=w^(1024 j), C[j]=w^(2^20 j)
    Fp* d_pow = nullptr;
};

static void host_pow_tables(const Fp& w, std::vector<Fp>& tab) {
    tab.resize(3 * 1024);
    Fp step = w;
    for (int lvl = 0; lvl < 3; ++lvl) {
        Fp cur = Fr::one();
        for (int j = 0; j < 1024; ++j) {
            tab[lvl * 1024 + j] = cur;
            cur = Fr::mul(cur, step);
        }
        step = cur;   // step^1024
    }
}

// out[idx] = w^(e(idx)) [* scale], e(idx) = (S * m' * sigma) with idx = m' * R + sigma
__global__ __launch_bounds__(256) void ntt_gen_pass_tw_kernel(Fp* __restrict__ out, uint64_t count,
                                                              int log_S, int B,
                                                              const Fp* __restrict__ pw, Fp scale,
                                                              int use_scale, int dbl_count) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    uint64_t mp = idx >> B, sg = idx & ((1u << B) - 1);
    uint64_t e = (mp * sg) << log_S;
    Fp v = Fr::mul(pw[e & 1023], pw[1024 + ((e >> 10) & 1023)]);
    v = Fr::mul(v, pw[2048 + (e >> 20)]);
    if (use_scale) v = Fr::mul(v, scale);
    for (int d = 0; d < dbl_count; ++d) v = Fr::add(v, v);   // 2^256-form -> 2^261-form
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// small transform: one workgroup, LDS radix-2 DIT
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) {
    return bits == 0 ? 0u : (__brev(x) >> (32 - bits));
}

__global__ __launch_bounds__(256) void ntt_small_kernel(const Fp* __restrict__ in, Fp* __restrict__ out,
                                                        int log_n, const Fp* __restrict__ tw,
                                                        Fp scale, int use_scale, uint64_t in_stride, uint64_t out_stride) {
    extern __shared__ uint4 lds_small[];
    Fp* s = reinterpret_cast<Fp*>(lds_small);
    const uint32_t n = 1u << log_n;
    const Fp* src = in + (uint64_t)blockIdx.x * in_stride;
    Fp* dst = out + (uint64_t)blockIdx.x * out_stride;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s[bitrev(i, log_n)] = src[i];
    __syncthreads();
    for (int lh = 0; lh < log_n; ++lh) {
        const uint32_t half = 1u << lh;
        const int shift = log_n - 1 - lh;   // twiddle stride = n / (2*half)
        for (uint32_t b = threadIdx.x; b < n / 2; b += blockDim.x) {
            uint32_t g = b >> lh, p = b & (half - 1);
            uint32_t iu = (g << (lh + 1)) + p, iv = iu + half;
            Fp u = s[iu];
            Fp v = Fr::mul(s[iv], tw[p << shift]);
            s[iu] = Fr::add(u, v);
            s[iv] = Fr::sub(u, v);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        Fp v = s[i];
        if (use_scale) v = Fr::mul(v, scale);
        dst[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// register butterflies (natural order in, natural order out)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf2(Fp& a, Fp& b) {
    Fp s = Fr::add(a, b);
    b = Fr::sub(a, b);
    a = s;
}
__device__ __forceinline__ void swap_fp(Fp& a, Fp& b) { Fp t = a; a = b; b = t; }

// size-4 DFT, w4 = w8^2.  1 multiplication.
__device__ __forceinline__ void radix4(Fp& x0, Fp& x1, Fp& x2, Fp& x3, const Fp& w4) {
    bf2(x0, x2);
    bf2(x1, x3); x3 = Fr::mul(x3, w4);
    bf2(x0, x1);
    bf2(x2, x3);
    swap_fp(x1, x2);
}

struct PassArgs {
    uint64_t batch_stride; // elements between consecutive (sub-)transforms in the scratch buffers, and the size of the fused tables (= N)
    // Elements between consecutive FULL vectors where the caller's memory is touched: the first pass's input and the last
    // pass's output (uzk_ntt_fr_batch_strided_device: a batch that reads / writes slots of a wider array, e.g. the 6n-slots
    // the prover keeps its coefficient polynomials in).  Contiguous vectors: the full vector length.
    uint64_t in_vec_stride, out_vec_stride;
    uint64_t stride;       // N / R
    int log_S;             // log2 of the product of earlier radices
    const Fp* tw256;       // omega_256^e, direction-specific
    const Fp* tw256c;      // the same as (w, wq) pairs for the constant-operand product; a pass of 9 .. 11 bits: omega_2048^e, e < 2048
    const Fp* twp;         // pass table [m'][sigma] (nullptr on the last pass)
    uint64_t twp_count;    // entries of twp (29-bit-limb kernels: tables are limb planes, see tw29_load)
    // ---- fused coset scaling and radix-3 stage (29-bit-limb kernels only; all zero = plain transform) ----
    // First pass, m3 != 0: the launch runs 3 * batch sub-transforms of size m3 of batch vectors of 3 * m3 elements
    // (blockIdx.y = 3 * vector + r).  Sub-transform r loads a = x[j], b = x[j + m3], c = x[j + 2 m3] and starts from
    //   y_j = (a + alpha[r] b + beta[r] c) * in_tw[r][j]       (decimation in frequency by 3; r = 0 without tables: a + b + c)
    // First pass, m3 == 0, in_tw[0] != null: y_j = x[j] * in_tw[0][j]  (coset pre-scaling k^j of a 2^k transform).
    // Last pass: element i' of sub-transform r goes to out[out_mul * i' + r] (out_mul = 3 with m3, else 1), times
    // out_tw[r][i'] when given (coset post-scaling of the inverse transforms).
    uint64_t m3;
    // First pass with m3 != 0: the caller's vectors hold at most in_len non-zero leading coefficients (0 = no statement).  With
    // in_len <= m3 the two upper thirds are zero BY CONTRACT: they are not read, the radix-3 combination is the element itself, and
    // elements at or beyond in_len start as zero (the prover's coset FFTs over 6n points of polynomials with n + 3 coefficients).
    uint64_t in_len;
    const Fp* in_tw[3];    // limb planes of N (or m3) entries each
    Fp alpha[3], beta[3];  // 2^261-form
    uint32_t use_ab;       // alpha / beta present (else plain sums for every r is NOT implied: r != 0 always needs them)
    const Fp* out_tw[3];
    uint32_t out_mul;      // 0 / 1: contiguous
    // Limb planes between the passes (plain power-of-two transforms of up to 2^20 elements per launch): the intermediate vectors live in the
    // two scratch buffers as three planes over plane_count = batch * N entries (limbs 0-3, 4-7, 8: 36 bytes per element, the
    // layout of the twiddle tables) instead of 8 x 32-bit words, so a pass boundary costs neither the 9 -> 8 word packing of the
    // store nor the 8 -> 9 unpacking of the next load (~47 instructions per element and boundary) for 12 % more bytes there.
    uint32_t in_planes, out_planes;
    uint64_t plane_count;
};

// ---------------------------------------------------------------------------------------------
// One Stockham pass of radix R = 2^B on the 29-bit-limb representation (fp29.hpp): data stays in 2^256-form, every
// twiddle table is in 2^261-form, so a twiddle product is one carry-free 9x9 product.  Additions and
// subtractions are 9 independent 32-bit adds; the bounds below (limb size / multiple of M) are what
// keeps every product inside its contract (limb product < 2^60.6):
//   loaded / exchanged values        limbs < 2^29     value < 2M
//   radix-4 outputs                  limbs < 2^31.5   value < 10M   -> twiddle product or reduce()
//   radix-2 outputs                  limbs < 2^31     value < 6M
// The sigma = 0 output of a butterfly has no twiddle; it is brought back with reduce() (a 9-MAD
// quotient-estimate subtraction, about a fifth of a product).
// ---------------------------------------------------------------------------------------------
using F9 = Fr29;

// Twiddle tables of the 29-bit-limb kernels are stored already repacked, as three planes over the
// table's `count` entries: limbs 0-3 (uint4), limbs 4-7 (uint4), limb 8 (u32) -- 36 bytes per entry
// instead of 32, and no shifting/masking per twiddle in the butterflies.
__device__ __forceinline__ L29 tw29_load(const Fp* base, uint64_t count, uint64_t idx) {
    const uint4* p0 = reinterpret_cast<const uint4*>(base);
    const uint4* p1 = p0 + count;
    const uint32_t* p2 = reinterpret_cast<const uint32_t*>(p1 + count);
    const uint4 a = p0[idx], b = p1[idx];
    L29 v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    v.l[8] = p2[idx];
    return v;
}
__device__ __forceinline__ void plane29_store(Fp* base, uint64_t count, uint64_t idx, const L29& v) {
    uint4* p0 = reinterpret_cast<uint4*>(base);
    uint4* p1 = p0 + count;
    uint32_t* p2 = reinterpret_cast<uint32_t*>(p1 + count);
    p0[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p1[idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p2[idx] = v.l[8];
}
__global__ __launch_bounds__(256) void ntt_repack_tw_kernel(const Fp* __restrict__ src, Fp* __restrict__ dst_base,
                                                            uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const L29 v = F9::from_fp(src[i]);
    uint4* p0 = reinterpret_cast<uint4*>(dst_base);
    uint4* p1 = p0 + count;
    uint32_t* p2 = reinterpret_cast<uint32_t*>(p1 + count);
    p0[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p1[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p2[i] = v.l[8];
}

// src[i] = w_i 2^261 mod M (canonical, 8 x 32 bits)  ->  two sets of limb planes: the plain values w_i and their companions
// floor(w_i 2^261 / M) = (w_i 2^261 mod M) * (-M^-1) mod 2^261 -- the operands of F9::mulc.
__global__ __launch_bounds__(256) void ntt_pair_tw_kernel(const Fp* __restrict__ src, Fp* __restrict__ dst_base, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const L29 rho = F9::from_fp(src[i]);
    const L29 w = F9::canon(F9::mul(rho, F9::constant(F9::Cfg::R1)));
    const L29 wq = F9::mullo(rho, F9::constant(F9::Cfg::NEGINV261));
    for (int half = 0; half < 2; ++half) {
        const L29& v = half ? wq : w;
        uint4* p0 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst_base) + (size_t)half * count * 36);
        uint4* p1 = p0 + count;
        uint32_t* p2 = reinterpret_cast<uint32_t*>(p1 + count);
        p0[i] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        p1[i] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
        p2[i] = v.l[8];
    }
}

// planes[j] = g^j * scale in 2^261-form (three-level power table of g), j < count: the coset / radix-3 input and
// output tables of the fused transforms
__global__ __launch_bounds__(256) void ntt_gen_pow_planes_kernel(Fp* __restrict__ dst_base, uint64_t count,
                                                                 const Fp* __restrict__ pw, Fp scale) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    Fp g = Fr::mul(pw[j & 1023], pw[1024 + ((j >> 10) & 1023)]);
    g = Fr::mul(g, pw[2048 + (j >> 20)]);
    g = Fr::mul(g, scale);
    for (int d = 0; d < 5; ++d) g = Fr::add(g, g);          // 2^256-form -> 2^261-form
    const L29 v = F9::from_fp(g);
    uint4* p0 = reinterpret_cast<uint4*>(dst_base);
    uint4* p1 = p0 + count;
    uint32_t* p2 = reinterpret_cast<uint32_t*>(p1 + count);
    p0[j] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    p1[j] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    p2[j] = v.l[8];
}

__device__ __forceinline__ void bf2_l(L29& a, L29& b) {
    L29 s = F9::add(a, b);
    b = F9::sub<4>(a, b);
    a = s;
}
// The radix-4 butterfly around the constant-operand product (F9::mulc: plain product, value < 3M out, value < 2^261 in).  Inputs
// < 3M (a mulc result, or a loaded element): first layer sums < 6M, differences < 7M; the second layer subtracts a sum of two
// (value < 6M, limbs < 2^30), hence the 8M offset; outputs < 12M / 14M / 10M / 11M, limbs < 2^31.4.
__device__ __forceinline__ void radix4_c(L29& x0, L29& x1, L29& x2, L29& x3, const L29& w4, const L29& w4q) {
    bf2_l(x0, x2);
    bf2_l(x1, x3); x3 = F9::mulcs(x3, w4, w4q);                   // omega_4 and its companion are wave-uniform: scalar registers
    { const L29 s = F9::add(x0, x1); x1 = F9::sub<8>(x0, x1); x0 = s; }
    bf2_l(x2, x3);
    L29 t = x1; x1 = x2; x2 = t;
}
// entry idx of a pair table of `count` twiddles: the plain value and its companion
__device__ __forceinline__ void tw29_load_pair(const Fp* base, uint64_t count, uint64_t idx, L29& w, L29& wq) {
    w = tw29_load(base, count, idx);
    wq = tw29_load(reinterpret_cast<const Fp*>(reinterpret_cast<const char*>(base) + count * 36), count, idx);
}

// two uint4 planes + one u32 plane per element slot; PL = slots per plane (tile + 64: the transposed layout needs T*(R+1))
template <int PL>
__device__ __forceinline__ void lds_put29(uint4* lds, int idx, const L29& v) {
    lds[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    lds[PL + idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    reinterpret_cast<uint32_t*>(lds + 2 * PL)[idx] = v.l[8];
}
template <int PL>
__device__ __forceinline__ L29 lds_get29(const uint4* lds, int idx) {
    const uint4 a = lds[idx], b = lds[PL + idx];
    L29 v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    v.l[8] = reinterpret_cast<const uint32_t*>(lds + 2 * PL)[idx];
    return v;
}

// TILE elements per workgroup (TILE / 4 threads): 2048 (two workgroups per CU) or 1024 (four, shorter turnover bubbles)
// The twiddles of the tile (omega_4 inside the butterflies, omega_256^e between the sub-passes -- 8.75 of the 10.75 products per
// element of a 2^22 transform) are multiplied in by the constant-operand product over the pair table a.tw256c; the pass
// twiddles and the fused coset / radix-3 tables are Montgomery products over their 2^261-form tables (streamed tables: a pair
// would double their traffic).  Both forms compute x * w mod M up to the lazy multiple of M, so they mix freely.
// Waves per SIMD: four for the first pass (114 .. 117 VGPRs).  The later passes of 6 .. 9 bits on 1024-element tiles need 139 .. 146 VGPRs:
// held to 128 (four waves) they spill 4 .. 10 words; compiled for THREE waves per SIMD they do not, and the transform is 3 % faster
// (round 6, profiles/r06_ntt_bounds.txt: 2^22 396 -> 381 us, ten transforms of 2^14 26.1 -> 25.5 us; the 2048-element tiles
// keep four: at three their 76 KB of LDS would leave one workgroup per CU).
template <int B, bool FIRST, int TILE>
__global__ __launch_bounds__(TILE / 4) __attribute__((amdgpu_waves_per_eu((!FIRST && TILE == 1024 && B >= 6 && B <= 9) ? 3 : 4))) void ntt_pass29_kernel(const Fp* __restrict__ in, Fp* __restrict__ out,
                                                             PassArgs a) {
    constexpr int R = 1 << B, T = TILE / R, Q = R / 4, TWN = B > 8 ? 2048 : 256, SH = (B > 8 ? 11 : 8) - B, NT = TILE / 4, PL = TILE + (TILE >= 1024 ? 64 : 16);
    constexpr int N4 = B / 2;
    constexpr bool TAIL2 = (B & 1) != 0;
    __shared__ uint4 lds[(9 * PL + 3) / 4];
    const int tid = threadIdx.x;
    const int col = tid % T, q = tid / T;
    // The untwiddled output of a butterfly has to become small again (it may be the next butterfly's subtrahend).  Measured and
    // not kept (round 4, profiles/r04_ab_ntt_reduce_lazy.txt): a table-driven form of this step -- q M from a 16-row LDS table, limb-wise
    // subtraction over a pre-lent M, one parallel carry step: ~45 instructions against ~90 -- pushed the kernels past 128 VGPRs; held
    // to 128 they spill, and the transform lost a third of what the constant-operand product had won (2^22: 0.947 -> 0.963 of the
    // Montgomery path's time).
    // Round 4, kept: with the constant-operand product the butterflies take inputs < 3M, so the untwiddled output only needs
    // F9::reduce3 (one MAD chain from the raw top limb, 37 instructions against ~65 of the full reduce()).
    auto shrink = [&](L29& v) { v = F9::reduce3(v); };
    // Passes of 10 / 11 bits: a tile of one or two columns reads 32 / 64 bytes of a 128-byte line; the workgroups that read the
    // rest of the line are made neighbours ON ONE XCD (workgroup ids go round the eight XCDs), so that the line is fetched into
    // one L2 once instead of into four (2^20: 111.5 -> 97.8 us, profiles/r05_ab_ntt_two_pass.txt columns 1 and 2)
    uint32_t bx = blockIdx.x;
    if constexpr (T < 4) {
        if ((gridDim.x % (8 * (4 / T))) == 0) {
            constexpr uint32_t G = 4 / T;
            const uint32_t xcd = bx & 7, k = bx >> 3;
            bx = ((k / G) * 8 + xcd) * G + (k % G);
        }
    }
    const uint64_t i0 = (uint64_t)bx * T;
    const uint64_t i = i0 + col;
    const Fp* in_base = in;
    Fp* out_base = out;
    // plain / coset transforms: the first pass reads and the last pass writes the caller's vectors (stride in_vec_stride /
    // out_vec_stride); the radix-3 forms address whole vectors through in_base / out_base below
    // limb-plane buffers are addressed by element index from the plane base: the vector's offset stays an index
    const uint64_t in_off = a.in_planes ? (uint64_t)blockIdx.y * a.batch_stride : 0, out_off = a.out_planes ? (uint64_t)blockIdx.y * a.batch_stride : 0;
    if (!a.in_planes) in += (uint64_t)blockIdx.y * ((FIRST && a.m3 == 0) ? a.in_vec_stride : a.batch_stride);
    if (!a.out_planes) out += (uint64_t)blockIdx.y * ((a.twp == nullptr && a.out_mul <= 1) ? a.out_vec_stride : a.batch_stride);
    auto load_in = [&](uint64_t idx) -> L29 { return a.in_planes ? tw29_load(in, a.plane_count, in_off + idx) : F9::from_fp(in[idx]); };
    auto store_out = [&](uint64_t idx, const L29& v) {        // v normalized, value < 2^256
        if (a.out_planes) plane29_store(out, a.plane_count, out_off + idx, v);
        else out[idx] = F9::to_fp(v);
    };

    L29 w4, w4q;
    tw29_load_pair(a.tw256c, TWN, TWN / 4, w4, w4q);
    w4 = F9::uniform(w4); w4q = F9::uniform(w4q);
    // one twiddled output of a butterfly: x * omega_256^e
    // Measured and not kept (round 5, profiles/r05_ab_ntt_split_mulc.txt): the product in two steps -- the companion wq first, the
    // quotient estimate from it, and only then the load of w, nine registers fewer alive across the product -- is 5..6 % SLOWER at
    // every size (2^22: 370.7 -> 390.4 us): the kernels no longer spill (profiles/r05_ntt_kernel_registers.txt: 104..114 VGPRs since the
    // experiment branches left them) and the second load's latency now sits in the middle of the product.
    auto twiddle = [&](L29& v, int e) { L29 w, wq; tw29_load_pair(a.tw256c, TWN, e, w, wq); v = F9::mulc(v, w, wq); };
    // The twiddle between two passes stays a Montgomery product over the 2^261-form table.  Measured and not kept (round 4,
    // profiles/r04_ab_ntt_reduce3_pairs.txt): (w, wq) pair tables for pass tables of up to 2^18 entries, i.e. the constant-operand
    // product here too -- no difference at any size (2^14 24.5 / 24.4 us, 2^22 370.0 / 375.0), the 36 instructions saved per product
    // against twice the table bytes per load.
    auto pass_twiddle = [&](const L29& v, uint64_t idx) -> L29 { return F9::mul(v, tw29_load(a.twp, a.twp_count, idx)); };
    auto butterfly = [&](L29& y0, L29& y1, L29& y2, L29& y3) { radix4_c(y0, y1, y2, y3, w4, w4q); };
    L29 x[4];
    int rows[4];

    // ---- sub-pass 0: radix 4 straight from global memory
    if constexpr (FIRST) {
        if (a.m3 != 0) {
            // decimation in frequency by 3 in front of sub-transform r of vector blockIdx.y / 3 (see PassArgs)
            const uint32_t r = blockIdx.y % 3;
            const Fp* vec = in_base + (uint64_t)(blockIdx.y / 3) * a.in_vec_stride;
            const bool tabled = a.in_tw[r] != nullptr;
            const bool short_in = a.in_len != 0 && a.in_len <= a.m3;          // wave-uniform
            L29 al, be;
            if (r != 0 && !short_in) { al = F9::from_fp(a.alpha[r]); be = F9::from_fp(a.beta[r]); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint64_t j = i + (uint64_t)(q + t * Q) * a.stride;
                if (short_in) {
                    // x_(j + m3) = x_(j + 2 m3) = 0: the combination is x_j itself (canonical: normalized, value < M)
                    if (j < a.in_len) { const L29 xa = F9::from_fp(vec[j]); x[t] = tabled ? F9::mul(xa, tw29_load(a.in_tw[r], a.m3, j)) : xa; }
                    else x[t] = F9::zero();
                    continue;
                }
                const L29 xa = F9::from_fp(vec[j]), xb = F9::from_fp(vec[j + a.m3]), xc = F9::from_fp(vec[j + 2 * a.m3]);
                L29 y;
                if (r == 0 && !a.use_ab) y = F9::add(F9::add(xa, xb), xc);                  // limbs < 3 * 2^29, value < 3M
                else if (r == 0) y = F9::add(xa, F9::mul2(xb, F9::from_fp(a.alpha[0]), xc, F9::from_fp(a.beta[0])));
                else y = F9::add(xa, F9::mul2(xb, al, xc, be));                            // limbs < 2^30, value < 2.1M
                x[t] = tabled ? F9::mul(y, tw29_load(a.in_tw[r], a.m3, j)) : F9::reduce(y);  // normalized, value < 2M
            }
        } else if (a.in_tw[0] != nullptr) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint64_t j = i + (uint64_t)(q + t * Q) * a.stride;
                x[t] = F9::mul(F9::from_fp(in[j]), tw29_load(a.in_tw[0], a.batch_stride, j));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) x[t] = F9::from_fp(in[i + (uint64_t)(q + t * Q) * a.stride]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = load_in(i + (uint64_t)(q + t * Q) * a.stride);
    }
    butterfly(x[0], x[1], x[2], x[3]);
    if constexpr (N4 > 1 || TAIL2) {
        shrink(x[0]);
#pragma unroll
        for (int s = 1; s < 4; ++s) twiddle(x[s], (q * s) << SH);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rows[s] = q * 4 + s;

    // ---- radix-4 sub-passes 1 .. N4-1
#pragma unroll
    for (int k = 1; k < N4; ++k) {
        const int S = 1 << (2 * k);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) lds_put29<PL>(lds, rows[s] * T + col, x[s]);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = lds_get29<PL>(lds, (q + t * Q) * T + col);
        butterfly(x[0], x[1], x[2], x[3]);
        const int mp = q >> (2 * k), sl = q & (S - 1);
        const bool more = (R >> (2 * k + 2)) > 1;
        if (more) {
            shrink(x[0]);
#pragma unroll
            for (int s = 1; s < 4; ++s) twiddle(x[s], (S * mp * s) << SH);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) rows[s] = (mp << (2 * k + 2)) + s * S + sl;
    }
    // ---- final radix-2 sub-pass (B odd)
    if constexpr (TAIL2) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) lds_put29<PL>(lds, rows[s] * T + col, x[s]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ib = q + u * Q;
            x[2 * u] = lds_get29<PL>(lds, ib * T + col);
            x[2 * u + 1] = lds_get29<PL>(lds, (ib + R / 2) * T + col);
            bf2_l(x[2 * u], x[2 * u + 1]);
            rows[2 * u] = ib;
            rows[2 * u + 1] = ib + R / 2;
        }
    }

    // ---- write back (lazy limbs: < 2^31.5, value < 10M)
    if constexpr (FIRST) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_put29<PL>(lds, col * (R + 1) + rows[j], x[j]);
        __syncthreads();
        const uint64_t base = i0 * R;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * NT;
            const int ce = e / R, re = e % R;
            L29 v = lds_get29<PL>(lds, ce * (R + 1) + re);
            v = pass_twiddle(v, base + e);                                // normalized, < 2M: fits 8 words
            store_out(base + e, v);
        }
    } else {
        const uint64_t mp = i >> a.log_S, sp = i & ((1ull << a.log_S) - 1);
        const uint64_t base = (mp << (a.log_S + B)) + sp;
        if (a.twp == nullptr && (a.out_mul > 1 || a.out_tw[0] != nullptr)) {
            // last pass of a fused transform: interleaved store (radix-3 outputs 3 i' + r) and / or coset post-scaling
            const uint32_t mul = a.out_mul > 1 ? a.out_mul : 1u;
            const uint32_t r = mul > 1 ? blockIdx.y % mul : 0u;
            Fp* dst = mul > 1 ? out_base + (uint64_t)(blockIdx.y / mul) * a.out_vec_stride : out;
            const Fp* otw = a.out_tw[r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t pos = base + ((uint64_t)rows[j] << a.log_S);
                L29 v = x[j];
                if (otw != nullptr) v = F9::mul(v, tw29_load(otw, a.batch_stride, pos));   // lazy v: limbs < 2^31.5, value < 10M
                v = F9::canon(v);
                dst[(uint64_t)mul * pos + r] = F9::to_fp(v);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            L29 v = x[j];
            if (a.twp != nullptr) v = pass_twiddle(v, (mp << B) + rows[j]);
            else v = F9::canon(v);                 // last pass: back to [0, M)
            store_out(base + ((uint64_t)rows[j] << a.log_S), v);
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// helpers for coset scaling and the 3 * 2^k domain
// ---------------------------------------------------------------------------------------------
// out[j] = in[j] * g^j, powers from a three-level table of g.
__global__ __launch_bounds__(256) void ntt_scale_pows_kernel(const Fp* __restrict__ in, Fp* __restrict__ out,
                                                             uint64_t n, const Fp* __restrict__ pw) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    in += (uint64_t)blockIdx.y * n;
    out += (uint64_t)blockIdx.y * n;
    Fp g = Fr::mul(pw[j & 1023], pw[1024 + ((j >> 10) & 1023)]);
    g = Fr::mul(g, pw[2048 + (j >> 20)]);
    out[j] = Fr::mul(in[j], g);
}
// sub[k*m + j] = in[3j + k]
__global__ __launch_bounds__(256) void ntt_decimate3_kernel(const Fp* __restrict__ in, Fp* __restrict__ sub, uint64_t m) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * m) return;
    in += (uint64_t)blockIdx.y * 3 * m;
    sub += (uint64_t)blockIdx.y * 3 * m;
    uint64_t j = idx / 3, k = idx % 3;
    sub[k * m + j] = in[idx];
}
// out[i] = (X0[i%m] + w^i X1[i%m] + w^2i X2[i%m]) * scale
__global__ __launch_bounds__(256) void ntt_combine3_kernel(const Fp* __restrict__ sub, Fp* __restrict__ out, uint64_t m,
                                                           const Fp* __restrict__ pw, Fp scale, int use_scale) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * m) return;
    sub += (uint64_t)blockIdx.y * 3 * m;
    out += (uint64_t)blockIdx.y * 3 * m;
    uint64_t j = i % m;
    Fp w = Fr::mul(pw[i & 1023], pw[1024 + ((i >> 10) & 1023)]);
    w = Fr::mul(w, pw[2048 + (i >> 20)]);
    Fp w2 = Fr::sqr(w);
    Fp v = Fr::add(sub[j], Fr::add(Fr::mul(w, sub[m + j]), Fr::mul(w2, sub[2 * m + j])));
    if (use_scale) v = Fr::mul(v, scale);
    out[i] = v;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int upload(Fp** dst, const std::vector<Fp>& src, hipStream_t st) {
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(Fp)));
    UZK_HIP(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(Fp), hipMemcpyHostToDevice, st));
    UZK_HIP(hipStreamSynchronize(st));
    return UZK_OK;
}

static int get_plan(Ctx& c, uint64_t n, bool inverse, bool scaled, NttPlan** out) {
    int k_of_n = 0;
    while ((1ull << k_of_n) < n) ++k_of_n;
    // 2^17 .. 2^21 in TWO passes of 9 .. 11 bits (round 5; uzk_tune("ntt_two_pass", 0): the three passes of 5 .. 8 bits they had)
    const bool two_pass = c.tune_ntt_two_pass && k_of_n >= 17 && k_of_n <= 21;
    const uint64_t key = (n << 3) | (inverse ? 1u : 0u) | (scaled ? 2u : 0u) | (two_pass ? 4u : 0u);
    auto it = c.ntt_plans.find(key);
    if (it != c.ntt_plans.end()) { *out = it->second; return UZK_OK; }
    NttPlan* p = new NttPlan();
    p->n = n;
    p->inverse = inverse;
    p->scaled = scaled;
    int k = 0;
    while ((1ull << k) < n) ++k;
    p->log_n = k;
    Fp w = fr_root_of_unity(n);
    if (inverse) w = fr_inv(w);
    p->scale = scaled ? fr_inv(fr_from_u64(n)) : Fr::one();
    if (k <= 11) {
        std::vector<Fp> tw(n >= 2 ? n / 2 : 1);
        Fp cur = Fr::one();
        for (size_t e = 0; e < tw.size(); ++e) { tw[e] = cur; cur = Fr::mul(cur, w); }
        UZK_TRY(upload(&p->d_small_tw, tw, c.stream));
    } else {
        // Split k into P = ceil(k/8) radices 2^b, b in 5..8, minimising the products per element:
        // a pass of radix 2^b costs floor(b/2) radix-4 sub-passes of one product per element each,
        // except that the last sub-pass of an even b carries no twiddles (a quarter of a product).
        p->npass = (k + 7) / 8;
        // Two passes instead of three for 2^17 .. 2^21 (round 5): a pass costs its products plus ~1.7
        // product-equivalents per element of load / re-limbing / exchanges / store, so one pass fewer pays as long as the tile
        // keeps >= 2 columns (64-byte accesses) in at least one of the two passes: interleaved A/B, identical bytes
        // (profiles/r05_ab_ntt_two_pass.txt, _shapes.txt): 2^17 36.6 -> 32.5 us, 2^19 62.5 -> 57.8, 2^20 104.3 -> 97.8, 2^21 192.4 ->
        // 181.9; eight transforms of 2^20: 701 -> 637.  2^22 = 11 + 11 has ONE column per tile in both passes and LOSES: 379.5 ->
        // 406.5 us (462 without the XCD mapping of the pass kernel): it keeps 8 + 8 + 6.
        if (two_pass) { p->npass = 2; p->bits[0] = (k + 1) / 2; p->bits[1] = k / 2; }
        else {
            int best_cost = 1 << 30, cur[4] = {0, 0, 0, 0};
            const int P = p->npass;
            // enumerate non-increasing b_0 >= b_1 >= ... (order does not change the cost), smallest
            // b_0 first so that ties go to the more balanced split
            auto cost_of = [&](const int* b) {
                int c4 = 0;
                for (int j = 0; j < P; ++j) c4 += 4 * (b[j] / 2) - ((b[j] & 1) ? 0 : 3);
                return c4;
            };
            for (cur[0] = 5; cur[0] <= 8; ++cur[0])
                for (cur[1] = (P > 1 ? cur[0] : 0); cur[1] >= (P > 1 ? 5 : 0); --cur[1])
                    for (cur[2] = (P > 2 ? cur[1] : 0); cur[2] >= (P > 2 ? 5 : 0); --cur[2])
                        for (cur[3] = (P > 3 ? cur[2] : 0); cur[3] >= (P > 3 ? 5 : 0); --cur[3]) {
                            if (cur[0] + cur[1] + cur[2] + cur[3] != k) continue;
                            const int cst = cost_of(cur);
                            if (cst < best_cost) { best_cost = cst; for (int j = 0; j < 4; ++j) p->bits[j] = cur[j]; }
                        }
            if (best_cost == (1 << 30)) { delete p; set_error("ntt: no radix split for 2^%d", k); return UZK_ERR_FFT; }
        }
        std::vector<Fp> pw;
        host_pow_tables(w, pw);
        UZK_TRY(upload(&p->d_pow, pw, c.stream));
        // omega_256^e = w^(e * n/256)
        if (two_pass) {                         // passes of more than 8 bits: omega_2048^e as (w, wq) pairs
            std::vector<Fp> t2k(2048);
            Fp w2k = f_pow_u64<Fr>(w, n / 2048), cur2 = Fr::one();
            for (uint32_t e = 0; e < 2048; ++e) { t2k[e] = cur2; cur2 = Fr::mul(cur2, w2k); }
            for (auto& t : t2k)
                for (int d = 0; d < 5; ++d) t = Fr::add(t, t);
            Fp* d_t2k = nullptr;
            UZK_TRY(upload(&d_t2k, t2k, c.stream));
            UZK_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_tw2048c), 2 * (size_t)2048 * 36 + 64));
            hipLaunchKernelGGL(ntt_pair_tw_kernel, dim3(8), dim3(256), 0, c.stream, d_t2k, p->d_tw2048c, (uint64_t)2048);
            UZK_HIP(hipStreamSynchronize(c.stream));
            UZK_HIP(hipFree(d_t2k));
        }
        const uint32_t TW = 256;
        std::vector<Fp> t256(TW);
        Fp w256 = f_pow_u64<Fr>(w, n / TW), cur = Fr::one();
        for (uint32_t e = 0; e < TW; ++e) { t256[e] = cur; cur = Fr::mul(cur, w256); }
        for (auto& t : t256)
            for (int d = 0; d < 5; ++d) t = Fr::add(t, t);     // 2^261-form
        UZK_TRY(upload(&p->d_tw256, t256, c.stream));
        // 29-bit-limb kernels read their twiddles as limb planes (36 B per entry): repack a table in place of
        // the 8 x 32-bit one
        auto to_planes = [&](Fp** tab, uint64_t count) -> int {
            Fp* planes = nullptr;
            UZK_HIP(hipMalloc(reinterpret_cast<void**>(&planes), count * 36 + 64));
            hipLaunchKernelGGL(ntt_repack_tw_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c.stream, *tab,
                               planes, count);
            UZK_HIP(hipStreamSynchronize(c.stream));
            UZK_HIP(hipFree(*tab));
            *tab = planes;
            return UZK_OK;
        };
        // the 256 tile twiddles once more as (w, floor(w 2^261 / M)) pairs for the constant-operand product
        UZK_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_tw256c), 2 * (size_t)TW * 36 + 64));
        hipLaunchKernelGGL(ntt_pair_tw_kernel, dim3(TW / 256), dim3(256), 0, c.stream, p->d_tw256, p->d_tw256c, (uint64_t)TW);
        UZK_TRY(to_planes(&p->d_tw256, TW));
        int log_S = 0;
        for (int j = 0; j + 1 < p->npass; ++j) {
            const uint64_t count = n >> log_S;   // (N / (S R)) * R
            UZK_HIP(hipMalloc(reinterpret_cast<void**>(&p->d_tw_pass[j]), count * sizeof(Fp)));
            const bool fold = scaled && j == 0;
            {
                KernelScope ks(c, "ntt_gen_pass_tw");
                hipLaunchKernelGGL(ntt_gen_pass_tw_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                                   c.stream, p->d_tw_pass[j], count, log_S, p->bits[j], p->d_pow, p->scale,
                                   fold ? 1 : 0, 5);
            }
            p->tw_count[j] = count;
            UZK_TRY(to_planes(&p->d_tw_pass[j], count));
            log_S += p->bits[j];
        }
        UZK_HIP(hipStreamSynchronize(c.stream));
    }
    c.ntt_plans[key] = p;
    *out = p;
    return UZK_OK;
}

static void release_ntt_caches(Ctx& c);
void ntt_free_plans(Ctx& c) {
    for (auto& kv : c.ntt_plans) {
        NttPlan* p = kv.second;
        if (p->d_tw256) (void)hipFree(p->d_tw256);
        if (p->d_tw256c) (void)hipFree(p->d_tw256c);
        if (p->d_tw2048c) (void)hipFree(p->d_tw2048c);
        if (p->d_small_tw) (void)hipFree(p->d_small_tw);
        if (p->d_pow) (void)hipFree(p->d_pow);
        for (auto* t : p->d_tw_pass) if (t) (void)hipFree(t);
        delete p;
    }
    c.ntt_plans.clear();
    release_ntt_caches(c);
}

template <int B>
static void launch_pass(Ctx& c, bool first, const Fp* in, Fp* out, const PassArgs& a, uint64_t n, uint32_t batch) {
    constexpr int R = 1 << B;
    // Tiles of 1024 elements (256 threads, four workgroups per CU) up to 2^24 elements in all: twice as many workgroups
    // on a chip the launch does not fill, shorter load / compute / store phases -- 25..32 % faster from 2^12 to 2^18 and
    // for the prover's shapes (10 x 2^14: 36 -> 26 us, 98304 on a coset: 48 -> 34 us); re-measured at the end of round 3
    // (profiles/r03f_ab_ntt_tile.txt, two boxes): 2^22 413 -> 391 / 392 -> 380 us, 2^23 851 -> 813 / 828 -> 809,
    // 2^24 1711 -> 1682 / 1682 -> 1673, 2^20 and below equal -- the finer turnover of four workgroups per CU wins there too.
    // uzk_tune("ntt_tile", 1024 | 2048) forces a tile (tests: both instantiations at every size).
    const bool small_tile = c.tune_ntt_tile == 1024 || (c.tune_ntt_tile != 2048 && n * (uint64_t)batch <= (1ull << 24));
    KernelScope ks(c, first ? "ntt_pass_first" : "ntt_pass");
    if constexpr (B <= 9) {
    if (small_tile) {
        const unsigned g2 = (unsigned)((n / R) / (1024 / R));
        if (first) hipLaunchKernelGGL((ntt_pass29_kernel<B, true, 1024>), dim3(g2, batch), dim3(256), 0, c.stream, in, out, a);
        else hipLaunchKernelGGL((ntt_pass29_kernel<B, false, 1024>), dim3(g2, batch), dim3(256), 0, c.stream, in, out, a);
        return;
    }
    }
    {
        const unsigned grid = (unsigned)((n / R) / (2048 / R));
        if (first) hipLaunchKernelGGL((ntt_pass29_kernel<B, true, 2048>), dim3(grid, batch), dim3(512), 0, c.stream, in, out, a);
        else hipLaunchKernelGGL((ntt_pass29_kernel<B, false, 2048>), dim3(grid, batch), dim3(512), 0, c.stream, in, out, a);
    }
}

// power-of-two transform d_in -> d_out (may alias)
// `batch` contiguous vectors of n elements each, d_in -> d_out (may alias)
// `fx` (optional): the fused stages of PassArgs -- first-pass input tables / radix-3 load, last-pass interleaved store
// and output tables; batch then counts sub-transforms (3 per vector when fx->m3 != 0).
// in_vs / out_vs: elements between consecutive FULL vectors of the caller's input / output (0 = contiguous).
static int ntt_pow2(Ctx& c, const Fp* d_in, Fp* d_out, uint64_t n, bool inverse, bool scaled, uint32_t batch,
                    const PassArgs* fx = nullptr, uint64_t in_vs = 0, uint64_t out_vs = 0) {
    NttPlan* p = nullptr;
    UZK_TRY(get_plan(c, n, inverse, scaled, &p));
    const uint64_t n_full = (fx && fx->m3) ? 3 * n : n;              // radix-3 forms: batch counts sub-transforms of n = m3
    if (in_vs == 0) in_vs = n_full;
    if (out_vs == 0) out_vs = n_full;
    if (p->log_n <= 11) {
        KernelScope ks(c, "ntt_small");
        const size_t shmem = (size_t)n * sizeof(Fp);
        hipLaunchKernelGGL(ntt_small_kernel, dim3(batch), dim3(256), shmem, c.stream, d_in, d_out, p->log_n,
                           p->d_small_tw, p->scale, scaled ? 1 : 0, in_vs, out_vs);
        UZK_HIP(hipGetLastError());
        return UZK_OK;
    }
    // limb planes between the passes: plain transforms on the 29-bit-limb kernels with at least one intermediate vector
    // (measured, profiles/r04_ab_ntt_planes.txt: 3..5 % faster up to 2^20 elements per launch, 5 % SLOWER from 2^21 on -- three
    // plane streams and 12 % more bytes where the passes start to feel HBM)
    const bool planes = fx == nullptr && p->npass >= 2 && n * (uint64_t)batch <= (1ull << 20);
    const size_t bytes = (size_t)n * batch * (planes ? 36 : sizeof(Fp)) + 64;
    UZK_TRY(c.ntt_scratch[0].reserve(bytes));
    Fp* s0 = c.ntt_scratch[0].as<Fp>();
    Fp* s1 = nullptr;
    const bool in_place = (d_in == d_out);
    // A strided output holds other data between its vectors: intermediate passes then stay in the two scratch buffers.
    const bool out_dense = out_vs == n_full && !planes;             // ... and so do limb-plane intermediates (36 bytes per element)
    if ((in_place && (p->npass & 1)) || (!out_dense && p->npass > 2)) {
        UZK_TRY(c.ntt_scratch[1].reserve(bytes));
        s1 = c.ntt_scratch[1].as<Fp>();
    }
    // buffer schedule: the last pass writes d_out; earlier passes alternate scratch/d_out and
    // never write the buffer they read.
    const Fp* src = d_in;
    int log_S = 0;
    for (int j = 0; j < p->npass; ++j) {
        const int remaining = p->npass - 1 - j;
        Fp* dst = (remaining % 2 == 0) ? d_out : s0;
        if (!out_dense && remaining > 0) dst = (j % 2 == 0) ? s0 : s1;      // scratch only until the last pass
        if (dst == src) dst = s1;   // only when in_place and npass is odd, at j == 0
        PassArgs a{};
        if (fx != nullptr) {
            if (j == 0) { a.m3 = fx->m3; a.in_len = fx->in_len; a.use_ab = fx->use_ab; for (int k = 0; k < 3; ++k) { a.in_tw[k] = fx->in_tw[k]; a.alpha[k] = fx->alpha[k]; a.beta[k] = fx->beta[k]; } }
            if (j == p->npass - 1) { a.out_mul = fx->out_mul; for (int k = 0; k < 3; ++k) a.out_tw[k] = fx->out_tw[k]; }
        }
        a.batch_stride = n;
        a.in_vec_stride = in_vs;
        a.out_vec_stride = out_vs;
        a.stride = n >> p->bits[j];
        a.log_S = log_S;
        a.tw256 = p->d_tw256;
        a.tw256c = p->bits[j] > 8 ? p->d_tw2048c : p->d_tw256c;
        a.twp = p->d_tw_pass[j];
        a.twp_count = p->tw_count[j];
        a.in_planes = planes && j > 0;
        a.out_planes = planes && remaining > 0;
        a.plane_count = (uint64_t)n * batch;
        const bool first = (j == 0);
        switch (p->bits[j]) {
            case 5: launch_pass<5>(c, first, src, dst, a, n, batch); break;
            case 6: launch_pass<6>(c, first, src, dst, a, n, batch); break;
            case 7: launch_pass<7>(c, first, src, dst, a, n, batch); break;
            case 8: launch_pass<8>(c, first, src, dst, a, n, batch); break;
            case 9: launch_pass<9>(c, first, src, dst, a, n, batch); break;         // two-pass plans (2^17 .. 2^21)
            case 10: launch_pass<10>(c, first, src, dst, a, n, batch); break;
            case 11: launch_pass<11>(c, first, src, dst, a, n, batch); break;
            default: set_error("ntt: bad radix bits %d", p->bits[j]); return UZK_ERR_FFT;
        }
        UZK_HIP(hipGetLastError());
        // after a diverted first pass (dst == s1) continue the normal alternation
        src = dst;
        log_S += p->bits[j];
    }
    return UZK_OK;
}

// three-level power table of an arbitrary element, cached by value (a proof alternates between k and 1/k and
// between the forward and inverse mixed-radix roots: all of them stay resident, a hit costs nothing)
static void release_fused(Ctx& c);
static void release_ntt_caches(Ctx& c) {
    for (auto& pc : c.pow_cache) { pc.buf.release(); pc.valid = false; }
    c.ntt_sub.release();
    release_fused(c);
}

static int pow_table_device(Ctx& c, const Fp& g, const Fp** out) {
    for (auto& pc : c.pow_cache)
        if (pc.valid && Fr::eq(pc.key, g)) { pc.stamp = ++c.pow_stamp; *out = pc.buf.as<Fp>(); return UZK_OK; }
    Ctx::PowCache* slot = &c.pow_cache[0];          // first free entry, else the least recently used one
    for (auto& pc : c.pow_cache) {
        if (!pc.valid) { slot = &pc; break; }
        if (pc.stamp < slot->stamp) slot = &pc;
    }
    std::vector<Fp> pw;
    host_pow_tables(g, pw);
    UZK_TRY(slot->buf.reserve(pw.size() * sizeof(Fp)));
    // the slot being replaced may still be read by kernels in flight on this stream; the copy is ordered after them
    UZK_HIP(hipMemcpyAsync(slot->buf.p, pw.data(), pw.size() * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));      // pw is a host vector of this call
    slot->key = g;
    slot->valid = true;
    slot->stamp = ++c.pow_stamp;
    *out = slot->buf.as<Fp>();
    return UZK_OK;
}

// ---- fused transforms: coset scaling and the radix-3 stage inside the first / last pass -----------------------
// Device tables of one (n, direction, coset shift): built once, kept (a proof uses a handful: k and 1/k on the 6n
// domain, nothing else), all dropped together when the cache is full or on shutdown.
struct NttFused {
    uint64_t n = 0;
    bool inverse = false, has_shift = false;
    Fp shift;
    PassArgs fx{};            // the fused fields of PassArgs, ready to copy
    Fp* tables[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
static void release_fused(Ctx& c) {
    for (void* v : c.ntt_fused) {
        NttFused* f = static_cast<NttFused*>(v);
        for (Fp* t : f->tables) if (t) (void)hipFree(t);
        delete f;
    }
    c.ntt_fused.clear();
}
// planes of count entries: base^j * scale (2^261-form)
static int gen_pow_planes(Ctx& c, const Fp& base, const Fp& scale, uint64_t count, Fp** out) {
    const Fp* pw = nullptr;
    UZK_TRY(pow_table_device(c, base, &pw));
    Fp* planes = nullptr;
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&planes), count * 36 + 64));
    KernelScope ks(c, "ntt_gen_tables");
    hipLaunchKernelGGL(ntt_gen_pow_planes_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c.stream, planes, count, pw, scale);
    UZK_HIP(hipGetLastError());
    *out = planes;
    return UZK_OK;
}
static Fp to_261(Fp v) { for (int d = 0; d < 5; ++d) v = Fr::add(v, v); return v; }

static int get_fused(Ctx& c, uint64_t n, bool inverse, const Fp* shift, NttFused** out) {
    for (void* v : c.ntt_fused) {
        NttFused* f = static_cast<NttFused*>(v);
        if (f->n == n && f->inverse == inverse && f->has_shift == (shift != nullptr) && (!shift || Fr::eq(f->shift, *shift))) { *out = f; return UZK_OK; }
    }
    if (c.ntt_fused.size() >= 16) { UZK_HIP(hipStreamSynchronize(c.stream)); release_fused(c); }
    NttFused* f = new NttFused();
    f->n = n; f->inverse = inverse; f->has_shift = shift != nullptr;
    if (shift) f->shift = *shift;
    int nt = 0;
    auto fail = [&](int rc) { for (Fp* t : f->tables) if (t) (void)hipFree(t); delete f; return rc; };
    const Fp one = Fr::one();
    if (n % 3 != 0) {
        // 2^k coset: forward pre-scales by shift^j in the first pass, inverse post-scales by shift^i in the last
        Fp* t = nullptr;
        int rc = gen_pow_planes(c, *shift, one, n, &t);
        if (rc != UZK_OK) return fail(rc);
        f->tables[nt++] = t;
        if (!inverse) f->fx.in_tw[0] = t; else f->fx.out_tw[0] = t;
    } else {
        // n = 3m.  X[3i' + r] = NTT_m( w^(rj) k^j (x_j + zeta^r k^m x_(j+m) + zeta^(2r) k^(2m) x_(j+2m)) )[i'],
        // w = group_gen(n) (inverted for the inverse transform), zeta = w^m, k = the forward coset shift (1 otherwise);
        // the inverse carries 1/n in its input tables and post-scales by shift^(3i' + r).
        const uint64_t m = n / 3;
        Fp w = fr_root_of_unity(n);
        if (inverse) w = fr_inv(w);
        const Fp zeta = f_pow_u64<Fr>(w, m);
        const Fp k = (!inverse && shift) ? *shift : one;
        const Fp km = f_pow_u64<Fr>(k, m), k2m = Fr::sqr(km);
        const Fp ninv = inverse ? fr_inv(fr_from_u64(n)) : one;
        f->fx.m3 = m;
        f->fx.out_mul = 3;
        f->fx.use_ab = (!inverse && shift) ? 1u : 0u;
        Fp zr = one, wr = one;                                   // zeta^r, w^r
        for (int r = 0; r < 3; ++r) {
            f->fx.alpha[r] = to_261(Fr::mul(zr, km));
            f->fx.beta[r] = to_261(Fr::mul(Fr::sqr(zr), k2m));
            const bool need_table = r != 0 || inverse || shift != nullptr;     // r = 0 plain forward: a + b + c, no products
            if (need_table) {
                Fp* t = nullptr;
                int rc = gen_pow_planes(c, Fr::mul(wr, k), ninv, m, &t);        // (w^r k)^j / n
                if (rc != UZK_OK) return fail(rc);
                f->tables[nt++] = t;
                f->fx.in_tw[r] = t;
            }
            zr = Fr::mul(zr, zeta);
            wr = Fr::mul(wr, w);
        }
        if (inverse && shift) {
            const Fp s3 = Fr::mul(Fr::sqr(*shift), *shift);
            Fp sr = one;
            for (int r = 0; r < 3; ++r) {
                Fp* t = nullptr;
                int rc = gen_pow_planes(c, s3, sr, m, &t);                      // shift^(3i' + r)
                if (rc != UZK_OK) return fail(rc);
                f->tables[nt++] = t;
                f->fx.out_tw[r] = t;
                sr = Fr::mul(sr, *shift);
            }
        }
    }
    c.ntt_fused.push_back(f);
    *out = f;
    return UZK_OK;
}

// `batch` independent transforms of n elements each (d_in -> d_out, may alias when the strides are equal); consecutive
// vectors are in_stride / out_stride elements apart (0 = contiguous).
// in_len (forward transforms over 3 * 2^k points only; 0 = no statement): every input vector is zero from index in_len on -- the
// first pass then neither reads nor combines the zero thirds (PassArgs::in_len).
int ntt_run(Ctx& c, const Fp* d_in, Fp* d_out, uint64_t n, bool inverse, const Fp* coset_shift_host, uint32_t batch,
            uint64_t in_stride, uint64_t out_stride, uint64_t in_len) {
    if (!domain_supported(n)) {
        set_error("no evaluation domain of size %llu (need 2^k, k <= %d, or 3 * 2^k, k <= %d)", (unsigned long long)n, UZK_NTT_MAX_LOG2, UZK_NTT_MAX_LOG2_MIXED);
        return UZK_ERR_FFT;
    }
    if (batch == 0) return UZK_OK;
    if (batch > 65535) { set_error("ntt: batch %u exceeds 65535", batch); return UZK_ERR_PARAMETER; }
    if (in_stride == n) in_stride = 0;
    if (out_stride == n) out_stride = 0;
    if ((in_stride && in_stride < n) || (out_stride && out_stride < n)) { set_error("ntt: a vector stride is smaller than n"); return UZK_ERR_PARAMETER; }
    if (d_in == d_out && in_stride != out_stride) { set_error("ntt: in-place transforms need equal strides"); return UZK_ERR_PARAMETER; }
    const bool strided = in_stride != 0 || out_stride != 0;
    // 29-bit-limb pass kernels: coset scaling and the radix-3 stage of 3 * 2^k domains run inside the first / last pass
    // (no separate scaling, decimation or combination kernels; 2 launches for the prover's 98304-point coset FFTs)
    const uint64_t sub_n = n % 3 == 0 ? n / 3 : n;
    if (sub_n >= 4096 && (n % 3 == 0 || coset_shift_host != nullptr)) {
        NttFused* f = nullptr;
        UZK_TRY(get_fused(c, n, inverse, coset_shift_host, &f));
        if (n % 3 == 0) {
            PassArgs fx = f->fx;
            fx.in_len = (!inverse && in_len != 0 && in_len <= sub_n) ? in_len : 0;
            return ntt_pow2(c, d_in, d_out, sub_n, inverse, false, 3 * batch, &fx, in_stride, out_stride);
        }
        return ntt_pow2(c, d_in, d_out, n, inverse, inverse, batch, &f->fx, in_stride, out_stride);
    }
    if (n % 3 != 0 && coset_shift_host == nullptr) return ntt_pow2(c, d_in, d_out, n, inverse, inverse, batch, nullptr, in_stride, out_stride);
    if (strided) {
        // the separate scaling / decimation kernels (small or unfused transforms) work on contiguous batches: one vector at a time
        for (uint32_t b = 0; b < batch; ++b)
            UZK_TRY(ntt_run(c, d_in + (uint64_t)b * (in_stride ? in_stride : n), d_out + (uint64_t)b * (out_stride ? out_stride : n), n, inverse,
                            coset_shift_host, 1, 0, 0, 0));
        return UZK_OK;
    }
    const dim3 egrid((unsigned)((n + 255) / 256), batch);
    const Fp* src = d_in;
    if (coset_shift_host != nullptr && !inverse) {
        const Fp* pw = nullptr;
        UZK_TRY(pow_table_device(c, *coset_shift_host, &pw));
        KernelScope ks(c, "ntt_scale_pows");
        hipLaunchKernelGGL(ntt_scale_pows_kernel, egrid, dim3(256), 0, c.stream, src, d_out, n, pw);
        src = d_out;
    }
    if (n % 3 != 0) {
        UZK_TRY(ntt_pow2(c, src, d_out, n, inverse, inverse, batch));
    } else {
        // decimation in time by 3: per vector, sub[k][j] = x[3j + k]; 3*batch transforms of size m
        const uint64_t m = n / 3;
        UZK_TRY(c.ntt_sub.reserve((size_t)n * batch * sizeof(Fp)));
        Fp* s = c.ntt_sub.as<Fp>();
        {
            KernelScope ks(c, "ntt_decimate3");
            hipLaunchKernelGGL(ntt_decimate3_kernel, egrid, dim3(256), 0, c.stream, src, s, m);
        }
        UZK_TRY(ntt_pow2(c, s, s, m, inverse, false, 3 * batch));
        Fp w = fr_root_of_unity(n);
        if (inverse) w = fr_inv(w);
        const Fp* pw = nullptr;
        UZK_TRY(pow_table_device(c, w, &pw));
        Fp scale = inverse ? fr_inv(fr_from_u64(n)) : Fr::one();
        KernelScope ks(c, "ntt_combine3");
        hipLaunchKernelGGL(ntt_combine3_kernel, egrid, dim3(256), 0, c.stream, s, d_out, m, pw, scale, inverse ? 1 : 0);
    }
    if (coset_shift_host != nullptr && inverse) {
        const Fp* pw = nullptr;
        UZK_TRY(pow_table_device(c, *coset_shift_host, &pw));
        KernelScope ks(c, "ntt_scale_pows");
        hipLaunchKernelGGL(ntt_scale_pows_kernel, egrid, dim3(256), 0, c.stream, d_out, d_out, n, pw);
    }
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

}  // namespace uzk
