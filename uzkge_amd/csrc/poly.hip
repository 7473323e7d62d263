// Polynomial helpers next to the hot path: the O(n) host loops the prover runs
// between its transforms and commits, kept on the device so evaluations need not leave HBM.
//   fr_scan_mul      prefix / suffix products over Fr (building block)
//   poly_eval_batch  p_b(x) for a batch of polynomials at one point -- FpPolynomial::eval,
//                    uzkge/src/poly_commit/field_polynomial.rs:198-209 (19 + 20 Horner evaluations
//                    per proof, uzkge/src/plonk/prover.rs:246-273, pcs.rs:126)
//   z_poly           the permutation grand product, uzkge/src/plonk/helpers.rs:160-220:
//                    z_0 = 1, z_{i+1} = z_i * prod_j (f_j(i) + beta k_j w^i + gamma)
//                                              / (f_j(i) + beta perm_j(i) + gamma)
//                    (rayon map + batch_inversion + a serial prefix product in the reference;
//                    here two product scans and ONE inversion)
// All values are exact field elements, so any evaluation order gives the reference's bytes.
#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "lz29.hpp"
#include "host_math.hpp"

namespace uzk {

constexpr int kScanPer = 8;                    // elements per lane
constexpr int kScanBlock = 256 * kScanPer;     // elements per workgroup

// Inclusive product scan of one block of `in` (reverse: from the right end), block totals out.
__global__ __launch_bounds__(256) void fr_scan_block_kernel(const Fp* __restrict__ in, Fp* __restrict__ out,
                                                            Fp* __restrict__ block_tot, uint64_t n, int reverse) {
    __shared__ Fp sh[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)tid * kScanPer;
    Fp v[kScanPer];
    Fp run = Fr::one();
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        const uint64_t i = base + e;
        Fp x = Fr::one();
        if (i < n) x = in[reverse ? n - 1 - i : i];
        run = Fr::mul(run, x);
        v[e] = run;
    }
    sh[tid] = run;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {   // Hillis-Steele inclusive scan of lane totals
        Fp t = (tid >= off) ? sh[tid - off] : Fr::one();
        __syncthreads();
        if (tid >= off) sh[tid] = Fr::mul(sh[tid], t);
        __syncthreads();
    }
    const Fp pre = tid ? sh[tid - 1] : Fr::one();
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        const uint64_t i = base + e;
        if (i < n) out[reverse ? n - 1 - i : i] = tid ? Fr::mul(v[e], pre) : v[e];
    }
    if (tid == 255 && block_tot) block_tot[blockIdx.x] = sh[255];
}
// out[i] *= (exclusive prefix of block totals)[block of i]
__global__ __launch_bounds__(256) void fr_scan_apply_kernel(Fp* __restrict__ out, const Fp* __restrict__ tot_scan, uint64_t n,
                                                            int reverse) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t blk = i / kScanBlock;
    if (blk == 0) return;
    const uint64_t pos = reverse ? n - 1 - i : i;
    out[pos] = Fr::mul(out[pos], tot_scan[blk - 1]);
}

// The two scans of z_poly (prefix products of the numerators, suffix products of the denominators, same length) as ONE launch
// sequence: blockIdx.y picks the array, blockIdx.z the lane of a lockstep batch (lane_stride elements apart).  in / out: [2][n]
// contiguous per lane; array 0 forward, array 1 reverse.  total_host (optional, pinned): the full product of array 1 of every
// lane -- the denominator z_poly inverts -- written by the lane that produces it.
__global__ __launch_bounds__(256) void fr_scan_block2_kernel(const Fp* __restrict__ in, Fp* __restrict__ out, Fp* __restrict__ block_tot,
                                                             uint64_t n, uint64_t nb, uint64_t lane_stride, Fp* __restrict__ total_host) {
    __shared__ Fp sh[256];
    const uint32_t tid = threadIdx.x, which = blockIdx.y;
    const int reverse = which;
    in += (uint64_t)blockIdx.z * lane_stride + (uint64_t)which * n; out += (uint64_t)blockIdx.z * lane_stride + (uint64_t)which * n;
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)tid * kScanPer;
    Fp v[kScanPer];
    Fp run = Fr::one();
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        const uint64_t i = base + e;
        Fp x = Fr::one();
        if (i < n) x = in[reverse ? n - 1 - i : i];
        run = Fr::mul(run, x);
        v[e] = run;
    }
    sh[tid] = run;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        Fp t = (tid >= off) ? sh[tid - off] : Fr::one();
        __syncthreads();
        if (tid >= off) sh[tid] = Fr::mul(sh[tid], t);
        __syncthreads();
    }
    const Fp pre = tid ? sh[tid - 1] : Fr::one();
#pragma unroll
    for (int e = 0; e < kScanPer; ++e) {
        const uint64_t i = base + e;
        if (i < n) out[reverse ? n - 1 - i : i] = tid ? Fr::mul(v[e], pre) : v[e];
    }
    if (tid == 255) {
        block_tot[((uint64_t)blockIdx.z * 2 + which) * nb + blockIdx.x] = sh[255];
        if (nb == 1 && which == 1 && total_host) total_host[blockIdx.z] = sh[255];
    }
}
// second level for both arrays (nb <= 256 blocks each: n <= 2^19) and the fix-up of every element, in one kernel: each
// workgroup rebuilds the exclusive prefix of the block totals it needs (nb - 1 dependent products at most, nb = 8 at n = 2^14)
__global__ __launch_bounds__(256) void fr_scan_apply2_kernel(Fp* __restrict__ out, const Fp* __restrict__ block_tot, uint64_t n, uint64_t nb,
                                                             uint64_t lane_stride, Fp* __restrict__ total_host) {
    const uint32_t which = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t blk = i / kScanBlock;
    if (blk == 0) return;
    const Fp* tot = block_tot + ((uint64_t)blockIdx.z * 2 + which) * nb;
    Fp pre = tot[0];
    for (uint64_t b = 1; b < blk; ++b) pre = Fr::mul(pre, tot[b]);
    Fp* o = out + (uint64_t)blockIdx.z * lane_stride + (uint64_t)which * n;
    const uint64_t pos = which ? n - 1 - i : i;
    const Fp v = Fr::mul(o[pos], pre);
    o[pos] = v;
    if (which == 1 && i == n - 1 && total_host) total_host[blockIdx.z] = v;     // nb > 1: the last element of the reversed scan lies in the last block
}
// d_in / d_out: [lanes][..][2][n] (lane_stride apart); n <= 32 * kScanBlock (the fix-up walks the block totals serially)
static int fr_scan_mul2(Ctx& c, const Fp* d_in, Fp* d_out, uint64_t n, DevBuf& tmp, uint32_t lanes = 1, uint64_t lane_stride = 0, Fp* total_host = nullptr) {
    const uint64_t nb = (n + kScanBlock - 1) / kScanBlock;
    UZK_TRY(tmp.reserve(2 * nb * lanes * sizeof(Fp)));
    KernelScope ks(c, "fr_scan");
    hipLaunchKernelGGL(fr_scan_block2_kernel, dim3((unsigned)nb, 2, lanes), dim3(256), 0, c.stream, d_in, d_out, tmp.as<Fp>(), n, nb, lane_stride, total_host);
    if (nb > 1)
        hipLaunchKernelGGL(fr_scan_apply2_kernel, dim3((unsigned)((n + 255) / 256), 2, lanes), dim3(256), 0, c.stream, d_out, tmp.as<Fp>(), n, nb, lane_stride, total_host);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// d_out[i] = prod_{t <= i} d_in[t]  (reverse: prod_{t >= i}); n <= 2048^2
static int fr_scan_mul(Ctx& c, const Fp* d_in, Fp* d_out, uint64_t n, bool reverse, DevBuf& tmp) {
    if (n == 0) return UZK_OK;
    const uint64_t nb = (n + kScanBlock - 1) / kScanBlock;
    if (nb > (uint64_t)kScanBlock) { set_error("scan: n = %llu too large", (unsigned long long)n); return UZK_ERR_PARAMETER; }
    UZK_TRY(tmp.reserve(2 * nb * sizeof(Fp)));
    Fp* tot = tmp.as<Fp>();
    Fp* tot_scan = tot + nb;
    KernelScope ks(c, "fr_scan");
    hipLaunchKernelGGL(fr_scan_block_kernel, dim3((unsigned)nb), dim3(256), 0, c.stream, d_in, d_out, tot, n, reverse ? 1 : 0);
    if (nb > 1) {
        hipLaunchKernelGGL(fr_scan_block_kernel, dim3(1), dim3(256), 0, c.stream, tot, tot_scan, (Fp*)nullptr, nb, 0);
        hipLaunchKernelGGL(fr_scan_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_out, tot_scan, n,
                           reverse ? 1 : 0);
    }
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- batched evaluation at one point -----------------------------------------------------------
__device__ inline Fp fr_pow_u64(Fp base, uint64_t e) {
    Fp acc = Fr::one();
    while (e) {
        if (e & 1) acc = Fr::mul(acc, base);
        base = Fr::sqr(base);
        e >>= 1;
    }
    return acc;
}
constexpr int kEvalPer = 16;                    // coefficients per lane (Horner)
constexpr int kEvalBlock = 256 * kEvalPer;      // coefficients per workgroup
// tab[t] = x^(16 t), t < 256;  tab[256 + b] = x^(4096 b), b < nblocks
__global__ __launch_bounds__(256) void poly_eval_tables_kernel(Fp* __restrict__ tab, Fp x, uint32_t nblocks) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 256) tab[t] = fr_pow_u64(x, (uint64_t)t * kEvalPer);
    else if (t < 256 + nblocks) tab[t] = fr_pow_u64(x, (uint64_t)(t - 256) * kEvalBlock);
}
// grid (nblocks, batch): partial[b][blk] = x^(4096 blk) * sum_{j in block} c_j x^(j - 4096 blk)
__global__ __launch_bounds__(256) void poly_eval_kernel(const Fp* __restrict__ coefs, uint64_t n, Fp x,
                                                        const Fp* __restrict__ tab, Fp* __restrict__ partial) {
    __shared__ Fp sh[256];
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, nblocks = gridDim.x;
    const Fp* c = coefs + (uint64_t)blockIdx.y * n;
    const uint64_t base = (uint64_t)blk * kEvalBlock + (uint64_t)tid * kEvalPer;
    Fp h = Fr::zero();
#pragma unroll
    for (int e = kEvalPer - 1; e >= 0; --e) {
        const uint64_t j = base + e;
        h = Fr::mul(h, x);
        if (j < n) h = Fr::add(h, c[j]);
    }
    sh[tid] = Fr::mul(h, tab[tid]);
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = Fr::add(sh[tid], sh[tid + s]);
        __syncthreads();
    }
    if (tid == 0) partial[(uint64_t)blockIdx.y * nblocks + blk] = Fr::mul(sh[0], tab[256 + blk]);
}
__global__ __launch_bounds__(64) void poly_eval_finish_kernel(const Fp* __restrict__ partial, uint32_t nblocks, uint32_t batch,
                                                             Fp* __restrict__ out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    Fp acc = Fr::zero();
    for (uint32_t k = 0; k < nblocks; ++k) acc = Fr::add(acc, partial[(uint64_t)b * nblocks + k]);
    out[b] = acc;
}


// Small polynomials (<= 256 blocks of 1024 coefficients: the prover's n = 2^14 and its 6n = 98304 quotient domain): ONE launch.
// grid (nblocks, batch), 256 lanes = one wave per SIMD.  Each workgroup builds its own power table x^(4 t), t < 256, in
// LDS by doubling (eight levels of one product and one squaring -- the separate table kernel's square-and-multiply per
// lane took 16 us), evaluates its 1024 coefficients, weights the sum by x^(1024 blk), and the last workgroup of a
// polynomial to arrive (a counter per polynomial, reset by that workgroup) adds the block sums and stores the value
// straight into pinned host memory: no table kernel, no finish kernel, no copy kernel.
constexpr int kEvalSmallPer = 4, kEvalSmallBlock = 256 * kEvalSmallPer, kEvalSmallMaxBlocks = 256;
constexpr uint32_t kEvalMaxPolys = 64, kEvalMaxPoints = 4;
// The polynomials of one launch: poly b has len[b] coefficients at p[b] and is evaluated at x[pt[b]] -- the prover's round 4
// (prover.rs:246-273: 15 polynomials at zeta, 4 at zeta * omega) is ONE launch.  By value: 1.2 KB of kernel arguments.
struct EvalArgs {
    const Fp* p[kEvalMaxPolys];
    uint32_t len[kEvalMaxPolys];
    uint8_t pt[kEvalMaxPolys];
    Fp x[kEvalMaxPoints];
};
__global__ __launch_bounds__(256) void poly_eval_small_kernel(EvalArgs a, Fp* __restrict__ partial,
                                                              uint32_t* __restrict__ counters, Fp* __restrict__ out_host) {
    __shared__ Fp pw[256];
    __shared__ Fp sh[256];
    __shared__ uint32_t last;
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, stride = gridDim.x, b = blockIdx.y;
    const uint64_t n = a.len[b];
    const uint32_t nblocks = (uint32_t)((n + kEvalSmallBlock - 1) / kEvalSmallBlock);
    if (n == 0 && blk == 0 && tid == 0) out_host[b] = Fr::zero();    // the zero polynomial (a real circuit's unused selectors trim to it)
    if (blk >= nblocks) return;                       // shorter polynomials of a mixed launch need fewer workgroups
    const Fp* c = a.p[b];
    const Fp x = a.x[a.pt[b]];
    const uint64_t base = (uint64_t)blk * kEvalSmallBlock + (uint64_t)tid * kEvalSmallPer;
    Fp h = Fr::zero();
#pragma unroll
    for (int e = kEvalSmallPer - 1; e >= 0; --e) {
        const uint64_t j = base + e;
        h = Fr::mul(h, x);
        if (j < n) h = Fr::add(h, c[j]);
    }
    // pw[t] = x^(4 t): pw[2^k + i] = pw[i] * s_k with s_k = x^(4 2^k) kept by every lane
    Fp s = Fr::sqr(Fr::sqr(x));
    if (tid == 0) pw[0] = Fr::one();
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t half = 1u << k;
        if (tid < half) pw[half + tid] = Fr::mul(pw[tid], s);
        s = Fr::sqr(s);
        __syncthreads();
    }
    // s = x^1024: the block's weight is s^blk (blk < 256)
    Fp wblk = Fr::one(), sp = s;
    for (uint32_t e = blk; e; e >>= 1) {
        if (e & 1) wblk = Fr::mul(wblk, sp);
        sp = Fr::sqr(sp);
    }
    sh[tid] = Fr::mul(h, pw[tid]);
    __syncthreads();
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (tid < st) sh[tid] = Fr::add(sh[tid], sh[tid + st]);
        __syncthreads();
    }
    if (tid == 0) {
        partial[(uint64_t)b * stride + blk] = Fr::mul(sh[0], wblk);
        __threadfence();
        last = (atomicAdd(&counters[b], 1u) == nblocks - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        Fp acc = Fr::zero();
        for (uint32_t k = 0; k < nblocks; ++k) acc = Fr::add(acc, partial[(uint64_t)b * stride + k]);
        out_host[b] = acc;
        counters[b] = 0;                       // ready for the next call on this stream
    }
}

static int poly_host_reserve(Ctx& c, size_t bytes) {
    if (c.poly_host_cap >= bytes) return UZK_OK;
    if (c.poly_host) { UZK_HIP(hipStreamSynchronize(c.stream)); (void)hipHostFree(c.poly_host); c.poly_host = nullptr; c.poly_host_cap = 0; }
    const size_t cap = std::max<size_t>(bytes, 1 << 16);
    UZK_HIP(hipHostMalloc(&c.poly_host, cap, hipHostMallocDefault));
    c.poly_host_cap = cap;
    return UZK_OK;
}

// One launch over `count` <= 64 polynomials (each <= 2^18 coefficients), results copied to out_host after the stream is idle.
static int poly_eval_small_launch(Ctx& c, const EvalArgs& a, uint32_t count, uint32_t max_blocks, Fp* out_host) {
    UZK_TRY(c.poly_tmp.reserve((size_t)count * max_blocks * sizeof(Fp)));
    if (c.poly_cnt.cap < (size_t)kEvalMaxPolys * 4) {
        UZK_TRY(c.poly_cnt.reserve(4096));
        UZK_HIP(hipMemsetAsync(c.poly_cnt.p, 0, c.poly_cnt.cap, c.stream));
    }
    UZK_TRY(poly_host_reserve(c, (size_t)kEvalMaxPolys * sizeof(Fp)));
    {
        KernelScope ks(c, "poly_eval");
        hipLaunchKernelGGL(poly_eval_small_kernel, dim3(max_blocks, count), dim3(256), 0, c.stream, a, c.poly_tmp.as<Fp>(),
                           c.poly_cnt.as<uint32_t>(), static_cast<Fp*>(c.poly_host));
    }
    UZK_HIP(hipGetLastError());
    UZK_HIP(hipStreamSynchronize(c.stream));
    std::memcpy(out_host, c.poly_host, (size_t)count * sizeof(Fp));
    return UZK_OK;
}

int poly_eval_batch(Ctx& c, const Fp* d_coefs, uint64_t n, uint32_t batch, const Fp& x, Fp* out_host) {
    if (batch == 0) return UZK_OK;
    if (n == 0) { for (uint32_t b = 0; b < batch; ++b) out_host[b] = Fr::zero(); return UZK_OK; }
    if (batch > 65535) { set_error("poly_eval: batch %u exceeds 65535", batch); return UZK_ERR_PARAMETER; }
    const uint64_t nb_small = (n + kEvalSmallBlock - 1) / kEvalSmallBlock;
    if (nb_small <= (uint64_t)kEvalSmallMaxBlocks) {
        for (uint32_t b0 = 0; b0 < batch; b0 += kEvalMaxPolys) {
            const uint32_t cnt = std::min<uint32_t>(kEvalMaxPolys, batch - b0);
            EvalArgs a{};
            a.x[0] = x;
            for (uint32_t k = 0; k < cnt; ++k) { a.p[k] = d_coefs + (uint64_t)(b0 + k) * n; a.len[k] = (uint32_t)n; a.pt[k] = 0; }
            UZK_TRY(poly_eval_small_launch(c, a, cnt, (uint32_t)nb_small, out_host + b0));
        }
        return UZK_OK;
    }
    const uint32_t nblocks = (uint32_t)((n + kEvalBlock - 1) / kEvalBlock);
    UZK_TRY(c.poly_tmp.reserve(((size_t)256 + nblocks + (size_t)batch * nblocks + batch) * sizeof(Fp)));
    Fp* tab = c.poly_tmp.as<Fp>();
    Fp* partial = tab + 256 + nblocks;
    Fp* d_out = partial + (size_t)batch * nblocks;
    {
        KernelScope ks(c, "poly_eval");
        hipLaunchKernelGGL(poly_eval_tables_kernel, dim3((256 + nblocks + 255) / 256), dim3(256), 0, c.stream, tab, x, nblocks);
        hipLaunchKernelGGL(poly_eval_kernel, dim3(nblocks, batch), dim3(256), 0, c.stream, d_coefs, n, x, tab, partial);
        hipLaunchKernelGGL(poly_eval_finish_kernel, dim3((batch + 63) / 64), dim3(64), 0, c.stream, partial, nblocks, batch, d_out);
    }
    UZK_HIP(hipGetLastError());
    UZK_HIP(hipMemcpyAsync(out_host, d_out, (size_t)batch * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}

// `count` device-resident polynomials, polynomial k evaluated at points[point_idx[k]]: one launch for count <= 64 and
// lengths <= 2^18 (the prover's round 4), otherwise one call per polynomial.
int poly_eval_ptrs(Ctx& c, const void* const* d_polys, const uint64_t* lens, const uint32_t* point_idx, uint32_t count,
                   const Fp* points_host, uint32_t n_points, Fp* out_host) {
    if (count == 0) return UZK_OK;
    if (n_points == 0) { set_error("poly_eval: no evaluation point"); return UZK_ERR_PARAMETER; }
    uint64_t max_len = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (point_idx[k] >= n_points) { set_error("poly_eval: polynomial %u names point %u of %u", k, point_idx[k], n_points); return UZK_ERR_PARAMETER; }
        if (lens[k] && !d_polys[k]) { set_error("poly_eval: polynomial %u is null", k); return UZK_ERR_PARAMETER; }
        max_len = std::max(max_len, lens[k]);
    }
    const uint64_t max_blocks = (max_len + kEvalSmallBlock - 1) / kEvalSmallBlock;
    if (count <= kEvalMaxPolys && n_points <= kEvalMaxPoints && max_blocks <= (uint64_t)kEvalSmallMaxBlocks && max_len > 0) {
        EvalArgs a{};
        for (uint32_t j = 0; j < n_points; ++j) a.x[j] = points_host[j];
        for (uint32_t k = 0; k < count; ++k) { a.p[k] = static_cast<const Fp*>(d_polys[k]); a.len[k] = (uint32_t)lens[k]; a.pt[k] = (uint8_t)point_idx[k]; }
        return poly_eval_small_launch(c, a, count, (uint32_t)max_blocks, out_host);
    }
    for (uint32_t k = 0; k < count; ++k)
        UZK_TRY(poly_eval_batch(c, static_cast<const Fp*>(d_polys[k]), lens[k], 1, points_host[point_idx[k]], out_host + k));
    return UZK_OK;
}

// ---- z_poly -------------------------------------------------------------------------------------
struct ZPolyArgs {
    const Fp* w;            // [n_wires][n] wire values (lane b: w + b * w_lane_stride)
    const uint32_t* perm;   // [n_wires][n] permutation (values < n_wires * n)
    const Fp* group;        // [n] omega^i
    Fp k[8];                // coset representatives k_j (n_wires <= 8)
    Fp beta, gamma;         // one lane: by value
    const Fp* bg;           // lanes: beta, gamma of lane b at bg[2 b], bg[2 b + 1] (device memory); null: the values above
    uint64_t w_lane_stride;
    uint32_t n, n_wires;
};
// i < n-1: num[i] = prod_j (f_j(i) + gamma + beta k_j w^i), den[i] = prod_j (f_j(i) + gamma + beta perm_j(i))
// work: [lanes][4][n] = num | den | P | S per lane
__global__ __launch_bounds__(256) void z_poly_terms_kernel(ZPolyArgs a, Fp* __restrict__ work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= a.n) return;
    Fp* num = work + (uint64_t)b * 4 * a.n;
    Fp* den = num + a.n;
    if (i + 1 == a.n) { num[i] = Fr::one(); den[i] = Fr::one(); return; }     // pads the two scans to n elements (fr_scan_mul2)
    const Fp beta = a.bg ? a.bg[2 * b] : a.beta, gamma = a.bg ? a.bg[2 * b + 1] : a.gamma;
    const Fp* w = a.w + (uint64_t)b * a.w_lane_stride;
    const Fp gi = a.group[i];
    Fp nm = Fr::one(), dn = Fr::one();
    for (uint32_t j = 0; j < a.n_wires; ++j) {
        const Fp f = w[(size_t)j * a.n + i];
        const Fp fg = Fr::add(f, gamma);
        nm = Fr::mul(nm, Fr::add(fg, Fr::mul(beta, Fr::mul(a.k[j], gi))));
        const uint32_t pv = a.perm[(size_t)j * a.n + i];
        const Fp px = Fr::mul(a.k[pv / a.n], a.group[pv % a.n]);      // p_of_x, helpers.rs:174-182
        dn = Fr::mul(dn, Fr::add(fg, Fr::mul(beta, px)));
    }
    num[i] = nm;
    den[i] = dn;
}
// z[0] = 1; z[i+1] = P[i] * S[i+1] * inv_total  (P = prefix products of num, S = suffix products of den,
// S[n-1] = 1, inv_total = 1 / S[0]); lane b: inv_total = inv[b], z at z + b * z_lane_stride
__global__ __launch_bounds__(256) void z_poly_combine_kernel(const Fp* __restrict__ work, const Fp* __restrict__ inv, Fp inv_one,
                                                             uint32_t n, Fp* __restrict__ z, uint64_t z_lane_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n) return;
    z += (uint64_t)b * z_lane_stride;
    if (i == 0) { z[0] = Fr::one(); return; }
    const Fp* P = work + (uint64_t)b * 4 * n + 2 * (uint64_t)n;
    const Fp* S = P + n;
    Fp v = Fr::mul(P[i - 1], inv ? inv[b] : inv_one);
    if (i < n - 1) v = Fr::mul(v, S[i]);
    z[i] = v;
}

// Device-resident core: d_w [n_wires][n], d_perm [n_wires][n], d_group [n]  ->  d_z [n].
int z_poly_device(Ctx& c, const Fp* d_w, const uint32_t* d_perm, const Fp* d_group, const Fp* k_host, const Fp& beta,
                  const Fp& gamma, uint32_t n, uint32_t n_wires, Fp* d_z) {
    if (n == 0) return UZK_OK;
    if (n_wires == 0 || n_wires > 8) { set_error("z_poly: n_wires must be 1..8"); return UZK_ERR_PARAMETER; }
    if (n == 1) {
        const Fp one = Fr::one();
        UZK_HIP(hipMemcpyAsync(d_z, &one, sizeof(Fp), hipMemcpyHostToDevice, c.stream));
        UZK_HIP(hipStreamSynchronize(c.stream));
        return UZK_OK;
    }
    // workspace: num | den | P | S
    UZK_TRY(c.zpoly_tmp.reserve(4 * (size_t)n * sizeof(Fp)));
    Fp* d_num = c.zpoly_tmp.as<Fp>();
    Fp* d_den = d_num + n;
    Fp* d_P = d_den + n;
    Fp* d_S = d_P + n;
    ZPolyArgs a;
    a.w = d_w; a.perm = d_perm; a.group = d_group; a.bg = nullptr; a.w_lane_stride = 0;
    for (uint32_t j = 0; j < 8; ++j) a.k[j] = j < n_wires ? k_host[j] : Fr::zero();
    a.beta = beta; a.gamma = gamma; a.n = n; a.n_wires = n_wires;
    const uint32_t m = n - 1;
    {
        KernelScope ks(c, "z_poly_terms");
        hipLaunchKernelGGL(z_poly_terms_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, a, d_num);
    }
    if (m <= 32ull * kScanBlock) {
        // num | den and P | S are adjacent pairs of the workspace: both scans in two launches instead of six.  The pairs are n
        // apart while the scans run over m = n - 1 elements: the kernels take the array distance from their `n`, so scan
        // n elements each -- the extra last element (index n - 1 of num / den) is never read by the combine step, it only has to be
        // defined: z_poly_terms gives it a one.
        UZK_TRY(fr_scan_mul2(c, d_num, d_P, n, c.poly_tmp));
    } else {
        UZK_TRY(fr_scan_mul(c, d_num, d_P, m, false, c.poly_tmp));
        UZK_TRY(fr_scan_mul(c, d_den, d_S, m, true, c.poly_tmp2));
    }
    Fp total;
    UZK_HIP(hipMemcpyAsync(&total, d_S, sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    if (Fr::is_zero(total)) {
        // a zero denominator: the reference's batch_inversion leaves zeros in place; mirror it by
        // refusing rather than inventing a value (cannot happen for random beta, gamma)
        set_error("z_poly: a permutation denominator is zero");
        return UZK_ERR_PARAMETER;
    }
    const Fp inv_total = fr_inv(total);     // the single inversion, on the host (~15 us)
    {
        KernelScope ks(c, "z_poly_combine");
        hipLaunchKernelGGL(z_poly_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, c.stream, d_num, (const Fp*)nullptr, inv_total, n, d_z, (uint64_t)0);
    }
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// The grand products of the `lanes` proofs of a lockstep batch (prover round 2): lane b's wires at d_w + b * w_lane_stride, its
// beta / gamma at d_bg[2 b], d_bg[2 b + 1] (device memory), z to d_z + b * z_lane_stride.  Three launches, ONE synchronisation
// and one host inversion (Montgomery's trick over the lanes' totals) for the whole batch; the inverses return to the device
// through `args`.  lane_ok[b] = 0 marks a lane whose denominator product is zero (its z is not written).  n <= 2^16.
int z_poly_lanes(Ctx& c, ArgArena& args, const Fp* d_w, uint64_t w_lane_stride, const uint32_t* d_perm, const Fp* d_group, const Fp* k_host,
                 const Fp* d_bg, uint32_t n, uint32_t n_wires, uint32_t lanes, Fp* d_z, uint64_t z_lane_stride, uint8_t* lane_ok) {
    if (n < 2 || (uint64_t)n > 32ull * kScanBlock || n_wires == 0 || n_wires > 8 || lanes == 0) { set_error("z_poly_lanes: bad shape"); return UZK_ERR_PARAMETER; }
    UZK_TRY(c.zpoly_tmp.reserve(4 * (size_t)n * lanes * sizeof(Fp)));
    UZK_TRY(poly_host_reserve(c, (size_t)lanes * sizeof(Fp)));
    Fp* work = c.zpoly_tmp.as<Fp>();
    Fp* totals = static_cast<Fp*>(c.poly_host);
    ZPolyArgs a;
    a.w = d_w; a.perm = d_perm; a.group = d_group; a.bg = d_bg; a.w_lane_stride = w_lane_stride;
    for (uint32_t j = 0; j < 8; ++j) a.k[j] = j < n_wires ? k_host[j] : Fr::zero();
    a.beta = Fr::zero(); a.gamma = Fr::zero(); a.n = n; a.n_wires = n_wires;
    {
        KernelScope ks(c, "z_poly_terms");
        hipLaunchKernelGGL(z_poly_terms_kernel, dim3((n + 255) / 256, lanes), dim3(256), 0, c.stream, a, work);
    }
    UZK_TRY(fr_scan_mul2(c, work, work + 2 * (uint64_t)n, n, c.poly_tmp, lanes, 4 * (uint64_t)n, totals));
    UZK_HIP(hipStreamSynchronize(c.stream));
    // 1 / total_b for every lane with one inversion: prefix products, invert the last, walk back
    const Fp* d_inv = nullptr;
    Fp* inv = args.push<Fp>(lanes, &d_inv);
    if (!inv) { set_error("z_poly_lanes: argument block full"); return UZK_ERR_PARAMETER; }
    std::vector<Fp> pre(lanes);
    Fp acc = Fr::one();
    for (uint32_t b = 0; b < lanes; ++b) {
        lane_ok[b] = Fr::is_zero(totals[b]) ? 0 : 1;
        pre[b] = acc;
        if (lane_ok[b]) acc = Fr::mul(acc, totals[b]);
    }
    Fp run = fr_inv(acc);
    for (uint32_t b = lanes; b-- > 0;) {
        if (!lane_ok[b]) { inv[b] = Fr::zero(); continue; }
        inv[b] = Fr::mul(run, pre[b]);
        run = Fr::mul(run, totals[b]);
    }
    UZK_TRY(args.upload(c.stream));
    {
        KernelScope ks(c, "z_poly_combine");
        hipLaunchKernelGGL(z_poly_combine_kernel, dim3((n + 255) / 256, lanes), dim3(256), 0, c.stream, work, d_inv, Fr::zero(), n, d_z, z_lane_stride);
    }
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

int z_poly_run(Ctx& c, const Fp* w_host, const uint32_t* perm_host, const Fp* group_host, const Fp* k_host,
               const Fp& beta, const Fp& gamma, uint32_t n, uint32_t n_wires, Fp* z_host) {
    if (n == 0) return UZK_OK;
    if (n_wires == 0 || n_wires > 8) { set_error("z_poly: n_wires must be 1..8"); return UZK_ERR_PARAMETER; }
    const size_t wn = (size_t)n_wires * n;
    // staging: w | group | z (Fp), perm (u32)
    UZK_TRY(c.poly_io.reserve((wn + 2 * (size_t)n) * sizeof(Fp) + wn * sizeof(uint32_t)));
    Fp* d_w = c.poly_io.as<Fp>();
    Fp* d_group = d_w + wn;
    Fp* d_z = d_group + n;
    uint32_t* d_perm = reinterpret_cast<uint32_t*>(d_z + n);
    UZK_HIP(hipMemcpyAsync(d_w, w_host, wn * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipMemcpyAsync(d_group, group_host, (size_t)n * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    UZK_HIP(hipMemcpyAsync(d_perm, perm_host, wn * sizeof(uint32_t), hipMemcpyHostToDevice, c.stream));
    UZK_TRY(z_poly_device(c, d_w, d_perm, d_group, k_host, beta, gamma, n, n_wires, d_z));
    UZK_HIP(hipMemcpyAsync(z_host, d_z, (size_t)n * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
    UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}

// ---------------------------------------------------------------------------------------------
// t_poly quotient evaluations (uzk_t_quotient_device).  One lane per point of the m-domain.  The
// field arithmetic is exact, so the expression is factored freely (the result is the same canonical
// element as the reference's term-by-term sum): the four selector-weighted curve-addition
// constraints share S = sum sel_ab and the six sums  sum sel_ab * q_{x,y,dxy}_ab; the two Anemoi
// fifth powers are shared by terms 8/10 and 9/11.  About 140 products per point (the reference
// spends about 330) and 60 coalesced 32-byte loads: 1.9 KB/point, HBM-stream bound.
// ---------------------------------------------------------------------------------------------
// What every lane of a launch shares (by value), and one lane's challenges (device memory, read with scalar loads).  The first
// ten vectors (the proof's own: w0..4, w_sel0..2, pi, z) of lane b start own_stride * b elements behind vec[0..9]; the 46 circuit
// tables are the same for every lane.
struct QuotientDev {
    uint32_t m, factor;
    const Fp* vec[56];
    uint64_t own_stride;
    Fp g, g_inv, g2p1, ea;
    Fp zhi[16];
};
struct QuotientLane {
    Fp ap[17];          // alpha^0 .. alpha^16
    Fp beta, gamma, bk[5];     // bk[j] = beta * k[j]
};

// The reference's loop body (helpers.rs:284-656) in four groups of terms of about equal length (34 / 43 / 36 / 38 products), so
// that small domains can put them on four waves (t_quotient_split_kernel); PART: 0 = terms 1-3 (gate, permutation),
// 1 = terms 4-11 (L1, booleanity, Anemoi round), 2 = terms 12, 13 (public-key curve addition), 3 = terms 14-18 (generator
// curve addition, ECC / wire-selector booleanity).  Parts 2 and 3 exist only with the "shuffle" feature.
template <int PART>
__device__ __forceinline__ Fp tq_part(const QuotientDev& a, const QuotientLane& l, uint64_t own, uint32_t point, uint32_t nxt) {
    const Fp one = Fr::one();
    auto L = [&](int slot) -> Fp { return slot < 10 ? a.vec[slot][own + point] : a.vec[slot][point]; };
    auto N = [&](int slot) -> Fp { return a.vec[slot][own + nxt]; };          // own vectors at the next row
    if constexpr (PART == 0) {
        const Fp w0 = L(0), w1 = L(1), w2 = L(2), w3 = L(3), w4 = L(4);
        const Fp z = L(9), zn = N(9);
        const Fp w0w1 = Fr::mul(w0, w1), w2w3 = Fr::mul(w2, w3);
        // term1: gate
        Fp acc = Fr::mul(L(10), w0);
        acc = Fr::add(acc, Fr::mul(L(11), w1));
        acc = Fr::add(acc, Fr::mul(L(12), w2));
        acc = Fr::add(acc, Fr::mul(L(13), w3));
        acc = Fr::add(acc, Fr::mul(L(14), w0w1));
        acc = Fr::add(acc, Fr::mul(L(15), w2w3));
        acc = Fr::add(acc, Fr::add(L(16), L(8)));
        acc = Fr::add(acc, Fr::mul(L(17), Fr::mul(Fr::mul(w0w1, w2w3), w4)));
        acc = Fr::sub(acc, Fr::mul(L(18), w4));
        // term2 - term3: permutation
        const Fp cq = L(30);
        const Fp wg[5] = {Fr::add(w0, l.gamma), Fr::add(w1, l.gamma), Fr::add(w2, l.gamma), Fr::add(w3, l.gamma), Fr::add(w4, l.gamma)};
        Fp t2 = Fr::mul(l.ap[1], z), t3 = Fr::mul(l.ap[1], zn);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            t2 = Fr::mul(t2, Fr::add(wg[j], Fr::mul(l.bk[j], cq)));
            t3 = Fr::mul(t3, Fr::add(wg[j], Fr::mul(l.beta, L(19 + j))));
        }
        return Fr::add(acc, Fr::sub(t2, t3));
    } else if constexpr (PART == 1) {
        const Fp w0 = L(0), w1 = L(1), w2 = L(2), w3 = L(3), w4 = L(4);
        const Fp w0n = N(0), w1n = N(1), w2n = N(2);
        const Fp z = L(9);
        // term4: alpha^2 L1 (z - 1)
        Fp acc = Fr::mul(Fr::mul(l.ap[2], L(24)), Fr::sub(z, one));
        // terms 5..7: qb * sum alpha^(2+i) w_i (w_i - 1)
        {
            Fp s = Fr::mul(l.ap[3], Fr::mul(w1, Fr::sub(w1, one)));
            s = Fr::add(s, Fr::mul(l.ap[4], Fr::mul(w2, Fr::sub(w2, one))));
            s = Fr::add(s, Fr::mul(l.ap[5], Fr::mul(w3, Fr::sub(w3, one))));
            acc = Fr::add(acc, Fr::mul(L(25), s));
        }
        // terms 8..11: Anemoi round, all weighted by q_prk3 and subtracted
        const Fp prk1 = L(26), prk2 = L(27), prk3 = L(28), prk4 = L(29);
        const Fp w3w0 = Fr::add(w0, w3), w2w1 = Fr::add(w1, w2);
        const Fp w3_2w0 = Fr::add(w0, w3w0), w2_2w1 = Fr::add(w1, w2w1);
        auto pow5 = [](const Fp& x) { const Fp x2 = Fr::sqr(x); return Fr::mul(Fr::sqr(x2), x); };
        const Fp tA = Fr::add(Fr::add(w3w0, Fr::mul(a.g, w2w1)), prk3);
        const Fp dA = pow5(Fr::sub(tA, w2n));
        Fp e8 = Fr::sub(Fr::add(dA, Fr::mul(a.g, Fr::sqr(tA))), Fr::add(Fr::add(w3_2w0, Fr::mul(a.g, w2_2w1)), prk1));
        Fp e10 = Fr::sub(Fr::add(Fr::add(dA, Fr::mul(a.g, Fr::sqr(w2n))), a.g_inv), w0n);
        const Fp tB = Fr::add(Fr::add(Fr::mul(a.g, w3w0), Fr::mul(a.g2p1, w2w1)), prk4);
        const Fp dB = pow5(Fr::sub(tB, w4));
        Fp e9 = Fr::sub(Fr::add(dB, Fr::mul(a.g, Fr::sqr(tB))),
                        Fr::add(Fr::add(Fr::mul(a.g, w3_2w0), Fr::mul(a.g2p1, w2_2w1)), prk2));
        Fp e11 = Fr::sub(Fr::add(Fr::add(dB, Fr::mul(a.g, Fr::sqr(w4))), a.g_inv), w1n);
        Fp s = Fr::mul(l.ap[6], e8);
        s = Fr::add(s, Fr::mul(l.ap[7], e9));
        s = Fr::add(s, Fr::mul(l.ap[8], e10));
        s = Fr::add(s, Fr::mul(l.ap[9], e11));
        return Fr::sub(acc, Fr::mul(prk3, s));
    } else {
        // the four selector weights of the curve-addition constraints, shared by terms 12..15
        const Fp ws0 = L(5), ws1 = L(6), ws2 = L(7), qecc = L(55);
        const Fp om0 = Fr::sub(one, ws0), om1 = Fr::sub(one, ws1);
        Fp sel[4];
        sel[0] = Fr::sub(Fr::add(Fr::mul(om0, om1), qecc), one);
        sel[1] = Fr::mul(ws0, om1);
        sel[2] = Fr::mul(om0, ws1);
        sel[3] = Fr::mul(ws0, ws1);
        const Fp S = Fr::add(Fr::add(sel[0], sel[1]), Fr::add(sel[2], sel[3]));
        const Fp ws2S = Fr::mul(ws2, S);
        constexpr int V0 = PART == 2 ? 0 : 3;           // sums of sel_ab * {x, y, dxy}_ab: the public key's (slots 31..42) or the generator's (43..54)
        Fp sums[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            Fp s = Fr::mul(sel[0], L(31 + 4 * (V0 + v)));
#pragma unroll
            for (int ab = 1; ab < 4; ++ab) s = Fr::add(s, Fr::mul(sel[ab], L(31 + 4 * (V0 + v) + ab)));
            sums[v] = s;
        }
        const Fp ws2Y = Fr::mul(ws2, sums[1]);
        if constexpr (PART == 2) {
            const Fp w0 = L(0), w1 = L(1), w0n = N(0), w1n = N(1);
            const Fp w01SD = Fr::mul(Fr::mul(w0, w1), sums[2]);
            // 12: ws2 w0n S - ws2 w0 SY - w1 SX + w0 w1 w0n SD
            Fp t12 = Fr::sub(Fr::add(Fr::mul(ws2S, w0n), Fr::mul(w01SD, w0n)), Fr::add(Fr::mul(ws2Y, w0), Fr::mul(w1, sums[0])));
            // 13: ws2 w1n S + a w0 SX - ws2 w1 SY - w0 w1 w1n SD
            Fp t13 = Fr::sub(Fr::add(Fr::mul(ws2S, w1n), Fr::mul(Fr::mul(a.ea, w0), sums[0])), Fr::add(Fr::mul(ws2Y, w1), Fr::mul(w01SD, w1n)));
            return Fr::add(Fr::mul(l.ap[10], t12), Fr::mul(l.ap[11], t13));
        } else {
            const Fp w2 = L(2), w3 = L(3), w4 = L(4), w2n = N(2);
            const Fp w23GD = Fr::mul(Fr::mul(w2, w3), sums[2]);
            // 14: ws2 w2n S - ws2 w2 GY - w3 GX + w2 w3 w2n GD
            Fp t14 = Fr::sub(Fr::add(Fr::mul(ws2S, w2n), Fr::mul(w23GD, w2n)), Fr::add(Fr::mul(ws2Y, w2), Fr::mul(w3, sums[0])));
            // 15: ws2 w4 S + a w2 GX - ws2 w3 GY - w2 w3 w4 GD
            Fp t15 = Fr::sub(Fr::add(Fr::mul(ws2S, w4), Fr::mul(Fr::mul(a.ea, w2), sums[0])), Fr::add(Fr::mul(ws2Y, w3), Fr::mul(w23GD, w4)));
            Fp s = Fr::add(Fr::mul(l.ap[12], t14), Fr::mul(l.ap[13], t15));
            const Fp omq = Fr::sub(one, qecc);
            // 16, 17: q_ecc ws (1 - ws) + (1 - q_ecc) ws ; 18: q_ecc (1 + ws2)(1 - ws2)
            s = Fr::add(s, Fr::mul(l.ap[14], Fr::mul(ws0, Fr::add(Fr::mul(qecc, om0), omq))));
            s = Fr::add(s, Fr::mul(l.ap[15], Fr::mul(ws1, Fr::add(Fr::mul(qecc, om1), omq))));
            return Fr::add(s, Fr::mul(l.ap[16], Fr::mul(qecc, Fr::mul(Fr::add(one, ws2), Fr::sub(one, ws2)))));
        }
    }
}

// The same four groups of terms on the 29-bit lazy-limb arithmetic (lz29.hpp; round 6): every loaded value enters the 2^261-form
// by re-limbing alone (32 a = x 2^261), a product is 162 multiply-adds with no carry chain instead of 128 + 128 add-with-carry,
// sums of two products share one reduction (mul2), additions and subtractions are nine independent 32-bit operations, and the
// bounds every operation needs are checked by the compiler (the Lz<K, V> types).  Exact field arithmetic throughout: the
// result is the same canonical element as tq_part's, bit for bit (tests/test_gpu_variants.py compares the two kernels).
// Returns the group's sum normalized, value < kTqPartV * M.
constexpr int kTqPartV = 160;
template <int PART>
__device__ __forceinline__ Lz<Fr29, 1, kTqPartV> tq_part29(const QuotientDev& a, const QuotientLane& l, uint64_t own, uint32_t point, uint32_t nxt) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Z = LzOps<Fr29>;
    auto L = [&](int slot) { return Z::ld(slot < 10 ? a.vec[slot][own + point] : a.vec[slot][point]); };
    auto N = [&](int slot) { return Z::ld(a.vec[slot][own + nxt]); };          // own vectors at the next row
    const auto one = Z::one();
    if constexpr (PART == 0) {
        const auto w0 = L(0), w1 = L(1), w2 = L(2), w3 = L(3), w4 = L(4);
        const auto z = L(9), zn = N(9);
        const auto w0w1 = Z::mul(w0, w1), w2w3 = Z::mul(w2, w3);
        // term1: gate -- pairs of products under one reduction
        const auto g1 = Z::mul2(L(10), w0, L(11), w1);
        const auto g2 = Z::mul2(L(12), w2, L(13), w3);
        const auto g3 = Z::mul2(L(14), w0w1, L(15), w2w3);
        const auto nw4 = Z::sub(Z::zero(), w4);                                 // -w4 (+ 33 M), limbs < 4 * 2^29
        const auto g4 = Z::mul2(L(17), Z::mul(Z::mul(w0w1, w2w3), w4), L(18), nw4);
        const auto gate = Z::norm(Z::add(Z::add(Z::add(g1, g2), Z::add(g3, g4)), Z::add(L(16), L(8))));
        // term2 - term3: permutation
        const auto cq = L(30);
        const auto gamma = Z::ld(l.gamma), beta = Z::ld(l.beta), ap1 = Z::ld(l.ap[1]);
        auto t2 = Z::template relax<1, 8>(Z::mul(ap1, z));
        auto t3 = Z::template relax<1, 8>(Z::mul(ap1, zn));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const auto wg = Z::add(j == 0 ? w0 : j == 1 ? w1 : j == 2 ? w2 : j == 3 ? w3 : w4, gamma);
            t2 = Z::template relax<1, 8>(Z::mul(t2, Z::add(wg, Z::mul(Z::ld(l.bk[j]), cq))));
            t3 = Z::template relax<1, 8>(Z::mul(t3, Z::add(wg, Z::mul(beta, L(19 + j)))));
        }
        return Z::template relax<1, kTqPartV>(Z::norm(Z::add(gate, Z::sub(t2, t3))));
    } else if constexpr (PART == 1) {
        const auto w0 = L(0), w1 = L(1), w2 = L(2), w3 = L(3), w4 = L(4);
        const auto w0n = N(0), w1n = N(1), w2n = N(2);
        const auto z = L(9);
        // term4: alpha^2 L1 (z - 1)
        const auto t4 = Z::mul(Z::mul(Z::ld(l.ap[2]), L(24)), Z::sub(z, one));
        // terms 5..7: qb * sum alpha^(2+i) w_i (w_i - 1)
        const auto b1 = Z::mul(w1, Z::sub(w1, one)), b2 = Z::mul(w2, Z::sub(w2, one)), b3 = Z::mul(w3, Z::sub(w3, one));
        const auto bs = Z::add(Z::mul2(Z::ld(l.ap[3]), b1, Z::ld(l.ap[4]), b2), Z::mul(Z::ld(l.ap[5]), b3));
        const auto t57 = Z::mul(L(25), bs);
        // terms 8..11: Anemoi round, all weighted by q_prk3 and subtracted
        const auto prk1 = L(26), prk2 = L(27), prk3 = L(28), prk4 = L(29);
        const auto g = Z::ld(a.g), g_inv = Z::ld(a.g_inv), g2p1 = Z::ld(a.g2p1);
        const auto w3w0 = Z::add(w0, w3), w2w1 = Z::add(w1, w2);
        const auto w3_2w0 = Z::add(w0, w3w0), w2_2w1 = Z::add(w1, w2w1);
        auto pow5 = [](const auto& x) { const auto x2 = Z::sqr(x); return Z::mul(Z::sqr(x2), x); };
        const auto tA = Z::norm(Z::add(Z::add(w3w0, Z::mul(g, w2w1)), prk3));
        const auto dA = pow5(Z::norm(Z::sub(tA, w2n)));
        const auto e8 = Z::norm(Z::sub(Z::add(dA, Z::mul(g, Z::sqr(tA))), Z::norm(Z::add(Z::add(w3_2w0, Z::mul(g, w2_2w1)), prk1))));
        const auto e10 = Z::norm(Z::sub(Z::add(Z::add(dA, Z::mul(g, Z::sqr(w2n))), g_inv), w0n));
        const auto tB = Z::norm(Z::add(Z::mul2(g, w3w0, g2p1, w2w1), prk4));
        const auto dB = pow5(Z::norm(Z::sub(tB, w4)));
        const auto e9 = Z::norm(Z::sub(Z::add(dB, Z::mul(g, Z::sqr(tB))), Z::norm(Z::add(Z::mul2(g, w3_2w0, g2p1, w2_2w1), prk2))));
        const auto e11 = Z::norm(Z::sub(Z::add(Z::add(dB, Z::mul(g, Z::sqr(w4))), g_inv), w1n));
        const auto s = Z::add(Z::mul2(Z::ld(l.ap[6]), e8, Z::ld(l.ap[7]), e9), Z::mul2(Z::ld(l.ap[8]), e10, Z::ld(l.ap[9]), e11));
        return Z::template relax<1, kTqPartV>(Z::norm(Z::sub(Z::add(t4, t57), Z::mul(prk3, s))));
    } else {
        // the four selector weights of the curve-addition constraints, shared by terms 12..15
        const auto ws0 = L(5), ws1 = L(6), ws2 = L(7), qecc = L(55);
        const auto om0 = Z::norm(Z::sub(one, ws0)), om1 = Z::norm(Z::sub(one, ws1));
        const auto sel0 = Z::norm(Z::sub(Z::add(Z::mul(om0, om1), qecc), one));
        const auto sel1 = Z::mul(ws0, om1), sel2 = Z::mul(om0, ws1), sel3 = Z::mul(ws0, ws1);
        const auto S = Z::norm(Z::add(Z::add(sel0, sel1), Z::add(sel2, sel3)));
        const auto ws2S = Z::mul(ws2, S);
        constexpr int V0 = PART == 2 ? 0 : 3;           // sums of sel_ab * {x, y, dxy}_ab: the public key's (slots 31..42) or the generator's (43..54)
        auto sum4 = [&](int v) {
            return Z::norm(Z::add(Z::mul2(sel0, L(31 + 4 * (V0 + v)), sel1, L(31 + 4 * (V0 + v) + 1)),
                                  Z::mul2(sel2, L(31 + 4 * (V0 + v) + 2), sel3, L(31 + 4 * (V0 + v) + 3))));
        };
        const auto SX = sum4(0), SY = sum4(1), SD = sum4(2);
        const auto ws2Y = Z::mul(ws2, SY);
        if constexpr (PART == 2) {
            const auto w0 = L(0), w1 = L(1), w0n = N(0), w1n = N(1);
            const auto w01SD = Z::mul(Z::mul(w0, w1), SD);
            // 12: ws2 w0n S + w0 w1 w0n SD - (ws2 w0 SY + w1 SX)
            const auto t12 = Z::norm(Z::sub(Z::mul2(ws2S, w0n, w01SD, w0n), Z::mul2(ws2Y, w0, w1, SX)));
            // 13: ws2 w1n S + a w0 SX - (ws2 w1 SY + w0 w1 w1n SD)
            const auto t13 = Z::norm(Z::sub(Z::mul2(ws2S, w1n, Z::mul(Z::ld(a.ea), w0), SX), Z::mul2(ws2Y, w1, w01SD, w1n)));
            return Z::template relax<1, kTqPartV>(Z::mul2(Z::ld(l.ap[10]), t12, Z::ld(l.ap[11]), t13));
        } else {
            const auto w2 = L(2), w3 = L(3), w4 = L(4), w2n = N(2);
            const auto w23GD = Z::mul(Z::mul(w2, w3), SD);
            // 14: ws2 w2n S + w2 w3 w2n GD - (ws2 w2 GY + w3 GX)
            const auto t14 = Z::norm(Z::sub(Z::mul2(ws2S, w2n, w23GD, w2n), Z::mul2(ws2Y, w2, w3, SX)));
            // 15: ws2 w4 S + a w2 GX - (ws2 w3 GY + w2 w3 w4 GD)
            const auto t15 = Z::norm(Z::sub(Z::mul2(ws2S, w4, Z::mul(Z::ld(a.ea), w2), SX), Z::mul2(ws2Y, w3, w23GD, w4)));
            const auto s1 = Z::mul2(Z::ld(l.ap[12]), t14, Z::ld(l.ap[13]), t15);
            const auto omq = Z::norm(Z::sub(one, qecc));
            // 16, 17: q_ecc ws (1 - ws) + (1 - q_ecc) ws ; 18: q_ecc (1 + ws2)(1 - ws2)
            const auto u16 = Z::mul(ws0, Z::norm(Z::add(Z::mul(qecc, om0), omq)));
            const auto u17 = Z::mul(ws1, Z::norm(Z::add(Z::mul(qecc, om1), omq)));
            const auto u18 = Z::mul(qecc, Z::mul(Z::add(one, ws2), Z::norm(Z::sub(one, ws2))));
            const auto s2 = Z::add(Z::mul2(Z::ld(l.ap[14]), u16, Z::ld(l.ap[15]), u17), Z::mul(Z::ld(l.ap[16]), u18));
            return Z::template relax<1, kTqPartV>(Z::norm(Z::add(s1, s2)));
        }
    }
#else
    return Lz<Fr29, 1, kTqPartV>{};
#endif
}

// SHUFFLE = false: the circuit has no shuffle / ECC selectors (uzkge built without the "shuffle" feature, e.g.
// zmatchmaking): terms 12..18 of helpers.rs:437-655 do not exist and their 28 vectors are not read.
template <bool SHUFFLE>
__global__ __launch_bounds__(256) void t_quotient_kernel(QuotientDev a, const QuotientLane* __restrict__ lanes, Fp* __restrict__ out, uint64_t out_stride) {
    const uint32_t point = blockIdx.x * blockDim.x + threadIdx.x;
    if (point >= a.m) return;
    const QuotientLane& l = lanes[blockIdx.y];
    const uint64_t own = (uint64_t)blockIdx.y * a.own_stride;
    uint32_t nxt = point + a.factor;
    if (nxt >= a.m) nxt -= a.m;
    Fp acc = Fr::add(tq_part<0>(a, l, own, point, nxt), tq_part<1>(a, l, own, point, nxt));
    if constexpr (SHUFFLE) acc = Fr::add(acc, Fr::add(tq_part<2>(a, l, own, point, nxt), tq_part<3>(a, l, own, point, nxt)));
    out[(uint64_t)blockIdx.y * out_stride + point] = Fr::mul(acc, a.zhi[point % a.factor]);
}
// The same with the groups of terms on the WAVES of a workgroup (64 points per workgroup, NPARTS waves): at the prover's size
// the loop has only 6n = 98 304 points -- 1.5 waves per SIMD, every lane a chain of ~150 dependent-latency products -- so the
// kernel is bound by the length of one lane's chain, not by issue slots or HBM; cutting the chain in four parts of 34..43
// products quadruples the waves (measured at n = 2^14, tools/ab_tq_split.py: 200 us in one piece, 143 us in two, 126 us in four --
// about 90 % of what 98 304 x 150 products cost in issue slots on the whole chip).  Field arithmetic is exact: the
// sum is the same element.  NPARTS = 4 with the shuffle-feature terms, 2 without.  blockIdx.y = the lane of a lockstep batch.
template <int NPARTS, int WPE>
__global__ __launch_bounds__(64 * NPARTS) __attribute__((amdgpu_waves_per_eu(WPE))) void t_quotient_split_kernel(QuotientDev a, const QuotientLane* __restrict__ lanes,
                                                                                                                  Fp* __restrict__ out, uint64_t out_stride) {
    __shared__ Fp part[NPARTS - 1][64];
    const uint32_t lane = threadIdx.x & 63, which = threadIdx.x >> 6;
    const uint32_t point = blockIdx.x * 64 + lane;
    const bool live = point < a.m;
    const QuotientLane& l = lanes[blockIdx.y];
    const uint64_t own = (uint64_t)blockIdx.y * a.own_stride;
    uint32_t nxt = point + a.factor;
    if (nxt >= a.m) nxt -= a.m;
    Fp acc = Fr::zero();
    if (live) {
        if (which == 0) acc = tq_part<0>(a, l, own, point, nxt);
        else if (which == 1) acc = tq_part<1>(a, l, own, point, nxt);
        else if (NPARTS > 2 && which == 2) acc = tq_part<2>(a, l, own, point, nxt);
        else if (NPARTS > 2) acc = tq_part<3>(a, l, own, point, nxt);
    }
    if (which) part[which - 1][lane] = acc;
    __syncthreads();
    if (which == 0 && live) {
#pragma unroll
        for (int p = 0; p < NPARTS - 1; ++p) acc = Fr::add(acc, part[p][lane]);
        // 1 / Z_H of this point's coset class, by selects: a dynamic index into the by-value argument block would make the compiler
        // copy all of it to scratch in every lane
        const uint32_t cls = point % a.factor;
        Fp zhi = a.zhi[0];
#pragma unroll
        for (int i = 1; i < 16; ++i)
            if (cls == (uint32_t)i) zhi = a.zhi[i];
        out[(uint64_t)blockIdx.y * out_stride + point] = Fr::mul(acc, zhi);
    }
}

// The two kernels above on the lazy 29-bit arithmetic (tq_part29): same launch shapes, same outputs.
template <bool SHUFFLE>
__global__ __launch_bounds__(256) void t_quotient29_kernel(QuotientDev a, const QuotientLane* __restrict__ lanes, Fp* __restrict__ out, uint64_t out_stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Z = LzOps<Fr29>;
    const uint32_t point = blockIdx.x * blockDim.x + threadIdx.x;
    if (point >= a.m) return;
    const QuotientLane& l = lanes[blockIdx.y];
    const uint64_t own = (uint64_t)blockIdx.y * a.own_stride;
    uint32_t nxt = point + a.factor;
    if (nxt >= a.m) nxt -= a.m;
    const uint32_t cls = point % a.factor;
    Fp zhi = a.zhi[0];
#pragma unroll
    for (int i = 1; i < 16; ++i)
        if (cls == (uint32_t)i) zhi = a.zhi[i];
    const auto s01 = Z::add(tq_part29<0>(a, l, own, point, nxt), tq_part29<1>(a, l, own, point, nxt));
    if constexpr (SHUFFLE) {
        const auto s = Z::add(s01, Z::add(tq_part29<2>(a, l, own, point, nxt), tq_part29<3>(a, l, own, point, nxt)));
        out[(uint64_t)blockIdx.y * out_stride + point] = Z::canon(Z::mul(s, Z::ldp(zhi)));
    } else {
        out[(uint64_t)blockIdx.y * out_stride + point] = Z::canon(Z::mul(s01, Z::ldp(zhi)));
    }
#endif
}
template <int NPARTS, int WPE>
__global__ __launch_bounds__(64 * NPARTS) __attribute__((amdgpu_waves_per_eu(WPE))) void t_quotient_split29_kernel(QuotientDev a, const QuotientLane* __restrict__ lanes,
                                                                                                                    Fp* __restrict__ out, uint64_t out_stride) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Z = LzOps<Fr29>;
    __shared__ uint32_t part[NPARTS - 1][9][64];         // limb planes: lane-contiguous, no bank conflicts
    const uint32_t lane = threadIdx.x & 63, which = threadIdx.x >> 6;
    const uint32_t point = blockIdx.x * 64 + lane;
    const bool live = point < a.m;
    const QuotientLane& l = lanes[blockIdx.y];
    const uint64_t own = (uint64_t)blockIdx.y * a.own_stride;
    uint32_t nxt = point + a.factor;
    if (nxt >= a.m) nxt -= a.m;
    Lz<Fr29, 1, kTqPartV> acc = Z::template relax<1, kTqPartV>(Z::zero());
    if (live) {
        if (which == 0) acc = tq_part29<0>(a, l, own, point, nxt);
        else if (which == 1) acc = tq_part29<1>(a, l, own, point, nxt);
        else if (NPARTS > 2 && which == 2) acc = tq_part29<2>(a, l, own, point, nxt);
        else if (NPARTS > 2) acc = tq_part29<3>(a, l, own, point, nxt);
    }
    if (which) {
#pragma unroll
        for (int i = 0; i < 9; ++i) part[which - 1][i][lane] = acc.v.l[i];
    }
    __syncthreads();
    if (which == 0 && live) {
        Lz<Fr29, NPARTS, NPARTS * kTqPartV> sum;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            uint32_t t = acc.v.l[i];
#pragma unroll
            for (int p = 0; p < NPARTS - 1; ++p) t += part[p][i][lane];
            sum.v.l[i] = t;
        }
        // 1 / Z_H of this point's coset class, by selects (see t_quotient_split_kernel); it stays in plain 2^256-form, so the
        // product of the 2^261-form sum with it is the quotient in the wire's 2^256-form
        const uint32_t cls = point % a.factor;
        Fp zhi = a.zhi[0];
#pragma unroll
        for (int i = 1; i < 16; ++i)
            if (cls == (uint32_t)i) zhi = a.zhi[i];
        out[(uint64_t)blockIdx.y * out_stride + point] = Z::canon(Z::mul(sum, Z::ldp(zhi)));
    }
#endif
}

struct QuotientArgsAbi {      // byte-for-byte uzk_quotient_args (include/uzkge_gpu.h): 8-byte aligned limbs
    uint32_t n, factor;
    const void* vec[56];
    uint64_t alpha[4], beta[4], gamma[4], k[5][4], anemoi_g[4], anemoi_g_inv[4], edwards_a[4];
    uint64_t z_h_inv[16][4];
};
static inline Fp fp_from_words(const uint64_t* w) { Fp r; std::memcpy(&r, w, sizeof(Fp)); return r; }

// the shared part of a launch from the C ABI's argument block; the lane part (one lane) into *lane
static int quotient_args(const QuotientArgsAbi& A, const void* d_out, QuotientDev& d, QuotientLane* lane, int* shuffle) {
    if (A.n == 0 || A.factor == 0 || A.factor > 16 || (uint64_t)A.n * A.factor >= (1ull << 31)) {
        set_error("t_quotient: need n > 0, 1 <= factor <= 16, n * factor < 2^31");
        return UZK_ERR_PARAMETER;
    }
    d.m = A.n * A.factor;
    d.factor = A.factor;
    d.own_stride = 0;
    // the 28 vectors of the "shuffle" feature (w_sel, q_shuffle_public_key, q_shuffle_generator, q_ecc) are either all
    // given or all null (a circuit without them: terms 12..18 vanish)
    auto is_shuffle_slot = [](int i) { return (i >= UZK_TQ_WSEL && i < UZK_TQ_PI) || i >= UZK_TQ_QPK; };
    int shuffle_present = 0;
    for (int i = 0; i < 56; ++i) if (is_shuffle_slot(i) && A.vec[i]) ++shuffle_present;
    if (shuffle_present != 0 && shuffle_present != 28) {
        set_error("t_quotient: the shuffle-feature vectors (w_sel, q_pk, q_g, q_ecc) must be all present or all null (%d of 28 given)", shuffle_present);
        return UZK_ERR_PARAMETER;
    }
    for (int i = 0; i < 56; ++i) {
        if (!A.vec[i] && !(is_shuffle_slot(i) && shuffle_present == 0)) { set_error("t_quotient: vec[%d] is null", i); return UZK_ERR_PARAMETER; }
        if (A.vec[i] && A.vec[i] == d_out) { set_error("t_quotient: output aliases vec[%d]", i); return UZK_ERR_PARAMETER; }
        d.vec[i] = static_cast<const Fp*>(A.vec[i]);
    }
    const Fp g = fp_from_words(A.anemoi_g);
    d.g = g; d.g_inv = fp_from_words(A.anemoi_g_inv); d.ea = fp_from_words(A.edwards_a);
    d.g2p1 = Fr::add(Fr::sqr(g), Fr::one());
    for (int i = 0; i < 16; ++i) d.zhi[i] = fp_from_words(A.z_h_inv[i]);
    if (lane) {
        Fp k[5];
        for (int j = 0; j < 5; ++j) k[j] = fp_from_words(A.k[j]);
        quotient_lane(fp_from_words(A.alpha), fp_from_words(A.beta), fp_from_words(A.gamma), k, lane);
    }
    *shuffle = shuffle_present ? 1 : 0;
    return UZK_OK;
}
void quotient_lane(const Fp& alpha, const Fp& beta, const Fp& gamma, const Fp* k, void* lane_out) {
    QuotientLane& l = *static_cast<QuotientLane*>(lane_out);
    l.ap[0] = Fr::one();
    l.ap[1] = alpha;
    for (int i = 2; i <= 16; ++i) l.ap[i] = Fr::mul(l.ap[i - 1], alpha);
    l.beta = beta; l.gamma = gamma;
    for (int j = 0; j < 5; ++j) l.bk[j] = Fr::mul(beta, k[j]);
}
size_t quotient_lane_bytes() { return sizeof(QuotientLane); }

static int quotient_launch(Ctx& c, const QuotientDev& d, int shuffle_present, const QuotientLane* d_lanes, uint32_t lanes, Fp* d_out, uint64_t out_stride) {
    KernelScope ks(c, "t_quotient");
    // small domains (the prover's 6n = 98 304 points): the term groups on separate waves per 64 points; large ones fill the chip
    // with one lane per point
    const bool split = d.m <= (1u << 19);
    if (c.tune_arith29 & 1) {           // the lazy 29-bit arithmetic (default); uzk_tune("arith29", 0): the 8 x 32-bit kernels below
        // (three waves per SIMD: 168 VGPRs with 44 spilled words beat two waves without spills, 1585 / 1557 against 1567 / 1527 proofs/s)
        if (shuffle_present && split) hipLaunchKernelGGL((t_quotient_split29_kernel<4, 3>), dim3((d.m + 63) / 64, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
        else if (split) hipLaunchKernelGGL((t_quotient_split29_kernel<2, 3>), dim3((d.m + 63) / 64, lanes), dim3(128), 0, c.stream, d, d_lanes, d_out, out_stride);
        else if (shuffle_present) hipLaunchKernelGGL(t_quotient29_kernel<true>, dim3((d.m + 255) / 256, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
        else hipLaunchKernelGGL(t_quotient29_kernel<false>, dim3((d.m + 255) / 256, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
        UZK_HIP(hipGetLastError());
        return UZK_OK;
    }
    if (shuffle_present && split) hipLaunchKernelGGL((t_quotient_split_kernel<4, 3>), dim3((d.m + 63) / 64, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
    else if (split) hipLaunchKernelGGL((t_quotient_split_kernel<2, 3>), dim3((d.m + 63) / 64, lanes), dim3(128), 0, c.stream, d, d_lanes, d_out, out_stride);
    else if (shuffle_present) hipLaunchKernelGGL(t_quotient_kernel<true>, dim3((d.m + 255) / 256, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
    else hipLaunchKernelGGL(t_quotient_kernel<false>, dim3((d.m + 255) / 256, lanes), dim3(256), 0, c.stream, d, d_lanes, d_out, out_stride);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// one lane, everything from the C ABI's argument block (uzk_t_quotient_device)
int t_quotient_run(Ctx& c, const void* args_c_abi, Fp* d_out) {
    QuotientDev d;
    QuotientLane lane;
    int shuffle = 0;
    UZK_TRY(quotient_args(*static_cast<const QuotientArgsAbi*>(args_c_abi), d_out, d, &lane, &shuffle));
    UZK_TRY(c.poly_args.reserve(sizeof lane));
    UZK_HIP(hipMemcpyAsync(c.poly_args.p, &lane, sizeof lane, hipMemcpyHostToDevice, c.stream));    // ordinary memory: staged before the call returns
    return quotient_launch(c, d, shuffle, c.poly_args.as<QuotientLane>(), 1, d_out, 0);
}

// `lanes` proofs of a lockstep batch (prover round 3): the argument block names lane 0's vectors, lane b's own ten lie
// own_stride * b elements behind them, its challenges at d_lanes[b] (quotient_lane), its output at d_out + b * out_stride.
int t_quotient_lanes(Ctx& c, const void* args_c_abi, uint64_t own_stride, const void* d_lanes, uint32_t lanes, Fp* d_out, uint64_t out_stride) {
    QuotientDev d;
    int shuffle = 0;
    UZK_TRY(quotient_args(*static_cast<const QuotientArgsAbi*>(args_c_abi), d_out, d, nullptr, &shuffle));
    d.own_stride = own_stride;
    return quotient_launch(c, d, shuffle, static_cast<const QuotientLane*>(d_lanes), lanes, d_out, out_stride);
}

// ---------------------------------------------------------------------------------------------
// Opening quotient (uzk_open_quotient_device): batch_prove's polynomial work,
// uzkge/src/poly_commit/pcs.rs:119-135:  h = sum_k alpha^k (p_k - p_k(z)),  q = h / (X - z)
// (div_rem, field_polynomial.rs:519-550, with the monic divisor X - z: synthetic division
//  q_{i-1} = h_i + z q_i, a linear recurrence from the top coefficient down).
// The recurrence is evaluated as a blocked scan: 16 coefficients per lane by Horner, the carries
// between lanes and between 4096-coefficient blocks by Hillis-Steele steps whose multipliers are
// the same for every lane (x_t += z^(16 * 2^k) x_(t + 2^k)).  n <= 2^20 coefficients per polynomial.
// ---------------------------------------------------------------------------------------------

constexpr uint32_t kLincombMax = 64;
struct LincombArgs {                      // by value: 2.8 KB of kernel arguments, no staging copy and no synchronisation
    const Fp* p[kLincombMax];
    uint32_t len[kLincombMax];
    Fp scalars[kLincombMax];
    uint32_t count;
};
// GS lanes per coefficient: lane `sub` takes the polynomials k = sub, sub + GS, ... and a shuffle tree adds the GS partial sums.
// At the prover's size (n + 3 = 16 387 coefficients: a quarter of a wave per SIMD) one lane per coefficient is a chain of
// `count` dependent products on an idle chip; four lanes make it a quarter as long (r_poly's 43 polynomials: 35 -> 12 us).
template <int GS>
__global__ __launch_bounds__(256) void poly_lincomb_kernel(LincombArgs a, Fp* __restrict__ out, uint64_t out_len, int accumulate) {
    const Fp* scalars = a.scalars;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = gid / GS;
    const uint32_t sub = (uint32_t)(gid % GS);
    const bool live = j < out_len;                    // whole groups leave the range together; dead lanes still take part in the shuffles
    Fp acc = (accumulate && live && sub == 0) ? out[j] : Fr::zero();
    if (live)
        for (uint32_t k = sub; k < a.count; k += GS)
            if (j < a.len[k]) acc = Fr::add(acc, Fr::mul(scalars[k], a.p[k][j]));
    if constexpr (GS > 1) {
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) {
            Fp q;
#pragma unroll
            for (int w = 0; w < 8; ++w) q.v[w] = (uint32_t)__shfl_down((int)acc.v[w], o);
            if (sub + (uint32_t)o < (uint32_t)GS) acc = Fr::add(acc, q);
        }
    }
    if (live && sub == 0) out[j] = acc;
}
static void launch_lincomb(Ctx& c, const LincombArgs& a, Fp* d_out, uint64_t out_len, int accumulate) {
    if (out_len <= (1ull << 17) && a.count >= 8)
        hipLaunchKernelGGL(poly_lincomb_kernel<4>, dim3((unsigned)((out_len * 4 + 255) / 256)), dim3(256), 0, c.stream, a, d_out, out_len, accumulate);
    else
        hipLaunchKernelGGL(poly_lincomb_kernel<1>, dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, c.stream, a, d_out, out_len, accumulate);
}

// s_i = h_i + z s_(i+1) within each block of 256 PER coefficients, carry-in 0; block_first[b] = s at the block's lowest index.
// zp[k] = z^(PER * 2^k), k < 8.
struct DivPows { Fp z; Fp zp[8]; };
// PER coefficients per lane (16; 4 for n <= 2^16: the prover's openings get 16 workgroups and a dependent chain of
// 16 products instead of 40), 256 PER per block; the host builds zp / ztab for the same PER.
template <int kDivPer>
__global__ __launch_bounds__(256) void open_div_block_kernel(const Fp* __restrict__ h, uint64_t n, DivPows pw,
                                                             Fp* __restrict__ s_out, Fp* __restrict__ block_first) {
    constexpr int kDivBlock = 256 * kDivPer;
    __shared__ Fp sh[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t lo = (uint64_t)blockIdx.x * kDivBlock + (uint64_t)tid * kDivPer;
    Fp v[kDivPer];
    Fp run = Fr::zero();
#pragma unroll
    for (int e = kDivPer - 1; e >= 0; --e) {             // local Horner, top index first
        const uint64_t i = lo + e;
        run = Fr::mul(run, pw.z);
        if (i < n) run = Fr::add(run, h[i]);
        v[e] = run;
    }
    // x_t = local value at the lane's lowest index; carry from the lanes above: x_t += z^16 x_(t+1) ...
    sh[tid] = run;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t off = 1u << k;
        Fp t = (tid + off < 256) ? sh[tid + off] : Fr::zero();
        __syncthreads();
        if (tid + off < 256) sh[tid] = Fr::add(sh[tid], Fr::mul(pw.zp[k], t));
        __syncthreads();
    }
    // carry into this lane = full value at the lowest index of the lane above
    Fp carry = (tid + 1 < 256) ? sh[tid + 1] : Fr::zero();
    Fp zp = pw.z;                                        // z^(hi - i) for i = hi-1 .. lo
#pragma unroll
    for (int e = kDivPer - 1; e >= 0; --e) {
        const uint64_t i = lo + e;
        if (i < n) s_out[i] = Fr::add(v[e], Fr::mul(zp, carry));
        zp = Fr::mul(zp, pw.z);
    }
    if (tid == 0) block_first[blockIdx.x] = sh[0];
}
// carries between blocks (nblocks <= 256): c_b = sum_{b' > b} first[b'] z^(4096 (b' - b - 1)); zb[k] = z^(4096 * 2^k)
// Also leaves ztab[t] = (z^PER)^t, t <= 256, for the apply kernel (no host table, no upload, no synchronisation): the lower 256
// lanes run the carry scan, the upper 256 build the table in LDS by doubling with the multipliers pw.zp[k] = z^(PER 2^k) --
// two independent chains of eight dependent products side by side.
__global__ __launch_bounds__(512) void open_div_carry_kernel(const Fp* __restrict__ block_first, uint32_t nblocks, DivPows pw, DivPows pb,
                                                             Fp* __restrict__ carry, Fp* __restrict__ ztab) {
    __shared__ Fp sh[256], zt[256];
    const uint32_t t = threadIdx.x & 255;
    const bool table = threadIdx.x >= 256;
    if (!table) sh[t] = t < nblocks ? block_first[t] : Fr::zero();
    else if (t == 0) zt[0] = Fr::one();
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t off = 1u << k;
        Fp x = Fr::zero();
        if (!table && t + off < 256) x = sh[t + off];
        __syncthreads();
        if (!table) { if (t + off < 256) sh[t] = Fr::add(sh[t], Fr::mul(pb.zp[k], x)); }
        else if (t < off) zt[off + t] = Fr::mul(zt[t], pw.zp[k]);
        __syncthreads();
    }
    if (!table) { if (t < nblocks) carry[t] = (t + 1 < 256) ? sh[t + 1] : Fr::zero(); }
    else {
        ztab[t] = zt[t];
        if (t == 0) ztab[256] = pb.zp[0];               // z^(256 PER)
    }
}
// q[i-1] = s_i + z^(block_hi - i) carry[block]  for 1 <= i < n ;  q[n-1 .. q_cap) = 0 ; ztab16[t] = z^(PER t), t <= 256
template <int kDivPer>
__global__ __launch_bounds__(256) void open_div_apply_kernel(const Fp* __restrict__ s, const Fp* __restrict__ carry,
                                                             const Fp* __restrict__ ztab16, Fp z, uint64_t n, uint64_t q_cap,
                                                             Fp* __restrict__ q) {
    constexpr int kDivBlock = 256 * kDivPer;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q_cap) return;
    if (i >= n) { q[i] = Fr::zero(); return; }
    if (i == 0) { q[n - 1] = Fr::zero(); return; }
    const uint64_t blk = i / kDivBlock;
    const uint32_t dist = (uint32_t)((blk + 1) * kDivBlock - i);      // 1 .. 4096
    Fp v = s[i];
    const Fp cb = carry[blk];
    if (!Fr::is_zero(cb)) {
        // z^dist = ztab16[dist / 16] * z^(dist % 16)
        Fp zp = ztab16[dist / kDivPer];
        for (uint32_t r = 0; r < dist % kDivPer; ++r) zp = Fr::mul(zp, z);
        v = Fr::add(v, Fr::mul(zp, cb));
    }
    q[i - 1] = v;
}


// h = sum_k alpha^k p_k over `count` device-resident polynomials of lens[k] coefficients; q = h div (X - z) written to d_q:
// hlen - 1 coefficients (hlen = the longest polynomial) followed by zeros up to q_cap.  The constant sum_k alpha^k p_k(z)
// that batch_prove subtracts first (pcs.rs:124-130) only changes h_0, i.e. the remainder -- never q -- so no evaluation is
// needed for the quotient; evals_host (optional) receives p_k(z) all the same (one more launch and a synchronisation).
// Without it the call is asynchronous: the power tables are built on the device.
int open_quotient_ptrs(Ctx& c, const void* const* d_polys, const uint64_t* lens, uint32_t count, const Fp& z, const Fp& alpha, Fp* d_q,
                       uint64_t q_cap, Fp* evals_host) {
    if (count == 0) { set_error("open_quotient: need at least one polynomial"); return UZK_ERR_PARAMETER; }
    uint64_t n = 0;
    for (uint32_t k = 0; k < count; ++k) {
        if (lens[k] && !d_polys[k]) { set_error("open_quotient: polynomial %u is null", k); return UZK_ERR_PARAMETER; }
        if (d_polys[k] == d_q) { set_error("open_quotient: output aliases polynomial %u", k); return UZK_ERR_PARAMETER; }
        n = std::max(n, lens[k]);
    }
    if (n == 0) { set_error("open_quotient: every polynomial is empty"); return UZK_ERR_PARAMETER; }
    if (n > (1ull << 20)) { set_error("open_quotient: n = %llu exceeds 2^20 coefficients", (unsigned long long)n); return UZK_ERR_PARAMETER; }
    if (q_cap < n) { set_error("open_quotient: d_q holds %llu elements, the quotient needs %llu", (unsigned long long)q_cap, (unsigned long long)n); return UZK_ERR_PARAMETER; }
    if (evals_host) {
        std::vector<uint32_t> pt(count, 0u);
        UZK_TRY(poly_eval_ptrs(c, d_polys, lens, pt.data(), count, &z, 1, evals_host));     // p_k(z), returned to the caller
    }
    const int per = (n <= (1ull << 16)) ? 4 : 16;       // coefficients per lane of the division kernels
    const int log_per = per == 4 ? 2 : 4;
    const uint64_t div_block = 256ull * per;
    const uint32_t nblocks = (uint32_t)((n + div_block - 1) / div_block);
    // layout: h[n] | s[n] | block_first[256] | carry[256] | ztab[257]
    UZK_TRY(c.open_tmp.reserve((2 * n + 256 + 256 + 257) * sizeof(Fp)));
    Fp* d_h = c.open_tmp.as<Fp>();
    Fp* d_s = d_h + n;
    Fp* d_first = d_s + n;
    Fp* d_carry = d_first + 256;
    Fp* d_ztab = d_carry + 256;
    DivPows pw, pb;
    pw.z = z; pb.z = z;
    {
        Fp p = z;
        for (int i = 0; i < log_per; ++i) p = Fr::sqr(p);                   // z^per
        for (int k = 0; k < 8; ++k) { pw.zp[k] = p; p = Fr::sqr(p); }       // z^(per * 2^k); p ends as z^(256 per): one block
        for (int k = 0; k < 8; ++k) { pb.zp[k] = p; p = Fr::sqr(p); }
    }
    KernelScope ks(c, "open_quotient");
    Fp a = Fr::one();
    for (uint32_t k0 = 0; k0 < count; k0 += kLincombMax) {                   // h = sum alpha^k p_k, 64 polynomials per launch
        LincombArgs la;
        la.count = std::min<uint32_t>(kLincombMax, count - k0);
        for (uint32_t k = 0; k < la.count; ++k) {
            la.p[k] = static_cast<const Fp*>(d_polys[k0 + k]);
            la.len[k] = (uint32_t)lens[k0 + k];
            la.scalars[k] = a;
            a = Fr::mul(a, alpha);
        }
        launch_lincomb(c, la, d_h, n, k0 ? 1 : 0);
    }
    if (per == 4) hipLaunchKernelGGL(open_div_block_kernel<4>, dim3(nblocks), dim3(256), 0, c.stream, d_h, n, pw, d_s, d_first);
    else hipLaunchKernelGGL(open_div_block_kernel<16>, dim3(nblocks), dim3(256), 0, c.stream, d_h, n, pw, d_s, d_first);
    hipLaunchKernelGGL(open_div_carry_kernel, dim3(1), dim3(512), 0, c.stream, d_first, nblocks, pw, pb, d_carry, d_ztab);
    if (per == 4) hipLaunchKernelGGL(open_div_apply_kernel<4>, dim3((unsigned)((q_cap + 255) / 256)), dim3(256), 0, c.stream, d_s, d_carry, d_ztab, z, n, q_cap, d_q);
    else hipLaunchKernelGGL(open_div_apply_kernel<16>, dim3((unsigned)((q_cap + 255) / 256)), dim3(256), 0, c.stream, d_s, d_carry, d_ztab, z, n, q_cap, d_q);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// The contiguous form (batch polynomials of n coefficients each): the same through a pointer list.
int open_quotient_run(Ctx& c, const Fp* d_polys, uint64_t n, uint32_t batch, const Fp& z, const Fp& alpha, Fp* d_q,
                      Fp* evals_host) {
    if (batch == 0 || n == 0) { set_error("open_quotient: need batch > 0 and n > 0"); return UZK_ERR_PARAMETER; }
    if (batch > 4096) { set_error("open_quotient: batch %u exceeds 4096", batch); return UZK_ERR_PARAMETER; }
    std::vector<const void*> ptrs(batch);
    std::vector<uint64_t> lens(batch, n);
    for (uint32_t k = 0; k < batch; ++k) ptrs[k] = d_polys + (uint64_t)k * n;
    if (evals_host && batch > kEvalMaxPolys) {           // more evaluations than one launch holds: the batched evaluator
        UZK_TRY(poly_eval_batch(c, d_polys, n, batch, z, evals_host));
        UZK_TRY(open_quotient_ptrs(c, ptrs.data(), lens.data(), batch, z, alpha, d_q, n, nullptr));
    } else {
        UZK_TRY(open_quotient_ptrs(c, ptrs.data(), lens.data(), batch, z, alpha, d_q, n, evals_host));
    }
    UZK_HIP(hipStreamSynchronize(c.stream));             // this entry point's contract: q is in d_q when it returns
    return UZK_OK;
}

// ---------------------------------------------------------------------------------------------
// The fold modulo X^N - 1 that precedes every Lagrange-basis commit of a polynomial with more than N
// coefficients (batch_prove, uzkge/src/poly_commit/pcs.rs:137-156; split_t_and_commit,
// uzkge/src/plonk/helpers.rs:1366-1383):  blinds[i] = -coefs[N + i];  out[i] = coefs[i] - blinds[i] for
// i < len - N, out[i] = coefs[i] otherwise.  out has N elements (zero beyond len when len < N).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_blinds_kernel(const Fp* __restrict__ coefs, uint64_t len, uint64_t N,
                                                          Fp* __restrict__ out, Fp* __restrict__ blinds) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    Fp v = i < len ? coefs[i] : Fr::zero();
    if (N + i < len) {
        const Fp hi = coefs[N + i];
        blinds[i] = Fr::neg(hi);
        v = Fr::add(v, hi);
    }
    out[i] = v;
}

int fold_blinds_run(Ctx& c, const Fp* d_coefs, uint64_t len, uint64_t N, Fp* d_out, Fp* blinds_host) {
    if (N == 0 || len > 2 * N) { set_error("fold_blinds: need N > 0 and len <= 2N (len %llu, N %llu)", (unsigned long long)len, (unsigned long long)N); return UZK_ERR_PARAMETER; }
    const uint64_t nb = len > N ? len - N : 0;
    UZK_TRY(c.poly_tmp2.reserve((size_t)(nb ? nb : 1) * sizeof(Fp)));
    {
        KernelScope ks(c, "fold_blinds");
        hipLaunchKernelGGL(fold_blinds_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c.stream, d_coefs, len, N, d_out,
                           c.poly_tmp2.as<Fp>());
    }
    UZK_HIP(hipGetLastError());
    if (nb) {
        UZK_HIP(hipMemcpyAsync(blinds_host, c.poly_tmp2.p, nb * sizeof(Fp), hipMemcpyDeviceToHost, c.stream));
        UZK_HIP(hipStreamSynchronize(c.stream));
    }
    return UZK_OK;
}

// ---------------------------------------------------------------------------------------------
// out[j] = sum_k scalar_k * p_k[j]  (p_k[j] = 0 beyond its length): the shape of r_poly
// (uzkge/src/plonk/helpers.rs:681-999: ~43 polynomials times scalars built from the evaluations and challenges --
// those O(1) scalar formulas stay with the caller, exactly where the reference computes them).
// ---------------------------------------------------------------------------------------------
int poly_lincomb_run(Ctx& c, const void* const* d_polys, const uint64_t* lens, const Fp* scalars_host, uint32_t count, Fp* d_out,
                     uint64_t out_len) {
    if (count == 0 || count > kLincombMax) { set_error("poly_lincomb: 1 <= count <= %u", kLincombMax); return UZK_ERR_PARAMETER; }
    LincombArgs a;
    a.count = count;
    for (uint32_t k = 0; k < count; ++k) {
        if (!d_polys[k] && lens[k]) { set_error("poly_lincomb: polynomial %u is null", k); return UZK_ERR_PARAMETER; }
        if (d_polys[k] == d_out) { set_error("poly_lincomb: output aliases polynomial %u", k); return UZK_ERR_PARAMETER; }
        if (lens[k] >= (1ull << 32)) { set_error("poly_lincomb: polynomial %u too long", k); return UZK_ERR_PARAMETER; }
        a.p[k] = static_cast<const Fp*>(d_polys[k]);
        a.len[k] = (uint32_t)lens[k];
    }
    if (out_len == 0) return UZK_OK;
    for (uint32_t k = 0; k < count; ++k) a.scalars[k] = scalars_host[k];
    {
        KernelScope ks(c, "poly_lincomb");
        launch_lincomb(c, a, d_out, out_len, 0);
    }
    UZK_HIP(hipGetLastError());
    return UZK_OK;                                 // asynchronous on the library stream
}

// hide_polynomial (uzkge/src/plonk/helpers.rs:139-158) on device-resident coefficients:
// coefs[i] += blind_i, coefs[zeroing_degree + i] -= blind_i  (adds (b_0 + b_1 X + ..)(X^zeroing_degree - 1)).
struct HideArgs { Fp blinds[16]; };
__global__ void poly_hide_kernel(Fp* __restrict__ coefs, uint64_t zeroing_degree, HideArgs h, uint32_t count) {
    const uint32_t i = threadIdx.x;
    if (i >= count) return;
    coefs[i] = Fr::add(coefs[i], h.blinds[i]);
    coefs[zeroing_degree + i] = Fr::sub(coefs[zeroing_degree + i], h.blinds[i]);
}
int poly_hide_run(Ctx& c, Fp* d_coefs, uint64_t len, const Fp* blinds_host, uint32_t hiding_degree, uint64_t zeroing_degree) {
    if (hiding_degree == 0) return UZK_OK;
    if (hiding_degree > 16 || hiding_degree > zeroing_degree || zeroing_degree + hiding_degree > len) {
        set_error("hide_polynomial: need hiding_degree <= min(16, zeroing_degree) and zeroing_degree + hiding_degree <= len");
        return UZK_ERR_PARAMETER;
    }
    HideArgs h;                                    // the blinds travel as kernel arguments: no staging copy, no synchronisation
    for (uint32_t i = 0; i < 16; ++i) h.blinds[i] = i < hiding_degree ? blinds_host[i] : Fr::zero();
    hipLaunchKernelGGL(poly_hide_kernel, dim3(1), dim3(64), 0, c.stream, d_coefs, zeroing_degree, h, hiding_degree);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// The evaluations of the PI polynomial (pi_poly, helpers.rs:111-131): dst_b[idx[j]] = val[b][j] for `batch` vectors dst_stride
// apart that the caller has zeroed.  idx / val sit in pinned host memory the kernel reads directly (a few hundred entries).
// The caller lists a repeated index once (the reference's find_position takes the first occurrence), so writes never collide.
__global__ __launch_bounds__(256) void poly_scatter_kernel(Fp* __restrict__ dst, uint64_t dst_stride, const uint32_t* __restrict__ idx,
                                                           const Fp* __restrict__ val, uint32_t count) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    dst[(uint64_t)blockIdx.y * dst_stride + idx[j]] = val[(uint64_t)blockIdx.y * count + j];
}
int poly_scatter_run(Ctx& c, Fp* d_dst, uint64_t dst_stride, const uint32_t* idx_pinned, const Fp* val_pinned, uint32_t count, uint32_t batch) {
    if (count == 0 || batch == 0) return UZK_OK;
    KernelScope ks(c, "poly_scatter");
    hipLaunchKernelGGL(poly_scatter_kernel, dim3((count + 255) / 256, batch), dim3(256), 0, c.stream, d_dst, dst_stride, idx_pinned, val_pinned, count);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- batched forms for the prover's rounds: one launch where the single-polynomial entry points take one per polynomial ----

// hide_polynomial for `count` polynomials stored `stride` elements apart, each holding len_in meaningful coefficients:
// the reference resizes the coefficient vector with zeros up to zeroing_degree + hiding_degree before it adds the blinds
// (helpers.rs:146-156), so slots [len_in, zeroing_degree + hiding_degree) are WRITTEN (zero, minus the blind), not read --
// whatever an earlier proof left there does not matter.  One workgroup per polynomial.
constexpr uint32_t kHideBatchMax = 64;       // count * hiding_degree
struct HideBatchArgs { Fp blinds[kHideBatchMax]; };
__global__ __launch_bounds__(256) void poly_hide_batch_kernel(Fp* __restrict__ coefs, uint64_t stride, uint64_t len_in,
                                                              uint64_t zeroing_degree, uint32_t hd, HideBatchArgs h) {
    Fp* c = coefs + (uint64_t)blockIdx.x * stride;
    const Fp* bl = h.blinds + (size_t)blockIdx.x * hd;
    const uint64_t top = zeroing_degree + hd;
    for (uint64_t j = len_in + threadIdx.x; j < zeroing_degree; j += blockDim.x) c[j] = Fr::zero();   // the resize's zeros below the top slots
    const uint32_t i = threadIdx.x;
    if (i < hd) {
        // low and top slots are distinct (hd <= zeroing_degree), so the two updates of a lane never touch the same element
        const Fp lo = i < len_in ? c[i] : Fr::zero();
        const uint64_t t = zeroing_degree + i;
        const Fp hi = t < len_in ? c[t] : Fr::zero();
        c[i] = Fr::add(lo, bl[i]);
        c[t] = Fr::sub(hi, bl[i]);
    }
    (void)top;
}
int poly_hide_batch_run(Ctx& c, Fp* d_coefs, uint64_t stride, uint64_t len_in, uint32_t count, const Fp* blinds_host, uint32_t hd,
                        uint64_t zeroing_degree) {
    if (count == 0 || hd == 0) return UZK_OK;
    if (hd > zeroing_degree || (uint64_t)count * hd > kHideBatchMax || zeroing_degree + hd > stride || len_in > stride) {
        set_error("hide_polynomial_batch: need hiding_degree <= zeroing_degree, count * hiding_degree <= %u, zeroing_degree + hiding_degree <= stride", kHideBatchMax);
        return UZK_ERR_PARAMETER;
    }
    if (len_in < hd) { set_error("hide_polynomial_batch: a polynomial shorter than its hiding degree"); return UZK_ERR_PARAMETER; }
    HideBatchArgs h;
    for (uint32_t i = 0; i < kHideBatchMax; ++i) h.blinds[i] = i < count * hd ? blinds_host[i] : Fr::zero();
    KernelScope ks(c, "poly_hide");
    hipLaunchKernelGGL(poly_hide_batch_kernel, dim3(count), dim3(256), 0, c.stream, d_coefs, stride, len_in, zeroing_degree, hd, h);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// The fold modulo X^N - 1 (pcs.rs:137-156, helpers.rs:1366-1383) for `batch` polynomials at once, and the scalars of
// apply_blind_factors (kzg_poly_commitment.rs:299-313) left ON THE DEVICE in the layout the commit consumes:
//   out[b][i]  = c_b[i] + c_b[N + i]  (i < len_b - N),  c_b[i]  (len_b - N <= i < min(len_b, N)),  0 beyond
//   tail[b][i] = -c_b[N + i] = blind_i,  tail[b][K + i] = c_b[N + i] = -blind_i   (i < len_b - N; zero up to K = tail_n / 2)
// so that fold -> FFT(N) -> MSM(evals || tail) runs without the blinds ever visiting the host.
constexpr uint32_t kFoldBatchMax = 16;
struct FoldBatchArgs { uint32_t len[kFoldBatchMax]; };
__global__ __launch_bounds__(256) void fold_blinds_batch_kernel(const Fp* __restrict__ polys, uint64_t in_stride, FoldBatchArgs lens,
                                                                uint64_t N, Fp* __restrict__ out, uint64_t out_stride,
                                                                Fp* __restrict__ tail, uint32_t tail_n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = blockIdx.y;
    if (i >= N) return;
    const uint64_t len = lens.len[b];
    const Fp* c = polys + (uint64_t)b * in_stride;
    Fp v = i < len ? c[i] : Fr::zero();
    const uint32_t K = tail_n / 2;
    if (N + i < len) {
        const Fp hi = c[N + i];
        v = Fr::add(v, hi);
        tail[(uint64_t)b * tail_n + i] = Fr::neg(hi);
        tail[(uint64_t)b * tail_n + K + i] = hi;
    } else if (i < K) {
        tail[(uint64_t)b * tail_n + i] = Fr::zero();
        tail[(uint64_t)b * tail_n + K + i] = Fr::zero();
    }
    out[(uint64_t)b * out_stride + i] = v;
}
int fold_blinds_batch_run(Ctx& c, const Fp* d_polys, uint64_t in_stride, const uint64_t* lens_host, uint64_t N, uint32_t batch, Fp* d_out,
                          uint64_t out_stride, Fp* d_tail, uint32_t tail_n, Fp* blinds_host) {
    if (batch == 0) return UZK_OK;
    if (batch > kFoldBatchMax) { set_error("fold_blinds_batch: batch %u exceeds %u", batch, kFoldBatchMax); return UZK_ERR_PARAMETER; }
    if (N == 0 || (tail_n & 1) || out_stride < N) { set_error("fold_blinds_batch: need N > 0, an even tail_n and out_stride >= N"); return UZK_ERR_PARAMETER; }
    FoldBatchArgs la{};
    for (uint32_t b = 0; b < batch; ++b) {
        const uint64_t len = lens_host[b];
        if (len > 2 * N || len > in_stride || (len > N && len - N > tail_n / 2)) {
            set_error("fold_blinds_batch: polynomial %u: need len <= min(2N, in_stride) and len - N <= tail_n / 2 (len %llu, N %llu, tail_n %u)", b,
                      (unsigned long long)len, (unsigned long long)N, tail_n);
            return UZK_ERR_PARAMETER;
        }
        la.len[b] = (uint32_t)len;
    }
    {
        KernelScope ks(c, "fold_blinds");
        hipLaunchKernelGGL(fold_blinds_batch_kernel, dim3((unsigned)((N + 255) / 256), batch), dim3(256), 0, c.stream, d_polys, in_stride, la, N,
                           d_out, out_stride, d_tail, tail_n);
    }
    UZK_HIP(hipGetLastError());
    if (blinds_host && tail_n) {        // optional: the blinds as the reference's Vec (tail_n / 2 per polynomial, zero padded); synchronises
        const uint32_t K = tail_n / 2;
        UZK_HIP(hipMemcpy2DAsync(blinds_host, K * sizeof(Fp), d_tail, tail_n * sizeof(Fp), K * sizeof(Fp), batch, hipMemcpyDeviceToHost, c.stream));
        UZK_HIP(hipStreamSynchronize(c.stream));
    }
    return UZK_OK;
}

// The split of t in split_t_and_commit (helpers.rs:1335-1363), `chunk` = the reference's `n` argument (n_constraints + 2):
//   i < last: coefs = t[i chunk .. (i+1) chunk) (zero beyond t_len), resized to chunk + 1; coefs[chunk] += rand_i; coefs[0] -= rand_(i-1)
//   last    : coefs = t[last chunk .. t_len)  (or [-rand_(last-1)] when that range is empty);   coefs[0] -= rand_(last-1)
// written to d_chunks + i * chunk_stride, zero padded to chunk_stride.  rand_(n_chunks-1) is drawn but unused, as in the reference.
constexpr uint32_t kSplitMax = 8;
struct SplitArgs { Fp rands[kSplitMax]; };
__global__ __launch_bounds__(256) void split_t_kernel(const Fp* __restrict__ t, uint64_t t_len, uint64_t chunk, uint32_t n_chunks,
                                                      SplitArgs r, Fp* __restrict__ out, uint64_t chunk_stride) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y;
    if (j >= chunk_stride) return;
    const bool last = i + 1 == n_chunks;
    const uint64_t start = (uint64_t)i * chunk;
    const uint64_t end = last ? t_len : min(t_len, start + chunk);
    Fp v = (start + j < end) ? t[start + j] : Fr::zero();
    if (!last && j == chunk) v = Fr::add(v, r.rands[i]);
    if (j == 0 && i > 0) v = Fr::sub(v, r.rands[i - 1]);
    out[(uint64_t)i * chunk_stride + j] = v;
}
int split_t_run(Ctx& c, const Fp* d_t, uint64_t t_len, uint64_t chunk, uint32_t n_chunks, const Fp* rands_host, Fp* d_chunks,
                uint64_t chunk_stride, uint64_t* lens_out) {
    if (n_chunks == 0 || n_chunks > kSplitMax || chunk == 0) { set_error("split_t: 1 <= n_chunks <= %u and chunk > 0", kSplitMax); return UZK_ERR_PARAMETER; }
    const uint64_t last_start = (uint64_t)(n_chunks - 1) * chunk;
    const uint64_t last_len = t_len > last_start ? t_len - last_start : 1;         // an empty last range becomes [-prev]
    if (chunk_stride < chunk + 1 || chunk_stride < last_len) { set_error("split_t: chunk_stride too small"); return UZK_ERR_PARAMETER; }
    SplitArgs r;
    for (uint32_t i = 0; i < kSplitMax; ++i) r.rands[i] = i < n_chunks ? rands_host[i] : Fr::zero();
    {
        KernelScope ks(c, "split_t");
        hipLaunchKernelGGL(split_t_kernel, dim3((unsigned)((chunk_stride + 255) / 256), n_chunks), dim3(256), 0, c.stream, d_t, t_len, chunk,
                           n_chunks, r, d_chunks, chunk_stride);
    }
    UZK_HIP(hipGetLastError());
    if (lens_out) {
        for (uint32_t i = 0; i + 1 < n_chunks; ++i) lens_out[i] = chunk + 1;
        lens_out[n_chunks - 1] = last_len;
    }
    return UZK_OK;
}

// FpPolynomial::from_coefs trims trailing zero coefficients (field_polynomial.rs:86-90) and the prover's control flow reads
// the result: t's coefs.len() decides the split (helpers.rs:1333), q.degree() the fold of batch_prove (pcs.rs:138).  On the
// device the polynomials keep their allocated length, so this gives the trimmed one: out[b] = 1 + the highest index below
// lens[b] holding a non-zero coefficient, 0 for the zero polynomial.  One launch for the batch; synchronises (the answer is
// control flow).
constexpr uint32_t kTrimMax = 16;
struct TrimArgs { uint32_t len[kTrimMax]; };
__global__ __launch_bounds__(256) void poly_trimmed_len_kernel(const Fp* __restrict__ polys, uint64_t stride, TrimArgs lens,
                                                               unsigned long long* __restrict__ out,
                                                               unsigned long long* __restrict__ clear_for_next) {
    __shared__ unsigned long long top_of_block;
    const uint32_t b = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && b == 0 && threadIdx.x < kTrimMax) clear_for_next[threadIdx.x] = 0;     // the set the NEXT call accumulates into
    if (threadIdx.x == 0) top_of_block = 0;
    __syncthreads();
    const bool nz = i < lens.len[b] && !Fr::is_zero(polys[(uint64_t)b * stride + i]);
    // highest non-zero index of the wave -> of the workgroup (LDS) -> one global atomic per workgroup that has any, and
    // only if it can raise the current value (most workgroups of a dense polynomial lie below it)
    const unsigned long long mask = __ballot(nz);
    if (mask != 0 && (threadIdx.x & 63) == 0) atomicMax(&top_of_block, (unsigned long long)((i & ~63ull) + (63 - __clzll(mask)) + 1));
    __syncthreads();
    if (threadIdx.x == 0 && top_of_block > out[b]) atomicMax(&out[b], top_of_block);
}
// out_host: ordinary memory (sync: the values are there on return) or, with sync == false, pinned memory of uzk_host_alloc
// that the stream fills in order -- the caller reads it after its next synchronising call.
int poly_trimmed_len_run(Ctx& c, const Fp* d_polys, uint64_t stride, const uint64_t* lens_host, uint32_t batch, uint64_t* out_host, bool sync) {
    if (batch == 0) return UZK_OK;
    if (batch > kTrimMax) { set_error("poly_trimmed_len: batch %u exceeds %u", batch, kTrimMax); return UZK_ERR_PARAMETER; }
    TrimArgs la{};
    uint64_t max_len = 0;
    for (uint32_t b = 0; b < batch; ++b) {
        if (lens_host[b] >= (1ull << 32) || (batch > 1 && lens_host[b] > stride)) { set_error("poly_trimmed_len: polynomial %u: bad length", b); return UZK_ERR_PARAMETER; }
        la.len[b] = (uint32_t)lens_host[b];
        max_len = std::max(max_len, lens_host[b]);
    }
    if (max_len == 0) { for (uint32_t b = 0; b < batch; ++b) out_host[b] = 0; return UZK_OK; }
    if (c.poly_cnt.cap < 4096) {
        UZK_TRY(c.poly_cnt.reserve(4096));
        UZK_HIP(hipMemsetAsync(c.poly_cnt.p, 0, c.poly_cnt.cap, c.stream));
    }
    // Result slots beyond the evaluation counters: two sets used alternately.  Both start zero (the buffer is cleared when it is
    // allocated); every launch accumulates into one set and clears the other for the call after it -- whose copy-out of that
    // set, issued by the call before this one, is behind it on the stream.  No fill launch.
    c.trim_flip ^= 1u;
    unsigned long long* sets = reinterpret_cast<unsigned long long*>(static_cast<char*>(c.poly_cnt.p) + 2048);
    unsigned long long* d_res = sets + c.trim_flip * kTrimMax;
    {
        KernelScope ks(c, "poly_trimmed_len");
        hipLaunchKernelGGL(poly_trimmed_len_kernel, dim3((unsigned)((max_len + 255) / 256), batch), dim3(256), 0, c.stream, d_polys, stride, la, d_res,
                           sets + (c.trim_flip ^ 1u) * kTrimMax);
    }
    UZK_HIP(hipGetLastError());
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "result words are copied as they are");
    UZK_HIP(hipMemcpyAsync(out_host, d_res, batch * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
    if (sync) UZK_HIP(hipStreamSynchronize(c.stream));
    return UZK_OK;
}

int poly_eval_batch_host(Ctx& c, const Fp* coefs_host, uint64_t n, uint32_t batch, const Fp& x, Fp* out_host) {
    if (batch == 0) return UZK_OK;
    const size_t bytes = (size_t)n * batch * sizeof(Fp);
    if (bytes) {
        UZK_TRY(c.poly_io.reserve(bytes));
        UZK_HIP(hipMemcpyAsync(c.poly_io.p, coefs_host, bytes, hipMemcpyHostToDevice, c.stream));
    }
    return poly_eval_batch(c, c.poly_io.as<Fp>(), n, batch, x, out_host);
}

void poly_free(Ctx& c) {
    c.poly_tmp.release();
    c.poly_tmp2.release();
    c.poly_io.release();
    c.zpoly_tmp.release();
    c.open_tmp.release();
    c.poly_args.release();
    c.poly_cnt.release();
    if (c.poly_host) { (void)hipHostFree(c.poly_host); c.poly_host = nullptr; c.poly_host_cap = 0; }
}

}  // namespace uzk
