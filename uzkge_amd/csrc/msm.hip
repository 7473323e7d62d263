// BN254 G1 multi-scalar multiplication (Pippenger bucket method) for gfx950.
// Replaces `G1Projective::msm(&points_raw, &coefs)` at
//   uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290
// (the `normalize_batch` of :287-288 disappears: the SRS is registered once, affine, in HBM).
//
// Pipeline (all on the library stream; see DESIGN.md "MSM"):
//   1 msm_digits      scalar -> canonical (one Montgomery mul by 1) -> signed c-bit digits
//                     d_w in [-2^(c-1), 2^(c-1)], stored [window][i] (coalesced both ways); at large n the same kernel
//                     leaves the first sort pass's histograms
//   2 sort            counting sort of the (window, bucket) keys, high bits first.  Two-pass sorts with 4-byte packed
//                     entries (the large sizes) keep their unit of work in registers: msm_radix_chunk (first pass: one
//                     workgroup per 32768-digit chunk, bins leave as runs of two lines), msm_radix_segment (last pass: one
//                     workgroup per segment, contiguous output, bucket tables); the generic msm_radix_{hist,scan,scatter}
//                     (LDS-tiled, work-item list) take every other shape and the over-long segments of skewed inputs
//   3 task tables     msm_scan_win (tasks per bucket, prefixes, length histogram) -> msm_bucket_fill: runs of <= L points,
//                     scheduled longest first, one lane per bucket
//   4 msm_accumulate  one lane per TASK: gathers the 64-byte affine point from HBM (next point
//                     prefetched under the current mixed add), XYZZ accumulator in VGPRs -- the
//                     dominant kernel; buckets are split so skewed scalar sets and small n still
//                     fill the chip and the grid has no long tail; a bucket that is one task is written in place
//   5 msm_combine /   partial sums of one bucket are folded <= 32 at a time (extra levels only
//     msm_finalize    when one bucket holds more than 32*L points; long folds by one wave each: msm_fold_big)
//   6 reduction       sum_b b*B_b: class sums (rows / columns of the bucket index, plain additions: msm_class_sums), then
//                     suffix scans + trees on quads over 2 x 256 class sums per window (msm_reduce_scan_quad); small
//                     windows: the scans alone
//   7 host            general mode: Horner over the W window sums (c doublings each)
//
// Two modes share every kernel:
//   general      W windows, each its own set of 2^(c-1) buckets (any bases, nothing cached)
//   precomputed  the SRS handle carries T[j][i] = 2^(c*j) * P_i (uzk_srs_precompute; 288 GB of HBM
//                make W*n*64 B cheap), so all W*n (point, window) pairs fall into ONE bucket set:
//                larger c, ~25 % fewer additions, no per-window reduction and no host Horner.
// No MFMA anywhere: the work is 254-bit modular integer arithmetic on v_mad_u64_u32.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <functional>
#include <memory>
#include <condition_variable>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "ec29.hpp"
#include "ecquad.hpp"
#include "ecquad29.hpp"
#include "ec29l.hpp"
#include "host_ec64.hpp"
#include "host_math.hpp"

namespace uzk {

constexpr uint32_t kSeg = 16;          // max buckets per lane in the reduction
constexpr uint32_t kSignBit = 0x80000000u;
constexpr uint32_t kCombineFan = 32;   // partial sums folded per lane and level
constexpr int kMaxPasses = 3;

struct MsmWork {
    DevBuf digits, bucket_count, bucket_start, sorted, buckets, partials, win_sums, class_sums, big;
    DevBuf ent[2];                           // radix ping-pong ({key, val} entries)
    DevBuf counts[kMaxPasses], segs_start[kMaxPasses], segs_len[kMaxPasses], items[kMaxPasses];
    DevBuf lvl_cnt[2], lvl_off[2], lvl_part[2], small, task_desc, exc;
    XYZZ* h_sums = nullptr;                  // pinned
    size_t h_sums_cap = 0;
    uint32_t* h_max = nullptr;               // pinned: largest bucket population of the current call
};

__host__ __device__ inline int msm_num_windows(int c) {
    int W = (254 + c - 1) / c;
    if (254 - (W - 1) * c == c) ++W;   // top window must leave room for the signed carry
    return W;
}

// ---- 1. digits -------------------------------------------------------------------------------
// Scalars above (r - 1) / 2 are replaced by r - k with every digit's sign flipped: k P = -(r - k) P.  Uniform scalars
// gain nothing, but the values a real witness is full of -- -1 = r - 1 (wire selectors, uzkge turbo/mod.rs:171-185),
// small negatives -- then populate one or two low windows instead of all of them.
__device__ __forceinline__ uint32_t msm_fold_scalar_sign(Fp& k) {
    // k > (r - 1) / 2  <=>  2k >= r + 1  <=>  2k > r
    bool gt = false, decided = false;
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        const uint32_t two_k = (k.v[j] << 1) | (j ? (k.v[j - 1] >> 31) : 0u);      // canonical k < r < 2^254: no bit 256
        if (!decided && two_k != FrCfg::M[j]) { gt = two_k > FrCfg::M[j]; decided = true; }
    }
    if (!gt) return 0u;
    k = Fr::neg(k);            // r - k (k != 0 here)
    return kSignBit;
}


// scalars: [batch][n]; digits: [batch][wcnt][n] for the windows w0 .. w0 + wcnt of the W-window recoding
// `zero8` (optional): eight counters this launch clears for the kernels behind it on the stream (the small pipeline's task /
// chunk / exception counters: one fill launch less per commit).
__global__ __launch_bounds__(256) void msm_digits_kernel(ScalarView scalars, uint32_t* __restrict__ digits,
                                                         uint32_t n, uint32_t batch, int c, int W, int w0, int wcnt,
                                                         uint32_t* __restrict__ zero8 = nullptr) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (zero8 != nullptr && t < 8) zero8[t] = 0;
    if (t >= (uint64_t)n * batch) return;
    const uint32_t b = (uint32_t)(t / n), i = (uint32_t)(t % n);
    Fp k = Fr::from_mont(scalars.at(b, i));
    const uint32_t flip = msm_fold_scalar_sign(k);
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint32_t carry = 0;
    uint32_t* dst = digits + (size_t)b * wcnt * n + i;
    for (int w = 0; w < w0 + wcnt && w < W; ++w) {       // the carry chain starts at window 0
        uint32_t d = (k.v[0] & mask) + carry;
        // k >>= c
#pragma unroll
        for (int j = 0; j < 7; ++j) k.v[j] = __funnelshift_r(k.v[j], k.v[j + 1], c);
        k.v[7] >>= c;
        uint32_t out;
        if (d > half) { out = ((1u << c) - d) | kSignBit; carry = 1; }
        else { out = d; carry = 0; }
        if (out & ~kSignBit) out ^= flip;                // a zero digit stays zero
        if (w >= w0) dst[(size_t)(w - w0) * n] = out;
    }
}

// The same, one workgroup per pass-0 sort chunk (grid (nch, batch), scalars [ch*cs, (ch+1)*cs) of a
// vector): besides the digits it leaves the chunk's pass-0 histograms counts[seg][ch][bins] for all
// windows, so the sort's first histogram kernel (a 4-byte read of every entry) is not needed.
// General mode, large n only (a chunk must be big enough to keep 1024 lanes busy).
__global__ __launch_bounds__(1024) void msm_digits_hist_kernel(ScalarView scalars, uint32_t* __restrict__ digits,
                                                               uint32_t n, int c, int W, uint32_t shift, uint32_t bins,
                                                               uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t hist[];       // [W][bins]
    const uint32_t ch = blockIdx.x, nch = gridDim.x, b = blockIdx.y;
    const uint32_t cs = (n + nch - 1) / nch;
    const uint32_t lo = min(n, ch * cs), hi = min(n, lo + cs);
    for (uint32_t k = threadIdx.x; k < (uint32_t)W * bins; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1), bmask = bins - 1;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        Fp k = Fr::from_mont(scalars.at(b, i));
        const uint32_t flip = msm_fold_scalar_sign(k);
        uint32_t carry = 0;
        uint32_t* dst = digits + (size_t)b * W * n + i;
        for (int w = 0; w < W; ++w) {
            uint32_t d = (k.v[0] & mask) + carry;
#pragma unroll
            for (int j = 0; j < 7; ++j) k.v[j] = __funnelshift_r(k.v[j], k.v[j + 1], c);
            k.v[7] >>= c;
            uint32_t out;
            if (d > half) { out = ((1u << c) - d) | kSignBit; carry = 1; }
            else { out = d; carry = 0; }
            if (out & ~kSignBit) out ^= flip;
            dst[(size_t)w * n] = out;
            const uint32_t mag = out & ~kSignBit;
            if (mag) atomicAdd(&hist[(uint32_t)w * bins + (((mag - 1) >> shift) & bmask)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < (uint32_t)W * bins; k += blockDim.x) {
        const uint32_t w = k / bins, bin = k % bins;
        counts[(((size_t)b * W + w) * nch + ch) * bins + bin] = hist[k];
    }
}

// ---- 2. counting sort --------------------------------------------------------------------------
// Bucket key = |digit| - 1 in [0, 2^(c-1)); zero digits are dropped.  A pass sorts every segment on
// one bit field of the key.  Pass 0 reads the digit array (FROM_DIGITS), later passes read 8-byte
// {key, val} entries; the last pass writes only val = point index | sign (OUT_VAL).
struct RadixArgs {
    const uint32_t* digits;      // FROM_DIGITS: [nseg][n]
    const uint2* in_entries;     // else: {key, val}
    const uint32_t* seg_start;   // else: absolute start / length of each input segment
    const uint32_t* seg_len;
    uint32_t n;                  // FROM_DIGITS: entries per segment
    uint32_t remap_cnt;          // FROM_DIGITS, precomputed mode: val = (k / cnt) * stride + off + k % cnt
    uint32_t remap_stride, remap_off;
    const uint32_t* item_off;    // else: work items (fixed-size chunks) of segment s are item_off[s] .. item_off[s+1]
    uint32_t nseg, chunk;        // else: number of segments, entries per work item
    uint32_t shift, mask, bins;  // bin = (key >> shift) & mask
    uint32_t* counts;            // pass 0: [seg][chunk][bins], later passes: [item][bins]  (written by the histogram)
    uint32_t* prefix;            // same shape: the scan's exclusive chunk prefixes, read by the scatter (= counts when one
                                 // workgroup scans a segment; a second array when several do, which re-read the counts)
    uint32_t* bin_base;          // [seg][bins] absolute output start of each bin
    uint32_t* bin_count;         // [seg][bins]
    uint2* out_entries;          // !OUT_VAL
    uint32_t* out_vals;          // OUT_VAL
    // Packed 4-byte intermediate entries (two-pass sorts whose fields fit 32 bits): the remaining low
    // key bits above the point index and its sign,  e = key_low << (idx_bits + 1) | sign << idx_bits | idx.
    // 0 = 8-byte {key, val} entries.
    uint32_t pk_in_bits, pk_out_bits;   // idx_bits of the input / output entry format (0 = unpacked)
    uint32_t seg_min_len;               // later passes: only segments LONGER than this are this path's (0 = all): the rest went to
                                        // msm_radix_segment_kernel
};
__device__ __forceinline__ uint32_t pk_key(uint32_t e, uint32_t idx_bits) { return e >> (idx_bits + 1); }
__device__ __forceinline__ uint32_t pk_val(uint32_t e, uint32_t idx_bits) {
    return (e & ((1u << idx_bits) - 1)) | (((e >> idx_bits) & 1u) << 31);
}
__device__ __forceinline__ uint32_t pk_make(uint32_t key_low, uint32_t val, uint32_t idx_bits) {
    return (key_low << (idx_bits + 1)) | ((val >> 31) << idx_bits) | (val & ~kSignBit);
}

// Which (segment, chunk) does this workgroup own, where are its counters, and which entries?
// Pass 0 (FROM_DIGITS): equal segments, grid (nch, nseg).  Later passes: segment lengths are data
// dependent (a skewed scalar set can put every entry into one segment), so the work is a 1-D list
// of fixed-size items built from the actual lengths; workgroups past the end of the list leave.
template <bool FROM_DIGITS>
__device__ __forceinline__ bool radix_work(const RadixArgs& a, uint32_t item, uint32_t& seg, size_t& cidx, uint32_t& base,
                                           uint32_t& lo, uint32_t& hi) {
    if constexpr (FROM_DIGITS) {
        const uint32_t ch = blockIdx.x, nch = gridDim.x;
        seg = blockIdx.y;
        cidx = ((size_t)seg * nch + ch) * a.bins;
        base = 0;
        const uint32_t cs = (a.n + nch - 1) / nch;
        lo = min(a.n, ch * cs);
        hi = min(a.n, lo + cs);
        return true;
    } else {
        if (item >= a.item_off[a.nseg]) return false;
        uint32_t l = 0, h = a.nseg;            // largest s with item_off[s] <= item (skips empty segments)
        while (h - l > 1) {
            const uint32_t mid = (l + h) >> 1;
            if (a.item_off[mid] <= item) l = mid; else h = mid;
        }
        seg = l;
        cidx = (size_t)item * a.bins;
        base = a.seg_start[seg];
        const uint32_t len = a.seg_len[seg];
        lo = min(len, (item - a.item_off[seg]) * a.chunk);
        hi = min(len, lo + a.chunk);
        return true;
    }
}
template <bool FROM_DIGITS, bool PK_IN>
__device__ __forceinline__ bool radix_load(const RadixArgs& a, uint32_t seg, uint32_t base, uint32_t k, uint32_t& key,
                                           uint32_t& val) {
    if constexpr (FROM_DIGITS) {
        const uint32_t d = a.digits[(size_t)seg * a.n + k];
        const uint32_t mag = d & ~kSignBit;
        key = mag - 1;
        uint32_t idx = k;
        if (a.remap_cnt) idx = (k / a.remap_cnt) * a.remap_stride + a.remap_off + (k % a.remap_cnt);
        val = idx | (d & kSignBit);
        return mag != 0;
    } else if constexpr (PK_IN) {
        const uint32_t e = reinterpret_cast<const uint32_t*>(a.in_entries)[(size_t)base + k];
        key = pk_key(e, a.pk_in_bits);
        val = pk_val(e, a.pk_in_bits);
        return true;
    } else {
        const uint2 e = a.in_entries[(size_t)base + k];
        key = e.x;
        val = e.y;
        return true;
    }
}

template <bool FROM_DIGITS, bool PK_IN>
__global__ __launch_bounds__(1024) void msm_radix_hist_kernel(RadixArgs a) {
    __shared__ uint32_t cnt[512];
    // later passes: the grid strides over the work items (their number is data dependent; the launch is sized for the chip)
    for (uint32_t item = blockIdx.x;; item += gridDim.x) {
        uint32_t seg, base, lo, hi;
        size_t cidx;
        if (!radix_work<FROM_DIGITS>(a, item, seg, cidx, base, lo, hi)) return;
        for (uint32_t b = threadIdx.x; b < a.bins; b += blockDim.x) cnt[b] = 0;
        __syncthreads();
        for (uint32_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
            if constexpr (FROM_DIGITS) {
                const uint32_t mag = a.digits[(size_t)seg * a.n + k] & ~kSignBit;
                if (mag) atomicAdd(&cnt[((mag - 1) >> a.shift) & a.mask], 1u);
            } else if constexpr (PK_IN) {
                const uint32_t e = reinterpret_cast<const uint32_t*>(a.in_entries)[(size_t)base + k];
                atomicAdd(&cnt[(pk_key(e, a.pk_in_bits) >> a.shift) & a.mask], 1u);
            } else {
                atomicAdd(&cnt[(a.in_entries[(size_t)base + k].x >> a.shift) & a.mask], 1u);
            }
        }
        __syncthreads();
        uint32_t* dst = a.counts + cidx;
        for (uint32_t b = threadIdx.x; b < a.bins; b += blockDim.x) dst[b] = cnt[b];
        if constexpr (FROM_DIGITS) return;
        __syncthreads();
    }
}

// grid (nseg, G), block 512: chunk prefixes per bin, bin totals, exclusive scan over bins.
// Absolute output start of the segment = out_start_of[seg] (or seg * out_stride when null).
// The chunk walk is a chain of dependent read-modify-writes per bin (512 chunks at n = 2^24: 170 us for the first pass's
// 15 workgroups); with G > 1 workgroup g of a segment walks only the chunks [g, g + 1) * nch / G, after summing the counts of
// the chunks before them (independent loads, read again rather than waited for), and the last one, which then holds the bin
// totals, writes the bin tables.
__global__ __launch_bounds__(512) void msm_radix_scan_kernel(RadixArgs a, uint32_t nch, const uint32_t* __restrict__ out_start_of,
                                                             uint32_t out_stride) {
    __shared__ uint32_t wsum[8];
    const uint32_t seg = blockIdx.x, b = threadIdx.x, G = gridDim.y, g = blockIdx.y;
    if (a.seg_min_len && a.seg_len[seg] <= a.seg_min_len) return;   // msm_radix_segment_kernel wrote this segment's tables
    uint32_t run = 0;
    if (b < a.bins) {
        size_t first = (size_t)seg * nch;
        uint32_t cnt = nch;
        if (a.item_off) { first = a.item_off[seg]; cnt = a.item_off[seg + 1] - a.item_off[seg]; }
        const uint32_t per = (cnt + G - 1) / G;
        const uint32_t c0 = min(cnt, g * per), c1 = min(cnt, c0 + per);
        const uint32_t* p0 = a.counts + first * a.bins + b;
        uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        uint32_t ch = 0;
        for (; ch + 4 <= c0; ch += 4) {
            r0 += p0[(size_t)ch * a.bins];
            r1 += p0[(size_t)(ch + 1) * a.bins];
            r2 += p0[(size_t)(ch + 2) * a.bins];
            r3 += p0[(size_t)(ch + 3) * a.bins];
        }
        for (; ch < c0; ++ch) r0 += p0[(size_t)ch * a.bins];
        run = r0 + r1 + r2 + r3;
        for (ch = c0; ch < c1; ++ch) {
            const size_t at = (first + ch) * a.bins + b;
            const uint32_t v = a.counts[at];
            a.prefix[at] = run;
            run += v;
        }
    }
    if (g + 1 != G) return;                       // the last workgroup's `run` is the bin total
    uint32_t incl = run;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up((int)incl, o);
        if ((int)(b & 63) >= o) incl += t;
    }
    if ((b & 63) == 63) wsum[b >> 6] = incl;
    __syncthreads();
    uint32_t pre = 0;
    for (uint32_t w = 0; w < (b >> 6); ++w) pre += wsum[w];
    const uint32_t excl = pre + incl - run;
    if (b < a.bins) {
        const uint32_t start = out_start_of ? out_start_of[seg] : seg * out_stride;
        a.bin_base[(size_t)seg * a.bins + b] = start + excl;
        a.bin_count[(size_t)seg * a.bins + b] = run;
    }
}

// item_off[s] = sum_{t < s} ceil(seg_len[t] / chunk), item_off[nseg] = total  (one workgroup)
__global__ __launch_bounds__(1024) void msm_radix_items_kernel(const uint32_t* __restrict__ seg_len, uint32_t nseg,
                                                               uint32_t chunk, uint32_t* __restrict__ item_off, uint32_t min_len) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nseg + 1023) / 1024;
    const uint32_t lo = min(nseg, tid * per), hi = min(nseg, lo + per);
    uint32_t s = 0;
    // segments of <= min_len entries belong to msm_radix_segment_kernel: no items
    auto items_of = [&](uint32_t k) { const uint32_t l = seg_len[k]; return l > min_len ? (l + chunk - 1) / chunk : 0u; };
    for (uint32_t k = lo; k < hi; ++k) s += items_of(k);
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t k = lo; k < hi; ++k) { item_off[k] = run; run += items_of(k); }
    if (tid == 1023) item_off[nseg] = part[1023];
}

// block TB, tile = TB * E entries sorted in LDS before they are written out
template <int TB, int E, bool FROM_DIGITS, bool OUT_VAL, bool PK_IN>
__global__ __launch_bounds__(TB) void msm_radix_scatter_kernel(RadixArgs a) {
    constexpr int TILE = TB * E;
    __shared__ uint2 buf[TILE];
    __shared__ uint32_t tcnt[512], toff[512], gcur[512], wsum[16];
    const uint32_t tid = threadIdx.x;
    for (uint32_t item = blockIdx.x;; item += gridDim.x) {   // later passes: grid-stride over the work items
    uint32_t seg, base, lo, hi;
    size_t cidx;
    if (!radix_work<FROM_DIGITS>(a, item, seg, cidx, base, lo, hi)) return;
    const uint32_t* coff = a.prefix + cidx;
    for (uint32_t b = tid; b < a.bins; b += TB) gcur[b] = a.bin_base[(size_t)seg * a.bins + b] + coff[b];
    for (uint32_t t0 = lo; t0 < hi; t0 += TILE) {
        for (uint32_t b = tid; b < a.bins; b += TB) tcnt[b] = 0;
        __syncthreads();
        uint32_t key[E], val[E], rank[E];
        bool ok[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t k = t0 + tid + e * TB;
            ok[e] = (k < hi) && radix_load<FROM_DIGITS, PK_IN>(a, seg, base, k, key[e], val[e]);
            if (ok[e]) rank[e] = atomicAdd(&tcnt[(key[e] >> a.shift) & a.mask], 1u);
        }
        __syncthreads();
        // exclusive scan of tcnt[0..bins) -> toff: each wave scans whole 64-bin groups (any TB >= 64)
        {
            const uint32_t ngroups = (a.bins + 63) / 64, lane = tid & 63;
            for (uint32_t g = tid >> 6; g < ngroups; g += TB / 64) {
                const uint32_t b = g * 64 + lane;
                const uint32_t v = (b < a.bins) ? tcnt[b] : 0;
                uint32_t incl = v;
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up((int)incl, o);
                    if ((int)lane >= o) incl += t;
                }
                if (b < a.bins) toff[b] = incl - v;
                if (lane == 63) wsum[g] = incl;
            }
            __syncthreads();
            for (uint32_t b = tid; b < a.bins; b += TB) {
                uint32_t pre = 0;
                for (uint32_t g = 0; g < (b >> 6); ++g) pre += wsum[g];
                toff[b] += pre;
            }
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (ok[e]) buf[toff[(key[e] >> a.shift) & a.mask] + rank[e]] = make_uint2(key[e], val[e]);
        }
        __syncthreads();
        const uint32_t last = a.bins - 1;
        const uint32_t total = toff[last] + tcnt[last];
        for (uint32_t k = tid; k < total; k += TB) {
            const uint2 en = buf[k];
            const uint32_t bin = (en.x >> a.shift) & a.mask;
            const uint32_t pos = gcur[bin] + (k - toff[bin]);
            if constexpr (OUT_VAL) a.out_vals[pos] = en.y;
            else if (a.pk_out_bits)
                reinterpret_cast<uint32_t*>(a.out_entries)[pos] = pk_make(en.x & ((1u << a.shift) - 1), en.y, a.pk_out_bits);
            else a.out_entries[pos] = en;
        }
        __syncthreads();
        for (uint32_t b = tid; b < a.bins; b += TB) gcur[b] += tcnt[b];
        __syncthreads();
    }
    if constexpr (FROM_DIGITS) return;
    }
}

// Last pass of a packed two-pass sort, ONE workgroup per segment (round 3).  After the first pass a segment -- the entries of
// one window that share the key's high bits -- is n / 512 entries long (32768 at n = 2^24): short enough to sit in one
// workgroup's registers (E entries per lane).  The workgroup then does the whole pass by itself: histogram of the low key bits
// (LDS atomics), exclusive scan, the bucket tables (start / population of every bucket of the segment), and the scatter
// through an LDS buffer that leaves as ONE contiguous run of full lines.  Compared with the generic pass this drops the separate
// histogram kernel (a second read of every entry), the per-chunk counter arrays and their scan, the work-item list, and the
// 32-byte partial-line writes of the 4096-entry tiles.  The LDS buffer may be smaller than the segment: the buckets are then
// written in rounds of whole bins (two workgroups per CU at n = 2^24, so one's loads run under the other's LDS phase); a single
// bin larger than the buffer (heavily skewed scalars) is written directly, its order being irrelevant.
// Segments longer than TB * E entries are left alone: the generic kernels behind this launch take exactly those
// (RadixArgs::seg_min_len), so skewed inputs keep their load-balanced path.
template <int TB, int E, int BUFN>
__global__ __launch_bounds__(TB, TB >= 512 ? 4 : 2) void msm_radix_segment_kernel(RadixArgs a) {   // 512 lanes: two workgroups per CU
    constexpr uint32_t CAP = (uint32_t)TB * E;
    constexpr int MAXBINS = 512;
    __shared__ uint32_t buf[BUFN];
    __shared__ uint32_t cnt[MAXBINS], off[MAXBINS + 1], cur[MAXBINS], wsum[8];
    const uint32_t seg = blockIdx.x, tid = threadIdx.x;
    const uint32_t len = a.seg_len[seg];
    if (len > CAP) return;                               // the generic path owns this segment
    const uint32_t start = a.seg_start[seg];
    const uint32_t bins = a.bins;
    for (uint32_t b = tid; b < bins; b += TB) cnt[b] = 0;
    __syncthreads();
    const uint32_t* __restrict__ in = reinterpret_cast<const uint32_t*>(a.in_entries) + start;
    uint32_t e[E];
    // entry i of this lane is k = tid + i * TB; it exists for i < mine (one compare per use instead of E kept predicates)
    const int mine = tid < len ? (int)((len - tid + TB - 1) / TB) : 0;
#pragma unroll
    for (int i = 0; i < E; ++i) e[i] = i < mine ? in[tid + (uint32_t)i * TB] : 0u;
#pragma unroll
    for (int i = 0; i < E; ++i)
        if (i < mine) atomicAdd(&cnt[(pk_key(e[i], a.pk_in_bits) >> a.shift) & a.mask], 1u);
    __syncthreads();
    {   // exclusive scan of cnt[0..bins) -> off, off[bins] = len
        const uint32_t ngroups = (bins + 63) / 64, lane = tid & 63;
        for (uint32_t g = tid >> 6; g < ngroups; g += TB / 64) {
            const uint32_t b = g * 64 + lane;
            const uint32_t v = (b < bins) ? cnt[b] : 0;
            uint32_t incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up((int)incl, o);
                if ((int)lane >= o) incl += t;
            }
            if (b < bins) off[b] = incl - v;
            if (lane == 63) wsum[g] = incl;
        }
        __syncthreads();
        for (uint32_t b = tid; b < bins; b += TB) {
            uint32_t pre = 0;
            for (uint32_t g = 0; g < (b >> 6); ++g) pre += wsum[g];
            const uint32_t o = off[b] + pre;
            off[b] = o;
            cur[b] = o;
            a.bin_base[(size_t)seg * bins + b] = start + o;
            a.bin_count[(size_t)seg * bins + b] = cnt[b];
        }
        if (tid == 0) off[bins] = len;
        __syncthreads();
    }
    uint32_t* __restrict__ out = a.out_vals + start;
    uint32_t b_lo = 0;
    while (b_lo < bins) {
        // this round: bins [b_lo, b_hi), as many as the buffer holds; a single bin beyond the buffer goes out directly
        const uint32_t base = off[b_lo];
        uint32_t l = b_lo, h = bins + 1;                 // largest b_hi with off[b_hi] - base <= BUFN
        while (h - l > 1) {
            const uint32_t mid = (l + h) >> 1;
            if (off[mid] - base <= (uint32_t)BUFN) l = mid; else h = mid;
        }
        const bool direct = (l == b_lo);
        const uint32_t b_hi = direct ? b_lo + 1 : l;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            // (opaque to the optimiser: the bin is recomputed from the entry here, two shifts, rather than kept in a
            // register per entry from the histogram loop -- E = 72 entries are all the registers two workgroups per CU leave)
            asm volatile("" : "+v"(e[i]));
            const uint32_t bin = (pk_key(e[i], a.pk_in_bits) >> a.shift) & a.mask;
            if (i < mine && bin >= b_lo && bin < b_hi) {
                const uint32_t pos = atomicAdd(&cur[bin], 1u);
                const uint32_t val = pk_val(e[i], a.pk_in_bits);
                if (direct) out[pos] = val; else buf[pos - base] = val;
            }
        }
        __syncthreads();
        if (!direct) {
            const uint32_t total = off[b_hi] - base;
            for (uint32_t k = tid; k < total; k += TB) out[base + k] = buf[k];
            __syncthreads();
        }
        b_lo = b_hi;
    }
}

// First pass of a packed two-pass sort with the same shape: ONE workgroup takes its whole chunk of a window's digits (<= TB * E,
// 32768 for the large sorts) into registers.  The chunk's bin populations are already known (the histogram pass, fused into the
// digit kernel at large n) and so are the bins' destinations (the scan), so the only LDS atomic is the one that hands out the
// position inside the bin; the packed entries leave through the LDS buffer in rounds of whole bins, every bin as ONE run of
// 64 entries on average (two full lines) instead of the 8-entry runs of a 4096-entry tile (1.85x write traffic at 2^24,
// rocprofv3 WRITE_SIZE), copied out by half-waves.  grid (chunks, segments) as for the generic scatter.
template <int TB, int E, int BUFN>
__global__ __launch_bounds__(TB, 4) void msm_radix_chunk_kernel(RadixArgs a) {
    constexpr int MAXBINS = 512;
    __shared__ uint32_t buf[BUFN];
    __shared__ uint32_t cnt[MAXBINS], off[MAXBINS + 1], cur[MAXBINS], gdst[MAXBINS], wsum[8];
    const uint32_t ch = blockIdx.x, nch = gridDim.x, seg = blockIdx.y, tid = threadIdx.x;
    const uint32_t bins = a.bins;
    const size_t cidx = ((size_t)seg * nch + ch) * bins;
    const uint32_t cs = (a.n + nch - 1) / nch;
    const uint32_t lo = min(a.n, ch * cs), hi = min(a.n, lo + cs), len = hi - lo;       // len <= TB * E (host)
    for (uint32_t b = tid; b < bins; b += TB) {
        cnt[b] = a.counts[cidx + b];
        gdst[b] = a.bin_base[(size_t)seg * bins + b] + a.prefix[cidx + b];
    }
    const uint32_t* __restrict__ in = a.digits + (size_t)seg * a.n + lo;
    uint32_t d[E];
    const int mine = tid < len ? (int)((len - tid + TB - 1) / TB) : 0;
#pragma unroll
    for (int i = 0; i < E; ++i) d[i] = i < mine ? in[tid + (uint32_t)i * TB] : 0u;       // a zero digit is dropped
    __syncthreads();
    {   // exclusive scan of cnt[0..bins) -> off, off[bins] = entries of the chunk
        const uint32_t ngroups = (bins + 63) / 64, lane = tid & 63;
        for (uint32_t g = tid >> 6; g < ngroups; g += TB / 64) {
            const uint32_t b = g * 64 + lane;
            const uint32_t v = (b < bins) ? cnt[b] : 0;
            uint32_t incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up((int)incl, o);
                if ((int)lane >= o) incl += t;
            }
            if (b < bins) off[b] = incl - v;
            if (lane == 63) wsum[g] = incl;
        }
        __syncthreads();
        uint32_t tot = 0;
        for (uint32_t b = tid; b < bins; b += TB) {
            uint32_t pre = 0;
            for (uint32_t g = 0; g < (b >> 6); ++g) pre += wsum[g];
            const uint32_t o = off[b] + pre;
            off[b] = o;
            cur[b] = o;
            if (b == bins - 1) tot = o + cnt[b];
        }
        if (tid == ((bins - 1) % TB)) off[bins] = tot;
        __syncthreads();
    }
    uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(a.out_entries);
    const uint32_t low_mask = (1u << a.shift) - 1;
    uint32_t b_lo = 0;
    while (b_lo < bins) {
        const uint32_t base = off[b_lo];
        uint32_t l = b_lo, h = bins + 1;                 // largest b_hi with off[b_hi] - base <= BUFN
        while (h - l > 1) {
            const uint32_t mid = (l + h) >> 1;
            if (off[mid] - base <= (uint32_t)BUFN) l = mid; else h = mid;
        }
        const bool direct = (l == b_lo);                 // one bin beyond the buffer (skewed scalars): written directly
        const uint32_t b_hi = direct ? b_lo + 1 : l;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            asm volatile("" : "+v"(d[i]));               // key and bin recomputed per round, not kept per entry
            const uint32_t mag = d[i] & ~kSignBit;
            const uint32_t key = mag - 1;
            const uint32_t bin = (key >> a.shift) & a.mask;
            if (mag != 0 && bin >= b_lo && bin < b_hi) {
                const uint32_t pos = atomicAdd(&cur[bin], 1u);
                const uint32_t word = pk_make(key & low_mask, (lo + tid + (uint32_t)i * TB) | (d[i] & kSignBit), a.pk_out_bits);
                if (direct) out[gdst[bin] + (pos - base)] = word; else buf[pos - base] = word;
            }
        }
        __syncthreads();
        if (!direct) {
            const uint32_t hl = tid & 31;
            for (uint32_t b = b_lo + (tid >> 5); b < b_hi; b += TB / 32) {
                const uint32_t c = cnt[b], src = off[b] - base, dst = gdst[b];
                for (uint32_t j = hl; j < c; j += 32) out[dst + j] = buf[src + j];
            }
            __syncthreads();
        }
        b_lo = b_hi;
    }
}

// ---- 3. task scans -----------------------------------------------------------------------------
// One workgroup per bucket window: v[b] = div ? ceil(cnt[b] / div) : cnt[b]; writes v (optional), the
// exclusive prefix of v within the window, the window total, and folds max(cnt) into *max_out.
// The window's counts are staged in LDS (dynamic, NB + NB/32 words: one pad word per 32 keeps a lane's run of
// NB/1024 consecutive entries off its neighbours' banks) so that every global access is coalesced; with the lanes
// walking their runs in global memory each instruction touched 64 cache lines and the kernel took 80 us at NB = 2^15.
__global__ __launch_bounds__(1024) void msm_scan_win_kernel(const uint32_t* __restrict__ cnt_in, uint32_t div,
                                                            uint32_t* __restrict__ v_out,
                                                            uint32_t* __restrict__ off_out,
                                                            uint32_t* __restrict__ win_total,
                                                            uint32_t* __restrict__ max_out, uint32_t NB,
                                                            uint32_t* __restrict__ len_hist = nullptr) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t lh[256];
    extern __shared__ uint32_t stage[];
    if (len_hist != nullptr) { if (threadIdx.x < 256) lh[threadIdx.x] = 0; __syncthreads(); }
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    const uint32_t per = (NB + 1023) / 1024;
    const uint32_t lo = min(NB, tid * per), hi = min(NB, lo + per);
    const uint32_t* cnt = cnt_in + (size_t)w * NB;
    // ceil(c / div) by one multiplication: magic = ceil(2^40 / div); exact for c + div < 2^30 (the error term
    // (magic div - 2^40) (c + div - 1) < div 2^30 <= 2^40 needs div <= 2^10, which the task and fold lengths are)
    const uint64_t magic = div ? ((1ull << 40) + div - 1) / div : 0;
    const bool fast = div && div <= 1024;
    auto val = [&](uint32_t c) -> uint32_t {
        if (!div) return c;
        if (fast && c < (1u << 29)) return (uint32_t)(((uint64_t)(c + div - 1) * magic) >> 40);
        return (c + div - 1) / div;
    };
    auto pad = [](uint32_t i) -> uint32_t { return i + (i >> 5); };
    uint32_t mx = 0;
    for (uint32_t i = tid; i < NB; i += 1024) {
        const uint32_t c = cnt[i];
        mx = max(mx, c);
        const uint32_t T = val(c);
        stage[pad(i)] = T;
        // level 0 (div = task length): the bucket's T tasks are r of q + 1 points and T - r of q (balanced split, task_extent):
        // their lengths go into the schedule's histogram here, in O(1) per bucket whatever T is
        if (len_hist != nullptr && c != 0) {
            const uint32_t q = c / T, r = c - q * T;
            if (r) atomicAdd(&lh[min(q + 1, 255u)], r);
            atomicAdd(&lh[min(q, 255u)], T - r);
        }
    }
    __syncthreads();
    if (len_hist != nullptr && tid < 256 && lh[tid]) atomicAdd(&len_hist[tid], lh[tid]);
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; ++b) s += stage[pad(b)];
    part[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
        uint32_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t b = lo; b < hi; ++b) {               // exclusive prefix in place
        const uint32_t v = stage[pad(b)];
        stage[pad(b)] = run;
        run += v;
    }
    __syncthreads();
    uint32_t* off_o = off_out + (size_t)w * NB;
    uint32_t* v_o = v_out ? v_out + (size_t)w * NB : nullptr;
    for (uint32_t i = tid; i < NB; i += 1024) {
        off_o[i] = stage[pad(i)];
        if (v_o) v_o[i] = val(cnt[i]);
    }
    if (tid == 1023) win_total[w] = part[1023];
    if (max_out) {
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_down((int)mx, o));
        if ((tid & 63) == 0 && mx) atomicMax(max_out, mx);
    }
}
// dynamic LDS of msm_scan_win_kernel for windows of NB buckets (above 64 KiB the runtime wants to be told once)
static size_t scan_win_lds(uint32_t NB) {
    static const bool raised = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(msm_scan_win_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   ((1 << 15) + (1 << 10) + 1) * 4) == hipSuccess;
    }();
    (void)raised;
    return ((size_t)NB + NB / 32 + 1) * 4;
}
// win_base[w] = sum of win_total[0..w), win_base[W] = grand total   (W <= 1024)
__global__ __launch_bounds__(1024) void msm_win_base_kernel(const uint32_t* __restrict__ win_total,
                                                            uint32_t* __restrict__ win_base, uint32_t W) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t v = tid < W ? win_total[tid] : 0;
    part[tid] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t t = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    if (tid < W) win_base[tid] = part[tid] - v;
    if (tid == 1023) win_base[W] = part[1023];
}

// ---- 4. bucket accumulation (dominant kernel) -------------------------------------------------
__device__ __forceinline__ Affine load_point(const Affine* __restrict__ pts, uint32_t idx) {
    return pts[idx];
}

// task id -> (window, bucket, j): binary search in win_base, then in the window's exclusive task
// prefix (the largest b with off[b] <= local id is the non-empty one).
__device__ __forceinline__ void find_task(uint32_t tid, const uint32_t* __restrict__ win_base, uint32_t W,
                                          const uint32_t* __restrict__ task_off, uint32_t NB, uint32_t& w,
                                          uint32_t& b, uint32_t& j) {
    uint32_t lo = 0, hi = W;           // invariant: win_base[lo] <= tid
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (win_base[mid] <= tid) lo = mid; else hi = mid;
    }
    w = lo;
    const uint32_t lt = tid - win_base[w];
    const uint32_t* off = task_off + (size_t)w * NB;
    lo = 0; hi = NB;                   // invariant: off[lo] <= lt, answer in [lo, hi)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= lt) lo = mid; else hi = mid;
    }
    b = lo;
    j = lt - off[lo];
}

// Length-ordered task schedule.  Lanes of a wave run consecutive schedule slots, so slots are grouped
// by trip count (longest first): a wave then has no idle lanes waiting for its longest run, and the
// grid's tail is made of the shortest tasks.  desc[slot] = {first sorted index (absolute), count}.
// Pass 1 counts task lengths (<= 255) per workgroup and reserves output ranges; pass 2 fills them.
struct TaskDesc { uint32_t start, cnt, task; uint32_t pad; };
constexpr uint32_t kLenBins = 256;

// cursor[len] = number of tasks strictly longer than len  (longest first)
__global__ __launch_bounds__(256) void msm_task_scan_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t h[kLenBins];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int l = kLenBins - 1; l >= 0; --l) { const uint32_t v = h[l]; h[l] = run; run += v; }
    }
    __syncthreads();
    cursor[threadIdx.x] = h[threadIdx.x];
}
// The same schedule without a search per task (round 3): one lane per BUCKET.  A bucket's tasks are two runs of equal lengths
// (r of q + 1 points, T - r of q), so the lane reserves two ranges of the schedule (LDS-aggregated counters, as above) and
// writes its descriptors; task ids are win_base[w] + task_off[bucket] + j as everywhere.  Buckets with more than kFillInline
// tasks (skewed scalars) are appended to a list and written by one wave each (msm_task_fill_big_kernel).  The histogram of
// the lengths comes from msm_scan_win_kernel.  (Round 3; it replaced a histogram and a fill kernel that did one binary search over
// the window's 2^15 prefixes per task, twice: 0.18 ms at 2^24.)
constexpr uint32_t kFillInline = 4;
struct BigBucket { uint32_t start, total, T, tid0; };
__global__ __launch_bounds__(256) void msm_bucket_fill_kernel(const uint32_t* __restrict__ win_base, const uint32_t* __restrict__ task_cnt,
                                                              const uint32_t* __restrict__ task_off,
                                                              const uint32_t* __restrict__ bucket_start,
                                                              const uint32_t* __restrict__ bucket_count, uint32_t NB, uint64_t TBK,
                                                              uint32_t* __restrict__ cursor, TaskDesc* __restrict__ desc,
                                                              uint32_t* __restrict__ big_count, BigBucket* __restrict__ big, uint32_t direct) {
    __shared__ uint32_t h[kLenBins], base[kLenBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t T = 0, total = 0, q = 0, r = 0, start = 0, tid0 = 0, rank_hi = 0, rank_lo = 0;
    bool inline_fill = false;
    if (t < TBK) {
        T = task_cnt[t];
        if (T != 0) {
            total = bucket_count[t];
            start = bucket_start[t];
            tid0 = win_base[t / NB] + task_off[t];
            q = total / T; r = total - q * T;
            if (T <= kFillInline) {
                inline_fill = true;
                if (r) rank_hi = atomicAdd(&h[min(q + 1, kLenBins - 1)], r);
                rank_lo = atomicAdd(&h[min(q, kLenBins - 1)], T - r);
            } else {
                BigBucket e; e.start = start; e.total = total; e.T = T; e.tid0 = tid0;
                big[atomicAdd(big_count, 1u)] = e;
            }
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (inline_fill) {
        const uint32_t p_hi = base[min(q + 1, kLenBins - 1)] + rank_hi, p_lo = base[min(q, kLenBins - 1)] + rank_lo;
        for (uint32_t j = 0; j < T; ++j) {
            TaskDesc d;
            d.start = start + j * q + min(j, r);
            d.cnt = q + (j < r ? 1u : 0u);
            d.task = (direct && T == 1) ? (kSignBit | (uint32_t)t) : tid0 + j;     // a one-task bucket's sum IS the bucket
            d.pad = 0;
            desc[j < r ? p_hi + j : p_lo + (j - r)] = d;
        }
    }
}
// one wave per listed bucket (grid-stride; the list is normally empty)
__global__ __launch_bounds__(64) void msm_task_fill_big_kernel(const uint32_t* __restrict__ big_count, const BigBucket* __restrict__ big,
                                                               uint32_t* __restrict__ cursor, TaskDesc* __restrict__ desc) {
    const uint32_t n_big = *big_count, lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < n_big; i += gridDim.x) {
        const BigBucket e = big[i];
        const uint32_t q = e.total / e.T, r = e.total - q * e.T;
        uint32_t p_hi = 0, p_lo = 0;
        if (lane == 0) {
            if (r) p_hi = atomicAdd(&cursor[min(q + 1, kLenBins - 1)], r);
            p_lo = atomicAdd(&cursor[min(q, kLenBins - 1)], e.T - r);
        }
        p_hi = (uint32_t)__shfl((int)p_hi, 0);
        p_lo = (uint32_t)__shfl((int)p_lo, 0);
        for (uint32_t j = lane; j < e.T; j += 64) {
            TaskDesc d;
            d.start = e.start + j * q + min(j, r);
            d.cnt = q + (j < r ? 1u : 0u);
            d.task = e.tid0 + j;
            d.pad = 0;
            desc[j < r ? p_hi + j : p_lo + (j - r)] = d;
        }
    }
}

// Where a task's sum goes: its slot among the partial sums, or -- a bucket that is ONE task (msm_bucket_fill_kernel marks it with
// the top bit when the launch sequence allows) -- straight into the bucket array, so that msm_finalize has nothing to copy for it.
__device__ __forceinline__ XYZZ& task_dst(XYZZ* __restrict__ partials, XYZZ* __restrict__ direct, uint32_t task) {
    return (task & kSignBit) ? direct[task & ~kSignBit] : partials[task];
}
// The same loop on the 29-bit-limb accumulator (ec29.hpp).  A task whose additions degenerate
// (the next point equals +-the running sum) is appended to exc_list and left to
// msm_accumulate_exc_kernel; for SRS-like inputs that list is empty.
__global__ __launch_bounds__(256, 4) void msm_accumulate29_kernel(const Affine* __restrict__ points,
                                                                  const uint32_t* __restrict__ sorted,
                                                                  const TaskDesc* __restrict__ desc,
                                                                  const uint32_t* __restrict__ win_base,
                                                                  XYZZ* __restrict__ partials, uint32_t W,
                                                                  uint32_t* __restrict__ exc_count,
                                                                  uint32_t* __restrict__ exc_list, XYZZ* __restrict__ direct = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= win_base[W]) return;
    const TaskDesc d = desc[slot];
    const uint32_t cnt = d.cnt;
    const uint32_t* run = sorted + d.start;
    Acc29 acc = acc29_inf();
    uint32_t e = run[0];
    Affine p = load_point(points, e & ~kSignBit);
    for (uint32_t k = 0; k < cnt; ++k) {
        const Affine cur = p;
        const bool neg = (e & kSignBit) != 0;
        if (k + 1 < cnt) {
            e = run[k + 1];
            p = load_point(points, e & ~kSignBit);
        }
        if (!acc29_madd(acc, cur, neg)) {
            exc_list[atomicAdd(exc_count, 1u)] = slot;
            return;
        }
    }
    task_dst(partials, direct, d.task) = acc29_to_xyzz(acc);
#endif
}
// Tasks the fast kernel gave up on, redone from scratch with the complete canonical group law.
__global__ __launch_bounds__(256) void msm_accumulate_exc_kernel(const Affine* __restrict__ points,
                                                                 const uint32_t* __restrict__ sorted,
                                                                 const TaskDesc* __restrict__ desc,
                                                                 XYZZ* __restrict__ partials,
                                                                 const uint32_t* __restrict__ exc_count,
                                                                 const uint32_t* __restrict__ exc_list, XYZZ* __restrict__ direct = nullptr) {
    const uint32_t total = *exc_count;                      // grid-stride: any grid size covers the whole list
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const TaskDesc d = desc[exc_list[i]];
        const uint32_t* run = sorted + d.start;
        XYZZ acc = xyzz_inf();
        for (uint32_t k = 0; k < d.cnt; ++k) {
            const uint32_t e = run[k];
            xyzz_madd(acc, load_point(points, e & ~kSignBit), (e & kSignBit) != 0);
        }
        task_dst(partials, direct, d.task) = acc;
    }
}

// Folding partial sums.  GS lanes of one wave cooperate on one output: lane `sub` adds the partials
// sub, sub + GS, ... and a shuffle tree adds the GS lane sums, so the dependent chain is
// ceil(cnt / GS) + log2(GS) additions instead of cnt (GS = 1 for large problems, where throughput
// matters and every lane has its own bucket; 4 or 16 for small n, where latency does).
__device__ __forceinline__ XYZZ xyzz_shfl_down(const XYZZ& p, int delta) {
    XYZZ r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int k = 0; k < 32; ++k) d[k] = (uint32_t)__shfl_down((int)s[k], delta);
    return r;
}
// the same on the lazy 29-bit limbs (ec29l.hpp): partial sums enter by re-limbing, the sum leaves by one exact division per coordinate
template <int GS>
__device__ __forceinline__ XYZZ fold_partials29(const XYZZ* __restrict__ src, uint32_t cnt, uint32_t sub) {
    P29 acc = p29_inf();
    for (uint32_t k = sub; k < cnt; k += GS) { const P29 q = p29_load(src[k]); p29_add(acc, q); }
    for (int o = GS / 2; o > 0; o >>= 1) {
        P29 q;
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(&acc);
        uint32_t* dp = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
        for (int w = 0; w < 36; ++w) dp[w] = (uint32_t)__shfl_down((int)sp[w], o);
        if (sub + (uint32_t)o < (uint32_t)GS) p29_add(acc, q);          // see fold_partials
    }
    return sub == 0 ? p29_store(acc) : xyzz_inf();   // valid in lane sub == 0
}
template <int GS>
__device__ __forceinline__ XYZZ fold_partials(const XYZZ* __restrict__ src, uint32_t cnt, uint32_t sub) {
    XYZZ acc = xyzz_inf();
    for (uint32_t k = sub; k < cnt; k += GS) { XYZZ q = src[k]; xyzz_add(acc, q); }
    for (int o = GS / 2; o > 0; o >>= 1) {
        XYZZ q = xyzz_shfl_down(acc, o);
        // lanes whose partner lies outside the group hold nothing lane 0 will use; adding there would feed a lane
        // its own value at the end of the wave and send the whole wave through the doubling branch
        if (sub + (uint32_t)o < (uint32_t)GS) xyzz_add(acc, q);
    }
    return acc;   // valid in lane sub == 0
}

// Intermediate level (only for buckets holding > G*L points): out task = sum of <= G partials.
// `big` (GS = 1 only, optional): outputs that fold more than big_thresh partial sums are not folded here but appended to a list
// {src offset, count, destination} for msm_fold_big_kernel -- see there.
struct BigFold { uint32_t src, cnt, dst, pad; };
template <int GS>
__global__ __launch_bounds__(256) void msm_combine_kernel(const XYZZ* __restrict__ in, const uint32_t* __restrict__ in_cnt,
                                                          const uint32_t* __restrict__ in_off,
                                                          const uint32_t* __restrict__ in_base,
                                                          const uint32_t* __restrict__ task_off,
                                                          const uint32_t* __restrict__ win_base, XYZZ* __restrict__ out,
                                                          uint32_t NB, uint32_t W, uint32_t G, uint32_t* __restrict__ big_count,
                                                          BigFold* __restrict__ big, uint32_t big_thresh, int arith29) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tid = gt / GS, sub = gt % GS;
    if (tid >= win_base[W]) return;          // whole groups leave together
    uint32_t w, b, j;
    find_task(tid, win_base, W, task_off, NB, w, b, j);
    const uint32_t total = in_cnt[(size_t)w * NB + b];
    const uint32_t first = j * G;
    const uint32_t cnt = min(G, total - first);
    const uint32_t at = in_base[w] + in_off[(size_t)w * NB + b] + first;
    if (GS == 1 && big != nullptr && cnt > big_thresh) {
        BigFold e; e.src = at; e.cnt = cnt; e.dst = tid; e.pad = 0;
        big[atomicAdd(big_count, 1u)] = e;
        return;
    }
    XYZZ acc = arith29 ? fold_partials29<GS>(in + at, cnt, sub) : fold_partials<GS>(in + at, cnt, sub);
    if (sub == 0) out[tid] = acc;
}
// Skewed scalar sets (a real witness is full of 0 / 1 / -1) put thousands of partial sums into a few buckets while everything
// else has one or two.  With one lane per output those few fold up to 32 partial sums in a row at a lone wave's latency
// (~10 us per addition): two levels were 0.61 of the 1.78 ms of a 2^20-point prover-mix MSM.  Here one WAVE takes a listed
// output: lane k loads partial sum k, five shuffle-tree steps add them -- 5 dependent additions instead of 31.
// grid-stride over the list (its length is data dependent), one wave per workgroup.  mode: 0 = write, 1 = add onto dst.
__global__ __launch_bounds__(64) void msm_fold_big_kernel(const XYZZ* __restrict__ in, const uint32_t* __restrict__ big_count,
                                                          const BigFold* __restrict__ big, XYZZ* __restrict__ out, int accumulate) {
    const uint32_t total = *big_count, lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
        const BigFold e = big[i];
        XYZZ acc = xyzz_inf();
        for (uint32_t k = lane; k < e.cnt; k += 64) { XYZZ q = in[e.src + k]; xyzz_add(acc, q); }
        for (int o = 32; o > 0; o >>= 1) {
            XYZZ q = xyzz_shfl_down(acc, o);
            if (lane + (uint32_t)o < 64) xyzz_add(acc, q);
        }
        if (lane == 0) {
            if (accumulate) { XYZZ prev = out[e.dst]; xyzz_add(prev, acc); out[e.dst] = prev; }
            else out[e.dst] = acc;
        }
    }
}

// Last level: one group per bucket folds its <= G partials into the dense bucket array.
template <int GS>
__global__ __launch_bounds__(256) void msm_finalize_kernel(const XYZZ* __restrict__ in, const uint32_t* __restrict__ in_cnt,
                                                           const uint32_t* __restrict__ in_off,
                                                           const uint32_t* __restrict__ in_base,
                                                           XYZZ* __restrict__ buckets, uint32_t NB, uint32_t W, int accumulate,
                                                           uint32_t* __restrict__ big_count, BigFold* __restrict__ big, uint32_t big_thresh,
                                                           int skip_single, int arith29) {
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t t = gt / GS;
    const uint32_t sub = (uint32_t)(gt % GS);
    if (t >= (size_t)W * NB) return;
    const uint32_t w = (uint32_t)(t / NB);
    const uint32_t cnt = in_cnt[t];
    if (skip_single && cnt == 1) return;                          // the accumulator wrote this bucket itself (task_dst)
    if (GS == 1 && big != nullptr && cnt > big_thresh) {          // msm_fold_big_kernel writes (or adds onto) this bucket
        BigFold e; e.src = in_base[w] + in_off[t]; e.cnt = cnt; e.dst = (uint32_t)t; e.pad = 0;
        big[atomicAdd(big_count, 1u)] = e;
        return;
    }
    XYZZ acc = arith29 ? fold_partials29<GS>(in + in_base[w] + in_off[t], cnt, sub) : fold_partials<GS>(in + in_base[w] + in_off[t], cnt, sub);
    if (sub == 0) {
        // streamed MSMs (msm_run_streamed) add every later point chunk's bucket sums onto the first one's
        if (accumulate && cnt != 0) { XYZZ prev = buckets[t]; xyzz_add(prev, acc); buckets[t] = prev; }
        else if (!accumulate) buckets[t] = acc;
    }
}

// Skewed inputs need extra fold levels, which re-materialise EVERY bucket's partial sums level by level: the sums the accumulator
// wrote straight into the bucket array go back to their slots first (one copy per one-task bucket; only ever runs for such inputs).
__global__ __launch_bounds__(256) void msm_undirect_kernel(const XYZZ* __restrict__ buckets, const uint32_t* __restrict__ cnt,
                                                           const uint32_t* __restrict__ off, const uint32_t* __restrict__ base,
                                                           XYZZ* __restrict__ partials, uint32_t NB, uint64_t TBK) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= TBK || cnt[t] != 1) return;
    partials[base[t / NB] + off[t]] = buckets[t];
}

// ---- 6. bucket reduction ------------------------------------------------------------------------
// Logical window sum = sum_{b=1}^{nbk} b * B_b.  Lane g owns buckets [g*seg + 1, (g+1)*seg]:
//   run_g = sum B_b, acc_g = sum (b - g*seg) B_b (running-sum trick), T_g = acc_g + (g*seg) run_g.
// A 256-lane workgroup tree-adds its T_g through LDS and writes one partial.  grid (groups, windows).
__global__ __launch_bounds__(256) void msm_reduce_kernel(const XYZZ* __restrict__ buckets, XYZZ* __restrict__ partials,
                                                         uint32_t nbk, uint32_t groups_per_window, uint32_t seg) {
    __shared__ XYZZ sh[256];
    const uint32_t w = blockIdx.y, tid = threadIdx.x;
    const uint32_t g = blockIdx.x * blockDim.x + tid;
    const XYZZ* bw = buckets + (size_t)w * nbk;
    XYZZ run = xyzz_inf(), acc = xyzz_inf();
    const uint64_t lo64 = (uint64_t)g * seg;   // bucket ids lo+1 .. lo+seg  (array index = id - 1)
    if (lo64 < nbk) {
        const uint32_t lo = (uint32_t)lo64;
        const uint32_t hi = min(nbk, lo + seg);
        for (uint32_t idx = hi; idx-- > lo;) {
            XYZZ bk = bw[idx];
            xyzz_add(run, bk);
            xyzz_add(acc, run);
        }
        // acc += lo * run   (double-and-add)
        if (lo != 0 && !xyzz_is_inf(run)) {
            XYZZ m = xyzz_inf();
            for (int bit = 31 - __clz(lo); bit >= 0; --bit) {
                m = xyzz_dbl(m);
                if ((lo >> bit) & 1) xyzz_add(m, run);
            }
            xyzz_add(acc, m);
        }
    }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            XYZZ a = sh[tid];
            XYZZ b2 = sh[tid + s];
            xyzz_add(a, b2);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) partials[(size_t)w * groups_per_window + blockIdx.x] = sh[0];
}
// out[w] = sum of partials[w][0..groups)   (one 256-lane workgroup per window)
__global__ __launch_bounds__(256) void msm_fold_partials_kernel(const XYZZ* __restrict__ partials, XYZZ* __restrict__ out,
                                                                uint32_t groups) {
    __shared__ XYZZ sh[256];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    XYZZ acc = xyzz_inf();
    for (uint32_t g = tid; g < groups; g += 256) { XYZZ q = partials[(size_t)w * groups + g]; xyzz_add(acc, q); }
    sh[tid] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            XYZZ a = sh[tid];
            XYZZ b2 = sh[tid + s];
            xyzz_add(a, b2);
            sh[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) out[w] = sh[0];
}

// The same reduction without the per-lane double-and-add.  With t the lane's index in its workgroup and
// G the workgroup's index in its window, a bucket's weight is  b = (G*256 + t)*seg + j,  j = 1..seg, so
//   sum b*B_b = sum_t acc_t  +  seg * sum_t t*run_t  +  256*seg * G * sum_t run_t
// and  sum_t t*run_t = sum_{t >= 1} Suffix_t  (Suffix_t = run_t + run_(t+1) + ...): an 8-step suffix scan
// of the run_t through LDS, log2(seg) doublings per lane and one tree sum give the first two terms
// (P1); the third is left to the fold kernel, which applies the same identity to the workgroup totals
// R_G.  Every lane does the same work (no data-dependent double-and-add), so short segments -- many
// lanes, short dependent chains -- become affordable.  grid (groups, windows).
// `width` = power of two >= the number of lanes that hold anything (lanes beyond it hold infinity)
__device__ __forceinline__ void wg_suffix_scan(XYZZ* sh, uint32_t tid, XYZZ& mine, uint32_t width) {
    sh[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < width; off <<= 1) {
        XYZZ v = (tid + off < 256) ? sh[tid + off] : xyzz_inf();
        __syncthreads();
        xyzz_add(mine, v);
        sh[tid] = mine;
        __syncthreads();
    }
}
__device__ __forceinline__ void wg_tree_sum(XYZZ* sh, uint32_t tid, const XYZZ& mine, uint32_t width) {
    sh[tid] = mine;
    __syncthreads();
    for (uint32_t s = width >> 1; s > 0; s >>= 1) {
        if (tid < s) {
            XYZZ a = sh[tid];
            XYZZ b2 = sh[tid + s];
            xyzz_add(a, b2);
            sh[tid] = a;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void msm_reduce_scan_kernel(const XYZZ* __restrict__ buckets, XYZZ* __restrict__ part_p1,
                                                              XYZZ* __restrict__ part_r, uint32_t nbk,
                                                              uint32_t groups_per_window, uint32_t seg, uint32_t log_seg) {
    __shared__ XYZZ sh[256];
    const uint32_t w = blockIdx.y, tid = threadIdx.x;
    const uint32_t g = blockIdx.x * blockDim.x + tid;
    const XYZZ* bw = buckets + (size_t)w * nbk;
    XYZZ run = xyzz_inf(), acc = xyzz_inf();
    const uint64_t lo64 = (uint64_t)g * seg;   // bucket ids lo+1 .. lo+seg  (array index = id - 1)
    if (lo64 < nbk) {
        const uint32_t lo = (uint32_t)lo64;
        const uint32_t hi = min(nbk, lo + seg);
        for (uint32_t idx = hi; idx-- > lo;) {
            XYZZ bk = bw[idx];
            xyzz_add(run, bk);
            xyzz_add(acc, run);
        }
    }
    // lanes of this workgroup that own buckets (rounded up to a power of two)
    const uint64_t lanes_w = ((uint64_t)nbk + seg - 1) / seg;
    const uint64_t mine_first = (uint64_t)blockIdx.x * 256;
    uint32_t active = lanes_w > mine_first ? (uint32_t)min((uint64_t)256, lanes_w - mine_first) : 1u;
    uint32_t width = 1;
    while (width < active) width <<= 1;
    XYZZ suf = run;
    wg_suffix_scan(sh, tid, suf, width);          // sh[t] = Suffix_t (inclusive)
    XYZZ above = (tid + 1 < 256) ? sh[tid + 1] : xyzz_inf();
    const XYZZ total = sh[0];
    __syncthreads();
    for (uint32_t d = 0; d < log_seg; ++d) above = xyzz_dbl(above);
    xyzz_add(acc, above);                         // acc_t + seg * Suffix_(t+1)
    wg_tree_sum(sh, tid, acc, width);
    if (tid == 0) {
        part_p1[(size_t)w * groups_per_window + blockIdx.x] = sh[0];
        part_r[(size_t)w * groups_per_window + blockIdx.x] = total;
    }
}
// out[w] = sum_G P1_G + 256*seg * sum_{G >= 1} SuffixR_G   (one workgroup per window, groups <= 256)
__global__ __launch_bounds__(256) void msm_fold_scan_kernel(const XYZZ* __restrict__ part_p1, const XYZZ* __restrict__ part_r,
                                                            XYZZ* __restrict__ out, uint32_t groups, uint32_t log_shift) {
    __shared__ XYZZ sh[256];
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    XYZZ r = tid < groups ? part_r[(size_t)w * groups + tid] : xyzz_inf();
    XYZZ p1 = tid < groups ? part_p1[(size_t)w * groups + tid] : xyzz_inf();
    uint32_t width = 1;
    while (width < groups) width <<= 1;
    if (groups > 1) {
        wg_suffix_scan(sh, tid, r, width);
        XYZZ above = (tid + 1 < 256) ? sh[tid + 1] : xyzz_inf();
        __syncthreads();
        if (!xyzz_is_inf(above)) {
            for (uint32_t d = 0; d < log_shift; ++d) above = xyzz_dbl(above);
            xyzz_add(p1, above);
        }
    }
    wg_tree_sum(sh, tid, p1, width);
    if (tid == 0) out[w] = sh[0];
}

// Class sums (round 3): the first level of a bucket reduction without weights.  With the bucket index i = h * 2^s + l,
//   sum_i (i + 1) B_i  =  2^s * sum_h h R_h  +  sum_l (l + 1) C_l,     R_h = sum_l B_(h,l)  (rows),  C_l = sum_h B_(h,l)  (columns):
// 2 additions per bucket like the running sums above, but PLAIN sums -- no per-lane chain of 2 * seg dependent additions, no
// double-and-add, no scan: every lane adds 8 buckets, the 256 partial sums of a workgroup are folded in LDS by fewer and fewer
// lanes (4, 4, 2 to one: 15 dependent additions in all), and what is left -- 2 * 256 class sums per window instead of 2^15 or
// 2^16 buckets -- goes through the scan kernels above as a problem of the size they are quick at.  The host's Horner takes
// the two sums of a window with its doublings split c - s | s (host_ec64.hpp horner_split).
// grid (tiles, logical windows, 2): z = 0 rows, z = 1 columns; a tile = 2048 buckets: 8 rows, or 2048 / H columns of all H
// rows.  out: [window][2][nb2] with rows at index h - 1 (weight h; R_0 has weight 0 and is dropped) and columns at l.
__global__ __launch_bounds__(256, 3) void msm_class_sums_kernel(const XYZZ* __restrict__ buckets, XYZZ* __restrict__ out, uint32_t nbk,
                                                              uint32_t log_s, uint32_t nb2, int arith29) {
    __shared__ XYZZ sh[256];
    const uint32_t tile = blockIdx.x, w = blockIdx.y, cols = blockIdx.z, t = threadIdx.x;
    const uint32_t Lc = 1u << log_s, H = nbk >> log_s;
    const XYZZ* bw = buckets + (size_t)w * nbk;
    XYZZ* ow = out + ((size_t)w * 2 + cols) * nb2;
    // G = partial sums per class inside the workgroup: a row is 2^s / 8 lanes, a column H / 8
    const uint32_t G = cols ? H / 8 : Lc / 8;
    // Every lane's eight additions -- nine tenths of the kernel's arithmetic -- on the lazy 29-bit limbs (ec29l.hpp: the buckets
    // enter by re-limbing, the lane's sum leaves through one product per coordinate); `arith29` = 0: the 8 x 32-bit words.
    const XYZZ* src = cols ? bw + (tile * (256 / G) + t / G) : bw + (size_t)tile * 2048 + (size_t)t * 8;
    const size_t step = cols ? Lc : 1, first = cols ? (size_t)(t % G) * 8 * Lc : 0;
    if (arith29) {
        P29 acc = p29_inf();
#pragma unroll 1
        for (uint32_t k = 0; k < 8; ++k) { const P29 q = p29_load(src[first + k * step]); p29_add(acc, q); }
        sh[t] = p29_store(acc);
    } else {
        XYZZ acc = xyzz_inf();
#pragma unroll 1
        for (uint32_t k = 0; k < 8; ++k) { XYZZ q = src[first + k * step]; xyzz_add(acc, q); }
        sh[t] = acc;
    }
    __syncthreads();
    // fold groups of G consecutive partial sums, 4 (or 2) to one per step; lane j takes partial sums j*f .. j*f + f - 1
    // From 64 outputs per step on, a QUAD takes each output (ecquad.hpp): these steps are chains of dependent additions on a
    // few lanes, where four lanes per addition are four times faster (3.2 us against 13 us for a lone wave's 8 x 32-bit addition).
    uint32_t live = 256;
    const uint32_t qi = t >> 2, ql = t & 3;
    for (uint32_t g = G; g > 1;) {
        const uint32_t f = (g % 4 == 0) ? 4 : 2;
        live /= f;
        XYZZ a;
        if (live <= 64) {
            if (qi < live) {
                a = sh[qi * f];
                for (uint32_t k = 1; k < f; ++k) { const XYZZ q = sh[qi * f + k]; xyzz_add_quad(a, q, ql); }
            }
            __syncthreads();
            if (qi < live && ql == 0) sh[qi] = a;
        } else {
            if (t < live) {
                a = sh[t * f];
                for (uint32_t k = 1; k < f; ++k) { XYZZ q = sh[t * f + k]; xyzz_add(a, q); }
            }
            __syncthreads();
            if (t < live) sh[t] = a;
        }
        __syncthreads();
        g /= f;
    }
    if (t < 256 / G) {
        const uint32_t cls = tile * (256 / G) + t;          // row h or column l
        if (!cols) { if (cls != 0) ow[cls - 1] = sh[t]; }
        else ow[cls] = sh[t];
    }
    // padding of the row array: weights H .. nb2 do not exist (and R_0's slot moved down by one)
    // (the column array likewise beyond its 2^s columns when there are more rows than columns)
    if (tile == 0) for (uint32_t k = (cols ? Lc : H - 1) + t; k < nb2; k += 256) ow[k] = xyzz_inf();
}

// The two kernels above with one quad per lane-role (block 256 = 64 quads, ecquad.hpp): the same identity, every
// dependent addition / doubling at a quarter of its latency.  t = quad index in the workgroup.
// 256 threads = one wave per SIMD: two waves share the SIMD's issue port and every dependent step takes twice as long
// (tools/microbench/quad_latency.hip: quad addition 3.2 us alone, 5.9 us with a second wave on the SIMD)
constexpr uint32_t kQuadLanes = 64, kLogQuadLanes = 6;
__device__ __forceinline__ void wgq_suffix_scan(XYZZ* sh, uint32_t t, uint32_t q, XYZZ& mine, uint32_t width) {
    if (q == 0) sh[t] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < width; off <<= 1) {
        const bool has = t + off < kQuadLanes;
        XYZZ v;
        if (has) v = sh[t + off];
        __syncthreads();
        if (has) { xyzz_add_quad(mine, v, q); if (q == 0) sh[t] = mine; }
        __syncthreads();
    }
}
__device__ __forceinline__ void wgq_tree_sum(XYZZ* sh, uint32_t t, uint32_t q, XYZZ mine, uint32_t width) {
    if (q == 0) sh[t] = mine;
    __syncthreads();
    for (uint32_t s = width >> 1; s > 0; s >>= 1) {
        if (t < s) {
            const XYZZ v = sh[t + s];
            xyzz_add_quad(mine, v, q);
            if (q == 0) sh[t] = mine;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(kQuadLanes * 4) void msm_reduce_scan_quad_kernel(const XYZZ* __restrict__ buckets, XYZZ* __restrict__ part_p1,
                                                                    XYZZ* __restrict__ part_r, uint32_t nbk,
                                                                    uint32_t groups_per_window, uint32_t seg, uint32_t log_seg) {
    __shared__ XYZZ sh[kQuadLanes];
    const uint32_t w = blockIdx.y, t = threadIdx.x >> 2, q = threadIdx.x & 3;
    const uint32_t g = blockIdx.x * kQuadLanes + t;
    const XYZZ* bw = buckets + (size_t)w * nbk;
    XYZZ run = xyzz_inf(), acc = xyzz_inf();
    const uint64_t lo64 = (uint64_t)g * seg;
    if (lo64 < nbk) {
        const uint32_t lo = (uint32_t)lo64;
        const uint32_t hi = min(nbk, lo + seg);
        for (uint32_t idx = hi; idx-- > lo;) {
            const XYZZ bk = bw[idx];
            xyzz_add_quad(run, bk, q);
            xyzz_add_quad(acc, run, q);
        }
    }
    const uint64_t lanes_w = ((uint64_t)nbk + seg - 1) / seg;
    const uint64_t mine_first = (uint64_t)blockIdx.x * kQuadLanes;
    const uint32_t active = lanes_w > mine_first ? (uint32_t)min((uint64_t)kQuadLanes, lanes_w - mine_first) : 1u;
    uint32_t width = 1;
    while (width < active) width <<= 1;
    XYZZ suf = run;
    wgq_suffix_scan(sh, t, q, suf, width);
    XYZZ above = (t + 1 < kQuadLanes) ? sh[t + 1] : xyzz_inf();
    const XYZZ total = sh[0];
    __syncthreads();
    for (uint32_t d = 0; d < log_seg; ++d) xyzz_dbl_quad(above, q);
    xyzz_add_quad(acc, above, q);
    wgq_tree_sum(sh, t, q, acc, width);
    if (threadIdx.x == 0) {
        part_p1[(size_t)w * groups_per_window + blockIdx.x] = sh[0];
        part_r[(size_t)w * groups_per_window + blockIdx.x] = total;
    }
}
__global__ __launch_bounds__(kQuadLanes * 4) void msm_fold_scan_quad_kernel(const XYZZ* __restrict__ part_p1, const XYZZ* __restrict__ part_r,
                                                                  XYZZ* __restrict__ out, uint32_t groups, uint32_t log_shift) {
    __shared__ XYZZ sh[kQuadLanes];
    const uint32_t w = blockIdx.x, t = threadIdx.x >> 2, q = threadIdx.x & 3;
    XYZZ r = t < groups ? part_r[(size_t)w * groups + t] : xyzz_inf();
    XYZZ p1 = t < groups ? part_p1[(size_t)w * groups + t] : xyzz_inf();
    uint32_t width = 1;
    while (width < groups) width <<= 1;
    if (groups > 1) {
        wgq_suffix_scan(sh, t, q, r, width);
        XYZZ above = (t + 1 < kQuadLanes) ? sh[t + 1] : xyzz_inf();
        __syncthreads();
        if (!xyzz_is_inf(above)) {
            for (uint32_t d = 0; d < log_shift; ++d) xyzz_dbl_quad(above, q);
            xyzz_add_quad(p1, above, q);
        }
    }
    wgq_tree_sum(sh, t, q, p1, width);
    if (threadIdx.x == 0) out[w] = sh[0];
}

// ---- small problems (n <= 2^15): one workgroup per (vector, window) slot ---------------------------
// The prover's real size is n = 2^14: the general pipeline above spends 17 launches and a
// host round trip on it.  Here a slot's whole sort -- histogram, prefixes, scatter, the task table and the
// tables of every fold level -- happens inside one workgroup's LDS, and nothing is read back before the
// window sums:
//   msm_digits -> msm_small_sort -> msm_accumulate29 (+ exceptions) -> msm_small_fold x levels -> msm_small_reduce
// Tasks are runs of <= L sorted entries of one bucket (balanced split, as above).  Their partial sums are folded
// 16 to 1 per level until every bucket holds one sum -- ceil(log16(n / L)) levels sized on the host, so the depth
// is bounded whatever the skew, and a bucket that is down to one sum takes no part in later levels (the uniform
// case: one level).  All sums live in one array P: level 0 = task partials, level k = the sums of level k's chunks.
// The slot's reduction workgroup finishes with sum_b b*B_b as a suffix scan + tree over its 2^(c-1) buckets in LDS.
struct SmallChunk { uint32_t first, cnt; };          // sums P[first .. first + cnt) of one bucket
constexpr uint32_t kSmallFan = 16;                   // sums folded per chunk
constexpr int kSmallMaxLevels = 4;                   // 16^4 >= 2^15 / 2 tasks of one bucket
constexpr uint32_t kSmallNone = 0xFFFFFFFFu;
struct SmallLevels {
    uint32_t nl;                                     // fold levels 1 .. nl
    uint32_t pbase[kSmallMaxLevels + 1];             // first sum of each level's region in P (pbase[0] = 0)
    uint32_t dbase[kSmallMaxLevels + 1];             // first chunk descriptor of each level
};

// exclusive prefix of v over the workgroup (all threads call; wsum: LDS[17]); total = sum over the workgroup
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* wsum, uint32_t tid, uint32_t nthreads,
                                                       uint32_t& total) {
    const uint32_t lane = tid & 63;
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if ((int)lane >= o) incl += t;
    }
    __syncthreads();                                  // wsum may still be read from the previous call
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (uint32_t w = 0; w < nthreads / 64; ++w) {
        const uint32_t x = wsum[w];
        if (w < (tid >> 6)) pre += x;
        tot += x;
    }
    total = tot;
    return pre + incl - v;
}
// largest b in [0, nb) with off[b] <= t   (off: exclusive prefix, non-decreasing; picks the non-empty bucket)
__device__ __forceinline__ uint32_t small_find(const uint32_t* off, uint32_t nb, uint32_t t) {
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// grid (slots), block 1024, dynamic LDS (5 NBL + 32 + n) words.  digits: [slot][n].
// counters[k]: running total over all slots of the tasks (k = 0) / the chunks of level k (dense packing of the tables).
__global__ __launch_bounds__(1024) void msm_small_sort_kernel(const uint32_t* __restrict__ digits, uint32_t n, uint32_t NBL,
                                                              uint32_t L, uint32_t* __restrict__ sorted,
                                                              TaskDesc* __restrict__ desc, SmallChunk* __restrict__ cdesc,
                                                              uint32_t* __restrict__ bucket_ref, uint2* __restrict__ slot_chunks,
                                                              uint32_t* __restrict__ counters, SmallLevels lv, uint32_t W,
                                                              uint32_t pre_stride, uint32_t pre_off) {
    extern __shared__ uint32_t sm_small[];
    uint32_t* cnt = sm_small;                 // [NBL]    bucket populations
    uint32_t* start = cnt + NBL;              // [NBL+1]  exclusive entry prefix
    uint32_t* toff = start + NBL + 1;         // [NBL+1]  exclusive task prefix
    uint32_t* coff = toff + NBL + 1;          // [NBL+1]  exclusive chunk prefix of the level being built
    uint32_t* cur = coff + NBL + 1;           // [NBL]    scatter cursors, then the buckets' current first sum
    uint32_t* misc = cur + NBL;               // [4]      table bases of this slot
    uint32_t* wsum = misc + 4;                // [17]
    uint32_t* stage = wsum + 17;              // [n]      the slot's sorted entries
    const uint32_t slot = blockIdx.x, tid = threadIdx.x;
    const uint32_t* dg = digits + (size_t)slot * n;
    for (uint32_t b = tid; b < NBL; b += 1024) { cnt[b] = 0; cur[b] = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t mag = dg[i] & ~kSignBit;
        if (mag) atomicAdd(&cnt[mag - 1], 1u);
    }
    __syncthreads();
    const uint32_t e = tid < NBL ? cnt[tid] : 0u;
    const uint32_t tk = (e + L - 1) / L;
    uint32_t tot_e, tot_t;
    const uint32_t es = block_excl_scan_u32(e, wsum, tid, 1024, tot_e);
    const uint32_t ts = block_excl_scan_u32(tk, wsum, tid, 1024, tot_t);
    if (tid < NBL) { start[tid] = es; toff[tid] = ts; }
    if (tid == 0) {
        start[NBL] = tot_e; toff[NBL] = tot_t;
        misc[0] = atomicAdd(&counters[0], tot_t);
    }
    __syncthreads();
    const uint32_t tbase = misc[0];
    // window table (uzk_srs_precompute): window w of every vector reads row w of T[w][i] = 2^(c w) P_i
    const uint32_t idx_base = pre_stride ? (slot % W) * pre_stride + pre_off : 0u;
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t d = dg[i], mag = d & ~kSignBit;
        if (mag) stage[start[mag - 1] + atomicAdd(&cur[mag - 1], 1u)] = (idx_base + i) | (d & kSignBit);
    }
    __syncthreads();
    uint32_t* so = sorted + (size_t)slot * n;
    for (uint32_t k = tid; k < tot_e; k += 1024) so[k] = stage[k];
    for (uint32_t t = tid; t < tot_t; t += 1024) {
        const uint32_t b = small_find(toff, NBL, t), j = t - toff[b];
        const uint32_t total = cnt[b], T = toff[b + 1] - toff[b];
        const uint32_t q = total / T, r = total - q * T;
        TaskDesc d;
        d.start = slot * n + start[b] + j * q + min(j, r);
        d.cnt = q + (j < r ? 1u : 0u);
        d.task = tbase + t;
        d.pad = 0;
        desc[tbase + t] = d;
    }
    // fold levels: thread b < NBL carries its bucket's (number of sums, index of the first one in P)
    uint32_t have = tk, first = tbase + ts;
    for (uint32_t lvl = 1; lvl <= lv.nl; ++lvl) {
        const uint32_t ck = have > 1 ? (have + kSmallFan - 1) / kSmallFan : 0u;     // a single sum is final
        uint32_t tot_c;
        const uint32_t cs = block_excl_scan_u32(ck, wsum, tid, 1024, tot_c);       // (starts with a barrier)
        if (tid < NBL) { coff[tid] = cs; cnt[tid] = have; cur[tid] = first; }
        if (tid == 0) {
            coff[NBL] = tot_c;
            misc[1] = atomicAdd(&counters[lvl], tot_c);
            slot_chunks[(size_t)slot * kSmallMaxLevels + (lvl - 1)] = make_uint2(misc[1], tot_c);   // this slot's chunks of the level
        }
        __syncthreads();
        const uint32_t cbase = misc[1];
        for (uint32_t ch = tid; ch < tot_c; ch += 1024) {
            const uint32_t b = small_find(coff, NBL, ch), j = ch - coff[b];
            SmallChunk cd;
            cd.first = cur[b] + j * kSmallFan;
            cd.cnt = min(kSmallFan, cnt[b] - j * kSmallFan);
            cdesc[lv.dbase[lvl] + cbase + ch] = cd;
        }
        if (ck) { have = ck; first = lv.pbase[lvl] + cbase + cs; }
    }
    if (tid < NBL) bucket_ref[(size_t)slot * NBL + tid] = have ? first : kSmallNone;
}

// One fold level: out[chunk] = sum of the chunk's <= 16 sums.  GS logical lanes per chunk (lane `sub` adds the sums
// sub, sub + GS, ..., a shuffle tree adds the GS lane sums); QUAD: every logical lane is a quad (ecquad.hpp) --
// 2.5x shorter dependent chains for 1.6x the work, for launches that cannot fill the chip anyway.
template <bool QUAD, int GS>
__global__ __launch_bounds__(256) void msm_small_fold_kernel(XYZZ* __restrict__ P, const SmallChunk* __restrict__ cdesc,
                                                             const uint32_t* __restrict__ count, uint32_t out_base, int arith29) {
    constexpr uint32_t LPL = QUAD ? 4 : 1;         // lanes per logical lane
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t group = gt / (LPL * GS), sub = (gt / LPL) % GS, q = gt & 3;
    if (group >= *count) return;               // whole groups leave together
    const SmallChunk cd = cdesc[group];
    if constexpr (QUAD) {
        if (arith29) {
            // the quad additions on the lazy 29-bit limbs (ec29l.hpp): operands enter by re-limbing, the sums travel between lanes
            // as 4 x 9 limbs, lane q of the quad turns coordinate q back into wire words at the end
            P29 acc = p29_inf();
            for (uint32_t k = sub; k < cd.cnt; k += GS) { const P29 v = p29_load(P[cd.first + k]); p29_add_quad(acc, v, q); }
#pragma unroll
            for (int o = GS / 2; o > 0; o >>= 1) {
                P29 v;
                const uint32_t* sp = reinterpret_cast<const uint32_t*>(&acc);
                uint32_t* dp = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                for (int w = 0; w < 36; ++w) dp[w] = (uint32_t)__shfl_down((int)sp[w], (int)LPL * o);
                if (sub + (uint32_t)o < (uint32_t)GS) p29_add_quad(acc, v, q);        // see fold_partials
            }
            if (sub == 0) {
                // (every lane of the quad holds the same sum; infinity: all four coordinates zero)
                const bool inf = p29_is_inf(acc);
                const Fp cq = inf ? Fq::zero() : p29_coord_to_fp(acc, q);
                reinterpret_cast<Fp*>(&P[out_base + group])[q] = cq;
            }
            return;
        }
    }
    XYZZ acc = xyzz_inf();
    for (uint32_t k = sub; k < cd.cnt; k += GS) {
        const XYZZ v = P[cd.first + k];
        if constexpr (QUAD) xyzz_add_quad(acc, v, q); else xyzz_add(acc, v);
    }
#pragma unroll
    for (int o = GS / 2; o > 0; o >>= 1) {
        const XYZZ v = xyzz_shfl_down(acc, (int)LPL * o);
        if (sub + (uint32_t)o < (uint32_t)GS) {        // see fold_partials
            if constexpr (QUAD) xyzz_add_quad(acc, v, q); else xyzz_add(acc, v);
        }
    }
    if (sub == 0 && (!QUAD || q == 0)) P[out_base + group] = acc;
}

// grid (slots).  First the slot's chunks of the fold levels >= 2 (they exist only where one bucket holds more than 16
// task sums -- skewed scalars; four quads per chunk), then B_b = the bucket's one remaining sum and
// window sum = sum_{b >= 1} b * B_b = sum over t of the inclusive suffix sums Suffix_t = B_(t+1) + B_(t+2) + ... (t = b - 1):
// log2(NBL) scan steps, then a tree over the NBL suffixes.
// QUAD: min(NBL, 64) quads (block 4 min(NBL, 64)); else one lane per bucket (block NBL).  Dynamic LDS 128 B per quad / lane.
// QMODE: 0 = one lane per bucket, 1 = quads on the 8 x 32-bit arithmetic (ecquad.hpp), 2 = quads on 29-bit limbs
// (ecquad29.hpp: buckets converted as they are loaded, the window sum converted back by the four lanes of quad 0).
template <int QMODE, int MAXT>
__global__ __launch_bounds__(MAXT) void msm_small_reduce_kernel(XYZZ* __restrict__ P, const uint32_t* __restrict__ bucket_ref,
                                                                const SmallChunk* __restrict__ cdesc,
                                                                const uint2* __restrict__ slot_chunks, SmallLevels lv,
                                                                XYZZ* __restrict__ win_sums, uint32_t NBL) {
    constexpr bool QUAD = QMODE != 0;
    static_assert(sizeof(P29) == 144 && sizeof(X29) == 144, "dynamic LDS: 144 bytes per quad in the 29-bit modes");
    extern __shared__ uint4 sm_red[];
    XYZZ* sh = reinterpret_cast<XYZZ*>(sm_red);
    const uint32_t slot = blockIdx.x, tid = threadIdx.x;
    const uint32_t t = QUAD ? tid >> 2 : tid, q = tid & 3;
    const bool writer = !QUAD || q == 0;
    for (uint32_t lvl = 2; lvl <= lv.nl; ++lvl) {
        const uint2 rng = slot_chunks[(size_t)slot * kSmallMaxLevels + (lvl - 1)];
        if (rng.y) {                                           // uniform over the workgroup
            constexpr uint32_t LPL = QUAD ? 4 : 1, GS = 4;     // lanes per logical lane, logical lanes per chunk
            const uint32_t sub = (tid / LPL) % GS, per_round = blockDim.x / (LPL * GS);
            for (uint32_t ch = tid / (LPL * GS); ch < rng.y; ch += per_round) {
                const SmallChunk cd = cdesc[lv.dbase[lvl] + rng.x + ch];
                XYZZ acc = xyzz_inf();
                for (uint32_t k = sub; k < cd.cnt; k += GS) {
                    const XYZZ v = P[cd.first + k];
                    if constexpr (QUAD) xyzz_add_quad(acc, v, q); else xyzz_add(acc, v);
                }
#pragma unroll
                for (int o = GS / 2; o > 0; o >>= 1) {
                    const XYZZ v = xyzz_shfl_down(acc, (int)LPL * o);
                    if (sub + (uint32_t)o < GS) {
                        if constexpr (QUAD) xyzz_add_quad(acc, v, q); else xyzz_add(acc, v);
                    }
                }
                if (sub == 0 && writer) P[lv.pbase[lvl] + rng.x + ch] = acc;
            }
            __threadfence();                                   // the sums are read by other waves of this workgroup
        }
        __syncthreads();
    }
    if constexpr (QMODE == 3) {
        // (QMODE 2's scheme with the buckets re-limbed into the 2^261-form instead of converted by a product: ec29l.hpp)
        // Q = min(NBL, 64) quads -- 256 lanes, ONE wave per SIMD (a second wave on the SIMD doubles the time of every
        // dependent step) -- quad t owns the r = NBL / Q consecutive buckets t r .. t r + r - 1 (array index i = b - 1):
        //   run_t = sum_j B, acc_t = sum_j (j + 1) B            (running sums from the top bucket down: 2 (r - 1) additions)
        //   sum_b b B_b = sum_t acc_t + r sum_{t >= 1} Suf_t,   Suf_t = run_t + run_(t+1) + ...   (log2 Q scan steps)
        // then a tree over V_t = acc_t + r Suf_t (t >= 1), V_0 = acc_0.
        const uint32_t Q = blockDim.x >> 2, r = NBL / Q;
        P29* sh29 = reinterpret_cast<P29*>(sm_red);
        P29 run = p29_inf(), acc = p29_inf();
        for (uint32_t j = r; j-- > 0;) {
            const uint32_t ref = bucket_ref[(size_t)slot * NBL + t * r + j];
            if (ref != kSmallNone) { const XYZZ bk = P[ref]; p29_add_quad(run, p29_load(bk), q); }
            if (r > 1) p29_add_quad(acc, run, q); else acc = run;
        }
        if (writer) sh29[t] = run;
        __syncthreads();
        for (uint32_t off = 1; off < Q; off <<= 1) {
            const bool has = t + off < Q;
            P29 v;
            if (has) v = sh29[t + off];
            __syncthreads();
            if (has) {
                p29_add_quad(run, v, q);
                if (writer) sh29[t] = run;
            }
            __syncthreads();
        }
        if (t >= 1) {
            for (uint32_t d = 1; d < r; d <<= 1) p29_dbl_quad(run, q);
            p29_add_quad(acc, run, q);
        }
        if (writer) sh29[t] = acc;
        __syncthreads();
        for (uint32_t s2 = Q >> 1; s2 > 0; s2 >>= 1) {
            if (t < s2) {
                const P29 v = sh29[t + s2];
                p29_add_quad(acc, v, q);
                if (writer) sh29[t] = acc;
            }
            __syncthreads();
        }
        if (t == 0) reinterpret_cast<Fp*>(&win_sums[slot])[q] = p29_coord_to_fp(acc, q);      // x | y | zz | zzz by the quad's four lanes
    } else     if constexpr (QMODE == 2) {
        // Q = min(NBL, 64) quads -- 256 lanes, ONE wave per SIMD (a second wave on the SIMD doubles the time of every
        // dependent step) -- quad t owns the r = NBL / Q consecutive buckets t r .. t r + r - 1 (array index i = b - 1):
        //   run_t = sum_j B, acc_t = sum_j (j + 1) B            (running sums from the top bucket down: 2 (r - 1) additions)
        //   sum_b b B_b = sum_t acc_t + r sum_{t >= 1} Suf_t,   Suf_t = run_t + run_(t+1) + ...   (log2 Q scan steps)
        // then a tree over V_t = acc_t + r Suf_t (t >= 1), V_0 = acc_0.
        const uint32_t Q = blockDim.x >> 2, r = NBL / Q;
        X29* sh29 = reinterpret_cast<X29*>(sm_red);
        X29 run = x29_inf(), acc = x29_inf();
        for (uint32_t j = r; j-- > 0;) {
            const uint32_t ref = bucket_ref[(size_t)slot * NBL + t * r + j];
            if (ref != kSmallNone) { const XYZZ bk = P[ref]; x29_add_quad(run, x29_from_xyzz_quad(bk, q), q); }
            if (r > 1) x29_add_quad(acc, run, q); else acc = run;
        }
        if (writer) sh29[t] = run;
        __syncthreads();
        for (uint32_t off = 1; off < Q; off <<= 1) {
            const bool has = t + off < Q;
            X29 v;
            if (has) v = sh29[t + off];
            __syncthreads();
            if (has) {
                x29_add_quad(run, v, q);
                if (writer) sh29[t] = run;
            }
            __syncthreads();
        }
        if (t >= 1) {
            for (uint32_t d = 1; d < r; d <<= 1) x29_dbl_quad(run, q);
            x29_add_quad(acc, run, q);
        }
        if (writer) sh29[t] = acc;
        __syncthreads();
        for (uint32_t s2 = Q >> 1; s2 > 0; s2 >>= 1) {
            if (t < s2) {
                const X29 v = sh29[t + s2];
                x29_add_quad(acc, v, q);
                if (writer) sh29[t] = acc;
            }
            __syncthreads();
        }
        if (t == 0) reinterpret_cast<Fp*>(&win_sums[slot])[q] = x29_coord_to_fp(acc, q);      // x | y | zz | zzz by the quad's four lanes
    } else if constexpr (QMODE == 1) {
        // the same on the 8 x 32-bit arithmetic
        const uint32_t Q = blockDim.x >> 2, r = NBL / Q;
        XYZZ run = xyzz_inf(), acc = xyzz_inf();
        for (uint32_t j = r; j-- > 0;) {
            const uint32_t ref = bucket_ref[(size_t)slot * NBL + t * r + j];
            if (ref != kSmallNone) { const XYZZ bk = P[ref]; xyzz_add_quad(run, bk, q); }
            if (r > 1) xyzz_add_quad(acc, run, q); else acc = run;
        }
        if (writer) sh[t] = run;
        __syncthreads();
        for (uint32_t off = 1; off < Q; off <<= 1) {
            const bool has = t + off < Q;
            XYZZ v;
            if (has) v = sh[t + off];
            __syncthreads();
            if (has) {
                xyzz_add_quad(run, v, q);
                if (writer) sh[t] = run;
            }
            __syncthreads();
        }
        if (t >= 1) {
            for (uint32_t d = 1; d < r; d <<= 1) xyzz_dbl_quad(run, q);
            xyzz_add_quad(acc, run, q);
        }
        if (writer) sh[t] = acc;
        __syncthreads();
        for (uint32_t s2 = Q >> 1; s2 > 0; s2 >>= 1) {
            if (t < s2) {
                const XYZZ v = sh[t + s2];
                xyzz_add_quad(acc, v, q);
                if (writer) sh[t] = acc;
            }
            __syncthreads();
        }
        if (tid == 0) win_sums[slot] = acc;
    } else {
        const uint32_t ref = bucket_ref[(size_t)slot * NBL + t];
        XYZZ mine = ref == kSmallNone ? xyzz_inf() : P[ref];
        sh[t] = mine;
        __syncthreads();
        for (uint32_t off = 1; off < NBL; off <<= 1) {
            const bool has = t + off < NBL;
            XYZZ v;
            if (has) v = sh[t + off];
            __syncthreads();
            if (has) {
                xyzz_add(mine, v);
                sh[t] = mine;
            }
            __syncthreads();
        }
        for (uint32_t s2 = NBL >> 1; s2 > 0; s2 >>= 1) {
            if (t < s2) {
                const XYZZ v = sh[t + s2];
                xyzz_add(mine, v);
                sh[t] = mine;
            }
            __syncthreads();
        }
        if (tid == 0) win_sums[slot] = mine;
    }
}

// ---- precomputation: T[j][i] = 2^c * T[j-1][i], affine --------------------------------------------
__device__ inline Fp fq_inv_pow(const Fp& a) {   // a^(p-2)
    const uint32_t e[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                           0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fp acc = Fq::one();
    for (int i = 253; i >= 0; --i) {
        acc = Fq::sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = Fq::mul(acc, a);
    }
    return acc;
}
constexpr int kPreRun = 16;   // points normalised per lane with one inversion (Montgomery's trick)
__global__ __launch_bounds__(64) void msm_precompute_kernel(const Affine* __restrict__ prev, Affine* __restrict__ out,
                                                            Fp* __restrict__ tmp_z, Fp* __restrict__ tmp_p, uint32_t n,
                                                            int c) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * kPreRun;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + kPreRun);
    Fp prod = Fq::one();
    for (uint32_t i = lo; i < hi; ++i) {
        XYZZ a = xyzz_from_affine(prev[i]);
        for (int d = 0; d < c; ++d) a = xyzz_dbl(a);
        Affine xy;
        Fp z;
        if (xyzz_is_inf(a)) {            // infinity base stays infinity in every window
            xy.x = Fq::zero(); xy.y = Fq::zero(); z = Fq::one();
        } else {
            xy.x = Fq::mul(a.x, a.zz);   // Jacobian-like (X*ZZ, Y*ZZZ, Z = ZZ)
            xy.y = Fq::mul(a.y, a.zzz);
            z = a.zz;
        }
        out[i] = xy;
        tmp_z[i] = z;
        tmp_p[i] = prod;
        prod = Fq::mul(prod, z);
    }
    Fp inv = fq_inv_pow(prod);
    for (uint32_t i = hi; i-- > lo;) {
        Fp zi = Fq::mul(inv, tmp_p[i]);
        inv = Fq::mul(inv, tmp_z[i]);
        Fp zi2 = Fq::sqr(zi);
        Affine xy = out[i];
        xy.x = Fq::mul(xy.x, zi2);
        xy.y = Fq::mul(xy.y, Fq::mul(zi2, zi));
        out[i] = xy;
    }
}

// The same table for a small SRS in ONE launch: lane i walks point i through all W - 1 levels (c doublings each),
// leaves the Jacobian-like values and their Z in place, and normalises its whole column with a single inversion.
// The level-by-level kernel above needs one inversion per lane and LEVEL and has n / 16 lanes: at the prover's
// n = 2^14 that is 31 launches of 1 ms; this one is ~3 ms in all.
__global__ __launch_bounds__(64) void msm_precompute_column_kernel(Affine* __restrict__ table, Fp* __restrict__ tmp_z,
                                                                   Fp* __restrict__ tmp_p, uint32_t n, int c, uint32_t W) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XYZZ a = xyzz_from_affine(table[i]);                 // level 0 = the SRS itself
    Fp prod = Fq::one();
    for (uint32_t j = 1; j < W; ++j) {
        for (int d = 0; d < c; ++d) a = xyzz_dbl(a);
        Affine xy;
        Fp z;
        if (xyzz_is_inf(a)) { xy.x = Fq::zero(); xy.y = Fq::zero(); z = Fq::one(); }      // an infinity base stays infinity
        else { xy.x = Fq::mul(a.x, a.zz); xy.y = Fq::mul(a.y, a.zzz); z = a.zz; }          // (X ZZ, Y ZZZ, Z = ZZ)
        const size_t at = (size_t)j * n + i;
        table[at] = xy;
        tmp_z[at] = z;
        tmp_p[at] = prod;                                // product of the z's of the earlier levels
        prod = Fq::mul(prod, z);
    }
    Fp inv = fq_inv_pow(prod);
    for (uint32_t j = W - 1; j >= 1; --j) {
        const size_t at = (size_t)j * n + i;
        const Fp zi = Fq::mul(inv, tmp_p[at]);           // 1 / z_j
        inv = Fq::mul(inv, tmp_z[at]);
        const Fp zi2 = Fq::sqr(zi);
        Affine xy = table[at];
        xy.x = Fq::mul(xy.x, zi2);
        xy.y = Fq::mul(xy.y, Fq::mul(zi2, zi));
        table[at] = xy;
    }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// Window size of the general mode, from measured sweeps (tools/sweep_c.py, tools/ab_msm.py, MI355X).
// Besides the W*n additions what matters is the population of the TOP window (254 mod c bits): a
// narrow top window piles n / 2^bits points into each of its few buckets and both the sort and the
// fold levels pay for it, so only sizes with a wide top window are used
// (c = 8: 6 bits, c = 15: 14 bits, c = 16: 14 bits, c = 17: 16 bits and only 15 windows).
static int choose_window_bits(size_t n, int forced) {
    if (forced >= 4 && forced <= 22) return forced;
    int lg = 0;
    while ((1ull << (lg + 1)) <= n) ++lg;
    // measured with the class-sum reduction (tools/sweep_c_large.py, profiles/r03c_sweep_c.txt): 2^17: c = 15 0.68 ms against
    // 0.72 at c = 8; 2^21: c = 17 3.01 against 3.04 at c = 16
    if (lg <= 16) return 8;
    if (lg <= 18) return 15;
    if (lg <= 20) return 16;
    return 17;
}
int msm_precompute_window_bits(size_t n, int forced) {
    if (forced >= 4 && forced <= 24) return forced;
    if (n <= (1u << 15)) return n <= 256 ? 5 : n <= 2048 ? 7 : 8;   // the small pipeline's window (small_window_bits)
    int lg = 0;
    while ((1ull << (lg + 1)) <= n) ++lg;
    return std::max(8, std::min(22, lg - 2));
}

void msm_free(Ctx& c) {
    if (!c.msm) return;
    for (int q = 0; q < 2; ++q) {
        MsmWork* m = &c.msm[q];
        m->digits.release(); m->bucket_count.release(); m->bucket_start.release(); m->sorted.release();
        m->buckets.release(); m->partials.release(); m->win_sums.release(); m->class_sums.release(); m->big.release(); m->small.release(); m->task_desc.release(); m->exc.release();
        for (int k = 0; k < 2; ++k) {
            m->ent[k].release(); m->lvl_cnt[k].release(); m->lvl_off[k].release(); m->lvl_part[k].release();
        }
        for (int k = 0; k < kMaxPasses; ++k) {
            m->counts[k].release(); m->segs_start[k].release(); m->segs_len[k].release(); m->items[k].release();
        }
        if (m->h_sums) (void)hipHostFree(m->h_sums);
        if (m->h_max) (void)hipHostFree(m->h_max);
    }
    delete[] c.msm;
    c.msm = nullptr;
    if (c.stream2) { (void)hipStreamDestroy(c.stream2); c.stream2 = nullptr; }
}

// Builds the window table of a registered SRS: table[j*n + i] = 2^(cb*j) * P_i, j < W.
int msm_build_table(Ctx& c, const Affine* d_points, size_t n, int cb, Affine** table_out, uint32_t* W_out) {
    const uint32_t W = (uint32_t)msm_num_windows(cb);
    if ((uint64_t)W * n >= (1ull << 31)) {
        set_error("precompute: W*n = %llu exceeds 2^31 - 1", (unsigned long long)W * n);
        return UZK_ERR_PARAMETER;
    }
    Affine* table = nullptr;
    UZK_HIP(hipMalloc(reinterpret_cast<void**>(&table), (size_t)W * n * sizeof(Affine)));
    Fp *tz = nullptr, *tp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&tz), n * sizeof(Fp));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tp), n * sizeof(Fp));
    if (e == hipSuccess) e = hipMemcpyAsync(table, d_points, n * sizeof(Affine), hipMemcpyDeviceToDevice, c.stream);
    if (e == hipSuccess && n <= (1u << 16) && W > 1) {
        // small SRS: whole columns in one launch (the scratch arrays then hold W * n elements each)
        (void)hipFree(tz); (void)hipFree(tp); tz = tp = nullptr;
        e = hipMalloc(reinterpret_cast<void**>(&tz), (size_t)W * n * sizeof(Fp));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tp), (size_t)W * n * sizeof(Fp));
        if (e == hipSuccess) {
            KernelScope ks(c, "msm_precompute");
            hipLaunchKernelGGL(msm_precompute_column_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, table, tz, tp,
                               (uint32_t)n, cb, W);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    } else if (e == hipSuccess) {
        const uint32_t threads = (uint32_t)((n + kPreRun - 1) / kPreRun);
        for (uint32_t j = 1; j < W; ++j) {
            KernelScope ks(c, "msm_precompute");
            hipLaunchKernelGGL(msm_precompute_kernel, dim3((threads + 63) / 64), dim3(64), 0, c.stream,
                               table + (size_t)(j - 1) * n, table + (size_t)j * n, tz, tp, (uint32_t)n, cb);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    }
    if (tz) (void)hipFree(tz);
    if (tp) (void)hipFree(tp);
    if (e != hipSuccess) {
        (void)hipFree(table);
        set_error("precompute failed: %s", hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    *table_out = table;
    *W_out = W;
    return UZK_OK;
}

struct SortPass { uint32_t shift, bins, nseg, nch, chunk, items_bound; };
// register capacities (TB * E) of the msm_radix_segment_kernel instantiations, smallest first
constexpr int kSegCfgs = 6;
constexpr uint32_t kSegCap[kSegCfgs] = {128 * 12, 256 * 12, 256 * 24, 256 * 40, 512 * 40, 512 * 72};

// One pipeline instance: the windows [w0, w0 + W) of every scalar vector, on its own stream with its
// own workspace.  One instance per call by default; an experimental two-instance mode lets the second
// group's digit extraction and sort run underneath the first group's bucket accumulation.
struct MsmGroup {
    MsmWork* m = nullptr;
    hipStream_t st = nullptr;
    // geometry
    bool pre = false;
    int cb = 0;
    uint32_t W_total = 0, w0 = 0, W = 0, n32 = 0, batch = 0;
    uint64_t per_poly = 0, entries = 0, TBK = 0, bound0 = 0, part_cap = 0;
    uint32_t kb = 0, NBL = 0, S0 = 0, seg_n = 0, NB = 0, Wd = 0, RW = 0, seg = 0, groups = 0, L = 0;
    int P = 0;
    bool scan_reduce = false, quad_reduce = false;
    bool direct = false;                                    // one-task buckets were written by the accumulator itself (task_dst)
    uint32_t class_s = 0, nb2 = 0;                          // > 0: class-sum reduction (msm_class_sums_kernel), 2 * nb2 class sums per window
    uint32_t rl = 256;                                      // reduction lanes (quads) per workgroup
    uint32_t pk_bits = 0;   // > 0: 4-byte packed entries between the two sort passes (index bits)
    SortPass sp[kMaxPasses];
    // state carried from phase 1 to phase 2
    uint32_t *cnt_cur = nullptr, *off_cur = nullptr, *base_cur = nullptr;
    XYZZ* part_cur = nullptr;
};

// idx_span (precomputed mode): the sorted values are table indices pre_stride * j + pre_off + i < idx_span, not positions in the
// scalar vector -- the packed sort entries must have room for those (0: no bound known, unpacked entries)
static int msm_group_plan(Ctx& c, MsmGroup& g, size_t n, uint32_t batch, int cb, bool pre, uint32_t W_total, uint32_t w0,
                          uint32_t W, uint64_t idx_span = 0) {
    g.pre = pre; g.cb = cb; g.W_total = W_total; g.w0 = w0; g.W = W; g.n32 = (uint32_t)n; g.batch = batch;
    g.per_poly = (uint64_t)W * n;
    g.entries = g.per_poly * batch;
    if (g.entries >= (1ull << 31)) { set_error("msm: batch*W*n overflows the 31-bit index space"); return UZK_ERR_PARAMETER; }
    g.kb = (uint32_t)cb - 1;
    g.NBL = 1u << g.kb;
    g.S0 = pre ? batch : batch * W;                         // sort segments
    g.seg_n = pre ? (uint32_t)g.per_poly : g.n32;
    // bucket windows of <= 2^15 buckets for the task scans, as many of them as the 1024-entry window table holds
    g.NB = std::min<uint32_t>(g.NBL, 1u << 15);
    while (((uint64_t)g.S0 * g.NBL) / g.NB > 1024 && g.NB < (1u << 15)) g.NB <<= 1;
    g.Wd = (uint32_t)(((uint64_t)g.S0 * g.NBL) / g.NB);
    g.TBK = (uint64_t)g.Wd * g.NB;
    if (g.Wd > 1024) { set_error("msm: too many bucket windows (%u): lower the batch", g.Wd); return UZK_ERR_PARAMETER; }
    g.RW = pre ? batch : batch * W;                         // logical windows in the reduction
    // scan-based reduction: segments of 8 buckets per lane (power of two), at most 256 workgroups per window
    // (measured: the scan form wins for windows of <= 2^14 buckets, the double-and-add form above that).
    // Up to 2^19 buckets in all, the scans run on quads (ecquad.hpp): segments of <= 16 buckets, one quad each, still
    // fit the chip at two waves per SIMD (32768 quads), and every dependent addition takes half the time -- reduce
    // 0.16 -> 0.09 ms at n = 2^16, 0.55 -> 0.35 ms at 2^19; beyond that the segments get long or the quads queue behind
    // each other, and the lane form is as fast (measured at 2^20 and 2^24: 0.59 vs 0.61 ms at best).
    const uint64_t total_buckets = (uint64_t)g.RW * g.NBL;
    g.quad_reduce = total_buckets <= (1u << 19);
    g.scan_reduce = g.quad_reduce || g.NBL <= (1u << 14);
    if (g.scan_reduce) {
        uint32_t sg = 8u;
        if (g.quad_reduce) {
            sg = 1;
            while ((uint64_t)sg * 32768 < total_buckets) sg <<= 1;
        }
        while (sg & (sg - 1)) sg &= sg - 1;                                  // power of two
        g.rl = g.quad_reduce ? kQuadLanes : 256u;
        // small windows (c = 8: 128 buckets): one workgroup per window, its partial needs no folding (0.094 -> 0.071 ms)
        if (g.quad_reduce && g.NBL <= 4 * g.rl) sg = std::max<uint32_t>(sg, g.NBL / g.rl);
        sg = std::max<uint32_t>(1, std::min<uint32_t>(sg, g.NBL / g.rl));
        while ((g.NBL + sg * g.rl - 1) / (sg * g.rl) > g.rl) sg <<= 1;
        g.seg = sg;
    } else {
        g.seg = std::max<uint32_t>(1, std::min<uint32_t>(kSeg, g.NBL / 256));
    }
    g.groups = (g.NBL + g.seg * g.rl - 1) / (g.seg * g.rl);
    // windows of >= 2^12 buckets: class sums first (2 x 256 sums per window), the scans above on those
    g.class_s = 0; g.nb2 = 0;
    if (g.NBL >= 4096 && g.NBL <= (1u << 18) && (uint64_t)g.RW * 2 <= 65535) {
        g.class_s = 8;
        g.nb2 = std::max<uint32_t>(256, g.NBL >> 8);
    }
    const uint64_t all_entries = (uint64_t)W_total * n * batch;
    // Task length: long enough that a typical bucket (4x the mean population) is ONE task -- its partial
    // sum then needs no folding -- but short enough that there are >= ~200k tasks to fill the chip
    // (measured: task-length sweeps at 2^14 .. 2^24, profiles/r01*_sweep*.txt).  Larger buckets are split evenly.
    const uint64_t mean_pop = (pre ? (uint64_t)W_total * n : (uint64_t)n) >> g.kb;
    g.L = (uint32_t)std::max<uint64_t>(16, std::min<uint64_t>(256, std::min<uint64_t>(4 * mean_pop, all_entries / 200000)));
    g.bound0 = g.entries / g.L + g.TBK;                     // upper bound on level-0 tasks
    g.part_cap = g.bound0 + 2 * g.TBK;                      // every later level fits too
    g.P = g.kb <= 9 ? 1 : (g.kb <= 18 ? 2 : 3);             // radix passes of <= 9 bits, high bits first
    // Two-pass sorts: take 9 bits first when that lets {remaining key bits, sign, index} fit 32 bits --
    // the entries between the passes are then 4 bytes instead of 8 (a third of the sort's traffic less).
    g.pk_bits = 0;
    if (g.P == 2 && (!pre || idx_span)) {
        uint32_t ib = 1;
        while ((1ull << ib) < (pre ? idx_span : (uint64_t)g.seg_n)) ++ib;
        const uint32_t first_bits = std::min<uint32_t>(9, g.kb - 1);
        if (ib + 1 + (g.kb - first_bits) <= 32) g.pk_bits = ib;
    }
    uint32_t rem = g.kb, nseg = g.S0;
    for (int p = 0; p < g.P; ++p) {
        uint32_t bits = (rem + (uint32_t)(g.P - p) - 1) / (uint32_t)(g.P - p);
        if (g.pk_bits && p == 0) bits = std::min<uint32_t>(9, g.kb - 1);
        rem -= bits;
        g.sp[p].shift = rem;
        g.sp[p].bins = 1u << bits;
        g.sp[p].nseg = nseg;
        const uint64_t avg = std::max<uint64_t>(1, (g.S0 * (uint64_t)g.seg_n) / nseg);
        g.sp[p].nch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024, avg / 32768));
        // packed two-pass sorts: chunks of <= 32768 digits, what one workgroup of msm_radix_chunk_kernel takes into registers
        if (p == 0 && g.pk_bits && avg <= (1ull << 25)) g.sp[p].nch = (uint32_t)((avg + 32767) / 32768);
        g.sp[p].chunk = (p == g.P - 1) ? 16384u : 32768u;   // later passes: fixed-size work items
        g.sp[p].items_bound = (uint32_t)(g.entries / g.sp[p].chunk) + nseg;
        nseg *= g.sp[p].bins;
    }
    MsmWork& m = *g.m;
    UZK_TRY(m.digits.reserve((size_t)g.entries * 4));
    UZK_TRY(m.sorted.reserve((size_t)g.entries * 4));
    UZK_TRY(m.bucket_count.reserve((size_t)g.TBK * 4));
    UZK_TRY(m.bucket_start.reserve((size_t)g.TBK * 4));
    UZK_TRY(m.buckets.reserve((size_t)g.TBK * sizeof(XYZZ)));
    UZK_TRY(m.partials.reserve((size_t)2 * g.RW * std::max<uint32_t>(g.groups, g.class_s ? 2 * (g.nb2 / 256 + 1) : 0) * sizeof(XYZZ)));
    UZK_TRY(m.win_sums.reserve((size_t)g.RW * 2 * sizeof(XYZZ)));
    if (g.class_s) UZK_TRY(m.class_sums.reserve((size_t)g.RW * 2 * g.nb2 * sizeof(XYZZ)));
    for (int p = 0; p < g.P; ++p) {
        UZK_TRY(m.counts[p].reserve((size_t)(p == 0 ? 2 * g.sp[p].nseg * g.sp[p].nch : g.sp[p].items_bound) * g.sp[p].bins * 4));
        if (p > 0) UZK_TRY(m.items[p].reserve(((size_t)g.sp[p].nseg + 1) * 4));
        if (p + 1 < g.P) {
            UZK_TRY(m.segs_start[p].reserve((size_t)g.sp[p].nseg * g.sp[p].bins * 4));
            UZK_TRY(m.segs_len[p].reserve((size_t)g.sp[p].nseg * g.sp[p].bins * 4));
            UZK_TRY(m.ent[p & 1].reserve((size_t)g.entries * sizeof(uint2)));
        }
    }
    for (int k = 0; k < 2; ++k) {
        UZK_TRY(m.lvl_cnt[k].reserve((size_t)g.TBK * 4));
        UZK_TRY(m.lvl_off[k].reserve((size_t)g.TBK * 4));
    }
    UZK_TRY(m.lvl_part[0].reserve((size_t)g.part_cap * sizeof(XYZZ)));
    UZK_TRY(m.small.reserve(16384));
    UZK_TRY(m.task_desc.reserve((size_t)g.bound0 * sizeof(TaskDesc)));
    UZK_TRY(m.exc.reserve((size_t)g.bound0 * 4));
    if (m.h_sums_cap < (size_t)g.RW * 2) {
        if (m.h_sums) (void)hipHostFree(m.h_sums);
        UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&m.h_sums), (size_t)g.RW * 2 * sizeof(XYZZ), hipHostMallocDefault));
        m.h_sums_cap = (size_t)g.RW * 2;
    }
    if (!m.h_max) UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&m.h_max), 64, hipHostMallocDefault));
    return UZK_OK;
}

// Phase 1 (asynchronous on g.st): digits, sort, task schedule, bucket accumulation, copy of the
// largest bucket population to the host.
// `direct_ok`: phase 2 will WRITE the bucket array (not add onto an earlier chunk's), so one-task buckets may be written here
static int msm_group_phase1(Ctx& c, MsmGroup& g, const Affine* points, const ScalarView& d_scalars, uint32_t pre_stride,
                            uint32_t pre_off, bool direct_ok = true) {
    MsmWork& m = *g.m;
    hipStream_t st = g.st;
    c.cur_stream = st;
    uint32_t* digits = m.digits.as<uint32_t>();
    uint32_t* sorted = m.sorted.as<uint32_t>();
    uint32_t* bcount = m.bucket_count.as<uint32_t>();
    uint32_t* bstart = m.bucket_start.as<uint32_t>();
    // small: [0..1024) window totals, [1024..2049) win_base ping, [2080..3105) win_base pong, [3200] max,
    //        [3328..3584) task-length histogram, [3584..3840) cursors
    uint32_t* sm = m.small.as<uint32_t>();
    uint32_t* win_tot = sm;
    uint32_t* d_max = sm + 3200;
    // ONE fill for everything this phase counts in: [3200] max, [3328, 3584) the task-length histogram, [3584, 3840) its cursors,
    // [3900] the exception count, [3908] the big-bucket count (four fills before round 5: three launches less per MSM)
    UZK_HIP(hipMemsetAsync(d_max, 0, (3912 - 3200) * 4, st));
    // general mode, whole window range, chunks of >= 32768 scalars: digits and pass-0 histograms in one kernel
    const bool fused_hist = !g.pre && g.w0 == 0 && g.W == g.W_total && g.sp[0].nch >= 256 &&
                            (size_t)g.W * g.sp[0].bins * 4 <= 64 * 1024;
    if (fused_hist) {
        KernelScope ks(c, "msm_digits");
        hipLaunchKernelGGL(msm_digits_hist_kernel, dim3(g.sp[0].nch, g.batch), dim3(1024), (size_t)g.W * g.sp[0].bins * 4, st,
                           d_scalars, digits, g.n32, g.cb, (int)g.W, g.sp[0].shift, g.sp[0].bins, m.counts[0].as<uint32_t>());
    } else {
        KernelScope ks(c, "msm_digits");
        const uint64_t tot = (uint64_t)g.n32 * g.batch;
        hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_scalars, digits, g.n32,
                           g.batch, g.cb, (int)g.W_total, (int)g.w0, (int)g.W, (uint32_t*)nullptr);
    }
    // ---- counting sort, high bits first
    for (int p = 0; p < g.P; ++p) {
        const bool first = (p == 0), last = (p == g.P - 1);
        const SortPass& sp = g.sp[p];
        RadixArgs a{};
        a.shift = sp.shift; a.bins = sp.bins; a.mask = sp.bins - 1;
        a.counts = m.counts[p].as<uint32_t>();
        a.prefix = first ? a.counts + (size_t)sp.nseg * sp.nch * sp.bins : a.counts;   // first pass: several scan workgroups per segment
        if (first) {
            a.digits = digits;
            a.n = g.seg_n;
            if (g.pre) { a.remap_cnt = g.n32; a.remap_stride = pre_stride; a.remap_off = pre_off; }
        } else {
            a.in_entries = m.ent[(p - 1) & 1].as<uint2>();
            a.seg_start = m.segs_start[p - 1].as<uint32_t>();
            a.seg_len = m.segs_len[p - 1].as<uint32_t>();
        }
        if (g.pk_bits) { if (first) a.pk_out_bits = g.pk_bits; else a.pk_in_bits = g.pk_bits; }
        if (last) { a.bin_base = bstart; a.bin_count = bcount; a.out_vals = sorted; }
        else {
            a.bin_base = m.segs_start[p].as<uint32_t>(); a.bin_count = m.segs_len[p].as<uint32_t>();
            a.out_entries = m.ent[p & 1].as<uint2>();
        }
        // Last pass of a packed two-pass sort: segments of the expected length (mean + 8 sigma of a uniform split fits the
        // kernel's registers) are sorted by one workgroup each (msm_radix_segment_kernel); the generic kernels below then
        // only see the longer ones (skewed scalar sets), normally none.
        int seg_cfg = -1;
        if (last && !first && g.P == 2 && a.pk_in_bits && c.tune_seg_sort && sp.bins <= 512) {
            const double mean = (double)g.seg_n / (double)g.sp[0].bins;
            const double need = mean + 8.0 * std::sqrt(mean) + 64.0;
            for (int k = 0; k < kSegCfgs; ++k)
                if ((double)kSegCap[k] >= need) { seg_cfg = k; break; }
            if (c.tune_seg_sort >= 10) seg_cfg = std::min(kSegCfgs - 1, c.tune_seg_sort - 10);   // tests: a given instantiation
            if (seg_cfg >= 0) a.seg_min_len = kSegCap[seg_cfg];
        }
        if (!first) {
            a.item_off = m.items[p].as<uint32_t>();
            a.nseg = sp.nseg;
            a.chunk = sp.chunk;
            KernelScope ks(c, "msm_sort_items");
            hipLaunchKernelGGL(msm_radix_items_kernel, dim3(1), dim3(1024), 0, st, a.seg_len, a.nseg, a.chunk,
                               m.items[p].as<uint32_t>(), a.seg_min_len);
        }
        // later passes: the work-item list is data dependent; the launch covers its bound up to a few workgroups per CU and
        // strides over the rest (behind the segment kernel the list is normally empty: 2048 workgroups that leave at once)
        const dim3 grid = first ? dim3(sp.nch, sp.nseg) : dim3(seg_cfg >= 0 ? std::min<uint32_t>(sp.items_bound, 2048u) : sp.items_bound);
        // Workspace guard (the round-1 fault -- a write past a sort counter array while the two-pass sort was being
        // written, DESIGN.md 3.1 -- must fail here, on the host, not on the device): every workgroup of this pass
        // owns `bins` counters, every segment `bins` entries of bin_base / bin_count, and both output arrays hold
        // all entries.
        {
            const size_t wgs = first ? 2 * (size_t)grid.x * grid.y : (size_t)sp.items_bound;   // counters: one set per work item (+ prefixes)
            const size_t ent_bytes = g.pk_bits ? 4 : sizeof(uint2);
            const bool ok = m.counts[p].cap >= wgs * sp.bins * 4 &&
                            (last ? (m.bucket_start.cap >= (size_t)sp.nseg * sp.bins * 4 && m.bucket_count.cap >= (size_t)sp.nseg * sp.bins * 4 &&
                                     m.sorted.cap >= (size_t)g.entries * 4)
                                  : (m.segs_start[p].cap >= (size_t)sp.nseg * sp.bins * 4 && m.segs_len[p].cap >= (size_t)sp.nseg * sp.bins * 4 &&
                                     m.ent[p & 1].cap >= (size_t)g.entries * ent_bytes)) &&
                            (first || m.items[p].cap >= ((size_t)sp.nseg + 1) * 4);
            if (!ok) { set_error("msm: sort pass %d: a workspace is smaller than its launch needs (internal error)", p); return UZK_ERR_DEVICE; }
        }
        if (seg_cfg >= 0) {
            KernelScope ks(c, "msm_sort_segment");
            const dim3 sgrid(sp.nseg);
            switch (seg_cfg) {
                case 0: hipLaunchKernelGGL((msm_radix_segment_kernel<128, 12, 1536>), sgrid, dim3(128), 0, st, a); break;
                case 1: hipLaunchKernelGGL((msm_radix_segment_kernel<256, 12, 3072>), sgrid, dim3(256), 0, st, a); break;
                case 2: hipLaunchKernelGGL((msm_radix_segment_kernel<256, 24, 6144>), sgrid, dim3(256), 0, st, a); break;
                case 3: hipLaunchKernelGGL((msm_radix_segment_kernel<256, 40, 10240>), sgrid, dim3(256), 0, st, a); break;
                case 4: hipLaunchKernelGGL((msm_radix_segment_kernel<512, 40, 18432>), sgrid, dim3(512), 0, st, a); break;
                default: hipLaunchKernelGGL((msm_radix_segment_kernel<512, 72, 18432>), sgrid, dim3(512), 0, st, a); break;
            }
        }
        if (!(first && fused_hist)) {
            KernelScope ks(c, "msm_sort_hist");
            if (first) hipLaunchKernelGGL((msm_radix_hist_kernel<true, false>), grid, dim3(1024), 0, st, a);
            else if (a.pk_in_bits) hipLaunchKernelGGL((msm_radix_hist_kernel<false, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((msm_radix_hist_kernel<false, false>), grid, dim3(256), 0, st, a);
        }
        {
            KernelScope ks(c, "msm_sort_scan");
            // first pass: few segments with many chunks each -- several workgroups per segment
            const uint32_t scan_g = first ? std::max<uint32_t>(1, std::min<uint32_t>(sp.nch / 16, 1024 / std::max<uint32_t>(1, sp.nseg))) : 1u;
            hipLaunchKernelGGL(msm_radix_scan_kernel, dim3(sp.nseg, scan_g), dim3(512), 0, st, a, sp.nch,
                               first ? (const uint32_t*)nullptr : a.seg_start, first ? g.seg_n : 0u);
        }
        {
            KernelScope ks(c, "msm_sort_scatter");
            const uint32_t cs0 = first ? (g.seg_n + sp.nch - 1) / sp.nch : 0u;
            if (first && !last && a.pk_out_bits && sp.bins <= 512 && cs0 <= 32768 && !g.pre)
                hipLaunchKernelGGL((msm_radix_chunk_kernel<512, 64, 18176>), grid, dim3(512), 0, st, a);
            else if (first && last) hipLaunchKernelGGL((msm_radix_scatter_kernel<1024, 8, true, true, false>), grid, dim3(1024), 0, st, a);
            else if (first && g.entries >= (1ull << 26))   // large sorts: 512 lanes, 4096-entry tiles (measured)
                hipLaunchKernelGGL((msm_radix_scatter_kernel<512, 8, true, false, false>), grid, dim3(512), 0, st, a);
            else if (first) hipLaunchKernelGGL((msm_radix_scatter_kernel<1024, 8, true, false, false>), grid, dim3(1024), 0, st, a);
            else if (last && a.pk_in_bits) hipLaunchKernelGGL((msm_radix_scatter_kernel<512, 8, false, true, true>), grid, dim3(512), 0, st, a);
            else if (last) hipLaunchKernelGGL((msm_radix_scatter_kernel<256, 16, false, true, false>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((msm_radix_scatter_kernel<1024, 8, false, false, false>), grid, dim3(1024), 0, st, a);
        }
    }
    // ---- level-0 tasks (runs of <= L indices), longest first
    g.cnt_cur = m.lvl_cnt[0].as<uint32_t>();
    g.off_cur = m.lvl_off[0].as<uint32_t>();
    g.base_cur = sm + 1024;
    g.part_cur = m.lvl_part[0].as<XYZZ>();
    uint32_t* len_hist = sm + 3328;
    uint32_t* len_cur = sm + 3328 + kLenBins;
    g.direct = direct_ok;
    XYZZ* direct_buckets = g.direct ? m.buckets.as<XYZZ>() : nullptr;
    {
        KernelScope ks(c, "msm_scan_win");
        hipLaunchKernelGGL(msm_scan_win_kernel, dim3(g.Wd), dim3(1024), scan_win_lds(g.NB), st, bcount, g.L, g.cnt_cur, g.off_cur, win_tot,
                           d_max, g.NB, len_hist);
        hipLaunchKernelGGL(msm_win_base_kernel, dim3(1), dim3(1024), 0, st, win_tot, g.base_cur, g.Wd);
    }
    TaskDesc* desc = m.task_desc.as<TaskDesc>();
    {
        // one lane per bucket; the lengths' histogram came out of msm_scan_win_kernel
        KernelScope ks(c, "msm_task_order");
        uint32_t* big_count = sm + 3908;
        UZK_TRY(m.big.reserve((size_t)(g.TBK + 1) * sizeof(BigBucket)));        // (shared with the fold lists of phase 2)
        hipLaunchKernelGGL(msm_task_scan_kernel, dim3(1), dim3(256), 0, st, len_hist, len_cur);
        hipLaunchKernelGGL(msm_bucket_fill_kernel, dim3((unsigned)((g.TBK + 255) / 256)), dim3(256), 0, st, g.base_cur, g.cnt_cur, g.off_cur,
                           bstart, bcount, g.NB, g.TBK, len_cur, desc, big_count, m.big.as<BigBucket>(), g.direct ? 1u : 0u);
        hipLaunchKernelGGL(msm_task_fill_big_kernel, dim3(1024), dim3(64), 0, st, big_count, m.big.as<BigBucket>(), len_cur, desc);
    }
    {
        KernelScope ks(c, "msm_accumulate");
        const dim3 grid((unsigned)((g.bound0 + 255) / 256));
        // the 29-bit-limb accumulator + the (normally empty) exception pass
        uint32_t* exc_count = sm + 3900;
        uint32_t* exc_list = m.exc.as<uint32_t>();
        hipLaunchKernelGGL(msm_accumulate29_kernel, grid, dim3(256), 0, st, points, sorted, desc, g.base_cur, g.part_cur,
                           g.Wd, exc_count, exc_list, direct_buckets);
        // (grid-stride over a list that is normally EMPTY: a launch of the accumulator's own grid was 66 us of idle workgroups per
        // MSM in the prover's lockstep commits -- 3.3 % of a proof's kernel time, profiles/r05d_lockstep8_kernel_stats.csv)
        hipLaunchKernelGGL(msm_accumulate_exc_kernel, dim3(std::min<unsigned>(grid.x, 128u)), dim3(256), 0, st, points, sorted, desc, g.part_cur,
                           exc_count, exc_list, direct_buckets);
    }
    UZK_HIP(hipGetLastError());
    // the largest bucket decides how many fold levels are needed (one tiny read-back)
    UZK_HIP(hipMemcpyAsync(m.h_max, d_max, 4, hipMemcpyDeviceToHost, st));
    return UZK_OK;
}

// Phase 2: wait for the read-back, fold the partial sums into the bucket array (`accumulate`: onto what an earlier point
// chunk left there), and -- unless more chunks follow -- reduce the buckets and copy the window sums.
static int msm_group_phase2(Ctx& c, MsmGroup& g, bool accumulate = false, bool reduce = true) {
    const int a29 = (c.tune_arith29 >> 2) & 1;       // the folds' additions on the lazy 29-bit limbs (ec29l.hpp)
    MsmWork& m = *g.m;
    hipStream_t st = g.st;
    c.cur_stream = st;
    UZK_HIP(hipStreamSynchronize(st));
    uint32_t* sm = m.small.as<uint32_t>();
    uint32_t* win_tot = sm;
    uint32_t* win_base[2] = {sm + 1024, sm + 2080};
    XYZZ* buckets = m.buckets.as<XYZZ>();
    XYZZ* partials = m.partials.as<XYZZ>();
    XYZZ* win_sums = m.win_sums.as<XYZZ>();
    const uint32_t G = kCombineFan;
    uint64_t tmax = ((uint64_t)m.h_max[0] + g.L - 1) / g.L;
    // lanes per fold group: latency mode for small problems, one lane per bucket for large ones
    // (measured at n = 2^14, batch 1..8: four lanes per bucket beat sixteen by 2..20 %)
    const uint32_t gs = g.TBK >= (1u << 18) ? 1u : (tmax > 2 ? 4u : 1u);
    // skewed inputs (the extra levels exist only for them; a last level that still folds more than kBigThresh partial sums
    // somewhere): the one-lane-per-output kernels hand their long folds to msm_fold_big_kernel, one wave each
    constexpr uint32_t kBigThresh = 4;
    static_assert(sizeof(BigFold) == sizeof(BigBucket), "the two lists share one buffer");
    const bool big_mode = gs == 1 && tmax > kBigThresh;
    uint32_t* big_count = sm + 3904;
    BigFold* big_list = nullptr;
    if (big_mode) {
        UZK_TRY(m.big.reserve((size_t)(g.bound0 / G + g.TBK + 1) * sizeof(BigFold)));       // every output of a level could be listed
        big_list = m.big.as<BigFold>();
    }
    if (g.direct && tmax > G) {          // extra levels ahead: the directly written one-task sums go back among the partial sums
        KernelScope ks(c, "msm_combine");
        hipLaunchKernelGGL(msm_undirect_kernel, dim3((unsigned)((g.TBK + 255) / 256)), dim3(256), 0, st, buckets, g.cnt_cur, g.off_cur,
                           g.base_cur, g.part_cur, g.NB, g.TBK);
        g.direct = false;
    }
    if (accumulate && g.direct) { set_error("msm: direct bucket writes in an accumulating chunk (internal error)"); return UZK_ERR_DEVICE; }
    const int skip_single = g.direct ? 1 : 0;
    int lvl = 0;
    uint64_t bound_prev = g.bound0;
    while (tmax > G) {
        const int nx = (lvl + 1) & 1;
        UZK_TRY(m.lvl_part[nx].reserve((size_t)g.part_cap * sizeof(XYZZ)));   // no-op for buffer 0
        uint32_t* cnt_nx = m.lvl_cnt[nx].as<uint32_t>();
        uint32_t* off_nx = m.lvl_off[nx].as<uint32_t>();
        uint32_t* base_nx = win_base[nx];
        const uint64_t bound_nx = bound_prev / G + g.TBK;
        XYZZ* part_nx = m.lvl_part[nx].as<XYZZ>();
        {
            KernelScope ks(c, "msm_scan_win");
            hipLaunchKernelGGL(msm_scan_win_kernel, dim3(g.Wd), dim3(1024), scan_win_lds(g.NB), st, g.cnt_cur, G, cnt_nx, off_nx, win_tot,
                               (uint32_t*)nullptr, g.NB);
            hipLaunchKernelGGL(msm_win_base_kernel, dim3(1), dim3(1024), 0, st, win_tot, base_nx, g.Wd);
        }
        {
            KernelScope ks(c, "msm_combine");
            const dim3 grid((unsigned)((bound_nx * gs + 255) / 256));
            if (gs == 16)
                hipLaunchKernelGGL(msm_combine_kernel<16>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                                   off_nx, base_nx, part_nx, g.NB, g.Wd, G, (uint32_t*)nullptr, (BigFold*)nullptr, 0u, a29);
            else if (gs == 4)
                hipLaunchKernelGGL(msm_combine_kernel<4>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                                   off_nx, base_nx, part_nx, g.NB, g.Wd, G, (uint32_t*)nullptr, (BigFold*)nullptr, 0u, a29);
            else {
                if (big_mode) UZK_HIP(hipMemsetAsync(big_count, 0, 4, st));
                hipLaunchKernelGGL(msm_combine_kernel<1>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                                   off_nx, base_nx, part_nx, g.NB, g.Wd, G, big_count, big_list, kBigThresh, a29);
                if (big_mode)
                    hipLaunchKernelGGL(msm_fold_big_kernel, dim3(2048), dim3(64), 0, st, g.part_cur, big_count, big_list, part_nx, 0);
            }
        }
        g.cnt_cur = cnt_nx; g.off_cur = off_nx; g.base_cur = base_nx; g.part_cur = part_nx;
        bound_prev = bound_nx;
        tmax = (tmax + G - 1) / G;
        lvl = nx;
    }
    {
        KernelScope ks(c, "msm_finalize");
        const dim3 grid((unsigned)((g.TBK * gs + 255) / 256));
        if (gs == 16)
            hipLaunchKernelGGL(msm_finalize_kernel<16>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                               buckets, g.NB, g.Wd, accumulate ? 1 : 0, (uint32_t*)nullptr, (BigFold*)nullptr, 0u, skip_single, a29);
        else if (gs == 4)
            hipLaunchKernelGGL(msm_finalize_kernel<4>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                               buckets, g.NB, g.Wd, accumulate ? 1 : 0, (uint32_t*)nullptr, (BigFold*)nullptr, 0u, skip_single, a29);
        else {
            const bool big_last = big_mode && tmax > kBigThresh;
            if (big_last) UZK_HIP(hipMemsetAsync(big_count, 0, 4, st));
            hipLaunchKernelGGL(msm_finalize_kernel<1>, grid, dim3(256), 0, st, g.part_cur, g.cnt_cur, g.off_cur, g.base_cur,
                               buckets, g.NB, g.Wd, accumulate ? 1 : 0, big_count, big_last ? big_list : (BigFold*)nullptr, kBigThresh, skip_single, a29);
            if (big_last)
                hipLaunchKernelGGL(msm_fold_big_kernel, dim3(2048), dim3(64), 0, st, g.part_cur, big_count, big_list, buckets, accumulate ? 1 : 0);
        }
    }
    if (!reduce) { UZK_HIP(hipGetLastError()); return UZK_OK; }
    if (g.class_s) {
        // level 1: class sums; level 2: the quad scans over 2 * RW logical windows of nb2 "buckets" each -- window sums
        // [w][0] = sum_h h R_h and [w][1] = sum_l (l + 1) C_l, combined by the host's Horner (2^s * [0] + [1])
        XYZZ* cls = m.class_sums.as<XYZZ>();
        const uint32_t RW2 = g.RW * 2;
        {
            KernelScope ks(c, "msm_reduce_class");
            hipLaunchKernelGGL(msm_class_sums_kernel, dim3(g.NBL / 2048, g.RW, 2), dim3(256), 0, st, buckets, cls, g.NBL, g.class_s, g.nb2, (c.tune_arith29 >> 2) & 1);
        }
        {
            KernelScope ks(c, "msm_reduce");
            const uint32_t seg2 = 4, log_seg2 = 2, groups2 = (g.nb2 + seg2 * kQuadLanes - 1) / (seg2 * kQuadLanes);
            XYZZ* part_r = partials + (size_t)RW2 * groups2;
            // one workgroup per logical window (nb2 = 256): its P1 partial IS the window sum
            hipLaunchKernelGGL(msm_reduce_scan_quad_kernel, dim3(groups2, RW2), dim3(kQuadLanes * 4), 0, st, cls, groups2 == 1 ? win_sums : partials,
                               part_r, g.nb2, groups2, seg2, log_seg2);
            if (groups2 > 1)
                hipLaunchKernelGGL(msm_fold_scan_quad_kernel, dim3(RW2), dim3(kQuadLanes * 4), 0, st, partials, part_r, win_sums, groups2,
                                   kLogQuadLanes + log_seg2);
        }
        UZK_HIP(hipGetLastError());
        UZK_HIP(hipMemcpyAsync(m.h_sums, win_sums, (size_t)RW2 * sizeof(XYZZ), hipMemcpyDeviceToHost, st));
        return UZK_OK;
    }
    {
        KernelScope ks(c, "msm_reduce");
        if (g.scan_reduce) {
            XYZZ* part_r = partials + (size_t)g.RW * g.groups;
            uint32_t log_seg = 0;
            while ((1u << log_seg) < g.seg) ++log_seg;
            if (g.quad_reduce) {
                hipLaunchKernelGGL(msm_reduce_scan_quad_kernel, dim3(g.groups, g.RW), dim3(kQuadLanes * 4), 0, st, buckets, partials, part_r,
                                   g.NBL, g.groups, g.seg, log_seg);
                hipLaunchKernelGGL(msm_fold_scan_quad_kernel, dim3(g.RW), dim3(kQuadLanes * 4), 0, st, partials, part_r, win_sums, g.groups,
                                   kLogQuadLanes + log_seg);
            } else {
                hipLaunchKernelGGL(msm_reduce_scan_kernel, dim3(g.groups, g.RW), dim3(256), 0, st, buckets, partials, part_r, g.NBL,
                                   g.groups, g.seg, log_seg);
                hipLaunchKernelGGL(msm_fold_scan_kernel, dim3(g.RW), dim3(256), 0, st, partials, part_r, win_sums, g.groups,
                                   8 + log_seg);
            }
        } else {
            hipLaunchKernelGGL(msm_reduce_kernel, dim3(g.groups, g.RW), dim3(256), 0, st, buckets, partials, g.NBL, g.groups, g.seg);
            hipLaunchKernelGGL(msm_fold_partials_kernel, dim3(g.RW), dim3(256), 0, st, partials, win_sums, g.groups);
        }
    }
    UZK_HIP(hipGetLastError());
    UZK_HIP(hipMemcpyAsync(m.h_sums, win_sums, (size_t)g.RW * sizeof(XYZZ), hipMemcpyDeviceToHost, st));
    return UZK_OK;
}

// A few persistent host threads for the per-vector Horner sums of a batched call (spawning
// std::threads per call cost 0.2 ms for 8 vectors -- more than the arithmetic).
namespace {
class HornerPool {
public:
    explicit HornerPool(unsigned nthreads) {
        for (unsigned t = 0; t < nthreads; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~HornerPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    // runs fn(i) for i in [0, count), the caller included; returns when all are done.  Several contexts may be inside at
    // once: every call is a job of its own in the list the workers serve (a caller only works on its own job), so a
    // second prover's batch neither waits for the first one's nor falls back to doing all its sums alone.
    void run(uint32_t count, const std::function<void(uint32_t)>& fn) {
        // every job owns its counters: a worker that wakes late still holds the job it saw under the lock and can only
        // find that job exhausted, never another call's indices
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->count = count;
        { std::lock_guard<std::mutex> lk(mu_); active_.push_back(job); }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return job->done.load() >= count; });
        for (size_t i = 0; i < active_.size(); ++i)
            if (active_[i] == job) { active_.erase(active_.begin() + (long)i); break; }
    }
private:
    struct Job {
        const std::function<void(uint32_t)>* fn = nullptr;
        uint32_t count = 0;
        std::atomic<uint32_t> next{0}, done{0};
    };
    void work(Job& j) {
        for (;;) {
            const uint32_t i = j.next.fetch_add(1);
            if (i >= j.count) break;
            (*j.fn)(i);
            if (j.done.fetch_add(1) + 1 >= j.count) { std::lock_guard<std::mutex> lk(mu_); cv_done_.notify_all(); }
        }
    }
    // a job that still has indices to hand out (mu_ held)
    std::shared_ptr<Job> pending_locked() const {
        for (const auto& j : active_) if (j->next.load() < j->count) return j;
        return nullptr;
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (j = pending_locked()) != nullptr; });
                if (stop_) return;
            }
            work(*j);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::vector<std::shared_ptr<Job>> active_;
    bool stop_ = false;
};
// process-wide, shared by every context, created on first use and joined at process exit
HornerPool& horner_pool() {
    static HornerPool pool(7);
    return pool;
}
}  // namespace

// Host: per scalar vector, Horner over its W window sums (c doublings per step).
// `class_s` > 0: window_sum(b, w) is the first of TWO consecutive sums of the window (class sums: 2^class_s * [0] + [1])
template <class WindowSum>
static void msm_horner_host(Ctx& c, uint32_t batch, uint32_t wpp, int cb, const WindowSum& window_sum, Jac* out_host, int class_s = 0) {
    // 4 x 64-bit host arithmetic, Jacobian doublings (host_ec64.hpp): 57 -> 32 us for the 32 windows of a 2^14-point commit
    auto horner = [&](uint32_t b) {
        if (class_s > 0)
            out_host[b] = h64::horner_split(wpp, std::max(cb, class_s), class_s, [&](uint32_t w) -> const XYZZ& { return (&window_sum(b, w))[0]; },
                                            [&](uint32_t w) -> const XYZZ& { return (&window_sum(b, w))[1]; });
        else
            out_host[b] = h64::horner(wpp, cb, [&](uint32_t w) -> const XYZZ& { return window_sum(b, w); });
    };
    HostScope hs_horner(c, "host_msm_horner");
    if (batch > 1 && wpp > 1) {
        // 254 dependent doublings per vector: ~0.04 ms each on one core, so spread the vectors over the pool
        const std::function<void(uint32_t)> job = [&](uint32_t b) { horner(b); };
        horner_pool().run(batch, job);
    } else {
        for (uint32_t b = 0; b < batch; ++b) horner(b);
    }
}

// ---- the small-problem pipeline (kernels above: msm_small_*) ------------------------------------------
static int small_window_bits(Ctx& c, size_t n, uint32_t batch) {
    if (c.msm_window_bits >= 4 && c.msm_window_bits <= 10) return c.msm_window_bits;
    (void)batch;
    if (n <= 256) return 5;
    if (n <= 2048) return 7;
    return 8;
}
bool msm_small_applies(Ctx& c, size_t n, uint32_t batch, int pre_c = 0) {
    if (!c.tune_small || n == 0 || n > (1u << 15)) return false;
    if (pre_c > 10 || (pre_c == 0 && c.msm_window_bits > 10)) return false;   // larger windows: the general pipeline
    const int cb = pre_c > 0 ? pre_c : small_window_bits(c, n, batch);
    const uint64_t S = (uint64_t)batch * (uint64_t)msm_num_windows(cb);
    return S <= 65535 && S * n < (1ull << 31);
}

// General mode: `points` are the bases, the host combines the W window sums by Horner.  Window-table mode (pre_c > 0):
// `points` is T[w][i] = 2^(pre_c w) P_i (row stride pre_stride, first column pre_off), every window keeps its own bucket
// set but reads its own row, and the window sums just add up -- no doublings on the host.
static int msm_run_small(Ctx& c, const Affine* points, const ScalarView& d_scalars, size_t n, uint32_t batch, Jac* out_host, int pre_c,
                         uint32_t pre_stride, uint32_t pre_off) {
    MsmWork& m = c.msm[0];
    hipStream_t st = c.stream;
    c.cur_stream = st;
    const int cb = pre_c > 0 ? pre_c : small_window_bits(c, n, batch);
    const uint32_t W = (uint32_t)msm_num_windows(cb), S = batch * W, NBL = 1u << (cb - 1), n32 = (uint32_t)n;
    const uint64_t entries = (uint64_t)S * n;
    // Task length: the accumulation is issue-bound from one wave per SIMD on (a wave alone already keeps its
    // SIMD's 64-bit multiplier busy), so the shortest dependent chain comes from filling k = 1..4 waves per
    // SIMD almost exactly: L = entries / (0.93 * 65536 * k) for the smallest k that keeps L <= 12.
    uint32_t L = 0;
    {
        const double lanes = 0.93 * 64.0 * 4.0 * (double)c.num_cus;
        for (int k = 1; k <= 4; ++k) {
            L = (uint32_t)((double)entries / (lanes * k)) + 1;
            if (L <= 12) break;
        }
        L = std::max<uint32_t>(L, 2);
    }
    const uint64_t task_cap = entries / L + (uint64_t)S * NBL;       // sum_b ceil(cnt_b / L) over non-empty buckets
    // fold levels (16 to 1): as many as the fullest possible bucket (all n entries of a slot) needs
    SmallLevels lv{};
    uint64_t cap[kSmallMaxLevels + 1];                               // capacity of each level's region (sums = chunks)
    cap[0] = task_cap;
    {
        uint64_t worst = (n + L - 1) / L;
        while (worst > 1 && lv.nl < (uint32_t)kSmallMaxLevels) { worst = (worst + kSmallFan - 1) / kSmallFan; ++lv.nl; }
        if (worst > 1) { set_error("msm: small pipeline: %u fold levels do not cover n = %zu at task length %u", lv.nl, n, L); return UZK_ERR_PARAMETER; }
    }
    uint64_t p_total = 0, d_total = 0;
    for (uint32_t k = 0; k <= lv.nl; ++k) {
        if (k > 0) cap[k] = cap[k - 1] / kSmallFan + (uint64_t)S * NBL;   // a chunk per 16 sums, plus one ragged chunk per bucket
        lv.pbase[k] = (uint32_t)p_total;
        lv.dbase[k] = (uint32_t)d_total;                                  // level k's descriptors follow level k - 1's
        p_total += cap[k];
        if (k > 0) d_total += cap[k];
    }
    if (p_total >= (1ull << 32)) { set_error("msm: small pipeline: too many partial sums"); return UZK_ERR_PARAMETER; }
    UZK_TRY(m.digits.reserve((size_t)entries * 4));
    UZK_TRY(m.sorted.reserve((size_t)entries * 4));
    UZK_TRY(m.task_desc.reserve((size_t)task_cap * sizeof(TaskDesc)));
    UZK_TRY(m.lvl_part[0].reserve((size_t)p_total * sizeof(XYZZ)));
    UZK_TRY(m.exc.reserve((size_t)task_cap * 4));
    UZK_TRY(m.partials.reserve((size_t)(d_total + 1) * sizeof(SmallChunk)));
    UZK_TRY(m.bucket_count.reserve((size_t)S * NBL * 4));
    UZK_TRY(m.bucket_start.reserve((size_t)S * kSmallMaxLevels * sizeof(uint2)));
    UZK_TRY(m.win_sums.reserve((size_t)S * sizeof(XYZZ)));
    UZK_TRY(m.small.reserve(16384));
    if (m.h_sums_cap < S) {
        if (m.h_sums) (void)hipHostFree(m.h_sums);
        UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&m.h_sums), (size_t)S * sizeof(XYZZ), hipHostMallocDefault));
        m.h_sums_cap = S;
    }
    uint32_t* digits = m.digits.as<uint32_t>();
    uint32_t* sorted = m.sorted.as<uint32_t>();
    TaskDesc* desc = m.task_desc.as<TaskDesc>();
    XYZZ* P = m.lvl_part[0].as<XYZZ>();
    SmallChunk* cdesc = m.partials.as<SmallChunk>();
    uint32_t* bucket_ref = m.bucket_count.as<uint32_t>();
    uint2* slot_chunks = m.bucket_start.as<uint2>();
    XYZZ* win_sums = m.h_sums;        // pinned, device-visible host memory: one 128-byte store per slot, no copy kernel afterwards
    uint32_t* counters = m.small.as<uint32_t>() + 3960;               // [0] tasks, [1..4] chunks per level, [6] exceptions
    const size_t sort_lds = ((size_t)5 * NBL + 32 + n32) * 4;
    {
        static std::once_flag attr_once;
        hipError_t attr_err = hipSuccess;
        std::call_once(attr_once, [&] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(msm_small_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (5 * 512 + 32 + 32768) * 4);
        });
        UZK_HIP(attr_err);
    }
    {
        HostScope hs(c, "host_msm_enqueue1");
        {
            KernelScope ks(c, "msm_digits");
            const uint64_t tot = (uint64_t)n32 * batch;                  // n >= 1: the grid has the lanes that clear the eight counters
            hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_scalars, digits, n32, batch,
                               cb, (int)W, 0, (int)W, counters);
        }
        {
            KernelScope ks(c, "msm_small_sort");
            hipLaunchKernelGGL(msm_small_sort_kernel, dim3(S), dim3(1024), sort_lds, st, digits, n32, NBL, L, sorted, desc, cdesc,
                               bucket_ref, slot_chunks, counters, lv, W, pre_c > 0 ? pre_stride : 0u, pre_off);
        }
        {
            KernelScope ks(c, "msm_accumulate");
            const dim3 grid((unsigned)((task_cap + 255) / 256));
            hipLaunchKernelGGL(msm_accumulate29_kernel, grid, dim3(256), 0, st, points, sorted, desc, counters, P, 0u,
                               counters + 6, m.exc.as<uint32_t>());
            hipLaunchKernelGGL(msm_accumulate_exc_kernel, dim3(256), dim3(256), 0, st, points, sorted, desc, P, counters + 6,
                               m.exc.as<uint32_t>());
        }
        // Fold mode per level: quads (four lanes per addition) while a level is latency-bound -- about one chunk per
        // non-empty bucket at level 1, far fewer later -- plain lanes once the chunks alone fill the chip.
        const uint64_t lanes_chip = (uint64_t)c.num_cus * 4 * 64;
        for (uint32_t k = 1; k <= std::min<uint32_t>(lv.nl, 1); ++k) {     // later levels: inside the reduction kernel
            KernelScope ks(c, "msm_small_fold");
            const uint64_t est = k == 1 ? (uint64_t)S * NBL : (uint64_t)S * 2;      // chunks that really exist (estimate)
            // measured (profiles/r02_small_msm_2e14.txt, n = 2^14): quads of 4 logical lanes match or beat plain lanes up
            // to batch 8 (uniform: 75-89 vs 77-96 us, skewed scalars: 27-38 vs 62-63 us); beyond, plain lanes fill the chip
            // (round 6, four lockstep provers keeping the chip busy: plain lanes here -- 1 / 1.6 of the quads' instructions -- made no
            // measurable difference, 1602 / 1583 against 1620 / 1576 proofs/s lockstep8 / shared_32, profiles/r06_ab_arith29.txt)
            const bool quad = est * 16 <= 10 * lanes_chip;
            const int gs = quad ? 4 : (est * 4 <= 4 * lanes_chip ? 4 : 2);
            const uint64_t lanes = cap[k] * (uint64_t)gs * (quad ? 4 : 1);
            const dim3 grid((unsigned)((lanes + 255) / 256));
            const SmallChunk* cd = cdesc + lv.dbase[k];
#define UZK_FOLD(QD, G) hipLaunchKernelGGL((msm_small_fold_kernel<QD, G>), grid, dim3(256), 0, st, P, cd, counters + k, lv.pbase[k], (c.tune_arith29 >> 2) & 1)
            if (quad) { if (gs >= 8) UZK_FOLD(true, 8); else if (gs >= 4) UZK_FOLD(true, 4); else UZK_FOLD(true, 2); }
            else { if (gs >= 8) UZK_FOLD(false, 8); else if (gs >= 4) UZK_FOLD(false, 4); else if (gs >= 2) UZK_FOLD(false, 2); else UZK_FOLD(false, 1); }
#undef UZK_FOLD
        }
        {
            KernelScope ks(c, "msm_small_reduce");
            const uint32_t Q = std::min<uint32_t>(NBL, 64);
            if ((c.tune_arith29 >> 2) & 1) hipLaunchKernelGGL((msm_small_reduce_kernel<3, 256>), dim3(S), dim3(4 * Q), (size_t)Q * 144, st, P, bucket_ref, cdesc,
                               slot_chunks, lv, win_sums, NBL);
            else hipLaunchKernelGGL((msm_small_reduce_kernel<2, 256>), dim3(S), dim3(4 * Q), (size_t)Q * 144, st, P, bucket_ref, cdesc,
                               slot_chunks, lv, win_sums, NBL);
        }
        UZK_HIP(hipGetLastError());
    }
    {
        HostScope hs(c, "host_msm_wait2");
        UZK_HIP(hipStreamSynchronize(st));      // the reduction kernel stored the window sums straight into pinned host memory
    }
    auto window_sum = [&](uint32_t b, uint32_t w) -> const XYZZ& { return m.h_sums[(size_t)b * W + w]; };
    msm_horner_host(c, batch, W, pre_c > 0 ? 0 : cb, window_sum, out_host);
    return UZK_OK;
}

// ---- streamed: host scalars of a large MSM arrive in point chunks while the previous chunk is being accumulated --------
// Only msm_digits reads the scalars, but the sort needs every digit of its input, so a monolithic MSM cannot start before
// the whole upload is in (2^24 scalars = 512 MiB = 10 ms of PCIe in front of 20 ms of compute).  Here the points are cut into
// chunks that run the whole pipeline up to the bucket array one after the other -- same window width as the full problem,
// ONE bucket set: every chunk's bucket sums are added onto the previous ones (msm_finalize, accumulate), the reduction runs
// once at the end -- so the total number of mixed additions is that of the monolithic MSM, and chunk k + 1 uploads under
// chunk k's accumulation.  uzk_msm_g1 (host scalars, general mode) takes this path from 2^22 points on.
int msm_run_streamed(Ctx& c, const Affine* points, const Fp* scalars_host, size_t n, Jac* out_host) {
    if (!c.msm) c.msm = new MsmWork[2];
    const int cb = choose_window_bits(n, c.msm_window_bits);
    const uint32_t W = (uint32_t)msm_num_windows(cb);
    // Chunk schedule: the first upload is the only one nothing hides, so it is small (2^20 points = 32 MiB = 0.6 ms); the
    // chunks then double up to 2^22, where the per-chunk costs (task tables, the pass over the bucket array) are a few
    // per cent of the accumulation.  Measured at 2^24 (tools/stream_msm.py): uniform 2^20 / 2^21 / 2^22 chunks 24.8 / 23.5 /
    // 25.1 ms, upload-then-compute 31.1 ms, device-resident scalars 20.6 ms.  uzk_tune("msm_stream_log", k): uniform 2^k.
    std::vector<size_t> bounds{0};
    {
        const size_t cap = (size_t)1 << (c.tune_stream_log > 0 ? std::max(16, std::min(26, c.tune_stream_log)) : 22);
        size_t sz = c.tune_stream_log > 0 ? cap : std::min<size_t>(cap, (size_t)1 << 20);
        int at_this_size = 0;
        while (bounds.back() < n) {
            bounds.push_back(std::min(n, bounds.back() + sz));
            if (++at_this_size >= (sz == ((size_t)1 << 20) ? 2 : 1) && sz < cap) { sz <<= 1; at_this_size = 0; }
        }
    }
    const size_t nchunks = bounds.size() - 1;
    size_t chunk = 0;
    for (size_t k = 0; k < nchunks; ++k) chunk = std::max(chunk, bounds[k + 1] - bounds[k]);
    if (!c.stream2) UZK_HIP(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
    hipStream_t copy_st = c.stream2;
    UZK_TRY(c.msm_scalars.reserve(2 * chunk * sizeof(Fp)));            // double buffer
    Fp* stage[2] = {c.msm_scalars.as<Fp>(), c.msm_scalars.as<Fp>() + chunk};
    hipEvent_t up_done[2] = {c.get_event(), c.get_event()}, consumed[2] = {c.get_event(), c.get_event()};
    auto give_back = [&] { for (int i = 0; i < 2; ++i) { c.event_pool.push_back(up_done[i]); c.event_pool.push_back(consumed[i]); } };
    MsmGroup g;
    g.m = &c.msm[0];
    g.st = c.stream;
    int rc = UZK_OK;
    auto upload = [&](size_t k) -> int {
        const size_t lo = bounds[k], len = bounds[k + 1] - lo;
        const int s = (int)(k & 1);
        if (k >= 2) UZK_HIP(hipStreamWaitEvent(copy_st, consumed[s], 0));        // the digits kernel of chunk k - 2 has read this half
        HostScope hs(c, "host_msm_upload");
        UZK_HIP(hipMemcpyAsync(stage[s], scalars_host + lo, len * sizeof(Fp), hipMemcpyHostToDevice, copy_st));
        UZK_HIP(hipEventRecord(up_done[s], copy_st));
        return UZK_OK;
    };
    rc = msm_group_plan(c, g, chunk, 1, cb, false, W, 0, W);        // size every workspace for the largest chunk before anything is queued
    if (rc == UZK_OK) rc = upload(0);
    for (size_t k = 0; k < nchunks && rc == UZK_OK; ++k) {
        const size_t lo = bounds[k], len = bounds[k + 1] - lo;
        const int s = (int)(k & 1);
        rc = msm_group_plan(c, g, len, 1, cb, false, W, 0, W);
        if (rc != UZK_OK) break;
        if (hipStreamWaitEvent(c.stream, up_done[s], 0) != hipSuccess) { rc = UZK_ERR_DEVICE; set_error("msm: stream wait failed"); break; }
        { HostScope hs(c, "host_msm_enqueue1"); rc = msm_group_phase1(c, g, points + lo, ScalarView::dense(stage[s], len), 0, 0, /*direct_ok*/ k == 0); }
        if (rc != UZK_OK) break;
        // phase 1 is queued: its first kernel (digits) is the only reader of the staging half.  Marking the half free after the
        // whole phase is simpler than an event in the middle of it and costs nothing: the next upload into this half is two
        // chunks away.
        (void)hipEventRecord(consumed[s], c.stream);
        if (k + 1 < nchunks) rc = upload(k + 1);           // pageable source: the call returns when the chunk is on its way; the GPU works meanwhile
        if (rc != UZK_OK) break;
        { HostScope hs(c, "host_msm_wait1_enqueue2"); rc = msm_group_phase2(c, g, /*accumulate*/ k > 0, /*reduce*/ k + 1 == nchunks); }
    }
    { HostScope hs(c, "host_msm_wait2"); (void)hipStreamSynchronize(c.stream); (void)hipStreamSynchronize(copy_st); }
    c.cur_stream = c.stream;
    give_back();
    UZK_TRY(rc);
    const uint32_t per = g.class_s ? 2u : 1u;
    auto window_sum = [&](uint32_t, uint32_t w) -> const XYZZ& { return g.m->h_sums[(size_t)w * per]; };
    msm_horner_host(c, 1, W, cb, window_sum, out_host, (int)g.class_s);
    return UZK_OK;
}

// Device-resident scalars beyond 2^24 points (general mode, one vector): the same chunk loop without the uploads.  The packed
// two-pass sort -- and with it the register-resident sort kernels -- holds point indices of 24 bits, so a 2^25 or 2^26-point
// MSM runs as 2 / 4 chunks of 2^24 points into ONE bucket set (full problem's window width, one reduction at the end) instead
// of one pass over 8-byte entries through the generic sort: 41.96 -> 39.5 ms at 2^25, 83.5 -> 78.7 ms at 2^26.
int msm_run_chunked(Ctx& c, const Affine* points, const Fp* d_scalars, size_t n, Jac* out_host) {
    if (!c.msm) c.msm = new MsmWork[2];
    const int cb = choose_window_bits(n, c.msm_window_bits);
    const uint32_t W = (uint32_t)msm_num_windows(cb);
    const size_t chunk = (size_t)1 << 24;
    MsmGroup g;
    g.m = &c.msm[0];
    g.st = c.stream;
    UZK_TRY(msm_group_plan(c, g, std::min(chunk, n), 1, cb, false, W, 0, W));
    int rc = UZK_OK;
    for (size_t lo = 0, k = 0; lo < n && rc == UZK_OK; lo += chunk, ++k) {
        const size_t len = std::min(chunk, n - lo);
        rc = msm_group_plan(c, g, len, 1, cb, false, W, 0, W);
        if (rc != UZK_OK) break;
        { HostScope hs(c, "host_msm_enqueue1"); rc = msm_group_phase1(c, g, points + lo, ScalarView::dense(d_scalars + lo, len), 0, 0, /*direct_ok*/ k == 0); }
        if (rc != UZK_OK) break;
        { HostScope hs(c, "host_msm_wait1_enqueue2"); rc = msm_group_phase2(c, g, /*accumulate*/ k > 0, /*reduce*/ lo + len >= n); }
    }
    { HostScope hs(c, "host_msm_wait2"); (void)hipStreamSynchronize(c.stream); }
    c.cur_stream = c.stream;
    UZK_TRY(rc);
    const uint32_t per = g.class_s ? 2u : 1u;
    auto window_sum = [&](uint32_t, uint32_t w) -> const XYZZ& { return g.m->h_sums[(size_t)w * per]; };
    msm_horner_host(c, 1, W, cb, window_sum, out_host, (int)g.class_s);
    return UZK_OK;
}

void msm_plan_info(Ctx& c, size_t n, int* window_bits, int* windows) {
    const int cb = msm_small_applies(c, n, 1) ? small_window_bits(c, n, 1) : choose_window_bits(n, c.msm_window_bits);
    *window_bits = cb;
    *windows = msm_num_windows(cb);
}

// `points`: base array the sorted indices refer to (the SRS slice, or the window table).
// Precomputed mode (pre_c > 0): `points` = table, entries of window j index pre_stride * j + pre_off + i.
// `batch` scalar vectors of n elements each share the same bases; out_host[batch].
int msm_run(Ctx& c, const Affine* points, const ScalarView& d_scalars, size_t n, uint32_t batch, Jac* out_host, int pre_c,
            uint32_t pre_stride, uint32_t pre_off) {
    if (batch == 0) return UZK_OK;
    if ((size_t)d_scalars.n_main + d_scalars.tail_n != n) { set_error("msm: scalar view covers %u + %u elements, n = %zu", d_scalars.n_main, d_scalars.tail_n, n); return UZK_ERR_PARAMETER; }
    if (n == 0) { for (uint32_t b = 0; b < batch; ++b) out_host[b] = jac_inf(); return UZK_OK; }
    if (n >= (1ull << 31)) { set_error("msm: n = %zu exceeds 2^31 - 1 points per call", n); return UZK_ERR_PARAMETER; }
    if (!c.msm) c.msm = new MsmWork[2];
    const bool pre = pre_c > 0;
    if (msm_small_applies(c, n, batch, pre_c)) return msm_run_small(c, points, d_scalars, n, batch, out_host, pre_c, pre_stride, pre_off);
    const int cb = pre ? pre_c : choose_window_bits(n, c.msm_window_bits);
    const uint32_t W = (uint32_t)msm_num_windows(cb);
    MsmGroup g[1];
    const int ngroups = 1;
    g[0].m = &c.msm[0]; g[0].st = c.stream;
    UZK_TRY(msm_group_plan(c, g[0], n, batch, cb, pre, W, 0, W, pre ? (uint64_t)pre_stride * (W - 1) + pre_off + n : 0));
    int rc = UZK_OK;
    { HostScope hs(c, "host_msm_enqueue1");
      for (int k = 0; k < ngroups && rc == UZK_OK; ++k) rc = msm_group_phase1(c, g[k], points, d_scalars, pre_stride, pre_off); }
    { HostScope hs(c, "host_msm_wait1_enqueue2");
      for (int k = 0; k < ngroups && rc == UZK_OK; ++k) rc = msm_group_phase2(c, g[k]); }
    { HostScope hs(c, "host_msm_wait2");
      for (int k = 0; k < ngroups; ++k) (void)hipStreamSynchronize(g[k].st); }
    c.cur_stream = c.stream;
    UZK_TRY(rc);

    // 7. host: per scalar vector, Horner over its logical windows (c doublings per step); one window
    //    each when precomputed.  Window w of vector b lives in the group that owns w.
    const uint32_t wpp = pre ? 1u : W;
    const uint32_t per = g[0].class_s ? 2u : 1u;
    auto window_sum = [&](uint32_t b, uint32_t w) -> const XYZZ& {
        if (pre) return g[0].m->h_sums[(size_t)b * per];
        return g[0].m->h_sums[((size_t)b * g[0].W + (w - g[0].w0)) * per];
    };
    msm_horner_host(c, batch, wpp, cb, window_sum, out_host, (int)g[0].class_s);
    return UZK_OK;
}

}  // namespace uzk
