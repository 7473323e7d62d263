// The per-proof kernels of the prover's five rounds with the LANES of a lockstep batch on a grid axis: one launch per step for
// all the proofs that advance together (prover.cpp), where poly.hip's single-polynomial entry points take one launch -- and some
// a synchronisation -- per proof.  Same arithmetic as their one-lane counterparts in poly.hip (exact field elements: any
// evaluation order gives the reference's bytes); what differs is addressing: lane b's data lies a fixed stride behind lane 0's,
// and everything that varies per lane and per round (challenges, blinds, lengths, pointer lists) is read from a device-resident
// argument block (ArgArena, ctx.hpp) that the host fills once per round.
//
//   hide_lanes            hide_polynomial, helpers.rs:139-158                      round 1 (wires, selectors, pi), round 2 (z)
//   poly_eval_lanes       FpPolynomial::eval at zeta / zeta omega, prover.rs:246-273   round 4
//   poly_lincomb_lanes    r(X), helpers.rs:1030-1080; h = sum alpha^k p_k, pcs.rs:119-131   round 5
//   open_div_lanes        h / (X - z), field_polynomial.rs:519-550                  round 5
//   split_t_lanes         split_t_and_commit's chunks, helpers.rs:1335-1363         round 3
//   fold_blinds_lanes     fold modulo X^N - 1 + the blind scalars, pcs.rs:137-156   rounds 3, 5
//   trimmed_len_lanes     FpPolynomial::from_coefs' trim, field_polynomial.rs:86-90 rounds 3, 5
#include <algorithm>
#include <cstring>

#include "ctx.hpp"
#include "host_math.hpp"
#include "lz29.hpp"

namespace uzk {

int ArgArena::init(size_t bytes) {
    release();
    UZK_HIP(hipHostMalloc(reinterpret_cast<void**>(&h), bytes, hipHostMallocDefault));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        h = nullptr; d = nullptr;
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return UZK_ERR_DEVICE;
    }
    cap = bytes;
    reset();
    return UZK_OK;
}
void ArgArena::release() {
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    h = nullptr; d = nullptr; cap = 0; used = 0; uploaded = 0;
}
int ArgArena::upload(hipStream_t s) {
    if (used > uploaded) {
        const size_t from = uploaded & ~(size_t)15;
        UZK_HIP(hipMemcpyAsync(d + from, h + from, used - from, hipMemcpyHostToDevice, s));
        uploaded = used;
    }
    return UZK_OK;
}

// ---- hide_polynomial ------------------------------------------------------------------------------------------------------
// Polynomial (slot, lane) holds n coefficients at coefs + lane * lane_stride + slot * slot_stride; its three blind slots
// [n, n + 3) are WRITTEN (the reference resizes with zeros first), so whatever an earlier proof left there does not matter --
// unused blinds are passed as zeros.  blinds: [lanes][slots][3].
__global__ __launch_bounds__(64) void hide_lanes_kernel(Fp* __restrict__ coefs, uint64_t lane_stride, uint64_t slot_stride, uint32_t n,
                                                        const Fp* __restrict__ blinds) {
    const uint32_t i = threadIdx.x;
    if (i >= 3) return;
    Fp* c = coefs + (uint64_t)blockIdx.y * lane_stride + (uint64_t)blockIdx.x * slot_stride;
    const Fp bl = blinds[((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + i];
    c[i] = Fr::add(c[i], bl);
    c[n + i] = Fr::neg(bl);
}
int hide_lanes(Ctx& c, Fp* d_coefs, uint64_t lane_stride, uint64_t slot_stride, uint32_t n, uint32_t slots, uint32_t lanes, const Fp* d_blinds) {
    if (slots == 0 || lanes == 0) return UZK_OK;
    KernelScope ks(c, "poly_hide");
    hipLaunchKernelGGL(hide_lanes_kernel, dim3(slots, lanes), dim3(64), 0, c.stream, d_coefs, lane_stride, slot_stride, n, d_blinds);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- evaluations ----------------------------------------------------------------------------------------------------------
// Polynomial k of lane b: len[k] coefficients at p[k] + b * lane_stride[k] (0 for a circuit polynomial every lane shares),
// evaluated at points[2 b + pt[k]].  grid (blocks, count, lanes); the value goes straight to pinned host memory,
// out_host[b * count + k].  Same scheme as poly_eval_small_kernel (poly.hip): 256 * PER coefficients per workgroup, a power table
// in LDS, the last workgroup of a polynomial adds the block sums.  The table and the block's weight cost every lane ~25 products
// whatever PER is: at PER = 4 (one proof: many small workgroups, short chains) that is six times the Horner steps themselves, at
// PER = 16 (several lanes fill the chip anyway) one and a half times -- 158 -> 60 us for 8 lanes x 19 evaluations at n = 2^14.
struct EvalPoly { const Fp* p; uint64_t lane_stride; uint32_t len, pt; };
template <int PER>
__global__ __launch_bounds__(256) void poly_eval_lanes_kernel(const EvalPoly* __restrict__ polys, const Fp* __restrict__ points, Fp* __restrict__ partial,
                                                              uint32_t* __restrict__ counters, Fp* __restrict__ out_host) {
    constexpr int BLOCK = 256 * PER;
    __shared__ Fp pw[256];
    __shared__ Fp sh[256];
    __shared__ uint32_t last;
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, stride = gridDim.x, k = blockIdx.y, b = blockIdx.z;
    const uint32_t slot = b * gridDim.y + k;
    const EvalPoly P = polys[k];
    const uint64_t n = P.len;
    const uint32_t nblocks = (uint32_t)((n + BLOCK - 1) / BLOCK);
    if (n == 0 && blk == 0 && tid == 0) out_host[slot] = Fr::zero();
    if (blk >= nblocks) return;
    const Fp* c = P.p + (uint64_t)b * P.lane_stride;
    const Fp x = points[2 * b + P.pt];
    const uint64_t base = (uint64_t)blk * BLOCK + (uint64_t)tid * PER;
    Fp h = Fr::zero();
#pragma unroll
    for (int e = PER - 1; e >= 0; --e) {
        const uint64_t j = base + e;
        h = Fr::mul(h, x);
        if (j < n) h = Fr::add(h, c[j]);
    }
    Fp s = x;                                  // x^PER
#pragma unroll
    for (int e = 1; e < PER; e <<= 1) s = Fr::sqr(s);
    if (tid == 0) pw[0] = Fr::one();
    __syncthreads();
#pragma unroll
    for (int lv = 0; lv < 8; ++lv) {
        const uint32_t half = 1u << lv;
        if (tid < half) pw[half + tid] = Fr::mul(pw[tid], s);
        s = Fr::sqr(s);
        __syncthreads();
    }
    Fp wblk = Fr::one(), sp = s;              // s = x^BLOCK: the block's weight is s^blk
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
        partial[(uint64_t)slot * stride + blk] = Fr::mul(sh[0], wblk);
        __threadfence();
        last = (atomicAdd(&counters[slot], 1u) == nblocks - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        Fp acc = Fr::zero();
        for (uint32_t j = 0; j < nblocks; ++j) acc = Fr::add(acc, partial[(uint64_t)slot * stride + j]);
        out_host[slot] = acc;
        counters[slot] = 0;                   // ready for the next launch on this stream
    }
}
// Evaluations: poly_eval_lanes_kernel's scheme -- PER Horner steps per lane, the power table (x^PER)^t in LDS by doubling, the block's
// weight, a tree over the 256 lane values -- with every product on the 29-bit limbs; the LDS arrays hold limb planes.  The block's
// partial value leaves in wire form (the last block's sum over <= 64 partials stays on 8 x 32-bit words: one lane).
template <int PER>
__global__ __launch_bounds__(256) void poly_eval_lanes29_kernel(const EvalPoly* __restrict__ polys, const Fp* __restrict__ points, Fp* __restrict__ partial,
                                                                uint32_t* __restrict__ counters, Fp* __restrict__ out_host) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Z = LzOps<Fr29>;
    using S8 = Lz<Fr29, 1, 8>;
    constexpr int BLOCK = 256 * PER;
    __shared__ uint32_t pw[9][256];
    __shared__ uint32_t sh[9][256];
    __shared__ uint32_t last;
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, stride = gridDim.x, k = blockIdx.y, b = blockIdx.z;
    const uint32_t slot = b * gridDim.y + k;
    const EvalPoly P = polys[k];
    const uint64_t n = P.len;
    const uint32_t nblocks = (uint32_t)((n + BLOCK - 1) / BLOCK);
    if (n == 0 && blk == 0 && tid == 0) out_host[slot] = Fr::zero();
    if (blk >= nblocks) return;
    const Fp* c = P.p + (uint64_t)b * P.lane_stride;
    const auto x = Z::ld(points[2 * b + P.pt]);
    const uint64_t base = (uint64_t)blk * BLOCK + (uint64_t)tid * PER;
    auto put = [&](uint32_t (*arr)[256], uint32_t i, const L29& v) {
#pragma unroll
        for (int w = 0; w < 9; ++w) arr[w][i] = v.l[w];
    };
    auto get = [&](uint32_t (*arr)[256], uint32_t i) { L29 v;
#pragma unroll
        for (int w = 0; w < 9; ++w) v.l[w] = arr[w][i];
        return v; };
    Lz<Fr29, 2, 48> h = Z::template relax<2, 48>(Z::zero());
#pragma unroll
    for (int e = PER - 1; e >= 0; --e) {
        const uint64_t j = base + e;
        const auto hx = Z::mul(h, x);                                   // < 11 M
        if (j < n) h = Z::template relax<2, 48>(Z::add(hx, Z::ld(c[j])));
        else h = Z::template relax<2, 48>(hx);
    }
    S8 s = Z::template relax<1, 8>(Z::sqr(x));                          // x^2 ... x^PER
#pragma unroll
    for (int e = 2; e < PER; e <<= 1) s = Z::template relax<1, 8>(Z::sqr(s));
    if (tid == 0) put(pw, 0, Z::one().v);
    __syncthreads();
#pragma unroll
    for (int lv = 0; lv < 8; ++lv) {
        const uint32_t half = 1u << lv;
        if (tid < half) { S8 t; t.v = get(pw, tid); put(pw, half + tid, Z::mul(t, s).v); }
        s = Z::template relax<1, 8>(Z::sqr(s));
        __syncthreads();
    }
    S8 wblk = Z::template relax<1, 8>(Z::one()), sp = s;                // s = x^BLOCK: the block's weight is s^blk
    for (uint32_t e = blk; e; e >>= 1) {
        if (e & 1) wblk = Z::template relax<1, 8>(Z::mul(wblk, sp));
        sp = Z::template relax<1, 8>(Z::sqr(sp));
    }
    { S8 t; t.v = get(pw, tid); put(sh, tid, Z::mul(h, t).v); }         // < 4 M each, normalized
    __syncthreads();
    // 256 values < 4 M: the tree's sums stay below 1024 M; one carry step per level keeps the limbs normalized
    for (uint32_t st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            Lz<Fr29, 1, 1024> u, v;
            u.v = get(sh, tid); v.v = get(sh, tid + st);
            put(sh, tid, Z::norm(Z::add(u, v)).v);
        }
        __syncthreads();
    }
    if (tid == 0) {
        Lz<Fr29, 1, 1024> tot;
        tot.v = get(sh, 0);
        partial[(uint64_t)slot * stride + blk] = Z::to_wire(Z::mul(tot, wblk));
        __threadfence();
        last = (atomicAdd(&counters[slot], 1u) == nblocks - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        Fp acc = Fr::zero();
        for (uint32_t j = 0; j < nblocks; ++j) acc = Fr::add(acc, partial[(uint64_t)slot * stride + j]);
        out_host[slot] = acc;
        counters[slot] = 0;                   // ready for the next launch on this stream
    }
#endif
}

// d_polys: `count` EvalPoly in device memory; d_counters: lanes * count zeroed words; out_host: pinned.  Asynchronous.
int poly_eval_lanes(Ctx& c, const void* d_polys, uint32_t count, uint64_t max_len, const Fp* d_points, uint32_t lanes, uint32_t* d_counters, Fp* out_host_pinned) {
    if (count == 0 || lanes == 0) return UZK_OK;
    if (max_len == 0 || max_len > (1ull << 18)) { set_error("poly_eval_lanes: lengths must be 1 .. 2^18"); return UZK_ERR_PARAMETER; }
    const bool wide = lanes >= 4 && max_len >= 4096;
    const uint64_t per_block = wide ? 4096 : 1024, blocks = (max_len + per_block - 1) / per_block;
    UZK_TRY(c.poly_tmp.reserve((size_t)lanes * count * blocks * sizeof(Fp)));
    KernelScope ks(c, "poly_eval");
    const dim3 grid((unsigned)blocks, count, lanes);
    const bool a29 = (c.tune_arith29 & 2) != 0;       // the products on the lazy 29-bit limbs (uzk_tune("arith29", mask) bit 1)
    if (wide && a29) hipLaunchKernelGGL(poly_eval_lanes29_kernel<16>, grid, dim3(256), 0, c.stream, static_cast<const EvalPoly*>(d_polys), d_points, c.poly_tmp.as<Fp>(), d_counters, out_host_pinned);
    else if (a29) hipLaunchKernelGGL(poly_eval_lanes29_kernel<4>, grid, dim3(256), 0, c.stream, static_cast<const EvalPoly*>(d_polys), d_points, c.poly_tmp.as<Fp>(), d_counters, out_host_pinned);
    else if (wide) hipLaunchKernelGGL(poly_eval_lanes_kernel<16>, grid, dim3(256), 0, c.stream, static_cast<const EvalPoly*>(d_polys), d_points, c.poly_tmp.as<Fp>(), d_counters, out_host_pinned);
    else hipLaunchKernelGGL(poly_eval_lanes_kernel<4>, grid, dim3(256), 0, c.stream, static_cast<const EvalPoly*>(d_polys), d_points, c.poly_tmp.as<Fp>(), d_counters, out_host_pinned);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}
void eval_poly_fill(void* host_entry, const void* p, uint64_t lane_stride, uint64_t len, uint32_t pt) {
    EvalPoly& e = *static_cast<EvalPoly*>(host_entry);
    e.p = static_cast<const Fp*>(p); e.lane_stride = lane_stride; e.len = (uint32_t)len; e.pt = pt;
}
size_t eval_poly_bytes() { return sizeof(EvalPoly); }

// ---- linear combinations --------------------------------------------------------------------------------------------------
// out_b[j] = sum_k scalars[b][k] * p_k,b[j], j < out_len; p_k,b = p[k] + b * lane_stride[k] holds lens[b][k] coefficients.
// GS lanes of the wave share a coefficient (poly.hip poly_lincomb_kernel: at n + 3 = 16 387 coefficients one lane per
// coefficient is a chain of `count` dependent products on an idle chip).
struct LinPoly { const Fp* p; uint64_t lane_stride; };
template <int GS>
__global__ __launch_bounds__(256) void poly_lincomb_lanes_kernel(const LinPoly* __restrict__ polys, uint32_t count, const uint32_t* __restrict__ lens,
                                                                 const Fp* __restrict__ scalars, Fp* __restrict__ out, uint64_t out_stride, uint64_t out_len) {
    const uint32_t b = blockIdx.y;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = gid / GS;
    const uint32_t sub = (uint32_t)(gid % GS);
    const bool live = j < out_len;
    lens += (uint64_t)b * count; scalars += (uint64_t)b * count;
    Fp acc = Fr::zero();
    if (live)
        for (uint32_t k = sub; k < count; k += GS)
            if (j < lens[k]) acc = Fr::add(acc, Fr::mul(scalars[k], polys[k].p[(uint64_t)b * polys[k].lane_stride + j]));
    if constexpr (GS > 1) {
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) {
            Fp q;
#pragma unroll
            for (int w = 0; w < 8; ++w) q.v[w] = (uint32_t)__shfl_down((int)acc.v[w], o);
            if (sub + (uint32_t)o < (uint32_t)GS) acc = Fr::add(acc, q);
        }
    }
    if (live && sub == 0) out[(uint64_t)b * out_stride + j] = acc;
}
// ---- the same two lane kernels on the lazy 29-bit limbs (lz29.hpp; round 6) ------------------------------------------------------
// Linear combination: two terms share one reduction (mul2); the running sum stays lazy between terms (one parallel carry step per
// pair).  Value bound: every pair adds < 14 M, the host admits count <= 128 (< 900 M in all: the top limb stays below 2^32).
template <int GS>
__global__ __launch_bounds__(256) void poly_lincomb_lanes29_kernel(const LinPoly* __restrict__ polys, uint32_t count, const uint32_t* __restrict__ lens,
                                                                   const Fp* __restrict__ scalars, Fp* __restrict__ out, uint64_t out_stride, uint64_t out_len) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Z = LzOps<Fr29>;
    using Acc = Lz<Fr29, 1, 1024>;
    const uint32_t b = blockIdx.y;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = gid / GS;
    const uint32_t sub = (uint32_t)(gid % GS);
    const bool live = j < out_len;
    lens += (uint64_t)b * count; scalars += (uint64_t)b * count;
    Acc acc = Z::template relax<1, 1024>(Z::zero());
    if (live) {
        auto term = [&](uint32_t k) { return polys[k].p[(uint64_t)b * polys[k].lane_stride + j]; };
        for (uint32_t k = sub; k < count; k += 2 * GS) {
            const uint32_t k2 = k + GS;
            const bool has1 = j < lens[k], has2 = k2 < count && j < lens[k2];
            if (has1 && has2) {
                const auto t = Z::mul2(Z::ld(scalars[k]), Z::ld(term(k)), Z::ld(scalars[k2]), Z::ld(term(k2)));
                acc = Z::template assume<1, 1024>(Z::norm(Z::add(acc, t)));
            } else if (has1 || has2) {
                const uint32_t kk = has1 ? k : k2;
                const auto t = Z::mul(Z::ld(scalars[kk]), Z::ld(term(kk)));
                acc = Z::template assume<1, 1024>(Z::norm(Z::add(acc, t)));
            }
        }
    }
    if constexpr (GS > 1) {
#pragma unroll
        for (int o = GS / 2; o > 0; o >>= 1) {
            Acc q;
#pragma unroll
            for (int w = 0; w < 9; ++w) q.v.l[w] = (uint32_t)__shfl_down((int)acc.v.l[w], o);
            if (sub + (uint32_t)o < (uint32_t)GS) acc = Z::template assume<1, 1024>(Z::norm(Z::add(acc, q)));
        }
    }
    if (live && sub == 0) out[(uint64_t)b * out_stride + j] = Z::to_wire(acc);
#endif
}


int poly_lincomb_lanes(Ctx& c, const void* d_polys, uint32_t count, const uint32_t* d_lens, const Fp* d_scalars, uint32_t lanes, Fp* d_out, uint64_t out_stride,
                       uint64_t out_len) {
    if (count == 0 || lanes == 0 || out_len == 0) return UZK_OK;
    KernelScope ks(c, "poly_lincomb");
    const LinPoly* polys = static_cast<const LinPoly*>(d_polys);
    const bool a29 = (c.tune_arith29 & 2) != 0 && count <= 128;      // (the 29-bit kernel's running sum is bounded for <= 128 terms)
    if (a29 && out_len <= (1ull << 17) && count >= 8)
        hipLaunchKernelGGL(poly_lincomb_lanes29_kernel<4>, dim3((unsigned)((out_len * 4 + 255) / 256), lanes), dim3(256), 0, c.stream, polys, count, d_lens, d_scalars, d_out, out_stride, out_len);
    else if (a29)
        hipLaunchKernelGGL(poly_lincomb_lanes29_kernel<1>, dim3((unsigned)((out_len + 255) / 256), lanes), dim3(256), 0, c.stream, polys, count, d_lens, d_scalars, d_out, out_stride, out_len);
    else if (out_len <= (1ull << 17) && count >= 8)
        hipLaunchKernelGGL(poly_lincomb_lanes_kernel<4>, dim3((unsigned)((out_len * 4 + 255) / 256), lanes), dim3(256), 0, c.stream, polys, count, d_lens, d_scalars, d_out, out_stride, out_len);
    else
        hipLaunchKernelGGL(poly_lincomb_lanes_kernel<1>, dim3((unsigned)((out_len + 255) / 256), lanes), dim3(256), 0, c.stream, polys, count, d_lens, d_scalars, d_out, out_stride, out_len);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}
void lin_poly_fill(void* host_entry, const void* p, uint64_t lane_stride) {
    LinPoly& e = *static_cast<LinPoly*>(host_entry);
    e.p = static_cast<const Fp*>(p); e.lane_stride = lane_stride;
}
size_t lin_poly_bytes() { return sizeof(LinPoly); }

// ---- division by X - z ----------------------------------------------------------------------------------------------------
// The blocked scan of poly.hip's open_div_* kernels with one opening per blockIdx.y: h_v of n coefficients at h + v * h_stride,
// its point's powers at pw[v] (device memory).  PER coefficients per lane.
struct DivPowsL { Fp z; Fp zp[8]; Fp zb[8]; };        // zp[k] = z^(PER 2^k), zb[k] = z^(256 PER 2^k)
template <int PER>
__global__ __launch_bounds__(256) void open_div_block_lanes_kernel(const Fp* __restrict__ h, uint64_t h_stride, uint64_t n, const DivPowsL* __restrict__ pws,
                                                                   Fp* __restrict__ s_out, Fp* __restrict__ block_first) {
    constexpr int BLOCK = 256 * PER;
    __shared__ Fp sh[256];
    const uint32_t tid = threadIdx.x, v = blockIdx.y;
    const DivPowsL& pw = pws[v];
    h += (uint64_t)v * h_stride; s_out += (uint64_t)v * n;
    const uint64_t lo = (uint64_t)blockIdx.x * BLOCK + (uint64_t)tid * PER;
    const Fp z = pw.z;
    Fp val[PER];
    Fp run = Fr::zero();
#pragma unroll
    for (int e = PER - 1; e >= 0; --e) {
        const uint64_t i = lo + e;
        run = Fr::mul(run, z);
        if (i < n) run = Fr::add(run, h[i]);
        val[e] = run;
    }
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
    Fp carry = (tid + 1 < 256) ? sh[tid + 1] : Fr::zero();
    Fp zp = z;
#pragma unroll
    for (int e = PER - 1; e >= 0; --e) {
        const uint64_t i = lo + e;
        if (i < n) s_out[i] = Fr::add(val[e], Fr::mul(zp, carry));
        zp = Fr::mul(zp, z);
    }
    if (tid == 0) block_first[(uint64_t)v * 256 + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(512) void open_div_carry_lanes_kernel(const Fp* __restrict__ block_first, uint32_t nblocks, const DivPowsL* __restrict__ pws,
                                                                   Fp* __restrict__ carry, Fp* __restrict__ ztab) {
    __shared__ Fp sh[256], zt[256];
    const uint32_t t = threadIdx.x & 255, v = blockIdx.x;
    const bool table = threadIdx.x >= 256;
    const DivPowsL& pw = pws[v];
    block_first += (uint64_t)v * 256; carry += (uint64_t)v * 256; ztab += (uint64_t)v * 257;
    if (!table) sh[t] = t < nblocks ? block_first[t] : Fr::zero();
    else if (t == 0) zt[0] = Fr::one();
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t off = 1u << k;
        Fp x = Fr::zero();
        if (!table && t + off < 256) x = sh[t + off];
        __syncthreads();
        if (!table) { if (t + off < 256) sh[t] = Fr::add(sh[t], Fr::mul(pw.zb[k], x)); }
        else if (t < off) zt[off + t] = Fr::mul(zt[t], pw.zp[k]);
        __syncthreads();
    }
    if (!table) { if (t < nblocks) carry[t] = (t + 1 < 256) ? sh[t + 1] : Fr::zero(); }
    else {
        ztab[t] = zt[t];
        if (t == 0) ztab[256] = pw.zb[0];
    }
}
template <int PER>
__global__ __launch_bounds__(256) void open_div_apply_lanes_kernel(const Fp* __restrict__ s, const Fp* __restrict__ carry, const Fp* __restrict__ ztab,
                                                                   const DivPowsL* __restrict__ pws, uint64_t n, uint64_t q_cap, Fp* __restrict__ q, uint64_t q_stride) {
    constexpr int BLOCK = 256 * PER;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = blockIdx.y;
    if (i >= q_cap) return;
    q += (uint64_t)v * q_stride;
    if (i >= n) { q[i] = Fr::zero(); return; }
    if (i == 0) { q[n - 1] = Fr::zero(); return; }
    const uint64_t blk = i / BLOCK;
    const uint32_t dist = (uint32_t)((blk + 1) * BLOCK - i);
    Fp val = s[(uint64_t)v * n + i];
    const Fp cb = carry[(uint64_t)v * 256 + blk];
    if (!Fr::is_zero(cb)) {
        const Fp z = pws[v].z;
        Fp zp = ztab[(uint64_t)v * 257 + dist / PER];
        for (uint32_t r = 0; r < dist % PER; ++r) zp = Fr::mul(zp, z);
        val = Fr::add(val, Fr::mul(zp, cb));
    }
    q[i - 1] = val;
}
// the powers of one opening's point for PER coefficients per lane
void div_pows_fill(void* host_entry, const Fp& z, int per) {
    DivPowsL& pw = *static_cast<DivPowsL*>(host_entry);
    pw.z = z;
    Fp p = z;
    for (int i = 1; i < per; i <<= 1) p = Fr::sqr(p);
    for (int k = 0; k < 8; ++k) { pw.zp[k] = p; p = Fr::sqr(p); }
    for (int k = 0; k < 8; ++k) { pw.zb[k] = p; p = Fr::sqr(p); }
}
size_t div_pows_bytes() { return sizeof(DivPowsL); }
int open_div_per(uint64_t n) { return n <= (1ull << 16) ? 4 : 16; }
// q_v = h_v div (X - z_v) for `count` openings: h_v (n coefficients) at d_h + v * h_stride, q_v (n - 1 coefficients, then zeros
// up to q_cap) at d_q + v * q_stride; d_pows: `count` entries of div_pows_fill(open_div_per(n)).  n <= 2^20.  Asynchronous.
int open_div_lanes(Ctx& c, const Fp* d_h, uint64_t h_stride, uint64_t n, const void* d_pows, uint32_t count, Fp* d_q, uint64_t q_stride, uint64_t q_cap) {
    if (count == 0) return UZK_OK;
    if (n == 0 || n > (1ull << 20) || q_cap < n) { set_error("open_div_lanes: need 1 <= n <= 2^20 and q_cap >= n"); return UZK_ERR_PARAMETER; }
    const int per = open_div_per(n);
    const uint32_t nblocks = (uint32_t)((n + 256ull * per - 1) / (256ull * per));
    // s[count][n] | block_first[count][256] | carry[count][256] | ztab[count][257]
    UZK_TRY(c.open_tmp.reserve(((size_t)count * (n + 256 + 256 + 257)) * sizeof(Fp)));
    Fp* d_s = c.open_tmp.as<Fp>();
    Fp* d_first = d_s + (size_t)count * n;
    Fp* d_carry = d_first + (size_t)count * 256;
    Fp* d_ztab = d_carry + (size_t)count * 256;
    const DivPowsL* pw = static_cast<const DivPowsL*>(d_pows);
    KernelScope ks(c, "open_quotient");
    if (per == 4) hipLaunchKernelGGL(open_div_block_lanes_kernel<4>, dim3(nblocks, count), dim3(256), 0, c.stream, d_h, h_stride, n, pw, d_s, d_first);
    else hipLaunchKernelGGL(open_div_block_lanes_kernel<16>, dim3(nblocks, count), dim3(256), 0, c.stream, d_h, h_stride, n, pw, d_s, d_first);
    hipLaunchKernelGGL(open_div_carry_lanes_kernel, dim3(count), dim3(512), 0, c.stream, d_first, nblocks, pw, d_carry, d_ztab);
    if (per == 4) hipLaunchKernelGGL(open_div_apply_lanes_kernel<4>, dim3((unsigned)((q_cap + 255) / 256), count), dim3(256), 0, c.stream, d_s, d_carry, d_ztab, pw, n, q_cap, d_q, q_stride);
    else hipLaunchKernelGGL(open_div_apply_lanes_kernel<16>, dim3((unsigned)((q_cap + 255) / 256), count), dim3(256), 0, c.stream, d_s, d_carry, d_ztab, pw, n, q_cap, d_q, q_stride);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- split of t -----------------------------------------------------------------------------------------------------------
// split_t_kernel (poly.hip) per lane: t_b (t_len[b] coefficients) at t + b * t_stride, rands[b][n_chunks], chunks to
// out + (b * n_chunks + i) * chunk_stride.
__global__ __launch_bounds__(256) void split_t_lanes_kernel(const Fp* __restrict__ t, uint64_t t_stride, const uint32_t* __restrict__ t_lens, uint64_t chunk,
                                                            uint32_t n_chunks, const Fp* __restrict__ rands, Fp* __restrict__ out, uint64_t chunk_stride) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y, b = blockIdx.z;
    if (j >= chunk_stride) return;
    t += (uint64_t)b * t_stride; rands += (uint64_t)b * n_chunks;
    const uint64_t t_len = t_lens[b];
    const bool last = i + 1 == n_chunks;
    const uint64_t start = (uint64_t)i * chunk;
    const uint64_t end = last ? t_len : min(t_len, start + chunk);
    Fp v = (start + j < end) ? t[start + j] : Fr::zero();
    if (!last && j == chunk) v = Fr::add(v, rands[i]);
    if (j == 0 && i > 0) v = Fr::sub(v, rands[i - 1]);
    out[((uint64_t)b * n_chunks + i) * chunk_stride + j] = v;
}
int split_t_lanes(Ctx& c, const Fp* d_t, uint64_t t_stride, const uint32_t* d_t_lens, uint64_t chunk, uint32_t n_chunks, const Fp* d_rands, uint32_t lanes,
                  Fp* d_chunks, uint64_t chunk_stride) {
    if (lanes == 0 || n_chunks == 0) return UZK_OK;
    KernelScope ks(c, "split_t");
    hipLaunchKernelGGL(split_t_lanes_kernel, dim3((unsigned)((chunk_stride + 255) / 256), n_chunks, lanes), dim3(256), 0, c.stream, d_t, t_stride, d_t_lens, chunk,
                       n_chunks, d_rands, d_chunks, chunk_stride);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- fold modulo X^N - 1 --------------------------------------------------------------------------------------------------
// fold_blinds_batch_kernel (poly.hip) with the lengths in device memory: polynomial v (lens[v] coefficients) at
// polys + v * in_stride; any number of polynomials.
__global__ __launch_bounds__(256) void fold_blinds_lanes_kernel(const Fp* __restrict__ polys, uint64_t in_stride, const uint32_t* __restrict__ lens, uint64_t N,
                                                                Fp* __restrict__ out, uint64_t out_stride, Fp* __restrict__ tail, uint32_t tail_n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = blockIdx.y;
    if (i >= N) return;
    const uint64_t len = lens[v];
    const Fp* c = polys + (uint64_t)v * in_stride;
    Fp val = i < len ? c[i] : Fr::zero();
    const uint32_t K = tail_n / 2;
    if (N + i < len) {
        const Fp hi = c[N + i];
        val = Fr::add(val, hi);
        tail[(uint64_t)v * tail_n + i] = Fr::neg(hi);
        tail[(uint64_t)v * tail_n + K + i] = hi;
    } else if (i < K) {
        tail[(uint64_t)v * tail_n + i] = Fr::zero();
        tail[(uint64_t)v * tail_n + K + i] = Fr::zero();
    }
    out[(uint64_t)v * out_stride + i] = val;
}
int fold_blinds_lanes(Ctx& c, const Fp* d_polys, uint64_t in_stride, const uint32_t* d_lens, uint64_t N, uint32_t count, Fp* d_out, uint64_t out_stride, Fp* d_tail,
                      uint32_t tail_n) {
    if (count == 0) return UZK_OK;
    KernelScope ks(c, "fold_blinds");
    hipLaunchKernelGGL(fold_blinds_lanes_kernel, dim3((unsigned)((N + 255) / 256), count), dim3(256), 0, c.stream, d_polys, in_stride, d_lens, N, d_out, out_stride,
                       d_tail, tail_n);
    UZK_HIP(hipGetLastError());
    return UZK_OK;
}

// ---- trimmed lengths ------------------------------------------------------------------------------------------------------
// out[v] = 1 + the highest index below `cap` holding a non-zero coefficient of polynomial v (polys + v * stride), 0 for none.
// results: two sets of `count_max` words used alternately -- every launch accumulates into one and clears the other for the
// launch after it (both zero to begin with), so no fill launch is needed (poly.hip poly_trimmed_len_kernel).
__global__ __launch_bounds__(256) void trimmed_len_lanes_kernel(const Fp* __restrict__ polys, uint64_t stride, uint32_t cap, unsigned long long* __restrict__ out,
                                                                unsigned long long* __restrict__ clear_for_next, uint32_t count_max) {
    __shared__ unsigned long long top_of_block;
    const uint32_t v = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && v == 0)
        for (uint32_t k = threadIdx.x; k < count_max; k += blockDim.x) clear_for_next[k] = 0;
    if (threadIdx.x == 0) top_of_block = 0;
    __syncthreads();
    const bool nz = i < cap && !Fr::is_zero(polys[(uint64_t)v * stride + i]);
    const unsigned long long mask = __ballot(nz);
    if (mask != 0 && (threadIdx.x & 63) == 0) atomicMax(&top_of_block, (unsigned long long)((i & ~63ull) + (63 - __clzll(mask)) + 1));
    __syncthreads();
    if (threadIdx.x == 0 && top_of_block > out[v]) atomicMax(&out[v], top_of_block);
}
// d_sets: 2 * count_max words, zero when first used; *flip: the caller's toggle; out_host_pinned receives `count` words in stream order.
int trimmed_len_lanes(Ctx& c, const Fp* d_polys, uint64_t stride, uint64_t cap, uint32_t count, uint64_t* d_sets, uint32_t count_max, uint32_t* flip,
                      uint64_t* out_host_pinned) {
    if (count == 0) return UZK_OK;
    if (count > count_max || cap == 0 || cap >= (1ull << 32)) { set_error("trimmed_len_lanes: bad shape"); return UZK_ERR_PARAMETER; }
    *flip ^= 1u;
    unsigned long long* sets = reinterpret_cast<unsigned long long*>(d_sets);
    unsigned long long* d_res = sets + (size_t)*flip * count_max;
    {
        KernelScope ks(c, "poly_trimmed_len");
        hipLaunchKernelGGL(trimmed_len_lanes_kernel, dim3((unsigned)((cap + 255) / 256), count), dim3(256), 0, c.stream, d_polys, stride, (uint32_t)cap, d_res,
                           sets + (size_t)(*flip ^ 1u) * count_max, count_max);
    }
    UZK_HIP(hipGetLastError());
    UZK_HIP(hipMemcpyAsync(out_host_pinned, d_res, count * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream));
    return UZK_OK;
}

}  // namespace uzk
