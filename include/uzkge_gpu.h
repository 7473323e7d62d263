/*
 * uzkge_gpu.h -- C ABI of the MI355X (gfx950) backend for uzkge's PlonK prover hot path.
 *
 * The reference (zypher-game/uzkge, /root/reference) has no FFI/plugin surface; the boundary is
 * the pair of Rust call sites that carry the heavy arithmetic:
 *
 *   G1Projective::msm(&points_raw, &coefs)   uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290
 *   domain.fft(&self.coefs)                   uzkge/src/poly_commit/field_polynomial.rs:585
 *   domain.ifft(&values)                      uzkge/src/poly_commit/field_polynomial.rs:595
 *
 * A Rust `uzkge-gpu-sys` shim binds exactly these symbols (INTEGRATION.md shows the binding).
 *
 * Data layout (bit-identical to ark-ff `Fp<MontBackend<_,4>,4>`'s inner `BigInt<4>`):
 *   field element  = 4 x uint64 little-endian limbs, Montgomery form, R = 2^256
 *   uzk_g1_affine  = x || y (64 B); the point at infinity is encoded as x = y = 0
 *   uzk_g1_jac     = x || y || z (96 B), Jacobian (x/z^2, y/z^3); infinity <=> z == 0
 * The Jacobian representative of a result is not unique; compare after affine normalisation.
 *
 * Ownership: every pointer is caller-owned for the duration of the call; the library keeps no
 * reference except data explicitly copied (or adopted) by uzk_srs_register*.  Outputs go to
 * caller memory.  All functions are thread-safe (one lock per context, see uzk_ctx_create); none panics or aborts.
 *
 * Error codes map onto `UzkgeError` (uzkge/src/errors.rs:5-44):
 *   UZK_ERR_DEGREE     -> UzkgeError::DegreeError      (commit: len > SRS len, kzg_poly_commitment.rs:283-285)
 *   UZK_ERR_FFT        -> UzkgeError::FFTError         (no evaluation domain of that size)
 *   UZK_ERR_COMMITMENT -> UzkgeError::CommitmentError  (length mismatch: ark `msm` Err(min_len))
 *   UZK_ERR_PARAMETER  -> UzkgeError::ParameterError   (bad handle / null pointer)
 *   UZK_ERR_DEVICE     -> (new) no gfx950 device, HIP runtime failure, out of device memory; also out of HOST memory inside the
 *                         library (no C++ exception leaves an entry point: each is a function-try-block)
 * There is NO CPU fallback: without a usable GPU every compute entry point returns UZK_ERR_DEVICE.
 */
#ifndef UZKGE_GPU_H
#define UZKGE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UZK_OK 0
#define UZK_ERR_PARAMETER 1
#define UZK_ERR_DEGREE 2
#define UZK_ERR_FFT 3
#define UZK_ERR_COMMITMENT 4
#define UZK_ERR_DEVICE 5

typedef struct { uint64_t x[4], y[4]; } uzk_g1_affine;
typedef struct { uint64_t x[4], y[4], z[4]; } uzk_g1_jac;

/* ---- lifecycle ------------------------------------------------------------------------ */
/* Bind the calling process to HIP device `device` (one process per GPU); idempotent for the
 * same ordinal.  Creates the library stream and workspaces.  The library never edits the process environment: a host
 * that runs more than four prover threads (one context each) should export GPU_MAX_HW_QUEUES=16 before the process
 * makes its first HIP call -- the HIP runtime multiplexes streams onto four hardware queues by default. */
int uzk_init(int device);
int uzk_shutdown(void);
/* Number of visible HIP devices (0 when none / no driver); never fails. */
int uzk_device_count(void);
/* Thread-local description of the last non-OK return on this thread. */
const char* uzk_last_error(void);
const char* uzk_version(void);

/* ---- contexts (optional) ---------------------------------------------------------------- */
/* A context = one stream, one set of workspaces, one lock.  Every entry point works on the calling thread's current
 * context; threads that never set one share the default context (and serialise on its lock).  Giving each prover
 * thread its own context lets independent proofs overlap on the GPU -- at the real circuit size (n = 2^14) a single
 * proof leaves most of the chip idle.  SRS handles are process-wide and may be used from every context; register and
 * precompute them before the contexts start sharing them. */
/* The new context starts with the creator's current tuning (uzk_tune, uzk_msm_set_window_bits). */
int uzk_ctx_create(uint64_t* ctx_out);
/* A context on a named device ("uzk_init(n_devices)"): one process can then drive several GPUs -- prover threads
 * with a context, circuits and provers each on their own device, or the point chunks of one MSM (uzk_msm_g1_sharded).
 * uzk_init keeps its meaning: the device of the default context and of uzk_ctx_create.  SRS handles, circuits and provers
 * live on the device of the context that made them; using one from a context on another device is UZK_ERR_PARAMETER. */
int uzk_ctx_create_on(int device, uint64_t* ctx_out);
/* the device context `ctx` (0: the default context) lives on */
int uzk_ctx_device(uint64_t ctx, int* device_out);
/* Makes `ctx` (0 = the default context) the calling thread's current context. */
int uzk_ctx_set_current(uint64_t ctx);
/* Frees the context's stream, plans and workspaces; no thread may be inside a call on it.  A thread whose current
 * context has been destroyed (here or by uzk_shutdown) works on the default context from its next call on. */
int uzk_ctx_destroy(uint64_t ctx);
/* The calling thread's current context waits -- on the device, the host does not block -- for everything queued so far on
 * context `other` (0 = the default context).  A prover thread that owns TWO contexts has two lanes, each with its own stream
 * and workspaces: steps of a proof that do not depend on each other run side by side (the coset FFTs of the wire polynomials
 * under the commit of the same round, uzkge/src/plonk/prover.rs:160-192 / helpers.rs:256-266; the two batch_prove openings,
 * prover.rs:349-372), and this call is the dependency edge between the lanes. */
int uzk_ctx_wait(uint64_t other);
/* The calling thread's current context (0 = the default one). */
int uzk_ctx_current(uint64_t* ctx_out);

/* ---- device memory ------------------------------------------------------------------------ */
/* Everything a host language needs to keep data resident between the *_device entry points: with these it links
 * libuzkge_gpu.so and nothing else (no HIP runtime, no hip headers).  The reference keeps every polynomial of a proof in
 * Vec<Fr> between its steps (uzkge/src/plonk/prover.rs:151-372); these are the device-side Vecs.
 * Copies and fills are ordered on the calling context's stream, i.e. with that context's kernels:
 *   UZK_COPY_D2H  the data is in `dst` when the call returns;
 *   UZK_COPY_H2D  `src` may be reused when the call returns -- except for memory of uzk_host_alloc (pinned), whose
 *                 uploads are asynchronous: keep it unchanged until the next synchronising call (uzk_sync, a D2H copy, a
 *                 commit);
 *   UZK_COPY_D2D  asynchronous.
 * uzk_dev_free / uzk_host_free wait for the calling context's stream first.  Allocations belong to the caller and
 * survive uzk_shutdown. */
#define UZK_COPY_H2D 0
#define UZK_COPY_D2H 1
#define UZK_COPY_D2D 2
int uzk_dev_alloc(size_t bytes, void** d_out);
int uzk_dev_free(void* d_ptr);
int uzk_host_alloc(size_t bytes, void** h_out);
int uzk_host_free(void* h_ptr);
int uzk_dev_copy(void* dst, const void* src, size_t bytes, int kind);
/* `rows` rows of `width` bytes; consecutive rows are dst_pitch / src_pitch bytes apart. */
int uzk_dev_copy2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows, int kind);
int uzk_dev_memset(void* d_dst, int byte, size_t bytes);
int uzk_dev_memset2d(void* d_dst, size_t pitch, int byte, size_t width, size_t rows);

/* ---- SRS (the static bases of KZG commit) --------------------------------------------- */
/* Copy `n` affine points to HBM once; replaces the per-commit `normalize_batch` of
 * kzg_poly_commitment.rs:287-288.  Returns an opaque handle (never 0). */
int uzk_srs_register(const uzk_g1_affine* points, size_t n, uint64_t* handle_out);
/* Adopt points that already live in device memory (no copy; caller keeps them alive). */
int uzk_srs_register_device(const void* d_points, size_t n, uint64_t* handle_out);
int uzk_srs_release(uint64_t handle);
/* Optional, for a static SRS (KZG): build the window table T[j][i] = 2^(c*j) * SRS[i] in HBM
 * (W * n * 64 bytes, W = ceil(254 / c); c = window_bits or automatic when 0).  Later MSMs on this
 * handle then use one shared bucket set (larger windows, no per-window reduction, no host
 * combination).  Results are identical; only speed and memory change. */
int uzk_srs_precompute(uint64_t handle, int window_bits);
int uzk_srs_len(uint64_t handle, size_t* n_out);

/* ---- sharded SRS: one MSM over several GPUs from one process ------------------------------------------------------------------
 * north_star / SURVEY.md 8e: "MSM shards by point-chunk across the 8 GPUs of one node".  The bases are cut into n_devices
 * contiguous chunks (chunk i = [i n / N, (i + 1) n / N)), chunk i is copied to the HBM of devices[i] once and stays there; an MSM
 * call runs the chunks side by side (one host thread each, each uploading only its part of the scalars) and folds the N 96-byte
 * partial sums on the host (uzk_g1_fold).  Bases and scalars never move between devices.  devices[] may name an ordinal more
 * than once (several chunks on one GPU: what the single-GPU tests do).  window_bits: -1 no window table, 0 automatic, 4 .. 24.
 * (One process per GPU with the partial sums exchanged by RCCL is the other form of the same split: uzkge_amd/sharded.py,
 * bench.py --gpus N.) */
int uzk_srs_register_sharded(const uzk_g1_affine* points, size_t n, const int* devices, uint32_t n_devices, int window_bits, uint64_t* handle_out);
int uzk_srs_release_sharded(uint64_t handle);
/* sum_i scalars[i] * SRS[i], i < n <= the SRS length (a short vector touches the first chunks only).  partials_out (optional):
 * the n_devices partial sums in chunk order. */
int uzk_msm_g1_sharded(uint64_t handle, const uint64_t* scalars_mont, size_t n, uzk_g1_jac* partials_out, uzk_g1_jac* out);
/* length, number of chunks, their devices (n_chunks entries) and bounds (2 n_chunks entries: lo, hi); every output optional */
int uzk_srs_sharded_info(uint64_t handle, size_t* n_out, uint32_t* n_chunks_out, int* devices_out, size_t* bounds_out);

/* ---- MSM: replaces G1Projective::msm (kzg_poly_commitment.rs:290) ---------------------- */
/* out = sum_{i<n} scalars[i] * SRS[offset + i].  n == 0 -> infinity.  offset + n > len ->
 * UZK_ERR_DEGREE.  Zero scalars and infinity bases contribute the identity. */
int uzk_msm_g1(uint64_t srs_handle, size_t offset, const uint64_t* scalars_mont, size_t n,
               uzk_g1_jac* out);
/* Same with the scalars already resident in device memory (the bench / pipelined path). */
int uzk_msm_g1_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t n,
                      uzk_g1_jac* out);
/* `batch` scalar vectors of n elements each (scalars[b*n + i]) against the SAME bases
 * SRS[offset .. offset+n): out[b] = sum_i scalars[b][i] * SRS[offset+i].  One launch sequence for
 * the prover's independent commits (5 wires + 3 selectors, prover.rs:160-192; the 5 chunks of t,
 * helpers.rs:1390); shorter polynomials are zero-padded by the caller. */
int uzk_msm_g1_batch(uint64_t srs_handle, size_t offset, const uint64_t* scalars_mont, size_t n, uint32_t batch,
                     uzk_g1_jac* out);
int uzk_msm_g1_batch_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t n,
                            uint32_t batch, uzk_g1_jac* out);
/* The prover's commit closure on its Lagrange branch (uzkge/src/plonk/prover.rs:132-142) in ONE batched MSM: vector b has
 * n scalars at d_scalars + b * stride (device) followed by tail_n "tail" scalars (tail + b * tail_n; host memory unless
 * tail_on_device):
 *   out[b] = sum_{i<n} scalars_b[i] * SRS[offset + i]  +  sum_{j<tail_n} tail_b[j] * SRS[offset + n + j].
 * With the bases registered as  lagrange[0..n) || pcs[0..K) || pcs[zd..zd+K)  and the tail  blinds || -blinds  this is
 * lagrange_pcs.commit(evals) followed by pcs.apply_blind_factors(cm, blinds, zd) (kzg_poly_commitment.rs:299-313): the blind
 * terms ride in the commit's own MSM.  The stride lets the evaluation vectors stay where they are (no gather into a staging
 * array); a device tail is what uzk_fold_blinds_batch_device leaves behind.  n + tail_n <= 2^26. */
int uzk_msm_g1_batch_tail_device(uint64_t srs_handle, size_t offset, const void* d_scalars_mont, size_t stride, size_t n,
                                 uint32_t batch, const void* tail_scalars_mont, uint32_t tail_n, int tail_on_device, uzk_g1_jac* out);
/* One-shot, nothing cached: points and scalars are host arrays of the same length. */
int uzk_msm_g1_raw(const uzk_g1_affine* points, const uint64_t* scalars_mont, size_t n,
                   uzk_g1_jac* out);
/* Host-side fold of partial sums (point-chunk sharded MSM: ranks all-gather their 96-byte
 * partials over RCCL, then every rank folds; EC addition is not an RCCL reduce op). */
int uzk_g1_fold(const uzk_g1_jac* partials, size_t count, uzk_g1_jac* out);
/* Host-side Jacobian -> affine (for comparing results). */
int uzk_g1_to_affine(const uzk_g1_jac* p, uzk_g1_affine* out);

/* ---- NTT: replaces EvaluationDomain::{fft, ifft} (field_polynomial.rs:585,595) --------- */
/* The largest transforms this library runs: n = 2^k with k <= UZK_NTT_MAX_LOG2, n = 3 * 2^k with k <= UZK_NTT_MAX_LOG2_MIXED.
 * These are exactly the largest sizes the parity suite compares with the CPU oracle on the whole vector
 * (tests/test_gpu_ntt_large.py: 2^25 is the first four-pass plan); a size is accepted only if its result is checked.
 * (Fr has domains up to 2^28 / 3 * 2^28; the prover of the reference uses n = 2^14 and 6n = 3 * 2^15.) */
#define UZK_NTT_MAX_LOG2 25
#define UZK_NTT_MAX_LOG2_MIXED 22
/* 1 if this library transforms over the size-n domain (n = 2^k, k <= UZK_NTT_MAX_LOG2, or n = 3 * 2^k,
 * k <= UZK_NTT_MAX_LOG2_MIXED), else 0: the shim's FpPolynomial::{evaluation_domain, quotient_evaluation_domain}
 * (field_polynomial.rs:554-567) returns None on 0, which its callers turn into UzkgeError::FFTError. */
int uzk_domain_supported(uint64_t n);
/* group_gen of the size-n domain (5^((r-1)/n)), Montgomery form; answers for every domain Fr has (2^k, 3 * 2^k, k <= 28). */
int uzk_domain_group_gen(uint64_t n, uint64_t out_mont[4]);
/* In-place transform of `n` elements in natural order over <group_gen(n)>.
 * inverse != 0: includes the 1/n scaling.  coset_shift (4 limbs, Montgomery) or NULL:
 * forward: c_j *= shift^j before the transform (coset_fft_with_domain, :589-591);
 * inverse: c_j *= shift^j after the transform -- pass k^-1 (coset_ifft_with_domain, :601-607).
 * The caller zero-pads short inputs (it owns the Vec). */
int uzk_ntt_fr(uint64_t* data, uint64_t n, int inverse, const uint64_t* coset_shift_mont);
/* Device-resident variant: d_in -> d_out (may alias), asynchronous on the library stream
 * unless `sync` != 0. */
int uzk_ntt_fr_device(const void* d_in, void* d_out, uint64_t n, int inverse,
                      const uint64_t* coset_shift_mont, int sync);

/* `batch` independent transforms of the same size over contiguous vectors (the prover's 5 wire +
 * 3 selector iFFTs, prover.rs:160-192, or the ten coset FFTs of t_poly, helpers.rs:256-266, in one
 * launch sequence instead of 8-10). */
int uzk_ntt_fr_batch(uint64_t* data, uint64_t n, uint32_t batch, int inverse, const uint64_t* coset_shift_mont);
int uzk_ntt_fr_batch_device(const void* d_in, void* d_out, uint64_t n, uint32_t batch, int inverse,
                            const uint64_t* coset_shift_mont, int sync);

/* The same with consecutive vectors in_stride / out_stride elements apart (>= n; d_in == d_out needs equal strides): a batch
 * that reads or writes slots of a wider array -- the prover's iFFT results go straight into the 6n-element slots its coset
 * FFTs read (uzkge/src/plonk/prover.rs:160-175 -> helpers.rs:256-266), no copy in between. */
int uzk_ntt_fr_batch_strided_device(const void* d_in, uint64_t in_stride, void* d_out, uint64_t out_stride, uint64_t n,
                                    uint32_t batch, int inverse, const uint64_t* coset_shift_mont, int sync);

/* ---- polynomial helpers next to the hot path ------------------------ */
/* out[b] = sum_j coefs[b*n + j] * x^j : FpPolynomial::eval (field_polynomial.rs:198-209) for a batch of
 * polynomials at one point (the prover's 19 + 20 openings at zeta / zeta*omega, prover.rs:246-273). */
int uzk_poly_eval_batch(const uint64_t* coefs, uint64_t n, uint32_t batch, const uint64_t* x_mont, uint64_t* out);
int uzk_poly_eval_batch_device(const void* d_coefs, uint64_t n, uint32_t batch, const uint64_t* x_mont, uint64_t* out);
/* The permutation grand product z_poly (uzkge/src/plonk/helpers.rs:160-220), evaluations only:
 * z[0] = 1, z[i+1] = z[i] * prod_j (w[j*n+i] + beta*k[j]*group[i] + gamma)
 *                         / (w[j*n+i] + beta*k[perm/n]*group[perm%n] + gamma),  perm = perm[j*n+i].
 * w: n_wires*n elements, perm: n_wires*n indices < n_wires*n, group: n elements (omega^i), k: n_wires. */
int uzk_z_poly(const uint64_t* w, const uint32_t* perm, const uint64_t* group, const uint64_t* k,
               const uint64_t* beta_mont, const uint64_t* gamma_mont, uint32_t n, uint32_t n_wires, uint64_t* z_out);

/* The same on device-resident inputs (w, group: device Fr vectors; perm: device u32; k and the challenges on the
 * host), z written to d_z (n elements, device). */
int uzk_z_poly_device(const void* d_w, const uint32_t* d_perm, const void* d_group, const uint64_t* k,
                      const uint64_t* beta_mont, const uint64_t* gamma_mont, uint32_t n, uint32_t n_wires, void* d_z);

/* batch_prove's polynomial work (uzkge/src/poly_commit/pcs.rs:119-135 with div_rem,
 * field_polynomial.rs:519-550): evals_out[k] = p_k(z);  h = sum_k alpha^k (p_k - p_k(z));
 * q = h / (X - z)  (the remainder is zero by construction).
 * d_polys: batch polynomials of n coefficients each (device, zero-padded to the common n);
 * d_q: n elements on the device, q's n - 1 coefficients followed by a zero; evals_out: batch x 4 words
 * (host).  n <= 2^20, batch <= 4096.  The caller commits q (or folds it for the Lagrange path,
 * pcs.rs:137-166) exactly as the reference does. */
int uzk_open_quotient_device(const void* d_polys, uint64_t n, uint32_t batch, const uint64_t* z_mont,
                             const uint64_t* alpha_mont, void* d_q, uint64_t* evals_out);

/* The same on host arrays (polys: batch * n elements; q_out: n elements; evals_out: batch elements). */
int uzk_open_quotient(const uint64_t* polys, uint64_t n, uint32_t batch, const uint64_t* z_mont, const uint64_t* alpha_mont,
                      uint64_t* q_out, uint64_t* evals_out);

/* The fold modulo X^N - 1 in front of every Lagrange-basis commit of a polynomial with len > N coefficients
 * (batch_prove, uzkge/src/poly_commit/pcs.rs:137-156; split_t_and_commit, uzkge/src/plonk/helpers.rs:1366-1383):
 *   blinds_out[i] = -coefs[N + i]  (i < len - N, host memory),
 *   d_out[i] = coefs[i] + coefs[N + i] for i < len - N, coefs[i] for len - N <= i < min(len, N), 0 beyond  (N elements, device).
 * len <= 2N.  The caller continues exactly as the reference: fft(N) of d_out, commit over the Lagrange SRS,
 * apply_blind_factors(blinds, N).  d_out may alias d_coefs. */
int uzk_fold_blinds_device(const void* d_coefs, uint64_t len, uint64_t n_fold, void* d_out, uint64_t* blinds_out);

/* d_out[j] = sum_k scalars[k] * d_polys[k][j] for j < out_len (a polynomial contributes zero beyond lens[k]): the shape of
 * r_poly (uzkge/src/plonk/helpers.rs:681-999, 1030-1090) -- about 43 device-resident polynomials times scalars that the
 * caller builds from the evaluations and challenges exactly as the reference does.  count <= 64; d_out aliases no input.
 * Asynchronous on the library stream (the scalars are copied before the call returns). */
int uzk_poly_lincomb_device(const void* const* d_polys, const uint64_t* lens, const uint64_t* scalars_mont, uint32_t count,
                            void* d_out, uint64_t out_len);
/* hide_polynomial (uzkge/src/plonk/helpers.rs:139-158) on device-resident coefficients (len >= zeroing_degree +
 * hiding_degree, zero-padded by the caller): coefs[i] += blinds[i], coefs[zeroing_degree + i] -= blinds[i].
 * hiding_degree <= 16.  Asynchronous on the library stream. */
int uzk_hide_polynomial_device(void* d_coefs, uint64_t len, const uint64_t* blinds_mont, uint32_t hiding_degree, uint64_t zeroing_degree);

/* ---- batched / pointer-list forms: one launch per prover step instead of one per polynomial ---------------------------- */
/* hide_polynomial (helpers.rs:139-158) for `count` polynomials `stride` elements apart holding len_in coefficients each:
 * as the reference's `resize`, slots [len_in, zeroing_degree + hiding_degree) are written (zeros, then minus the blinds),
 * not read.  blinds: count * hiding_degree elements (host), count * hiding_degree <= 64.  Asynchronous. */
int uzk_hide_polynomial_batch_device(void* d_coefs, uint64_t stride, uint64_t len_in, uint32_t count, const uint64_t* blinds_mont,
                                     uint32_t hiding_degree, uint64_t zeroing_degree);
/* uzk_fold_blinds_device for `batch` <= 16 polynomials (lens[b] coefficients, in_stride apart), results out_stride apart, and
 * the scalars of apply_blind_factors left on the device as the tail of the following commit:
 *   d_tail[b * tail_n + i] = blind_i = -coefs_b[N + i],  d_tail[b * tail_n + tail_n / 2 + i] = -blind_i   (zero padded),
 * so fold -> uzk_ntt_fr_batch_strided_device -> uzk_msm_g1_batch_tail_device(.., d_tail, tail_n, 1, ..) is the tail of
 * batch_prove (pcs.rs:137-166) / split_t_and_commit (helpers.rs:1366-1394) without a host round trip.  blinds_out (optional,
 * host, batch * tail_n / 2 elements): the blinds as the reference's Vec; asking for them synchronises. */
int uzk_fold_blinds_batch_device(const void* d_polys, uint64_t in_stride, const uint64_t* lens, uint64_t n_fold, uint32_t batch,
                                 void* d_out, uint64_t out_stride, void* d_tail, uint32_t tail_n, uint64_t* blinds_out);
/* FpPolynomial::from_coefs trims trailing zeros (field_polynomial.rs:86-90) and the prover branches on the result: t's
 * coefs.len() (helpers.rs:1333), q.degree() (pcs.rs:138).  out_lens[b] = 1 + the highest index < lens[b] with a non-zero
 * coefficient of polynomial b (d_polys + b * stride), 0 for the zero polynomial.  batch <= 16.
 * sync != 0: the values are in out_lens on return.  sync == 0: out_lens must be uzk_host_alloc memory; the stream fills it in
 * order and the caller reads it after its next synchronising call -- a prover that KNOWS the lengths a well-formed proof has
 * (t: 5n + 11 coefficients, q: n + 2) goes on with them and checks at the round's commit that the device agrees, instead of
 * paying a synchronisation for an answer it already has. */
int uzk_poly_trimmed_len_device(const void* d_polys, uint64_t stride, const uint64_t* lens, uint32_t batch, uint64_t* out_lens, int sync);
/* The split of t in split_t_and_commit (helpers.rs:1335-1363); `chunk` is the reference's `n` argument (n_constraints + 2):
 * chunk i < last = t[i chunk .. (i+1) chunk) resized to chunk + 1 with coefs[chunk] += rands[i], coefs[0] -= rands[i-1];
 * the last chunk = t[last chunk .. t_len) (or [-rands[last-1]] if empty) with coefs[0] -= rands[last-1].  Written to
 * d_chunks + i * chunk_stride, zero padded to chunk_stride; lens_out[i] (optional, host) = the reference's coefs.len().
 * n_chunks <= 8.  Asynchronous. */
int uzk_split_t_device(const void* d_t, uint64_t t_len, uint64_t chunk, uint32_t n_chunks, const uint64_t* rands_mont,
                       void* d_chunks, uint64_t chunk_stride, uint64_t* lens_out);
/* out[k] = p_k(points[point_idx[k]]) for `count` device-resident polynomials of lens[k] coefficients: the prover's round 4
 * (prover.rs:246-273: 15 polynomials at zeta, 4 at zeta * omega) in one launch (count <= 64, n_points <= 4, lens <= 2^18;
 * beyond that one call per polynomial). */
int uzk_poly_eval_ptrs_device(const void* const* d_polys, const uint64_t* lens, const uint32_t* point_idx, uint32_t count,
                              const uint64_t* points_mont, uint32_t n_points, uint64_t* out);
/* uzk_open_quotient_device over a pointer list (polynomials of different lengths, wherever they live):
 * q = (sum_k alpha^k p_k) div (X - z) -> d_q: hlen - 1 coefficients (hlen = max lens[k]), zeros up to q_cap >= hlen.
 * evals_out (optional, host, count elements) = p_k(z); without it nothing is evaluated -- the constant batch_prove subtracts
 * (pcs.rs:124-130) only changes the remainder.  The division is queued on the context's stream and the call returns: d_q is
 * complete after the next synchronising call (uzk_sync, a commit, a D2H copy). */
int uzk_open_quotient_ptrs_device(const void* const* d_polys, const uint64_t* lens, uint32_t count, const uint64_t* z_mont,
                                  const uint64_t* alpha_mont, void* d_q, uint64_t q_cap, uint64_t* evals_out);

/* The quotient evaluations of t_poly on the coset k[1]*<g_m> (uzkge/src/plonk/helpers.rs:284-656 with
 * the "shuffle" feature; gate function turbo/mod.rs:193-222): for every point of the m = factor*n
 * domain, the 18 terms (gate, permutation, L1, booleanity, Anemoi round, shuffle/ECC selectors) are
 * combined with the powers of alpha and multiplied by z_h_inv[point % factor] -- the body of the
 * reference's cfg_into_iter!(0..m) loop, one lane per point.  The caller produces the inputs with the
 * coset transforms above (uzk_ntt_fr_batch_device with coset_shift = k[1]) and finishes with the
 * inverse coset transform, exactly as the reference does around that loop.
 * vec[slot]: DEVICE pointers to vectors of m elements (Montgomery form); "next" values (z, w0..w2 at
 * (point + factor) % m) are read from the same vectors. */
enum {
    UZK_TQ_W = 0,               /* 5: wire polynomials w[0..4] */
    UZK_TQ_WSEL = 5,            /* 3: wire selectors w_sel[0..2] */
    UZK_TQ_PI = 8,              /* public-input polynomial */
    UZK_TQ_Z = 9,               /* permutation grand product */
    UZK_TQ_Q = 10,              /* 9: selectors q_coset_evals (turbo/mod.rs:197-210 order) */
    UZK_TQ_S = 19,              /* 5: s_coset_evals */
    UZK_TQ_L1 = 24,
    UZK_TQ_QB = 25,
    UZK_TQ_QPRK = 26,           /* 4: q_prk1..4 */
    UZK_TQ_COSET_QUOTIENT = 30, /* prover_params.coset_quotient */
    UZK_TQ_QPK = 31,            /* 12: q_shuffle_public_key: x_00,x_01,x_10,x_11, y_.., dxy_.. */
    UZK_TQ_QG = 43,             /* 12: q_shuffle_generator, same order */
    UZK_TQ_QECC = 55,
    UZK_TQ_NVEC = 56
};
typedef struct {
    uint32_t n, factor;                 /* constraint-system size, m / n (6, or 16 for n <= 8); factor <= 16 */
    const void* vec[UZK_TQ_NVEC];
    uint64_t alpha[4], beta[4], gamma[4], k[5][4], anemoi_g[4], anemoi_g_inv[4], edwards_a[4];
    uint64_t z_h_inv[16][4];            /* 1 / (k[1]^n * g_m^(n*i) - 1), i < factor (helpers.rs:242-252) */
} uzk_quotient_args;
/* d_out: m elements on the device (must not alias an input).  A circuit built without the "shuffle" feature (zmatchmaking)
 * passes NULL in all 28 slots of UZK_TQ_WSEL, UZK_TQ_QPK, UZK_TQ_QG and UZK_TQ_QECC: terms 12..18 (helpers.rs:437-655,
 * #[cfg(feature = "shuffle")]) are then not evaluated. */
int uzk_t_quotient_device(const uzk_quotient_args* args, void* d_out, int sync);

/* ---- circuits and the five prover rounds ------------------------------------------------------------------------------------
 * prover_with_lagrange (uzkge/src/plonk/prover.rs:88-394) behind the C ABI: the per-circuit data of PlonkProverParams
 * (uzkge/src/plonk/indexer.rs:76-138) lives in HBM as a CIRCUIT, one proof's polynomials in a PROVER (a workspace), and one call
 * per Fiat-Shamir round does everything the reference does between two transcript draws.  The host keeps what is O(1) or
 * protocol: the transcript, the prng (blinds), r_poly's scalars.  Every host language -- the C++ driver tests/cpp/prover_rounds.cpp,
 * the Python driver tools/prover_chain.py, rust/uzkge-glue/gpu_prover.rs -- runs the same tested implementation.
 *
 * Slots of a circuit's polynomials = the quotient kernel's order (UZK_TQ_Q .. UZK_TQ_QECC) minus UZK_TQ_Q.
 * Coset evaluations over the quotient domain (indexer.rs:316-470: q_coset_evals, s_coset_evals, ...) are never uploaded: the
 * library derives them from the coefficient forms with one batched coset FFT (k[1], the 6n domain) -- exact arithmetic, so the
 * bytes are the indexer's, and the ordering of the 6n domain becomes internal to the library (every vector over that domain is
 * produced and consumed here: the proof does not depend on which primitive 6n-th root enumerates it). */
#define UZK_CIRCUIT_SLOTS 46
enum {
    UZK_CS_Q = 0,               /* 9: q_polys */
    UZK_CS_S = 9,               /* 5: s_polys */
    UZK_CS_L1 = 14,             /* l1_coefs */
    UZK_CS_QB = 15,             /* qb_poly */
    UZK_CS_QPRK = 16,           /* 4: q_prk_polys */
    UZK_CS_COSET_QUOTIENT = 20, /* prover_params.coset_quotient = k[1] * g_m^i: pass NULL, the library builds it (= the polynomial X) */
    UZK_CS_QPK = 21,            /* 12: q_shuffle_public_key_polys */
    UZK_CS_QG = 33,             /* 12: q_shuffle_generator_polys */
    UZK_CS_QECC = 45            /* q_ecc_poly */
};
typedef struct {
    uint32_t n;                 /* cs.size(): a power of two, 16 <= n <= 2^20 */
    uint32_t shuffle;           /* != 0: built with the "shuffle" feature -- all 46 slots; 0: slots 0..20 only (zmatchmaking) */
    uint32_t precompute;        /* window tables over the commit bases (uzk_srs_precompute): 0 none; 1 automatic -- two tables, 8-bit
                                   windows for provers of one proof (shortest chain of dependent additions) and 15-bit windows for
                                   lockstep batches (17 instead of 32 additions per scalar), 50 MiB at n = 2^14; 4 .. 24: one table
                                   of that window width for every prover */
    uint32_t reserved;
    const uzk_g1_affine* lagrange_bases;  /* n points: lagrange_pcs.public_parameter_group_1 (prover.rs:125-130) */
    const uzk_g1_affine* blind_bases;     /* 6 points: pcs.public_parameter_group_1[0..3) || [n..n+3) (apply_blind_factors) */
    const uint32_t* permutation;          /* 5 n entries < 5 n: prover_params.permutation */
    uint64_t k[5][4];                     /* verifier_params.k */
    uint64_t anemoi_g[4], anemoi_g_inv[4], edwards_a[4];
    uint64_t group_gen[4];                /* domain.group_gen of the size-n domain: must equal uzk_domain_group_gen(n), else UZK_ERR_FFT */
    const uint64_t* polys[UZK_CIRCUIT_SLOTS];   /* coefficient forms (host); slots >= 21 are ignored without shuffle; slot 20: NULL
                                                   (a polynomial given there replaces X: the frozen test vectors use a random one) */
    uint64_t poly_lens[UZK_CIRCUIT_SLOTS];      /* coefs.len() <= n (FpPolynomial::from_coefs has trimmed them); 0 = the zero polynomial */
} uzk_circuit_desc;
/* Uploads the circuit, derives the 46 (21) coset tables, registers the commit bases.  The handle is process-wide: provers of
 * every context may use it at the same time. */
int uzk_circuit_create(const uzk_circuit_desc* desc, uint64_t* circuit_out);
/* Replaces `count` polynomials from `first_slot` on (coefficient forms, host; lens[i] <= n) and re-derives their coset tables:
 * what refresh_prover_params_public_key (shuffle/src/gen_params/params.rs:57-129) does to q_shuffle_public_key_polys /
 * _coset_evals once per game -- first_slot = UZK_CS_QPK, count = 12.  Copy on write: a proof in flight (between its round 1 and
 * its round 5) keeps the tables it started with, the next round 1 sees the new ones; nothing is freed under a running kernel. */
int uzk_circuit_update_tables(uint64_t circuit, uint32_t first_slot, uint32_t count, const uint64_t* const* polys, const uint64_t* lens);
/* The per-table loop of refresh_prover_params_public_key (params.rs:88-121) and indexer_with_lagrange (indexer.rs:316-470) as
 * one device call: `count` evaluation vectors over the size-n domain (host, n elements each, evals + i * n)
 *   -> batched iFFT(n) -> batched coset FFT(6n, k[1]) -> batched Lagrange commit of the evaluations,
 * the results installed in slots first_slot .. first_slot + count (as uzk_circuit_update_tables would).  Outputs, each optional
 * (NULL): polys_out = count x n coefficients (zero padded; lens_out[i] = coefs.len() after from_coefs' trimming), coset_out =
 * count x 6n coset evaluations in the library's enumeration of the 6n domain (uzk_domain_group_gen(6 n)), commitments_out =
 * count commitments (lagrange_pcs.commit(evals), no blinds). */
int uzk_circuit_refresh_tables(uint64_t circuit, uint32_t first_slot, uint32_t count, const uint64_t* evals, uint64_t* polys_out,
                               uint64_t* lens_out, uint64_t* coset_out, uzk_g1_jac* commitments_out);
/* The same loop without a circuit -- what indexer_with_lagrange (indexer.rs:316-470) runs once per deck size for its 34 (9)
 * selector / permutation vectors: `count` evaluation vectors (host, n each) -> batched iFFT(n) -> polys_out (count x n, zero padded;
 * lens_out = coefs.len()), [coset_out: batched coset FFT over the 6n domain with shift k1: count x 6n], [commitments_out: one batched
 * MSM of the evaluations over the first n bases of `lagrange_srs` (uzk_srs_register)].  Every output is optional. */
int uzk_preprocess_tables(uint64_t lagrange_srs, uint32_t n, uint32_t count, const uint64_t* evals, const uint64_t* k1_mont, uint64_t* polys_out,
                          uint64_t* lens_out, uint64_t* coset_out, uzk_g1_jac* commitments_out);
/* Device address and coefficient count of a slot's polynomial (which = 0; n elements allocated) or coset table (which = 1; 6n
 * elements) as of now -- tests and diagnostics. */
int uzk_circuit_table(uint64_t circuit, uint32_t slot, int which, const void** d_out, uint64_t* len_out);
/* Forgets the handle.  Tables, commit bases and window tables are freed with the last proof that still holds them: a proof in
 * flight (between its round 1 and its round 5) finishes with what it started with. */
int uzk_circuit_release(uint64_t circuit);
/* n, the evaluations round 4 writes per proof (15, or 19 for a shuffle circuit), the r_poly scalars round 5 reads per proof
 * (19 or 43) and the device the circuit lives on; every output optional (NULL). */
int uzk_circuit_info(uint64_t circuit, uint32_t* n_out, uint32_t* evals_per_proof_out, uint32_t* r_scalars_per_proof_out, int* device_out);

/* A prover = the device buffers of `batch` proofs over circuits of size n that advance in lockstep (a server with several
 * witnesses of one circuit waiting runs them as ONE sequence of wider launches -- commits of 8 x batch vectors, transforms of
 * 10 x batch).  Use a prover from one thread at a time.  It lives on the calling context's device; all five rounds of a proof
 * must come from the context that ran its round 1 (else UZK_ERR_PARAMETER: they are ordered on that context's stream).
 *
 * batch = 1 is the reference's shape -- prover_with_lagrange proves one proof per call from whatever thread the application
 * runs (prover.rs:88-100; shuffle/src/sdk.rs:196-214) -- and at n = 2^14 one proof leaves most of the chip idle.  Provers of
 * one proof are therefore SHARED by default: when several threads stand at the same round of proofs over the same circuit, the
 * library runs their calls as one lockstep launch sequence on a pooled workspace with its own stream and hands each caller its
 * own outputs -- bit for bit what the call alone would have returned (tests/test_gpu_coalesce.py).  No new API: each thread
 * keeps calling uzk_prove_round1..5 on its own prover, from any context.  A thread whose prover is the only idle one never
 * waits; see uzk_coalesce_config.
 *
 * Device memory: a lane (the buffers of one proof) takes about 150 x n x 32 bytes -- 79 MB at n = 2^14, 315 MB at 2^16, 5 GB at
 * 2^20.  A prover of `batch` proofs holds `batch` lanes.  A shared prover holds one lane of its own (made on first need); the
 * pooled workspaces hold max_lanes lanes each and at most `groups` of them exist per (n, device): 8 x 4 x 79 MB = 2.5 GB at
 * n = 2^14 with the defaults.  Sharing only pays where one proof leaves the chip idle, so provers of n > UZK_SHARED_MAX_N
 * (2^16) are never shared: they own their single lane as before -- the pool cannot grow past 8 x 4 x 315 MB = 10 GB. */
#define UZK_SHARED_MAX_N (1u << 16)
int uzk_prover_create(uint32_t n, uint32_t batch, uint64_t* prover_out);
/* The same, but the prover owns its lanes whatever `batch` and the sharing configuration are: its proofs run on the calling
 * context's stream, alone, and uzk_prover_buffer can show its buffers (tests, diagnostics, latency measurements). */
int uzk_prover_create_private(uint32_t n, uint32_t batch, uint64_t* prover_out);
int uzk_prover_destroy(uint64_t prover);
/* How provers of one proof made FROM NOW ON are shared (process-wide; existing provers keep their kind):
 *   max_lanes          most proofs per lockstep launch (default 8; <= 64).  0 or 1: off -- such provers own their lane, as
 *                      provers with batch >= 2 and those of uzk_prover_create_private always do
 *   gather_wait_us     how long the first caller of a round 1 waits for company (0 = the default, 2000); it does not wait at all
 *                      unless another shared prover of its size and device is about to start a proof too (one that finished a
 *                      proof within the last 5 ms, or stands in the last round of one)
 *   straggler_wait_us  how long the callers of rounds 2..5 wait for a member of their group before its proof is moved to a
 *                      workspace of its own and the rest go on (0 = the default, 20000)
 *   groups             the provers at work are spread over this many launch sequences side by side (0 = the default, 4: measured
 *                      best at n = 2^14 -- four streams of B proofs each beat one of 4 B): a group takes at most
 *                      ceil(provers at work / groups) proofs; up to three provers keep a stream each and are never merged,
 *                      from four on nobody is left alone (four provers run as 2 + 2, five as 3 + 2, seven as 3 + 2 + 2).
 *                      At most four of the library's own streams are busy on a device whatever `groups` is: the compute front
 *                      end has four pipes (a fifth busy queue shares one: profiles/r05_gaps_lockstep_5x8.txt) */
int uzk_coalesce_config(uint32_t max_lanes, uint32_t gather_wait_us, uint32_t straggler_wait_us, uint32_t groups);
/* What sharing has done since the last uzk_coalesce_config or the last reset (out == NULL resets the counters): out[0] shared rounds run, out[1] round calls they served (out[1] /
 * out[0] = proofs per launch sequence), out[2] the most calls one round served, out[3] proofs moved to a workspace of their own
 * because their caller stayed away, out[4] groups formed at a round 1, out[5] / out[6] the average host time in microseconds
 * between the end of a group's round and the start of its next (its callers wake, take their results and come back with the next
 * challenges: the price of keeping the one-proof API), out[7] / out[4] the average time in microseconds a group stayed open before
 * its first round, out[8 + i] how many groups started with i + 1 callers (the last entry: 8 or more). */
int uzk_coalesce_stats(uint64_t out[16]);
/* Per-proof arrays below are [batch][...]: element b of every input / output belongs to proof b.
 *
 * Round 1 (prover.rs:151-192): PI polynomial, wire and wire-selector polynomials: iFFT(n), hide_polynomial, commit with blinds.
 *   witness   batch x 5 n: cs.extend_witness(w) (wire-major).          wsel  batch x 3 n: cs.compute_witness_selectors(), or NULL
 *   inputs_on_device != 0: witness / wsel are device addresses (a witness produced on the device); host memory of
 *     uzk_host_alloc is uploaded asynchronously, other host memory synchronously
 *   pi_index  pi_count constraint indices (verifier_params.public_vars_constraint_indices), shared by the batch;
 *   pi_value  batch x pi_count online values (helpers.rs:111-131; a repeated index takes its first value, as find_position does)
 *   hiding    5 (+3 with wsel) hiding degrees <= 3: cs.get_hiding_degree(i), then 2 per wire selector (prover.rs:186)
 *   blinds    batch x (5 or 8) x 3 elements: the draws of hide_polynomial in the reference's order, unused third slots zero
 *   cm_out    batch x (5 or 8) commitments: cm_w_vec, then cm_w_sel_vec
 * Takes the circuit's CURRENT tables for the whole proof. */
int uzk_prove_round1(uint64_t prover, uint64_t circuit, const void* witness, const void* wsel, int inputs_on_device,
                     const uint32_t* pi_index, const uint64_t* pi_value, uint32_t pi_count, const uint32_t* hiding,
                     const uint64_t* blinds, uzk_g1_jac* cm_out);
/* Round 2 (prover.rs:194-209): z_poly, iFFT, hide with 3 blinds, commit.  beta, gamma: batch x 4 limbs; blinds_z: batch x 3. */
int uzk_prove_round2(uint64_t prover, const uint64_t* beta, const uint64_t* gamma, const uint64_t* blinds_z, uzk_g1_jac* cm_z_out);
/* Round 3 (prover.rs:211-239): t_poly (helpers.rs:223-678) and split_t_and_commit with n_constraints + 2 (helpers.rs:1323-1408).
 * alpha: batch x 4 limbs; t_rands: batch x 5 (one draw per chunk, chunk order); cm_t_out: batch x 5.
 * from_coefs trims t and the trimmed length drives the split: the library goes on with the length a satisfied circuit gives
 * (5 n - 2 + the wires' hiding degrees), measures the real one on the device meanwhile and redoes the split if they differ.
 * UZK_ERR_COMMITMENT: t is longer than 5 (n + 2) + 1 coefficients (the witness does not satisfy the circuit; the reference
 * indexes past its n + 3 SRS powers in apply_blind_factors and aborts there) or a chunk is shorter than n. */
int uzk_prove_round3(uint64_t prover, const uint64_t* alpha, const uint64_t* t_rands, uzk_g1_jac* cm_t_out);
/* Round 4 (prover.rs:241-273): the opening evaluations, in the reference's order of computation:
 *   w_polys_eval_zeta (5), s_polys_eval_zeta (4), prk_3, prk_4, z_eval_zeta_omega, w_polys_eval_zeta_omega (3)
 *   and, for a shuffle circuit, q_ecc_poly_eval_zeta, w_sel_polys_eval_zeta (3):  15 or 19 elements per proof.
 * zeta: batch x 4 limbs; zeta * omega is formed here.  evals_cap: the elements evals_out can hold (>= batch x 15 or 19:
 * uzk_circuit_info; less is UZK_ERR_PARAMETER, nothing is written past it). */
int uzk_prove_round4(uint64_t prover, const uint64_t* zeta, uint64_t* evals_out, size_t evals_cap);
/* Round 5 (prover.rs:296-372): r(X) = sum_k r_scalars[k] * p_k over (in this order) q_polys (9), z, s_polys[4], qb, q_prk1,
 * q_prk2, [q_shuffle_public_key (12), q_shuffle_generator (12)], t chunks (5): 19 or 43 scalars per proof, which the caller gets
 * from the reference's r_poly_or_comm (helpers.rs:681-1002); then both batch_prove calls (pcs.rs:107-168) with their transcript
 * challenges alpha_zeta / alpha_zeta_omega: quotient, fold, FFT(n), Lagrange commit, blind factors.
 * r_count: the elements r_scalars holds -- exactly batch x 19 or 43 (uzk_circuit_info), else UZK_ERR_PARAMETER.
 * openings_out: batch x 2 (opening_witness_zeta, opening_witness_zeta_omega).  Ends the proof: the circuit tables are released. */
int uzk_prove_round5(uint64_t prover, const uint64_t* r_scalars, size_t r_count, const uint64_t* alpha_zeta, const uint64_t* alpha_zeta_omega,
                     uzk_g1_jac* openings_out);
/* Device address and element count (per proof: proof b's part starts b * elems_out elements in) of a prover buffer -- tests
 * (provers that own their lanes only: batch >= 2, or uzk_prover_create_private):
 *   0 evals (10 n: w0..4, w_sel0..2, pi, z)  1 coefs (10 x 6n slots, same order)  2 coset evaluations (10 x 6n)
 *   3 quotient evaluations (6n)  4 t (6n)  5 t chunks (5 x (n + 8))  6 folded (5 n)  7 tails (5 x 6)
 *   8 opening quotients (2 x (n + 8))  9 r (n + 8).
 * Without wire selectors the slot order of 0..2 is w0..4, pi, z (7 slots used). */
int uzk_prover_buffer(uint64_t prover, int which, void** d_out, uint64_t* elems_out);

/* ---- synthetic workloads (bench / tests; generated on device, nothing uploaded) -------- */
/* d_points[i] = (i + 1) * Q with Q = seed_scalar * G: n distinct valid G1 points whose discrete
 * logs relative to Q are known, so MSM(points, s) == (sum_i s_i (i+1)) * Q for any size. */
int uzk_synth_points_arith(void* d_points, size_t n, const uint64_t* seed_scalar_mont);
/* d_points[i] = hash-to-curve of a SplitMix64 counter stream (try-and-increment on x):
 * pseudo-random G1 points with unknown discrete logs. */
int uzk_synth_points_random(void* d_points, size_t n, uint64_t seed);
/* Uniform Fr elements (Montgomery form) from a SplitMix64 counter stream. */
int uzk_synth_scalars(void* d_scalars, size_t n, uint64_t seed);
/* The illustrative prover-like mix (SURVEY.md 8d set B): by a hash of the index 50 % zero, 20 % one,
 * 10 % r - 1, 10 % below 2^16, 10 % uniform. */
int uzk_synth_scalars_mix(void* d_scalars, size_t n, uint64_t seed);

/* ---- element-wise field arithmetic on host arrays (the host mirrors' helper) --------------- */
/* The device's field primitives applied element-wise to host arrays: what the host-side mirrors use for format conversion and the
 * few O(n) field operations they own (include/uzkge_poly_commit.hpp: the SRS blob's canonical coordinates -> Montgomery form,
 * apply_blind_factors' negations).  field: 0 = Fq, 1 = Fr.  op: 0 mul, 1 add, 2 sub, 4 sqr, 5 neg, 6 from_mont, 7 to_mont
 * (unary ops ignore b).  Any other op: UZK_ERR_PARAMETER -- the known-answer opcodes of the arithmetic cores live in
 * include/uzkge_gpu_test.h (uzk_test_field_kat), outside the drop-in ABI. */
int uzk_field_op_device(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

/* ---- measurement ---------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by hipEvents on the library stream. */
int uzk_profile_enable(int on);
int uzk_profile_reset(void);
/* Total elapsed ms and launch count of kernel `name` since the last reset. */
int uzk_profile_get(const char* name, double* total_ms, uint64_t* launches);
/* Newline-separated "name launches total_ms" table into buf. */
int uzk_profile_dump(char* buf, size_t cap);
/* Block until the library stream is idle. */
int uzk_sync(void);
/* The library's hipStream_t (so callers can order their own work against it). */
void* uzk_stream(void);
/* Tuning knobs (0 = automatic): MSM window bits.  Like uzk_tune, this sets the calling thread's CURRENT CONTEXT only;
 * a context created afterwards inherits its creator's settings. */
int uzk_msm_set_window_bits(int c);
/* What a general-mode MSM over n points will use: signed window width and window count (every
 * point is added into one bucket per window: n * windows mixed additions). */
int uzk_msm_plan_info(size_t n, int* window_bits, int* windows);
/* The switches of the calling thread's current context (a context made afterwards inherits them).  None changes a result.
 *   "msm_no_precompute"   1: ignore window tables (uzk_srs_precompute): the general pipeline over the plain bases -- what a host
 *                         sets when HBM is short; the A/B of table against no table
 *   "msm_stream_log"      log2 of the point chunk in which the host scalars of a large MSM are uploaded while the previous chunk
 *                         is accumulated (0 = 21); -1: never stream -- upload, then one MSM
 *   "msm_stream_min_log"  host-scalar MSMs of at least 2^this points are streamed (default 22)
 *   "msm_chunk_log"       points per sort pass of one MSM (default 26, the index space of the sort; lower = less workspace)
 *   "msm_small"           0: n <= 2^15 takes the general pipeline too (default 1: one workgroup per (vector, window))
 *   "msm_seg_sort"        0: the generic last sort pass instead of one workgroup per segment; 10 + k: segment kernel k at any size
 *   "ntt_tile"            1024 / 2048: elements per workgroup of an NTT pass at every size (default 0: by size)
 *   "ntt_two_pass"        0: transforms of 2^17 .. 2^21 elements in three passes of 5 .. 8 bits (default 1: two passes of 9 .. 11)
 *   "arith29"             bit mask, default 7: which of the prover's kernels run on the lazy 29-bit limbs (csrc/lz29.hpp) instead of
 *                         8 x 32-bit Montgomery words -- 1: the quotient kernel, 2: the lane evaluations and linear combinations,
 *                         4: the MSM's bucket-side additions (class sums, folds, the small pipeline's quads).  Same bytes either way
 * The last five exist so that the tests reach every pipeline and instantiation at sizes the CPU oracle can check
 * (tests/test_gpu_variants.py, tests/test_gpu_arith29.py).  Unknown keys are UZK_ERR_PARAMETER. */
int uzk_tune(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* UZKGE_GPU_H */
