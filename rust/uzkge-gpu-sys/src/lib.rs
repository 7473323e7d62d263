//! Bindings to `libuzkge_gpu.so` (C ABI: `include/uzkge_gpu.h`), the MI355X backend for the two hot paths of
//! `uzkge::plonk::prover`:
//!
//! * `G1Projective::msm(&points_raw, &coefs)`  -- `uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290`
//! * `domain.fft(&self.coefs)` / `domain.ifft(&values)` -- `uzkge/src/poly_commit/field_polynomial.rs:585,595`
//!
//! `ffi` is generated from the header (`tools/gen_rust_bindings.py`); this file adds the error mapping and a few
//! safe conveniences.  The crate knows nothing about arkworks: field elements travel as `[u64; 4]` Montgomery limbs
//! (`Fp<MontBackend<_, 4>, 4>`'s inner `BigInt<4>`), the arkworks-side repacking lives in uzkge (`src/gpu.rs` of the
//! patch next to this crate).
pub mod ffi;
pub use ffi::*;

use std::ffi::CStr;
use std::os::raw::c_int;

/// The non-OK return codes, named after the `UzkgeError` variants they map to (`uzkge/src/errors.rs:5-44`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Error {
    Parameter,
    Degree,
    Fft,
    Commitment,
    /// no gfx950 device, HIP failure, out of device memory: there is no CPU fallback
    Device,
    Unknown(c_int),
}

pub fn check(rc: c_int) -> Result<(), Error> {
    match rc {
        UZK_OK => Ok(()),
        UZK_ERR_PARAMETER => Err(Error::Parameter),
        UZK_ERR_DEGREE => Err(Error::Degree),
        UZK_ERR_FFT => Err(Error::Fft),
        UZK_ERR_COMMITMENT => Err(Error::Commitment),
        UZK_ERR_DEVICE => Err(Error::Device),
        other => Err(Error::Unknown(other)),
    }
}

/// Description of the last non-OK return on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = uzk_last_error();
        if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}

/// A device-resident SRS (the static bases of KZG commit); released on drop.
pub struct Srs {
    handle: u64,
    len: usize,
}

impl Srs {
    /// Copies `points` to HBM once; replaces the per-commit `normalize_batch` (kzg_poly_commitment.rs:287-288).
    pub fn register(points: &[uzk_g1_affine]) -> Result<Self, Error> {
        let mut handle = 0u64;
        check(unsafe { uzk_srs_register(points.as_ptr(), points.len(), &mut handle) })?;
        Ok(Srs { handle, len: points.len() })
    }
    /// Optional for a static SRS: window table in HBM (`uzk_srs_precompute`); identical results, shorter calls.
    pub fn precompute(&self, window_bits: c_int) -> Result<(), Error> {
        check(unsafe { uzk_srs_precompute(self.handle, window_bits) })
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn handle(&self) -> u64 { self.handle }

    /// sum_i scalars[i] * SRS[offset + i]  (`G1Projective::msm`, kzg_poly_commitment.rs:290).
    pub fn msm(&self, offset: usize, scalars_mont: &[[u64; 4]]) -> Result<uzk_g1_jac, Error> {
        let mut out = uzk_g1_jac::default();
        check(unsafe { uzk_msm_g1(self.handle, offset, scalars_mont.as_ptr() as *const u64, scalars_mont.len(), &mut out) })?;
        Ok(out)
    }
    /// `batch` vectors of `n` scalars each against the same bases (the prover's independent commits in one call).
    pub fn msm_batch(&self, offset: usize, scalars_mont: &[[u64; 4]], n: usize) -> Result<Vec<uzk_g1_jac>, Error> {
        if n == 0 || scalars_mont.len() % n != 0 {
            return Err(Error::Parameter);
        }
        let batch = scalars_mont.len() / n;
        let mut out = vec![uzk_g1_jac::default(); batch];
        check(unsafe { uzk_msm_g1_batch(self.handle, offset, scalars_mont.as_ptr() as *const u64, n, batch as u32, out.as_mut_ptr()) })?;
        Ok(out)
    }
}

impl Srs {
    /// The indexer's per-table loop without a circuit (`uzk_preprocess_tables`): `evals.len() / n` evaluation vectors of n elements
    /// -> (coefficient forms: n each, zero padded; their trimmed lengths; coset evaluations over the 6n domain shifted by `k1`:
    /// 6n each; the commitments over this Lagrange SRS of n bases, empty unless `want_commit`).
    #[allow(clippy::type_complexity)]
    pub fn preprocess_tables(&self, evals: &[[u64; 4]], n: usize, k1: &[u64; 4], want_commit: bool) -> Result<(Vec<[u64; 4]>, Vec<u64>, Vec<[u64; 4]>, Vec<uzk_g1_jac>), Error> {
        if n == 0 || evals.is_empty() || evals.len() % n != 0 || n > u32::MAX as usize {
            return Err(Error::Parameter);
        }
        let count = evals.len() / n;
        let mut polys = vec![[0u64; 4]; evals.len()];
        let mut lens = vec![0u64; count];
        let mut coset = vec![[0u64; 4]; 6 * evals.len()];
        let mut cms = vec![uzk_g1_jac::default(); if want_commit { count } else { 0 }];
        let cms_ptr = if want_commit { cms.as_mut_ptr() } else { std::ptr::null_mut() };
        check(unsafe {
            uzk_preprocess_tables(self.handle, n as u32, count as u32, evals.as_ptr() as *const u64, k1.as_ptr(), polys.as_mut_ptr() as *mut u64, lens.as_mut_ptr(),
                                  coset.as_mut_ptr() as *mut u64, cms_ptr)
        })?;
        Ok((polys, lens, coset, cms))
    }
}

impl Drop for Srs {
    fn drop(&mut self) {
        unsafe { uzk_srs_release(self.handle) };
    }
}

/// An SRS cut into contiguous point chunks over the GPUs of a node, driven from this one process (`uzk_srs_register_sharded`):
/// chunk i = [i n / N, (i + 1) n / N) lives on `devices[i]`; `msm` runs the chunks side by side and folds the 96-byte partial
/// sums on the host -- north_star's "MSM shards by point-chunk across the 8 GPUs of one node" behind one call.
pub struct ShardedSrs {
    handle: u64,
    len: usize,
}
impl ShardedSrs {
    /// `window_bits`: -1 no window table, 0 automatic, 4..24.
    pub fn register(points: &[uzk_g1_affine], devices: &[c_int], window_bits: c_int) -> Result<Self, Error> {
        let mut handle = 0u64;
        check(unsafe { uzk_srs_register_sharded(points.as_ptr(), points.len(), devices.as_ptr(), devices.len() as u32, window_bits, &mut handle) })?;
        Ok(ShardedSrs { handle, len: points.len() })
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    /// sum_i scalars[i] * SRS[i]
    pub fn msm(&self, scalars_mont: &[[u64; 4]]) -> Result<uzk_g1_jac, Error> {
        let mut out = uzk_g1_jac::default();
        check(unsafe { uzk_msm_g1_sharded(self.handle, scalars_mont.as_ptr() as *const u64, scalars_mont.len(), std::ptr::null_mut(), &mut out) })?;
        Ok(out)
    }
}
impl Drop for ShardedSrs {
    fn drop(&mut self) {
        unsafe { uzk_srs_release_sharded(self.handle) };
    }
}

/// In-place transform over the size-`data.len()` domain, natural order (`EvaluationDomain::{fft, ifft}`);
/// `coset_shift`: forward = pre-scale by shift^j, inverse = post-scale by shift^j (pass k^-1).
pub fn ntt(data: &mut [[u64; 4]], inverse: bool, coset_shift: Option<&[u64; 4]>) -> Result<(), Error> {
    let shift = coset_shift.map_or(std::ptr::null(), |s| s.as_ptr());
    check(unsafe { uzk_ntt_fr(data.as_mut_ptr() as *mut u64, data.len() as u64, inverse as c_int, shift) })
}

/// group_gen of the size-n domain as this library defines it (5^((r-1)/n), Montgomery limbs).
pub fn domain_group_gen(n: u64) -> Result<[u64; 4], Error> {
    let mut out = [0u64; 4];
    check(unsafe { uzk_domain_group_gen(n, out.as_mut_ptr()) })?;
    Ok(out)
}

// ---------------------------------------------------------------------------------------------------------------------
// The device-resident prover (uzkge/src/plonk/gpu_prover.rs): circuits, provers and the five rounds.  Nothing here links the
// HIP runtime, and nothing here decides anything: lengths, folds, table snapshots and error conditions live in the library.
// ---------------------------------------------------------------------------------------------------------------------
use std::os::raw::c_void;

/// One field element on the wire: 4 little-endian u64 limbs, Montgomery form.
pub type Limbs = [u64; 4];

/// A context (one stream, one set of workspaces, one lock) for a prover thread; see `uzk_ctx_create`.
pub struct Context(u64);
impl Context {
    pub fn new() -> Result<Self, Error> {
        let mut h = 0u64;
        check(unsafe { uzk_ctx_create(&mut h) })?;
        Ok(Context(h))
    }
    /// A context on a named device (`uzk_ctx_create_on`): circuits, provers and SRS handles made while it is current live there,
    /// so one process can run a pool of prover threads over all the GPUs of a node.
    pub fn on_device(device: c_int) -> Result<Self, Error> {
        let mut h = 0u64;
        check(unsafe { uzk_ctx_create_on(device, &mut h) })?;
        Ok(Context(h))
    }
    pub fn device(&self) -> Result<c_int, Error> {
        let mut d: c_int = 0;
        check(unsafe { uzk_ctx_device(self.0, &mut d) })?;
        Ok(d)
    }
    /// Makes this context the calling thread's current one.
    pub fn make_current(&self) -> Result<(), Error> {
        check(unsafe { uzk_ctx_set_current(self.0) })
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe {
            uzk_ctx_set_current(0);
            uzk_ctx_destroy(self.0);
        }
    }
}

/// A circuit resident in HBM (`uzk_circuit_create`): commit bases, permutation, the 46 (21) polynomials and their coset tables.
/// Process-wide: provers of every thread may use it at the same time.  Released on drop.
pub struct Circuit {
    handle: u64,
}
impl Circuit {
    /// `desc` and everything it points to are read before this returns.
    pub fn create(desc: &uzk_circuit_desc) -> Result<Self, Error> {
        let mut handle = 0u64;
        check(unsafe { uzk_circuit_create(desc, &mut handle) })?;
        Ok(Circuit { handle })
    }
    /// Replaces the polynomials of slots first_slot.. (coefficient forms, trimmed as `FpPolynomial::from_coefs` leaves them) and
    /// re-derives their coset tables; copy on write: a proof in flight keeps the tables it started with.
    pub fn update_tables(&self, first_slot: u32, polys: &[Vec<Limbs>]) -> Result<(), Error> {
        let ptrs: Vec<*const u64> = polys.iter().map(|p| p.as_ptr() as *const u64).collect();
        let lens: Vec<u64> = polys.iter().map(|p| p.len() as u64).collect();
        check(unsafe { uzk_circuit_update_tables(self.handle, first_slot, polys.len() as u32, ptrs.as_ptr(), lens.as_ptr()) })
    }
    /// The refresh / indexer loop as one device call (`uzk_circuit_refresh_tables`): `evals.len() / n` evaluation vectors of n
    /// elements -> (coefficient forms: n each, zero padded; their trimmed lengths; coset evaluations: 6n each, empty unless
    /// `want_coset`; the Lagrange commitments of the evaluations), installed in the slots.
    #[allow(clippy::type_complexity)]
    pub fn refresh_tables(&self, first_slot: u32, evals: &[Limbs], n: usize, want_coset: bool) -> Result<(Vec<Limbs>, Vec<u64>, Vec<Limbs>, Vec<uzk_g1_jac>), Error> {
        if n == 0 || evals.len() % n != 0 {
            return Err(Error::Parameter);
        }
        let count = evals.len() / n;
        let mut polys = vec![[0u64; 4]; evals.len()];
        let mut lens = vec![0u64; count];
        let mut coset = vec![[0u64; 4]; if want_coset { 6 * evals.len() } else { 0 }];
        let mut cms = vec![uzk_g1_jac::default(); count];
        let coset_ptr = if want_coset { coset.as_mut_ptr() as *mut u64 } else { std::ptr::null_mut() };
        check(unsafe {
            uzk_circuit_refresh_tables(self.handle, first_slot, count as u32, evals.as_ptr() as *const u64, polys.as_mut_ptr() as *mut u64, lens.as_mut_ptr(), coset_ptr,
                                       cms.as_mut_ptr())
        })?;
        Ok((polys, lens, coset, cms))
    }
    /// (n, evaluations round 4 writes per proof, r_poly scalars round 5 reads per proof, device) -- `uzk_circuit_info`.
    pub fn info(&self) -> Result<(u32, u32, u32, c_int), Error> {
        let (mut n, mut ev, mut rs, mut dev) = (0u32, 0u32, 0u32, 0 as c_int);
        check(unsafe { uzk_circuit_info(self.handle, &mut n, &mut ev, &mut rs, &mut dev) })?;
        Ok((n, ev, rs, dev))
    }
    pub fn handle(&self) -> u64 { self.handle }
}
impl Drop for Circuit {
    fn drop(&mut self) {
        unsafe { uzk_circuit_release(self.handle) };
    }
}

/// How provers of ONE proof made from now on are shared between threads (`uzk_coalesce_config`): the library runs round calls of
/// several threads that stand at the same round of proofs over the same circuit as one lockstep launch sequence.  On by default
/// (at most 8 proofs per sequence, the provers at work spread over 4 sequences, 2000 us gathering wait -- include/uzkge_gpu.h; 0 = a default);
/// `max_lanes` <= 1 switches it off.
pub fn coalesce_config(max_lanes: u32, gather_wait_us: u32, straggler_wait_us: u32, groups: u32) -> Result<(), Error> {
    check(unsafe { uzk_coalesce_config(max_lanes, gather_wait_us, straggler_wait_us, groups) })
}

/// The device buffers of `batch` proofs in lockstep over circuits of size n (`uzk_prover_create`) and the five rounds.  Per-proof
/// arrays are [batch][..]; lengths are checked here and again by the library (a short slice is a `Parameter` error, never a read
/// or write past it).  `batch` = 1 is a SHARED prover: each thread keeps its own and calls the rounds of its own proof, and the
/// library merges the calls of threads that prove over the same circuit at the same time.
pub struct Prover {
    handle: u64,
    n: usize,
    batch: usize,
}
impl Prover {
    pub fn new(n: u32, batch: u32) -> Result<Self, Error> {
        let mut handle = 0u64;
        check(unsafe { uzk_prover_create(n, batch, &mut handle) })?;
        Ok(Prover { handle, n: n as usize, batch: batch as usize })
    }
    /// A prover that owns its lanes whatever `batch` is (`uzk_prover_create_private`): its proofs run alone on the calling
    /// context's stream (latency measurements, diagnostics).
    pub fn new_private(n: u32, batch: u32) -> Result<Self, Error> {
        let mut handle = 0u64;
        check(unsafe { uzk_prover_create_private(n, batch, &mut handle) })?;
        Ok(Prover { handle, n: n as usize, batch: batch as usize })
    }
    /// prover.rs:151-192.  witness: batch x 5n; wsel: batch x 3n or empty; hiding: 5 (+3); blinds: batch x (5 | 8) x 3.
    #[allow(clippy::too_many_arguments)]
    pub fn round1(&self, circuit: &Circuit, witness: &[Limbs], wsel: &[Limbs], pi_index: &[u32], pi_value: &[Limbs], hiding: &[u32], blinds: &[Limbs]) -> Result<Vec<uzk_g1_jac>, Error> {
        let n_first = if wsel.is_empty() { 5 } else { 8 };
        if witness.len() != self.batch * 5 * self.n || (!wsel.is_empty() && wsel.len() != self.batch * 3 * self.n) || hiding.len() != n_first
            || blinds.len() != self.batch * n_first * 3 || pi_value.len() != self.batch * pi_index.len() {
            return Err(Error::Parameter);
        }
        let mut out = vec![uzk_g1_jac::default(); self.batch * n_first];
        let wsel_ptr = if wsel.is_empty() { std::ptr::null() } else { wsel.as_ptr() as *const c_void };
        check(unsafe {
            uzk_prove_round1(self.handle, circuit.handle, witness.as_ptr() as *const c_void, wsel_ptr, 0, pi_index.as_ptr(), pi_value.as_ptr() as *const u64,
                             pi_index.len() as u32, hiding.as_ptr(), blinds.as_ptr() as *const u64, out.as_mut_ptr())
        })?;
        Ok(out)
    }
    /// prover.rs:194-209 for every proof of the batch: beta, gamma: one each; blinds_z: three each; returns cm_z per proof.
    pub fn round2(&self, beta: &[Limbs], gamma: &[Limbs], blinds_z: &[Limbs]) -> Result<Vec<uzk_g1_jac>, Error> {
        if beta.len() != self.batch || gamma.len() != self.batch || blinds_z.len() != 3 * self.batch {
            return Err(Error::Parameter);
        }
        let mut out = vec![uzk_g1_jac::default(); self.batch];
        check(unsafe { uzk_prove_round2(self.handle, beta.as_ptr() as *const u64, gamma.as_ptr() as *const u64, blinds_z.as_ptr() as *const u64, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// prover.rs:211-239: alpha: one per proof; t_rands: five per proof; returns the five commitments of t's chunks per proof.
    pub fn round3(&self, alpha: &[Limbs], t_rands: &[Limbs]) -> Result<Vec<uzk_g1_jac>, Error> {
        if alpha.len() != self.batch || t_rands.len() != 5 * self.batch {
            return Err(Error::Parameter);
        }
        let mut out = vec![uzk_g1_jac::default(); 5 * self.batch];
        check(unsafe { uzk_prove_round3(self.handle, alpha.as_ptr() as *const u64, t_rands.as_ptr() as *const u64, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// prover.rs:241-273: zeta: one per proof; returns `per` evaluations per proof, packed: 19 for a circuit with the shuffle
    /// feature's terms (`shuffle`), 15 without.  The library is told how many elements the buffer holds and refuses the round
    /// (`Parameter`) when the circuit of the proof gives more.
    pub fn round4(&self, zeta: &[Limbs], shuffle: bool) -> Result<Vec<Limbs>, Error> {
        if zeta.len() != self.batch {
            return Err(Error::Parameter);
        }
        let per = if shuffle { 19 } else { 15 };
        let mut out = vec![[0u64; 4]; per * self.batch];
        check(unsafe { uzk_prove_round4(self.handle, zeta.as_ptr() as *const u64, out.as_mut_ptr() as *mut u64, out.len()) })?;
        Ok(out)
    }
    /// prover.rs:296-372: r_scalars: 19 | 43 per proof (the order of uzk_prove_round5); the two opening challenges: one each per
    /// proof; returns the two opening commitments per proof.  The library checks the count against the circuit of the proof.
    pub fn round5(&self, r_scalars: &[Limbs], alpha_zeta: &[Limbs], alpha_zeta_omega: &[Limbs]) -> Result<Vec<uzk_g1_jac>, Error> {
        if alpha_zeta.len() != self.batch || alpha_zeta_omega.len() != self.batch || (r_scalars.len() != 19 * self.batch && r_scalars.len() != 43 * self.batch) {
            return Err(Error::Parameter);
        }
        let mut out = vec![uzk_g1_jac::default(); 2 * self.batch];
        check(unsafe {
            uzk_prove_round5(self.handle, r_scalars.as_ptr() as *const u64, r_scalars.len(), alpha_zeta.as_ptr() as *const u64, alpha_zeta_omega.as_ptr() as *const u64,
                             out.as_mut_ptr())
        })?;
        Ok(out)
    }
    pub fn batch(&self) -> usize { self.batch }
}
impl Drop for Prover {
    fn drop(&mut self) {
        unsafe { uzk_prover_destroy(self.handle) };
    }
}
