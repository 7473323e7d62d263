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
        assert!(n > 0 && scalars_mont.len() % n == 0);
        let batch = scalars_mont.len() / n;
        let mut out = vec![uzk_g1_jac::default(); batch];
        check(unsafe { uzk_msm_g1_batch(self.handle, offset, scalars_mont.as_ptr() as *const u64, n, batch as u32, out.as_mut_ptr()) })?;
        Ok(out)
    }
}

impl Drop for Srs {
    fn drop(&mut self) {
        unsafe { uzk_srs_release(self.handle) };
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
// Device residency: what the device-resident prover flow (uzkge/src/plonk/gpu_prover.rs) builds on.  Nothing here links the
// HIP runtime: buffers come from uzk_dev_alloc, copies from uzk_dev_copy*.
// ---------------------------------------------------------------------------------------------------------------------
use std::os::raw::c_void;

/// One field element on the wire: 4 little-endian u64 limbs, Montgomery form.
pub type Limbs = [u64; 4];
const FR_BYTES: usize = 32;

/// `count` field elements of device memory (uzk_dev_alloc); freed on drop.
pub struct DevBuf {
    ptr: *mut c_void,
    count: usize,
}
// the pointer names device memory; the library serialises access per context
unsafe impl Send for DevBuf {}
unsafe impl Sync for DevBuf {}

impl DevBuf {
    pub fn new(count: usize) -> Result<Self, Error> {
        let mut ptr: *mut c_void = std::ptr::null_mut();
        check(unsafe { uzk_dev_alloc(count * FR_BYTES, &mut ptr) })?;
        Ok(DevBuf { ptr, count })
    }
    pub fn zeroed(count: usize) -> Result<Self, Error> {
        let b = Self::new(count)?;
        check(unsafe { uzk_dev_memset(b.ptr, 0, count * FR_BYTES) })?;
        Ok(b)
    }
    pub fn from_host(data: &[Limbs]) -> Result<Self, Error> {
        let b = Self::new(data.len())?;
        b.upload(0, data)?;
        Ok(b)
    }
    pub fn len(&self) -> usize { self.count }
    pub fn is_empty(&self) -> bool { self.count == 0 }
    /// Device address of element `elem`.
    pub fn at(&self, elem: usize) -> *mut c_void {
        assert!(elem <= self.count);
        unsafe { (self.ptr as *mut u8).add(elem * FR_BYTES) as *mut c_void }
    }
    pub fn as_ptr(&self) -> *mut c_void { self.ptr }
    /// Host -> device at element offset `elem`; `data` may be reused when this returns.
    pub fn upload(&self, elem: usize, data: &[Limbs]) -> Result<(), Error> {
        assert!(elem + data.len() <= self.count);
        check(unsafe { uzk_dev_copy(self.at(elem), data.as_ptr() as *const c_void, data.len() * FR_BYTES, UZK_COPY_H2D) })
    }
    /// Device -> host; synchronises the calling context's stream.
    pub fn download(&self, elem: usize, count: usize) -> Result<Vec<Limbs>, Error> {
        assert!(elem + count <= self.count);
        let mut out = vec![[0u64; 4]; count];
        check(unsafe { uzk_dev_copy(out.as_mut_ptr() as *mut c_void, self.at(elem), count * FR_BYTES, UZK_COPY_D2H) })?;
        Ok(out)
    }
}

impl Drop for DevBuf {
    fn drop(&mut self) {
        unsafe { uzk_dev_free(self.ptr) };
    }
}

/// Raw device bytes (the permutation: u32 indices).
pub struct DevBytes {
    ptr: *mut c_void,
}
unsafe impl Send for DevBytes {}
unsafe impl Sync for DevBytes {}
impl DevBytes {
    pub fn from_u32(data: &[u32]) -> Result<Self, Error> {
        let mut ptr: *mut c_void = std::ptr::null_mut();
        check(unsafe { uzk_dev_alloc(data.len() * 4, &mut ptr) })?;
        let b = DevBytes { ptr };
        check(unsafe { uzk_dev_copy(b.ptr, data.as_ptr() as *const c_void, data.len() * 4, UZK_COPY_H2D) })?;
        Ok(b)
    }
    pub fn as_ptr(&self) -> *mut c_void { self.ptr }
}
impl Drop for DevBytes {
    fn drop(&mut self) {
        unsafe { uzk_dev_free(self.ptr) };
    }
}

/// A context (one stream, one set of workspaces, one lock) for a prover thread; see `uzk_ctx_create`.
pub struct Context(u64);
impl Context {
    pub fn new() -> Result<Self, Error> {
        let mut h = 0u64;
        check(unsafe { uzk_ctx_create(&mut h) })?;
        Ok(Context(h))
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

fn opt_ptr(s: Option<&Limbs>) -> *const u64 {
    s.map_or(std::ptr::null(), |v| v.as_ptr())
}

/// `batch` transforms of n elements, consecutive vectors in_stride / out_stride elements apart (uzk_ntt_fr_batch_strided_device).
pub fn ntt_strided(d_in: *const c_void, in_stride: usize, d_out: *mut c_void, out_stride: usize, n: usize, batch: u32, inverse: bool,
                   coset_shift: Option<&Limbs>) -> Result<(), Error> {
    check(unsafe { uzk_ntt_fr_batch_strided_device(d_in, in_stride as u64, d_out, out_stride as u64, n as u64, batch, inverse as c_int, opt_ptr(coset_shift), 0) })
}

impl Srs {
    /// commit(evals) + apply_blind_factors in one batched MSM (uzk_msm_g1_batch_tail_device): host tail.
    pub fn commit_with_tail(&self, d_scalars: *const c_void, stride: usize, n: usize, batch: u32, tail: &[Limbs], tail_n: u32) -> Result<Vec<uzk_g1_jac>, Error> {
        assert_eq!(tail.len(), batch as usize * tail_n as usize);
        let mut out = vec![uzk_g1_jac::default(); batch as usize];
        check(unsafe { uzk_msm_g1_batch_tail_device(self.handle, 0, d_scalars, stride, n, batch, tail.as_ptr() as *const c_void, tail_n, 0, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// The same with the tail left on the device by `fold_blinds_batch`.
    pub fn commit_with_device_tail(&self, d_scalars: *const c_void, stride: usize, n: usize, batch: u32, d_tail: *const c_void, tail_n: u32) -> Result<Vec<uzk_g1_jac>, Error> {
        let mut out = vec![uzk_g1_jac::default(); batch as usize];
        check(unsafe { uzk_msm_g1_batch_tail_device(self.handle, 0, d_scalars, stride, n, batch, d_tail, tail_n, 1, out.as_mut_ptr()) })?;
        Ok(out)
    }
}

/// hide_polynomial for `count` polynomials `stride` apart (uzk_hide_polynomial_batch_device); blinds: count * hiding_degree.
pub fn hide_batch(d_coefs: *mut c_void, stride: usize, len_in: usize, count: u32, blinds: &[Limbs], hiding_degree: u32, zeroing_degree: usize) -> Result<(), Error> {
    assert_eq!(blinds.len(), (count * hiding_degree) as usize);
    check(unsafe { uzk_hide_polynomial_batch_device(d_coefs, stride as u64, len_in as u64, count, blinds.as_ptr() as *const u64, hiding_degree, zeroing_degree as u64) })
}

/// z_poly on device-resident wires / permutation / domain (uzk_z_poly_device).
pub fn z_poly(d_w: *const c_void, d_perm: *const c_void, d_group: *const c_void, k: &[Limbs], beta: &Limbs, gamma: &Limbs, n: usize, d_z: *mut c_void) -> Result<(), Error> {
    check(unsafe { uzk_z_poly_device(d_w, d_perm as *const u32, d_group, k.as_ptr() as *const u64, beta.as_ptr(), gamma.as_ptr(), n as u32, k.len() as u32, d_z) })
}

/// A few u64 words of pinned host memory (uzk_host_alloc) that the device stream writes in order: results a prover reads
/// after its next synchronising call instead of waiting for them.
pub struct PinnedWords {
    ptr: *mut u64,
    count: usize,
}
unsafe impl Send for PinnedWords {}
impl PinnedWords {
    pub fn new(count: usize) -> Result<Self, Error> {
        let mut p: *mut c_void = std::ptr::null_mut();
        check(unsafe { uzk_host_alloc(count * 8, &mut p) })?;
        Ok(PinnedWords { ptr: p as *mut u64, count })
    }
    /// Word `i`; meaningful once the stream has passed the call that writes it (after a synchronising call).
    pub fn get(&self, i: usize) -> u64 {
        assert!(i < self.count);
        unsafe { std::ptr::read_volatile(self.ptr.add(i)) }
    }
    pub fn at(&self, i: usize) -> *mut u64 {
        assert!(i <= self.count);
        unsafe { self.ptr.add(i) }
    }
}
impl Drop for PinnedWords {
    fn drop(&mut self) {
        unsafe { uzk_host_free(self.ptr as *mut c_void) };
    }
}

/// Trimmed lengths (FpPolynomial::from_coefs) of `lens.len()` device polynomials `stride` apart (uzk_poly_trimmed_len_device),
/// measured asynchronously into `out` (pinned): read them after the next synchronising call.
pub fn trimmed_len_async(d_polys: *const c_void, stride: usize, lens: &[u64], out: *mut u64) -> Result<(), Error> {
    check(unsafe { uzk_poly_trimmed_len_device(d_polys, stride as u64, lens.as_ptr(), lens.len() as u32, out, 0) })
}

/// The split of t (uzk_split_t_device); returns the chunks' coefs.len().
pub fn split_t(d_t: *const c_void, t_len: usize, chunk: usize, rands: &[Limbs], d_chunks: *mut c_void, chunk_stride: usize) -> Result<Vec<u64>, Error> {
    let mut lens = vec![0u64; rands.len()];
    check(unsafe { uzk_split_t_device(d_t, t_len as u64, chunk as u64, rands.len() as u32, rands.as_ptr() as *const u64, d_chunks, chunk_stride as u64, lens.as_mut_ptr()) })?;
    Ok(lens)
}

/// Fold modulo X^N - 1 for a batch, blinds left on the device as the next commit's tail (uzk_fold_blinds_batch_device).
pub fn fold_blinds_batch(d_polys: *const c_void, in_stride: usize, lens: &[u64], n_fold: usize, d_out: *mut c_void, out_stride: usize, d_tail: *mut c_void, tail_n: u32) -> Result<(), Error> {
    check(unsafe { uzk_fold_blinds_batch_device(d_polys, in_stride as u64, lens.as_ptr(), n_fold as u64, lens.len() as u32, d_out, out_stride as u64, d_tail, tail_n, std::ptr::null_mut()) })
}

/// out[k] = p_k(points[point_idx[k]]) (uzk_poly_eval_ptrs_device).
pub fn eval_ptrs(polys: &[(*const c_void, u64)], point_idx: &[u32], points: &[Limbs]) -> Result<Vec<Limbs>, Error> {
    assert_eq!(polys.len(), point_idx.len());
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.0).collect();
    let lens: Vec<u64> = polys.iter().map(|p| p.1).collect();
    let mut out = vec![[0u64; 4]; polys.len()];
    check(unsafe { uzk_poly_eval_ptrs_device(ptrs.as_ptr(), lens.as_ptr(), point_idx.as_ptr(), polys.len() as u32, points.as_ptr() as *const u64, points.len() as u32, out.as_mut_ptr() as *mut u64) })?;
    Ok(out)
}

/// d_out[j] = sum_k scalars[k] * p_k[j] (uzk_poly_lincomb_device).
pub fn lincomb(polys: &[(*const c_void, u64)], scalars: &[Limbs], d_out: *mut c_void, out_len: usize) -> Result<(), Error> {
    assert_eq!(polys.len(), scalars.len());
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.0).collect();
    let lens: Vec<u64> = polys.iter().map(|p| p.1).collect();
    check(unsafe { uzk_poly_lincomb_device(ptrs.as_ptr(), lens.as_ptr(), scalars.as_ptr() as *const u64, polys.len() as u32, d_out, out_len as u64) })
}

/// q = (sum_k alpha^k p_k) div (X - z) -> d_q (uzk_open_quotient_ptrs_device, no evaluations: asynchronous).
pub fn open_quotient(polys: &[(*const c_void, u64)], z: &Limbs, alpha: &Limbs, d_q: *mut c_void, q_cap: usize) -> Result<(), Error> {
    let ptrs: Vec<*const c_void> = polys.iter().map(|p| p.0).collect();
    let lens: Vec<u64> = polys.iter().map(|p| p.1).collect();
    check(unsafe { uzk_open_quotient_ptrs_device(ptrs.as_ptr(), lens.as_ptr(), polys.len() as u32, z.as_ptr(), alpha.as_ptr(), d_q, q_cap as u64, std::ptr::null_mut()) })
}

/// The quotient evaluations of t_poly (uzk_t_quotient_device), asynchronous.
pub fn t_quotient(args: &uzk_quotient_args, d_out: *mut c_void) -> Result<(), Error> {
    check(unsafe { uzk_t_quotient_device(args, d_out, 0) })
}

/// Device-to-device pitched copy (uzk_dev_copy2d).
pub fn copy2d_d2d(dst: *mut c_void, dst_pitch: usize, src: *const c_void, src_pitch: usize, width: usize, rows: usize) -> Result<(), Error> {
    check(unsafe { uzk_dev_copy2d(dst, dst_pitch, src, src_pitch, width, rows, UZK_COPY_D2D) })
}
