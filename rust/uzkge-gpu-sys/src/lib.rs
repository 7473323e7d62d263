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
