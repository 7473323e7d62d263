//! uzkge/src/gpu.rs -- the arkworks side of the MI355X backend (cargo feature `gpu`).
//!
//! Everything the two call sites need and nothing else:
//!   * `commit`  replaces `normalize_batch` + `G1Projective::msm`   (poly_commit/kzg_poly_commitment.rs:287-290)
//!   * `fft`     replaces `domain.fft(..)` / `domain.ifft(..)`      (poly_commit/field_polynomial.rs:585, 595)
//!     and fuses the serial `mul_var` of the coset pair                 (field_polynomial.rs:589-591, 601-607)
//!
//! Layout: arkworks' `Fp<MontBackend<_, 4>, 4>` wraps `BigInt<4>([u64; 4])` holding the Montgomery representation
//! (R = 2^256) -- the wire format of include/uzkge_gpu.h -- but the structs are `repr(Rust)`, so limbs are copied
//! field by field, never transmuted.
use std::any::TypeId;
use std::collections::{HashMap, HashSet};
use std::sync::Mutex;

use ark_bn254::{Fq, Fr, G1Affine, G1Projective};
use ark_ec::CurveGroup;
use ark_ff::{BigInt, PrimeField};
use ark_poly::EvaluationDomain;
use lazy_static::lazy_static;
use uzkge_gpu_sys as sys;

use crate::errors::UzkgeError;

fn map_err(e: sys::Error) -> UzkgeError {
    match e {
        sys::Error::Degree => UzkgeError::DegreeError,
        sys::Error::Fft => UzkgeError::FFTError,
        sys::Error::Commitment => UzkgeError::CommitmentError,
        _ => UzkgeError::ParameterError,
    }
}

#[inline]
fn fr_limbs(s: &Fr) -> [u64; 4] {
    (s.0).0
}
#[inline]
fn fr_from_limbs(l: [u64; 4]) -> Fr {
    Fr::new_unchecked(BigInt(l))
}
#[inline]
fn fq_limbs(s: &Fq) -> [u64; 4] {
    (s.0).0
}
fn affine_to_wire(p: &G1Affine) -> sys::uzk_g1_affine {
    if p.infinity {
        sys::uzk_g1_affine::default() // infinity = (0, 0), never on y^2 = x^3 + 3
    } else {
        sys::uzk_g1_affine { x: fq_limbs(&p.x), y: fq_limbs(&p.y) }
    }
}
fn jac_from_wire(j: &sys::uzk_g1_jac) -> G1Projective {
    // z == 0 is the identity in arkworks' Jacobian representation too
    G1Projective::new_unchecked(Fq::new_unchecked(BigInt(j.x)), Fq::new_unchecked(BigInt(j.y)), Fq::new_unchecked(BigInt(j.z)))
}

lazy_static! {
    /// Device-resident SRS per parameter vector: keyed by address, length and a fingerprint of three points, so a
    /// reallocated vector at the same address does not alias a released one.
    static ref SRS: Mutex<HashMap<(usize, usize, [u64; 12]), sys::Srs>> = Mutex::new(HashMap::new());
    /// Domain sizes whose generator has been compared with the library's.
    static ref GENERATOR_CHECKED: Mutex<HashSet<u64>> = Mutex::new(HashSet::new());
}

fn fingerprint(wire: &[sys::uzk_g1_affine]) -> [u64; 12] {
    let mut f = [0u64; 12];
    for (slot, idx) in [0usize, wire.len() / 2, wire.len() - 1].iter().enumerate() {
        f[4 * slot..4 * slot + 4].copy_from_slice(&wire[*idx].x);
    }
    f
}

/// Runs `f` on the device-resident copy of `public_parameter_group_1`, uploading it first if this is the first use:
/// the bases are normalised and copied once per parameter vector, later commits only move scalars.  A static SRS also
/// gets the window table (`uzk_srs_precompute`): same commitments, shorter calls.
fn with_srs<T>(public_parameter_group_1: &[G1Projective], f: impl FnOnce(&sys::Srs) -> Result<T, sys::Error>) -> Result<T, UzkgeError> {
    let len = public_parameter_group_1.len();
    let probe = G1Projective::normalize_batch(&[public_parameter_group_1[0], public_parameter_group_1[len / 2], public_parameter_group_1[len - 1]]);
    let mut fp = [0u64; 12];
    for (slot, p) in probe.iter().enumerate() {
        fp[4 * slot..4 * slot + 4].copy_from_slice(&affine_to_wire(p).x);
    }
    let key = (public_parameter_group_1.as_ptr() as usize, len, fp);
    let mut cache = SRS.lock().unwrap();
    if !cache.contains_key(&key) {
        let wire: Vec<sys::uzk_g1_affine> = G1Projective::normalize_batch(public_parameter_group_1).iter().map(affine_to_wire).collect();
        debug_assert_eq!(fingerprint(&wire), fp);
        let srs = sys::Srs::register(&wire).map_err(map_err)?;
        srs.precompute(0).map_err(map_err)?;
        cache.insert(key, srs);
    }
    f(&cache[&key]).map_err(map_err)
}

/// `commit` of kzg_poly_commitment.rs:278-293 for `coefs = polynomial.coefs[..=degree]` (the caller has done the
/// DegreeError check).
pub fn commit(public_parameter_group_1: &[G1Projective], coefs: &[Fr]) -> Result<G1Projective, UzkgeError> {
    let scalars: Vec<[u64; 4]> = coefs.iter().map(fr_limbs).collect();
    with_srs(public_parameter_group_1, |srs| srs.msm(0, &scalars)).map(|j| jac_from_wire(&j))
}

/// The prover's independent commits in one call (five wires + three selectors, prover.rs:160-192; the five chunks of
/// t, helpers.rs:1390): every polynomial zero-padded to the longest.
pub fn commit_batch(public_parameter_group_1: &[G1Projective], polys: &[&[Fr]]) -> Result<Vec<G1Projective>, UzkgeError> {
    let n = polys.iter().map(|p| p.len()).max().unwrap_or(0);
    if n == 0 {
        return Ok(vec![G1Projective::default(); polys.len()]);
    }
    let mut flat = vec![[0u64; 4]; n * polys.len()];
    for (b, p) in polys.iter().enumerate() {
        for (d, c) in flat[b * n..].iter_mut().zip(p.iter()) {
            *d = fr_limbs(c);
        }
    }
    with_srs(public_parameter_group_1, |srs| srs.msm_batch(0, &flat, n)).map(|v| v.iter().map(jac_from_wire).collect())
}

/// Once per domain size: arkworks' `group_gen` must be the generator the library transforms over (5^((r-1)/n)).
/// A fork with another `LARGE_SUBGROUP_ROOT_OF_UNITY` would otherwise order the 3 * 2^k coset evaluations
/// differently from the CPU-built `q_coset_evals` / `z_h_inv` tables and yield invalid proofs silently.
fn assert_same_generator(n: u64, group_gen: &Fr) {
    let mut seen = GENERATOR_CHECKED.lock().unwrap();
    if seen.contains(&n) {
        return;
    }
    let ours = sys::domain_group_gen(n).expect("uzk_domain_group_gen");
    assert_eq!(
        fr_limbs(group_gen),
        ours,
        "uzkge-gpu: the size-{} evaluation domain of this arkworks fork uses another generator than libuzkge_gpu.so",
        n
    );
    seen.insert(n);
}

/// `domain.fft(coefs)` / `domain.ifft(values)` on the GPU, `None` when F is not BN254's scalar field (the caller then
/// takes the arkworks path).  `coset_shift`: forward = p(kX) (pass k), inverse = post-scale by k^-j (pass k^-1) --
/// the serial `mul_var` of field_polynomial.rs:470-477 runs inside the transform.
pub fn fft<F: PrimeField, E: EvaluationDomain<F>>(domain: &E, input: &[F], inverse: bool, coset_shift: Option<&F>) -> Option<Vec<F>> {
    if TypeId::of::<F>() != TypeId::of::<Fr>() {
        return None;
    }
    // F is Fr: same type, so these are identity casts, not reinterpretations
    let input: &[Fr] = unsafe { &*(input as *const [F] as *const [Fr]) };
    let group_gen: Fr = unsafe { *(&domain.group_gen() as *const F as *const Fr) };
    let shift: Option<[u64; 4]> = coset_shift.map(|k| fr_limbs(unsafe { &*(k as *const F as *const Fr) }));
    let n = domain.size();
    assert!(input.len() <= n);
    assert_same_generator(n as u64, &group_gen);
    let mut buf = vec![[0u64; 4]; n]; // the caller owns the Vec: short inputs are zero-padded here
    for (d, c) in buf.iter_mut().zip(input.iter()) {
        *d = fr_limbs(c);
    }
    if let Err(e) = sys::ntt(&mut buf, inverse, shift.as_ref()) {
        panic!("uzk_ntt_fr: {:?}: {}", e, sys::last_error());
    }
    let out: Vec<Fr> = buf.into_iter().map(fr_from_limbs).collect();
    // identity cast back to Vec<F>
    let mut out = std::mem::ManuallyDrop::new(out);
    Some(unsafe { Vec::from_raw_parts(out.as_mut_ptr() as *mut F, out.len(), out.capacity()) })
}
