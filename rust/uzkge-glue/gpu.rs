//! uzkge/src/gpu.rs -- the arkworks side of the MI355X backend (cargo feature `gpu`).
//!
//! The two call sites of the hot path and the conversions the device-resident prover (`plonk/gpu_prover.rs`) shares:
//!   * `commit`  replaces `normalize_batch` + `G1Projective::msm`   (poly_commit/kzg_poly_commitment.rs:287-290)
//!   * `fft`     replaces `domain.fft(..)` / `domain.ifft(..)`      (poly_commit/field_polynomial.rs:585, 595)
//!     and fuses the serial `mul_var` of the coset pair                 (field_polynomial.rs:589-591, 601-607)
//!   * `preprocess_tables`  one indexer step's tables (iFFT, coset FFT over the quotient domain, Lagrange commit) as one
//!     device call                                                      (plonk/indexer.rs:316-470)
//!
//! Both return `None` when the device cannot serve the call (no GPU, a HIP failure, out of device memory, a field that is
//! not BN254's Fr): the caller then takes the arkworks path it always had.  Nothing here panics on a device error -- the
//! reference builds with `panic = "abort"` (Cargo.toml:69), so a lost GPU must not take the prover down.
//!
//! Layout: arkworks' `Fp<MontBackend<_, 4>, 4>` wraps `BigInt<4>([u64; 4])` holding the Montgomery representation
//! (R = 2^256) -- the wire format of include/uzkge_gpu.h -- but the structs are `repr(Rust)`, so limbs are copied
//! field by field, never transmuted.
use std::any::TypeId;
use std::collections::HashMap;
use std::sync::{Arc, Mutex};

use ark_bn254::{Fq, Fr, G1Affine, G1Projective};
use ark_ec::CurveGroup;
use ark_ff::{BigInt, PrimeField};
use ark_poly::EvaluationDomain;
use lazy_static::lazy_static;
use uzkge_gpu_sys as sys;

use crate::errors::UzkgeError;
use crate::poly_commit::{field_polynomial::FpPolynomial, pcs::PolyComScheme};

pub(crate) fn map_err(e: sys::Error) -> UzkgeError {
    match e {
        sys::Error::Degree => UzkgeError::DegreeError,
        sys::Error::Fft => UzkgeError::FFTError,
        sys::Error::Commitment => UzkgeError::CommitmentError,
        _ => UzkgeError::ParameterError,
    }
}

#[inline]
pub(crate) fn fr_limbs(s: &Fr) -> [u64; 4] {
    (s.0).0
}
#[inline]
pub(crate) fn fr_from_limbs(l: [u64; 4]) -> Fr {
    Fr::new_unchecked(BigInt(l))
}
#[inline]
fn fq_limbs(s: &Fq) -> [u64; 4] {
    (s.0).0
}
pub(crate) fn affine_to_wire(p: &G1Affine) -> sys::uzk_g1_affine {
    if p.infinity {
        sys::uzk_g1_affine::default() // infinity = (0, 0), never on y^2 = x^3 + 3
    } else {
        sys::uzk_g1_affine { x: fq_limbs(&p.x), y: fq_limbs(&p.y) }
    }
}
pub(crate) fn jac_from_wire(j: &sys::uzk_g1_jac) -> G1Projective {
    // z == 0 is the identity in arkworks' Jacobian representation too
    G1Projective::new_unchecked(Fq::new_unchecked(BigInt(j.x)), Fq::new_unchecked(BigInt(j.y)), Fq::new_unchecked(BigInt(j.z)))
}

/// `&[F]` as `&[Fr]` when F is BN254's scalar field (an identity cast: same type), else None.
pub(crate) fn as_fr_slice<F: PrimeField>(v: &[F]) -> Option<&[Fr]> {
    if TypeId::of::<F>() != TypeId::of::<Fr>() {
        return None;
    }
    Some(unsafe { &*(v as *const [F] as *const [Fr]) })
}
/// `Vec<Fr>` as `Vec<F>` under the same condition.
pub(crate) fn from_fr_vec<F: PrimeField>(v: Vec<Fr>) -> Vec<F> {
    assert_eq!(TypeId::of::<F>(), TypeId::of::<Fr>());
    let mut v = std::mem::ManuallyDrop::new(v);
    unsafe { Vec::from_raw_parts(v.as_mut_ptr() as *mut F, v.len(), v.capacity()) }
}

/// The most device-resident SRS copies kept at once (least recently used goes first).  A process normally holds two or
/// three: the monomial SRS, one Lagrange SRS per circuit size, the prover's combined commit bases.
const SRS_CACHE_CAP: usize = 8;
/// Below this many bases the window table (`uzk_srs_precompute`) shortens every commit (measured: 216 vs 248 us at 2^14);
/// above it the general pipeline without a table is as fast or faster and the table would cost W * n * 64 bytes of HBM
/// (21.43 vs 20.64 ms and 13 GiB at 2^24).
const PRECOMPUTE_MAX_LEN: usize = 1 << 15;

struct SrsEntry {
    srs: Arc<sys::Srs>,
    stamp: u64,
}
lazy_static! {
    /// Device-resident SRS per parameter vector, keyed by length and a fingerprint of three points -- not by address: a
    /// freshly loaded copy of the same parameters (`load_srs_params` allocates a new Vec every time) finds the resident one.
    static ref SRS: Mutex<(HashMap<(usize, [u64; 12]), SrsEntry>, u64)> = Mutex::new((HashMap::new(), 0));
    /// Domain sizes whose generator has been compared with the library's: true = equal.
    static ref GENERATOR_CHECKED: Mutex<HashMap<u64, bool>> = Mutex::new(HashMap::new());
}

fn fingerprint_of(public_parameter_group_1: &[G1Projective]) -> [u64; 12] {
    let len = public_parameter_group_1.len();
    let probe = G1Projective::normalize_batch(&[public_parameter_group_1[0], public_parameter_group_1[len / 2], public_parameter_group_1[len - 1]]);
    let mut fp = [0u64; 12];
    for (slot, p) in probe.iter().enumerate() {
        fp[4 * slot..4 * slot + 4].copy_from_slice(&affine_to_wire(p).x);
    }
    fp
}

/// The device-resident copy of `public_parameter_group_1`, uploaded on first use: the bases are normalised and copied once
/// per parameter set, later commits only move scalars.  The registry lock is held for the lookup only -- never across a
/// device call -- so commits from several prover threads (one context each) overlap on the GPU.
pub(crate) fn resident_srs(public_parameter_group_1: &[G1Projective]) -> Result<Arc<sys::Srs>, sys::Error> {
    let key = (public_parameter_group_1.len(), fingerprint_of(public_parameter_group_1));
    {
        let mut guard = SRS.lock().unwrap();
        let (map, clock) = &mut *guard;
        *clock += 1;
        if let Some(e) = map.get_mut(&key) {
            e.stamp = *clock;
            return Ok(e.srs.clone());
        }
    }
    // not resident: normalise and upload outside the lock (two threads may race here; the loser's copy is dropped)
    let wire: Vec<sys::uzk_g1_affine> = G1Projective::normalize_batch(public_parameter_group_1).iter().map(affine_to_wire).collect();
    let srs = sys::Srs::register(&wire)?;
    if wire.len() <= PRECOMPUTE_MAX_LEN {
        srs.precompute(0)?;
    }
    let srs = Arc::new(srs);
    let mut guard = SRS.lock().unwrap();
    let (map, clock) = &mut *guard;
    *clock += 1;
    if let Some(e) = map.get(&key) {
        return Ok(e.srs.clone());
    }
    if map.len() >= SRS_CACHE_CAP {
        // evict the least recently used entry; commits still running on it keep their Arc until they return
        if let Some(oldest) = map.iter().min_by_key(|(_, e)| e.stamp).map(|(k, _)| *k) {
            map.remove(&oldest);
        }
    }
    map.insert(key, SrsEntry { srs: srs.clone(), stamp: *clock });
    Ok(srs)
}

/// Drops the device copy of this parameter set (HBM is released when the last commit using it returns).
pub fn release_srs(public_parameter_group_1: &[G1Projective]) {
    if public_parameter_group_1.is_empty() {
        return;
    }
    let key = (public_parameter_group_1.len(), fingerprint_of(public_parameter_group_1));
    SRS.lock().unwrap().0.remove(&key);
}

/// Drops every device-resident SRS copy.
pub fn release_all_srs() {
    SRS.lock().unwrap().0.clear();
}

/// `commit` of kzg_poly_commitment.rs:278-293 for `coefs = polynomial.coefs[..=degree]` (the caller has done the
/// DegreeError check).  `Ok(None)`: the device could not serve the call, take the arkworks path.
pub fn commit(public_parameter_group_1: &[G1Projective], coefs: &[Fr]) -> Result<Option<G1Projective>, UzkgeError> {
    let srs = match resident_srs(public_parameter_group_1) {
        Ok(s) => s,
        Err(sys::Error::Device) => return Ok(None),
        Err(e) => return Err(map_err(e)),
    };
    let scalars: Vec<[u64; 4]> = coefs.iter().map(fr_limbs).collect();
    match srs.msm(0, &scalars) {
        Ok(j) => Ok(Some(jac_from_wire(&j))),
        Err(sys::Error::Device) => Ok(None),
        Err(e) => Err(map_err(e)),
    }
}

/// The prover's independent commits in one call (five wires + three selectors, prover.rs:160-192; the five chunks of
/// t, helpers.rs:1390): every polynomial zero-padded to the longest.
pub fn commit_batch(public_parameter_group_1: &[G1Projective], polys: &[&[Fr]]) -> Result<Option<Vec<G1Projective>>, UzkgeError> {
    let n = polys.iter().map(|p| p.len()).max().unwrap_or(0);
    if n == 0 {
        return Ok(Some(vec![G1Projective::default(); polys.len()]));
    }
    let mut flat = vec![[0u64; 4]; n * polys.len()];
    for (b, p) in polys.iter().enumerate() {
        for (d, c) in flat[b * n..].iter_mut().zip(p.iter()) {
            *d = fr_limbs(c);
        }
    }
    let srs = match resident_srs(public_parameter_group_1) {
        Ok(s) => s,
        Err(sys::Error::Device) => return Ok(None),
        Err(e) => return Err(map_err(e)),
    };
    match srs.msm_batch(0, &flat, n) {
        Ok(v) => Ok(Some(v.iter().map(jac_from_wire).collect())),
        Err(sys::Error::Device) => Ok(None),
        Err(e) => Err(map_err(e)),
    }
}

/// Once per domain size: is arkworks' `group_gen` the generator the library transforms over (5^((r-1)/n))?  A fork with
/// another `LARGE_SUBGROUP_ROOT_OF_UNITY` would order the 3 * 2^k coset evaluations differently from the CPU-built
/// `q_coset_evals` / `z_h_inv` tables; such a domain size simply stays on the arkworks path (`false`), no wrong proofs and
/// no panic.
pub(crate) fn same_generator(n: u64, group_gen: &Fr) -> bool {
    let mut seen = GENERATOR_CHECKED.lock().unwrap();
    if let Some(ok) = seen.get(&n) {
        return *ok;
    }
    let ok = match sys::domain_group_gen(n) {
        Ok(ours) => ours == fr_limbs(group_gen),
        Err(_) => false,
    };
    seen.insert(n, ok);
    ok
}

/// `domain.fft(coefs)` / `domain.ifft(values)` on the GPU; `None` when F is not BN254's scalar field, when this arkworks
/// fork's generator of the domain is not the library's, or when the device cannot serve the call -- the caller then takes
/// the arkworks path.  `coset_shift`: forward = p(kX) (pass k), inverse = post-scale by k^-j (pass k^-1) -- the serial
/// `mul_var` of field_polynomial.rs:470-477 runs inside the transform.
pub fn fft<F: PrimeField, E: EvaluationDomain<F>>(domain: &E, input: &[F], inverse: bool, coset_shift: Option<&F>) -> Option<Vec<F>> {
    let input: &[Fr] = as_fr_slice(input)?;
    // F is Fr: same type, so these are identity casts, not reinterpretations
    let group_gen: Fr = unsafe { *(&domain.group_gen() as *const F as *const Fr) };
    let shift: Option<[u64; 4]> = coset_shift.map(|k| fr_limbs(unsafe { &*(k as *const F as *const Fr) }));
    let n = domain.size();
    if input.len() > n || !same_generator(n as u64, &group_gen) {
        return None;
    }
    let mut buf = vec![[0u64; 4]; n]; // the caller owns the Vec: short inputs are zero-padded here
    for (d, c) in buf.iter_mut().zip(input.iter()) {
        *d = fr_limbs(c);
    }
    if sys::ntt(&mut buf, inverse, shift.as_ref()).is_err() {
        return None; // UZK_ERR_DEVICE (or an unsupported size): arkworks computes the same vector
    }
    Some(from_fr_vec(buf.into_iter().map(fr_from_limbs).collect()))
}

/// What the indexer keeps of one table: its polynomial, its coset evaluations over the quotient domain, its commitment.
pub type TableTriple<PCS> = (FpPolynomial<<PCS as PolyComScheme>::Field>, Vec<<PCS as PolyComScheme>::Field>, <PCS as PolyComScheme>::Commitment);

/// The per-table loop of one indexer step (plonk/indexer.rs:316-470: `ifft_with_domain`, `coset_fft_with_domain(&domain_m, &k[1])`,
/// the Lagrange branch of the `commit` closure) for `evals.len()` tables as ONE device call (`uzk_preprocess_tables`), in order.
/// `commit == false` (verifier parameters were handed in): the commitments are `Default`, as the reference leaves them.
/// `None` = not BN254 KZG with a Lagrange SRS of this size, other roots of unity than the library's, or no device: the caller's
/// loop runs on the CPU.
pub fn preprocess_tables<PCS: PolyComScheme>(
    lagrange_pcs: Option<&PCS>, root: &PCS::Field, root_m: &PCS::Field, k1: &PCS::Field, evals: &[&[PCS::Field]], commit: bool,
) -> Option<Vec<TableTriple<PCS>>> {
    let lagrange = &lagrange_pcs?.as_kzg_bn254()?.public_parameter_group_1;
    let n = lagrange.len();
    if n < 2 || evals.is_empty() || evals.iter().any(|e| e.len() != n) {
        return None;
    }
    let scalars = as_fr_slice(&[*root, *root_m, *k1])?.to_vec();
    if !same_generator(n as u64, &scalars[0]) || !same_generator(6 * n as u64, &scalars[1]) {
        return None; // the coefficient forms and coset evaluations come back in the library's enumeration of the two domains
    }
    let srs = resident_srs(lagrange).ok()?;
    let mut flat: Vec<[u64; 4]> = Vec::with_capacity(evals.len() * n);
    for e in evals {
        flat.extend(as_fr_slice(e)?.iter().map(fr_limbs));
    }
    let (polys, lens, coset, cms) = srs.preprocess_tables(&flat, n, &fr_limbs(&scalars[2]), commit).ok()?;
    let field_of = |l: &[[u64; 4]]| from_fr_vec::<PCS::Field>(l.iter().map(|x| fr_from_limbs(*x)).collect());
    let mut out = Vec::with_capacity(evals.len());
    for t in 0..evals.len() {
        let cm = if commit { PCS::commitment_from_g1(jac_from_wire(cms.get(t)?))? } else { PCS::Commitment::default() };
        let len = (*lens.get(t)? as usize).min(n);
        out.push((FpPolynomial::from_coefs(field_of(&polys[t * n..t * n + len])), field_of(&coset[t * 6 * n..(t + 1) * 6 * n]), cm));
    }
    Some(out)
}
