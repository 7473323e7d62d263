//! uzkge/src/plonk/gpu_prover.rs -- `prover_with_lagrange` (prover.rs:88-394) with every polynomial of the proof resident on
//! the MI355X (cargo feature `gpu`).
//!
//! The CPU prover keeps its polynomials in `Vec<Fr>` between steps; here they stay in HBM and only challenges, blinds,
//! evaluations and commitments cross PCIe.  The steps, their order, the transcript and the prng draws are the reference's
//! (same proof bytes); the device calls are those of tests/cpp/prover_rounds.cpp in the backend's repository, one for one:
//!
//!   round 1   uzk_ntt_fr_batch_strided_device (iFFT x8 into 6n-slots), uzk_hide_polynomial_batch_device,
//!             uzk_msm_g1_batch_tail_device (8 commits + blind factors)                                   prover.rs:151-192
//!   round 2   uzk_z_poly_device, iFFT, hide, commit                                                      prover.rs:199-209
//!   round 3   uzk_ntt_fr_batch_device (coset FFT x10), uzk_t_quotient_device, coset iFFT                 helpers.rs:223-678
//!             uzk_poly_trimmed_len_device (asynchronous), uzk_split_t_device, uzk_fold_blinds_batch_device, FFT(n) x5,
//!             uzk_msm_g1_batch_tail_device (device tail)                                                 helpers.rs:1323-1408
//!   round 4   uzk_poly_eval_ptrs_device (15 + 4 evaluations, one launch)                                 prover.rs:246-273
//!   round 5   uzk_poly_lincomb_device (r_poly: the scalars come from the reference's own r_poly_or_comm,
//!             run on a symbolic element), uzk_open_quotient_ptrs_device x2, trimmed lengths, fold, FFT(n) x2,
//!             commit                                                                            helpers.rs:681-1090, pcs.rs:107-168
//!
//! `prove` returns `Ok(None)` -- and the caller continues on the CPU path, nothing consumed -- when the scheme is not BN254
//! KZG with a Lagrange SRS of the circuit's size, or when the device cannot hold the circuit.  Once the transcript has
//! been touched there is no way back: a device failure from then on is `Err(UzkgeError::ProofError)`, never a panic (the
//! reference builds with `panic = "abort"`).
use std::cell::RefCell;
use std::collections::HashMap;
use std::os::raw::c_void;
use std::sync::{Arc, Mutex};

use ark_bn254::{Fr, G1Projective};
use ark_ff::{batch_inversion, Field, One, PrimeField, UniformRand, Zero};
use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
use ark_std::rand::{CryptoRng, RngCore};
use lazy_static::lazy_static;
use serde::{Deserialize, Serialize};
use uzkge_gpu_sys as sys;

use super::{
    constraint_system::ConstraintSystem,
    helpers::{first_lagrange_poly, pi_poly, r_poly_or_comm, PlonkChallenges},
    indexer::{PlonkProof, PlonkProverParams},
};
use crate::{
    errors::UzkgeError,
    gpu::{as_fr_slice, fr_limbs, from_fr_vec, fr_from_limbs, jac_from_wire, resident_srs, same_generator},
    poly_commit::{
        field_polynomial::FpPolynomial,
        kzg_poly_commitment::KZGCommitmentSchemeBN254,
        pcs::{HomomorphicPolyComElem, PolyComScheme, ToBytes},
    },
    utils::transcript::Transcript,
};

type Limbs = [u64; 4];
const N_WIRES: usize = 5;
#[cfg(feature = "shuffle")]
const N_WSEL: usize = 3;
#[cfg(not(feature = "shuffle"))]
const N_WSEL: usize = 0;
const N_PROOF_POLYS: usize = 10; // slots of d_coefs / d_coset: w0..4, w_sel0..2, pi, z  (UZK_TQ_W .. UZK_TQ_Z)
const N_TABLES: usize = 46; //       slots UZK_TQ_Q .. UZK_TQ_QECC of the quotient kernel, minus UZK_TQ_Q
const T_Q: usize = 0;
const T_S: usize = 9;
const T_L1: usize = 14;
const T_QB: usize = 15;
const T_QPRK: usize = 16;
const T_CQ: usize = 20;
#[cfg(feature = "shuffle")]
const T_QPK: usize = 21;
#[cfg(feature = "shuffle")]
const T_QG: usize = 33;
#[cfg(feature = "shuffle")]
const T_QECC: usize = 45;

fn dev<T>(r: Result<T, sys::Error>) -> Result<T, UzkgeError> {
    r.map_err(|_| UzkgeError::ProofError)
}
fn limbs_of<F: PrimeField>(v: &[F]) -> Vec<Limbs> {
    as_fr_slice(v).expect("checked by prove(): the field is BN254 Fr").iter().map(fr_limbs).collect()
}
fn limb_of<F: PrimeField>(v: &F) -> Limbs {
    limbs_of(std::slice::from_ref(v))[0]
}
fn field_of<F: PrimeField>(l: Limbs) -> F {
    from_fr_vec::<F>(vec![fr_from_limbs(l)])[0]
}

/// What one circuit keeps in HBM for all its proofs: the commit bases, the circuit's polynomials in coefficient form (for
/// the evaluations, r(X) and the openings) and on the quotient coset (for the quotient kernel), permutation and domain.
struct Circuit {
    n: usize,
    /// lagrange[0..n) || pcs[0..3) || pcs[n..n+3): commit(evals) + apply_blind_factors is one MSM with a 6-element tail
    bases: Arc<sys::Srs>,
    d_tpolys: sys::DevBuf, // N_TABLES x n, zero padded
    tlen: Vec<u64>,        // coefs.len() of each (FpPolynomial::from_coefs trims)
    d_tables: sys::DevBuf, // N_TABLES x 6n
    d_perm: sys::DevBytes,
    d_group: sys::DevBuf,
}

/// The polynomials of ONE proof (per thread, reused by the next proof of the same size).
struct Workspace {
    n: usize,
    d_evals: sys::DevBuf,  // 8n: extended witness (5n), witness selectors (3n)
    d_coefs: sys::DevBuf,  // 10 x 6n, zero beyond n + 3
    d_coset: sys::DevBuf,  // 10 x 6n
    d_tq: sys::DevBuf,
    d_t: sys::DevBuf,
    d_z: sys::DevBuf,
    d_chunks: sys::DevBuf, // 5 x (n + 8)
    d_fold: sys::DevBuf,   // 5 x n
    d_tail: sys::DevBuf,   // 5 x 6
    d_q: sys::DevBuf,      // 2 x (n + 8)
    d_r: sys::DevBuf,
    h_lens: sys::PinnedWords, // measured trimmed lengths: [0] t, [1], [2] the opening quotients
}

lazy_static! {
    static ref CIRCUITS: Mutex<HashMap<(usize, usize, [u64; 8]), Arc<Circuit>>> = Mutex::new(HashMap::new());
}
thread_local! {
    static WORKSPACE: RefCell<Option<Workspace>> = RefCell::new(None);
}

impl Workspace {
    fn new(n: usize) -> Result<Self, sys::Error> {
        let (m, cs) = (6 * n, n + 8);
        Ok(Workspace {
            n,
            d_evals: sys::DevBuf::new(8 * n)?,
            d_coefs: sys::DevBuf::zeroed(N_PROOF_POLYS * m)?,
            d_coset: sys::DevBuf::zeroed(N_PROOF_POLYS * m)?,
            d_tq: sys::DevBuf::new(m)?,
            d_t: sys::DevBuf::new(m)?,
            d_z: sys::DevBuf::new(n)?,
            d_chunks: sys::DevBuf::new(5 * cs)?,
            d_fold: sys::DevBuf::new(5 * n)?,
            d_tail: sys::DevBuf::new(5 * 6)?,
            d_q: sys::DevBuf::new(2 * cs)?,
            d_r: sys::DevBuf::new(cs)?,
            h_lens: sys::PinnedWords::new(4)?,
        })
    }
}

impl Circuit {
    fn build<PCS: PolyComScheme>(pcs: &KZGCommitmentSchemeBN254, lagrange: &KZGCommitmentSchemeBN254, p: &PlonkProverParams<PCS>, n: usize) -> Result<Self, sys::Error> {
        let m = 6 * n;
        // commit bases: the Lagrange SRS followed by the six monomial powers apply_blind_factors touches (kzg_poly_commitment.rs:299-313)
        let mut g1: Vec<G1Projective> = lagrange.public_parameter_group_1[..n].to_vec();
        g1.extend_from_slice(&pcs.public_parameter_group_1[0..3]);
        g1.extend_from_slice(&pcs.public_parameter_group_1[n..n + 3]);
        let bases = resident_srs(&g1)?;
        // the circuit's polynomials, slot order = quotient-kernel order
        let mut polys: Vec<Option<&FpPolynomial<PCS::Field>>> = vec![None; N_TABLES];
        let mut evals: Vec<Option<&[PCS::Field]>> = vec![None; N_TABLES];
        for i in 0..9 {
            polys[T_Q + i] = Some(&p.q_polys[i]);
            evals[T_Q + i] = Some(&p.q_coset_evals[i]);
        }
        for i in 0..N_WIRES {
            polys[T_S + i] = Some(&p.s_polys[i]);
            evals[T_S + i] = Some(&p.s_coset_evals[i]);
        }
        polys[T_L1] = Some(&p.l1_coefs);
        evals[T_L1] = Some(&p.l1_coset_evals);
        polys[T_QB] = Some(&p.qb_poly);
        evals[T_QB] = Some(&p.qb_coset_eval);
        for i in 0..4 {
            polys[T_QPRK + i] = Some(&p.q_prk_polys[i]);
            evals[T_QPRK + i] = Some(&p.q_prk_coset_evals[i]);
        }
        evals[T_CQ] = Some(&p.coset_quotient);
        #[cfg(feature = "shuffle")]
        {
            for i in 0..12 {
                polys[T_QPK + i] = Some(&p.q_shuffle_public_key_polys[i]);
                evals[T_QPK + i] = Some(&p.q_shuffle_public_key_coset_evals[i]);
                polys[T_QG + i] = Some(&p.q_shuffle_generator_polys[i]);
                evals[T_QG + i] = Some(&p.q_shuffle_generator_coset_evals[i]);
            }
            polys[T_QECC] = Some(&p.q_ecc_poly);
            evals[T_QECC] = Some(&p.q_ecc_coset_eval);
        }
        let d_tpolys = sys::DevBuf::zeroed(N_TABLES * n)?;
        let d_tables = sys::DevBuf::zeroed(N_TABLES * m)?;
        let mut tlen = vec![0u64; N_TABLES];
        for slot in 0..N_TABLES {
            if let Some(poly) = polys[slot] {
                let c = poly.get_coefs_ref();
                assert!(c.len() <= n);
                tlen[slot] = c.len() as u64;
                d_tpolys.upload(slot * n, &limbs_of(c))?;
            }
            if let Some(e) = evals[slot] {
                assert_eq!(e.len(), m);
                d_tables.upload(slot * m, &limbs_of(e))?;
            }
        }
        let perm: Vec<u32> = p.permutation.iter().map(|v| *v as u32).collect();
        Ok(Circuit {
            n,
            bases,
            d_tpolys,
            tlen,
            d_tables,
            d_perm: sys::DevBytes::from_u32(&perm)?,
            d_group: sys::DevBuf::from_host(&limbs_of(&p.group))?,
        })
    }

    fn tpoly(&self, slot: usize) -> (*const c_void, u64) {
        (self.d_tpolys.at(slot * self.n) as *const c_void, self.tlen[slot])
    }
}

fn circuit_for<PCS: PolyComScheme>(pcs: &KZGCommitmentSchemeBN254, lagrange: &KZGCommitmentSchemeBN254, p: &PlonkProverParams<PCS>, n: usize) -> Result<Arc<Circuit>, sys::Error> {
    // identity of a circuit: where its parameters live, its size, and a few of its selector coefficients
    let mut fp = [0u64; 8];
    let c0 = limbs_of(&p.q_polys[0].get_coefs_ref()[..p.q_polys[0].get_coefs_ref().len().min(1)]);
    let s0 = limbs_of(&p.s_polys[0].get_coefs_ref()[..1]);
    if let Some(c) = c0.first() {
        fp[..4].copy_from_slice(c);
    }
    fp[4..].copy_from_slice(&s0[0]);
    let key = (p as *const PlonkProverParams<PCS> as usize, n, fp);
    if let Some(c) = CIRCUITS.lock().unwrap().get(&key) {
        return Ok(c.clone());
    }
    let c = Arc::new(Circuit::build(pcs, lagrange, p, n)?); // built outside the lock: other circuits' proofs go on
    let mut map = CIRCUITS.lock().unwrap();
    if map.len() >= 4 {
        map.clear(); // a handful of circuits per process (one per deck size); beyond that start over
    }
    Ok(map.entry(key).or_insert(c).clone())
}

/// Drops every circuit's device residency (HBM is released when the proofs in flight return).
pub fn release_circuits() {
    CIRCUITS.lock().unwrap().clear();
}

// ---------------------------------------------------------------------------------------------------------------------
// r_poly's scalars without restating its formulas: the reference's r_poly_or_comm (helpers.rs:681-999) is generic over
// "polynomial or commitment"; run on a third kind of element -- a formal linear combination of its input polynomials -- it
// returns exactly the scalar each polynomial is multiplied by.
// ---------------------------------------------------------------------------------------------------------------------
#[derive(Clone, Default, Serialize, Deserialize)]
#[serde(bound = "")]
struct Symbolic<F: PrimeField> {
    #[serde(skip)]
    c: Vec<F>, // coefficient of input polynomial k
}
impl<F: PrimeField> Symbolic<F> {
    fn basis(k: usize) -> Self {
        let mut c = vec![F::zero(); k + 1];
        c[k] = F::one();
        Symbolic { c }
    }
}
impl<F: PrimeField> ToBytes for Symbolic<F> {
    fn to_bytes(&self) -> Vec<u8> {
        Vec::new()
    }
    fn to_transcript_bytes(&self) -> Vec<u8> {
        Vec::new()
    }
}
impl<F: PrimeField> HomomorphicPolyComElem for Symbolic<F> {
    type Scalar = F;
    fn get_base() -> Self {
        Self::default()
    }
    fn get_identity() -> Self {
        Self::default()
    }
    fn add(&self, other: &Self) -> Self {
        let mut r = self.clone();
        r.add_assign(other);
        r
    }
    fn add_assign(&mut self, other: &Self) {
        if self.c.len() < other.c.len() {
            self.c.resize(other.c.len(), F::zero());
        }
        for (a, b) in self.c.iter_mut().zip(other.c.iter()) {
            *a += b;
        }
    }
    fn sub(&self, other: &Self) -> Self {
        let mut r = self.clone();
        r.sub_assign(other);
        r
    }
    fn sub_assign(&mut self, other: &Self) {
        if self.c.len() < other.c.len() {
            self.c.resize(other.c.len(), F::zero());
        }
        for (a, b) in self.c.iter_mut().zip(other.c.iter()) {
            *a -= b;
        }
    }
    fn mul(&self, scalar: &F) -> Self {
        Symbolic { c: self.c.iter().map(|a| *a * scalar).collect() }
    }
    fn mul_assign(&mut self, scalar: &F) {
        for a in self.c.iter_mut() {
            *a *= scalar;
        }
    }
}

/// blinds || -blinds, three slots each: the tail of a commit (apply_blind_factors, kzg_poly_commitment.rs:299-313)
fn tail_of(blinds: &[Fr]) -> [Limbs; 6] {
    let mut t = [[0u64; 4]; 6];
    for (i, b) in blinds.iter().enumerate() {
        t[i] = fr_limbs(b);
        t[3 + i] = fr_limbs(&-*b);
    }
    t
}

/// The largest power of two <= degree, as the reference computes it (pcs.rs:139-145, helpers.rs:1367-1373).
fn max_power_of_2(degree: usize) -> usize {
    let mut max_power_of_2 = degree;
    for i in (0..=degree).rev() {
        if (i & i.wrapping_sub(1)) == 0 {
            max_power_of_2 = i;
            break;
        }
    }
    max_power_of_2
}

/// The Lagrange branch of batch_prove / split_t_and_commit for polynomials whose max_power_of_2 is NOT the circuit size
/// (never the case for a well-formed proof; kept so that whatever the reference does for odd lengths, this does too):
/// one polynomial at a time, the blind factors applied on the host by the reference's own apply_blind_factors.
fn commit_folded_generic<PCS: PolyComScheme>(pcs: &PCS, circuit: &Circuit, ws: &Workspace, d_poly: *const c_void, len: usize, degree: usize) -> Result<PCS::Commitment, UzkgeError> {
    let npow = max_power_of_2(degree);
    if npow == 0 || npow > circuit.n || !npow.is_power_of_two() {
        return Err(UzkgeError::FFTError);
    }
    let mut blinds_l = vec![[0u64; 4]; len.saturating_sub(npow).max(1)];
    sys::check(unsafe { sys::uzk_fold_blinds_device(d_poly, len as u64, npow as u64, ws.d_fold.as_ptr(), blinds_l.as_mut_ptr() as *mut u64) }).map_err(|_| UzkgeError::ProofError)?;
    blinds_l.truncate(len.saturating_sub(npow));
    dev(sys::ntt_strided(ws.d_fold.as_ptr(), npow, ws.d_fold.as_ptr(), npow, npow, 1, false, None))?;
    let cm = dev(circuit.bases.commit_with_tail(ws.d_fold.as_ptr(), npow, npow, 1, &[], 0))?;
    let cm = PCS::commitment_from_g1(jac_from_wire(&cm[0])).ok_or(UzkgeError::ProofError)?;
    let blinds: Vec<PCS::Field> = blinds_l.into_iter().map(field_of::<PCS::Field>).collect();
    Ok(pcs.apply_blind_factors(&cm, &blinds, npow))
}

/// The device-resident body of `prover_with_lagrange`, from "1. Build the PI polynomial" to the returned proof
/// (prover.rs:151-393).  The caller has initialised the transcript and selected `lagrange_pcs` (prover.rs:101-130).
#[allow(clippy::too_many_arguments)]
pub(super) fn prove<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prng: &mut R,
    transcript: &mut Transcript,
    pcs: &PCS,
    lagrange_pcs: Option<&PCS>,
    cs: &CS,
    prover_params: &PlonkProverParams<PCS>,
    w: &[PCS::Field],
    domain: &Radix2EvaluationDomain<PCS::Field>,
    online_values: &[PCS::Field],
) -> Result<Option<PlonkProof<PCS>>, UzkgeError> {
    // ---- is this a proof the device flow covers?  (nothing is consumed before the answer is yes)
    let (kzg, lagrange) = match (pcs.as_kzg_bn254(), lagrange_pcs.and_then(|l| l.as_kzg_bn254())) {
        (Some(a), Some(b)) => (a, b),
        _ => return Ok(None),
    };
    let n = cs.size();
    let m = cs.quot_eval_dom_size();
    if as_fr_slice(w).is_none() || CS::n_wires_per_gate() != N_WIRES || m != 6 * n || n < 8 || kzg.public_parameter_group_1.len() < n + 3 {
        return Ok(None);
    }
    let root = domain.group_gen;
    let domain_m = match FpPolynomial::<PCS::Field>::quotient_evaluation_domain(m) {
        Some(d) => d,
        None => return Ok(None),
    };
    let root_fr = as_fr_slice(std::slice::from_ref(&root)).unwrap()[0];
    let root_m_fr = as_fr_slice(std::slice::from_ref(&domain_m.group_gen())).unwrap()[0];
    if !same_generator(n as u64, &root_fr) || !same_generator(m as u64, &root_m_fr) {
        return Ok(None); // another root of unity than the library's: stay on the arkworks path
    }
    let circuit = match circuit_for(kzg, lagrange, prover_params, n) {
        Ok(c) => c,
        Err(_) => return Ok(None), // no device / out of device memory: the CPU path
    };
    let ws_fresh = WORKSPACE.with(|w| w.borrow().as_ref().map_or(true, |ws| ws.n != n));
    if ws_fresh {
        match Workspace::new(n) {
            Ok(ws) => WORKSPACE.with(|w| *w.borrow_mut() = Some(ws)),
            Err(_) => return Ok(None),
        }
    }
    WORKSPACE.with(|cell| {
        let guard = cell.borrow();
        let ws = guard.as_ref().unwrap();
        prove_on_device::<R, PCS, CS>(prng, transcript, pcs, cs, prover_params, w, domain, &domain_m, online_values, &circuit, ws).map(Some)
    })
}

#[allow(clippy::too_many_arguments)]
fn prove_on_device<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prng: &mut R,
    transcript: &mut Transcript,
    pcs: &PCS,
    cs: &CS,
    prover_params: &PlonkProverParams<PCS>,
    w: &[PCS::Field],
    domain: &Radix2EvaluationDomain<PCS::Field>,
    domain_m: &impl EvaluationDomain<PCS::Field>,
    online_values: &[PCS::Field],
    circuit: &Circuit,
    ws: &Workspace,
) -> Result<PlonkProof<PCS>, UzkgeError> {
    let n = cs.size();
    let (m, csz) = (6 * n, n + 8);
    let root = domain.group_gen;
    let k = &prover_params.verifier_params.k;
    let k_l = limbs_of(k);
    let mut challenges = PlonkChallenges::new();
    let wrap = |j: &sys::uzk_g1_jac| PCS::commitment_from_g1(jac_from_wire(j)).ok_or(UzkgeError::ProofError);
    let coef = |slot: usize, len: usize| (ws.d_coefs.at(slot * m) as *const c_void, len as u64);

    // 1. the PI polynomial (prover.rs:151-152): n coefficients, uploaded into its 6n-slot
    let pi = pi_poly::<PCS, Radix2EvaluationDomain<_>>(prover_params, online_values, domain);
    let mut pi_coefs = limbs_of(pi.get_coefs_ref());
    pi_coefs.resize(n, [0u64; 4]); // the slot's first n elements are rewritten every proof
    dev(ws.d_coefs.upload(8 * m, &pi_coefs))?;

    // 2. + 3. witness and witness-selector polynomials (prover.rs:154-192): upload the evaluations, ONE batched iFFT into the
    // 6n-slots, ONE hide, ONE batched commit.  The blinds are drawn in the reference's order: wire by wire, then selectors.
    let extended_witness = cs.extend_witness(w);
    dev(ws.d_evals.upload(0, &limbs_of(&extended_witness)))?;
    #[cfg(feature = "shuffle")]
    for (i, sel) in cs.compute_witness_selectors().iter().enumerate() {
        dev(ws.d_evals.upload((N_WIRES + i) * n, &limbs_of(sel)))?;
    }
    let n_first = N_WIRES + N_WSEL;
    let mut hiding: Vec<usize> = (0..N_WIRES).map(|i| cs.get_hiding_degree(i)).collect();
    hiding.extend(std::iter::repeat(2).take(N_WSEL));
    if hiding.iter().any(|h| *h > 3) {
        return Err(UzkgeError::ProofError);
    }
    let mut blinds: Vec<Vec<Fr>> = Vec::new(); // hide_polynomial's draws (helpers.rs:145-153)
    for h in hiding.iter() {
        blinds.push((0..*h).map(|_| Fr::rand(prng)).collect());
    }
    dev(sys::ntt_strided(ws.d_evals.as_ptr(), n, ws.d_coefs.as_ptr(), m, n, n_first as u32, true, None))?;
    // one hiding degree per batched call: three slots each, an unused third slot holds a zero blind (adds nothing)
    let mut hide_l = vec![[0u64; 4]; n_first * 3];
    let mut tails = vec![[0u64; 4]; n_first * 6];
    for (i, b) in blinds.iter().enumerate() {
        for (j, v) in b.iter().enumerate() {
            hide_l[i * 3 + j] = fr_limbs(v);
        }
        tails[i * 6..i * 6 + 6].copy_from_slice(&tail_of(b));
    }
    dev(sys::hide_batch(ws.d_coefs.as_ptr(), m, n, n_first as u32, &hide_l, 3, n))?;
    let cms = dev(circuit.bases.commit_with_tail(ws.d_evals.as_ptr(), n, n, n_first as u32, &tails, 6))?;
    let mut cm_w_vec = Vec::with_capacity(N_WIRES);
    for j in cms[..N_WIRES].iter() {
        let cm_w = wrap(j)?;
        transcript.append_commitment::<PCS::Commitment>(&cm_w);
        cm_w_vec.push(cm_w);
    }
    #[cfg(feature = "shuffle")]
    let mut cm_w_sel_vec = Vec::with_capacity(N_WSEL);
    #[cfg(feature = "shuffle")]
    for j in cms[N_WIRES..].iter() {
        let cm_w_sel = wrap(j)?;
        transcript.append_commitment::<PCS::Commitment>(&cm_w_sel);
        cm_w_sel_vec.push(cm_w_sel);
    }
    let w_len: Vec<usize> = hiding[..N_WIRES].iter().map(|h| n + h).collect(); // coefs.len() after hide_polynomial

    // 4. beta, gamma (prover.rs:194-198)
    let beta: PCS::Field = transcript.get_challenge_field_elem(b"beta");
    transcript.append_single_byte(b"gamma", 0x01);
    let gamma: PCS::Field = transcript.get_challenge_field_elem(b"gamma");
    challenges.insert_beta_gamma(beta, gamma).unwrap(); // safe unwrap

    // 5. z: grand product on the device, iFFT, hide with three blinds, commit (prover.rs:199-209)
    dev(sys::z_poly(ws.d_evals.as_ptr(), circuit.d_perm.as_ptr(), circuit.d_group.as_ptr(), &k_l, &limb_of(&beta), &limb_of(&gamma), n, ws.d_z.as_ptr()))?;
    dev(sys::ntt_strided(ws.d_z.as_ptr(), n, ws.d_coefs.at(9 * m), m, n, 1, true, None))?;
    let z_blinds: Vec<Fr> = (0..3).map(|_| Fr::rand(prng)).collect();
    let z_blinds_l: Vec<Limbs> = z_blinds.iter().map(fr_limbs).collect();
    dev(sys::hide_batch(ws.d_coefs.at(9 * m), m, n, 1, &z_blinds_l, 3, n))?;
    let cm_z = wrap(&dev(circuit.bases.commit_with_tail(ws.d_z.as_ptr(), n, n, 1, &tail_of(&z_blinds), 6))?[0])?;
    transcript.append_commitment::<PCS::Commitment>(&cm_z);

    // 6. alpha
    let alpha: PCS::Field = transcript.get_challenge_field_elem(b"alpha");
    challenges.insert_alpha(alpha).unwrap();

    // 7. t (helpers.rs:223-678): ten coset FFTs over the 6n domain in one call, the quotient kernel, the inverse coset transform
    dev(sys::ntt_strided(ws.d_coefs.as_ptr(), m, ws.d_coset.as_ptr(), m, m, N_PROOF_POLYS as u32, false, Some(&k_l[1])))?;
    let mut qa: sys::uzk_quotient_args = unsafe { std::mem::zeroed() };
    qa.n = n as u32;
    qa.factor = 6;
    for i in 0..N_WIRES {
        qa.vec[sys::UZK_TQ_W + i] = ws.d_coset.at(i * m);
    }
    #[cfg(feature = "shuffle")]
    for i in 0..N_WSEL {
        qa.vec[sys::UZK_TQ_WSEL + i] = ws.d_coset.at((N_WIRES + i) * m);
    }
    qa.vec[sys::UZK_TQ_PI] = ws.d_coset.at(8 * m);
    qa.vec[sys::UZK_TQ_Z] = ws.d_coset.at(9 * m);
    for slot in 0..N_TABLES {
        // without the "shuffle" feature the 25 shuffle / ECC tables stay NULL: terms 12..18 are then not evaluated
        if cfg!(feature = "shuffle") || slot < 21 {
            qa.vec[sys::UZK_TQ_Q + slot] = circuit.d_tables.at(slot * m);
        }
    }
    qa.alpha = limb_of(&alpha);
    qa.beta = limb_of(&beta);
    qa.gamma = limb_of(&gamma);
    for i in 0..N_WIRES {
        qa.k[i] = k_l[i];
    }
    qa.anemoi_g = limb_of(&prover_params.verifier_params.anemoi_generator);
    qa.anemoi_g_inv = limb_of(&prover_params.verifier_params.anemoi_generator_inv);
    #[cfg(feature = "shuffle")]
    {
        qa.edwards_a = limb_of(&cs.get_edwards_a());
    }
    {
        // 1 / Z_H on the coset (helpers.rs:242-252)
        let one = PCS::Field::one();
        let group_gen_pow_n = domain_m.group_gen().pow(&[n as u64]);
        let mut multiplier = k[1].pow(&[n as u64]);
        let mut z_h_inv = Vec::with_capacity(6);
        for _ in 0..6 {
            z_h_inv.push(multiplier - one);
            multiplier *= group_gen_pow_n;
        }
        batch_inversion(&mut z_h_inv);
        for (i, v) in z_h_inv.iter().enumerate() {
            qa.z_h_inv[i] = limb_of(v);
        }
    }
    dev(sys::t_quotient(&qa, ws.d_tq.as_ptr()))?;
    let k_inv = k[1].inverse().ok_or(UzkgeError::DivisionByZero)?;
    dev(sys::ntt_strided(ws.d_tq.as_ptr(), m, ws.d_t.as_ptr(), m, m, 1, true, Some(&limb_of(&k_inv))))?;
    // FpPolynomial::from_coefs trimmed t (helpers.rs:673-677) and its coefs.len() drives the split.  A well-formed proof has
    // deg t = deg z + sum_j deg w_j - n, i.e. 5n - 2 + sum_j hiding_j coefficients: go on with that while the device measures
    // the trimmed length into pinned memory, compare after the commit (which synchronises), redo with the measured length if
    // they ever differ -- same proof bytes as the reference either way, and no synchronisation spent on an answer known in advance.
    let t_len_expected = 5 * n - 2 + hiding[..N_WIRES].iter().sum::<usize>();
    dev(sys::trimmed_len_async(ws.d_t.as_ptr(), m, &[m as u64], ws.h_lens.at(0)))?;

    // split_t_and_commit (helpers.rs:1323-1408) with n = n_constraints + 2: one rand per chunk, drawn in chunk order
    let t_rands: Vec<Fr> = (0..N_WIRES).map(|_| Fr::rand(prng)).collect();
    let t_rands_l: Vec<Limbs> = t_rands.iter().map(fr_limbs).collect();
    let split_and_commit = |t_len: usize| -> Result<(Vec<u64>, Vec<PCS::Commitment>), UzkgeError> {
        let chunk_lens = dev(sys::split_t(ws.d_t.as_ptr(), t_len, n + 2, &t_rands_l, ws.d_chunks.as_ptr(), csz))?;
        let mut cms: Vec<PCS::Commitment> = Vec::with_capacity(N_WIRES);
        if chunk_lens.iter().all(|l| max_power_of_2(*l as usize) == n && *l as usize <= n + 3) {
            // degree = coefs.len() (helpers.rs:1367): every chunk folds onto n coefficients -- one fold, one FFT, one commit
            dev(sys::fold_blinds_batch(ws.d_chunks.as_ptr(), csz, &chunk_lens, n, ws.d_fold.as_ptr(), n, ws.d_tail.as_ptr(), 6))?;
            dev(sys::ntt_strided(ws.d_fold.as_ptr(), n, ws.d_fold.as_ptr(), n, n, N_WIRES as u32, false, None))?;
            for j in dev(circuit.bases.commit_with_device_tail(ws.d_fold.as_ptr(), n, n, N_WIRES as u32, ws.d_tail.as_ptr(), 6))?.iter() {
                cms.push(wrap(j)?);
            }
        } else {
            for (i, l) in chunk_lens.iter().enumerate() {
                cms.push(commit_folded_generic(pcs, circuit, ws, ws.d_chunks.at(i * csz), *l as usize, *l as usize)?);
            }
        }
        Ok((chunk_lens, cms))
    };
    let (mut chunk_lens, mut cm_t_vec) = split_and_commit(t_len_expected)?;
    let t_len = ws.h_lens.get(0) as usize;
    if t_len != t_len_expected {
        let redo = split_and_commit(t_len)?;
        chunk_lens = redo.0;
        cm_t_vec = redo.1;
    }
    for cm_t in cm_t_vec.iter() {
        transcript.append_commitment::<PCS::Commitment>(cm_t);
    }

    // 8. zeta
    let zeta: PCS::Field = transcript.get_challenge_field_elem(b"zeta");
    challenges.insert_zeta(zeta).unwrap();
    let zeta_omega = root * zeta;

    // 9. a) the openings' evaluations (prover.rs:246-273), all in one launch: point 0 = zeta, 1 = zeta * omega
    let mut ev_polys: Vec<(*const c_void, u64)> = Vec::new();
    let mut ev_point: Vec<u32> = Vec::new();
    for i in 0..N_WIRES {
        ev_polys.push(coef(i, w_len[i]));
        ev_point.push(0);
    }
    for i in 0..N_WIRES - 1 {
        ev_polys.push(circuit.tpoly(T_S + i));
        ev_point.push(0);
    }
    ev_polys.push(circuit.tpoly(T_QPRK + 2));
    ev_point.push(0);
    ev_polys.push(circuit.tpoly(T_QPRK + 3));
    ev_point.push(0);
    ev_polys.push(coef(9, n + 3));
    ev_point.push(1);
    for i in 0..3 {
        ev_polys.push(coef(i, w_len[i]));
        ev_point.push(1);
    }
    #[cfg(feature = "shuffle")]
    {
        ev_polys.push(circuit.tpoly(T_QECC));
        ev_point.push(0);
        for i in 0..N_WSEL {
            ev_polys.push(coef(N_WIRES + i, n + 2));
            ev_point.push(0);
        }
    }
    let ev_l = dev(sys::eval_ptrs(&ev_polys, &ev_point, &[limb_of(&zeta), limb_of(&zeta_omega)]))?;
    let ev: Vec<PCS::Field> = ev_l.into_iter().map(field_of::<PCS::Field>).collect();
    let w_polys_eval_zeta: Vec<PCS::Field> = ev[0..5].to_vec();
    let s_polys_eval_zeta: Vec<PCS::Field> = ev[5..9].to_vec();
    let prk_3_poly_eval_zeta = ev[9];
    let prk_4_poly_eval_zeta = ev[10];
    let z_eval_zeta_omega = ev[11];
    let w_polys_eval_zeta_omega: Vec<PCS::Field> = ev[12..15].to_vec();
    #[cfg(feature = "shuffle")]
    let q_ecc_poly_eval_zeta = ev[15];
    #[cfg(feature = "shuffle")]
    let w_sel_polys_eval_zeta: Vec<PCS::Field> = ev[16..19].to_vec();

    //  b) the transcript, in the reference's order (prover.rs:275-294)
    for eval_zeta in w_polys_eval_zeta.iter().chain(s_polys_eval_zeta.iter()) {
        transcript.append_challenge(eval_zeta);
    }
    #[cfg(feature = "shuffle")]
    for eval_zeta in w_sel_polys_eval_zeta.iter() {
        transcript.append_challenge(eval_zeta);
    }
    transcript.append_challenge(&prk_3_poly_eval_zeta);
    transcript.append_challenge(&prk_4_poly_eval_zeta);
    transcript.append_challenge(&z_eval_zeta_omega);
    #[cfg(feature = "shuffle")]
    transcript.append_challenge(&q_ecc_poly_eval_zeta);
    for eval_zeta_omega in w_polys_eval_zeta_omega.iter() {
        transcript.append_challenge(eval_zeta_omega);
    }

    // 10. u
    let u: PCS::Field = transcript.get_challenge_field_elem(b"u");
    challenges.insert_u(u).unwrap();

    // r(X) (helpers.rs:1030-1080): the reference's own formulas give each polynomial's scalar (see `Symbolic`); the device
    // forms the combination.  Basis: q (9), qb, q_prk1, q_prk2, [q_g (12), q_pk (12)], last s, z, t chunks (5).
    let w_refs: Vec<&PCS::Field> = w_polys_eval_zeta.iter().collect();
    #[cfg(feature = "shuffle")]
    let w_omega_refs: Vec<&PCS::Field> = w_polys_eval_zeta_omega.iter().collect();
    let s_refs: Vec<&PCS::Field> = s_polys_eval_zeta.iter().collect();
    #[cfg(feature = "shuffle")]
    let w_sel_refs: Vec<&PCS::Field> = w_sel_polys_eval_zeta.iter().collect();
    let (z_h_eval_zeta, first_lagrange_eval_zeta) = first_lagrange_poly::<PCS>(&challenges, cs.size() as u64);
    let mut r_polys: Vec<(*const c_void, u64)> = Vec::new();
    let mut basis = |p: (*const c_void, u64)| {
        r_polys.push(p);
        Symbolic::<PCS::Field>::basis(r_polys.len() - 1)
    };
    let sym_q: Vec<Symbolic<PCS::Field>> = (0..9).map(|i| basis(circuit.tpoly(T_Q + i))).collect();
    let sym_qb = basis(circuit.tpoly(T_QB));
    let sym_prk1 = basis(circuit.tpoly(T_QPRK));
    let sym_prk2 = basis(circuit.tpoly(T_QPRK + 1));
    #[cfg(feature = "shuffle")]
    let sym_qg: Vec<Symbolic<PCS::Field>> = (0..12).map(|i| basis(circuit.tpoly(T_QG + i))).collect();
    #[cfg(feature = "shuffle")]
    let sym_qpk: Vec<Symbolic<PCS::Field>> = (0..12).map(|i| basis(circuit.tpoly(T_QPK + i))).collect();
    let sym_s_last = basis(circuit.tpoly(T_S + N_WIRES - 1));
    let sym_z = basis(coef(9, n + 3));
    let sym_t: Vec<Symbolic<PCS::Field>> = (0..N_WIRES).map(|i| basis((ws.d_chunks.at(i * csz) as *const c_void, chunk_lens[i]))).collect();
    let multipliers = CS::eval_selector_multipliers(&w_refs).unwrap(); // safe unwrap
    let r_sym = r_poly_or_comm::<PCS::Field, Symbolic<PCS::Field>>(
        &multipliers,
        &sym_q,
        &sym_qb,
        &sym_prk1,
        &sym_prk2,
        #[cfg(feature = "shuffle")]
        &sym_qg,
        #[cfg(feature = "shuffle")]
        &sym_qpk,
        #[cfg(feature = "shuffle")]
        &q_ecc_poly_eval_zeta,
        #[cfg(feature = "shuffle")]
        &w_sel_refs,
        k,
        #[cfg(feature = "shuffle")]
        &cs.get_edwards_a(),
        &sym_s_last,
        &sym_z,
        &w_refs,
        #[cfg(feature = "shuffle")]
        &w_omega_refs,
        &s_refs,
        &prk_3_poly_eval_zeta,
        &z_eval_zeta_omega,
        &challenges,
        &sym_t,
        &first_lagrange_eval_zeta,
        &z_h_eval_zeta,
        n + 2,
    );
    let mut r_scalars = limbs_of(&r_sym.c);
    r_scalars.resize(r_polys.len(), [0u64; 4]);
    dev(sys::lincomb(&r_polys, &r_scalars, ws.d_r.as_ptr(), n + 3))?;

    // the two batch_prove calls (prover.rs:329-372, pcs.rs:107-168): both transcripts first -- batch_prove appends nothing
    // after drawing its alpha -- then both quotients, ONE fold, ONE FFT, ONE commit
    let mut open_zeta: Vec<(*const c_void, u64)> = (0..N_WIRES).map(|i| coef(i, w_len[i])).collect();
    for i in 0..N_WIRES - 1 {
        open_zeta.push(circuit.tpoly(T_S + i));
    }
    open_zeta.push(circuit.tpoly(T_QPRK + 2));
    open_zeta.push(circuit.tpoly(T_QPRK + 3));
    #[cfg(feature = "shuffle")]
    {
        open_zeta.push(circuit.tpoly(T_QECC));
        for i in 0..N_WSEL {
            open_zeta.push(coef(N_WIRES + i, n + 2));
        }
    }
    open_zeta.push((ws.d_r.as_ptr() as *const c_void, (n + 3) as u64));
    let open_zeta_omega = vec![coef(9, n + 3), coef(0, w_len[0]), coef(1, w_len[1]), coef(2, w_len[2])];
    PCS::init_pcs_batch_eval_transcript(transcript, n + 2, &zeta);
    let alpha_1: PCS::Field = transcript.get_challenge_field_elem(b"alpha");
    PCS::init_pcs_batch_eval_transcript(transcript, n + 2, &zeta_omega);
    let alpha_2: PCS::Field = transcript.get_challenge_field_elem(b"alpha");
    dev(sys::open_quotient(&open_zeta, &limb_of(&zeta), &limb_of(&alpha_1), ws.d_q.as_ptr(), csz))?;
    dev(sys::open_quotient(&open_zeta_omega, &limb_of(&zeta_omega), &limb_of(&alpha_2), ws.d_q.at(csz), csz))?;
    // degree = q.degree() (pcs.rs:138) = the trimmed length minus one; expected: hlen - 1 coefficients, hlen = the longest
    // polynomial of the opening.  Measured asynchronously, compared after the commit, as for t.
    let hlen = |polys: &[(*const c_void, u64)]| polys.iter().map(|p| p.1).max().unwrap_or(1);
    let q_expected = vec![hlen(&open_zeta).saturating_sub(1), hlen(&open_zeta_omega).saturating_sub(1)];
    dev(sys::trimmed_len_async(ws.d_q.as_ptr(), csz, &[(n + 3) as u64, (n + 3) as u64], ws.h_lens.at(1)))?;
    let fold_and_commit = |q_lens: &[u64]| -> Result<Vec<PCS::Commitment>, UzkgeError> {
        let degrees: Vec<usize> = q_lens.iter().map(|l| (*l as usize).saturating_sub(1)).collect();
        let mut out: Vec<PCS::Commitment> = Vec::with_capacity(2);
        if degrees.iter().zip(q_lens.iter()).all(|(d, l)| max_power_of_2(*d) == n && *l as usize <= n + 3) {
            dev(sys::fold_blinds_batch(ws.d_q.as_ptr(), csz, q_lens, n, ws.d_fold.as_ptr(), n, ws.d_tail.as_ptr(), 6))?;
            dev(sys::ntt_strided(ws.d_fold.as_ptr(), n, ws.d_fold.as_ptr(), n, n, 2, false, None))?;
            for j in dev(circuit.bases.commit_with_device_tail(ws.d_fold.as_ptr(), n, n, 2, ws.d_tail.as_ptr(), 6))?.iter() {
                out.push(wrap(j)?);
            }
        } else {
            for j in 0..2 {
                out.push(commit_folded_generic(pcs, circuit, ws, ws.d_q.at(j * csz), q_lens[j] as usize, degrees[j])?);
            }
        }
        Ok(out)
    };
    let mut openings = fold_and_commit(&q_expected)?;
    let q_measured = vec![ws.h_lens.get(1), ws.h_lens.get(2)];
    if q_measured != q_expected {
        openings = fold_and_commit(&q_measured)?;
    }
    let opening_witness_zeta_omega = openings.pop().unwrap();
    let opening_witness_zeta = openings.pop().unwrap();

    Ok(PlonkProof {
        cm_w_vec,
        #[cfg(feature = "shuffle")]
        cm_w_sel_vec,
        cm_t_vec,
        cm_z,
        prk_3_poly_eval_zeta,
        prk_4_poly_eval_zeta,
        w_polys_eval_zeta,
        w_polys_eval_zeta_omega,
        z_eval_zeta_omega,
        s_polys_eval_zeta,
        #[cfg(feature = "shuffle")]
        q_ecc_poly_eval_zeta,
        #[cfg(feature = "shuffle")]
        w_sel_polys_eval_zeta,
        opening_witness_zeta,
        opening_witness_zeta_omega,
    })
}
