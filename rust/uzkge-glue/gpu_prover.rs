//! uzkge/src/plonk/gpu_prover.rs -- `prover_with_lagrange` (prover.rs:88-394) on the MI355X (cargo feature `gpu`): marshalling only.
//!
//! The five Fiat-Shamir rounds, the circuit's residency in HBM and every length / fold decision live in libuzkge_gpu.so
//! (`uzk_circuit_*`, `uzk_prove_round1..5`, include/uzkge_gpu.h) -- the implementation the backend's GPU tests hold to the reference's
//! verifier.  What stays here is what only the Rust side has: the transcript, the prng draws (in the reference's order) and
//! `r_poly_or_comm`'s scalars.  `prove` returns `Ok(None)` -- nothing consumed, the caller continues on the CPU path -- when the
//! scheme is not BN254 KZG with a Lagrange SRS of the circuit's size or the device cannot hold the circuit; after the transcript
//! has been touched a device failure is `Err(ProofError)`, never a panic (the reference builds with `panic = "abort"`).
use std::cell::RefCell;
use std::collections::HashMap;
use std::sync::{Arc, Mutex, RwLock};

use ark_bn254::{Fr, G1Projective};
use ark_ec::CurveGroup;
use ark_ff::{PrimeField, UniformRand};
use ark_poly::Radix2EvaluationDomain;
use ark_std::rand::{CryptoRng, RngCore};
use lazy_static::lazy_static;
use serde::{Deserialize, Serialize};
use uzkge_gpu_sys as sys;

use super::{
    constraint_system::ConstraintSystem,
    helpers::{first_lagrange_poly, r_poly_or_comm, PlonkChallenges},
    indexer::{PlonkProof, PlonkProverParams, PlonkVerifierParams},
};
use crate::{
    errors::UzkgeError,
    gpu::{affine_to_wire, as_fr_slice, fr_from_limbs, fr_limbs, from_fr_vec, jac_from_wire},
    poly_commit::{field_polynomial::FpPolynomial, pcs::{HomomorphicPolyComElem, PolyComScheme, ToBytes}},
    utils::transcript::Transcript,
};

type Limbs = [u64; 4];
const N_WIRES: usize = 5;
const N_WSEL: usize = if cfg!(feature = "shuffle") { 3 } else { 0 };

fn limbs_of<F: PrimeField>(v: &[F]) -> Option<Vec<Limbs>> {
    Some(as_fr_slice(v)?.iter().map(fr_limbs).collect())
}
fn field_of<F: PrimeField>(l: &[Limbs]) -> Vec<F> {
    from_fr_vec::<F>(l.iter().map(|x| fr_from_limbs(*x)).collect())
}
fn bytes_of<C: ToBytes>(cms: &[C]) -> Vec<u8> {
    cms.iter().flat_map(|c| c.to_bytes()).collect()
}

/// A circuit resident on the device and the public-key tables it currently holds.
struct Resident {
    circuit: sys::Circuit,
    public_key: Vec<u8>, // verifier_params.cm_shuffle_public_key_vec as bytes
}
lazy_static! {
    /// Keyed by what IDENTIFIES a circuit -- its verifier-key commitments (binding fingerprints of every selector and
    /// permutation polynomial, which the reference maintains anyway) and cs_size -- never by an address.  The public-key
    /// commitments are NOT part of the key: they change once per game and name the tables the circuit currently holds.
    static ref CIRCUITS: Mutex<HashMap<Vec<u8>, Arc<RwLock<Resident>>>> = Mutex::new(HashMap::new());
}
thread_local! {
    /// One context per prover thread (its own stream, workspaces and lock), created on first use and made the thread's current
    /// one: whatever this thread asks of the library besides the rounds -- table refreshes, plain commits -- no longer queues
    /// behind other threads on the default context.
    static CONTEXT: RefCell<Option<sys::Context>> = RefCell::new(None);
    /// (n, proofs per call, prover).  Proofs per call = 1 is a SHARED prover: when several threads prove over the same circuit
    /// at the same time the library runs their round calls as one lockstep launch sequence (include/uzkge_gpu.h,
    /// uzk_coalesce_config) -- the one-proof-per-call API of `prover_with_lagrange` reaches the lockstep throughput unchanged.
    static PROVER: RefCell<Option<(usize, usize, sys::Prover)>> = RefCell::new(None);
}

/// Makes sure the calling thread has its context and that it is current.
fn thread_context() -> Option<()> {
    CONTEXT.with(|cell| {
        let mut c = cell.borrow_mut();
        if c.is_none() {
            let ctx = sys::Context::new().ok()?;
            ctx.make_current().ok()?;
            *c = Some(ctx);
        }
        Some(())
    })
}
/// The thread's prover of `batch` proofs of size n (made again when either changes).
fn thread_prover(n: usize, batch: usize) -> Option<()> {
    let fresh = PROVER.with(|cell| cell.borrow().as_ref().map_or(true, |(size, b, _)| *size != n || *b != batch));
    if fresh {
        let pr = sys::Prover::new(n as u32, batch as u32).ok()?;
        PROVER.with(|cell| *cell.borrow_mut() = Some((n, batch, pr)));
    }
    Some(())
}

fn circuit_key<PCS: PolyComScheme>(vp: &PlonkVerifierParams<PCS>) -> Vec<u8> {
    let mut key = bytes_of(&vp.cm_q_vec);
    key.extend(bytes_of(&vp.cm_s_vec));
    key.extend(vp.cm_qb.to_bytes());
    key.extend(bytes_of(&vp.cm_prk_vec));
    #[cfg(feature = "shuffle")]
    {
        key.extend(vp.cm_q_ecc.to_bytes());
        key.extend(bytes_of(&vp.cm_shuffle_generator_vec));
    }
    key.extend((vp.cs_size as u64).to_le_bytes());
    key
}
fn public_key_of<PCS: PolyComScheme>(_vp: &PlonkVerifierParams<PCS>) -> Vec<u8> {
    #[cfg(feature = "shuffle")]
    let key = bytes_of(&_vp.cm_shuffle_public_key_vec);
    #[cfg(not(feature = "shuffle"))]
    let key = Vec::new();
    key
}
fn coefs_of<F: PrimeField>(polys: &[&FpPolynomial<F>]) -> Option<Vec<Vec<Limbs>>> {
    polys.iter().map(|q| limbs_of(q.get_coefs_ref())).collect()
}

/// The circuit of `p` on the device, built on first use; `refresh_prover_params_public_key` (shuffle/src/gen_params/params.rs:57-129)
/// replaces the twelve public-key selector polynomials in place once per game: when the verifier key's
/// `cm_shuffle_public_key_vec` no longer matches what the device holds, those twelve tables are replaced (copy on write).
fn resident<PCS: PolyComScheme>(kzg: &[G1Projective], lagrange: &[G1Projective], p: &PlonkProverParams<PCS>, n: usize, root: &Limbs) -> Option<Arc<RwLock<Resident>>> {
    let vp = &p.verifier_params;
    let key = circuit_key(vp);
    if let Some(r) = CIRCUITS.lock().ok()?.get(&key) {
        return Some(r.clone());
    }
    let mut slots: Vec<Option<&FpPolynomial<PCS::Field>>> = vec![None; sys::UZK_CIRCUIT_SLOTS];
    for i in 0..9 { slots[sys::UZK_CS_Q + i] = Some(&p.q_polys[i]); }
    for i in 0..N_WIRES { slots[sys::UZK_CS_S + i] = Some(&p.s_polys[i]); }
    slots[sys::UZK_CS_L1] = Some(&p.l1_coefs);
    slots[sys::UZK_CS_QB] = Some(&p.qb_poly);
    for i in 0..4 { slots[sys::UZK_CS_QPRK + i] = Some(&p.q_prk_polys[i]); }
    #[cfg(feature = "shuffle")]
    {
        for i in 0..12 {
            slots[sys::UZK_CS_QPK + i] = Some(&p.q_shuffle_public_key_polys[i]);
            slots[sys::UZK_CS_QG + i] = Some(&p.q_shuffle_generator_polys[i]);
        }
        slots[sys::UZK_CS_QECC] = Some(&p.q_ecc_poly);
    }
    let coefs: Vec<Option<Vec<Limbs>>> = slots.iter().map(|s| s.and_then(|q| limbs_of(q.get_coefs_ref()))).collect();
    let lagrange_wire: Vec<sys::uzk_g1_affine> = G1Projective::normalize_batch(&lagrange[..n]).iter().map(affine_to_wire).collect();
    let blind_g1: Vec<G1Projective> = kzg[0..3].iter().chain(kzg[n..n + 3].iter()).cloned().collect();
    let blind_wire: Vec<sys::uzk_g1_affine> = G1Projective::normalize_batch(&blind_g1).iter().map(affine_to_wire).collect();
    let permutation: Vec<u32> = p.permutation.iter().map(|v| *v as u32).collect();
    let mut d: sys::uzk_circuit_desc = unsafe { std::mem::zeroed() };
    d.n = n as u32;
    d.shuffle = cfg!(feature = "shuffle") as u32;
    d.precompute = 1;
    d.lagrange_bases = lagrange_wire.as_ptr();
    d.blind_bases = blind_wire.as_ptr();
    d.permutation = permutation.as_ptr();
    for (dst, k) in d.k.iter_mut().zip(limbs_of(&vp.k)?.iter()) { *dst = *k; }
    d.anemoi_g = limbs_of(&[vp.anemoi_generator])?[0];
    d.anemoi_g_inv = limbs_of(&[vp.anemoi_generator_inv])?[0];
    #[cfg(feature = "shuffle")]
    { d.edwards_a = limbs_of(&[vp.edwards_a])?[0]; }
    d.group_gen = *root;
    for (slot, c) in coefs.iter().enumerate() {
        if let Some(c) = c {
            d.polys[slot] = c.as_ptr() as *const u64;
            d.poly_lens[slot] = c.len() as u64;
        }
    }
    let circuit = sys::Circuit::create(&d).ok()?; // no device, out of device memory, another root of unity: the CPU path
    let r = Arc::new(RwLock::new(Resident { circuit, public_key: public_key_of(vp) }));
    let mut map = CIRCUITS.lock().ok()?;
    if map.len() >= 8 { map.clear(); } // a handful of circuits per process (one per deck size)
    Some(map.entry(key).or_insert(r).clone())
}

/// `refresh_prover_params_public_key`'s per-table loop (shuffle/src/gen_params/params.rs:88-121) as ONE device call: the twelve
/// selector vectors -> iFFT(n) -> coset FFT(6n) -> Lagrange commit, the resident circuit's tables replaced on the way.  Returns
/// what the caller stores in its parameters (polynomials, coset evaluations, commitments); `None` = not BN254 KZG with a
/// Lagrange SRS of this size, another 6n-th root of unity than the library's, or no device: the caller runs its CPU loop (and
/// the next proof re-uploads the twelve polynomials, see `rounds`).
#[cfg(feature = "shuffle")]
#[allow(clippy::type_complexity)]
pub fn refresh_public_key<PCS: PolyComScheme>(
    pcs: &PCS, lagrange_pcs: Option<&PCS>, p: &PlonkProverParams<PCS>, root: &PCS::Field, root_m: &PCS::Field, evals: &[Vec<PCS::Field>],
) -> Option<(Vec<FpPolynomial<PCS::Field>>, Vec<Vec<PCS::Field>>, Vec<PCS::Commitment>)> {
    let (kzg, lagrange) = (&pcs.as_kzg_bn254()?.public_parameter_group_1, &lagrange_pcs?.as_kzg_bn254()?.public_parameter_group_1);
    let n = p.verifier_params.cs_size;
    if evals.len() != 12 || evals.iter().any(|e| e.len() != n) || n < 16 || kzg.len() < n + 3 || lagrange.len() != n {
        return None;
    }
    if limbs_of(&[*root_m])?[0] != sys::domain_group_gen(6 * n as u64).ok()? {
        return None; // the coset evaluations below come back in the library's enumeration of the 6n domain
    }
    let entry = resident(kzg, lagrange, p, n, &limbs_of(&[*root])?[0])?;
    let flat: Vec<Limbs> = evals.iter().flat_map(|e| limbs_of(e).unwrap_or_default()).collect();
    let mut r = entry.write().ok()?; // exclusive: no proof takes its snapshot of the tables while they are replaced
    let (polys, lens, coset, cms) = r.circuit.refresh_tables(sys::UZK_CS_QPK as u32, &flat, n, true).ok()?;
    let cms: Vec<PCS::Commitment> = cms.iter().map(|j| PCS::commitment_from_g1(jac_from_wire(j))).collect::<Option<_>>()?;
    r.public_key = bytes_of(&cms);
    let polys = (0..12).map(|t| FpPolynomial::from_coefs(field_of(&polys[t * n..t * n + lens[t] as usize]))).collect();
    let coset = (0..12).map(|t| field_of(&coset[t * 6 * n..(t + 1) * 6 * n])).collect();
    Some((polys, coset, cms))
}

/// Drops every circuit's device residency (a proof in flight keeps its tables until it returns).
pub fn release_circuits() {
    if let Ok(mut m) = CIRCUITS.lock() { m.clear(); }
}

// r_poly's scalars without restating its formulas: the reference's r_poly_or_comm (helpers.rs:681-999) is generic over
// "polynomial or commitment"; run on a formal linear combination of its inputs it returns the scalar of each.
#[derive(Clone, Default, Serialize, Deserialize)]
#[serde(bound = "")]
struct Symbolic<F: PrimeField> {
    #[serde(skip)]
    c: Vec<F>,
}
impl<F: PrimeField> Symbolic<F> {
    fn basis(k: usize) -> Self {
        let mut c = vec![F::zero(); k + 1];
        c[k] = F::one();
        Symbolic { c }
    }
    fn zip(&mut self, o: &Self, f: impl Fn(&mut F, &F)) {
        if self.c.len() < o.c.len() { self.c.resize(o.c.len(), F::zero()); }
        for (a, b) in self.c.iter_mut().zip(o.c.iter()) { f(a, b); }
    }
}
impl<F: PrimeField> ToBytes for Symbolic<F> {
    fn to_bytes(&self) -> Vec<u8> { Vec::new() }
    fn to_transcript_bytes(&self) -> Vec<u8> { Vec::new() }
}
impl<F: PrimeField> HomomorphicPolyComElem for Symbolic<F> {
    type Scalar = F;
    fn get_base() -> Self { Self::default() }
    fn get_identity() -> Self { Self::default() }
    fn add(&self, o: &Self) -> Self { let mut r = self.clone(); r.add_assign(o); r }
    fn add_assign(&mut self, o: &Self) { self.zip(o, |a, b| *a += b) }
    fn sub(&self, o: &Self) -> Self { let mut r = self.clone(); r.sub_assign(o); r }
    fn sub_assign(&mut self, o: &Self) { self.zip(o, |a, b| *a -= b) }
    fn mul(&self, s: &F) -> Self { Symbolic { c: self.c.iter().map(|a| *a * s).collect() } }
    fn mul_assign(&mut self, s: &F) { for a in self.c.iter_mut() { *a *= s; } }
}

/// The device-resident body of `prover_with_lagrange`, from "1. Build the PI polynomial" to the returned proof
/// (prover.rs:151-393).  The caller has initialised the transcript and selected `lagrange_pcs` (prover.rs:101-130).
#[allow(clippy::too_many_arguments)]
pub(super) fn prove<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prng: &mut R, transcript: &mut Transcript, pcs: &PCS, lagrange_pcs: Option<&PCS>, cs: &CS, prover_params: &PlonkProverParams<PCS>,
    w: &[PCS::Field], domain: &Radix2EvaluationDomain<PCS::Field>, online_values: &[PCS::Field],
) -> Result<Option<PlonkProof<PCS>>, UzkgeError> {
    let mut prngs = [prng];
    let mut transcripts = [transcript];
    let proofs = prove_lanes::<R, PCS, CS>(&mut prngs, &mut transcripts, pcs, lagrange_pcs, &[cs], prover_params, &[w], domain, &[online_values])?;
    Ok(proofs.and_then(|mut v| v.pop()))
}

/// Several proofs over ONE circuit in lockstep, for hosts that hold several witnesses at once (a dealer proving every
/// player's shuffle, a server draining a queue): proof i is byte for byte what `prover_with_lagrange` makes of
/// (prngs[i], transcripts[i], css[i], ws[i], online_values[i]) alone -- each has its own prng and transcript -- but every step
/// of the five rounds is one launch over all of them (`uzk_prover_create(n, proofs)`).  At most 64 proofs per call.
/// `Ok(None)`: not a proof the device flow covers; nothing was consumed.
#[allow(clippy::too_many_arguments)]
pub fn prove_batch<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prngs: &mut [R], transcripts: &mut [Transcript], pcs: &PCS, lagrange_pcs: Option<&PCS>, css: &[&CS], prover_params: &PlonkProverParams<PCS>,
    ws: &[&[PCS::Field]], domain: &Radix2EvaluationDomain<PCS::Field>, online_values: &[&[PCS::Field]],
) -> Result<Option<Vec<PlonkProof<PCS>>>, UzkgeError> {
    let mut prng_refs: Vec<&mut R> = prngs.iter_mut().collect();
    let mut transcript_refs: Vec<&mut Transcript> = transcripts.iter_mut().collect();
    prove_lanes::<R, PCS, CS>(&mut prng_refs, &mut transcript_refs, pcs, lagrange_pcs, css, prover_params, ws, domain, online_values)
}

#[allow(clippy::too_many_arguments)]
fn prove_lanes<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prngs: &mut [&mut R], transcripts: &mut [&mut Transcript], pcs: &PCS, lagrange_pcs: Option<&PCS>, css: &[&CS], prover_params: &PlonkProverParams<PCS>,
    ws: &[&[PCS::Field]], domain: &Radix2EvaluationDomain<PCS::Field>, online_values: &[&[PCS::Field]],
) -> Result<Option<Vec<PlonkProof<PCS>>>, UzkgeError> {
    // ---- are these proofs the device flow covers?  (nothing is consumed before the answer is yes)
    let lanes = css.len();
    if lanes == 0 || lanes > 64 || prngs.len() != lanes || transcripts.len() != lanes || ws.len() != lanes || online_values.len() != lanes {
        return Ok(None);
    }
    let (kzg, lagrange) = match (pcs.as_kzg_bn254(), lagrange_pcs.and_then(|l| l.as_kzg_bn254())) {
        (Some(a), Some(b)) => (&a.public_parameter_group_1, &b.public_parameter_group_1),
        _ => return Ok(None),
    };
    let cs = css[0];
    let n = cs.size();
    let hiding: Vec<u32> = (0..N_WIRES).map(|i| cs.get_hiding_degree(i) as u32).chain(std::iter::repeat(2).take(N_WSEL)).collect();
    if CS::n_wires_per_gate() != N_WIRES || cs.quot_eval_dom_size() != 6 * n || n < 16 || kzg.len() < n + 3 || lagrange.len() < n || hiding.iter().any(|h| *h > 3) {
        return Ok(None);
    }
    // one circuit: every lane has its size and its hiding degrees
    if css.iter().any(|c| c.size() != n || (0..N_WIRES).any(|i| c.get_hiding_degree(i) as u32 != hiding[i])) {
        return Ok(None);
    }
    let root = match limbs_of(&[domain.group_gen]) { Some(r) => r[0], None => return Ok(None) };
    if thread_context().is_none() {
        return Ok(None);
    }
    let entry = match resident(kzg, lagrange, prover_params, n, &root) { Some(e) => e, None => return Ok(None) };
    let mut extended_witness: Vec<Limbs> = Vec::with_capacity(lanes * N_WIRES * n);
    for (c, w) in css.iter().zip(ws.iter()) {
        match limbs_of(&c.extend_witness(w)) { Some(v) => extended_witness.extend(v), None => return Ok(None) }
    }
    if thread_prover(n, lanes).is_none() {
        return Ok(None);
    }
    PROVER.with(|cell| {
        let guard = cell.borrow();
        let prover = &guard.as_ref().unwrap().2;
        rounds::<R, PCS, CS>(prngs, transcripts, css, prover_params, domain, online_values, &entry, prover, &extended_witness, &hiding).map(Some)
    })
}

/// The five rounds for `css.len()` proofs in lockstep: per-proof arrays are [proof][..] as uzk_prove_round1..5 take them; every
/// transcript and every prng sees exactly the sequence `prover_with_lagrange` gives it.
#[allow(clippy::too_many_arguments)]
fn rounds<R: CryptoRng + RngCore, PCS: PolyComScheme, CS: ConstraintSystem<PCS::Field>>(
    prngs: &mut [&mut R], transcripts: &mut [&mut Transcript], css: &[&CS], prover_params: &PlonkProverParams<PCS>, domain: &Radix2EvaluationDomain<PCS::Field>,
    online_values: &[&[PCS::Field]], entry: &Arc<RwLock<Resident>>, prover: &sys::Prover, extended_witness: &[Limbs], hiding: &[u32],
) -> Result<Vec<PlonkProof<PCS>>, UzkgeError> {
    let dev = |_: sys::Error| UzkgeError::ProofError;
    let wrap = |j: &sys::uzk_g1_jac| PCS::commitment_from_g1(jac_from_wire(j)).ok_or(UzkgeError::ProofError);
    let one = |f: &PCS::Field| limbs_of(std::slice::from_ref(f)).map(|v| v[0]).ok_or(UzkgeError::ProofError);
    let vp = &prover_params.verifier_params;
    let lanes = css.len();
    let n = css[0].size();
    let mut challenges: Vec<PlonkChallenges<PCS::Field>> = (0..lanes).map(|_| PlonkChallenges::new()).collect();
    // 1.-3. (prover.rs:151-192): blinds drawn in the reference's order -- wire by wire, then the selectors; three slots each
    let n_first = N_WIRES + N_WSEL;
    let mut blinds = vec![[0u64; 4]; lanes * n_first * 3];
    for (b, prng) in prngs.iter_mut().enumerate() {
        for (i, h) in hiding.iter().enumerate() {
            for j in 0..*h as usize { blinds[(b * n_first + i) * 3 + j] = fr_limbs(&Fr::rand(&mut **prng)); }
        }
    }
    #[cfg(feature = "shuffle")]
    let wsel: Vec<Limbs> = css.iter().flat_map(|c| c.compute_witness_selectors().iter().flat_map(|s| limbs_of(s).unwrap_or_default()).collect::<Vec<Limbs>>()).collect();
    #[cfg(not(feature = "shuffle"))]
    let wsel: Vec<Limbs> = Vec::new();
    let pi_index: Vec<u32> = vp.public_vars_constraint_indices.iter().map(|i| *i as u32).collect();
    let mut pi_value: Vec<Limbs> = Vec::with_capacity(lanes * pi_index.len());
    for ov in online_values.iter() {
        if ov.len() != pi_index.len() { return Err(UzkgeError::ProofError); }
        pi_value.extend(limbs_of(ov).ok_or(UzkgeError::ProofError)?);
    }
    // The public-key tables are checked against the verifier key and the circuit's tables taken by round 1 under ONE guard: a
    // refresh on another thread cannot slip between the two; from round 1 on these proofs own a snapshot of the tables.  The guard
    // is SHARED (RwLock read): round 1 is where provers of other threads join this one's cohort (coalesce_core.hpp `enter`), so
    // every thread proving over this circuit must be able to stand in round 1 at the same time -- an exclusive lock here would make
    // every cohort one lane wide and have its holder wait out the gathering time for team-mates that are blocked on the lock.
    // Only replacing the tables (here on a mismatch, and `refresh_public_key`) takes the lock exclusively.
    let public_key = public_key_of(vp);
    let cms = loop {
        {
            let r = entry.read().map_err(|_| UzkgeError::ProofError)?;
            if r.public_key == public_key {
                break prover.round1(&r.circuit, extended_witness, &wsel, &pi_index, &pi_value, hiding, &blinds).map_err(dev)?;
            }
        }
        let mut r = entry.write().map_err(|_| UzkgeError::ProofError)?;
        if r.public_key != public_key {
            #[cfg(feature = "shuffle")]
            {
                let polys: Vec<&FpPolynomial<PCS::Field>> = prover_params.q_shuffle_public_key_polys.iter().collect();
                r.circuit.update_tables(sys::UZK_CS_QPK as u32, &coefs_of(&polys).ok_or(UzkgeError::ProofError)?).map_err(dev)?;
            }
            r.public_key = public_key.clone();
        }
        // the write guard drops here; the loop takes the shared guard again and re-checks (std's RwLock cannot downgrade)
    };
    let mut cm_w_vecs: Vec<Vec<PCS::Commitment>> = Vec::with_capacity(lanes);
    #[cfg(feature = "shuffle")]
    let mut cm_w_sel_vecs: Vec<Vec<PCS::Commitment>> = Vec::with_capacity(lanes);
    for (b, transcript) in transcripts.iter_mut().enumerate() {
        let mine = &cms[b * n_first..(b + 1) * n_first];
        let cm_w_vec: Vec<PCS::Commitment> = mine[..N_WIRES].iter().map(wrap).collect::<Result<_, _>>()?;
        for cm in cm_w_vec.iter() { transcript.append_commitment::<PCS::Commitment>(cm); }
        cm_w_vecs.push(cm_w_vec);
        #[cfg(feature = "shuffle")]
        {
            let cm_w_sel_vec: Vec<PCS::Commitment> = mine[N_WIRES..].iter().map(wrap).collect::<Result<_, _>>()?;
            for cm in cm_w_sel_vec.iter() { transcript.append_commitment::<PCS::Commitment>(cm); }
            cm_w_sel_vecs.push(cm_w_sel_vec);
        }
    }
    // 4.-5. beta, gamma; z (prover.rs:194-209)
    let (mut betas, mut gammas, mut z_blinds) = (Vec::with_capacity(lanes), Vec::with_capacity(lanes), Vec::with_capacity(3 * lanes));
    for b in 0..lanes {
        let beta: PCS::Field = transcripts[b].get_challenge_field_elem(b"beta");
        transcripts[b].append_single_byte(b"gamma", 0x01);
        let gamma: PCS::Field = transcripts[b].get_challenge_field_elem(b"gamma");
        challenges[b].insert_beta_gamma(beta, gamma).unwrap(); // safe unwrap
        betas.push(one(&beta)?);
        gammas.push(one(&gamma)?);
        for _ in 0..3 { z_blinds.push(fr_limbs(&Fr::rand(&mut *prngs[b]))); }
    }
    let cm_zs: Vec<PCS::Commitment> = prover.round2(&betas, &gammas, &z_blinds).map_err(dev)?.iter().map(wrap).collect::<Result<_, _>>()?;
    if cm_zs.len() != lanes { return Err(UzkgeError::ProofError); }
    // 6.-7. alpha; t, split_t_and_commit: one rand per chunk, drawn in chunk order (helpers.rs:1351)
    let (mut alphas, mut t_rands) = (Vec::with_capacity(lanes), Vec::with_capacity(N_WIRES * lanes));
    for b in 0..lanes {
        transcripts[b].append_commitment::<PCS::Commitment>(&cm_zs[b]);
        let alpha: PCS::Field = transcripts[b].get_challenge_field_elem(b"alpha");
        challenges[b].insert_alpha(alpha).unwrap();
        alphas.push(one(&alpha)?);
        for _ in 0..N_WIRES { t_rands.push(fr_limbs(&Fr::rand(&mut *prngs[b]))); }
    }
    let cm_ts = prover.round3(&alphas, &t_rands).map_err(dev)?;
    if cm_ts.len() != N_WIRES * lanes { return Err(UzkgeError::ProofError); }
    // 8.-9. zeta; the evaluations (prover.rs:241-273) and their transcript order (prover.rs:275-294)
    let mut cm_t_vecs: Vec<Vec<PCS::Commitment>> = Vec::with_capacity(lanes);
    let (mut zetas, mut zeta_fields) = (Vec::with_capacity(lanes), Vec::with_capacity(lanes));
    for b in 0..lanes {
        let cm_t_vec: Vec<PCS::Commitment> = cm_ts[b * N_WIRES..(b + 1) * N_WIRES].iter().map(wrap).collect::<Result<_, _>>()?;
        for cm_t in cm_t_vec.iter() { transcripts[b].append_commitment::<PCS::Commitment>(cm_t); }
        cm_t_vecs.push(cm_t_vec);
        let zeta: PCS::Field = transcripts[b].get_challenge_field_elem(b"zeta");
        challenges[b].insert_zeta(zeta).unwrap();
        zetas.push(one(&zeta)?);
        zeta_fields.push(zeta);
    }
    let per = if cfg!(feature = "shuffle") { 19 } else { 15 };
    let all_ev: Vec<PCS::Field> = field_of(&prover.round4(&zetas, cfg!(feature = "shuffle")).map_err(dev)?);
    if all_ev.len() != per * lanes { return Err(UzkgeError::ProofError); }
    // 10. u; r(X)'s scalars from the reference's own formulas, in uzk_prove_round5's order:
    // q (9), z, the last s, qb, q_prk1, q_prk2, [q_pk (12), q_g (12)], t chunks (5)
    let mut r_scalars: Vec<Limbs> = Vec::new();
    let (mut alpha_1s, mut alpha_2s) = (Vec::with_capacity(lanes), Vec::with_capacity(lanes));
    for b in 0..lanes {
        let ev = &all_ev[b * per..(b + 1) * per];
        let transcript = &mut *transcripts[b];
        let (w_polys_eval_zeta, s_polys_eval_zeta) = (ev[0..5].to_vec(), ev[5..9].to_vec());
        let (prk_3_poly_eval_zeta, prk_4_poly_eval_zeta, z_eval_zeta_omega) = (ev[9], ev[10], ev[11]);
        let w_polys_eval_zeta_omega = ev[12..15].to_vec();
        #[cfg(feature = "shuffle")]
        let (q_ecc_poly_eval_zeta, w_sel_polys_eval_zeta) = (ev[15], ev[16..19].to_vec());
        for e in w_polys_eval_zeta.iter().chain(s_polys_eval_zeta.iter()) { transcript.append_challenge(e); }
        #[cfg(feature = "shuffle")]
        for e in w_sel_polys_eval_zeta.iter() { transcript.append_challenge(e); }
        transcript.append_challenge(&prk_3_poly_eval_zeta);
        transcript.append_challenge(&prk_4_poly_eval_zeta);
        transcript.append_challenge(&z_eval_zeta_omega);
        #[cfg(feature = "shuffle")]
        transcript.append_challenge(&q_ecc_poly_eval_zeta);
        for e in w_polys_eval_zeta_omega.iter() { transcript.append_challenge(e); }
        let u: PCS::Field = transcript.get_challenge_field_elem(b"u");
        challenges[b].insert_u(u).unwrap();
        let mut next = 0usize;
        let mut basis = |count: usize| -> Vec<Symbolic<PCS::Field>> { next += count; (next - count..next).map(Symbolic::basis).collect() };
        let (sym_q, sym_z, sym_s_last, sym_qb, sym_prk1, sym_prk2) = (basis(9), basis(1), basis(1), basis(1), basis(1), basis(1));
        #[cfg(feature = "shuffle")]
        let (sym_qpk, sym_qg) = (basis(12), basis(12));
        let sym_t = basis(N_WIRES);
        let w_refs: Vec<&PCS::Field> = w_polys_eval_zeta.iter().collect();
        let s_refs: Vec<&PCS::Field> = s_polys_eval_zeta.iter().collect();
        #[cfg(feature = "shuffle")]
        let (w_omega_refs, w_sel_refs): (Vec<&PCS::Field>, Vec<&PCS::Field>) = (w_polys_eval_zeta_omega.iter().collect(), w_sel_polys_eval_zeta.iter().collect());
        let (z_h_eval_zeta, first_lagrange_eval_zeta) = first_lagrange_poly::<PCS>(&challenges[b], n as u64);
        let r_sym = r_poly_or_comm::<PCS::Field, Symbolic<PCS::Field>>(
            &CS::eval_selector_multipliers(&w_refs).unwrap(), // safe unwrap
            &sym_q, &sym_qb[0], &sym_prk1[0], &sym_prk2[0],
            #[cfg(feature = "shuffle")] &sym_qg,
            #[cfg(feature = "shuffle")] &sym_qpk,
            #[cfg(feature = "shuffle")] &q_ecc_poly_eval_zeta,
            #[cfg(feature = "shuffle")] &w_sel_refs,
            &vp.k,
            #[cfg(feature = "shuffle")] &css[b].get_edwards_a(),
            &sym_s_last[0], &sym_z[0], &w_refs,
            #[cfg(feature = "shuffle")] &w_omega_refs,
            &s_refs, &prk_3_poly_eval_zeta, &z_eval_zeta_omega, &challenges[b], &sym_t, &first_lagrange_eval_zeta, &z_h_eval_zeta, n + 2,
        );
        let mut mine = limbs_of(&r_sym.c).ok_or(UzkgeError::ProofError)?;
        mine.resize(next, [0u64; 4]);
        r_scalars.extend(mine);
        // the two batch_prove calls (prover.rs:329-372, pcs.rs:107-118): both transcripts first -- batch_prove appends nothing after
        // drawing its alpha -- then one device round
        let zeta_omega = domain.group_gen * zeta_fields[b];
        PCS::init_pcs_batch_eval_transcript(transcript, n + 2, &zeta_fields[b]);
        let alpha_1: PCS::Field = transcript.get_challenge_field_elem(b"alpha");
        PCS::init_pcs_batch_eval_transcript(transcript, n + 2, &zeta_omega);
        let alpha_2: PCS::Field = transcript.get_challenge_field_elem(b"alpha");
        alpha_1s.push(one(&alpha_1)?);
        alpha_2s.push(one(&alpha_2)?);
    }
    let openings = prover.round5(&r_scalars, &alpha_1s, &alpha_2s).map_err(dev)?;
    if openings.len() != 2 * lanes { return Err(UzkgeError::ProofError); }
    let mut proofs = Vec::with_capacity(lanes);
    #[cfg(feature = "shuffle")]
    let mut cm_w_sel_iter = cm_w_sel_vecs.into_iter();
    for (b, ((cm_w_vec, cm_t_vec), cm_z)) in cm_w_vecs.into_iter().zip(cm_t_vecs.into_iter()).zip(cm_zs.into_iter()).enumerate() {
        let ev = &all_ev[b * per..(b + 1) * per];
        proofs.push(PlonkProof {
            cm_w_vec,
            #[cfg(feature = "shuffle")]
            cm_w_sel_vec: cm_w_sel_iter.next().ok_or(UzkgeError::ProofError)?,
            cm_t_vec,
            cm_z,
            prk_3_poly_eval_zeta: ev[9],
            prk_4_poly_eval_zeta: ev[10],
            w_polys_eval_zeta: ev[0..5].to_vec(),
            w_polys_eval_zeta_omega: ev[12..15].to_vec(),
            z_eval_zeta_omega: ev[11],
            s_polys_eval_zeta: ev[5..9].to_vec(),
            #[cfg(feature = "shuffle")]
            q_ecc_poly_eval_zeta: ev[15],
            #[cfg(feature = "shuffle")]
            w_sel_polys_eval_zeta: ev[16..19].to_vec(),
            opening_witness_zeta: wrap(&openings[2 * b])?,
            opening_witness_zeta_omega: wrap(&openings[2 * b + 1])?,
        });
    }
    Ok(proofs)
}
