#!/usr/bin/env python3
"""Regenerates rust/uzkge-gpu.patch: applies the `gpu`-feature edits to copies of the reference's files (read from
/root/reference, which exists only in the build container) and writes the unified diff.  The edits are small anchors
(cfg-gated hooks); everything of substance lives in rust/uzkge-glue/{gpu.rs, gpu_prover.rs} and rust/uzkge-gpu-sys.
usage: python tools/make_rust_patch.py [--check]"""
import difflib
import os
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust", "uzkge-gpu.patch")


def sub(text, old, new, path):
    assert text.count(old) == 1, f"{path}: anchor not found exactly once: {old[:60]!r}"
    return text.replace(old, new)


EDITS = {
    "Cargo.toml": [('  "matchmaking",\n', '  "matchmaking",\n  "uzkge-gpu-sys",\n')],
    "uzkge/Cargo.toml": [
        ('tera = { workspace = true, optional = true }\n',
         'tera = { workspace = true, optional = true }\nuzkge-gpu-sys = { path = "../uzkge-gpu-sys", optional = true }\n'),
        ('debug = []\n',
         'debug = []\n# MI355X backend: commit / fft / ifft (src/gpu.rs) and the device-resident prover (src/plonk/gpu_prover.rs);\n'
         '# needs libuzkge_gpu.so, see uzkge-gpu-sys/build.rs\ngpu = ["uzkge-gpu-sys"]\n'),
    ],
    "uzkge/src/lib.rs": [
        ('/// Module for anemoi hash.\n',
         '/// MI355X backend glue (arkworks <-> libuzkge_gpu.so).\n#[cfg(feature = "gpu")]\npub mod gpu;\n\n/// Module for anemoi hash.\n'),
    ],
    "uzkge/src/plonk/mod.rs": [
        ('/// Module for help functions.\npub(crate) mod helpers;\n',
         '/// Module for help functions.\npub(crate) mod helpers;\n\n'
         '/// `prover_with_lagrange` with the proof\'s polynomials resident on the MI355X.\n#[cfg(feature = "gpu")]\nmod gpu_prover;\n'
         '#[cfg(feature = "gpu")]\npub use gpu_prover::{prove_batch as gpu_prove_batch, release_circuits};\n'
         '#[cfg(all(feature = "gpu", feature = "shuffle"))]\npub use gpu_prover::refresh_public_key as gpu_refresh_public_key;\n'),
    ],
    "shuffle/Cargo.toml": [
        ('no_vk = []\n', 'no_vk = []\n# MI355X backend (uzkge/gpu): device-resident prover and the public-key refresh as one device call\ngpu = ["uzkge/gpu"]\n'),
    ],
    "shuffle/src/gen_params/params.rs": [
        ('    let q_shuffle_public_key_evals = params.cs.compute_shuffle_public_key_selectors();\n',
         '    let q_shuffle_public_key_evals = params.cs.compute_shuffle_public_key_selectors();\n\n'
         '    // MI355X: the loop below (12 x iFFT, coset FFT over the 6n domain, Lagrange commit) as one device call that also replaces\n'
         '    // the resident circuit\'s public-key tables; `None` = not applicable or no device: the CPU loop runs as before.\n'
         '    #[cfg(feature = "gpu")]\n'
         '    if let Some((polys, coset_evals, cms)) = uzkge::plonk::gpu_refresh_public_key(\n'
         '        &pcs,\n        lagrange_pcs,\n        &params.prover_params,\n        &domain.group_gen,\n        &domain_m.group_gen,\n'
         '        &q_shuffle_public_key_evals,\n    ) {\n'
         '        let res: Vec<_> = cms.iter().map(|c| c.0).collect();\n'
         '        params.prover_params.q_shuffle_public_key_polys = polys;\n'
         '        params.prover_params.q_shuffle_public_key_coset_evals = coset_evals;\n'
         '        params\n            .prover_params\n            .verifier_params\n            .cm_shuffle_public_key_vec = cms;\n'
         '        return Ok(res);\n    }\n'),
    ],
    # The indexer's six per-table steps (indexer.rs:316-470) each as one device call.  `gpu_tables` is defined for both settings of the
    # feature (without it: always None), so the six hook sites are ordinary code that both builds type-check.
    "uzkge/src/plonk/indexer.rs": [
        ('    // Step 1: compute permutation polynomials and commit them.\n',
         '    // MI355X: the per-table loop of each step below (iFFT, coset FFT over the quotient domain, Lagrange commit) as ONE device\n'
         '    // call per step; `None` = not BN254 KZG over a Lagrange SRS of this size, or no device: the step\'s CPU code runs as before.\n'
         '    #[cfg(feature = "gpu")]\n'
         '    let gpu_tables = |evals: &[&[PCS::Field]]| {\n'
         '        crate::gpu::preprocess_tables(\n            lagrange_pcs,\n            &domain.group_gen,\n            &domain_m.group_gen,\n'
         '            &k[1],\n            evals,\n            no_verifier,\n        )\n    };\n'
         '    #[cfg(not(feature = "gpu"))]\n'
         '    let gpu_tables = |_evals: &[&[PCS::Field]]| -> Option<\n        Vec<(\n            FpPolynomial<PCS::Field>,\n            Vec<PCS::Field>,\n            PCS::Commitment,\n        )>,\n    > { None };\n\n'
         '    // Step 1: compute permutation polynomials and commit them.\n'),
        ('    let mut cm_s_vec = vec![];\n    for i in 0..n_wires_per_gate {\n',
         '    let mut cm_s_vec = vec![];\n'
         '    let mut s_on_device = gpu_tables(\n        &(0..n_wires_per_gate)\n            .map(|i| &encoded_perm[i * n..(i + 1) * n])\n            .collect::<Vec<_>>(),\n    )\n    .map(|t| t.into_iter());\n'
         '    for i in 0..n_wires_per_gate {\n'
         '        if let Some((s_coefs, coset, cm_s)) = s_on_device.as_mut().and_then(|t| t.next()) {\n'
         '            s_coset_evals[i].extend(coset);\n            if no_verifier {\n                cm_s_vec.push(cm_s);\n            }\n'
         '            s_polys.push(s_coefs);\n            continue;\n        }\n'),
        ('    let mut cm_q_vec = vec![];\n    for (i, q_coset_eval) in q_coset_evals.iter_mut().enumerate() {\n',
         '    let mut cm_q_vec = vec![];\n'
         '    let mut q_on_device = {\n        let mut selectors = Vec::with_capacity(CS::num_selectors());\n'
         '        for i in 0..CS::num_selectors() {\n            selectors.push(cs.selector(i)?);\n        }\n'
         '        gpu_tables(&selectors).map(|t| t.into_iter())\n    };\n'
         '    for (i, q_coset_eval) in q_coset_evals.iter_mut().enumerate() {\n'
         '        if let Some((q_coefs, coset, cm_q)) = q_on_device.as_mut().and_then(|t| t.next()) {\n'
         '            q_coset_eval.extend(coset);\n            if no_verifier {\n                cm_q_vec.push(cm_q);\n            }\n'
         '            q_polys.push(q_coefs);\n            continue;\n        }\n'),
        ('            qb[*i] = PCS::Field::one();\n        }\n',
         '            qb[*i] = PCS::Field::one();\n        }\n'
         '        let qb_on_device = gpu_tables(&[qb.as_slice()]).and_then(|t| t.into_iter().next());\n'
         '        if let Some((qb_coef, qb_coset_eval, cm_qb)) = qb_on_device {\n'
         '            (qb_coset_eval, qb_coef, cm_qb)\n        } else {\n'),
        ('        (qb_coset_eval, qb_coef, cm_qb)\n    };\n',
         '        (qb_coset_eval, qb_coef, cm_qb)\n        }\n    };\n'),
        ('        let q_prk_evals = cs.compute_anemoi_jive_selectors().to_vec();\n',
         '        let q_prk_evals = cs.compute_anemoi_jive_selectors().to_vec();\n'
         '        let on_device = gpu_tables(&q_prk_evals.iter().map(|p| p.as_slice()).collect::<Vec<_>>());\n'
         '        if let Some(tables) = on_device {\n'
         '            let (mut coset_evals, mut polys, mut cms) = (vec![], vec![], vec![]);\n'
         '            for (poly, coset_eval, cm) in tables {\n                coset_evals.push(coset_eval);\n                polys.push(poly);\n'
         '                if no_verifier {\n                    cms.push(cm);\n                }\n            }\n'
         '            (coset_evals, polys, cms)\n        } else {\n'),
        ('        (q_prk_coset_evals, q_prk_polys, cm_prk_vec)\n    };\n',
         '        (q_prk_coset_evals, q_prk_polys, cm_prk_vec)\n        }\n    };\n'),
        ('                q_ecc[*i + j] = PCS::Field::one();\n            }\n        }\n',
         '                q_ecc[*i + j] = PCS::Field::one();\n            }\n        }\n'
         '        let q_ecc_on_device = gpu_tables(&[q_ecc.as_slice()]).and_then(|t| t.into_iter().next());\n'
         '        if let Some((q_ecc_coef, q_ecc_coset_eval, cm_q_ecc)) = q_ecc_on_device {\n'
         '            (q_ecc_coset_eval, q_ecc_coef, cm_q_ecc)\n        } else {\n'),
        ('        (q_ecc_coset_eval, q_ecc_coef, cm_q_ecc)\n    };\n',
         '        (q_ecc_coset_eval, q_ecc_coef, cm_q_ecc)\n        }\n    };\n'),
        ('        let q_shuffle_generator_evals = cs.compute_shuffle_generator_selectors();\n',
         '        let q_shuffle_generator_evals = cs.compute_shuffle_generator_selectors();\n'
         '        let on_device = gpu_tables(\n            &q_shuffle_generator_evals\n                .iter()\n                .map(|p| p.as_slice())\n                .collect::<Vec<_>>(),\n        );\n'
         '        if let Some(tables) = on_device {\n'
         '            let (mut coset_evals, mut polys, mut cms) = (vec![], vec![], vec![]);\n'
         '            for (poly, coset_eval, cm) in tables {\n                coset_evals.push(coset_eval);\n                polys.push(poly);\n'
         '                if no_verifier {\n                    cms.push(cm);\n                }\n            }\n'
         '            (coset_evals, polys, cms)\n        } else {\n'),
        ('            cm_shuffle_generator_vec,\n        )\n    };\n',
         '            cm_shuffle_generator_vec,\n        )\n        }\n    };\n'),
    ],
    "uzkge/src/plonk/helpers.rs": [
        ('fn r_poly_or_comm<F: PrimeField, PCSType: HomomorphicPolyComElem<Scalar = F>>(',
         'pub(super) fn r_poly_or_comm<F: PrimeField, PCSType: HomomorphicPolyComElem<Scalar = F>>('),
    ],
    "uzkge/src/plonk/prover.rs": [
        ('            None\n        };\n\n    let commit = |evals: Vec<PCS::Field>,',
         '            None\n        };\n\n'
         '    // MI355X: BN254 KZG with a Lagrange SRS of the circuit\'s size proves with every polynomial resident on the device\n'
         '    // (same transcript, same prng draws, same proof bytes); `None` = not applicable or no device: continue below.\n'
         '    #[cfg(feature = "gpu")]\n'
         '    if let Some(proof) = super::gpu_prover::prove(\n'
         '        prng,\n        transcript,\n        pcs,\n        lagrange_pcs,\n        cs,\n        prover_params,\n        w,\n        &domain,\n        &online_values,\n'
         '    )? {\n        return Ok(proof);\n    }\n\n'
         '    let commit = |evals: Vec<PCS::Field>,'),
    ],
    "uzkge/src/poly_commit/pcs.rs": [
        ('    /// Batch proof for polynomial evaluation.\n    /// `param` stores the instance parameters to be appended to the transcript.\n',
         '    /// MI355X backend: the BN254 KZG scheme behind this PCS, if that is what it is (the device-resident prover asks).\n'
         '    #[cfg(feature = "gpu")]\n'
         '    fn as_kzg_bn254(&self) -> Option<&crate::poly_commit::kzg_poly_commitment::KZGCommitmentSchemeBN254> {\n        None\n    }\n\n'
         '    /// MI355X backend: a G1 point computed on the device as this scheme\'s commitment type.\n'
         '    #[cfg(feature = "gpu")]\n'
         '    fn commitment_from_g1(_point: ark_bn254::G1Projective) -> Option<Self::Commitment> {\n        None\n    }\n\n'
         '    /// Batch proof for polynomial evaluation.\n    /// `param` stores the instance parameters to be appended to the transcript.\n'),
    ],
    "uzkge/src/poly_commit/kzg_poly_commitment.rs": [
        ('        let points_raw =\n            G1Projective::normalize_batch(&self.public_parameter_group_1[0..degree + 1]);\n',
         '        // MI355X: resident SRS + device MSM (trailing zero coefficients are already trimmed: degree + 1 scalars);\n'
         '        // `None` = no usable device, continue on the arkworks path\n'
         '        #[cfg(feature = "gpu")]\n'
         '        if let Some(cm) = crate::gpu::commit(&self.public_parameter_group_1, &coefs[0..degree + 1])? {\n'
         '            return Ok(KZGCommitment(cm));\n        }\n\n'
         '        let points_raw =\n            G1Projective::normalize_batch(&self.public_parameter_group_1[0..degree + 1]);\n'),
        ('    fn eval(&self, poly: &FpPolynomial<Self::Field>, point: &Self::Field) -> Self::Field {\n        poly.eval(point)\n    }\n',
         '    #[cfg(feature = "gpu")]\n    fn as_kzg_bn254(&self) -> Option<&KZGCommitmentSchemeBN254> {\n        Some(self)\n    }\n\n'
         '    #[cfg(feature = "gpu")]\n    fn commitment_from_g1(point: G1Projective) -> Option<Self::Commitment> {\n        Some(KZGCommitment(point))\n    }\n\n'
         '    fn eval(&self, poly: &FpPolynomial<Self::Field>, point: &Self::Field) -> Self::Field {\n        poly.eval(point)\n    }\n'),
    ],
    "uzkge/src/poly_commit/field_polynomial.rs": [
        ('        assert!(domain.size() > self.degree());\n        domain.fft(&self.coefs)\n',
         '        assert!(domain.size() > self.degree());\n        #[cfg(feature = "gpu")]\n'
         '        if let Some(evals) = crate::gpu::fft(domain, &self.coefs, false, None) {\n            return evals;\n        }\n'
         '        domain.fft(&self.coefs)\n'),
        ('        self.mul_var(k).fft_with_domain(domain)\n',
         '        #[cfg(feature = "gpu")]\n        {\n            assert!(domain.size() > self.degree());\n'
         '            // the k^j scaling of mul_var runs inside the device transform\n'
         '            if let Some(evals) = crate::gpu::fft(domain, &self.coefs, false, Some(k)) {\n                return evals;\n            }\n        }\n'
         '        self.mul_var(k).fft_with_domain(domain)\n'),
        ('        let coefs = domain.ifft(&values);\n',
         '        #[cfg(feature = "gpu")]\n        if let Some(coefs) = crate::gpu::fft(domain, values, true, None) {\n            return Self::from_coefs(coefs);\n        }\n'
         '        let coefs = domain.ifft(&values);\n'),
        ('        Self::ifft_with_domain(domain, values).mul_var(k_inv)\n',
         '        #[cfg(feature = "gpu")]\n        if let Some(coefs) = crate::gpu::fft(domain, values, true, Some(k_inv)) {\n            return Self::from_coefs(coefs);\n        }\n'
         '        Self::ifft_with_domain(domain, values).mul_var(k_inv)\n'),
    ],
}


def render() -> str:
    out = []
    for rel in sorted(EDITS):
        old = open(os.path.join(REF, rel)).read()
        new = old
        for a, b in EDITS[rel]:
            new = sub(new, a, b, rel)
        diff = difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True), f"a/{rel}", f"b/{rel}", n=3)
        out.append("".join(diff))
    return "".join(out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference tree is not present: the committed patch is the artefact")
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT} ({text.count(chr(10))} lines)")
