"""CPU: the reference's own golden proof accepted by the verifier restatement (tests/plonk_golden_verifier.py).

Data: tests/golden/plonk_52_golden.json -- the 52-card shuffle proof, decks and public-key commitments of the reference's
contracts/solidity/test/plonk_52.js ("shuffle 52 verify must success") and the verifier key the reference generated for that
circuit; the G2 elements come from parameters/srs-padding.bin.  What this pins: the transcript (Keccak-256 slots, challenge
order), `eval_pi_poly`, `r_eval_zeta` and the 43 scalars of `r_commitment` INCLUDING the shuffle gadget's (the proof's wire
selector and q_ecc evaluations are live), `PolyComScheme::batch`, `batch_verify_diff_points`, the pairing -- i.e. the
verifier-side formulas of tests/plonk_verifier_oracle.py that the GPU prover chain is held to in tests/test_gpu_plonk_verifier.py."""
import copy

import bn254_py as opy
import plonk_golden_verifier as gv


def test_keccak256_known_answers():
    assert gv.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert gv.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert len({gv.keccak256(bytes([i]) * n) for i in range(3) for n in (135, 136, 137, 272)}) == 12   # around the 136-byte rate


def test_fixture_matches_the_pinned_constants():
    vk, proof, pi = gv.load_golden()
    assert vk["cs_size"] == 16384 and len(pi) == 416 and len(vk["pi_lagrange"]) == 416
    assert vk["root"] == opy.root_of_unity(16384)                                  # arkworks' omega, as pinned by the Lagrange SRS
    # the key's Lagrange constants are c_j = omega^j / n at the listed rows (compute_lagrange_constant, helpers.rs:1170-1180)
    ninv = pow(16384, -1, opy.R)
    assert all(c == rp * ninv % opy.R for c, rp in zip(vk["pi_lagrange"], vk["pi_root_powers"]))
    assert pow(vk["pi_root_powers"][0], 16384, opy.R) == 1
    assert sum(1 for v in proof["wsel"] + [proof["q_ecc"]] if v) >= 3                # the shuffle gadget's evaluations are live


def test_reference_golden_proof_is_accepted():
    import json, os
    from util import GOLDEN
    vk, proof, pi = gv.load_golden()
    assert gv.verify(vk, proof, pi)
    # the byte codec (PlonkProof::to_bytes_be / from_bytes_be, indexer.rs:539-700) round-trips the reference's 1632 bytes
    raw = bytes.fromhex(json.load(open(os.path.join(GOLDEN, "plonk_52_golden.json")))["proof_hex"])
    assert gv.proof_to_bytes(proof) == raw


def test_tampered_inputs_are_rejected():
    vk, proof, pi = gv.load_golden()
    p2 = copy.deepcopy(proof); p2["w"][3] = (p2["w"][3] + 1) % opy.R
    assert not gv.verify(vk, p2, pi)                                               # an evaluation
    pi2 = list(pi); pi2[100] = (pi2[100] + 1) % opy.R
    assert not gv.verify(vk, proof, pi2)                                           # a public input (a card coordinate)
    p3 = copy.deepcopy(proof); p3["cm_t"][2] = opy.g1_add(p3["cm_t"][2], opy.G1_GEN)
    assert not gv.verify(vk, p3, pi)                                               # a quotient chunk commitment
    vk2 = copy.deepcopy(vk); vk2["cm_shuffle_public_key"][5] = opy.g1_add(vk2["cm_shuffle_public_key"][5], opy.G1_GEN)
    assert not gv.verify(vk2, proof, pi)                                           # a public-key selector commitment (r_commitment's shuffle scalars)
    assert not gv.verify(vk, proof, pi, n_cards=51)                                # the external transcript
