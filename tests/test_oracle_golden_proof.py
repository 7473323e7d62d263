"""CPU: the reference's own golden proofs accepted by the verifier restatement (tests/plonk_golden_verifier.py).

Data: tests/golden/plonk_52_golden.json and plonk_20_golden.json -- the 52-card and the 20-card shuffle proof, decks and public-key
commitments of the reference's contracts/solidity/test/plonk_52.js / plonk_20.js ("shuffle .. verify must success") and the verifier
keys the reference generated for those circuits (cs_size 16384 and 4096: two circuit sizes, two domain roots); the G2 elements come
from parameters/srs-padding.bin.  What this pins: the transcript (Keccak-256 slots, challenge
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


import pytest

CASES = [(52, 16384), (20, 4096)]


@pytest.mark.parametrize("cards,cs_size", CASES)
def test_fixture_matches_the_pinned_constants(cards, cs_size):
    vk, proof, pi = gv.load_golden(cards)
    assert vk["cs_size"] == cs_size and len(pi) == 8 * cards and len(vk["pi_lagrange"]) == 8 * cards
    assert vk["root"] == opy.root_of_unity(cs_size)                                # arkworks' omega, as pinned by the Lagrange SRS of that size
    # the key's Lagrange constants are c_j = omega^j / n at the listed rows (compute_lagrange_constant, helpers.rs:1170-1180)
    ninv = pow(cs_size, -1, opy.R)
    assert all(c == rp * ninv % opy.R for c, rp in zip(vk["pi_lagrange"], vk["pi_root_powers"]))
    assert pow(vk["pi_root_powers"][0], cs_size, opy.R) == 1
    assert sum(1 for v in proof["wsel"] + [proof["q_ecc"]] if v) >= 3                # the shuffle gadget's evaluations are live


@pytest.mark.parametrize("cards,cs_size", CASES)
def test_reference_golden_proof_is_accepted(cards, cs_size):
    import json, os
    from util import GOLDEN
    vk, proof, pi = gv.load_golden(cards)
    assert gv.verify(vk, proof, pi, n_cards=cards)
    # the byte codec (PlonkProof::to_bytes_be / from_bytes_be, indexer.rs:539-700) round-trips the reference's 1632 bytes
    raw = bytes.fromhex(json.load(open(os.path.join(GOLDEN, "plonk_%d_golden.json" % cards)))["proof_hex"])
    assert gv.proof_to_bytes(proof) == raw


@pytest.mark.parametrize("cards,cs_size", CASES)
def test_tampered_inputs_are_rejected(cards, cs_size):
    vk, proof, pi = gv.load_golden(cards)
    p2 = copy.deepcopy(proof); p2["w"][3] = (p2["w"][3] + 1) % opy.R
    assert not gv.verify(vk, p2, pi, n_cards=cards)                                # an evaluation
    pi2 = list(pi); pi2[100] = (pi2[100] + 1) % opy.R
    assert not gv.verify(vk, proof, pi2, n_cards=cards)                            # a public input (a card coordinate)
    p3 = copy.deepcopy(proof); p3["cm_t"][2] = opy.g1_add(p3["cm_t"][2], opy.G1_GEN)
    assert not gv.verify(vk, p3, pi, n_cards=cards)                                # a quotient chunk commitment
    vk2 = copy.deepcopy(vk); vk2["cm_shuffle_public_key"][5] = opy.g1_add(vk2["cm_shuffle_public_key"][5], opy.G1_GEN)
    assert not gv.verify(vk2, proof, pi, n_cards=cards)                            # a public-key selector commitment (r_commitment's shuffle scalars)
    assert not gv.verify(vk, proof, pi, n_cards=cards - 1)                         # the external transcript


def test_a_proof_is_not_accepted_under_the_other_circuits_key():
    vk52, proof52, pi52 = gv.load_golden(52)
    vk20, proof20, pi20 = gv.load_golden(20)
    assert not gv.verify(vk20, proof52, pi20, n_cards=20) and not gv.verify(vk52, proof20, pi52, n_cards=52)
