"""TEST INFRASTRUCTURE ONLY -- the reference's verifier run end to end on the reference's own golden proof.

`tests/plonk_verifier_oracle.py` restates the verifier-side formulas (r(X)'s scalars, r(zeta)) that the GPU chain is held to.  This
file pins THAT restatement on reference-held data: the 52-card shuffle proof, decks and public-key commitments of
contracts/solidity/test/plonk_52.js ("shuffle 52 verify must success") with the verifier key the reference generated for the
circuit (tests/golden/plonk_52_golden.json, extracted by tests/golden/make_plonk52_fixture.py) must be ACCEPTED by

  Transcript                      uzkge/src/utils/transcript.rs:8-70 (Keccak-256 over 32-byte slots; a challenge replaces the state)
  verify_shuffle                  shuffle/src/build_cs.rs:99-129 ("Plonk shuffle Proof", n_cards, public inputs = deck || new deck)
  transcript_init_plonk           uzkge/src/plonk/transcript.rs:9-31
  compute_challenges              uzkge/src/plonk/verifier.rs:166-222
  verifier                        uzkge/src/plonk/verifier.rs:17-164  (first_lagrange_poly, eval_pi_poly, r_eval_zeta, r_commitment,
                                  pcs.batch x 2, batch_verify_diff_points)
  PolyComScheme::batch            uzkge/src/poly_commit/pcs.rs:170-190, init_pcs_batch_eval_transcript :228-246
  batch_verify_diff_points        uzkge/src/poly_commit/kzg_poly_commitment.rs:373-422
  PlonkProof::from_bytes_be       uzkge/src/plonk/indexer.rs:539-590 (layout)

with the pairing over the G2 elements of parameters/srs-padding.bin (oracle/bn254_pairing.py).  Keccak-256 (the pre-standard
padding 0x01, not SHA3-256's 0x06) is restated from the Keccak specification; hashlib does not provide it."""
import json
import os

import bn254_py as opy
import bn254_pairing as pr
import plonk_verifier_oracle as pv
from util import GOLDEN

R, P = opy.R, opy.P

# ---- Keccak-256 ----------------------------------------------------------------------------------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(v, s):
    return ((v << s) | (v >> (64 - s))) & _M if s else v


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data) + b"\x01"
    msg += b"\x00" * ((-len(msg)) % rate)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


# ---- transcript (utils/transcript.rs) --------------------------------------------------------------------------------------
class Transcript:
    def __init__(self, msg: bytes):
        self.state = b""
        self.append_message(msg)

    def append_message(self, msg: bytes):
        if len(msg) < 32:
            msg = b"\x00" * (32 - len(msg)) + msg
        else:
            assert len(msg) % 32 == 0
        self.state += msg

    def append_u64(self, v: int):
        self.state += v.to_bytes(32, "big")

    def append_single_byte(self, b: int):
        self.state += bytes([b])

    def append_commitment(self, pt):                          # to_transcript_bytes: x || y big-endian, infinity as zeros
        x, y = (0, 0) if pt is None else pt
        self.append_message(x.to_bytes(32, "big") + y.to_bytes(32, "big"))

    def append_challenge(self, v: int):
        self.append_message((v % R).to_bytes(32, "big"))

    def challenge(self) -> int:
        c = int.from_bytes(keccak256(self.state), "big") % R
        self.state = c.to_bytes(32, "big")
        return c


def _pt(xy):
    x, y = int(xy[0]), int(xy[1])
    if x == 0 and y == 0:
        return None
    assert opy.g1_is_on_curve((x, y)), "a commitment of the golden case is not on the curve"
    return (x, y)


def proof_from_bytes(raw: bytes):
    """PlonkProof::from_bytes_be (indexer.rs:592-700; layout of to_bytes_be :539-590): 14 uncompressed points, 19 scalars, 2 points,
    32-byte big-endian words."""
    assert len(raw) == 1632
    words = [int.from_bytes(raw[32 * i: 32 * i + 32], "big") for i in range(51)]
    pts = lambda lo, cnt: [_pt(words[lo + 2 * i: lo + 2 * i + 2]) for i in range(cnt)]
    return {"cm_w": pts(0, 5), "cm_wsel": pts(10, 3), "cm_t": pts(16, 5), "cm_z": pts(26, 1)[0],
            "prk3": words[28], "prk4": words[29], "w": words[30:35], "w_omega": words[35:38], "z_omega": words[38],
            "s": words[39:43], "q_ecc": words[43], "wsel": words[44:47], "open_zeta": pts(47, 1)[0], "open_zeta_omega": pts(49, 1)[0]}


def proof_to_bytes(proof) -> bytes:
    """PlonkProof::to_bytes_be (indexer.rs:539-590)"""
    def pt(p):
        x, y = (0, 0) if p is None else p
        return x.to_bytes(32, "big") + y.to_bytes(32, "big")
    sc = lambda v: (v % R).to_bytes(32, "big")
    out = b"".join(pt(p) for p in proof["cm_w"] + proof["cm_wsel"] + proof["cm_t"] + [proof["cm_z"]])
    out += sc(proof["prk3"]) + sc(proof["prk4"]) + b"".join(sc(v) for v in proof["w"] + proof["w_omega"]) + sc(proof["z_omega"])
    out += b"".join(sc(v) for v in proof["s"]) + sc(proof["q_ecc"]) + b"".join(sc(v) for v in proof["wsel"])
    return out + pt(proof["open_zeta"]) + pt(proof["open_zeta_omega"])


def load_golden(cards=52):
    """The reference's golden verification case for a deck of `cards` (52: cs_size 16384; 20: cs_size 4096)."""
    d = json.load(open(os.path.join(GOLDEN, "plonk_%d_golden.json" % cards)))
    proof = proof_from_bytes(bytes.fromhex(d["proof_hex"]))
    vk = {"cm_q": [_pt(p) for p in d["cm_q"]], "cm_s": [_pt(p) for p in d["cm_s"]], "cm_qb": _pt(d["cm_qb"]),
          "cm_prk": [_pt(p) for p in d["cm_prk"]], "cm_q_ecc": _pt(d["cm_q_ecc"]),
          "cm_shuffle_generator": [_pt(p) for p in d["cm_shuffle_generator"]],
          "cm_shuffle_public_key": [_pt(d["pkc"][2 * i: 2 * i + 2]) for i in range(12)],
          "anemoi_g": int(d["anemoi_generator"]), "anemoi_g_inv": int(d["anemoi_generator_inv"]), "k": [int(v) for v in d["k"]],
          "edwards_a": int(d["edwards_a"]), "root": int(d["root"]), "cs_size": int(d["cs_size"]),
          "pi_root_powers": [int(v) for v in d["pi_root_powers"]], "pi_lagrange": [int(v) for v in d["pi_lagrange_constants"]]}
    pi = [int(v) for v in d["deck1"]] + [int(v) for v in d["deck2"]]
    return vk, proof, pi


def verify(vk, proof, pi, n_cards=52, g2=None):
    """verify_shuffle -> verifier: True iff the proof is accepted."""
    if g2 is None:
        g2 = pr.parse_srs_g2(open(os.path.join(GOLDEN, "srs-padding.bin"), "rb").read())
    n = vk["cs_size"]
    t = Transcript(b"Plonk shuffle Proof")
    t.append_u64(n_cards)
    # transcript_init_plonk
    t.append_message(b"PLONK")
    t.append_u64(n)
    t.append_message(R.to_bytes(32, "big"))
    for c in vk["cm_q"] + vk["cm_s"]:
        t.append_commitment(c)
    t.append_challenge(vk["root"])
    for k in vk["k"]:
        t.append_challenge(k)
    for v in pi:
        t.append_challenge(v)
    # compute_challenges
    for c in proof["cm_w"] + proof["cm_wsel"]:
        t.append_commitment(c)
    beta = t.challenge()
    t.append_single_byte(0x01)
    gamma = t.challenge()
    t.append_commitment(proof["cm_z"])
    alpha = t.challenge()
    for c in proof["cm_t"]:
        t.append_commitment(c)
    zeta = t.challenge()
    for v in proof["w"] + proof["s"] + proof["wsel"] + [proof["prk3"], proof["prk4"], proof["z_omega"], proof["q_ecc"]] + proof["w_omega"]:
        t.append_challenge(v)
    u = t.challenge()
    ch = {"alpha": alpha, "beta": beta, "gamma": gamma, "zeta": zeta, "anemoi_g": vk["anemoi_g"], "edwards_a": vk["edwards_a"]}
    ev = {"w": proof["w"], "s": proof["s"], "prk3": proof["prk3"], "prk4": proof["prk4"], "z_omega": proof["z_omega"],
          "w_omega": proof["w_omega"], "q_ecc": proof["q_ecc"], "wsel": proof["wsel"]}
    zh, _ = pv.first_lagrange_poly(zeta, n)
    # eval_pi_poly with the key's own constants: sum_j pi_j c_j / (zeta - root^idx_j) * Z_H(zeta)
    acc = 0
    for v, c, rp in zip(pi, vk["pi_lagrange"], vk["pi_root_powers"]):
        acc += v % R * c % R * pow((zeta - rp) % R, -1, R)
    pi_eval = acc % R * zh % R
    r_eval = pv.r_eval_zeta(ch, n, ev, pi_eval, True, anemoi_g_inv=vk["anemoi_g_inv"])
    scalars = pv.r_scalars(ch, vk["k"], n, ev, True)
    bases = vk["cm_q"] + [proof["cm_z"], vk["cm_s"][4], vk["cm_qb"], vk["cm_prk"][0], vk["cm_prk"][1]] + vk["cm_shuffle_public_key"] + \
        vk["cm_shuffle_generator"] + proof["cm_t"]
    assert len(bases) == len(scalars) == 43
    cm_r = None
    for b, s in zip(bases, scalars):
        if b is not None and s:
            cm_r = opy.g1_add(cm_r, opy.g1_mul(b, s))

    def batch(cms, vals, point):                              # PolyComScheme::batch
        t.append_message(b"New PCS-Batch-Eval Protocol")
        t.append_message(R.to_bytes(32, "big"))
        t.append_u64(n + 2)
        t.append_challenge(point)
        a = t.challenge()
        c_comb, v_comb, mult = None, 0, 1
        for c, v in zip(cms, vals):
            if c is not None:
                c_comb = opy.g1_add(c_comb, opy.g1_mul(c, mult))
            v_comb = (v_comb + mult * v) % R
            mult = mult * a % R
        return c_comb, v_comb
    zeta_omega = zeta * vk["root"] % R
    cms = proof["cm_w"] + vk["cm_s"][:4] + [vk["cm_prk"][2], vk["cm_prk"][3], vk["cm_q_ecc"]] + proof["cm_wsel"] + [cm_r]
    vals = proof["w"] + proof["s"] + [proof["prk3"], proof["prk4"], proof["q_ecc"]] + proof["wsel"] + [r_eval]
    comm, val = batch(cms, vals, zeta)
    comm_o, val_o = batch([proof["cm_z"]] + proof["cm_w"][:3], [proof["z_omega"]] + proof["w_omega"], zeta_omega)
    # batch_verify_diff_points with the challenge u
    g1_0 = opy.G1_GEN
    pi0, pi1 = proof["open_zeta"], opy.g1_mul(proof["open_zeta_omega"], u)
    left = opy.g1_add(pi0, pi1)
    right = opy.g1_add(opy.g1_mul(pi0, zeta), opy.g1_mul(pi1, zeta_omega))
    right = opy.g1_add(right, opy.g1_neg(opy.g1_mul(g1_0, (val + u * val_o) % R)))
    right = opy.g1_add(right, opy.g1_add(comm, opy.g1_mul(comm_o, u)))
    return pr.pairing_product_is_one([(left, g2[1]), (opy.g1_neg(right), g2[0])])
