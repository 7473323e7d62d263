#!/usr/bin/env python3
"""Generates tests/golden/plonk_52_golden.json and plonk_20_golden.json from the reference tree (run where /root/reference exists;
the JSON files are what travels).

DATA ONLY -- the numbers of the reference's own two golden verification cases, a 52-card shuffle proof (cs_size 16384) and a
20-card one (cs_size 4096: another circuit, another domain root):
  * the proof bytes, the two decks (the 416 / 160 public inputs) and the 12 public-key commitments of
    contracts/solidity/test/plonk_52.js ("shuffle 52 verify must success") and plonk_20.js ("shuffle 20 verify must success");
  * the verifier key the reference generated for that circuit (uzkge gen-params output, emitted as constants in
    contracts/solidity/contracts/shuffle/VerifierKey_52.sol / _20.sol: 32 commitment coordinates pairs, anemoi generator and inverse, k (5),
    edwards a, the domain's root, cs_size), the public-input positions as powers of the root (VerifierKeyExtra1_*.sol) and their
    Lagrange constants (VerifierKeyExtra2_*.sol).
No source text is copied: the script reads `mstore(add(vk, OFFSET), VALUE)` / `NAME[i] = VALUE;` pairs and writes the values."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def extract(cards: int, cs_size: int):
    sol = lambda name: open(os.path.join(REF, "contracts/solidity/contracts/shuffle", name % cards)).read()
    js = open(os.path.join(REF, "contracts/solidity/test/plonk_%d.js" % cards)).read()
    key = sol("VerifierKey_%d.sol")
    vk = {int(o, 16): int(v, 0) for o, v in re.findall(r"mstore\(add\(vk, (0x[0-9a-f]+)\), (0x[0-9a-f]+|\d+)\)", key)}
    n_pi = int(re.search(r"mstore\(add\(pi, 0x0\), (\d+)\)", key).group(1))
    idx = [int(v, 16) for v in re.findall(r"PI_POLY_INDICES_LOC\[\d+\] = (0x[0-9a-f]+);", sol("VerifierKeyExtra1_%d.sol"))]
    lag = [int(v, 16) for v in re.findall(r"PI_POLY_LAGRANGE_LOC\[\d+\] = (0x[0-9a-f]+);", sol("VerifierKeyExtra2_%d.sol"))]
    assert len(idx) == len(lag) == n_pi == 8 * cards               # two decks of `cards` masked cards, four coordinates each

    def words(lo, count):
        return [vk.get(lo + 32 * i, 0) for i in range(count)]

    def points(lo, count):
        w = words(lo, 2 * count)
        return [[w[2 * i], w[2 * i + 1]] for i in range(count)]

    # layout of the verifier key block (PlonkVerifier.sol CM_Q0_X_LOC .. CS_SIZE_LOC, relative to CM_Q0_X_LOC)
    out = {
        "source": "zypher-game/uzkge: contracts/solidity/test/plonk_%d.js + contracts/shuffle/VerifierKey{,Extra1,Extra2}_%d.sol (generated constants)" % (cards, cards),
        "cards": cards,
        "cm_q": points(0x000, 9), "cm_s": points(0x240, 5), "cm_qb": points(0x380, 1)[0], "cm_prk": points(0x3c0, 4),
        "cm_q_ecc": points(0x4c0, 1)[0], "cm_shuffle_generator": points(0x500, 12),
    }
    # scalars behind the 12 + 12 shuffle commitments: CM_SHUFFLE_PUBLIC_KEY_0 starts 12 points after the generators
    pk_lo = 0x500 + 12 * 64
    tail = pk_lo + 12 * 64
    out.update({
        "anemoi_generator": vk.get(tail + 0x00, 0), "anemoi_generator_inv": vk.get(tail + 0x20, 0),
        "k": words(tail + 0x40, 5), "edwards_a": vk.get(tail + 0xe0, 0), "root": vk.get(tail + 0x100, 0), "cs_size": vk.get(tail + 0x120, 0),
        "pi_root_powers": idx, "pi_lagrange_constants": lag,
    })
    proof = re.search(r'const proof = "0x([0-9a-f]+)"', js).group(1)
    out["proof_hex"] = proof
    for name in ("deck1", "deck2", "pkc"):
        m = re.search(r"const %s =\s*\[(.*?)\]" % name, js, re.S)
        out[name] = [int(v, 16) for v in re.findall(r"0x[0-9a-f]+", m.group(1))]
    assert len(out["deck1"]) == len(out["deck2"]) == 4 * cards and len(out["pkc"]) == 24 and len(proof) == 2 * 1632
    assert tail == 0xb00 and out["cs_size"] == cs_size, (hex(tail), out["cs_size"])
    json.dump(out, open(os.path.join(HERE, "plonk_%d_golden.json" % cards), "w"))      # Python's json carries big integers exactly
    print("written", cards, {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    extract(52, 16384)
    extract(20, 4096)
