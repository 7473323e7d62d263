"""Shared helpers for the parity tests (test infrastructure; imports the oracle)."""
import os
import random

import numpy as np

import bn254_py as opy
import oracle_c as oc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_fr(n, seed):
    rng = random.Random(seed)
    return [rng.randrange(opy.R) for _ in range(n)]


def rand_fr_wire(n, seed):
    """n uniform Fr elements straight in wire format (any value < r is a valid Montgomery word)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)   # < 2^252 < r
    return a


def load_srs(name):
    """reference SRS file -> [n,8] wire array of affine points (infinity -> zeros)."""
    with open(os.path.join(GOLDEN, name), "rb") as f:
        pts = opy.parse_srs_g1(f.read())
    return oc.points_from_affine(pts), pts


def affine_of(jac_wire):
    return oc.jac_to_affine_ints(np.ascontiguousarray(jac_wire, dtype=np.uint64))


def weighted_index_sum(wire):
    """sum_i s_i * (i+1) mod r for wire-format (Montgomery) scalars, as a canonical int.
    Exact: 16-bit pieces times indices < 2^25, summed over chunks of 2^20, stay below 2^64."""
    n = wire.shape[0]
    idx = np.arange(1, n + 1, dtype=np.uint64)
    total = 0
    for limb in range(4):
        for part in range(4):
            piece = (wire[:, limb] >> np.uint64(16 * part)) & np.uint64(0xFFFF)
            acc = 0
            for lo in range(0, n, 1 << 20):
                acc += int(np.dot(piece[lo:lo + (1 << 20)], idx[lo:lo + (1 << 20)]))
            total += acc << (64 * limb + 16 * part)
    return opy.from_mont(total % opy.R, opy.R)
