"""CPU: the C oracle's t_poly quotient (oracle_t_quotient) against a big-int evaluation of the same
18 terms written directly from uzkge/src/plonk/helpers.rs:284-656 -- two independent restatements
of the reference loop (this row has no reference fixture: parity unpinned)."""
import numpy as np

import bn254_py as opy
import oracle_c as oc
from util import rand_fr_wire

R = opy.R


def _quotient_py(n, factor, V, alpha, beta, gamma, k, g, g_inv, ea, zhi):
    m = n * factor
    out = []
    ap = [pow(alpha, i, R) for i in range(17)]
    for p in range(m):
        nx = (p + factor) % m
        w = [V[j][p] for j in range(5)]
        ws = [V[5 + j][p] for j in range(3)]
        pi, z, zn = V[8][p], V[9][p], V[9][nx]
        w0n, w1n, w2n = V[0][nx], V[1][nx], V[2][nx]
        q = [V[10 + j][p] for j in range(9)]
        t1 = (q[0] * w[0] + q[1] * w[1] + q[2] * w[2] + q[3] * w[3] + q[4] * w[0] * w[1] + q[5] * w[2] * w[3]
              + q[6] + pi + q[7] * w[0] * w[1] * w[2] * w[3] * w[4] - q[8] * w[4])
        t2 = alpha * z
        t3 = alpha * zn
        for j in range(5):
            t2 = t2 * (w[j] + gamma + beta * k[j] * V[30][p]) % R
            t3 = t3 * (w[j] + gamma + beta * V[19 + j][p]) % R
        t4 = ap[2] * V[24][p] * (z - 1)
        qb = V[25][p]
        t5 = ap[3] * qb * w[1] * (w[1] - 1)
        t6 = ap[4] * qb * w[2] * (w[2] - 1)
        t7 = ap[5] * qb * w[3] * (w[3] - 1)
        prk1, prk2, prk3, prk4 = (V[26 + j][p] for j in range(4))
        w3w0, w2w1 = w[0] + w[3], w[1] + w[2]
        w3_2w0, w2_2w1 = w[0] + w3w0, w[1] + w2w1
        tmp = w3w0 + g * w2w1 + prk3
        t8 = ap[6] * prk3 * (pow(tmp - w2n, 5, R) + g * tmp * tmp - (w3_2w0 + g * w2_2w1 + prk1))
        t10 = ap[8] * prk3 * (pow(tmp - w2n, 5, R) + g * w2n * w2n + g_inv - w0n)
        g21 = g * g + 1
        tmp = g * w3w0 + g21 * w2w1 + prk4
        t9 = ap[7] * prk3 * (pow(tmp - w[4], 5, R) + g * tmp * tmp - (g * w3_2w0 + g21 * w2_2w1 + prk2))
        t11 = ap[9] * prk3 * (pow(tmp - w[4], 5, R) + g * w[4] * w[4] + g_inv - w1n)
        qecc = V[55][p]
        sel = [(1 - ws[0]) * (1 - ws[1]) + qecc - 1, ws[0] * (1 - ws[1]), (1 - ws[0]) * ws[1], ws[0] * ws[1]]
        t12 = t13 = t14 = t15 = 0
        for ab in range(4):
            pkx, pky, pkd = V[31 + ab][p], V[35 + ab][p], V[39 + ab][p]
            gx, gy, gd = V[43 + ab][p], V[47 + ab][p], V[51 + ab][p]
            t12 += sel[ab] * (ws[2] * w0n - ws[2] * w[0] * pky - w[1] * pkx + w[0] * w[1] * w0n * pkd)
            t13 += sel[ab] * (ws[2] * w1n + w[0] * ea * pkx - ws[2] * w[1] * pky - w[0] * w[1] * w1n * pkd)
            t14 += sel[ab] * (ws[2] * w2n - ws[2] * w[2] * gy - w[3] * gx + w[2] * w[3] * w2n * gd)
            t15 += sel[ab] * (ws[2] * w[4] + w[2] * ea * gx - ws[2] * w[3] * gy - w[2] * w[3] * w[4] * gd)
        t16 = ap[14] * (qecc * ws[0] * (1 - ws[0]) + (1 - qecc) * ws[0])
        t17 = ap[15] * (qecc * ws[1] * (1 - ws[1]) + (1 - qecc) * ws[1])
        t18 = ap[16] * qecc * (1 + ws[2]) * (1 - ws[2])
        num = (t1 + t2 + (t4 - t3) + t5 + t6 + t7 - t8 - t9 - t10 - t11
               + ap[10] * t12 + ap[11] * t13 + ap[12] * t14 + ap[13] * t15 + t16 + t17 + t18)
        out.append(num * zhi[p % factor] % R)
    return out


def test_c_oracle_matches_bigint_reading():
    for n, factor in ((4, 16), (16, 6)):
        m = n * factor
        vecs = rand_fr_wire(56 * m, 40 + n).reshape(56, m, 4)
        s = rand_fr_wire(12, 41 + n)
        g_inv = oc.fr_inv(s[8])
        zhi = oc.z_h_inv(s[4], n, factor)
        got = oc.fr_to_ints(oc.t_quotient(n, factor, vecs, s[0], s[1], s[2], s[3:8], s[8], g_inv, s[9], zhi))
        V = [oc.fr_to_ints(vecs[i]) for i in range(56)]
        si = oc.fr_to_ints(s)
        want = _quotient_py(n, factor, V, si[0], si[1], si[2], si[3:8], si[8], oc.fr_to_ints(g_inv[None, :])[0], si[9],
                            oc.fr_to_ints(zhi))
        assert got == want
