#!/usr/bin/env python3
"""Generates uzkge_amd/csrc/mul29_gfx950.inc: the 9 x 29-bit-limb Montgomery product, squaring and dual
product of fp29.hpp with the multiply-adds of one column emitted as ONE inline-asm statement
(v_mad_u64_u32 chained through a single 64-bit accumulator).

Why: from C++ the compiler splits every column sum into two accumulation chains and joins them with a
64-bit add (v_lshl_add_u64, 4.7 issue cycles) -- 16 per product; chaining the MADs literally removes
them.  The carry-out operand of the MAD goes to VCC and is ignored (a column never overflows 64 bits,
fp29.hpp).  Statements are split at the 30-operand limit of inline asm; the quotient digit, the mask
and the 29-bit column shift stay in C++ (they need the low word of the accumulator)."""
MAXOPS = 30


def stmt(terms):
    """terms: (x, y, y_is_sgpr).  One asm statement: acc += sum x*y."""
    ops, body = [], []
    for (x, y, ys) in terms:
        ix = len(ops) + 1; ops.append(f'"v"({x})')
        iy = len(ops) + 1; ops.append(f'"{"s" if ys else "v"}"({y})')
        body.append(f"v_mad_u64_u32 %0, vcc, %{ix}, %{iy}, %0")
    text = "\\n\\t".join(body)
    return f'    asm("{text}" : "+v"(acc) : {", ".join(ops)} : "vcc");'


def emit(terms, L):
    per = (MAXOPS - 1) // 2
    # balanced split
    n = (len(terms) + per - 1) // per if terms else 0
    for i in range(n):
        lo = i * len(terms) // n; hi = (i + 1) * len(terms) // n
        L.append(stmt(terms[lo:hi]))


def gen(kind):
    L = []
    if kind == "mul":
        L.append("template <class C>\n__device__ __forceinline__ L29 l29_mul_asm(const L29& a, const L29& b) {")
    elif kind == "sqr":
        L.append("template <class C>\n__device__ __forceinline__ L29 l29_sqr_asm(const L29& a) {")
    else:
        L.append("template <class C>\n__device__ __forceinline__ L29 l29_mul2_asm(const L29& a, const L29& b, const L29& c, const L29& d) {")
    L.append("    constexpr uint32_t MASK = (1u << 29) - 1;")
    L.append("    uint64_t acc = 0;")
    L.append("    uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8;")
    L.append("    L29 r;")
    if kind == "sqr":
        L.append("    uint32_t d0 = a.l[0] << 1, d1 = a.l[1] << 1, d2 = a.l[2] << 1, d3 = a.l[3] << 1, d4 = a.l[4] << 1,")
        L.append("             d5 = a.l[5] << 1, d6 = a.l[6] << 1, d7 = a.l[7] << 1, d8 = a.l[8] << 1;")
    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        if kind == "mul":
            prod = [(f"a.l[{i}]", f"b.l[{k - i}]", False) for i in range(lo, hi + 1)]
        elif kind == "mul2":
            prod = [(f"a.l[{i}]", f"b.l[{k - i}]", False) for i in range(lo, hi + 1)]
            prod += [(f"c.l[{i}]", f"d.l[{k - i}]", False) for i in range(lo, hi + 1)]
        else:
            prod = [(f"a.l[{i}]", f"d{k - i}", False) for i in range(lo, hi + 1) if 2 * i < k]
            if k % 2 == 0:
                prod.append((f"a.l[{k // 2}]", f"a.l[{k // 2}]", False))
        red = [(f"m{i}", f"C::M[{k - i}]", True) for i in range(lo, hi + 1) if not (k < 9 and i == k)]
        emit(prod + red, L)
        if k < 9:
            L.append(f"    m{k} = ((uint32_t)acc * C::INV) & MASK;")
            L.append(stmt([(f"m{k}", "C::M[0]", True)]))
        else:
            L.append(f"    r.l[{k - 9}] = (uint32_t)acc & MASK;")
        L.append("    acc >>= 29;")
    L.append("    r.l[8] = (uint32_t)acc;")
    L.append("    return r;\n}")
    return "\n".join(L)


def gen_mulc(scalar_w=False):
    """x * w mod M for a PRECOMPUTED operand w (a twiddle) with its companion wq = floor(w 2^261 / M) -- Shoup's form on 29-bit
    limbs: no quotient digits, no serial chain through them.
      step 1  q~ = the top nine limbs of x * wq, from columns 7..16 only (the dropped low columns move q~ by at most one):
              q~ in {Q - 2, Q - 1, Q} for the true quotient Q = floor(x w / M)                                  53 MADs
      step 2  r = (x * w + q~ * (2^261 - M)) mod 2^261 = x w - q~ M, columns 0..8 only                          90 MADs
    143 MADs + 35 shifts / masks against 162 + 43 for the Montgomery product.  x: value < 2^261, limbs < 2^31.5; w, wq:
    normalized.  Result: normalized, value < 3M, the plain product (no Montgomery factor)."""
    # scalar_w: w and wq are wave-uniform and sit in SGPRs (omega_4 of the NTT butterflies): eighteen VGPRs fewer
    name = "l29_mulcs_asm" if scalar_w else "l29_mulc_asm"
    L = [f"template <class C>\n__device__ __forceinline__ L29 {name}(const L29& x, const L29& w, const L29& wq) {{"]
    L.append("    constexpr uint32_t MASK = (1u << 29) - 1;")
    L.append("    uint64_t acc = 0;")
    L.append("    uint32_t q0, q1, q2, q3, q4, q5, q6, q7, q8;")
    L.append("    L29 r;")
    for k in range(7, 17):
        lo, hi = max(0, k - 8), min(k, 8)
        emit([(f"x.l[{i}]", f"wq.l[{k - i}]", scalar_w) for i in range(lo, hi + 1)], L)
        if k >= 9:
            L.append(f"    q{k - 9} = (uint32_t)acc & MASK;")
        L.append("    acc >>= 29;")
    L.append("    q8 = (uint32_t)acc;")
    L.append("    acc = 0;")
    for k in range(9):
        emit([(f"x.l[{i}]", f"w.l[{k - i}]", scalar_w) for i in range(k + 1)] + [(f"q{i}", f"C::MC[{k - i}]", True) for i in range(k + 1)], L)
        L.append(f"    r.l[{k}] = (uint32_t)acc & MASK;")
        if k < 8:
            L.append("    acc >>= 29;")
    L.append("    return r;\n}")
    return "\n".join(L)


def main():
    out = "// GENERATED by tools/gen_mul29_asm.py -- do not edit.\n" + "\n".join(gen(k) for k in ("mul", "sqr", "mul2")) + "\n" + gen_mulc() + "\n" + gen_mulc(True) + "\n"
    open("uzkge_amd/csrc/mul29_gfx950.inc", "w").write(out)
    print(len(out.splitlines()), "lines")


if __name__ == "__main__":
    main()
