#!/usr/bin/env python3
"""Generates uzkge_amd/csrc/mont_mul_gfx950.inc: the device-side 256-bit Montgomery product as
finely-integrated product scanning (FIPS) over 32-bit limbs, written with inline gfx950 assembly.

Why assembly: each partial product is one `v_mad_u64_u32` into a 64-bit column accumulator; the
65th bit (carry-out, in VCC) must be counted by a `v_addc_co_u32` into a third word.  hipcc cannot
express "mad with carry-out" from C++ (it falls back to 64-bit adds plus moves, about 2.3x the
instruction cycles -- measured with tools/microbench/int_rates.hip), so the MAC pairs are emitted
literally.  Statements group a whole column (split only at the 30-operand limit of inline asm) so
the compiler's one-state pad after each asm statement is paid ~20 times per product, not 128.
Carries travel through VCC inside one statement only (never live across statements).
"""
import sys

MAXOPS = 30


def emit_stmt(terms, lines):
    """terms: list of (x_expr, y_expr, y_is_sgpr).  Emits one asm statement adding all products
    into (acc, top)."""
    ops_in = []
    body = []
    for (x, y, ys) in terms:
        ix = len(ops_in) + 2
        ops_in.append(f'"v"({x})')
        iy = len(ops_in) + 2
        ops_in.append(f'"{"s" if ys else "v"}"({y})')
        body.append(f"v_mad_u64_u32 %0, vcc, %{ix}, %{iy}, %0")
        body.append("v_addc_co_u32_e32 %1, vcc, 0, %1, vcc")
    text = "\\n\\t".join(body)
    lines.append(f'    asm("{text}"\n        : "+v"(acc), "+v"(top)\n        : {", ".join(ops_in)}\n        : "vcc");')


def emit_terms(terms, lines):
    per = (MAXOPS - 2) // 2
    for i in range(0, len(terms), per):
        emit_stmt(terms[i:i + per], lines)


def gen(square=False, relaxed=False):
    L = []
    name = "mont_sqr_fips" if square else ("mont_mul_fips_relaxed" if relaxed else "mont_mul_fips")
    args = "const Fp& a" if square else "const Fp& a, const Fp& b"
    L.append("template <class C>")
    L.append(f"__device__ __forceinline__ Fp {name}({args}) {{")
    L.append("    uint64_t acc = 0;")
    L.append("    uint32_t top = 0;")
    L.append("    uint32_t m0, m1, m2, m3, m4, m5, m6, m7;")
    L.append("    Fp r;")
    bname = "a" if square else "b"
    for k in range(16):
        lo = max(0, k - 7)
        hi = min(k, 7)
        terms = [(f"a.v[{i}]", f"{bname}.v[{k - i}]", False) for i in range(lo, hi + 1)]
        mterms = [(f"m{i}", f"C::M[{k - i}]", True) for i in range(lo, hi + 1) if not (k < 8 and i == k)]
        allterms = terms + mterms
        emit_terms(allterms, L)
        if k < 8:
            L.append(f"    m{k} = (uint32_t)acc * C::INV;")
            emit_stmt([(f"m{k}", "C::M[0]", True)], L)
        else:
            L.append(f"    r.v[{k - 8}] = (uint32_t)acc;")
        if k < 15:
            L.append("    acc = (acc >> 32) | ((uint64_t)top << 32);")
            L.append("    top = 0;")
    if not relaxed:
        L.append("    fp_reduce_once_asm<C>(r);")
    L.append("    return r;")
    L.append("}")
    return "\n".join(L)


def gen_addsub(marr="M", suf=""):
    """add / sub / dbl-free helpers: carry chains through VCC, modulus limbs in VGPRs (a VOP2 with
    carry-in cannot also read an SGPR: one constant-bus read per instruction on gfx9)."""
    L = []
    # ---- conditional subtract: r in [0, 2M) -> [0, M)
    L.append("template <class C>")
    L.append(f"__device__ __forceinline__ void fp_reduce_once{suf}_asm(Fp& r) {{")
    L.append("    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;")
    body = ["v_sub_co_u32_e32 %8, vcc, %0, %16"]
    for i in range(1, 8):
        body.append(f"v_subb_co_u32_e32 %{8 + i}, vcc, %{i}, %{16 + i}, vcc")
    for i in range(8):
        body.append(f"v_cndmask_b32_e32 %{i}, %{8 + i}, %{i}, vcc")      # borrow ? r : t
    outs = ", ".join([f'"+v"(r.v[{i}])' for i in range(8)] + [f'"=&v"(t{i})' for i in range(8)])
    ins = ", ".join(f'"v"(C::{marr}[{i}])' for i in range(8))
    L.append('    asm("' + "\\n\\t".join(body) + '"\n        : ' + outs + "\n        : " + ins + '\n        : "vcc");')
    L.append("}")
    # ---- add
    L.append("template <class C>")
    L.append(f"__device__ __forceinline__ Fp fp_add{suf}_asm(const Fp& a, const Fp& b) {{")
    L.append("    Fp r = a;")
    body = ["v_add_co_u32_e32 %0, vcc, %0, %8"]
    for i in range(1, 8):
        body.append(f"v_addc_co_u32_e32 %{i}, vcc, %{i}, %{8 + i}, vcc")
    outs = ", ".join(f'"+v"(r.v[{i}])' for i in range(8))
    ins = ", ".join(f'"v"(b.v[{i}])' for i in range(8))
    L.append('    asm("' + "\\n\\t".join(body) + '"\n        : ' + outs + "\n        : " + ins + '\n        : "vcc");')
    L.append(f"    fp_reduce_once{suf}_asm<C>(r);")
    L.append("    return r;")
    L.append("}")
    # ---- sub: d = a - b; mask = borrow ? ~0 : 0; d += M & mask   (M stays in SGPRs here)
    L.append("template <class C>")
    L.append(f"__device__ __forceinline__ Fp fp_sub{suf}_asm(const Fp& a, const Fp& b) {{")
    L.append("    Fp r = a;")
    L.append("    uint32_t mask;")
    L.append("    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;")
    body = ["v_sub_co_u32_e32 %0, vcc, %0, %9"]
    for i in range(1, 8):
        body.append(f"v_subb_co_u32_e32 %{i}, vcc, %{i}, %{9 + i}, vcc")
    body.append("v_cndmask_b32_e64 %8, 0, -1, vcc")
    outs = ", ".join([f'"+v"(r.v[{i}])' for i in range(8)] + ['"=&v"(mask)'])
    ins = ", ".join(f'"v"(b.v[{i}])' for i in range(8))
    L.append('    asm("' + "\\n\\t".join(body) + '"\n        : ' + outs + "\n        : " + ins + '\n        : "vcc");')
    body = [f"v_and_b32_e32 %{8 + i}, %{17 + i}, %16" for i in range(8)]
    body.append("v_add_co_u32_e32 %0, vcc, %0, %8")
    for i in range(1, 8):
        body.append(f"v_addc_co_u32_e32 %{i}, vcc, %{i}, %{8 + i}, vcc")
    outs = ", ".join([f'"+v"(r.v[{i}])' for i in range(8)] + [f'"=&v"(t{i})' for i in range(8)])
    ins = ", ".join(['"v"(mask)'] + [f'"s"(C::{marr}[{i}])' for i in range(8)])
    L.append('    asm("' + "\\n\\t".join(body) + '"\n        : ' + outs + "\n        : " + ins + '\n        : "vcc");')
    L.append("    return r;")
    L.append("}")
    return "\n".join(L)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "uzkge_amd/csrc/mont_mul_gfx950.inc"
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_mont_asm.py -- do not edit.\n")
        f.write("// Montgomery product a*b*2^-256 mod M for gfx950: FIPS over 32-bit limbs, one\n")
        f.write("// v_mad_u64_u32 + v_addc_co_u32 per partial product (128 pairs + 8 v_mul_lo_u32).\n")
        f.write(gen_addsub())
        f.write("\n\n")
        f.write(gen(False))
        f.write("\n\n")
        f.write("// ---- relaxed domain [0, 2M): the product skips its final conditional subtraction\n")
        f.write("// (inputs < 2M give outputs < 2M because 4M < 2^256); add/sub wrap at 2M.\n")
        f.write(gen_addsub("M2", "_2m"))
        f.write("\n\n")
        f.write(gen(False, relaxed=True))
        f.write("\n")


if __name__ == "__main__":
    main()
