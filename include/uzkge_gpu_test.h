/* uzkge_gpu_test.h -- TEST HOOKS of libuzkge_gpu.so.  NOT part of the drop-in ABI (include/uzkge_gpu.h): no host binding
 * declares them (rust/uzkge-gpu-sys binds uzkge_gpu.h only).  They are exported by the same shared library so that the known-answer
 * tests run against the very binary that ships -- a separate test build would check other machine code than the product's.
 * Users: tests/ (through uzkge_amd/_native.py TEST_PROTOTYPES) and tests/cpp/prover_rounds.cpp. */
#ifndef UZKGE_GPU_TEST_H
#define UZKGE_GPU_TEST_H
#include "uzkge_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- known-answer entry points: the DEVICE primitives applied element-wise to host arrays -------- */
/* field: 0 = Fq, 1 = Fr.  op: 0 mul (assembly FIPS), 1 add, 2 sub, 3 mul (portable CIOS), 4 sqr,
 * 5 neg, 6 from_mont, 7 to_mont, 8 add (portable), 9 sub (portable); 10..23 exercise the 9 x 29-bit
 * limb representation of the hot loops (fp29.hpp): 10 mul, 11 add, 12/13 sub with 4M / 12M offsets,
 * 14 a lazy-carry chain, 15 form round trip, 16/17 squaring vs product of a lazy operand, 18/19 the
 * multi-subtrahend offsets of ec29.hpp, 20 the dual product, 21..23 the C++ forms of the
 * assembly products 10 / 16 / 20; 24 / 25 the constant-operand product (a * b as PLAIN integers mod M, b canonical; 25 with a lazy
 * first operand 2 (a + 4M)), 26 its companion constant floor(b 2^261 / M) mod 2^256, 27 the NTT's lazy reduction of a + b + 4M, raw
 * (value < 3M, congruent to a + b); 28..33 the typed lazy arithmetic of lz29.hpp on canonical operands: 28 re-limb at offset -5 and
 * back by exact division by 32 (identity), 29 a b, 30 a - b, 31 a b + b b (ld -> typed operation -> to_wire), 32 into the 2^266-form
 * and back by exact division by 2^10 (identity), 33 a lazy chain 4 a - b across every offset the types generate.
 * a, b, out: n elements (host memory). */
int uzk_test_field_kat(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* op: 0 a + b (mixed add), 1 a + b (full XYZZ add), 2 2a, 3 a - b, 4 2(a + b); 5..7 the four-lane addition of the
 * small-MSM folds (ecquad.hpp): 5 a + b, 6 2(a + b) (its doubling branch), 7 (a + b) + (a - b); 8..10 the same three on the
 * 29-bit-limb form (ecquad29.hpp), 11 4(a + b) by two quad doublings, 12 2(a + b) by one, 13 4a; 14..17 the one-lane additions on
 * the lazy 29-bit limbs with re-limbed operands (ec29l.hpp): 14 a + b, 15 2(a + b) (doubling branch), 16 (a + b) + (a - b),
 * 17 ((a + b) - (a + b)) + a + (b + infinity) (cancellation, infinity on either side); 18..21 the same four by quads.
 * Inputs affine (infinity = zeros), outputs Jacobian. */
int uzk_test_g1_kat(int op, const uzk_g1_affine* a, const uzk_g1_affine* b, uzk_g1_jac* out, size_t n);

/* ---- synthetic circuits ----
 * TEST / TIMING ONLY -- changes results.  Marks the circuit as synthetic (random polynomials no witness satisfies, the frozen
 * parity vectors and the timing chains): round 3 then takes t as its first 5 n - 2 + sum(hiding) coefficients, as
 * tests/chain_oracle.py does, and the unsatisfied-witness check is off.  Never set it on a real circuit. */
int uzk_test_circuit_truncate_t(uint64_t circuit, int on);

#ifdef __cplusplus
}
#endif
#endif
