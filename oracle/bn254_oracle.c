/*
 * CPU ORACLE (plain C, gcc) for the uzkge PlonK hot path: BN254 G1 MSM and Fr NTT.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported CPU baseline.
 * The product path (uzkge_amd/, include/) never links or calls it.
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   MSM  : G1Projective::msm(&points_raw, &coefs)   uzkge/src/poly_commit/kzg_poly_commitment.rs:287-290
 *   NTT  : domain.fft / domain.ifft                 uzkge/src/poly_commit/field_polynomial.rs:583-597
 *   coset: mul_var_assign + fft / ifft + mul_var    uzkge/src/poly_commit/field_polynomial.rs:470-477,589-607
 *   domains 2^k and 3*2^k                           uzkge/src/poly_commit/field_polynomial.rs:554-567
 * The arithmetic proper is in un-vendored crates (ark-ec-zypher / ark-poly-zypher / ark-ff-zypher /
 * ark-bn254-zypher "0.4", Cargo.toml:28-38; no lockfile, no sources in the container), so this file
 * restates the published algorithms: Montgomery CIOS on 4x64-bit limbs with R = 2^256 (ark-ff
 * MontBackend), Jacobian/mixed addition on y^2 = x^3 + 3, Pippenger bucket method
 * (ark-ec VariableBaseMSM), radix-2 Cooley-Tukey with omega_n = 5^((r-1)/n) (ark-poly
 * Radix2EvaluationDomain / MixedRadixEvaluationDomain; natural order in and out).
 * Pinned against the reference's own SRS files by tests/test_oracle_pinning.py.
 *
 * Wire format everywhere: field element = 4 x uint64 little-endian limbs, Montgomery form;
 * affine point = x||y (64 B), infinity = all zero; Jacobian = x||y||z (96 B), infinity <=> z == 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;             /* field element, Montgomery form */
typedef struct { fe x, y; } g1a;                  /* affine; (0,0) = infinity */
typedef struct { fe x, y, z; } g1j;               /* Jacobian; z == 0 = infinity */
typedef struct { fe m; fe r; fe r2; uint64_t inv; } field;

static const field FQ = {
    {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}},
    0x87d20782e4866389ULL};
static const field FR = {
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}},
    0xc2e1f593efffffffULL};

/* ------------------------------------------------------------------ field arithmetic */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }
static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; --i) { if (a->l[i] != b->l[i]) return a->l[i] > b->l[i]; }
    return 1;
}
static inline uint64_t add4(fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; r->l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static inline uint64_t sub4(fe *r, const fe *a, const fe *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - br; r->l[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void f_add(const field *F, fe *r, const fe *a, const fe *b) {
    fe t; add4(&t, a, b);                     /* moduli are < 2^254: no carry out */
    if (fe_geq(&t, &F->m)) sub4(&t, &t, &F->m);
    *r = t;
}
static inline void f_sub(const field *F, fe *r, const fe *a, const fe *b) {
    fe t; if (sub4(&t, a, b)) add4(&t, &t, &F->m);
    *r = t;
}
static inline void f_neg(const field *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub4(r, &F->m, a);
}
static inline void f_dbl(const field *F, fe *r, const fe *a) { f_add(F, r, a, a); }
/* Montgomery product a*b*2^-256 mod m (CIOS, 64-bit words). */
static void f_mul(const field *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->m.l[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * F->m.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &F->m)) sub4(&o, &o, &F->m);
    *r = o;
}
static inline void f_sqr(const field *F, fe *r, const fe *a) { f_mul(F, r, a, a); }
static void f_from_mont(const field *F, fe *r, const fe *a) { fe one = {{1, 0, 0, 0}}; f_mul(F, r, a, &one); }
static void f_to_mont(const field *F, fe *r, const fe *a) { f_mul(F, r, a, &F->r2); }
/* a^e, e = canonical 256-bit exponent (not Montgomery) */
static void f_pow(const field *F, fe *r, const fe *a, const fe *e) {
    fe acc = F->r, base = *a;
    for (int i = 0; i < 256; ++i) {
        if ((e->l[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &acc, &base);
        f_sqr(F, &base, &base);
    }
    *r = acc;
}
static void f_inv(const field *F, fe *r, const fe *a) {   /* Fermat: a^(m-2) */
    fe e = F->m, two = {{2, 0, 0, 0}}; sub4(&e, &e, &two);
    f_pow(F, r, a, &e);
}
static void f_from_u64(const field *F, fe *r, uint64_t v) { fe t = {{v, 0, 0, 0}}; f_to_mont(F, r, &t); }

/* ------------------------------------------------------------------ G1: y^2 = x^3 + 3 */
static inline int a_is_inf(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void j_set_inf(g1j *p) { memset(p, 0, sizeof *p); p->x = FQ.r; p->y = FQ.r; }
static inline void j_from_a(g1j *r, const g1a *p) {
    if (a_is_inf(p)) { j_set_inf(r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ.r;
}
static void j_double(g1j *r, const g1j *p) {      /* dbl-2009-l (a = 0) */
    if (fe_is_zero(&p->z)) { *r = *p; return; }
    fe A, B, C, D, E, Fv, t, X3, Y3, Z3;
    f_sqr(&FQ, &A, &p->x); f_sqr(&FQ, &B, &p->y); f_sqr(&FQ, &C, &B);
    f_add(&FQ, &t, &p->x, &B); f_sqr(&FQ, &t, &t); f_sub(&FQ, &t, &t, &A); f_sub(&FQ, &t, &t, &C);
    f_dbl(&FQ, &D, &t);
    f_dbl(&FQ, &E, &A); f_add(&FQ, &E, &E, &A);
    f_sqr(&FQ, &Fv, &E);
    f_dbl(&FQ, &t, &D); f_sub(&FQ, &X3, &Fv, &t);
    f_sub(&FQ, &t, &D, &X3); f_mul(&FQ, &Y3, &E, &t);
    f_dbl(&FQ, &t, &C); f_dbl(&FQ, &t, &t); f_dbl(&FQ, &t, &t); f_sub(&FQ, &Y3, &Y3, &t);
    f_mul(&FQ, &Z3, &p->y, &p->z); f_dbl(&FQ, &Z3, &Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void j_add(g1j *r, const g1j *p, const g1j *q) {   /* add-2007-bl, complete via branches */
    if (fe_is_zero(&p->z)) { *r = *q; return; }
    if (fe_is_zero(&q->z)) { *r = *p; return; }
    fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, X3, Y3, Z3;
    f_sqr(&FQ, &Z1Z1, &p->z); f_sqr(&FQ, &Z2Z2, &q->z);
    f_mul(&FQ, &U1, &p->x, &Z2Z2); f_mul(&FQ, &U2, &q->x, &Z1Z1);
    f_mul(&FQ, &S1, &p->y, &q->z); f_mul(&FQ, &S1, &S1, &Z2Z2);
    f_mul(&FQ, &S2, &q->y, &p->z); f_mul(&FQ, &S2, &S2, &Z1Z1);
    if (fe_eq(&U1, &U2)) {
        if (fe_eq(&S1, &S2)) { j_double(r, p); return; }
        j_set_inf(r); return;
    }
    f_sub(&FQ, &H, &U2, &U1);
    f_dbl(&FQ, &I, &H); f_sqr(&FQ, &I, &I);
    f_mul(&FQ, &J, &H, &I);
    f_sub(&FQ, &rr, &S2, &S1); f_dbl(&FQ, &rr, &rr);
    f_mul(&FQ, &V, &U1, &I);
    f_sqr(&FQ, &X3, &rr); f_sub(&FQ, &X3, &X3, &J); f_dbl(&FQ, &t, &V); f_sub(&FQ, &X3, &X3, &t);
    f_sub(&FQ, &t, &V, &X3); f_mul(&FQ, &Y3, &rr, &t);
    f_mul(&FQ, &t, &S1, &J); f_dbl(&FQ, &t, &t); f_sub(&FQ, &Y3, &Y3, &t);
    f_add(&FQ, &Z3, &p->z, &q->z); f_sqr(&FQ, &Z3, &Z3); f_sub(&FQ, &Z3, &Z3, &Z1Z1);
    f_sub(&FQ, &Z3, &Z3, &Z2Z2); f_mul(&FQ, &Z3, &Z3, &H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void j_add_affine(g1j *r, const g1j *p, const g1a *q, int negate) {   /* madd-2007-bl */
    if (a_is_inf(q)) { *r = *p; return; }
    fe qy = q->y; if (negate) f_neg(&FQ, &qy, &qy);
    if (fe_is_zero(&p->z)) { r->x = q->x; r->y = qy; r->z = FQ.r; return; }
    fe Z1Z1, U2, S2, H, HH, I, J, rr, V, t, X3, Y3, Z3;
    f_sqr(&FQ, &Z1Z1, &p->z);
    f_mul(&FQ, &U2, &q->x, &Z1Z1);
    f_mul(&FQ, &S2, &qy, &p->z); f_mul(&FQ, &S2, &S2, &Z1Z1);
    if (fe_eq(&p->x, &U2)) {
        if (fe_eq(&p->y, &S2)) { j_double(r, p); return; }
        j_set_inf(r); return;
    }
    f_sub(&FQ, &H, &U2, &p->x); f_sqr(&FQ, &HH, &H);
    f_dbl(&FQ, &I, &HH); f_dbl(&FQ, &I, &I);
    f_mul(&FQ, &J, &H, &I);
    f_sub(&FQ, &rr, &S2, &p->y); f_dbl(&FQ, &rr, &rr);
    f_mul(&FQ, &V, &p->x, &I);
    f_sqr(&FQ, &X3, &rr); f_sub(&FQ, &X3, &X3, &J); f_dbl(&FQ, &t, &V); f_sub(&FQ, &X3, &X3, &t);
    f_sub(&FQ, &t, &V, &X3); f_mul(&FQ, &Y3, &rr, &t);
    f_mul(&FQ, &t, &p->y, &J); f_dbl(&FQ, &t, &t); f_sub(&FQ, &Y3, &Y3, &t);
    f_add(&FQ, &Z3, &p->z, &H); f_sqr(&FQ, &Z3, &Z3); f_sub(&FQ, &Z3, &Z3, &Z1Z1); f_sub(&FQ, &Z3, &Z3, &HH);
    r->x = X3; r->y = Y3; r->z = Z3;
}
static void j_to_affine(g1a *r, const g1j *p) {
    if (fe_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
    fe zi, zi2, zi3;
    f_inv(&FQ, &zi, &p->z); f_sqr(&FQ, &zi2, &zi); f_mul(&FQ, &zi3, &zi2, &zi);
    f_mul(&FQ, &r->x, &p->x, &zi2); f_mul(&FQ, &r->y, &p->y, &zi3);
}
/* k = canonical scalar (NOT Montgomery), 256 bits */
static void j_mul_canon(g1j *r, const g1j *p, const fe *k) {
    g1j acc; j_set_inf(&acc);
    for (int i = 255; i >= 0; --i) {
        j_double(&acc, &acc);
        if ((k->l[i >> 6] >> (i & 63)) & 1) j_add(&acc, &acc, p);
    }
    *r = acc;
}

/* ------------------------------------------------------------------ exported: field / group KATs */
void oracle_fq_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { f_mul(&FQ, (fe *)out, (const fe *)a, (const fe *)b); }
void oracle_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *out) { f_mul(&FR, (fe *)out, (const fe *)a, (const fe *)b); }
void oracle_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *out) { f_add(&FR, (fe *)out, (const fe *)a, (const fe *)b); }
void oracle_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *out) { f_sub(&FR, (fe *)out, (const fe *)a, (const fe *)b); }
void oracle_fr_inv(const uint64_t *a, uint64_t *out) { f_inv(&FR, (fe *)out, (const fe *)a); }
void oracle_fq_inv(const uint64_t *a, uint64_t *out) { f_inv(&FQ, (fe *)out, (const fe *)a); }
void oracle_fr_to_mont(const uint64_t *a, uint64_t *out) { f_to_mont(&FR, (fe *)out, (const fe *)a); }
void oracle_fr_from_mont(const uint64_t *a, uint64_t *out) { f_from_mont(&FR, (fe *)out, (const fe *)a); }
void oracle_fq_to_mont(const uint64_t *a, uint64_t *out) { f_to_mont(&FQ, (fe *)out, (const fe *)a); }
void oracle_fq_from_mont(const uint64_t *a, uint64_t *out) { f_from_mont(&FQ, (fe *)out, (const fe *)a); }
void oracle_g1_add(const uint64_t *p, const uint64_t *q, uint64_t *out) { j_add((g1j *)out, (const g1j *)p, (const g1j *)q); }
void oracle_g1_double(const uint64_t *p, uint64_t *out) { j_double((g1j *)out, (const g1j *)p); }
void oracle_g1_add_affine(const uint64_t *p, const uint64_t *q, uint64_t *out) { j_add_affine((g1j *)out, (const g1j *)p, (const g1a *)q, 0); }
void oracle_g1_to_affine(const uint64_t *p, uint64_t *out) { j_to_affine((g1a *)out, (const g1j *)p); }
/* scalar in Montgomery form (wire format) */
void oracle_g1_mul(const uint64_t *p_affine, const uint64_t *scalar_mont, uint64_t *out_jac) {
    fe k; f_from_mont(&FR, &k, (const fe *)scalar_mont);
    g1j b; j_from_a(&b, (const g1a *)p_affine);
    j_mul_canon((g1j *)out_jac, &b, &k);
}
int oracle_g1_is_on_curve(const uint64_t *p_affine) {
    const g1a *p = (const g1a *)p_affine;
    if (a_is_inf(p)) return 1;
    fe l, r3, three;
    f_sqr(&FQ, &l, &p->y); f_sqr(&FQ, &r3, &p->x); f_mul(&FQ, &r3, &r3, &p->x);
    f_from_u64(&FQ, &three, 3); f_add(&FQ, &r3, &r3, &three);
    return fe_eq(&l, &r3);
}

/* ------------------------------------------------------------------ MSM */
/* Definition used by the reference's own test_commit (kzg_poly_commitment.rs:526-548):
 * one double-and-add per term, summed. */
void oracle_msm_naive(const uint64_t *points, const uint64_t *scalars_mont, size_t n, uint64_t *out_jac) {
    const g1a *P = (const g1a *)points; const fe *S = (const fe *)scalars_mont;
    g1j acc; j_set_inf(&acc);
    for (size_t i = 0; i < n; ++i) {
        fe k; f_from_mont(&FR, &k, &S[i]);
        if (fe_is_zero(&k) || a_is_inf(&P[i])) continue;
        g1j b, t; j_from_a(&b, &P[i]); j_mul_canon(&t, &b, &k); j_add(&acc, &acc, &t);
    }
    *(g1j *)out_jac = acc;
}

/* One job = one window over one contiguous range of points; jobs are handed out through a shared
 * counter so any thread count (up to windows x parts) keeps every core busy. */
typedef struct {
    const g1a *P; fe *K; const fe *S; size_t n; int c, nwin, parts; g1j *win_part;   /* [nwin][parts] */
    volatile long next_job, next_conv;
} msm_shared;

static inline uint32_t get_bits(const fe *k, int lo, int c) {
    int limb = lo >> 6, sh = lo & 63;
    if (limb >= 4) return 0;
    uint64_t v = k->l[limb] >> sh;
    if (sh + c > 64 && limb + 1 < 4) v |= k->l[limb + 1] << (64 - sh);
    return (uint32_t)(v & ((1ULL << c) - 1));
}
static void *msm_conv_worker(void *arg) {        /* scalars out of Montgomery form, 4096 at a time */
    msm_shared *sh = (msm_shared *)arg;
    for (;;) {
        long blk = __sync_fetch_and_add(&sh->next_conv, 1);
        size_t lo = (size_t)blk * 4096, hi = lo + 4096;
        if (lo >= sh->n) break;
        if (hi > sh->n) hi = sh->n;
        for (size_t i = lo; i < hi; ++i) f_from_mont(&FR, &sh->K[i], &sh->S[i]);
    }
    return NULL;
}
static void *msm_worker(void *arg) {
    msm_shared *sh = (msm_shared *)arg;
    size_t nb = (size_t)1 << sh->c;
    g1j *buckets = (g1j *)malloc(nb * sizeof(g1j));
    for (;;) {
        long job = __sync_fetch_and_add(&sh->next_job, 1);
        if (job >= (long)sh->nwin * sh->parts) break;
        int w = (int)(job / sh->parts), part = (int)(job % sh->parts);
        size_t lo = sh->n * (size_t)part / sh->parts, hi = sh->n * (size_t)(part + 1) / sh->parts;
        for (size_t b = 0; b < nb; ++b) j_set_inf(&buckets[b]);
        for (size_t i = lo; i < hi; ++i) {
            uint32_t d = get_bits(&sh->K[i], w * sh->c, sh->c);
            if (d) j_add_affine(&buckets[d], &buckets[d], &sh->P[i], 0);
        }
        g1j run, acc; j_set_inf(&run); j_set_inf(&acc);
        for (size_t b = nb - 1; b >= 1; --b) { j_add(&run, &run, &buckets[b]); j_add(&acc, &acc, &run); }
        sh->win_part[(size_t)w * sh->parts + part] = acc;
    }
    free(buckets);
    return NULL;
}
/* Pippenger bucket method, plain (unsigned) windows of c bits, threads over (window, point range)
 * jobs.  This is the timed "port" CPU baseline in bench.py. */
void oracle_msm_pippenger(const uint64_t *points, const uint64_t *scalars_mont, size_t n,
                          int c, int threads, uint64_t *out_jac) {
    if (c < 1) {
        c = 3; while ((1ULL << (c + 3)) < n && c < 16) ++c;     /* ~ log2(n) - 3 */
    }
    int nwin = (254 + c - 1) / c;
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    /* point ranges per window: enough jobs for every thread, none shorter than 8 buckets' worth */
    int parts = (threads + nwin - 1) / nwin;
    if (threads > 1 && parts < 2 && (threads % nwin)) parts = 2;
    while (parts > 1 && n / (size_t)parts < ((size_t)8 << c)) --parts;
    msm_shared sh;
    sh.P = (const g1a *)points; sh.S = (const fe *)scalars_mont; sh.n = n; sh.c = c; sh.nwin = nwin; sh.parts = parts;
    sh.K = (fe *)malloc((n ? n : 1) * sizeof(fe));
    sh.win_part = (g1j *)malloc((size_t)nwin * parts * sizeof(g1j));
    sh.next_job = 0; sh.next_conv = 0;
    pthread_t *th = (pthread_t *)malloc(threads * sizeof(pthread_t));
    for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, msm_conv_worker, &sh);
    msm_conv_worker(&sh);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, msm_worker, &sh);
    msm_worker(&sh);
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    g1j total; j_set_inf(&total);
    for (int w = nwin - 1; w >= 0; --w) {
        for (int b = 0; b < c; ++b) j_double(&total, &total);
        for (int part = 0; part < parts; ++part) j_add(&total, &total, &sh.win_part[(size_t)w * parts + part]);
    }
    *(g1j *)out_jac = total;
    free(sh.K); free(sh.win_part); free(th);
}

/* ------------------------------------------------------------------ NTT */
static void fr_root_of_unity(fe *w, uint64_t n) {   /* 5^((r-1)/n), Montgomery form */
    fe e = FR.m, one = {{1, 0, 0, 0}}, q = {{0, 0, 0, 0}};
    sub4(&e, &e, &one);
    u128 rem = 0;                                   /* q = (r-1) / n, long division by a word */
    for (int i = 3; i >= 0; --i) { u128 cur = (rem << 64) | e.l[i]; q.l[i] = (uint64_t)(cur / n); rem = cur % n; }
    fe g; f_from_u64(&FR, &g, 5);
    f_pow(&FR, w, &g, &q);
}
int oracle_domain_supported(uint64_t n) {
    if (n == 0) return 0;
    uint64_t m = (n % 3 == 0) ? n / 3 : n;
    return (m & (m - 1)) == 0 && m <= (1ULL << 28);
}
void oracle_root_of_unity(uint64_t n, uint64_t *out_mont) { fr_root_of_unity((fe *)out_mont, n); }

typedef struct { fe *a; const fe *tw; size_t n, half, step, lo, hi; } ntt_job;
static void *ntt_stage_worker(void *arg) {
    ntt_job *j = (ntt_job *)arg;
    /* butterflies b in [lo,hi): group = b / half, pos = b % half */
    for (size_t b = j->lo; b < j->hi; ++b) {
        size_t g = b / j->half, p = b % j->half;
        fe *u = &j->a[g * 2 * j->half + p], *v = u + j->half, t;
        f_mul(&FR, &t, v, &j->tw[p * j->step]);
        f_sub(&FR, v, u, &t); f_add(&FR, u, u, &t);
    }
    return NULL;
}
/* In-place radix-2 DIT on a power-of-two length with root w (bit-reverse first). */
static void ntt_pow2(fe *a, size_t n, const fe *w, int threads) {
    if (n <= 1) return;
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    fe *tw = (fe *)malloc((n / 2) * sizeof(fe));
    tw[0] = FR.r;
    for (size_t i = 1; i < n / 2; ++i) f_mul(&FR, &tw[i], &tw[i - 1], w);
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(threads * sizeof(pthread_t));
    ntt_job *jobs = (ntt_job *)malloc(threads * sizeof(ntt_job));
    for (size_t half = 1; half < n; half <<= 1) {
        size_t step = n / (2 * half), nb = n / 2;
        int T = (nb < 4096) ? 1 : threads;
        for (int t = 0; t < T; ++t) {
            jobs[t] = (ntt_job){a, tw, n, half, step, nb * t / T, nb * (t + 1) / T};
            if (T == 1) ntt_stage_worker(&jobs[t]); else pthread_create(&th[t], NULL, ntt_stage_worker, &jobs[t]);
        }
        if (T > 1) for (int t = 0; t < T; ++t) pthread_join(th[t], NULL);
    }
    free(tw); free(th); free(jobs);
}
/* `EvaluationDomain::fft` / `ifft` over the size-n domain (n = 2^k or 3*2^k), in place,
 * natural order in and out; `data` holds n Montgomery elements (caller zero-pads). */
int oracle_ntt(uint64_t *data, uint64_t n, int inverse, int threads) {
    if (!oracle_domain_supported(n)) return 1;
    fe *a = (fe *)data, w;
    fr_root_of_unity(&w, n);
    if (inverse) f_inv(&FR, &w, &w);
    if (n % 3 != 0) {
        ntt_pow2(a, n, &w, threads);
    } else {
        /* decimation in time by 3: x_k[j] = a[3j+k]; X[i] = sum_k w^(ik) X_k[i mod m] */
        size_t m = n / 3;
        fe *sub = (fe *)malloc(n * sizeof(fe)), w3, wi = FR.r;
        for (size_t j = 0; j < m; ++j) for (int k = 0; k < 3; ++k) sub[k * m + j] = a[3 * j + k];
        f_sqr(&FR, &w3, &w); f_mul(&FR, &w3, &w3, &w);
        for (int k = 0; k < 3; ++k) ntt_pow2(sub + k * m, m, &w3, threads);
        for (size_t i = 0; i < n; ++i) {
            fe t1, t2, wi2;
            f_sqr(&FR, &wi2, &wi);
            f_mul(&FR, &t1, &wi, &sub[m + i % m]); f_mul(&FR, &t2, &wi2, &sub[2 * m + i % m]);
            f_add(&FR, &a[i], &sub[i % m], &t1); f_add(&FR, &a[i], &a[i], &t2);
            f_mul(&FR, &wi, &wi, &w);
        }
        free(sub);
    }
    if (inverse) {
        fe ninv; f_from_u64(&FR, &ninv, n); f_inv(&FR, &ninv, &ninv);
        for (size_t i = 0; i < n; ++i) f_mul(&FR, &a[i], &a[i], &ninv);
    }
    return 0;
}
/* mul_var_assign (field_polynomial.rs:470-477): c_j *= k^j, serial like the reference. */
void oracle_mul_var(uint64_t *data, uint64_t len, const uint64_t *k_mont) {
    fe *a = (fe *)data, x = FR.r;
    for (uint64_t j = 0; j < len; ++j) { f_mul(&FR, &a[j], &a[j], &x); f_mul(&FR, &x, &x, (const fe *)k_mont); }
}
/* Horner evaluation (field_polynomial.rs:198-209). */
void oracle_poly_eval(const uint64_t *coefs, uint64_t len, const uint64_t *x_mont, uint64_t *out) {
    fe acc = {{0, 0, 0, 0}};
    for (uint64_t j = len; j-- > 0;) { f_mul(&FR, &acc, &acc, (const fe *)x_mont); f_add(&FR, &acc, &acc, &((const fe *)coefs)[j]); }
    *(fe *)out = acc;
}

/* z_poly (uzkge/src/plonk/helpers.rs:160-220), restated literally: per-row numerator/denominator,
 * batch inversion (Montgomery's trick, as ark_ff::batch_inversion), serial prefix product. */
void oracle_z_poly(const uint64_t *w_, const uint32_t *perm, const uint64_t *group_, const uint64_t *k_,
                   const uint64_t *beta_, const uint64_t *gamma_, uint32_t n, uint32_t n_wires, uint64_t *z_out) {
    const fe *w = (const fe *)w_, *group = (const fe *)group_, *k = (const fe *)k_;
    const fe *beta = (const fe *)beta_, *gamma = (const fe *)gamma_;
    fe *z = (fe *)z_out;
    if (n == 0) return;
    uint32_t m = n - 1;
    fe *num = (fe *)malloc((m ? m : 1) * sizeof(fe)), *den = (fe *)malloc((m ? m : 1) * sizeof(fe));
    for (uint32_t i = 0; i < m; ++i) {
        fe nm = FR.r, dn = FR.r;
        for (uint32_t j = 0; j < n_wires; ++j) {
            fe kx, t, a, px, b;
            f_mul(&FR, &kx, &k[j], &group[i]);
            const fe *fx = &w[(size_t)j * n + i];
            f_add(&FR, &a, fx, gamma); f_mul(&FR, &t, beta, &kx); f_add(&FR, &a, &a, &t);
            f_mul(&FR, &nm, &nm, &a);
            uint32_t pv = perm[(size_t)j * n + i];
            /* p_of_x: k[i] * group[perm % n] for the coset the index falls in (helpers.rs:174-182) */
            f_mul(&FR, &px, &k[pv / n], &group[pv % n]);
            f_add(&FR, &b, fx, gamma); f_mul(&FR, &t, beta, &px); f_add(&FR, &b, &b, &t);
            f_mul(&FR, &dn, &dn, &b);
        }
        num[i] = nm; den[i] = dn;
    }
    /* batch inversion of den */
    fe *pre = (fe *)malloc((m ? m : 1) * sizeof(fe)), acc = FR.r;
    for (uint32_t i = 0; i < m; ++i) { pre[i] = acc; f_mul(&FR, &acc, &acc, &den[i]); }
    fe inv; f_inv(&FR, &inv, &acc);
    for (uint32_t i = m; i-- > 0;) { fe t; f_mul(&FR, &t, &inv, &pre[i]); f_mul(&FR, &inv, &inv, &den[i]); den[i] = t; }
    fe prev = FR.r;
    z[0] = prev;
    for (uint32_t i = 0; i < m; ++i) { fe t; f_mul(&FR, &t, &num[i], &den[i]); f_mul(&FR, &prev, &prev, &t); z[i + 1] = prev; }
    free(num); free(den); free(pre);
}

/* t_poly's quotient evaluations on the coset (uzkge/src/plonk/helpers.rs:284-656, feature "shuffle"),
 * restated term by term in the reference's own order; the gate function is
 * uzkge/src/plonk/constraint_system/turbo/mod.rs:193-222.  vec[] = 56 vectors of m = factor * n
 * elements in the slot order of include/uzkge_gpu.h (UZK_TQ_*); out[point] = numerator * z_h_inv[point % factor].
 * No fixture of the reference pins these intermediate values: parity for this row is unpinned. */
typedef struct {
    uint32_t n, factor;
    const uint64_t *vec[56];
    uint64_t alpha[4], beta[4], gamma[4], k[5][4], anemoi_g[4], anemoi_g_inv[4], edwards_a[4];
    uint64_t z_h_inv[16][4];
} oracle_quotient_args;

static void f_pow5(const field *F, fe *r, const fe *a) { fe a2, a4; f_sqr(F, &a2, a); f_sqr(F, &a4, &a2); f_mul(F, r, &a4, a); }

void oracle_t_quotient(const oracle_quotient_args *A, uint64_t *out_) {
    const field *F = &FR;
    const size_t m = (size_t)A->n * A->factor, factor = A->factor;
    fe *out = (fe *)out_;
#define V(slot, p) ((const fe *)A->vec[slot] + (p))
    const fe *alpha = (const fe *)A->alpha, *beta = (const fe *)A->beta, *gamma = (const fe *)A->gamma;
    const fe *g = (const fe *)A->anemoi_g, *g_inv = (const fe *)A->anemoi_g_inv, *ea = (const fe *)A->edwards_a;
    fe ap[17];                      /* alpha^1 .. alpha^16 */
    ap[1] = *alpha;
    for (int i = 2; i <= 16; ++i) f_mul(F, &ap[i], &ap[i - 1], alpha);
    const fe one = F->r;
    fe g2p1; f_sqr(F, &g2p1, g); f_add(F, &g2p1, &g2p1, &one);
    for (size_t point = 0; point < m; ++point) {
        const size_t nxt = (point + factor) % m;
        fe w[5], ws[3];
        for (int j = 0; j < 5; ++j) w[j] = *V(0 + j, point);
        for (int j = 0; j < 3; ++j) ws[j] = *V(5 + j, point);
        const fe pi = *V(8, point), z = *V(9, point), zn = *V(9, nxt);
        const fe w0n = *V(0, nxt), w1n = *V(1, nxt), w2n = *V(2, nxt);
        fe t, u, v;
        /* term1: gate function (turbo/mod.rs:197-221) */
        fe term1, acc;
        f_mul(F, &term1, V(10, point), &w[0]);
        f_mul(F, &t, V(11, point), &w[1]); f_add(F, &term1, &term1, &t);
        f_mul(F, &t, V(12, point), &w[2]); f_add(F, &term1, &term1, &t);
        f_mul(F, &t, V(13, point), &w[3]); f_add(F, &term1, &term1, &t);
        f_mul(F, &u, &w[0], &w[1]); f_mul(F, &t, V(14, point), &u); f_add(F, &term1, &term1, &t);
        f_mul(F, &u, &w[2], &w[3]); f_mul(F, &t, V(15, point), &u); f_add(F, &term1, &term1, &t);
        f_add(F, &t, V(16, point), &pi); f_add(F, &term1, &term1, &t);
        f_mul(F, &t, V(17, point), &w[0]); f_mul(F, &t, &t, &w[1]); f_mul(F, &t, &t, &w[2]); f_mul(F, &t, &t, &w[3]);
        f_mul(F, &t, &t, &w[4]); f_add(F, &term1, &term1, &t);
        f_mul(F, &t, V(18, point), &w[4]); f_sub(F, &term1, &term1, &t);
        /* term2 (helpers.rs:300-307) */
        fe term2; f_mul(F, &term2, alpha, &z);
        for (int j = 0; j < 5; ++j) {
            f_mul(F, &t, (const fe *)A->k[j], V(30, point)); f_mul(F, &t, beta, &t);
            f_add(F, &u, &w[j], gamma); f_add(F, &u, &u, &t);
            f_mul(F, &term2, &term2, &u);
        }
        /* term3 (:310-319) */
        fe term3; f_mul(F, &term3, alpha, &zn);
        for (int j = 0; j < 5; ++j) {
            f_mul(F, &t, beta, V(19 + j, point));
            f_add(F, &u, &w[j], gamma); f_add(F, &u, &u, &t);
            f_mul(F, &term3, &term3, &u);
        }
        /* term4 (:322-324) */
        fe term4; f_mul(F, &term4, &ap[2], V(24, point)); f_sub(F, &t, &z, &one); f_mul(F, &term4, &term4, &t);
        /* term5..7 (:328-350) */
        fe term5, term6, term7;
        const fe qb = *V(25, point);
        f_mul(F, &term5, &ap[3], &qb); f_mul(F, &term5, &term5, &w[1]); f_sub(F, &t, &w[1], &one); f_mul(F, &term5, &term5, &t);
        f_mul(F, &term6, &ap[4], &qb); f_mul(F, &term6, &term6, &w[2]); f_sub(F, &t, &w[2], &one); f_mul(F, &term6, &term6, &t);
        f_mul(F, &term7, &ap[5], &qb); f_mul(F, &term7, &term7, &w[3]); f_sub(F, &t, &w[3], &one); f_mul(F, &term7, &term7, &t);
        /* Anemoi terms 8..11 (:352-417) */
        const fe prk1 = *V(26, point), prk2 = *V(27, point), prk3 = *V(28, point), prk4 = *V(29, point);
        fe w3w0, w2w1, w3_2w0, w2_2w1, tmp, d5, sq, rhs;
        f_add(F, &w3w0, &w[0], &w[3]); f_add(F, &w2w1, &w[1], &w[2]);
        f_add(F, &w3_2w0, &w[0], &w3w0); f_add(F, &w2_2w1, &w[1], &w2w1);
        f_mul(F, &t, g, &w2w1); f_add(F, &tmp, &w3w0, &t); f_add(F, &tmp, &tmp, &prk3);
        fe term8, term9, term10, term11;
        f_sub(F, &t, &tmp, &w2n); f_pow5(F, &d5, &t);
        f_sqr(F, &sq, &tmp); f_mul(F, &sq, g, &sq);
        f_mul(F, &t, g, &w2_2w1); f_add(F, &rhs, &w3_2w0, &t); f_add(F, &rhs, &rhs, &prk1);
        f_add(F, &u, &d5, &sq); f_sub(F, &u, &u, &rhs);
        f_mul(F, &term8, &ap[6], &prk3); f_mul(F, &term8, &term8, &u);
        f_sqr(F, &sq, &w2n); f_mul(F, &sq, g, &sq);
        f_add(F, &u, &d5, &sq); f_add(F, &u, &u, g_inv); f_sub(F, &u, &u, &w0n);
        f_mul(F, &term10, &ap[8], &prk3); f_mul(F, &term10, &term10, &u);
        f_mul(F, &t, g, &w3w0); f_mul(F, &u, &g2p1, &w2w1); f_add(F, &tmp, &t, &u); f_add(F, &tmp, &tmp, &prk4);
        f_sub(F, &t, &tmp, &w[4]); f_pow5(F, &d5, &t);
        f_sqr(F, &sq, &tmp); f_mul(F, &sq, g, &sq);
        f_mul(F, &t, g, &w3_2w0); f_mul(F, &u, &g2p1, &w2_2w1); f_add(F, &rhs, &t, &u); f_add(F, &rhs, &rhs, &prk2);
        f_add(F, &u, &d5, &sq); f_sub(F, &u, &u, &rhs);
        f_mul(F, &term9, &ap[7], &prk3); f_mul(F, &term9, &term9, &u);
        f_sqr(F, &sq, &w[4]); f_mul(F, &sq, g, &sq);
        f_add(F, &u, &d5, &sq); f_add(F, &u, &u, g_inv); f_sub(F, &u, &u, &w1n);
        f_mul(F, &term11, &ap[9], &prk3); f_mul(F, &term11, &term11, &u);
        /* shuffle terms 12..18 (:419-627) */
        const fe qecc = *V(55, point);
        fe sel[4], om0, om1;
        f_sub(F, &om0, &one, &ws[0]); f_sub(F, &om1, &one, &ws[1]);
        f_mul(F, &sel[0], &om0, &om1); f_add(F, &sel[0], &sel[0], &qecc); f_sub(F, &sel[0], &sel[0], &one);
        f_mul(F, &sel[1], &ws[0], &om1);
        f_mul(F, &sel[2], &om0, &ws[1]);
        f_mul(F, &sel[3], &ws[0], &ws[1]);
        fe term12 = {{0, 0, 0, 0}}, term13 = term12, term14 = term12, term15 = term12;
        for (int ab = 0; ab < 4; ++ab) {
            const fe pkx = *V(31 + ab, point), pky = *V(35 + ab, point), pkd = *V(39 + ab, point);
            const fe gx = *V(43 + ab, point), gy = *V(47 + ab, point), gd = *V(51 + ab, point);
            fe e;
            /* 12: ws2*w0n - ws2*w0*pky - w1*pkx + w0*w1*w0n*pkd */
            f_mul(F, &e, &ws[2], &w0n);
            f_mul(F, &t, &ws[2], &w[0]); f_mul(F, &t, &t, &pky); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[1], &pkx); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[0], &w[1]); f_mul(F, &t, &t, &w0n); f_mul(F, &t, &t, &pkd); f_add(F, &e, &e, &t);
            f_mul(F, &e, &sel[ab], &e); f_add(F, &term12, &term12, &e);
            /* 13: ws2*w1n + w0*a*pkx - ws2*w1*pky - w0*w1*w1n*pkd */
            f_mul(F, &e, &ws[2], &w1n);
            f_mul(F, &t, &w[0], ea); f_mul(F, &t, &t, &pkx); f_add(F, &e, &e, &t);
            f_mul(F, &t, &ws[2], &w[1]); f_mul(F, &t, &t, &pky); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[0], &w[1]); f_mul(F, &t, &t, &w1n); f_mul(F, &t, &t, &pkd); f_sub(F, &e, &e, &t);
            f_mul(F, &e, &sel[ab], &e); f_add(F, &term13, &term13, &e);
            /* 14: ws2*w2n - ws2*w2*gy - w3*gx + w2*w3*w2n*gd */
            f_mul(F, &e, &ws[2], &w2n);
            f_mul(F, &t, &ws[2], &w[2]); f_mul(F, &t, &t, &gy); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[3], &gx); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[2], &w[3]); f_mul(F, &t, &t, &w2n); f_mul(F, &t, &t, &gd); f_add(F, &e, &e, &t);
            f_mul(F, &e, &sel[ab], &e); f_add(F, &term14, &term14, &e);
            /* 15: ws2*w4 + w2*a*gx - ws2*w3*gy - w2*w3*w4*gd */
            f_mul(F, &e, &ws[2], &w[4]);
            f_mul(F, &t, &w[2], ea); f_mul(F, &t, &t, &gx); f_add(F, &e, &e, &t);
            f_mul(F, &t, &ws[2], &w[3]); f_mul(F, &t, &t, &gy); f_sub(F, &e, &e, &t);
            f_mul(F, &t, &w[2], &w[3]); f_mul(F, &t, &t, &w[4]); f_mul(F, &t, &t, &gd); f_sub(F, &e, &e, &t);
            f_mul(F, &e, &sel[ab], &e); f_add(F, &term15, &term15, &e);
        }
        f_mul(F, &term12, &ap[10], &term12); f_mul(F, &term13, &ap[11], &term13);
        f_mul(F, &term14, &ap[12], &term14); f_mul(F, &term15, &ap[13], &term15);
        fe term16, term17, term18, omq;
        f_sub(F, &omq, &one, &qecc);
        f_mul(F, &t, &qecc, &ws[0]); f_mul(F, &t, &t, &om0); f_mul(F, &u, &omq, &ws[0]); f_add(F, &t, &t, &u); f_mul(F, &term16, &ap[14], &t);
        f_mul(F, &t, &qecc, &ws[1]); f_mul(F, &t, &t, &om1); f_mul(F, &u, &omq, &ws[1]); f_add(F, &t, &t, &u); f_mul(F, &term17, &ap[15], &t);
        f_add(F, &t, &one, &ws[2]); f_sub(F, &u, &one, &ws[2]); f_mul(F, &v, &qecc, &t); f_mul(F, &v, &v, &u); f_mul(F, &term18, &ap[16], &v);
        /* numerator (:629-652) */
        acc = term1;
        f_add(F, &acc, &acc, &term2);
        f_sub(F, &t, &term4, &term3); f_add(F, &acc, &acc, &t);
        f_add(F, &acc, &acc, &term5); f_add(F, &acc, &acc, &term6); f_add(F, &acc, &acc, &term7);
        f_sub(F, &acc, &acc, &term8); f_sub(F, &acc, &acc, &term9); f_sub(F, &acc, &acc, &term10); f_sub(F, &acc, &acc, &term11);
        f_add(F, &acc, &acc, &term12); f_add(F, &acc, &acc, &term13); f_add(F, &acc, &acc, &term14); f_add(F, &acc, &acc, &term15);
        f_add(F, &acc, &acc, &term16); f_add(F, &acc, &acc, &term17); f_add(F, &acc, &acc, &term18);
        f_mul(F, &out[point], &acc, (const fe *)A->z_h_inv[point % factor]);
    }
#undef V
}

/* z_h_inv_coset_evals (helpers.rs:242-252): 1 / (k1^n * (g_m^n)^i - 1), i < factor */
void oracle_z_h_inv(const uint64_t *k1_mont, uint32_t n, uint32_t factor, uint64_t *out) {
    const field *F = &FR;
    fe gm; fr_root_of_unity(&gm, (uint64_t)n * factor);
    fe e = {{n, 0, 0, 0}}, gn, mult;
    f_pow(F, &gn, &gm, &e);
    f_pow(F, &mult, (const fe *)k1_mont, &e);
    for (uint32_t i = 0; i < factor; ++i) {
        fe t; f_sub(F, &t, &mult, &F->r);
        f_inv(F, (fe *)out + i, &t);
        f_mul(F, &mult, &mult, &gn);
    }
}

/* batch_prove's polynomial work (uzkge/src/poly_commit/pcs.rs:119-135), restated literally:
 * for each poly: eval at the point (Horner, field_polynomial.rs:198-209), subtract the value, scale by
 * the running power of alpha, accumulate into h; then div_rem by X - z (field_polynomial.rs:519-550,
 * general long division specialised to the two-coefficient divisor {-z, 1}).
 * polys: batch x n coefficients; q_out: n elements (n - 1 quotient coefficients, then zero);
 * evals_out: batch elements.  Returns 1 if the remainder is zero (PCSProveEvalError otherwise). */
int oracle_open_quotient(const uint64_t *polys_, uint64_t n, uint32_t batch, const uint64_t *z_, const uint64_t *alpha_,
                         uint64_t *q_out, uint64_t *evals_out) {
    const field *F = &FR;
    const fe *polys = (const fe *)polys_, *z = (const fe *)z_, *alpha = (const fe *)alpha_;
    fe *q = (fe *)q_out, *ev = (fe *)evals_out;
    fe *h = (fe *)calloc(n, sizeof(fe));
    fe mult = F->r;
    for (uint32_t k = 0; k < batch; ++k) {
        const fe *p = polys + (size_t)k * n;
        fe e = {{0, 0, 0, 0}};
        for (uint64_t j = n; j-- > 0;) { f_mul(F, &e, &e, z); f_add(F, &e, &e, &p[j]); }
        ev[k] = e;
        for (uint64_t j = 0; j < n; ++j) {
            fe c = p[j];
            if (j == 0) f_sub(F, &c, &c, &e);
            f_mul(F, &c, &c, &mult);
            f_add(F, &h[j], &h[j], &c);
        }
        f_mul(F, &mult, &mult, alpha);
    }
    /* div_rem with divisor coefs d = {-z, 1}: bl_inv = 1; for i = k-l .. 0: qi = rem[i+1]; rem[i] -= qi*(-z); rem[i+1] -= qi */
    fe nz; f_neg(F, &nz, z);
    for (uint64_t i = 0; i < n; ++i) q[i] = (fe){{0, 0, 0, 0}};
    for (uint64_t i = n - 1; i-- > 0;) {
        fe qi = h[i + 1], a;
        f_mul(F, &a, &qi, &nz); f_sub(F, &h[i], &h[i], &a);
        f_sub(F, &h[i + 1], &h[i + 1], &qi);
        q[i] = qi;
    }
    int ok = fe_is_zero(&h[0]);
    free(h);
    return ok;
}
