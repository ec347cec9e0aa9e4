/*
 * bn254_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the barretenberg (AztecProtocol/aztec-2.0) algorithms on the
 * MSM + NTT hot path.  It is the CHECKER for the HIP kernels and the "port" CPU baseline of
 * bench.py; it is never linked into, imported by or called from the product library
 * (aztec-2.0_amd/csrc/libbbg.so).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Parity pinning: every function below is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by the REAL reference code compiled from /root/reference
 * (oracle/ref_driver.cpp -> oracle/_ref/libbbref.so, generator tests/golden/gen_golden.py)
 * and against the known-answer constants in the reference's own unit tests
 * (fr.test.cpp:50-87, fq.test.cpp:71-165, g1.test.cpp:39-121,284-299).
 *
 * Reference paths are relative to /root/reference/barretenberg/src/aztec/ ("B/").
 *
 * Representation: 4 x u64 little-endian limbs, Montgomery form with R = 2^256, exactly the
 * reference's `fr` / `fq` memory layout (B/ecc/fields/field.hpp:24,86).  The reference keeps
 * values coarsely reduced in [0,2p); this file keeps them strictly in [0,p) internally and
 * accepts any 256-bit input -- comparisons are therefore made on canonical values, which is
 * also what the reference's operator== does (B/ecc/fields/field_impl.hpp:221-227).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* OpenMP team size of every oracle routine: a PRIVATE cap applied with num_threads() clauses.  The restatement is a checker, not a
 * benchmark -- on hosts with hundreds of hardware threads an uncapped team spends its time in barriers (a 2^19 NTT took 20 s with 256
 * threads against 0.3 s with 16) -- and the process-wide OpenMP setting must stay untouched: the compiled reference shares the
 * runtime and its CPU pippenger misbehaves when its team size changes under it. */
static int g_oracle_team = 32;
static int oracle_team(void)
{
#ifdef _OPENMP
    const int m = omp_get_max_threads();
    return m < g_oracle_team ? m : g_oracle_team;
#else
    return 1;
#endif
}

typedef unsigned __int128 u128;
typedef struct { uint64_t d[4]; } fe;

typedef struct {
    uint64_t mod[4];
    uint64_t r2[4];  /* R^2 mod p */
    uint64_t one[4]; /* R mod p   */
    uint64_t inv;    /* -p^-1 mod 2^64 */
} fparams;

/* B/ecc/curves/bn254/fr.hpp:12-20,42 */
static const fparams FR = {
    { 0x43E1F593F0000001ULL, 0x2833E84879B97091ULL, 0xB85045B68181585DULL, 0x30644E72E131A029ULL },
    { 0x1BB8E645AE216DA7ULL, 0x53FE3AB1E35C59E3ULL, 0x8C49833D53BB8085ULL, 0x0216D0B17F4E44A5ULL },
    { 0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL },
    0xc2e1f593efffffffULL
};
/* B/ecc/curves/bn254/fq.hpp:11-19,41 */
static const fparams FQ = {
    { 0x3C208C16D87CFD47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL },
    { 0xF32CFC5B538AFA89ULL, 0xB5E71911D44501FBULL, 0x47AB1EFF0A417FF6ULL, 0x06D89F71CAB8351FULL },
    { 0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL },
    0x87d20782e4866389ULL
};

/* ------------------------------------------------------------------ limbs */
static int ge4(const uint64_t* a, const uint64_t* b)
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static uint64_t sub4(uint64_t* r, const uint64_t* a, const uint64_t* b)
{
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)x;
        br = (uint64_t)(x >> 64) & 1;
    }
    return br;
}
static uint64_t add4(uint64_t* r, const uint64_t* a, const uint64_t* b)
{
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)a[i] + b[i] + c;
        r[i] = (uint64_t)x;
        c = (uint64_t)(x >> 64);
    }
    return c;
}
static int is_zero4(const uint64_t* a) { return (a[0] | a[1] | a[2] | a[3]) == 0; }

/* any 256-bit value -> [0,p)   (generalises reduce_once, B/ecc/fields/field_impl.hpp:100-112) */
static fe fe_canon(const fparams* P, fe a)
{
    while (ge4(a.d, P->mod)) sub4(a.d, a.d, P->mod);
    return a;
}

/* Montgomery product, CIOS on 64-bit limbs (B/ecc/fields/field_impl_generic.hpp:392-442). in,out < p */
static fe fe_mul(const fparams* P, fe a, fe b)
{
    uint64_t t[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) {
            u128 x = (u128)a.d[j] * b.d[i] + t[j] + c;
            t[j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        u128 y = (u128)t[4] + c;
        t[4] = (uint64_t)y;
        t[5] = (uint64_t)(y >> 64);
        uint64_t m = t[0] * P->inv;
        u128 x = (u128)m * P->mod[0] + t[0];
        c = (uint64_t)(x >> 64);
        for (int j = 1; j < 4; j++) {
            x = (u128)m * P->mod[j] + t[j] + c;
            t[j - 1] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        y = (u128)t[4] + c;
        t[3] = (uint64_t)y;
        t[4] = t[5] + (uint64_t)(y >> 64);
    }
    fe r = { { t[0], t[1], t[2], t[3] } };
    if (t[4] || ge4(r.d, P->mod)) sub4(r.d, r.d, P->mod);
    return r;
}
static fe fe_sqr(const fparams* P, fe a) { return fe_mul(P, a, a); }
static fe fe_add(const fparams* P, fe a, fe b)
{
    fe r;
    uint64_t c = add4(r.d, a.d, b.d);
    if (c || ge4(r.d, P->mod)) sub4(r.d, r.d, P->mod);
    return r;
}
static fe fe_sub(const fparams* P, fe a, fe b)
{
    fe r;
    if (sub4(r.d, a.d, b.d)) add4(r.d, r.d, P->mod);
    return r;
}
static fe fe_neg(const fparams* P, fe a)
{
    fe r = { { 0, 0, 0, 0 } };
    if (is_zero4(a.d)) return r;
    sub4(r.d, P->mod, a.d);
    return r;
}
static fe fe_one(const fparams* P) { fe r; memcpy(r.d, P->one, 32); return r; }
static fe fe_zero(void) { fe r = { { 0, 0, 0, 0 } }; return r; }
static fe fe_to_mont(const fparams* P, fe a) { fe r2; memcpy(r2.d, P->r2, 32); return fe_mul(P, fe_canon(P, a), r2); }
static fe fe_from_mont(const fparams* P, fe a) { fe o = { { 1, 0, 0, 0 } }; return fe_mul(P, a, o); }
static int fe_eq(fe a, fe b) { return memcmp(a.d, b.d, 32) == 0; }

/* a^e, e = 4-limb plain integer (B/ecc/fields/field_impl.hpp:285-318 pow) */
static fe fe_pow(const fparams* P, fe a, const uint64_t* e)
{
    fe acc = fe_one(P);
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr(P, acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = fe_mul(P, acc, a);
    }
    return acc;
}
static fe fe_pow64(const fparams* P, fe a, uint64_t e)
{
    uint64_t ee[4] = { e, 0, 0, 0 };
    return fe_pow(P, a, ee);
}
/* a^(p-2) (B/ecc/fields/field_impl.hpp:323-329 invert) */
static fe fe_inv(const fparams* P, fe a)
{
    uint64_t e[4], two[4] = { 2, 0, 0, 0 };
    sub4(e, P->mod, two);
    return fe_pow(P, a, e);
}

/* ------------------------------------------------------------ Fr constants */
/* B/ecc/curves/bn254/fr.hpp:27-30 primitive 2^28-th root of unity, Montgomery form */
static const fe FR_PRIMITIVE_ROOT = { { 0x636e735580d13d9cULL, 0xa22bf3742445ffd6ULL, 0x56452ac01eb203d8ULL,
                                        0x1860ef942963f9e7ULL } };
/* B/ecc/curves/bn254/fr.hpp:22-25 cube root of unity lambda, Montgomery form */
static const fe FR_CUBE_ROOT = { { 0x93e7cede4a0329b3ULL, 0x7d4fdca77a96c167ULL, 0x8be4ba08b19a750aULL,
                                   0x1cbd5653a5661c25ULL } };
/* B/ecc/curves/bn254/fq.hpp:21-24 cube root of unity beta, Montgomery form */
static const fe FQ_CUBE_ROOT = { { 0x71930c11d782e155ULL, 0xa6bb947cffbe3323ULL, 0xaa303344d4741444ULL,
                                   0x2c3b3f0d26594943ULL } };

/* omega_n for n = 2^k: square the 2^28-th root 28-k times (B/ecc/fields/field_impl.hpp:496-503) */
static fe fr_root_of_unity(unsigned log2n)
{
    fe r = FR_PRIMITIVE_ROOT;
    for (unsigned i = 28; i > log2n; i--) r = fe_sqr(&FR, r);
    return r;
}
/* coset generator 0 == 5 (B/ecc/curves/bn254/fr.hpp:44-59 column 0; field.hpp:127-137) */
static fe fr_coset_generator(void)
{
    fe five = { { 5, 0, 0, 0 } };
    return fe_to_mont(&FR, five);
}

/* ------------------------------------------------------- GLV endomorphism */
static void mul_512(const uint64_t* a, const uint64_t* b, uint64_t* r)
{
    memset(r, 0, 64);
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) {
            u128 x = (u128)a[j] * b[i] + r[i + j] + c;
            r[i + j] = (uint64_t)x;
            c = (uint64_t)(x >> 64);
        }
        r[i + 4] = c;
    }
}
/*
 * k (NON-Montgomery, any 256-bit rep) -> k1, k2 < 2^128 with k = k1 - k2*lambda (mod r).
 * Restates field::split_into_endomorphism_scalars, B/ecc/fields/field.hpp:236-282, with the
 * Fr constants of B/ecc/curves/bn254/fr.hpp:32-40.  NOTE the reference multiplies t1 (a plain
 * integer) by the Montgomery-form cube root with a Montgomery product, which yields the plain
 * product t1*lambda -- reproduced here.
 */
static void fr_split_endo(fe k, uint64_t k1[2], uint64_t k2[2])
{
    static const uint64_t g1[4] = { 0x7a7bd9d4391eb18dULL, 0x4ccef014a773d2cfULL, 0x0000000000000002ULL, 0 };
    static const uint64_t g2[4] = { 0xd91d232ec7e0b3d7ULL, 0x0000000000000002ULL, 0, 0 };
    static const uint64_t minus_b1[4] = { 0x8211bbeb7d4f1128ULL, 0x6f4d8248eeb859fcULL, 0, 0 };
    static const uint64_t b2[4] = { 0x89d3256894d213e3ULL, 0, 0, 0 };
    fe input = fe_canon(&FR, k);
    uint64_t c1[8], c2[8], q1[8], q2[8];
    mul_512(g2, input.d, c1);
    mul_512(g1, input.d, c2);
    mul_512(c1 + 4, minus_b1, q1);
    mul_512(c2 + 4, b2, q2);
    fe q1lo = { { q1[0], q1[1], q1[2], q1[3] } }, q2lo = { { q2[0], q2[1], q2[2], q2[3] } };
    fe t1 = fe_sub(&FR, fe_canon(&FR, q2lo), fe_canon(&FR, q1lo));
    fe t2 = fe_add(&FR, fe_mul(&FR, t1, FR_CUBE_ROOT), input);
    k2[0] = t1.d[0]; k2[1] = t1.d[1];
    k1[0] = t2.d[0]; k1[1] = t2.d[1];
}

/* ---------------------------------------------------------------- G1 group */
/* y^2 = x^3 + 3, generator (1,2) (B/ecc/curves/bn254/g1.hpp:8-19).  Infinity is flagged by
 * bit 63 of x.d[3] (B/ecc/groups/element_impl.hpp:497-516, affine_element_impl.hpp:74-93). */
typedef struct { fe x, y; } g1_affine;
typedef struct { fe x, y, z; } g1_jac;
#define INF_BIT (1ULL << 63)

static int aff_is_inf(const g1_affine* p) { return (p->x.d[3] & INF_BIT) != 0; }
static int jac_is_inf(const g1_jac* p) { return (p->x.d[3] & INF_BIT) != 0; }
static void jac_set_inf(g1_jac* p) { memset(p, 0, sizeof(*p)); p->x.d[3] = INF_BIT; }
static void aff_set_inf(g1_affine* p) { memset(p, 0, sizeof(*p)); p->x.d[3] = INF_BIT; }

/* 2P, Jacobian, a = 0 (B/ecc/groups/element_impl.hpp:70-139 self_dbl) */
static void jac_dbl(g1_jac* r, const g1_jac* p)
{
    const fparams* F = &FQ;
    if (jac_is_inf(p) || is_zero4(p->y.d)) { jac_set_inf(r); return; }
    fe A = fe_sqr(F, p->x), B = fe_sqr(F, p->y), C = fe_sqr(F, B);
    fe D = fe_sub(F, fe_sqr(F, fe_add(F, p->x, B)), fe_add(F, A, C));
    D = fe_add(F, D, D);
    fe E = fe_add(F, fe_add(F, A, A), A);
    fe Fq_ = fe_sqr(F, E);
    fe X3 = fe_sub(F, Fq_, fe_add(F, D, D));
    fe C8 = fe_add(F, C, C); C8 = fe_add(F, C8, C8); C8 = fe_add(F, C8, C8);
    fe Y3 = fe_sub(F, fe_mul(F, E, fe_sub(F, D, X3)), C8);
    fe Z3 = fe_mul(F, p->y, p->z); Z3 = fe_add(F, Z3, Z3);
    r->x = X3; r->y = Y3; r->z = Z3;
}
/* P + Q, Q affine (B/ecc/groups/element_impl.hpp:243-330 operator+=(affine)) incl. edge cases */
static void jac_madd(g1_jac* r, const g1_jac* p, const g1_affine* q)
{
    const fparams* F = &FQ;
    if (aff_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) { r->x = q->x; r->y = q->y; r->z = fe_one(F); return; }
    fe Z1Z1 = fe_sqr(F, p->z);
    fe U2 = fe_mul(F, q->x, Z1Z1);
    fe S2 = fe_mul(F, fe_mul(F, q->y, p->z), Z1Z1);
    fe H = fe_sub(F, U2, p->x);
    fe Rr = fe_sub(F, S2, p->y);
    if (is_zero4(H.d)) {
        if (is_zero4(Rr.d)) { g1_jac t = *p; jac_dbl(r, &t); return; }
        jac_set_inf(r); return;
    }
    fe HH = fe_sqr(F, H), HHH = fe_mul(F, H, HH), V = fe_mul(F, p->x, HH);
    fe X3 = fe_sub(F, fe_sub(F, fe_sqr(F, Rr), HHH), fe_add(F, V, V));
    fe Y3 = fe_sub(F, fe_mul(F, Rr, fe_sub(F, V, X3)), fe_mul(F, p->y, HHH));
    fe Z3 = fe_mul(F, p->z, H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
/* P + Q, both Jacobian (B/ecc/groups/element_impl.hpp:354-441 operator+=(element)) */
static void jac_add(g1_jac* r, const g1_jac* p, const g1_jac* q)
{
    const fparams* F = &FQ;
    if (jac_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) { *r = *q; return; }
    fe Z1Z1 = fe_sqr(F, p->z), Z2Z2 = fe_sqr(F, q->z);
    fe U1 = fe_mul(F, p->x, Z2Z2), U2 = fe_mul(F, q->x, Z1Z1);
    fe S1 = fe_mul(F, fe_mul(F, p->y, q->z), Z2Z2), S2 = fe_mul(F, fe_mul(F, q->y, p->z), Z1Z1);
    fe H = fe_sub(F, U2, U1), Rr = fe_sub(F, S2, S1);
    if (is_zero4(H.d)) {
        if (is_zero4(Rr.d)) { g1_jac t = *p; jac_dbl(r, &t); return; }
        jac_set_inf(r); return;
    }
    fe HH = fe_sqr(F, H), HHH = fe_mul(F, H, HH), V = fe_mul(F, U1, HH);
    fe X3 = fe_sub(F, fe_sub(F, fe_sqr(F, Rr), HHH), fe_add(F, V, V));
    fe Y3 = fe_sub(F, fe_mul(F, Rr, fe_sub(F, V, X3)), fe_mul(F, S1, HHH));
    fe Z3 = fe_mul(F, fe_mul(F, p->z, q->z), H);
    r->x = X3; r->y = Y3; r->z = Z3;
}
/* Jacobian -> affine (B/ecc/groups/element_impl.hpp:51-68) */
static void jac_to_affine(g1_affine* r, const g1_jac* p)
{
    const fparams* F = &FQ;
    if (jac_is_inf(p)) { aff_set_inf(r); return; }
    fe zi = fe_inv(F, p->z), zi2 = fe_sqr(F, zi), zi3 = fe_mul(F, zi2, zi);
    r->x = fe_mul(F, p->x, zi2);
    r->y = fe_mul(F, p->y, zi3);
}
static g1_affine aff_canon(const g1_affine* p)
{
    g1_affine r;
    if (aff_is_inf(p)) { aff_set_inf(&r); return r; }
    r.x = fe_canon(&FQ, p->x);
    r.y = fe_canon(&FQ, p->y);
    return r;
}
static g1_affine aff_neg(const g1_affine* p)
{
    g1_affine r = *p;
    if (!aff_is_inf(p)) r.y = fe_neg(&FQ, p->y);
    return r;
}
/* k*P by plain double-and-add; k is a NON-Montgomery canonical scalar.  The reference uses
 * GLV + wNAF (B/ecc/groups/element_impl.hpp:593-680); the group element is the same. */
static void jac_mul(g1_jac* r, const g1_affine* p, const uint64_t* k)
{
    g1_jac acc;
    jac_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        g1_jac t;
        jac_dbl(&t, &acc);
        acc = t;
        if ((k[i >> 6] >> (i & 63)) & 1) { jac_madd(&t, &acc, p); acc = t; }
    }
    *r = acc;
}

/* ============================================================ exported API
 * All arrays are uint64 little-endian limb arrays in the reference's memory layout. */

/* --- field known-answer helpers: which = 0 Fr, 1 Fq.  Inputs any 256-bit rep, outputs canonical */
static const fparams* PF(int which) { return which ? &FQ : &FR; }
void oracle_fe_mul(int which, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x, y; memcpy(x.d, a + 4 * i, 32); memcpy(y.d, b + 4 * i, 32);
        fe z = fe_mul(P, fe_canon(P, x), fe_canon(P, y));
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_add(int which, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x, y; memcpy(x.d, a + 4 * i, 32); memcpy(y.d, b + 4 * i, 32);
        fe z = fe_add(P, fe_canon(P, x), fe_canon(P, y));
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_sub(int which, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x, y; memcpy(x.d, a + 4 * i, 32); memcpy(y.d, b + 4 * i, 32);
        fe z = fe_sub(P, fe_canon(P, x), fe_canon(P, y));
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_inv(int which, const uint64_t* a, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x; memcpy(x.d, a + 4 * i, 32);
        fe z = fe_inv(P, fe_canon(P, x));
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_to_mont(int which, const uint64_t* a, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x; memcpy(x.d, a + 4 * i, 32);
        fe z = fe_to_mont(P, x);
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_from_mont(int which, const uint64_t* a, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x; memcpy(x.d, a + 4 * i, 32);
        fe z = fe_from_mont(P, fe_canon(P, x));
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fe_canon(int which, const uint64_t* a, uint64_t* r, size_t n)
{
    const fparams* P = PF(which);
    for (size_t i = 0; i < n; i++) {
        fe x; memcpy(x.d, a + 4 * i, 32);
        fe z = fe_canon(P, x);
        memcpy(r + 4 * i, z.d, 32);
    }
}
void oracle_fr_root_of_unity(unsigned log2n, uint64_t* r) { fe w = fr_root_of_unity(log2n); memcpy(r, w.d, 32); }

/* scalars (Montgomery, as the MSM receives them) -> k1,k2; out[i] = {k1.lo,k1.hi,k2.lo,k2.hi}
 * == what compute_wnaf_states feeds to the wNAF (B/ecc/curves/bn254/scalar_multiplication/scalar_multiplication.cpp:223-225) */
void oracle_endo_split(const uint64_t* scalars_mont, uint64_t* out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fe s; memcpy(s.d, scalars_mont + 4 * i, 32);
        fe k = fe_from_mont(&FR, fe_canon(&FR, s));
        fr_split_endo(k, out + 4 * i, out + 4 * i + 2);
    }
}

/* --- group helpers.  Points: affine Montgomery (x||y, 8 limbs). Results: canonical Montgomery affine;
 * infinity = x.d[3] bit 63 set and everything else 0. */
void oracle_g1_generator(uint64_t* out)
{
    g1_affine g;
    g.x = fe_one(&FQ);
    fe two = { { 2, 0, 0, 0 } };
    g.y = fe_to_mont(&FQ, two);
    memcpy(out, &g, 64);
}
int oracle_g1_on_curve(const uint64_t* p)
{
    g1_affine a; memcpy(&a, p, 64);
    if (aff_is_inf(&a)) return 1;
    a = aff_canon(&a);
    fe three = { { 3, 0, 0, 0 } };
    fe rhs = fe_add(&FQ, fe_mul(&FQ, fe_sqr(&FQ, a.x), a.x), fe_to_mont(&FQ, three));
    return fe_eq(fe_sqr(&FQ, a.y), rhs);
}
/* out = k*P, k Montgomery-form Fr (like element::operator*(fr), B/ecc/groups/element_impl.hpp:682-690) */
void oracle_g1_mul(const uint64_t* p, const uint64_t* k_mont, uint64_t* out)
{
    g1_affine a; memcpy(&a, p, 64); a = aff_canon(&a);
    fe k; memcpy(k.d, k_mont, 32);
    k = fe_from_mont(&FR, fe_canon(&FR, k));
    g1_jac r; jac_mul(&r, &a, k.d);
    g1_affine o; jac_to_affine(&o, &r);
    memcpy(out, &o, 64);
}
void oracle_g1_add(const uint64_t* p, const uint64_t* q, uint64_t* out)
{
    g1_affine a, b; memcpy(&a, p, 64); memcpy(&b, q, 64); a = aff_canon(&a); b = aff_canon(&b);
    g1_jac j, r;
    if (aff_is_inf(&a)) jac_set_inf(&j); else { j.x = a.x; j.y = a.y; j.z = fe_one(&FQ); }
    jac_madd(&r, &j, &b);
    g1_affine o; jac_to_affine(&o, &r);
    memcpy(out, &o, 64);
}
/* Jacobian (12 limbs, any coarse rep) -> canonical affine */
void oracle_g1_jac_to_affine(const uint64_t* jac, uint64_t* out)
{
    g1_jac j; memcpy(&j, jac, 96);
    g1_affine o;
    if (jac_is_inf(&j)) aff_set_inf(&o);
    else {
        j.x = fe_canon(&FQ, j.x); j.y = fe_canon(&FQ, j.y); j.z = fe_canon(&FQ, j.z);
        jac_to_affine(&o, &j);
    }
    memcpy(out, &o, 64);
}
/* sum of n Jacobian points (g1_sum, B/ecc/curves/bn254/scalar_multiplication/c_bind.cpp:39-46) -> canonical affine */
void oracle_g1_sum(const uint64_t* jacs, size_t n, uint64_t* out)
{
    g1_jac acc; jac_set_inf(&acc);
    for (size_t i = 0; i < n; i++) {
        g1_jac j, t; memcpy(&j, jacs + 12 * i, 96);
        if (!jac_is_inf(&j)) { j.x = fe_canon(&FQ, j.x); j.y = fe_canon(&FQ, j.y); j.z = fe_canon(&FQ, j.z); }
        jac_add(&t, &acc, &j); acc = t;
    }
    g1_affine o; jac_to_affine(&o, &acc);
    memcpy(out, &o, 64);
}
/* affine Montgomery point -> the 64 bytes the prover writes into its transcript
 * (affine_element::serialize_to_buffer, B/ecc/groups/affine_element.hpp:38-45; field write,
 * B/ecc/fields/field.hpp:449-458): y then x, each 32-byte big-endian canonical NON-Montgomery;
 * infinity sets bit 7 of byte 0 (coordinates written as zero here). */
void oracle_g1_to_buffer(const uint64_t* p, uint8_t* buf)
{
    g1_affine a; memcpy(&a, p, 64);
    fe x = fe_zero(), y = fe_zero();
    int inf = aff_is_inf(&a);
    if (!inf) { a = aff_canon(&a); x = fe_from_mont(&FQ, a.x); y = fe_from_mont(&FQ, a.y); }
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) {
            buf[i * 8 + b] = (uint8_t)(y.d[3 - i] >> (56 - 8 * b));
            buf[32 + i * 8 + b] = (uint8_t)(x.d[3 - i] >> (56 - 8 * b));
        }
    if (inf) buf[0] |= 0x80;
}

/* synthetic SRS: P_i = A + i*S (i = 0..n-1), A = a*G, S = s*G with plain 64-bit a, s.
 * Sequential Jacobian madd + one batch inversion (batch_normalize, B/ecc/groups/element_impl.hpp:843-900). */
void oracle_srs_linear(uint64_t a, uint64_t s, size_t n, uint64_t* out_points)
{
    if (n == 0) return;
    g1_affine G; oracle_g1_generator((uint64_t*)&G);
    uint64_t ka[4] = { a, 0, 0, 0 }, ks[4] = { s, 0, 0, 0 };
    g1_jac A, S; jac_mul(&A, &G, ka); jac_mul(&S, &G, ks);
    g1_affine Sa; jac_to_affine(&Sa, &S);
    g1_jac* pts = (g1_jac*)malloc(n * sizeof(g1_jac));
    fe* prod = (fe*)malloc(n * sizeof(fe));
    pts[0] = A;
    for (size_t i = 1; i < n; i++) jac_madd(&pts[i], &pts[i - 1], &Sa);
    fe acc = fe_one(&FQ);
    for (size_t i = 0; i < n; i++) { prod[i] = acc; acc = fe_mul(&FQ, acc, pts[i].z); }
    fe inv = fe_inv(&FQ, acc);
    for (size_t i = n; i-- > 0;) {
        fe zi = fe_mul(&FQ, inv, prod[i]);
        inv = fe_mul(&FQ, inv, pts[i].z);
        fe zi2 = fe_sqr(&FQ, zi), zi3 = fe_mul(&FQ, zi2, zi);
        g1_affine o; o.x = fe_mul(&FQ, pts[i].x, zi2); o.y = fe_mul(&FQ, pts[i].y, zi3);
        memcpy(out_points + 8 * i, &o, 64);
    }
    free(pts); free(prod);
}
/* synthetic SRS without small linear relations: P_i = k_i * G, k_i = mix64(seed + i) | 1 (splitmix64
 * finaliser).  pippenger_unsafe assumes linearly independent bases (scalar_multiplication.cpp:908-921); the
 * A + i*S family violates that (P_0 + P_3 = P_1 + P_2) and trips "attempted to invert zero" (:317-318). */
static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void oracle_srs_hashed(uint64_t seed, size_t n, uint64_t* out_points)
{
    g1_affine G; oracle_g1_generator((uint64_t*)&G);
#pragma omp parallel for schedule(dynamic, 64) num_threads(oracle_team())
    for (size_t i = 0; i < n; i++) {
        uint64_t k[4] = { mix64(seed + (uint64_t)i) | 1ULL, 0, 0, 0 };
        g1_jac acc; jac_set_inf(&acc);
        for (int b = 63; b >= 0; b--) {
            g1_jac t; jac_dbl(&t, &acc); acc = t;
            if ((k[0] >> b) & 1) { jac_madd(&t, &acc, &G); acc = t; }
        }
        g1_affine o; jac_to_affine(&o, &acc);
        memcpy(out_points + 8 * i, &o, 64);
    }
}
/* structured SRS [x^i]G, i = 0..n-1, x Montgomery Fr (what an Ignition transcript holds) */
void oracle_srs_powers(const uint64_t* x_mont, size_t n, uint64_t* out_points)
{
    g1_affine G; oracle_g1_generator((uint64_t*)&G);
    fe x; memcpy(x.d, x_mont, 32); x = fe_canon(&FR, x);
#pragma omp parallel for schedule(dynamic, 16) num_threads(oracle_team())
    for (size_t i = 0; i < n; i++) {
        fe xi = fe_from_mont(&FR, fe_pow64(&FR, x, (uint64_t)i));
        g1_jac r; jac_mul(&r, &G, xi.d);
        g1_affine o; jac_to_affine(&o, &r);
        memcpy(out_points + 8 * i, &o, 64);
    }
}
/* endo table: table[2i] = P_i, table[2i+1] = (beta*x_i, -y_i)
 * (generate_pippenger_point_table, B/ecc/curves/bn254/scalar_multiplication/scalar_multiplication.cpp:104-112) */
void oracle_point_table(const uint64_t* points, size_t n, uint64_t* table)
{
    for (size_t i = n; i-- > 0;) {
        g1_affine p; memcpy(&p, points + 8 * i, 64); p = aff_canon(&p);
        g1_affine e; e.x = fe_mul(&FQ, FQ_CUBE_ROOT, p.x); e.y = fe_neg(&FQ, p.y);
        memcpy(table + 16 * i, &p, 64);
        memcpy(table + 16 * i + 8, &e, 64);
    }
}

/* ------------------------------------------------------------------- MSM */
/* B/ecc/curves/bn254/scalar_multiplication/runtime_states.hpp:9-57 */
size_t oracle_optimal_bucket_width(size_t num_points)
{
    if (num_points >= 14617149) return 21;
    if (num_points >= 1139094) return 18;
    if (num_points >= 155975) return 15;
    if (num_points >= 144834) return 14;
    if (num_points >= 25067) return 12;
    if (num_points >= 13926) return 11;
    if (num_points >= 7659) return 10;
    if (num_points >= 2436) return 9;
    if (num_points >= 376) return 7;
    if (num_points >= 231) return 6;
    if (num_points >= 97) return 5;
    if (num_points >= 35) return 4;
    if (num_points >= 10) return 3;
    if (num_points >= 2) return 2;
    return 1;
}
#define SCALAR_BITS 127 /* B/ecc/groups/wnaf.hpp:8 */

static int msb64(uint64_t x) { return 63 - __builtin_clzll(x); }
/* B/ecc/groups/wnaf.hpp:110-136 get_wnaf_bits */
static uint64_t wnaf_bits_at(const uint64_t* scalar, uint64_t bits, uint64_t pos)
{
    size_t lo_idx = (size_t)(pos >> 6), hi_idx = (size_t)((pos + bits - 1) >> 6);
    uint64_t lo_shift = pos & 63, mask = (1ULL << bits) - 1;
    uint64_t lo = scalar[lo_idx] >> lo_shift;
    uint64_t hi = 0;
    if (lo_idx != hi_idx && hi_idx < 2) hi = scalar[hi_idx] << (64 - lo_shift);
    return (lo | hi) & mask;
}
/*
 * Signed fixed-window recoding of one 128-bit half-scalar into per-round schedule words
 * (fixed_wnaf_with_counts, B/ecc/groups/wnaf.hpp:230-283; word format scalar_multiplication.hpp:24-29):
 * word = point_index<<32 | negative<<31 | (|digit|-1)/2 ; 0xffff..ff = no entry.  Round 0 is the
 * MOST significant window.  `skew` = 1 when the half-scalar was even (one extra -P at the end).
 */
static void fixed_wnaf_with_counts(const uint64_t* scalar, uint64_t* wnaf, uint8_t* skew, uint64_t* round_counts,
                                   uint64_t point_index, size_t stride, size_t wnaf_bits)
{
    size_t max_entries = (SCALAR_BITS + wnaf_bits - 1) / wnaf_bits;
    if ((scalar[0] | scalar[1]) == 0) {
        *skew = 0;
        for (size_t r = 0; r < max_entries; r++) wnaf[r * stride] = ~0ULL;
        return;
    }
    size_t nbits = (size_t)(scalar[1] ? msb64(scalar[1]) + 64 : msb64(scalar[0])) + 1;
    *skew = (scalar[0] & 1) == 0;
    uint64_t previous = wnaf_bits_at(scalar, wnaf_bits, 0) + (uint64_t)*skew;
    size_t entries = (nbits + wnaf_bits - 1) / wnaf_bits;
    if (entries == 1) {
        wnaf[(max_entries - 1) * stride] = (previous >> 1) | point_index;
        round_counts[max_entries - 1]++;
        for (size_t j = entries; j < max_entries; j++) wnaf[(max_entries - 1 - j) * stride] = ~0ULL;
        return;
    }
    for (size_t r = 1; r < entries - 1; r++) {
        uint64_t slice = wnaf_bits_at(scalar, wnaf_bits, r * wnaf_bits);
        uint64_t pred = (slice & 1) == 0;
        round_counts[max_entries - r]++;
        wnaf[(max_entries - r) * stride] =
            ((((previous - (pred << wnaf_bits)) ^ (0ULL - pred)) >> 1) | (pred << 31)) | point_index;
        previous = slice + pred;
    }
    size_t final_bits = nbits - wnaf_bits * (entries - 1);
    uint64_t slice = wnaf_bits_at(scalar, final_bits, (entries - 1) * wnaf_bits);
    uint64_t pred = (slice & 1) == 0;
    round_counts[max_entries - entries + 1]++;
    wnaf[(max_entries - entries + 1) * stride] =
        ((((previous - (pred << wnaf_bits)) ^ (0ULL - pred)) >> 1) | (pred << 31)) | point_index;
    round_counts[max_entries - entries]++;
    wnaf[(max_entries - entries) * stride] = ((slice + pred) >> 1) | point_index;
    for (size_t j = entries; j < max_entries; j++) wnaf[(max_entries - 1 - j) * stride] = ~0ULL;
}

/* compute_wnaf_states restated (scalar_multiplication.cpp:188-252): schedule[round*2n + 2j(+1)] */
void oracle_wnaf_schedule(const uint64_t* scalars_mont, size_t n, size_t wnaf_bits, uint64_t* schedule, uint8_t* skew,
                          uint64_t* round_counts)
{
    size_t rounds = (SCALAR_BITS + wnaf_bits - 1) / wnaf_bits;
    for (size_t r = 0; r < rounds; r++) round_counts[r] = 0;
    for (size_t j = 0; j < n; j++) {
        fe s; memcpy(s.d, scalars_mont + 4 * j, 32);
        fe k = fe_from_mont(&FR, fe_canon(&FR, s));
        uint64_t k1[2], k2[2];
        fr_split_endo(k, k1, k2);
        fixed_wnaf_with_counts(k1, schedule + 2 * j, skew + 2 * j, round_counts, (uint64_t)(2 * j) << 32, 2 * n, wnaf_bits);
        fixed_wnaf_with_counts(k2, schedule + 2 * j + 1, skew + 2 * j + 1, round_counts, (uint64_t)(2 * j + 1) << 32, 2 * n,
                               wnaf_bits);
    }
}

/*
 * Bucket-method MSM over the endo table (pippenger_internal / evaluate_pippenger_rounds,
 * scalar_multiplication.cpp:720-851): per round, add every scheduled (possibly negated) table point
 * into bucket (|digit|-1)/2, combine buckets as sum (2k+1) B_k = 2*sum_{k>=1} running_k + running_0
 * (:773-783), shift the accumulator by wnaf_bits doublings between rounds (:822-827), subtract the
 * skew points after the last round (:809-820).  Differences from the reference, none of which change
 * the group element: no per-round radix sort and no affine-trick batching (buckets are Jacobian
 * accumulators with complete mixed additions, so equal/opposite points are handled like the
 * handle_edge_cases=true path), and any n is processed in one pass instead of the power-of-two head +
 * recursive tail of pippenger() (:891-905).
 */
static void msm_bucket(const uint64_t* scalars_mont, const uint64_t* table, size_t n, g1_jac* result)
{
    size_t c = oracle_optimal_bucket_width(n), w = c + 1;
    size_t rounds = (SCALAR_BITS + w - 1) / w, nb = (size_t)1 << c;
    uint64_t* sched = (uint64_t*)malloc(rounds * 2 * n * sizeof(uint64_t));
    uint8_t* skew = (uint8_t*)malloc(2 * n);
    uint64_t counts[256];
    oracle_wnaf_schedule(scalars_mont, n, w, sched, skew, counts);
    g1_jac* round_sum = (g1_jac*)malloc(rounds * sizeof(g1_jac));
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_team())
    for (size_t r = 0; r < rounds; r++) {
        g1_jac* buckets = (g1_jac*)malloc(nb * sizeof(g1_jac));
        for (size_t k = 0; k < nb; k++) jac_set_inf(&buckets[k]);
        const uint64_t* s = sched + r * 2 * n;
        for (size_t e = 0; e < 2 * n; e++) {
            uint64_t wd = s[e];
            if (wd == ~0ULL) continue;
            size_t idx = (size_t)(wd >> 32), b = (size_t)(wd & 0x7fffffffULL);
            g1_affine p; memcpy(&p, table + 8 * idx, 64);
            if (wd & 0x80000000ULL) p = aff_neg(&p);
            g1_jac t; jac_madd(&t, &buckets[b], &p); buckets[b] = t;
        }
        g1_jac running, acc, t;
        jac_set_inf(&running); jac_set_inf(&acc);
        for (size_t k = nb - 1; k > 0; k--) {
            jac_add(&t, &running, &buckets[k]); running = t;
            jac_add(&t, &acc, &running); acc = t;
        }
        jac_add(&t, &running, &buckets[0]); running = t;
        jac_dbl(&t, &acc); acc = t;
        jac_add(&t, &acc, &running); acc = t;
        round_sum[r] = acc;
        free(buckets);
    }
    g1_jac total, t; jac_set_inf(&total);
    for (size_t r = 0; r < rounds; r++) {
        if (r > 0) for (size_t k = 0; k < w; k++) { jac_dbl(&t, &total); total = t; }
        jac_add(&t, &total, &round_sum[r]); total = t;
    }
    for (size_t e = 0; e < 2 * n; e++) {
        if (!skew[e]) continue;
        g1_affine p; memcpy(&p, table + 8 * e, 64); p = aff_neg(&p);
        jac_madd(&t, &total, &p); total = t;
    }
    *result = total;
    free(sched); free(skew); free(round_sum);
}

/* Pippenger MSM: scalars Montgomery Fr, points = PLAIN affine points (stride 64 B).  out = canonical
 * Montgomery affine (8 limbs).  Mirrors pippenger()/pippenger_unsafe() results
 * (scalar_multiplication.cpp:853-929) after g1::affine_element(result). */
void oracle_pippenger(const uint64_t* scalars_mont, const uint64_t* points, size_t n, uint64_t* out)
{
    g1_affine o;
    if (n == 0) { aff_set_inf(&o); memcpy(out, &o, 64); return; }
    uint64_t* table = (uint64_t*)malloc(n * 128);
    oracle_point_table(points, n, table);
    g1_jac r; msm_bucket(scalars_mont, table, n, &r);
    jac_to_affine(&o, &r);
    memcpy(out, &o, 64);
    free(table);
}
/* naive sum_i s_i * P_i -- the oracle every reference MSM test uses
 * (scalar_multiplication.test.cpp:655-686: element::operator* then +=, .normalize()) */
void oracle_msm_naive(const uint64_t* scalars_mont, const uint64_t* points, size_t n, uint64_t* out)
{
    g1_jac acc; jac_set_inf(&acc);
#pragma omp parallel num_threads(oracle_team())
    {
        g1_jac local; jac_set_inf(&local);
#pragma omp for schedule(dynamic, 8) nowait
        for (size_t i = 0; i < n; i++) {
            g1_affine p; memcpy(&p, points + 8 * i, 64); p = aff_canon(&p);
            fe k; memcpy(k.d, scalars_mont + 4 * i, 32);
            k = fe_from_mont(&FR, fe_canon(&FR, k));
            g1_jac r, t; jac_mul(&r, &p, k.d);
            jac_add(&t, &local, &r); local = t;
        }
#pragma omp critical
        { g1_jac t; jac_add(&t, &acc, &local); acc = t; }
    }
    g1_affine o; jac_to_affine(&o, &acc);
    memcpy(out, &o, 64);
}

/* ------------------------------------------------------------------- NTT */
/* B/polynomials/polynomial_arithmetic.cpp:39-46 */
static uint32_t reverse_bits(uint32_t x, uint32_t bit_length)
{
    x = (((x & 0xaaaaaaaa) >> 1) | ((x & 0x55555555) << 1));
    x = (((x & 0xcccccccc) >> 2) | ((x & 0x33333333) << 2));
    x = (((x & 0xf0f0f0f0) >> 4) | ((x & 0x0f0f0f0f) << 4));
    x = (((x & 0xff00ff00) >> 8) | ((x & 0x00ff00ff) << 8));
    return (((x >> 16) | (x << 16))) >> (32 - bit_length);
}
/*
 * In-place radix-2 DIT, natural order in and out: A_i = sum_j a_j root^(ij).
 * Restates fft_inner_serial / fft_inner_parallel (B/polynomials/polynomial_arithmetic.cpp:59-95,140-255):
 * bit-reversal, a twiddle-free first stage, then stages m = 2,4,..,n/2 with twiddle
 * round_roots[log2 m - 1][j] = root^(j * n/(2m)) (B/polynomials/evaluation_domain.cpp:33-55).
 */
static void fft_inner(fe* a, size_t n, fe root)
{
    if (n <= 1) return;
    unsigned lg = (unsigned)msb64(n);
    for (size_t i = 0; i < n; i++) {
        uint32_t j = reverse_bits((uint32_t)i, lg);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (size_t k = 0; k < n; k += 2) {
        fe t = a[k + 1];
        a[k + 1] = fe_sub(&FR, a[k], t);
        a[k] = fe_add(&FR, a[k], t);
    }
    for (size_t m = 2; m < n; m *= 2) {
        fe round_root = fe_pow64(&FR, root, (uint64_t)(n / (2 * m)));
        fe* tw = (fe*)malloc(m * sizeof(fe));
        tw[0] = fe_one(&FR);
        for (size_t j = 1; j < m; j++) tw[j] = fe_mul(&FR, tw[j - 1], round_root);
#pragma omp parallel for schedule(static) if (n >= 4096) num_threads(oracle_team())
        for (size_t kk = 0; kk < n / (2 * m); kk++) {
            size_t k = kk * 2 * m;
            for (size_t j = 0; j < m; j++) {
                fe t = fe_mul(&FR, tw[j], a[k + j + m]);
                a[k + j + m] = fe_sub(&FR, a[k + j], t);
                a[k + j] = fe_add(&FR, a[k + j], t);
            }
        }
        free(tw);
    }
}
/* target[i] = coeffs[i] * start * shift^i for i < size (scale_by_generator, polynomial_arithmetic.cpp:97-117) */
static void scale_by_generator(fe* a, size_t size, fe start, fe shift)
{
    fe g = start;
    for (size_t i = 0; i < size; i++) { a[i] = fe_mul(&FR, a[i], g); g = fe_mul(&FR, g, shift); }
}
static void scale_all(fe* a, size_t n, fe v)
{
#pragma omp parallel for schedule(static) if (n >= 4096) num_threads(oracle_team())
    for (size_t i = 0; i < n; i++) a[i] = fe_mul(&FR, a[i], v);
}

/*
 * One entry point for the whole family (B/polynomials/polynomial_arithmetic.cpp:374-484).
 *   op 0 fft                              (:374)
 *   op 1 ifft                             (:379)
 *   op 2 coset_fft(coeffs, domain)        (:395)   scales the first generator_size coeffs by g^i
 *   op 3 coset_ifft                       (:480)   ifft then g^-i over the whole domain
 *   op 4 fft_with_constant(value)         (:387)
 *   op 5 coset_fft_with_constant(value)   (:458)   start = value
 *   op 6 coset_fft_with_generator_shift(c)(:465)   generator = g*c
 *   op 7 ifft_with_constant(value)        (:471)
 * coeffs: n x 4 limbs Montgomery (any rep) in place -> canonical Montgomery.  constant: Montgomery or NULL.
 */
int oracle_ntt(uint64_t* coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant)
{
    size_t n = (size_t)1 << log2n;
    if (log2n > 28) return -1;
    fe* a = (fe*)coeffs;
    for (size_t i = 0; i < n; i++) a[i] = fe_canon(&FR, a[i]);
    fe root = fr_root_of_unity(log2n), root_inv = fe_inv(&FR, root);
    fe nfe = { { (uint64_t)n, 0, 0, 0 } };
    fe n_inv = fe_inv(&FR, fe_to_mont(&FR, nfe));
    fe g = fr_coset_generator(), g_inv = fe_inv(&FR, g);
    fe c = fe_one(&FR);
    if (constant) { memcpy(c.d, constant, 32); c = fe_canon(&FR, c); }
    if (generator_size == 0 || generator_size > n) generator_size = n;
    switch (op) {
    case 0: fft_inner(a, n, root); break;
    case 1: fft_inner(a, n, root_inv); scale_all(a, n, n_inv); break;
    case 2: scale_by_generator(a, generator_size, fe_one(&FR), g); fft_inner(a, n, root); break;
    case 3: fft_inner(a, n, root_inv); scale_all(a, n, n_inv); scale_by_generator(a, n, fe_one(&FR), g_inv); break;
    case 4: fft_inner(a, n, root); scale_all(a, n, c); break;
    case 5: scale_by_generator(a, generator_size, c, g); fft_inner(a, n, root); break;
    case 6: scale_by_generator(a, generator_size, fe_one(&FR), fe_mul(&FR, g, c)); fft_inner(a, n, root); break;
    case 7: fft_inner(a, n, root_inv); scale_all(a, n, fe_mul(&FR, n_inv, c)); break;
    default: return -2;
    }
    return 0;
}
/*
 * coset_fft(coeffs, small_domain, large_domain, ext) (polynomial_arithmetic.cpp:401-456): coeffs holds n
 * coefficients in a buffer of ext*n; the result is the coset FFT over the ext*n domain computed as ext
 * size-n coset FFTs with generators g * omega_{ext*n}^k, interleaved at index ext*i + k.
 */
int oracle_coset_fft_split(uint64_t* coeffs, unsigned log2n, size_t ext)
{
    size_t n = (size_t)1 << log2n;
    unsigned lext = (unsigned)msb64(ext);
    if (((size_t)1 << lext) != ext || log2n + lext > 28) return -1;
    fe* a = (fe*)coeffs;
    for (size_t i = 0; i < n; i++) a[i] = fe_canon(&FR, a[i]);
    fe* scratch = (fe*)malloc(ext * n * sizeof(fe));
    fe prim = fr_root_of_unity(log2n + lext), root = fr_root_of_unity(log2n);
    fe gk = fr_coset_generator();
    for (size_t k = 0; k < ext; k++) {
        memcpy(scratch + k * n, a, n * sizeof(fe));
        scale_by_generator(scratch + k * n, n, fe_one(&FR), gk);
        fft_inner(scratch + k * n, n, root);
        gk = fe_mul(&FR, gk, prim);
    }
    for (size_t i = 0; i < n; i++)
        for (size_t k = 0; k < ext; k++) a[ext * i + k] = scratch[k * n + i];
    free(scratch);
    return 0;
}
/* Horner evaluation sum a_j z^j (evaluate, polynomial_arithmetic.cpp:507-538) -- pins FFT ordering
 * exactly as fft_with_small_degree does (polynomial_arithmetic.test.cpp:45-68). */
void oracle_poly_eval(const uint64_t* coeffs, size_t n, const uint64_t* z_mont, uint64_t* out)
{
    fe z; memcpy(z.d, z_mont, 32); z = fe_canon(&FR, z);
    fe acc = fe_zero();
    for (size_t i = n; i-- > 0;) {
        fe c; memcpy(c.d, coeffs + 4 * i, 32);
        acc = fe_add(&FR, fe_mul(&FR, acc, z), fe_canon(&FR, c));
    }
    memcpy(out, acc.d, 32);
}

/* r = a (op) b pointwise, op 0 add / 1 sub / 2 mul (polynomial_arithmetic.cpp:486-505) */
void oracle_poly_binop(int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fe x, y; memcpy(x.d, a + 4 * i, 32); memcpy(y.d, b + 4 * i, 32);
        x = fe_canon(&FR, x); y = fe_canon(&FR, y);
        fe z = op == 0 ? fe_add(&FR, x, y) : op == 1 ? fe_sub(&FR, x, y) : fe_mul(&FR, x, y);
        memcpy(r + 4 * i, z.d, 32);
    }
}
/* compute_kate_opening_coefficients (polynomial_arithmetic.cpp:727-750), the reference's own recurrence:
 * f = F(z); dest[0] = (src[0] - f) * (-1/z); dest[i] = (src[i] - dest[i-1]) * (-1/z).  Returns f in f_out. */
void oracle_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t* z_mont, uint64_t* f_out)
{
    fe z; memcpy(z.d, z_mont, 32); z = fe_canon(&FR, z);
    fe f; oracle_poly_eval(src, n, z_mont, f.d);
    fe divisor = fe_neg(&FR, fe_inv(&FR, z));
    fe prev = f;
    for (size_t i = 0; i < n; i++) {
        fe c; memcpy(c.d, src + 4 * i, 32); c = fe_canon(&FR, c);
        prev = fe_mul(&FR, fe_sub(&FR, c, prev), divisor);
        memcpy(dest + 4 * i, prev.d, 32);
    }
    memcpy(f_out, f.d, 32);
}
/* divide_by_pseudo_vanishing_polynomial (polynomial_arithmetic.cpp:628-725) with compute_multiplicative_subgroup (:119-138):
 * evals[i] *= 1/((g w_ext^(i mod ext))^n - 1) * prod_{k < cut} (g w_T^i - w_src^-(k+1)),  n = 2^log2_src, T = 2^log2_target */
int oracle_divide_by_pseudo_vanishing(uint64_t* evals, unsigned log2_src, unsigned log2_target, size_t cut)
{
    if (log2_target < log2_src || log2_target > 28) return -1;
    unsigned lext = log2_target - log2_src;
    size_t ext = (size_t)1 << lext, T = (size_t)1 << log2_target;
    fe* sub = (fe*)malloc(ext * sizeof(fe));
    fe g = fr_coset_generator();
    fe acc = g;
    for (unsigned i = 0; i < log2_src; i++) acc = fe_sqr(&FR, acc);
    fe subroot = fr_root_of_unity(lext);
    sub[0] = acc;
    for (size_t j = 1; j < ext; j++) sub[j] = fe_mul(&FR, sub[j - 1], subroot);
    for (size_t j = 0; j < ext; j++) sub[j] = fe_inv(&FR, fe_sub(&FR, sub[j], fe_one(&FR)));
    fe root_inv = fe_inv(&FR, fr_root_of_unity(log2_src));
    fe* numer = (fe*)malloc((cut ? cut : 1) * sizeof(fe));
    if (cut) {
        numer[0] = fe_neg(&FR, root_inv);
        for (size_t k = 1; k < cut; k++) numer[k] = fe_mul(&FR, numer[k - 1], root_inv);
    }
    fe wT = fr_root_of_unity(log2_target);
    fe work = g;
    for (size_t i = 0; i < T; i++) {
        fe v; memcpy(v.d, evals + 4 * i, 32); v = fe_canon(&FR, v);
        v = fe_mul(&FR, v, sub[i & (ext - 1)]);
        for (size_t k = 0; k < cut; k++) v = fe_mul(&FR, v, fe_add(&FR, work, numer[k]));
        memcpy(evals + 4 * i, v.d, 32);
        work = fe_mul(&FR, work, wT);
    }
    free(sub); free(numer);
    return 0;
}

void oracle_set_threads(int n) { g_oracle_team = n < 1 ? 1 : n; }

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return oracle_team();
#else
    return 1;
#endif
}

/* Permutation grand product z (ProverPermutationWidget<4,false>::compute_round_commitments, reference
 * plonk/proof_system/widgets/random_widgets/permutation_widget_impl.hpp:48-268, steps 1-3 without the blinding and the ifft):
 *   z[0] = 1,  z[j+1] = z[j] * prod_k (w_k[j] + gamma + beta K_k w^j) / prod_k (w_k[j] + gamma + beta sigma_k[j]),  j < n - 1
 * K_0 = 1, K_1..K_3 = coset generators.  wires / sigmas: 4 arrays of n Lagrange-base values each (concatenated). */
void oracle_permutation_z(const uint64_t* wires, const uint64_t* sigmas, unsigned log2n, const uint64_t* beta_mont,
                          const uint64_t* gamma_mont, const uint64_t* k_mont /* 3 x 4 limbs */, uint64_t* z)
{
    const size_t n = (size_t)1 << log2n;
    fe beta, gamma, K[4];
    memcpy(beta.d, beta_mont, 32); memcpy(gamma.d, gamma_mont, 32);
    K[0] = fe_one(&FR);
    for (int k = 0; k < 3; k++) memcpy(K[k + 1].d, k_mont + 4 * k, 32);
    const fe root = fr_root_of_unity(log2n);
    fe acc = fe_one(&FR), root_beta = beta; /* beta * w^j */
    memcpy(z, acc.d, 32);
    for (size_t j = 0; j + 1 < n; j++) {
        fe num = fe_one(&FR), den = fe_one(&FR);
        for (int k = 0; k < 4; k++) {
            fe w, s; memcpy(w.d, wires + 4 * (k * n + j), 32); memcpy(s.d, sigmas + 4 * (k * n + j), 32);
            fe wpg = fe_add(&FR, fe_canon(&FR, w), gamma);
            num = fe_mul(&FR, num, fe_add(&FR, wpg, fe_mul(&FR, K[k], root_beta)));
            den = fe_mul(&FR, den, fe_add(&FR, wpg, fe_mul(&FR, fe_canon(&FR, s), beta)));
        }
        acc = fe_mul(&FR, acc, fe_mul(&FR, num, fe_inv(&FR, den)));
        fe c = fe_canon(&FR, acc);
        memcpy(z + 4 * (j + 1), c.d, 32);
        root_beta = fe_mul(&FR, root_beta, root);
    }
}

/* ------------------------------------------------------------------- quotient widgets of a TurboPLONK prover
 * CPU restatement of what ProverBase::execute_fourth_round's widgets add to the quotient on the 4n coset domain (reference
 * plonk/proof_system/widgets: random_widgets/permutation_widget_impl.hpp:316-420 and transition_widgets/transition_widget.hpp:262-290
 * with turbo_arithmetic_widget.hpp, turbo_fixed_base_widget.hpp, turbo_range_widget.hpp, turbo_logic_widget.hpp), written from
 * the identities they enforce.  polys[] in the order of include/bbg.h's bbg_quotient_poly (w_1..4, z, sigma_1..4, q_1..5, q_m, q_c,
 * q_arith, q_ecc_1, q_range, q_logic, L_1), each 2^log2_large values; challenges = alpha_base, alpha, beta, gamma,
 * public_input_delta, g, k1, k2, k3 (Montgomery).  widget 0 assigns the quotient, 1..4 accumulate; alpha_out = the
 * alpha_base handed to the next widget.  Pinned by tests/golden/widgets.json (recorded from the reference's widget objects). */
static fe fr_small(uint64_t k) { fe r = { { k, 0, 0, 0 } }; return fe_to_mont(&FR, r); }
static fe fr_ld(const uint64_t* p, size_t i) { fe r; memcpy(r.d, p + 4 * i, 32); return fe_canon(&FR, r); }
#define FM(a, b) fe_mul(&FR, (a), (b))
#define FA(a, b) fe_add(&FR, (a), (b))
#define FS(a, b) fe_sub(&FR, (a), (b))
static fe fr_x4(fe v) { fe d = FA(v, v); return FA(d, d); }
static fe fr_quad(fe d) /* D (D-1)(D-2)(D-3) */
{
    fe t = FS(FM(d, d), d);
    t = FM(t, FS(d, fr_small(2)));
    return FM(t, FS(d, fr_small(3)));
}
int oracle_quotient_widget(int widget, const uint64_t* const* polys, unsigned log2_large, const uint64_t* challenges, uint64_t* quotient,
                           uint64_t* alpha_out)
{
    if (widget < 0 || widget > 7 || log2_large < 3) return -1;
    const size_t m = (size_t)1 << log2_large, mask = m - 1;
    fe ch[9];
    for (int k = 0; k < 9; k++) { memcpy(ch[k].d, challenges + 4 * k, 32); ch[k] = fe_canon(&FR, ch[k]); }
    const fe alpha_base = ch[0], alpha = ch[1], beta = ch[2], gamma = ch[3], delta_pi = ch[4], g = ch[5];
    const fe K[4] = { fe_one(&FR), ch[6], ch[7], ch[8] };
    fe ap[8];
    ap[0] = alpha_base;
    for (int k = 1; k < 8; k++) ap[k] = FM(ap[k - 1], alpha);
    const fe one = fe_one(&FR);
    const uint64_t *W1 = polys[0], *W2 = polys[1], *W3 = polys[2], *W4 = polys[3], *Z = polys[4];
    const uint64_t *Q1 = polys[9], *Q2 = polys[10], *Q3 = polys[11], *Q4 = polys[12], *Q5 = polys[13], *QM = polys[14], *QC = polys[15];
    fe next;
    if (widget == 0 || widget == 5) { /* 5 = StandardPLONK: three wire columns */
        const int width = widget == 0 ? 4 : 3;
        const fe root = fr_root_of_unity(log2_large);
        fe rb = FM(beta, g); /* beta * g * w^i */
        const fe ab2 = FM(alpha_base, alpha_base);
        for (size_t i = 0; i < m; i++) {
            fe num = one, den = one;
            for (int k = 0; k < width; k++) {
                fe wpg = FA(fr_ld(polys[k], i), gamma);
                num = FM(num, FA(wpg, FM(K[k], rb)));
                den = FM(den, FA(wpg, FM(fr_ld(polys[5 + k], i), beta)));
            }
            const fe z = fr_ld(Z, i), zw = fr_ld(Z, (i + 4) & mask);
            num = FM(num, z);
            den = FM(den, zw);
            num = FA(num, FM(FM(FS(zw, delta_pi), alpha_base), fr_ld(polys[20], (i + 4 + 16) & mask)));
            num = FA(num, FM(FM(FS(z, one), ab2), fr_ld(polys[20], i)));
            fe q = FM(FS(num, den), alpha_base);
            memcpy(quotient + 4 * i, q.d, 32);
            rb = FM(rb, root);
        }
        next = FM(ab2, ab2);
    } else if (widget == 6) { /* StandardPLONK arithmetic gate (arithmetic_widget.hpp) */
        for (size_t i = 0; i < m; i++) {
            const fe w1 = fr_ld(W1, i), w2 = fr_ld(W2, i);
            fe gate = FM(FM(w1, w2), fr_ld(QM, i));
            gate = FA(gate, FM(w1, fr_ld(Q1, i)));
            gate = FA(gate, FM(w2, fr_ld(Q2, i)));
            gate = FA(gate, FM(fr_ld(W3, i), fr_ld(Q3, i)));
            gate = FA(gate, fr_ld(QC, i));
            fe q = FA(fr_ld(quotient, i), FM(gate, ap[0]));
            memcpy(quotient + 4 * i, q.d, 32);
        }
        next = ap[1];
    } else if (widget == 7) { /* MiMC round gate (mimc_widget.hpp:17-52): polys[21] = q_mimc_coefficient, polys[22] = q_mimc_selector */
        for (size_t i = 0; i < m; i++) {
            const fe w2 = fr_ld(W2, i);
            const fe t0 = FA(FA(fr_ld(W1, i), fr_ld(W3, i)), fr_ld(polys[21], i));
            const fe t1 = FS(FM(FM(t0, t0), t0), w2);                         /* (w1 + w3 + c)^3 - w2 */
            const fe t2 = FS(FM(FM(w2, w2), t0), fr_ld(W3, (i + 4) & mask));  /* w2^2 (w1 + w3 + c) - w3(wX) */
            const fe t3 = FA(FM(t1, ap[0]), FM(t2, ap[1]));
            fe q = FA(fr_ld(quotient, i), FM(t3, fr_ld(polys[22], i)));
            memcpy(quotient + 4 * i, q.d, 32);
        }
        next = ap[2]; /* two relations */
    } else {
        for (size_t i = 0; i < m; i++) {
            const size_t ish = (i + 4) & mask;
            const fe w1 = fr_ld(W1, i), w2 = fr_ld(W2, i), w3 = fr_ld(W3, i), w4 = fr_ld(W4, i);
            const fe w1n = fr_ld(W1, ish), w2n = fr_ld(W2, ish), w3n = fr_ld(W3, ish), w4n = fr_ld(W4, ish);
            fe add;
            if (widget == 1) { /* arithmetic gate + high-bit extraction */
                const fe qa = fr_ld(polys[16], i);
                fe gate = FM(FM(w1, w2), fr_ld(QM, i));
                gate = FA(gate, FM(w1, fr_ld(Q1, i)));
                gate = FA(gate, FM(w2, fr_ld(Q2, i)));
                gate = FA(gate, FM(w3, fr_ld(Q3, i)));
                gate = FA(gate, FM(w4, fr_ld(Q4, i)));
                gate = FA(gate, fr_ld(QC, i));
                fe t = FM(FM(FS(FM(w4, w4), w4), FS(w4, fr_small(2))), alpha);
                gate = FM(FA(gate, FM(t, fr_ld(Q5, i))), qa);
                const fe d = FS(w3, fr_x4(w4));
                fe h = FS(FS(FM(fr_small(9), d), FM(fr_small(2), FM(d, d))), fr_small(7));
                h = FM(FM(h, d), FS(FM(qa, qa), qa));
                add = FM(FA(gate, h), ap[0]);
            } else if (widget == 2) { /* fixed-base ladder over y^2 = x^3 - 17 */
                const fe qc = fr_ld(QC, i), qe = fr_ld(polys[17], i);
                const fe dl = FS(w4n, fr_x4(w4));
                fe lin = FM(FM(FM(dl, dl), ap[1]), fr_ld(Q1, i));
                lin = FA(lin, FM(ap[1], fr_ld(Q2, i)));
                fe t3 = FM(FM(FM(FS(w1n, w1), dl), w3n), ap[3]);
                fe u = FM(FM(FM(dl, w3n), w2), ap[2]);
                lin = FA(lin, FM(FA(t3, FA(u, u)), fr_ld(Q3, i)));
                fe init = FM(FM(w3, ap[5]), fr_ld(Q4, i));
                init = FA(init, FM(FM(FS(one, w4), ap[5]), fr_ld(Q5, i)));
                init = FA(init, FM(FM(w3, ap[6]), fr_ld(QM, i)));
                lin = FA(lin, FM(init, qc));
                const fe three = fr_small(3);
                fe gate = FM(FM(FM(FA(dl, one), FA(dl, three)), FM(FS(dl, one), FS(dl, three))), ap[0]);
                gate = FS(gate, FM(w3n, ap[1]));
                const fe dx = FS(w3n, w1);
                fe xacc = FM(FA(FA(w1n, w1), w3n), FM(dx, dx));
                xacc = FS(xacc, FS(FA(FM(FM(w3n, w3n), w3n), FM(w2, w2)), fr_small(17)));
                fe tdy = FM(FM(dl, w2), qe);
                xacc = FA(xacc, FA(tdy, tdy));
                gate = FA(gate, FM(xacc, ap[2]));
                fe yacc = FA(FM(FA(w2n, w2), dx), FM(FS(w1, w1n), FS(w2, FM(qe, dl))));
                gate = FA(gate, FM(yacc, ap[3]));
                const fe w4m1 = FS(w4, one);
                fe gi = FM(FM(w4m1, FS(w4m1, w3)), ap[4]);
                gi = FS(gi, FM(FM(w1, w3), ap[5]));
                gi = FA(gi, FM(FS(FM(FS(one, w4), qc), FM(w2, w3)), ap[6]));
                gate = FA(gate, FM(gi, qc));
                add = FM(FA(lin, gate), qe);
            } else if (widget == 3) { /* base-4 range raster */
                fe sum = FM(fr_quad(FS(w3, fr_x4(w4))), ap[0]);
                sum = FA(sum, FM(fr_quad(FS(w2, fr_x4(w3))), ap[1]));
                sum = FA(sum, FM(fr_quad(FS(w1, fr_x4(w2))), ap[2]));
                sum = FA(sum, FM(fr_quad(FS(w4n, fr_x4(w1))), ap[3]));
                add = FM(sum, fr_ld(polys[18], i));
            } else { /* AND / XOR on quads */
                const fe qa = FS(w1n, fr_x4(w1)), qb = FS(w2n, fr_x4(w2)), qq = FS(w4n, fr_x4(w4));
                const fe sm = FA(qa, qb), sq = FA(FM(qa, qa), FM(qb, qb));
                fe id = FM(FM(fr_small(2), FS(FM(qa, qb), w3)), alpha);
                id = FM(FA(id, fr_quad(qa)), alpha);
                id = FM(FA(id, fr_quad(qb)), alpha);
                fe e = FA(FS(fr_x4(w3), FM(fr_small(18), sm)), fr_small(81));
                e = FM(e, w3);
                e = FA(e, FA(FS(FM(fr_small(18), sq), FM(fr_small(81), sm)), fr_small(83)));
                e = FM(e, w3);
                fe tail = FS(FM(fr_small(3), FA(qq, sm)), FA(e, e));
                tail = FA(tail, FM(FS(FM(fr_small(9), qq), FM(fr_small(3), sm)), fr_ld(QC, i)));
                add = FM(FM(FA(id, tail), ap[0]), fr_ld(polys[19], i));
            }
            fe q = FA(fr_ld(quotient, i), add);
            memcpy(quotient + 4 * i, q.d, 32);
        }
        static const int NREL[5] = { 0, 2, 7, 4, 4 }; /* independent relations per widget: update_alpha */
        next = FM(ap[NREL[widget] - 1], alpha);
    }
    next = fe_canon(&FR, next);
    memcpy(alpha_out, next.d, 32);
    return 0;
}
#undef FM
#undef FA
#undef FS
