// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (not product code, never shipped in libbbg.so).
//
// A thin extern "C" shell, written for this repo, around the REAL barretenberg hot path.  It is
// compiled together with the reference's own translation units straight from /root/reference
// (see oracle/Makefile; g++ only -- ROCm clang mis-executes the reference's aliasing idiom,
// SURVEY.md fact 2) into oracle/_ref/libbbref.so.  No reference source is copied into the repo.
// Uses: (1) generating the golden fixtures under tests/golden/ (tests/golden/gen_golden.py),
// (2) validating oracle/bn254_oracle.c limb-for-limb in this container, (3) the
// "cpu_baseline.kind = reference" leg of bench.py on the GPU box (the prebuilt .so travels).
//
// All outputs are canonicalised exactly the way the reference's own tests compare values:
// field elements through reduce_once() (field_impl.hpp:100-112), points through
// g1::affine_element(result) (element_impl.hpp:51-68) + reduce_once of both coordinates.
#include <cstdint>
#include <cstring>
#include <chrono>
#include <vector>

#include <ecc/curves/bn254/fq.hpp>
#include <ecc/curves/bn254/fr.hpp>
#include <ecc/curves/bn254/g1.hpp>
#include <ecc/curves/bn254/scalar_multiplication/pippenger.hpp>
#include <ecc/curves/bn254/scalar_multiplication/scalar_multiplication.hpp>
#include <polynomials/evaluation_domain.hpp>
#include <polynomials/polynomial_arithmetic.hpp>

using namespace barretenberg;

namespace {
template <class F> F load(const uint64_t* p)
{
    F r;
    r.data[0] = p[0]; r.data[1] = p[1]; r.data[2] = p[2]; r.data[3] = p[3];
    return r;
}
template <class F> void store(uint64_t* p, const F& a)
{
    F r = a.reduce_once();
    p[0] = r.data[0]; p[1] = r.data[1]; p[2] = r.data[2]; p[3] = r.data[3];
}
g1::affine_element load_aff(const uint64_t* p)
{
    g1::affine_element a;
    a.x = load<fq>(p);
    a.y = load<fq>(p + 4);
    return a;
}
void store_aff(uint64_t* out, const g1::affine_element& a)
{
    if (a.is_point_at_infinity()) {
        memset(out, 0, 64);
        out[3] = 1ULL << 63;
        return;
    }
    store<fq>(out, a.x);
    store<fq>(out + 4, a.y);
}
void store_jac_as_aff(uint64_t* out, const g1::element& e)
{
    if (e.is_point_at_infinity()) {
        memset(out, 0, 64);
        out[3] = 1ULL << 63;
        return;
    }
    store_aff(out, g1::affine_element(e));
}
template <class F> void binop(int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        F x = load<F>(a + 4 * i), y = b ? load<F>(b + 4 * i) : F::zero(), z;
        switch (op) {
        case 0: z = x * y; break;
        case 1: z = x + y; break;
        case 2: z = x - y; break;
        case 3: z = x.invert(); break;
        case 4: z = x.to_montgomery_form(); break;
        case 5: z = x.from_montgomery_form(); break;
        case 6: z = x.sqr(); break;
        default: z = x; break;
        }
        store<F>(r + 4 * i, z);
    }
}
struct MsmCtx {
    g1::affine_element* table;
    size_t n;
    scalar_multiplication::pippenger_runtime_state* state;
};
struct NttCtx {
    evaluation_domain* dom;
};
} // namespace

#include <omp.h>

extern "C" {

int ref_num_threads() { return (int)max_threads::compute_num_threads(); }
// OpenMP team size of everything the reference runs afterwards.  Its compute_wnaf_states misbehaves when the team is large
// relative to the input (SIGSEGV seen with 128 threads at n = 2^14), so callers cap it for small inputs.
void ref_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }

// op: 0 mul 1 add 2 sub 3 invert 4 to_mont 5 from_mont 6 sqr ; which: 0 Fr 1 Fq
void ref_fe_op(int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n)
{
    if (which == 0) binop<fr>(op, a, b, r, n);
    else binop<fq>(op, a, b, r, n);
}

void ref_fr_root_of_unity(unsigned log2n, uint64_t* out) { store<fr>(out, fr::get_root_of_unity(log2n)); }
void ref_fr_coset_generator(uint64_t* out) { store<fr>(out, fr::coset_generator(0)); }

// exactly what compute_wnaf_states feeds the wNAF (scalar_multiplication.cpp:223-225), without the UB alias
void ref_endo_split(const uint64_t* scalars_mont, uint64_t* out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fr k = load<fr>(scalars_mont + 4 * i).from_montgomery_form();
        fr k1{ 0, 0, 0, 0 }, k2{ 0, 0, 0, 0 };
        fr::split_into_endomorphism_scalars(k, k1, k2);
        out[4 * i] = k1.data[0]; out[4 * i + 1] = k1.data[1];
        out[4 * i + 2] = k2.data[0]; out[4 * i + 3] = k2.data[1];
    }
}

void ref_g1_generator(uint64_t* out) { store_aff(out, g1::affine_one); }
int ref_g1_on_curve(const uint64_t* p) { return load_aff(p).on_curve() ? 1 : 0; }
void ref_g1_mul(const uint64_t* p, const uint64_t* k_mont, uint64_t* out)
{
    g1::element e(load_aff(p));
    store_jac_as_aff(out, e * load<fr>(k_mont));
}
void ref_g1_add(const uint64_t* p, const uint64_t* q, uint64_t* out)
{
    g1::element e(load_aff(p));
    e += load_aff(q);
    store_jac_as_aff(out, e);
}
void ref_g1_dbl(const uint64_t* p, uint64_t* out)
{
    g1::element e(load_aff(p));
    e.self_dbl();
    store_jac_as_aff(out, e);
}
void ref_g1_to_buffer(const uint64_t* p, uint8_t* buf)
{
    g1::affine_element a = load_aff(p);
    g1::affine_element::serialize_to_buffer(a, buf);
}
void ref_point_table(const uint64_t* points, size_t n, uint64_t* table)
{
    std::vector<g1::affine_element> pts(n);
    for (size_t i = 0; i < n; i++) pts[i] = load_aff(points + 8 * i);
    g1::affine_element* t = scalar_multiplication::point_table_alloc<g1::affine_element>(n);
    scalar_multiplication::generate_pippenger_point_table(pts.data(), t, n);
    for (size_t i = 0; i < 2 * n; i++) store_aff(table + 8 * i, t[i]);
    aligned_free(t);
}

// compute_wnaf_states (scalar_multiplication.cpp:188-252).  n must be a power of two >= threads.
// schedule: rounds*2n words, skew: 2n bytes, round_counts: 256 words.  returns wnaf_bits.
int ref_wnaf_schedule(const uint64_t* scalars_mont, size_t n, uint64_t* schedule, uint8_t* skew, uint64_t* round_counts)
{
    std::vector<fr> sc(n);
    for (size_t i = 0; i < n; i++) sc[i] = load<fr>(scalars_mont + 4 * i);
    bool* sk = (bool*)aligned_alloc(64, 2 * n + 64);
    uint64_t* sched = (uint64_t*)aligned_alloc(64, (scalar_multiplication::get_num_rounds(2 * n) * 2 * n + 256) * 8);
    scalar_multiplication::compute_wnaf_states(sched, sk, round_counts, sc.data(), n);
    size_t rounds = scalar_multiplication::get_num_rounds(2 * n);
    memcpy(schedule, sched, rounds * 2 * n * 8);
    for (size_t i = 0; i < 2 * n; i++) skew[i] = sk[i] ? 1 : 0;
    aligned_free(sk);
    aligned_free(sched);
    return (int)scalar_multiplication::get_optimal_bucket_width(n) + 1;
}

// ---- MSM with a persistent context (SRS table + runtime state built once, like Pippenger + proving_key)
void* ref_msm_new(const uint64_t* points, size_t n)
{
    MsmCtx* c = new MsmCtx;
    c->n = n;
    c->table = scalar_multiplication::point_table_alloc<g1::affine_element>(n ? n : 1);
    for (size_t i = 0; i < n; i++) c->table[i] = load_aff(points + 8 * i);
    if (n) scalar_multiplication::generate_pippenger_point_table(c->table, c->table, n);
    c->state = new scalar_multiplication::pippenger_runtime_state(n ? n : 1);
    return c;
}
void ref_msm_free(void* h)
{
    MsmCtx* c = (MsmCtx*)h;
    aligned_free(c->table);
    delete c->state;
    delete c;
}
// unsafe != 0 -> pippenger_unsafe (:923) else pippenger(handle_edge_cases=true) (:853).  Returns seconds.
double ref_msm_run(void* h, const uint64_t* scalars_mont, size_t from, size_t n, int unsafe, uint64_t* out_affine)
{
    MsmCtx* c = (MsmCtx*)h;
    fr* sc = (fr*)aligned_alloc(64, (n ? n : 1) * sizeof(fr));
    for (size_t i = 0; i < n; i++) sc[i] = load<fr>(scalars_mont + 4 * i);
    auto t0 = std::chrono::steady_clock::now();
    g1::element r = unsafe ? scalar_multiplication::pippenger_unsafe(sc, c->table + 2 * from, n, *c->state)
                           : scalar_multiplication::pippenger(sc, c->table + 2 * from, n, *c->state, true);
    auto t1 = std::chrono::steady_clock::now();
    store_jac_as_aff(out_affine, r);
    aligned_free(sc);
    return std::chrono::duration<double>(t1 - t0).count();
}
// raw Jacobian result limbs (coarse representation, as returned by value) for shim tests
void ref_msm_run_jac(void* h, const uint64_t* scalars_mont, size_t from, size_t n, uint64_t* out_jac)
{
    MsmCtx* c = (MsmCtx*)h;
    fr* sc = (fr*)aligned_alloc(64, (n ? n : 1) * sizeof(fr));
    for (size_t i = 0; i < n; i++) sc[i] = load<fr>(scalars_mont + 4 * i);
    g1::element r = scalar_multiplication::pippenger_unsafe(sc, c->table + 2 * from, n, *c->state);
    memcpy(out_jac, &r, 96);
    aligned_free(sc);
}
// naive sum s_i * P_i exactly as the reference tests compute their expectation (scalar_multiplication.test.cpp:668-676)
void ref_msm_naive(const uint64_t* scalars_mont, const uint64_t* points, size_t n, uint64_t* out_affine)
{
    g1::element acc;
    acc.self_set_infinity();
    for (size_t i = 0; i < n; i++) {
        g1::element t = g1::element(load_aff(points + 8 * i)) * load<fr>(scalars_mont + 4 * i);
        acc += t;
    }
    store_jac_as_aff(out_affine, acc);
}
void ref_g1_sum(const uint64_t* jacs, size_t n, uint64_t* out_affine)
{
    g1::element acc;
    acc.self_set_infinity();
    for (size_t i = 0; i < n; i++) {
        g1::element e;
        memcpy(&e, jacs + 12 * i, 96);
        acc += e;
    }
    store_jac_as_aff(out_affine, acc);
}

// ---- NTT family with a persistent evaluation_domain (compute_lookup_table outside the timed call)
void* ref_domain_new(unsigned log2n, size_t generator_size)
{
    NttCtx* c = new NttCtx;
    c->dom = new evaluation_domain((size_t)1 << log2n, generator_size);
    c->dom->compute_lookup_table();
    return c;
}
void ref_domain_free(void* h)
{
    NttCtx* c = (NttCtx*)h;
    delete c->dom;
    delete c;
}
// op numbering identical to oracle_ntt().  In place on coeffs (Montgomery limbs); canonicalised on return.
// Returns seconds spent inside the reference call only.
double ref_ntt_run(void* h, uint64_t* coeffs, int op, const uint64_t* constant)
{
    NttCtx* c = (NttCtx*)h;
    const evaluation_domain& d = *c->dom;
    fr* a = (fr*)aligned_alloc(64, d.size * sizeof(fr));
    for (size_t i = 0; i < d.size; i++) a[i] = load<fr>(coeffs + 4 * i);
    fr k = constant ? load<fr>(constant) : fr::one();
    auto t0 = std::chrono::steady_clock::now();
    switch (op) {
    case 0: polynomial_arithmetic::fft(a, d); break;
    case 1: polynomial_arithmetic::ifft(a, d); break;
    case 2: polynomial_arithmetic::coset_fft(a, d); break;
    case 3: polynomial_arithmetic::coset_ifft(a, d); break;
    case 4: polynomial_arithmetic::fft_with_constant(a, d, k); break;
    case 5: polynomial_arithmetic::coset_fft_with_constant(a, d, k); break;
    case 6: polynomial_arithmetic::coset_fft_with_generator_shift(a, d, k); break;
    case 7: polynomial_arithmetic::ifft_with_constant(a, d, k); break;
    default: break;
    }
    auto t1 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < d.size; i++) store<fr>(coeffs + 4 * i, a[i]);
    aligned_free(a);
    return std::chrono::duration<double>(t1 - t0).count();
}
// coset_fft(coeffs, small, large, ext) (polynomial_arithmetic.cpp:401-456); coeffs has ext*n slots
void ref_coset_fft_split(uint64_t* coeffs, unsigned log2n, size_t ext)
{
    size_t n = (size_t)1 << log2n;
    evaluation_domain small(n, n), large(n * ext, n);
    small.compute_lookup_table();
    large.compute_lookup_table();
    fr* a = (fr*)aligned_alloc(64, n * ext * sizeof(fr));
    for (size_t i = 0; i < n; i++) a[i] = load<fr>(coeffs + 4 * i);
    for (size_t i = n; i < n * ext; i++) a[i] = fr::zero();
    polynomial_arithmetic::coset_fft(a, small, large, ext);
    for (size_t i = 0; i < n * ext; i++) store<fr>(coeffs + 4 * i, a[i]);
    aligned_free(a);
}
void ref_poly_eval(const uint64_t* coeffs, size_t n, const uint64_t* z_mont, uint64_t* out)
{
    std::vector<fr> a(n);
    for (size_t i = 0; i < n; i++) a[i] = load<fr>(coeffs + 4 * i);
    store<fr>(out, polynomial_arithmetic::evaluate(a.data(), load<fr>(z_mont), n));
}
// pointwise add / sub / mul over a domain (polynomial_arithmetic.cpp:486-505); op 0 add, 1 sub, 2 mul
void ref_poly_binop(int op, const uint64_t* a, const uint64_t* b, uint64_t* r, unsigned log2n)
{
    const size_t n = (size_t)1 << log2n;
    evaluation_domain d(n);
    std::vector<fr> x(n), y(n), z(n);
    for (size_t i = 0; i < n; i++) { x[i] = load<fr>(a + 4 * i); y[i] = load<fr>(b + 4 * i); }
    if (op == 0) polynomial_arithmetic::add(x.data(), y.data(), z.data(), d);
    else if (op == 1) polynomial_arithmetic::sub(x.data(), y.data(), z.data(), d);
    else polynomial_arithmetic::mul(x.data(), y.data(), z.data(), d);
    for (size_t i = 0; i < n; i++) store<fr>(r + 4 * i, z[i]);
}
void ref_kate_opening(const uint64_t* src, uint64_t* dest, size_t n, const uint64_t* z_mont, uint64_t* f_out)
{
    std::vector<fr> a(n), w(n);
    for (size_t i = 0; i < n; i++) a[i] = load<fr>(src + 4 * i);
    fr f = polynomial_arithmetic::compute_kate_opening_coefficients(a.data(), w.data(), load<fr>(z_mont), n);
    for (size_t i = 0; i < n; i++) store<fr>(dest + 4 * i, w[i]);
    store<fr>(f_out, f);
}
void ref_divide_by_pseudo_vanishing(uint64_t* evals, unsigned log2_src, unsigned log2_target, size_t cut)
{
    const size_t T = (size_t)1 << log2_target;
    evaluation_domain src((size_t)1 << log2_src), tgt(T);
    std::vector<fr> a(T);
    for (size_t i = 0; i < T; i++) a[i] = load<fr>(evals + 4 * i);
    polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial(a.data(), src, tgt, cut);
    for (size_t i = 0; i < T; i++) store<fr>(evals + 4 * i, a[i]);
}
} // extern "C"
