// shim_check.cpp -- link-compatibility + parity proof for the shim (test infrastructure; built only where
// /root/reference exists, the prebuilt binary oracle/_ref/shim_check travels to the GPU box).
//
// This program is written against barretenberg's OWN public API and is linked against barretenberg's OWN translation
// units, unmodified.  The final link wraps the MSM / FFT entry points (-Wl,--wrap=..., shim/wrap_flags.txt), so the
// plain calls below land in shim/bbg_barretenberg_shim.cpp -> libbbg.so -> MI355X, while __real_* reaches the
// reference's CPU implementation in the same process.  Every result is compared the way the reference's tests do
// (operator== on field elements, g1::affine_element equality).
#include <chrono>
#include <omp.h>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <ecc/curves/bn254/scalar_multiplication/pippenger.hpp>
#include <ecc/curves/bn254/scalar_multiplication/scalar_multiplication.hpp>
#include <polynomials/evaluation_domain.hpp>
#include <polynomials/polynomial_arithmetic.hpp>

using namespace barretenberg;

extern "C" size_t bbg_shim_cached_tables(void);
extern "C" uint64_t bbg_shim_stale_tables(void);
extern "C" void bbg_shim_register_point_table(const void* endo_table, size_t num_points);
extern "C" void bbg_shim_unregister_point_table(const void* endo_table);

#define REAL(m) asm("__real_" m)
namespace real {
g1::element pippenger_unsafe(fr*, g1::affine_element*, const size_t, scalar_multiplication::pippenger_runtime_state&)
    REAL("_ZN12barretenberg21scalar_multiplication16pippenger_unsafeEPNS_5fieldINS_13Bn254FrParamsEEEPNS_14group_elements14affine_elementINS1_INS_13Bn254FqParamsEEES3_NS_13Bn254G1ParamsEEEmRNS0_23pippenger_runtime_stateE");
g1::element pippenger(fr*, g1::affine_element*, const size_t, scalar_multiplication::pippenger_runtime_state&, bool)
    REAL("_ZN12barretenberg21scalar_multiplication9pippengerEPNS_5fieldINS_13Bn254FrParamsEEEPNS_14group_elements14affine_elementINS1_INS_13Bn254FqParamsEEES3_NS_13Bn254G1ParamsEEEmRNS0_23pippenger_runtime_stateEb");
void fft(fr*, const evaluation_domain&) REAL("_ZN12barretenberg21polynomial_arithmetic3fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void ifft(fr*, const evaluation_domain&) REAL("_ZN12barretenberg21polynomial_arithmetic4ifftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void coset_fft(fr*, const evaluation_domain&) REAL("_ZN12barretenberg21polynomial_arithmetic9coset_fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void coset_fft4(fr*, const evaluation_domain&, const evaluation_domain&, const size_t)
    REAL("_ZN12barretenberg21polynomial_arithmetic9coset_fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainES7_m");
void coset_ifft(fr*, const evaluation_domain&) REAL("_ZN12barretenberg21polynomial_arithmetic10coset_ifftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void fft_with_constant(fr*, const evaluation_domain&, const fr&)
    REAL("_ZN12barretenberg21polynomial_arithmetic17fft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void coset_fft_with_constant(fr*, const evaluation_domain&, const fr&)
    REAL("_ZN12barretenberg21polynomial_arithmetic23coset_fft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void coset_fft_with_generator_shift(fr*, const evaluation_domain&, const fr&)
    REAL("_ZN12barretenberg21polynomial_arithmetic30coset_fft_with_generator_shiftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void ifft_with_constant(fr*, const evaluation_domain&, const fr&)
    REAL("_ZN12barretenberg21polynomial_arithmetic18ifft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
fr evaluate(const fr*, const fr&, const size_t) REAL("_ZN12barretenberg21polynomial_arithmetic8evaluateEPKNS_5fieldINS_13Bn254FrParamsEEERS4_m");
fr compute_kate_opening_coefficients(const fr*, fr*, const fr&, const size_t)
    REAL("_ZN12barretenberg21polynomial_arithmetic33compute_kate_opening_coefficientsEPKNS_5fieldINS_13Bn254FrParamsEEEPS3_RS4_m");
void divide_by_pseudo_vanishing_polynomial(fr*, const evaluation_domain&, const evaluation_domain&, const size_t)
    REAL("_ZN12barretenberg21polynomial_arithmetic37divide_by_pseudo_vanishing_polynomialEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainES7_m");
} // namespace real

static uint64_t sm_state = 0xBB254;
static uint64_t splitmix()
{
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static fr rand_fr()
{
    fr r{ splitmix(), splitmix(), splitmix(), splitmix() & 0x0FFFFFFFFFFFFFFFULL };
    return r;
}
static int failures = 0;
static void expect(bool ok, const char* what)
{
    std::printf("%s %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) failures++;
}
static bool same(const std::vector<fr>& a, const std::vector<fr>& b)
{
    for (size_t i = 0; i < a.size(); i++)
        if (!(a[i] == b[i])) return false;
    return true;
}

static void on_segv(int sig)
{
    void* frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "shim_check: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    _exit(128 + sig);
}

int main(int argc, char** argv)
{
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    // The reference's CPU pippenger mis-sizes its per-thread slices when the host has many more cores than the
    // 8 it was tuned on (observed: SIGSEGV in compute_wnaf_states at n = 2^14 with 128 threads); pin the CPU side.
    omp_set_num_threads(8);
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t log2n = argc > 1 ? (size_t)atoi(argv[1]) : 14;
    std::printf("shim_check: n = 2^%zu\n", log2n);
    const size_t n = (size_t)1 << log2n;
    // ---- SRS: P_i = k_i * G (independent bases), then the reference's own endomorphism table
    g1::affine_element* table = scalar_multiplication::point_table_alloc<g1::affine_element>(n);
    {
        std::vector<g1::element> pts(n);
#pragma omp parallel for
        for (size_t i = 0; i < n; i++) {
            fr k{ 0x9E3779B97F4A7C15ULL * (i + 1), i ^ 0xBB254, 0, 0 };
            pts[i] = g1::one * k.to_montgomery_form();
        }
        g1::element::batch_normalize(pts.data(), n);
        for (size_t i = 0; i < n; i++) table[i] = { pts[i].x, pts[i].y };
    }
    scalar_multiplication::generate_pippenger_point_table(table, table, n);
    std::printf("table built\n");
    scalar_multiplication::pippenger_runtime_state rs(n);
    std::vector<fr> scalars(n);
    for (auto& s : scalars) s = rand_fr();

    auto t0 = std::chrono::steady_clock::now();
    g1::element gpu = scalar_multiplication::pippenger_unsafe(scalars.data(), table, n, rs); // -> wrapped -> MI355X
    auto t1 = std::chrono::steady_clock::now();
    g1::element cpu = real::pippenger_unsafe(scalars.data(), table, n, rs);
    auto t2 = std::chrono::steady_clock::now();
    expect(g1::affine_element(gpu) == g1::affine_element(cpu), "pippenger_unsafe(n) GPU == reference CPU");
    std::printf("     first call incl. SRS upload + table precompute %.1f ms, CPU %.1f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    t0 = std::chrono::steady_clock::now();
    gpu = scalar_multiplication::pippenger_unsafe(scalars.data(), table, n, rs);
    t1 = std::chrono::steady_clock::now();
    std::printf("     second call (SRS resident, host scalars) %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
    expect(g1::affine_element(gpu) == g1::affine_element(cpu), "pippenger_unsafe repeat");
    // sub-range addressing: Pippenger::pippenger_unsafe(scalars, from, range) == pippenger_unsafe(scalars, table + 2*from, range)
    const size_t from = n / 4 + 3, range = n / 2 + 1;
    gpu = scalar_multiplication::pippenger_unsafe(scalars.data(), table + 2 * from, range, rs);
    cpu = real::pippenger_unsafe(scalars.data(), table + 2 * from, range, rs);
    expect(g1::affine_element(gpu) == g1::affine_element(cpu), "pippenger_unsafe(table + 2*from, range)");
    gpu = scalar_multiplication::pippenger(scalars.data(), table, 17, rs, true);
    cpu = real::pippenger(scalars.data(), table, 17, rs, true);
    expect(g1::affine_element(gpu) == g1::affine_element(cpu), "pippenger(17 points, handle_edge_cases)");
    gpu = scalar_multiplication::pippenger(scalars.data(), table, 0, rs, true);
    expect(gpu.is_point_at_infinity(), "pippenger(0 points) is infinity");

    // ---- FFT family on the n-domain with generator_size n/4 (as the prover's 4n domain uses, proving_key.cpp:21-22)
    evaluation_domain dom(n, n / 4);
    dom.compute_lookup_table();
    std::vector<fr> base(n);
    for (auto& c : base) c = rand_fr();
    const fr k = rand_fr();
    struct Case { const char* name; void (*gpu)(fr*, const evaluation_domain&); void (*cpu)(fr*, const evaluation_domain&); };
    const Case plain[] = { { "fft", polynomial_arithmetic::fft, real::fft }, { "ifft", polynomial_arithmetic::ifft, real::ifft },
                           { "coset_fft", polynomial_arithmetic::coset_fft, real::coset_fft },
                           { "coset_ifft", polynomial_arithmetic::coset_ifft, real::coset_ifft } };
    for (const Case& c : plain) {
        std::vector<fr> a = base, b = base;
        c.gpu(a.data(), dom);
        c.cpu(b.data(), dom);
        expect(same(a, b), c.name);
    }
    struct CaseK { const char* name; void (*gpu)(fr*, const evaluation_domain&, const fr&); void (*cpu)(fr*, const evaluation_domain&, const fr&); };
    const CaseK withk[] = { { "fft_with_constant", polynomial_arithmetic::fft_with_constant, real::fft_with_constant },
                            { "coset_fft_with_constant", polynomial_arithmetic::coset_fft_with_constant, real::coset_fft_with_constant },
                            { "coset_fft_with_generator_shift", polynomial_arithmetic::coset_fft_with_generator_shift, real::coset_fft_with_generator_shift },
                            { "ifft_with_constant", polynomial_arithmetic::ifft_with_constant, real::ifft_with_constant } };
    for (const CaseK& c : withk) {
        std::vector<fr> a = base, b = base;
        c.gpu(a.data(), dom, k);
        c.cpu(b.data(), dom, k);
        expect(same(a, b), c.name);
    }
    {
        const size_t m = n / 4;
        evaluation_domain small(m, m), large(n, m);
        small.compute_lookup_table();
        large.compute_lookup_table();
        std::vector<fr> a(n, fr::zero()), b(n, fr::zero());
        for (size_t i = 0; i < m; i++) a[i] = b[i] = base[i];
        polynomial_arithmetic::coset_fft(a.data(), small, large, 4);
        real::coset_fft4(b.data(), small, large, 4);
        expect(same(a, b), "coset_fft(coeffs, small, large, 4)");
    }
    { // the O(n) helpers
        const fr z = rand_fr();
        expect(polynomial_arithmetic::evaluate(base.data(), z, n) == real::evaluate(base.data(), z, n), "evaluate");
        expect(polynomial_arithmetic::evaluate(base.data(), z, n - 3) == real::evaluate(base.data(), z, n - 3), "evaluate (ragged n)");
        std::vector<fr> a = base, b = base; // in place, as KateCommitmentScheme::batch_open calls it
        const fr fa = polynomial_arithmetic::compute_kate_opening_coefficients(a.data(), a.data(), z, n);
        const fr fb = real::compute_kate_opening_coefficients(b.data(), b.data(), z, n);
        expect(fa == fb && same(a, b), "compute_kate_opening_coefficients (dest == src)");
        const size_t m = n / 4;
        evaluation_domain small(m, m), large(n, m);
        small.compute_lookup_table();
        large.compute_lookup_table();
        a = base;
        b = base;
        polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial(a.data(), small, large, 4);
        real::divide_by_pseudo_vanishing_polynomial(b.data(), small, large, 4);
        expect(same(a, b), "divide_by_pseudo_vanishing_polynomial(4n coset, 4 roots cut)");
    }
    { // lifetime of the device copies of point tables (shim/bbg_barretenberg_shim.cpp: registered / implicit / transient)
        expect(bbg_shim_cached_tables() == 1, "the table seen inside pippenger is cached once (implicit entry)");
        const size_t small_n = 64; // a verifier-sized table: uploaded, used and freed within the call
        g1::affine_element* small_table = scalar_multiplication::point_table_alloc<g1::affine_element>(small_n);
        for (size_t i = 0; i < small_n; i++) small_table[i] = table[2 * (i + 5)];
        scalar_multiplication::generate_pippenger_point_table(small_table, small_table, small_n);
        scalar_multiplication::pippenger_runtime_state small_rs(small_n);
        g1::element g = scalar_multiplication::pippenger(scalars.data(), small_table, small_n, small_rs, true);
        g1::element c = real::pippenger(scalars.data(), small_table, small_n, small_rs, true);
        expect(g1::affine_element(g) == g1::affine_element(c) && bbg_shim_cached_tables() == 1, "a small table is served from a transient copy (not retained)");
        aligned_free(small_table);
        // owner-announced lifetime: register, use (also through a sub-range), unregister
        const size_t reg_n = n / 2;
        g1::affine_element* reg_table = scalar_multiplication::point_table_alloc<g1::affine_element>(reg_n);
        for (size_t i = 0; i < reg_n; i++) reg_table[i] = table[2 * (n - 1 - i)];
        scalar_multiplication::generate_pippenger_point_table(reg_table, reg_table, reg_n);
        bbg_shim_register_point_table(reg_table, reg_n);
        expect(bbg_shim_cached_tables() == 2, "registered table cached");
        scalar_multiplication::pippenger_runtime_state reg_rs(reg_n);
        g = scalar_multiplication::pippenger_unsafe(scalars.data(), reg_table + 2 * 7, reg_n - 7, reg_rs);
        c = real::pippenger_unsafe(scalars.data(), reg_table + 2 * 7, reg_n - 7, reg_rs);
        expect(g1::affine_element(g) == g1::affine_element(c), "MSM over a sub-range of a registered table");
        bbg_shim_unregister_point_table(reg_table);
        expect(bbg_shim_cached_tables() == 1, "unregister drops the device copy");
        // the same memory, new contents, nobody told the shim: the stale implicit entry must not be used
        for (size_t i = 0; i < reg_n; i++) reg_table[i] = table[2 * i];
        scalar_multiplication::generate_pippenger_point_table(reg_table, reg_table, reg_n);
        g = scalar_multiplication::pippenger_unsafe(scalars.data(), reg_table, reg_n, reg_rs); // implicit entry #2
        for (size_t i = 0; i < reg_n; i++) reg_table[i] = table[2 * (i + 1 < n ? i + 1 : i)];
        scalar_multiplication::generate_pippenger_point_table(reg_table, reg_table, reg_n);
        g = scalar_multiplication::pippenger_unsafe(scalars.data(), reg_table, reg_n, reg_rs);
        c = real::pippenger_unsafe(scalars.data(), reg_table, reg_n, reg_rs);
        expect(g1::affine_element(g) == g1::affine_element(c), "a table rewritten in place is re-uploaded (sampled validation inside the call's own range)");
        // ONE point rewritten, one the samples do not look at (they take every (n / 1024)-th point and the last): the full content check of
        // the call's range (bbg_shim_verify.hpp) must find it -- the commitment is the one over the memory as it is now, and the stale copy
        // is counted.  With BBG_SHIM_VERIFY=sample this is exactly the silent-wrong case rounds 4-5 documented as a tripwire's limit.
        {
            const bool full = !(std::getenv("BBG_SHIM_VERIFY") && std::string(std::getenv("BBG_SHIM_VERIFY")) == "sample");
            const uint64_t stale0 = bbg_shim_stale_tables();
            const size_t poke = reg_n > 2048 ? 1 : reg_n; // an unsampled index where the table has them
            if (poke < reg_n) {
                const g1::affine_element keep = reg_table[2 * poke], keep_endo = reg_table[2 * poke + 1];
                reg_table[2 * poke] = reg_table[2 * (poke + 2)];
                reg_table[2 * poke + 1] = reg_table[2 * (poke + 2) + 1];
                g = scalar_multiplication::pippenger_unsafe(scalars.data(), reg_table, reg_n, reg_rs);
                c = real::pippenger_unsafe(scalars.data(), reg_table, reg_n, reg_rs);
                if (full) {
                    expect(g1::affine_element(g) == g1::affine_element(c), "ONE unsampled point rewritten in a cached table: the MSM is over the table as it is now");
                    expect(bbg_shim_stale_tables() == stale0 + 1, "... and the stale device copy was found by the full check, once");
                    g = scalar_multiplication::pippenger_unsafe(scalars.data(), reg_table + 2 * 3, reg_n - 3, reg_rs); // the fresh copy, a sub-range
                    c = real::pippenger_unsafe(scalars.data(), reg_table + 2 * 3, reg_n - 3, reg_rs);
                    expect(g1::affine_element(g) == g1::affine_element(c) && bbg_shim_stale_tables() == stale0 + 1, "the re-uploaded copy verifies (no second upload)");
                } else {
                    std::printf("note  BBG_SHIM_VERIFY=sample: a single unsampled point goes unnoticed (%s)\n",
                                g1::affine_element(g) == g1::affine_element(c) ? "noticed here" : "stale result, as documented");
                }
                reg_table[2 * poke] = keep;
                reg_table[2 * poke + 1] = keep_endo;
            }
        }
        // many tables without hooks: the implicit cache stays bounded
        std::vector<g1::affine_element*> many;
        for (int t = 0; t < 7; t++) {
            g1::affine_element* tb = scalar_multiplication::point_table_alloc<g1::affine_element>(reg_n);
            for (size_t i = 0; i < reg_n; i++) tb[i] = table[2 * ((i + 3 * (size_t)t + 2) % n)];
            scalar_multiplication::generate_pippenger_point_table(tb, tb, reg_n);
            g = scalar_multiplication::pippenger_unsafe(scalars.data(), tb, reg_n, reg_rs);
            many.push_back(tb);
        }
        c = real::pippenger_unsafe(scalars.data(), many.back(), reg_n, reg_rs);
        expect(g1::affine_element(g) == g1::affine_element(c) && bbg_shim_cached_tables() <= 4, "implicit cache bounded (least recently used entries go)");
        for (auto* tb : many) aligned_free(tb);
        aligned_free(reg_table);
    }
    aligned_free(table);
    std::printf(failures ? "shim_check FAILED (%d)\n" : "shim_check PASS\n", failures);
    return failures ? 1 : 0;
}
