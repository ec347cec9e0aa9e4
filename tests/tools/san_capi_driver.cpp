// san_capi_driver.cpp -- TEST INFRASTRUCTURE: drives the host side of libbbg.so (bbg_capi.hip, msm.hip's dispatch, prover.hip, multi.hip)
// through the C ABI in a build whose HOST code is compiled with AddressSanitizer + UndefinedBehaviorSanitizer (scripts/sanitize_build.sh).
// What it walks: the SRS registry and its reference counts (a prover outliving its SRS handle), the scratch arenas' regrowth (sizes going
// up and down), the MSM batch descriptors, key replacement on a live prover, the memory report / trim, a device group of four contexts on
// one GPU (bbg_multi_*: per-context buffers growing and shrinking), and the error paths.  Values are checked for self-consistency only
// (parity is the GPU suite's job): a batch equals its members issued alone, an NTT round trip is the identity, proofs' commitments repeat.
// The reference's precedent for a sanitizer configuration: barretenberg CMakeLists.txt:5-8 (MEMORY_CHECKS -> -fsanitize=address).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bbg.h"

#define CK(expr)                                                                                         \
    do {                                                                                                 \
        int _rc = (expr);                                                                                \
        if (_rc != BBG_OK) {                                                                             \
            std::fprintf(stderr, "FAIL %s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #expr, _rc, bbg_last_error()); \
            std::exit(1);                                                                                \
        }                                                                                                \
    } while (0)
#define EXPECT(cond)                                                                                     \
    do {                                                                                                 \
        if (!(cond)) {                                                                                   \
            std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #cond);                          \
            std::exit(1);                                                                                \
        }                                                                                                \
    } while (0)

static uint64_t g_seed = 0xBB254;
static uint64_t next64()
{
    g_seed += 0x9E3779B97F4A7C15ULL;
    uint64_t z = g_seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static std::vector<uint64_t> scalars(size_t n)
{
    std::vector<uint64_t> v(4 * n);
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 4; k++) v[4 * i + k] = next64();
        v[4 * i + 3] &= 0x0fffffffffffffffULL;
    }
    return v;
}
static std::vector<uint64_t> affine(bbg_ctx* ctx, const uint64_t* jac, size_t count)
{
    std::vector<uint64_t> out(8 * count);
    CK(bbg_g1_normalize(ctx, jac, count, out.data()));
    return out;
}

int main()
{
    bbg_ctx* ctx = nullptr;
    CK(bbg_init(0, &ctx));
    // ---- SRS + MSM: sizes up and down (arena regrowth), batches, ranges
    bbg_srs* srs = nullptr;
    const size_t N = 1 << 13;
    CK(bbg_srs_synth_hashed(ctx, 0xBB254, N, &srs));
    EXPECT(bbg_srs_num_points(srs) == N);
    for (size_t n : { (size_t)0, (size_t)1, (size_t)100, (size_t)4096, N, (size_t)17, (size_t)2048 }) {
        std::vector<uint64_t> sc = scalars(n ? n : 1);
        uint64_t out[12];
        CK(bbg_msm(ctx, srs, sc.data(), 0, n, out));
    }
    {
        const size_t lens[5] = { 1000, 0, 1, 4097, 333 }, from[5] = { 0, 5, 8191, 100, 7000 };
        std::vector<std::vector<uint64_t>> sc;
        const uint64_t* ptrs[5];
        for (int k = 0; k < 5; k++) {
            sc.push_back(scalars(lens[k] ? lens[k] : 1));
            ptrs[k] = sc.back().data();
        }
        uint64_t batch[5 * 12], single[5 * 12];
        CK(bbg_msm_batch(ctx, srs, 5, ptrs, from, lens, batch));
        for (int k = 0; k < 5; k++) CK(bbg_msm(ctx, srs, ptrs[k], from[k], lens[k], single + 12 * k));
        EXPECT(affine(ctx, batch, 5) == affine(ctx, single, 5));
        EXPECT(bbg_msm_batch(ctx, srs, 9, ptrs, from, lens, batch) == BBG_E_INVALID);
        const size_t bad[5] = { 0, 0, 8192, 0, 0 };
        EXPECT(bbg_msm_batch(ctx, srs, 5, ptrs, bad, lens, batch) == BBG_E_INVALID);
        int c = 0, w = 0;
        CK(bbg_msm_plan(ctx, srs, N, &c, &w));
        EXPECT(c == 13 && w == 20); // the small-circuit configuration
    }
    // ---- NTT family, sizes up and down (domain cache, scratch regrowth)
    for (unsigned lg : { 10u, 13u, 4u, 12u, 1u, 11u }) {
        const size_t n = (size_t)1 << lg;
        std::vector<uint64_t> a = scalars(n), b = a;
        CK(bbg_ntt(ctx, b.data(), lg, BBG_COSET_FFT, 0, nullptr));
        CK(bbg_ntt(ctx, b.data(), lg, BBG_COSET_IFFT, 0, nullptr));
        std::vector<uint64_t> c = a;
        CK(bbg_ntt(ctx, c.data(), lg, BBG_FFT, 0, nullptr));
        CK(bbg_ntt(ctx, c.data(), lg, BBG_IFFT, 0, nullptr));
        // canonical comparison through the field self-test entry (from_montgomery of both)
        std::vector<uint64_t> ca(4 * n), cb(4 * n), cc(4 * n);
        CK(bbg_field_op(ctx, 0, 4, a.data(), a.data(), ca.data(), n));
        CK(bbg_field_op(ctx, 0, 4, b.data(), b.data(), cb.data(), n));
        CK(bbg_field_op(ctx, 0, 4, c.data(), c.data(), cc.data(), n));
        EXPECT(ca == cb && ca == cc);
        if (lg >= 4) {
            std::vector<uint64_t> ext(4 * (4 * n + 4));
            CK(bbg_coset_fft_extend(ctx, a.data(), lg, lg + 2, ext.data()));
            std::vector<uint64_t> split(a);
            split.resize(4 * 4 * n, 0);
            CK(bbg_coset_fft_split(ctx, split.data(), lg, 4));
        }
    }
    EXPECT(bbg_ntt(ctx, nullptr, 10, BBG_FFT, 0, nullptr) == BBG_E_INVALID);
    // ---- resident prover: key registration, two proofs, key replacement on the live handle, SRS handle released first
    {
        const unsigned lg = 10;
        const size_t n = (size_t)1 << lg;
        uint64_t gens[16];
        {
            uint64_t raw[16] = { 5, 0, 0, 0, 5, 0, 0, 0, 6, 0, 0, 0, 7, 0, 0, 0 };
            CK(bbg_field_op(ctx, 0, 5, raw, raw, gens, 4));
        }
        bbg_srs* psrs = nullptr;
        CK(bbg_srs_synth_hashed(ctx, 77, n, &psrs));
        bbg_prover* p = nullptr;
        CK(bbg_prover_create(ctx, psrs, lg, 4, gens, &p));
        bbg_srs_free(psrs); // the prover shares ownership: the tables must survive
        for (int id = BBG_QP_SIGMA_1; id < BBG_QP_LAGRANGE_1; id++) {
            std::vector<uint64_t> poly = scalars(n);
            CK(bbg_prover_set_key_poly(p, id, BBG_FORM_COEFF, poly.data()));
        }
        CK(bbg_prover_finalize_key(p));
        std::vector<std::vector<uint64_t>> wires;
        const uint64_t* wp[4];
        for (int k = 0; k < 4; k++) {
            wires.push_back(scalars(n));
            wp[k] = wires.back().data();
        }
        std::vector<uint64_t> ch = scalars(40);
        auto proof = [&](uint64_t* commitments /* 11 x 12 */) {
            uint64_t ev[32 * 4];
            const int ids16[16] = { 0, 0, 1, 1, 2, 2, 3, 3, 4, 15, 16, 17, 5, 6, 7, 21 }, sh16[16] = { 0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
            const int lin[12] = { 4, 8, 9, 10, 11, 12, 13, 14, 15, 18, 19, 16 };
            const int at_zeta[14] = { 0, 1, 2, 3, 15, 16, 17, 5, 6, 7, 23, 24, 25, 26 }, at_omega[5] = { 0, 1, 2, 3, 4 };
            CK(bbg_prover_round1(p, wp, commitments));
            CK(bbg_prover_round3(p, &ch[0], &ch[4], &ch[8], commitments + 48));
            CK(bbg_prover_round4(p, &ch[20], &ch[24], commitments + 60));
            CK(bbg_prover_evaluate(p, 16, ids16, sh16, &ch[28], ev));
            CK(bbg_prover_linearise(p, 12, lin, &ch[32], &ch[28], ev));
            std::vector<uint64_t> nu = scalars(19);
            g_seed -= 0; // (the opening scalars differ per call on purpose: only the commitments before round 6 are compared)
            CK(bbg_prover_round6(p, 14, at_zeta, nu.data(), 5, at_omega, nu.data() + 56, &ch[28], &ch[36], nullptr, commitments + 108, commitments + 120));
        };
        uint64_t c1[11 * 12], c2[11 * 12], c3[11 * 12];
        proof(c1);
        proof(c2);
        EXPECT(affine(ctx, c1, 9) == affine(ctx, c2, 9)); // W_1..4, Z, T_1..4 repeat (same wires, same challenges)
        EXPECT(bbg_prover_round3(p, &ch[0], &ch[4], &ch[8], c3) == BBG_E_INVALID); // call order is guarded
        std::vector<uint64_t> q = scalars(n);
        CK(bbg_prover_set_key_poly(p, BBG_QP_Q_M, BBG_FORM_COEFF, q.data())); // replaced on the live handle
        EXPECT(bbg_prover_round1(p, wp, c3) == BBG_E_INVALID);                // ... which must be finalized again
        CK(bbg_prover_finalize_key(p));
        proof(c3);
        EXPECT(affine(ctx, c1, 5) == affine(ctx, c3, 5) && !(affine(ctx, c1 + 60, 4) == affine(ctx, c3 + 60, 4))); // the quotient changed
        size_t pb = 0;
        CK(bbg_prover_device_bytes(p, &pb));
        bbg_memory_info info;
        CK(bbg_memory_report(ctx, &info));
        EXPECT(info.live_provers == 1 && info.prover_keys == pb && info.live_srs == 2);
        bbg_prover_destroy(p);
        CK(bbg_memory_report(ctx, &info));
        EXPECT(info.live_provers == 0 && info.live_srs == 1); // the prover held the last reference of its SRS
    }
    // ---- memory trim, then everything again
    {
        size_t released = 0;
        CK(bbg_memory_trim(ctx, 1, &released));
        EXPECT(released > 0);
        std::vector<uint64_t> sc = scalars(N);
        uint64_t out[12];
        CK(bbg_msm(ctx, srs, sc.data(), 0, N, out));
        std::vector<uint64_t> a = scalars(1 << 12);
        CK(bbg_ntt(ctx, a.data(), 12, BBG_FFT, 0, nullptr));
    }
    // ---- device group: four contexts on device 0; per-context buffers grow and shrink
    {
        const int devices[4] = { 0, 0, 0, 0 };
        bbg_multi* m = nullptr;
        CK(bbg_multi_create(devices, 4, &m));
        EXPECT(bbg_multi_count(m) == 4);
        CK(bbg_multi_srs_synth_hashed(m, 0xBB254, N));
        for (size_t n : { (size_t)1000, N, (size_t)3, (size_t)5000 }) {
            std::vector<uint64_t> sc = scalars(n);
            uint64_t got[12], want[12];
            CK(bbg_multi_msm(m, sc.data(), 0, n, got));
            CK(bbg_msm(ctx, srs, sc.data(), 0, n, want));
            EXPECT(affine(ctx, got, 1) == affine(ctx, want, 1));
        }
        for (unsigned lg : { 10u, 13u, 8u }) {
            std::vector<uint64_t> a = scalars((size_t)1 << lg), b = a;
            CK(bbg_multi_ntt(m, a.data(), lg, BBG_COSET_FFT));
            CK(bbg_ntt(ctx, b.data(), lg, BBG_COSET_FFT, 0, nullptr));
            std::vector<uint64_t> ca(a.size()), cb(b.size());
            CK(bbg_field_op(ctx, 0, 4, a.data(), a.data(), ca.data(), a.size() / 4));
            CK(bbg_field_op(ctx, 0, 4, b.data(), b.data(), cb.data(), b.size() / 4));
            EXPECT(ca == cb);
        }
        EXPECT(bbg_multi_set_option(m, "exchange", 1) != BBG_OK); // RCCL needs distinct devices: refused, nothing leaked
        EXPECT(bbg_multi_set_option(m, "no_such_option", 1) == BBG_E_INVALID);
        bbg_multi_destroy(m);
        const int none[1] = { 99 };
        EXPECT(bbg_multi_create(none, 1, &m) != BBG_OK);
    }
    bbg_srs_free(srs);
    bbg_destroy(ctx);
    std::printf("san_capi_driver PASS\n");
    return 0;
}
