// bbg_barretenberg_shim.cpp -- the header-compatible C++ side of the drop-in boundary.
//
// Compiled INSIDE a barretenberg build (it includes barretenberg's own headers for fr, g1, evaluation_domain and
// pippenger_runtime_state; nothing from the reference is copied here).  It provides, with the reference's exact
// signatures and value conventions,
//
//   barretenberg::scalar_multiplication::pippenger / pippenger_unsafe       (scalar_multiplication.hpp:139-148)
//   barretenberg::polynomial_arithmetic::fft / ifft / coset_fft (x2) / coset_ifft / fft_with_constant /
//       coset_fft_with_constant / coset_fft_with_generator_shift / ifft_with_constant
//                                                                            (polynomial_arithmetic.hpp:23-39)
//
// forwarding to libbbg.so's C ABI (include/bbg.h).  Two ways to put it in front of the stock prover WITHOUT editing
// a single reference file:
//   (a) link-time wrapping (what shim/Makefile and shim/shim_check.cpp do):  every definition below is emitted under
//       the symbol name  __wrap_<mangled reference name>  and the final link passes  -Wl,--wrap=<mangled name>  for
//       each entry point (shim/wrap_flags.txt).  work_queue::process_queue, polynomial::fft, compute_verification_key
//       ... then reach the GPU; the reference's own CPU bodies stay reachable as __real_<mangled name>.
//   (b) or drop the two reference TUs' bodies for these functions and compile this file with -DBBG_SHIM_DEFINE_DIRECT,
//       which emits the plain barretenberg:: symbols instead.
//
// Semantics kept from the reference: scalars / coefficients in Montgomery form, any representative in [0, 2p);
// `points` is the interleaved endomorphism table of Pippenger (stride 2) and `points + 2*from` addresses a sub-range
// (pippenger.cpp:27-31); the MSM result is a Jacobian g1::element (its representation differs from the CPU one, the
// group element is identical -- compare through g1::affine_element, as the prover does, work_queue.hpp:233-239);
// FFTs are in place on coeffs[domain.size]; `pippenger_runtime_state` is accepted and ignored (the scratch arena lives
// on the device); a failing call throws std::runtime_error like throw_or_abort (common/throw_or_abort.hpp:5-13).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

#include <ecc/curves/bn254/scalar_multiplication/scalar_multiplication.hpp>
#include <polynomials/evaluation_domain.hpp>
#include <polynomials/polynomial_arithmetic.hpp>

#include "../include/bbg.h"
#include "bbg_shim_verify.hpp"

namespace {
using barretenberg::evaluation_domain;
using barretenberg::fr;
using barretenberg::g1;
using barretenberg::scalar_multiplication::pippenger_runtime_state;

// Device copies of Pippenger point tables, keyed by the table's ADDRESS.  Two kinds of entry:
//   registered  by bbg_shim_register_point_table from the place that owns the table (Pippenger / ReferenceString construction,
//               pippenger.cpp:7-25) and dropped by bbg_shim_unregister_point_table from its destructor (:33-36): the owner vouches
//               for the lifetime, lookups never touch host memory.
//   implicit    a table first seen inside pippenger*() of a build that calls neither hook.  Nothing announces its death, so an entry
//               remembers sampled points and is reused only when every sample INSIDE THE RANGE THE CURRENT CALL PASSES (memory the
//               caller guarantees readable right now -- freed or unmapped memory is never probed) still matches; at most
//               MAX_IMPLICIT of them are kept (least recently used goes first), and tables below MIN_CACHED_POINTS -- the
//               verifier's per-proof element table, verifier.cpp:165-170 -- are uploaded, used and freed within the call.
// The samples are only the quick look that spares a GPU call over an obviously dead entry.  What makes a cached table SAFE (round 6,
// bbg_shim_verify.hpp): every entry keeps a 64-bit hash of each of its points, taken when the device copy was made, and every MSM over a
// cached entry -- registered or implicit -- re-hashes ALL the points of the range it passes on host threads while the GPU computes; the
// result is returned only if every point still hashes to what was recorded, otherwise the entry is dropped, the table uploaded again and
// the MSM repeated.  One rewritten point anywhere in the range can therefore not produce a commitment over the stale copy.
// (BBG_SHIM_VERIFY=sample: samples only, for hosts whose tables are immutable.)
struct ShimState {
    static constexpr size_t MAX_IMPLICIT = 4, MIN_CACHED_POINTS = 4096, SAMPLES = 1024;
    std::mutex mu;
    bbg_ctx* ctx = nullptr;
    struct Entry {
        const g1::affine_element* base = nullptr;
        size_t n = 0;
        bbg_srs* srs = nullptr;
        bool registered = false;
        uint64_t last_use = 0;
        std::vector<std::pair<size_t, g1::affine_element>> samples; // the quick look
        std::vector<uint64_t> point_hash;                           // full mode: one hash per point, of the memory the device copy was made from
        static uint64_t hash_point(const g1::affine_element* base, size_t i)
        {
            return bbg_shim_verify::hash_words(reinterpret_cast<const uint64_t*>(&base[2 * i]), sizeof(g1::affine_element) / 8, (uint64_t)i);
        }
        void take_hashes()
        {
            point_hash.resize(n);
            constexpr size_t CH = 4096;
            bbg_shim_verify::parallel_chunks((n + CH - 1) / CH, [&](size_t k) {
                for (size_t i = k * CH, e = std::min(n, i + CH); i < e; i++) point_hash[i] = hash_point(base, i);
            });
        }
        // every point of [from, from + count) -- memory the current call passes -- still hashes to what was recorded
        bool range_verifies(size_t from, size_t count) const
        {
            if (point_hash.size() != n || from > n || count > n - from) return false;
            constexpr size_t CH = 4096;
            std::atomic<bool> ok{ true };
            bbg_shim_verify::parallel_chunks((count + CH - 1) / CH, [&](size_t k) {
                if (!ok.load(std::memory_order_relaxed)) return;
                for (size_t i = from + k * CH, e = std::min(from + count, i + CH); i < e; i++)
                    if (hash_point(base, i) != point_hash[i]) {
                        ok.store(false, std::memory_order_relaxed);
                        return;
                    }
            });
            return ok.load();
        }
        // true iff at least one sample lies in [from, from + count) and all of those match the host table
        bool range_still_matches(size_t from, size_t count) const
        {
            bool any = false;
            for (const auto& sm : samples) {
                if (sm.first < from || sm.first >= from + count) continue;
                if (std::memcmp((const void*)&base[2 * sm.first], (const void*)&sm.second, sizeof(g1::affine_element)) != 0) return false;
                any = true;
            }
            return any;
        }
    };
    std::map<const g1::affine_element*, Entry> tables; // keyed by table base pointer (get_monomials() identity)
    uint64_t clock = 0;
    // Several GPUs: an MSM of at least multi_min points over a cached table is sharded by point range over the device group
    // (bbg_multi_msm).  BBG_SHIM_DEVICES = comma-separated device indices (repeats allowed) or "all" (default when more than
    // one device is visible); BBG_SHIM_MULTI_MIN_POINTS = threshold (default 2^22: below it one GPU finishes before the others start).
    bbg_multi* multi = nullptr;
    bool multi_probed = false;
    size_t multi_min = (size_t)1 << 22;
    const g1::affine_element* multi_table = nullptr; // the table whose shards the group currently holds
    size_t multi_points = 0;
    ~ShimState()
    {
        if (multi) bbg_multi_destroy(multi);
        for (auto& kv : tables) bbg_srs_free(kv.second.srs);
        if (ctx) bbg_destroy(ctx);
    }
};
ShimState& state()
{
    static ShimState s;
    return s;
}
[[noreturn]] void fail(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + bbg_last_error());
}
bbg_ctx* context()
{
    ShimState& s = state();
    if (!s.ctx && bbg_init(0, &s.ctx) != BBG_OK) fail("bbg_init");
    return s.ctx;
}
bbg_multi* device_group()
{
    ShimState& s = state();
    if (s.multi_probed) return s.multi;
    s.multi_probed = true;
    std::vector<int> devices;
    const char* env = std::getenv("BBG_SHIM_DEVICES");
    if (env && std::string(env) != "all") {
        for (const char* p = env; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
    } else {
        const int n = bbg_device_count();
        if (n > 1 || env)
            for (int d = 0; d < n; d++) devices.push_back(d);
    }
    if (const char* mn = std::getenv("BBG_SHIM_MULTI_MIN_POINTS")) s.multi_min = (size_t)std::strtoull(mn, nullptr, 10);
    if (devices.size() > 1 && bbg_multi_create(devices.data(), (int)devices.size(), &s.multi) != BBG_OK) fail("bbg_multi_create");
    // BBG_SHIM_EXCHANGE=rccl: the group's exchanges (all-gather of the MSM partials) go through RCCL instead of peer copies
    if (const char* ex = std::getenv("BBG_SHIM_EXCHANGE"))
        if (s.multi && std::string(ex) == "rccl" && bbg_multi_set_option(s.multi, "exchange", 1) != BBG_OK) fail("bbg_multi_set_option(exchange = rccl)");
    return s.multi;
}
bbg_srs* upload(const g1::affine_element* points, size_t num_points)
{
    bbg_srs* srs = nullptr;
    if (bbg_srs_register(context(), reinterpret_cast<const uint64_t*>(points), num_points, sizeof(g1::affine_element) * 2, &srs) != BBG_OK)
        fail("bbg_srs_register");
    return srs;
}
// the entry whose table contains `points`, or end()
std::map<const g1::affine_element*, ShimState::Entry>::iterator containing(const g1::affine_element* points)
{
    ShimState& s = state();
    auto it = s.tables.upper_bound(points);
    if (it == s.tables.begin()) return s.tables.end();
    --it;
    return (size_t)(points - it->second.base) < 2 * it->second.n ? it : s.tables.end();
}
void drop(std::map<const g1::affine_element*, ShimState::Entry>::iterator it)
{
    if (state().multi_table == it->second.base) state().multi_table = nullptr; // its shards are re-uploaded if the address is seen again
    bbg_srs_free(it->second.srs);
    state().tables.erase(it);
}
void take_samples(ShimState::Entry& e)
{
    const size_t step = e.n > ShimState::SAMPLES ? e.n / ShimState::SAMPLES : 1;
    for (size_t i = 0; i < e.n; i += step) e.samples.emplace_back(i, e.base[2 * i]);
    e.samples.emplace_back(e.n - 1, e.base[2 * (e.n - 1)]);
    if (bbg_shim_verify::full_mode()) e.take_hashes();
}
// The full check of one call's range, on a host thread of its own (which fans out) while the GPU works on the call; joined before the call
// returns -- also when the call throws.
struct RangeCheck {
    std::thread th;
    bool ok = true;
    void start(const ShimState::Entry& e, size_t from, size_t count)
    {
        th = std::thread([this, &e, from, count]() { ok = e.range_verifies(from, count); });
    }
    bool finish()
    {
        if (th.joinable()) th.join();
        return ok;
    }
    ~RangeCheck()
    {
        if (th.joinable()) th.join();
    }
};
// every cached entry whose host range overlaps [table, table + 2 num_points) except the one AT `table`: the caller vouches that this
// memory holds its table now, so whatever else was remembered there is dead (a table freed without the unregister hook)
void drop_overlapping(const g1::affine_element* table, size_t num_points)
{
    ShimState& s = state();
    for (auto it = s.tables.begin(); it != s.tables.end();) {
        const ShimState::Entry& e = it->second;
        const bool overlaps = e.base != table && e.base < table + 2 * num_points && table < e.base + 2 * e.n;
        auto next = std::next(it);
        if (overlaps) drop(it);
        it = next;
    }
}
// owner-announced table: uploaded once, valid until bbg_shim_unregister_point_table
bbg_srs* register_table(const g1::affine_element* table, size_t num_points)
{
    ShimState& s = state();
    drop_overlapping(table, num_points);
    auto it = s.tables.find(table);
    if (it != s.tables.end()) {
        if (it->second.registered && it->second.n >= num_points && it->second.range_still_matches(0, num_points) &&
            (!bbg_shim_verify::full_mode() || it->second.range_verifies(0, num_points)))
            return it->second.srs;
        drop(it); // an implicit entry at this address, a shorter registration, or other contents than were registered: replace
    }
    ShimState::Entry e;
    e.base = table;
    e.n = num_points;
    e.srs = upload(table, num_points);
    e.registered = true;
    take_samples(e);
    s.tables[table] = std::move(e);
    return s.tables[table].srs;
}
// Device SRS for an MSM over points[0 .. 2 num_points); *from = index of points[0] in it; *transient = the caller frees it after use.
bbg_srs* lookup_srs(const g1::affine_element* points, size_t num_points, size_t& from, bool& transient, bool& fresh)
{
    ShimState& s = state();
    transient = false;
    fresh = false; // true: the entry was uploaded (and hashed) inside this call
    auto it = containing(points);
    bool registered_but_short = false;
    if (it != s.tables.end()) {
        ShimState::Entry& e = it->second;
        const size_t off = (size_t)(points - e.base);
        // Registered entries are sampled as well: an owner that forgot the unregister hook must not turn into wrong commitments.  Only
        // memory inside the range THIS call passes is compared.
        if (off % 2 == 0 && off / 2 + num_points <= e.n && e.range_still_matches(off / 2, num_points)) {
            e.last_use = ++s.clock;
            from = off / 2;
            return e.srs;
        }
        // An entry that no longer describes this memory (any kind), or an implicit one that is too short: forget it.  A REGISTERED
        // table that still matches but is asked for more points than were announced is served from a transient copy.
        const bool stale = off % 2 != 0 || !e.range_still_matches(off / 2, std::min(num_points, e.n - std::min(e.n, off / 2)));
        registered_but_short = e.registered && !stale;
        if (!e.registered || stale) drop(it);
    }
    from = 0;
    if (num_points < ShimState::MIN_CACHED_POINTS || registered_but_short) {
        transient = true;
        return upload(points, num_points);
    }
    size_t implicit = 0;
    auto oldest = s.tables.end();
    for (auto jt = s.tables.begin(); jt != s.tables.end(); ++jt) {
        if (jt->second.registered) continue;
        implicit++;
        if (oldest == s.tables.end() || jt->second.last_use < oldest->second.last_use) oldest = jt;
    }
    if (implicit >= ShimState::MAX_IMPLICIT) drop(oldest);
    ShimState::Entry e;
    e.base = points;
    e.n = num_points;
    e.srs = upload(points, num_points);
    e.last_use = ++s.clock;
    take_samples(e);
    s.tables[points] = std::move(e);
    fresh = true;
    return s.tables[points].srs;
}
uint64_t g_stale_tables = 0; // cached tables found rewritten by the full check and uploaded again (bbg_shim_stale_tables)
g1::element msm(fr* scalars, g1::affine_element* points, size_t n)
{
    std::lock_guard<std::mutex> lk(state().mu);
    g1::element out;
    if (n == 0) {
        out = g1::one;
        out.self_set_infinity();
        return out;
    }
    // at most two passes: the second only after the full check found the cached copy stale -- it uploads (fresh) and needs no check
    for (int pass = 0;; pass++) {
        size_t from = 0;
        bool transient = false, fresh = false;
        bbg_srs* srs = lookup_srs(points, n, from, transient, fresh);
        RangeCheck check; // joined when it goes out of scope, whatever happens below
        if (!transient && !fresh && bbg_shim_verify::full_mode()) check.start(containing(points)->second, from, n);
        bool done = false;
        if (!transient) {
            bbg_multi* group = device_group();
            if (group && n >= state().multi_min) { // large MSM over a cached table: every GPU of the group takes a point range
                ShimState& s = state();
                auto it = containing(points);
                const ShimState::Entry& e = it->second;
                if (s.multi_table != e.base || s.multi_points != e.n) {
                    if (bbg_multi_srs_register(group, reinterpret_cast<const uint64_t*>(e.base), e.n, sizeof(g1::affine_element) * 2) != BBG_OK)
                        fail("bbg_multi_srs_register");
                    s.multi_table = e.base;
                    s.multi_points = e.n;
                }
                if (bbg_multi_msm(group, reinterpret_cast<const uint64_t*>(scalars), from, n, reinterpret_cast<uint64_t*>(&out)) != BBG_OK) fail("bbg_multi_msm");
                done = true;
            }
        }
        if (!done) {
            const int rc = bbg_msm(context(), srs, reinterpret_cast<const uint64_t*>(scalars), from, n, reinterpret_cast<uint64_t*>(&out));
            if (transient) bbg_srs_free(srs);
            if (rc != BBG_OK) fail("bbg_msm");
        }
        if (check.finish()) return out;
        // the host table is not what the device copy was made from: forget the copy and compute the MSM again over the memory as it is now
        if (pass > 0) throw std::runtime_error("bbg_shim: a point table keeps changing while it is used");
        g_stale_tables++;
        auto it = containing(points);
        if (it != state().tables.end()) drop(it);
    }
}
void ntt(fr* coeffs, const evaluation_domain& d, int op, const fr* constant)
{
    std::lock_guard<std::mutex> lk(state().mu);
    if (bbg_ntt(context(), reinterpret_cast<uint64_t*>(coeffs), (unsigned)d.log2_size, op, d.generator_size,
                reinterpret_cast<const uint64_t*>(constant)) != BBG_OK)
        fail("bbg_ntt");
}
} // namespace

#ifdef BBG_SHIM_DEFINE_DIRECT
#define SHIM_NAME(mangled) asm(mangled)
#else
#define SHIM_NAME(mangled) asm("__wrap_" mangled)
#endif

// Lifetime hooks for the place that owns the table (Pippenger / ReferenceString construction and destruction,
// pippenger.cpp:7-25, :33-36): upload once before the first proof, release with the table.
extern "C" void bbg_shim_register_point_table(const void* endo_table, size_t num_points)
{
    std::lock_guard<std::mutex> lk(state().mu);
    (void)register_table(static_cast<const g1::affine_element*>(endo_table), num_points);
}
extern "C" void bbg_shim_unregister_point_table(const void* endo_table)
{
    std::lock_guard<std::mutex> lk(state().mu);
    auto it = state().tables.find(static_cast<const g1::affine_element*>(endo_table));
    if (it != state().tables.end()) drop(it);
}
// For the resident prover (shim/bbg_resident_prover.hpp): the shim's device context, and the device SRS of a table its owner has
// announced (registers it if it has not been: the caller -- a ResidentKey -- holds the proving key and with it the reference string).
extern "C" bbg_ctx* bbg_shim_context(void)
{
    std::lock_guard<std::mutex> lk(state().mu);
    return context();
}
extern "C" bbg_srs* bbg_shim_srs_for(const void* endo_table, size_t num_points)
{
    std::lock_guard<std::mutex> lk(state().mu);
    return register_table(static_cast<const g1::affine_element*>(endo_table), num_points);
}
// cached tables the full content check found rewritten and uploaded again (tests; BBG_SHIM_VERIFY=sample never counts)
extern "C" uint64_t bbg_shim_stale_tables(void)
{
    std::lock_guard<std::mutex> lk(state().mu);
    return g_stale_tables;
}
// number of cached device tables (tests: bounded cache, transient tables not retained)
extern "C" size_t bbg_shim_cached_tables(void)
{
    std::lock_guard<std::mutex> lk(state().mu);
    return state().tables.size();
}

namespace bbg_shim {
g1::element pippenger(fr* scalars, g1::affine_element* points, const size_t num_points, pippenger_runtime_state&, bool)
    SHIM_NAME("_ZN12barretenberg21scalar_multiplication9pippengerEPNS_5fieldINS_13Bn254FrParamsEEEPNS_14group_elements14affine_elementINS1_INS_13Bn254FqParamsEEES3_NS_13Bn254G1ParamsEEEmRNS0_23pippenger_runtime_stateEb");
g1::element pippenger_unsafe(fr* scalars, g1::affine_element* points, const size_t num_points, pippenger_runtime_state&)
    SHIM_NAME("_ZN12barretenberg21scalar_multiplication16pippenger_unsafeEPNS_5fieldINS_13Bn254FrParamsEEEPNS_14group_elements14affine_elementINS1_INS_13Bn254FqParamsEEES3_NS_13Bn254G1ParamsEEEmRNS0_23pippenger_runtime_stateE");
void fft(fr* coeffs, const evaluation_domain& domain)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic3fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void ifft(fr* coeffs, const evaluation_domain& domain)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic4ifftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void coset_fft(fr* coeffs, const evaluation_domain& domain)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic9coset_fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void coset_fft_split(fr* coeffs, const evaluation_domain& small_domain, const evaluation_domain& large_domain, const size_t ext)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic9coset_fftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainES7_m");
void coset_ifft(fr* coeffs, const evaluation_domain& domain)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic10coset_ifftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainE");
void fft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& value)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic17fft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void coset_fft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& constant)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic23coset_fft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void coset_fft_with_generator_shift(fr* coeffs, const evaluation_domain& domain, const fr& constant)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic30coset_fft_with_generator_shiftEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
void ifft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& value)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic18ifft_with_constantEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainERKS3_");
// the O(n) helpers between the FFTs and the MSMs (polynomial_arithmetic.hpp:16,56,64)
fr evaluate(const fr* coeffs, const fr& z, const size_t n)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic8evaluateEPKNS_5fieldINS_13Bn254FrParamsEEERS4_m");
fr compute_kate_opening_coefficients(const fr* src, fr* dest, const fr& z, const size_t n)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic33compute_kate_opening_coefficientsEPKNS_5fieldINS_13Bn254FrParamsEEEPS3_RS4_m");
void divide_by_pseudo_vanishing_polynomial(fr* coeffs, const evaluation_domain& src_domain, const evaluation_domain& target_domain,
                                           const size_t num_roots_cut_out_of_vanishing_polynomial)
    SHIM_NAME("_ZN12barretenberg21polynomial_arithmetic37divide_by_pseudo_vanishing_polynomialEPNS_5fieldINS_13Bn254FrParamsEEERKNS_17evaluation_domainES7_m");

g1::element pippenger(fr* scalars, g1::affine_element* points, const size_t num_points, pippenger_runtime_state&, bool)
{
    return msm(scalars, points, num_points);
}
g1::element pippenger_unsafe(fr* scalars, g1::affine_element* points, const size_t num_points, pippenger_runtime_state&)
{
    return msm(scalars, points, num_points);
}
void fft(fr* coeffs, const evaluation_domain& domain) { ntt(coeffs, domain, BBG_FFT, nullptr); }
void ifft(fr* coeffs, const evaluation_domain& domain) { ntt(coeffs, domain, BBG_IFFT, nullptr); }
void coset_fft(fr* coeffs, const evaluation_domain& domain) { ntt(coeffs, domain, BBG_COSET_FFT, nullptr); }
void coset_fft_split(fr* coeffs, const evaluation_domain& small_domain, const evaluation_domain&, const size_t ext)
{
    std::lock_guard<std::mutex> lk(state().mu);
    if (bbg_coset_fft_split(context(), reinterpret_cast<uint64_t*>(coeffs), (unsigned)small_domain.log2_size, ext) != BBG_OK)
        fail("bbg_coset_fft_split");
}
void coset_ifft(fr* coeffs, const evaluation_domain& domain) { ntt(coeffs, domain, BBG_COSET_IFFT, nullptr); }
void fft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& value) { ntt(coeffs, domain, BBG_FFT_WITH_CONSTANT, &value); }
void coset_fft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& constant)
{
    ntt(coeffs, domain, BBG_COSET_FFT_WITH_CONSTANT, &constant);
}
void coset_fft_with_generator_shift(fr* coeffs, const evaluation_domain& domain, const fr& constant)
{
    ntt(coeffs, domain, BBG_COSET_FFT_WITH_GENERATOR_SHIFT, &constant);
}
void ifft_with_constant(fr* coeffs, const evaluation_domain& domain, const fr& value) { ntt(coeffs, domain, BBG_IFFT_WITH_CONSTANT, &value); }
fr evaluate(const fr* coeffs, const fr& z, const size_t n)
{
    std::lock_guard<std::mutex> lk(state().mu);
    fr out;
    if (bbg_poly_evaluate(context(), reinterpret_cast<const uint64_t*>(coeffs), n, reinterpret_cast<const uint64_t*>(&z),
                          reinterpret_cast<uint64_t*>(&out)) != BBG_OK)
        fail("bbg_poly_evaluate");
    return out;
}
fr compute_kate_opening_coefficients(const fr* src, fr* dest, const fr& z, const size_t n)
{
    std::lock_guard<std::mutex> lk(state().mu);
    fr f;
    if (bbg_kate_opening(context(), reinterpret_cast<const uint64_t*>(src), reinterpret_cast<uint64_t*>(dest), n,
                         reinterpret_cast<const uint64_t*>(&z), reinterpret_cast<uint64_t*>(&f)) != BBG_OK)
        fail("bbg_kate_opening");
    return f;
}
void divide_by_pseudo_vanishing_polynomial(fr* coeffs, const evaluation_domain& src_domain, const evaluation_domain& target_domain,
                                           const size_t num_roots_cut_out_of_vanishing_polynomial)
{
    std::lock_guard<std::mutex> lk(state().mu);
    if (bbg_divide_by_pseudo_vanishing(context(), reinterpret_cast<uint64_t*>(coeffs), (unsigned)src_domain.log2_size,
                                       (unsigned)target_domain.log2_size, num_roots_cut_out_of_vanishing_polynomial) != BBG_OK)
        fail("bbg_divide_by_pseudo_vanishing");
}
} // namespace bbg_shim
