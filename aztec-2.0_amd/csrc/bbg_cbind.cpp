// libbbg_cbind.so -- the reference's OWN extern "C" names for this path, exported with exactly its signatures, on top of libbbg.so:
//
//   bbmalloc, bbfree, new_pippenger, delete_pippenger, pippenger_unsafe, g1_sum
//                                         ecc/curves/bn254/scalar_multiplication/c_bind.hpp:9-19, c_bind.cpp:11-46
//   coset_fft_with_generator_shift, ifft, new_evaluation_domain, delete_evaluation_domain
//                                         plonk/proof_system/prover/c_bind.cpp:99-120
//
// A host that speaks the reference's C / WASM offload protocol (barretenberg.js-style workers: fetch a work item's data, call these,
// put the result back) binds this library instead of the reference's and needs no glue.  The names are global and generic (`ifft`,
// `g1_sum`), which is why they live in their own shared object and not in libbbg.so.  Plain C++ host code: no HIP, no torch; every
// call forwards to include/bbg.h.  The reference has no status codes (throw_or_abort aborts under WASM, common/throw_or_abort.hpp):
// a failing call prints bbg_last_error() and aborts.  Device: BBG_CBIND_DEVICE (default 0).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/bbg.h"

namespace {
std::mutex g_mu;
bbg_ctx* g_ctx = nullptr;

[[noreturn]] void die(const char* what)
{
    std::fprintf(stderr, "libbbg_cbind: %s: %s\n", what, bbg_last_error());
    std::abort();
}
bbg_ctx* context()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) {
        const char* dev = std::getenv("BBG_CBIND_DEVICE");
        if (bbg_init(dev ? std::atoi(dev) : 0, &g_ctx) != BBG_OK) die("bbg_init");
    }
    return g_ctx;
}
struct Domain { // what new_evaluation_domain hands out: opaque to the caller, as evaluation_domain* is in the WASM protocol
    unsigned log2n;
};
} // namespace

#define BBG_EXPORT __attribute__((visibility("default")))

extern "C" {

// c_bind.cpp:11-19: 64-byte aligned host memory
BBG_EXPORT void* bbmalloc(size_t size)
{
    void* p = nullptr;
    if (posix_memalign(&p, 64, size ? size : 64) != 0) return nullptr;
    return p;
}
BBG_EXPORT void bbfree(void* ptr) { std::free(ptr); }

// c_bind.cpp:21-24 -> Pippenger(uint8_t const* points, size_t num_points) (pippenger.cpp:7-17): `points` = (num_points - 1) x 64 bytes
// in the transcript encoding (srs/io.cpp:47-67); monomials[0] = G.  The table lives on the device (window tables built once).
BBG_EXPORT void* new_pippenger(uint8_t* points, size_t num_points)
{
    bbg_srs* srs = nullptr;
    if (bbg_srs_register_transcript_buffer(context(), points, num_points, &srs) != BBG_OK) die("new_pippenger");
    return srs;
}
BBG_EXPORT void delete_pippenger(void* pippenger) { bbg_srs_free(static_cast<bbg_srs*>(pippenger)); }

// c_bind.cpp:31-37: *result (g1::element, 96 bytes) = sum_{i < range} scalars[i] * monomials[from + i]
BBG_EXPORT void pippenger_unsafe(void* pippenger, void* scalars, size_t from, size_t range, void* result)
{
    if (bbg_msm(context(), static_cast<bbg_srs*>(pippenger), static_cast<const uint64_t*>(scalars), from, range, static_cast<uint64_t*>(result)) != BBG_OK)
        die("pippenger_unsafe");
}
// c_bind.cpp:39-46: *result = sum of num_points g1::element values
BBG_EXPORT void g1_sum(void* points, const size_t num_points, void* result)
{
    if (bbg_g1_sum(context(), static_cast<const uint64_t*>(points), num_points, static_cast<uint64_t*>(result)) != BBG_OK) die("g1_sum");
}

// prover/c_bind.cpp:109-120: a domain of circuit_size points with its lookup table built
BBG_EXPORT void* new_evaluation_domain(size_t circuit_size)
{
    unsigned lg = 0;
    while (((size_t)1 << lg) < circuit_size) lg++;
    if (((size_t)1 << lg) != circuit_size || lg > 28) {
        std::fprintf(stderr, "libbbg_cbind: new_evaluation_domain: size must be a power of two <= 2^28\n");
        std::abort();
    }
    if (bbg_ntt_prepare(context(), lg) != BBG_OK) die("new_evaluation_domain");
    return new Domain{ lg };
}
BBG_EXPORT void delete_evaluation_domain(void* domain) { delete static_cast<Domain*>(domain); }

// prover/c_bind.cpp:99-107 (types: fr* = 4 x u64 Montgomery limbs; evaluation_domain* = the handle above)
BBG_EXPORT void coset_fft_with_generator_shift(void* coefficients, void* constant, void* domain)
{
    if (bbg_ntt(context(), static_cast<uint64_t*>(coefficients), static_cast<Domain*>(domain)->log2n, BBG_COSET_FFT_WITH_GENERATOR_SHIFT, 0,
                static_cast<const uint64_t*>(constant)) != BBG_OK)
        die("coset_fft_with_generator_shift");
}
BBG_EXPORT void ifft(void* coefficients, void* domain)
{
    if (bbg_ntt(context(), static_cast<uint64_t*>(coefficients), static_cast<Domain*>(domain)->log2n, BBG_IFFT, 0, nullptr) != BBG_OK) die("ifft");
}

} // extern "C"
