// bbg_prover_wrap.cpp -- the resident prover behind the reference's OWN entry point, with zero source edits.
//
// Every caller in the reference makes a proof with waffle::ProverBase<settings>::construct_proof()
// (plonk/proof_system/prover/prover.cpp:420-436).  The four instantiations (:445-448) are out-of-line weak symbols and the header
// declares them `extern template` (prover.hpp:99-102), so every call site references the symbol -- which makes the function wrappable
// at link time exactly like the fourteen MSM / FFT entry points of bbg_barretenberg_shim.cpp.  Linking this file plus
//
//     $(cat shim/wrap_flags.txt) $(cat shim/wrap_flags_prover.txt)
//
// in front of an unmodified barretenberg build sends `prover.construct_proof()` to bbg_shim::construct_proof
// (bbg_resident_prover.hpp: every O(n) step of the proof on the device, transcript and challenge algebra on the reference's own
// objects).  A prover whose widget list the device rounds do not implement falls through to the reference's body (__real_...), whose
// MSMs and FFTs still reach the GPU through the wrapped entry points.
//
// The device copy of a proving key (bbg_shim::ResidentKey) is cached per `proving_key` object:
//   * keyed by the key's address; the entry HOLDS the std::shared_ptr<proving_key>, so the address cannot be reused while the entry lives;
//   * an entry whose key nobody else holds any more (use_count() == 1: every prover / composer that shared it is gone) is released at the
//     next proof, or at once by bbg_shim_resident_trim() -- "key destroyed -> entry released" without a hook in proving_key's (implicit,
//     inline, hence unwrappable) destructor;
//   * bounded by device bytes: least recently used entries go first when the total exceeds the budget (BBG_SHIM_RESIDENT_MAX_BYTES,
//     default half of the device's memory); the key being proved with is never evicted;
//   * BBG_SHIM_RESIDENT=0 (or bbg_shim_resident_set_enabled(0)) opts out: construct_proof() is the reference's body again.
//
// Differences from the reference body a caller could observe: the host arrays of the witness are left as the composer produced them
// (wires in Lagrange form, blinding rows written) instead of in coefficient form, and key->wire_ffts / quotient_large / linear_poly /
// opening_poly are not filled in -- the proof is the only output.  The blinding scalars come from the kernel CSPRNG in one call per
// element (bbg_resident_prover.hpp: os_random_fr) instead of 64 std::random_device reads; bbg_shim_resident_set_random plugs a
// caller's source (the parity tests replay the scalars a reference proof drew and compare the proofs byte for byte).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

#include "bbg_resident_prover.hpp"

namespace {
using bbg_shim::ResidentKey;

struct ResidentCache {
    std::mutex mu;
    struct Entry {
        std::unique_ptr<ResidentKey> rk;
        size_t bytes = 0;
        uint64_t last_use = 0;
        uint64_t fingerprint = 0; // of the host key polynomials the device copy was made from (key_fingerprint below)
    };
    std::map<const waffle::proving_key*, Entry> entries;
    uint64_t clock = 0;
    bool enabled = true;
    size_t budget = 0; // 0 = not initialised yet
    uint64_t proofs = 0, fallbacks = 0, evictions = 0, reuploads = 0;
    void (*draw)(void* user, uint64_t out[4]) = nullptr;
    void* draw_user = nullptr;
    ResidentCache()
    {
        if (const char* e = std::getenv("BBG_SHIM_RESIDENT")) enabled = !(e[0] == '0' && e[1] == 0);
        if (const char* b = std::getenv("BBG_SHIM_RESIDENT_MAX_BYTES")) budget = (size_t)std::strtoull(b, nullptr, 10);
    }
    size_t total() const
    {
        size_t t = 0;
        for (const auto& kv : entries) t += kv.second.bytes;
        return t;
    }
    // entries whose proving key only the cache still holds
    size_t sweep()
    {
        size_t dropped = 0;
        for (auto it = entries.begin(); it != entries.end();) {
            if (it->second.rk->key().use_count() == 1) {
                it = entries.erase(it);
                dropped++;
            } else
                ++it;
        }
        return dropped;
    }
    // least recently used entries go until the total fits the budget; `keep` stays
    void evict_to_budget(const waffle::proving_key* keep)
    {
        while (total() > budget) {
            auto oldest = entries.end();
            for (auto it = entries.begin(); it != entries.end(); ++it)
                if (it->first != keep && (oldest == entries.end() || it->second.last_use < oldest->second.last_use)) oldest = it;
            if (oldest == entries.end()) return;
            entries.erase(oldest);
            evictions++;
        }
    }
};
// Never destroyed: at process exit the entries would otherwise run ~proving_key -> ~ProverReferenceString -> the shim's table registry
// after that registry's own static destruction (the order between two translation units' function statics is the order of first use,
// reversed).  Device memory goes back to the driver with the process; a host that wants it earlier calls bbg_shim_resident_clear().
ResidentCache& cache()
{
    static ResidentCache* c = new ResidentCache;
    return *c;
}

// The device copy is only as good as the host polynomials it was made from.  A host that rewrites a selector or a permutation polynomial
// of a key it has already proved with (the reference never does: compute_proving_key fills them once, composer_base.hpp) would otherwise
// get proofs over the OLD polynomials without an error.  Cheap check per proof: for every key polynomial the buffer address, its size and
// 16 evenly spaced coefficients go into a 64-bit FNV-1a; a different value re-uploads the key.  A SAMPLED fingerprint: it catches
// reallocation, resizing and any rewrite that touches the sampled rows (a recomputed polynomial does, a single poked coefficient may not) --
// keys must be treated as immutable once proved with; this is a tripwire, not a guarantee (INTEGRATION.md 2a').
uint64_t key_fingerprint(const waffle::proving_key& key)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) {
        for (int b = 0; b < 8; b++) {
            h ^= (v >> (8 * b)) & 0xff;
            h *= 1099511628211ull;
        }
    };
    auto take = [&](const barretenberg::polynomial& poly) {
        const size_t size = poly.get_size();
        mix((uint64_t)(uintptr_t)&poly[0]);
        mix((uint64_t)size);
        if (size == 0) return;
        for (size_t k = 0; k < 16; k++) {
            const uint64_t* w = reinterpret_cast<const uint64_t*>(&poly[(size - 1) * k / 15]);
            for (int l = 0; l < 4; l++) mix(w[l]);
        }
    };
    mix(key.n);
    mix(key.num_public_inputs);
    for (const auto& kv : key.constraint_selectors) take(kv.second);
    for (const auto& kv : key.permutation_selectors) take(kv.second);
    return h;
}

barretenberg::fr draw_adapter(void*)
{
    ResidentCache& c = cache();
    barretenberg::fr v;
    c.draw(c.draw_user, reinterpret_cast<uint64_t*>(&v));
    return v;
}

template <typename settings>
waffle::plonk_proof& resident_or_real(waffle::ProverBase<settings>* self, waffle::plonk_proof& (*real)(waffle::ProverBase<settings>*))
{
    ResidentCache& c = cache();
    std::unique_lock<std::mutex> lk(c.mu);
    if (!c.enabled || !self->key || !bbg_shim::resident_supported(*self)) {
        lk.unlock();
        return real(self);
    }
    c.sweep();
    if (c.budget == 0) {
        bbg_memory_info info;
        c.budget = bbg_memory_report(bbg_shim_context(), &info) == BBG_OK && info.device_total ? info.device_total / 2 : ((size_t)64 << 30);
    }
    const waffle::proving_key* id = self->key.get();
    const uint64_t fingerprint = key_fingerprint(*self->key);
    auto it = c.entries.find(id);
    if (it != c.entries.end() && it->second.fingerprint != fingerprint) { // the host polynomials changed under a cached key: upload again
        c.entries.erase(it);
        it = c.entries.end();
        c.reuploads++;
    }
    if (it == c.entries.end()) {
        std::unique_ptr<ResidentKey> rk;
        for (int attempt = 0; attempt < 2 && !rk; attempt++) {
            try {
                rk = std::make_unique<ResidentKey>(self->key, settings::program_width);
            } catch (const std::exception& e) {
                if (attempt == 0 && !c.entries.empty()) { // most likely device memory: give back every other key and try once more
                    c.evictions += c.entries.size();
                    c.entries.clear();
                    continue;
                }
                // the reference's body still proves (its MSMs / FFTs are wrapped onto the GPU): slower, never wrong
                if (c.fallbacks++ == 0) std::fprintf(stderr, "bbg_shim: resident key not created (%s); construct_proof() takes the reference body\n", e.what());
                lk.unlock();
                return real(self);
            }
        }
        ResidentCache::Entry e;
        size_t bytes = 0;
        if (bbg_prover_device_bytes(rk->handle(), &bytes) != BBG_OK) bytes = 0;
        e.bytes = bytes;
        e.fingerprint = fingerprint;
        e.rk = std::move(rk);
        it = c.entries.emplace(id, std::move(e)).first;
    }
    it->second.last_use = ++c.clock;
    c.evict_to_budget(id);
    bbg_shim::ResidentOptions opt;
    if (c.draw) opt.random = &draw_adapter;
    c.proofs++;
    // the lock is held for the whole proof: the reference's prover is not re-entrant either (process-global FFT scratch,
    // polynomial_arithmetic.cpp:13-34), and the shim's device context is one stream
    try {
        return bbg_shim::construct_proof(*self, *it->second.rk, opt);
    } catch (const std::exception& e) {
        // A device error in the middle of a proof (out of memory, a HIP failure): the reference's construct_proof() never throws for
        // that, so neither does this one.  The resident rounds have written only the blinding rows of the host witness (which the
        // reference's preamble draws again) and the transcript, which ProverBase::reset() (prover.cpp:438-442) rebuilds from its
        // manifest -- so the reference body can still make the proof from the start.  The key's device copy goes: whatever state the
        // failed round left in it is not trusted.
        std::fprintf(stderr, "bbg_shim: resident proof failed (%s); construct_proof() repeats it with the reference body\n", e.what());
        c.entries.erase(id);
        c.fallbacks++;
        lk.unlock();
        self->reset();
        return real(self);
    }
}
} // namespace

extern "C" {
// 1 = construct_proof() takes the resident path where it can (default), 0 = the reference body
void bbg_shim_resident_set_enabled(int on)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    cache().enabled = on != 0;
}
int bbg_shim_resident_enabled(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    return cache().enabled ? 1 : 0;
}
// source of the blinding scalars (Montgomery-form fr, 4 limbs); NULL restores the kernel CSPRNG
void bbg_shim_resident_set_random(void (*draw)(void* user, uint64_t out[4]), void* user)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    cache().draw = draw;
    cache().draw_user = user;
}
// device-byte budget of the key cache (least recently used keys are released above it)
void bbg_shim_resident_set_budget(size_t bytes)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    cache().budget = bytes;
    if (bytes) cache().evict_to_budget(nullptr);
}
size_t bbg_shim_resident_cached_keys(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    return cache().entries.size();
}
size_t bbg_shim_resident_bytes(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    return cache().total();
}
// releases the device copies of keys nobody but the cache holds any more; returns the number of entries left
size_t bbg_shim_resident_trim(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    cache().sweep();
    return cache().entries.size();
}
// releases every entry (a process that is done proving, or a test)
void bbg_shim_resident_clear(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    cache().entries.clear();
}
// counters: [0] proofs through the resident path, [1] proofs that fell back to the reference body (failed key upload, or a device error
// in the middle of a resident proof), [2] evictions
void bbg_shim_resident_stats(uint64_t out[3])
{
    std::lock_guard<std::mutex> lk(cache().mu);
    out[0] = cache().proofs;
    out[1] = cache().fallbacks;
    out[2] = cache().evictions;
}
// keys uploaded again because the host polynomials of a cached proving key had changed (key_fingerprint)
uint64_t bbg_shim_resident_reuploads(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    return cache().reuploads;
}
}

// ---- the four wrapped symbols.  Itanium ABI: a non-static member function takes `this` as its first argument and a reference comes
// back as a pointer, so a free function of this shape IS the member function as far as the linker and the callers are concerned.
namespace waffle {
#define BBG_WRAP_CONSTRUCT_PROOF(settings, mangled)                                                                    \
    plonk_proof& bbg_real_construct_proof_##settings(ProverBase<settings>* self) asm("__real_" mangled);              \
    plonk_proof& bbg_wrap_construct_proof_##settings(ProverBase<settings>* self) asm("__wrap_" mangled);              \
    plonk_proof& bbg_wrap_construct_proof_##settings(ProverBase<settings>* self)                                      \
    {                                                                                                                  \
        return resident_or_real<settings>(self, &bbg_real_construct_proof_##settings);                                \
    }
BBG_WRAP_CONSTRUCT_PROOF(turbo_settings, "_ZN6waffle10ProverBaseINS_14turbo_settingsEE15construct_proofEv")
BBG_WRAP_CONSTRUCT_PROOF(standard_settings, "_ZN6waffle10ProverBaseINS_17standard_settingsEE15construct_proofEv")
BBG_WRAP_CONSTRUCT_PROOF(unrolled_turbo_settings, "_ZN6waffle10ProverBaseINS_23unrolled_turbo_settingsEE15construct_proofEv")
BBG_WRAP_CONSTRUCT_PROOF(unrolled_standard_settings, "_ZN6waffle10ProverBaseINS_26unrolled_standard_settingsEE15construct_proofEv")
#undef BBG_WRAP_CONSTRUCT_PROOF
} // namespace waffle
