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
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "bbg_resident_prover.hpp"
#include "bbg_shim_verify.hpp"

namespace {
using bbg_shim::ResidentKey;

struct ResidentCache {
    std::mutex mu;
    struct Entry {
        std::shared_ptr<ResidentKey> rk; // shared with a round-by-round proof in progress: an eviction cannot pull the key from under it
        size_t bytes = 0;
        uint64_t last_use = 0;
        uint64_t fingerprint = 0; // of the host key polynomials the device copy was made from (key_fingerprint below): the quick look
        uint64_t content = 0;     // full mode: hash of EVERY coefficient that was uploaded (key_content_hash below)
    };
    std::map<const waffle::proving_key*, Entry> entries;
    // proofs a host drives round by round (execute_preamble_round ... execute_sixth_round), keyed by the prover object
    struct ProofIface {
        virtual ~ProofIface() {}
        virtual void step(int k) = 0;
    };
    struct Progress {
        std::shared_ptr<ResidentKey> rk;
        std::unique_ptr<ProofIface> proof; // null: this prover's current proof runs on the reference rounds
        int next = 0;                      // the round expected next (resident mode)
        const waffle::proving_key* id = nullptr;
        uint64_t touched = 0;              // cache clock of the last round: abandoned proofs age out (sweep_progress)
        uint64_t content = 0;              // the key's recorded content hash when the proof began
        std::future<uint64_t> check;       // full mode: the host key re-hashed while the rounds run; looked at before the last round
    };
    std::map<const void*, Progress> in_progress;
    static constexpr size_t MAX_IN_PROGRESS = 16;      // round-by-round proofs that may be open at once
    static constexpr uint64_t PROGRESS_MAX_AGE = 64;   // ... and for how many cache uses an untouched one is kept
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
    // device bytes the cache answers for: its entries, plus keys that are no longer entries but are still pinned by a proof in progress
    size_t total() const
    {
        size_t t = 0;
        for (const auto& kv : entries) t += kv.second.bytes;
        for (const auto& kv : in_progress) {
            const Progress& pg = kv.second;
            if (!pg.rk) continue;
            auto it = entries.find(pg.id);
            if (it != entries.end() && it->second.rk == pg.rk) continue; // counted above
            size_t bytes = 0;
            if (bbg_prover_device_bytes(pg.rk->handle(), &bytes) == BBG_OK) t += bytes;
        }
        return t;
    }
    // Round-by-round proofs nobody will finish (round-5 advisor finding): the prover was destroyed after an error, the proof abandoned.
    // Such an entry pins its key's device memory outside the LRU budget and keeps a reference to a dead prover.  Dropped here: proofs whose
    // proving key only the cache still holds (the prover that shared it is gone -- the rule sweep() applies to entries), proofs not touched
    // for PROGRESS_MAX_AGE cache uses, and the oldest ones beyond MAX_IN_PROGRESS.
    void sweep_progress()
    {
        for (auto it = in_progress.begin(); it != in_progress.end();) {
            const Progress& pg = it->second;
            const bool orphan = pg.rk && pg.rk->key().use_count() == 1; // only the device copy itself still holds the proving key
            const bool old = clock - pg.touched > PROGRESS_MAX_AGE;
            if (orphan || old) it = in_progress.erase(it);
            else ++it;
        }
        while (in_progress.size() > MAX_IN_PROGRESS) {
            auto oldest = in_progress.begin();
            for (auto it = in_progress.begin(); it != in_progress.end(); ++it)
                if (it->second.touched < oldest->second.touched) oldest = it;
            in_progress.erase(oldest);
        }
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

// Full mode (bbg_shim_verify.hpp): every coefficient ResidentKey uploads -- the first n coefficients of each selector and permutation
// polynomial of the manifest (bbg_resident_prover.hpp: bbg_prover_set_key_poly) -- in one 64-bit digest.  Recorded when the device copy is
// made; recomputed on host threads WHILE each later proof over the cached copy runs on the GPU and compared before the proof is handed
// back: one poked coefficient anywhere gives a re-upload and a repeated proof, never a proof over the stale copy.  The Lagrange / coset
// forms the reference keeps beside the coefficient forms are derived on the device from what was uploaded; a host that edits ONLY those
// has a key whose forms contradict each other (compute_proving_key's invariant, composer_base.cpp:180-262) -- not covered, not coverable.
uint64_t key_content_hash(const waffle::proving_key& key)
{
    std::vector<bbg_shim_verify::Span> spans;
    for (const auto& info : key.polynomial_manifest) {
        if (info.source == waffle::PolynomialSource::WITNESS) continue;
        const std::string label(info.polynomial_label);
        const auto& table = info.source == waffle::PolynomialSource::SELECTOR ? key.constraint_selectors : key.permutation_selectors;
        auto it = table.find(label);
        if (it == table.end() || it->second.get_size() == 0) {
            spans.push_back({ nullptr, 0 });
            continue;
        }
        spans.push_back({ reinterpret_cast<const uint64_t*>(&it->second[0]), std::min<size_t>(key.n, it->second.get_size()) * 4 });
    }
    return bbg_shim_verify::hash_spans(spans) ^ (uint64_t)key.n;
}

barretenberg::fr draw_adapter(void*)
{
    ResidentCache& c = cache();
    barretenberg::fr v;
    c.draw(c.draw_user, reinterpret_cast<uint64_t*>(&v));
    return v;
}
// The blinding scalars of a proof, recorded while it is made so that a proof that has to be REPEATED (the full check found the cached key
// stale) draws the same ones: the repeated proof is the proof the first attempt would have been over the key as it is now.
struct BlindingTape {
    std::vector<barretenberg::fr> drawn;
    size_t next = 0;
    bool replay = false;
    static barretenberg::fr draw(void* user)
    {
        BlindingTape& t = *static_cast<BlindingTape*>(user);
        if (t.replay && t.next < t.drawn.size()) return t.drawn[t.next++];
        ResidentCache& c = cache();
        const barretenberg::fr v = c.draw ? draw_adapter(nullptr) : bbg_shim::os_random_fr();
        t.drawn.push_back(v);
        return v;
    }
};

// The device copy of self's proving key: cached, re-uploaded when the host polynomials changed (key_fingerprint), created (evicting every
// other key once if the first attempt fails: most likely device memory) -- or null, counted as a fallback, when it cannot be had.
// Called with the cache locked.
template <typename settings>
ResidentCache::Entry* acquire_key(ResidentCache& c, waffle::ProverBase<settings>* self, const char* what, bool* fresh = nullptr)
{
    if (fresh) *fresh = false;
    c.sweep();
    c.sweep_progress();
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
        std::shared_ptr<ResidentKey> rk;
        for (int attempt = 0; attempt < 2 && !rk; attempt++) {
            try {
                rk = std::make_shared<ResidentKey>(self->key, settings::program_width);
            } catch (const std::exception& e) {
                if (attempt == 0 && !c.entries.empty()) { // most likely device memory: give back every other key and try once more
                    c.evictions += c.entries.size();
                    c.entries.clear();
                    continue;
                }
                // the reference's body still proves (its MSMs / FFTs are wrapped onto the GPU): slower, never wrong
                if (c.fallbacks++ == 0) std::fprintf(stderr, "bbg_shim: resident key not created (%s); %s takes the reference body\n", e.what(), what);
                return nullptr;
            }
        }
        ResidentCache::Entry e;
        size_t bytes = 0;
        if (bbg_prover_device_bytes(rk->handle(), &bytes) != BBG_OK) bytes = 0;
        e.bytes = bytes;
        e.fingerprint = fingerprint;
        if (bbg_shim_verify::full_mode()) e.content = key_content_hash(*self->key);
        e.rk = std::move(rk);
        it = c.entries.emplace(id, std::move(e)).first;
        if (fresh) *fresh = true;
    }
    it->second.last_use = ++c.clock;
    c.evict_to_budget(id);
    return &it->second;
}

// set while a reference body runs on this thread on behalf of a wrapped entry point: should a reference body reach another wrapped symbol
// (it does not when prover.cpp is one translation unit -- its internal calls are not undefined references --, but a build may split it), the
// call goes straight to the reference
thread_local int t_in_reference = 0;
struct ReferenceScope {
    ReferenceScope() { t_in_reference++; }
    ~ReferenceScope() { t_in_reference--; }
};

template <typename settings>
waffle::plonk_proof& resident_or_real(waffle::ProverBase<settings>* self, waffle::plonk_proof& (*real)(waffle::ProverBase<settings>*))
{
    ResidentCache& c = cache();
    if (t_in_reference) return real(self);
    std::unique_lock<std::mutex> lk(c.mu);
    c.in_progress.erase(self); // whatever a host began round by round on this prover is superseded
    if (!c.enabled || !self->key || !bbg_shim::resident_supported(*self)) {
        lk.unlock();
        ReferenceScope rs;
        return real(self);
    }
    bool fresh = false;
    ResidentCache::Entry* entry = acquire_key(c, self, "construct_proof()", &fresh);
    if (!entry) {
        lk.unlock();
        ReferenceScope rs;
        return real(self);
    }
    const waffle::proving_key* id = self->key.get();
    std::shared_ptr<ResidentKey> rk = entry->rk;
    BlindingTape tape;
    bbg_shim::ResidentOptions opt;
    opt.random = &BlindingTape::draw;
    opt.user = &tape;
    c.proofs++;
    // the lock is held for the whole proof: the reference's prover is not re-entrant either (process-global FFT scratch,
    // polynomial_arithmetic.cpp:13-34), and the shim's device context is one stream
    try {
        // a cached device copy: the host key is re-hashed on other host threads while the GPU makes the proof (this thread mostly waits for
        // it), and the proof is handed back only if every uploaded coefficient is still what it was
        std::future<uint64_t> check;
        if (!fresh && bbg_shim_verify::full_mode()) check = std::async(std::launch::async, [key = self->key]() { return key_content_hash(*key); });
        const uint64_t recorded = entry->content;
        waffle::plonk_proof& proof = bbg_shim::construct_proof(*self, *rk, opt);
        if (!check.valid() || check.get() == recorded) return proof;
        // the host polynomials changed under the cached copy: upload the key as it is now and make the proof again, with the same blinding
        c.entries.erase(id);
        c.reuploads++;
        rk.reset();
        entry = acquire_key(c, self, "construct_proof()", &fresh);
        if (!entry) {
            lk.unlock();
            self->reset();
            ReferenceScope rs;
            return real(self);
        }
        rk = entry->rk;
        self->reset();
        tape.replay = true;
        tape.next = 0;
        return bbg_shim::construct_proof(*self, *rk, opt);
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
        ReferenceScope rs;
        return real(self);
    }
}

// ---- round by round.  A host that drives the prover through execute_preamble_round() ... execute_sixth_round() with process_queue()
// between them (the reference's C binding, plonk/proof_system/prover/c_bind.cpp:59-92; construct_proof() itself is exactly that sequence,
// prover.cpp:420-436) gets the same device rounds one at a time.  The preamble decides: a prover the device rounds implement, called in
// order, runs resident (each wrapped round leaves its commitments in the transcript and the work queue empty, so the process_queue() that
// follows finds nothing to do); anything else -- unsupported widgets, the wrap switched off, a round called out of order, a round for a
// proof that did not begin with the preamble -- runs the reference round.  A device error inside round r rebuilds the transcript
// (ProverBase::reset) and replays the reference rounds 0 .. r with their process_queue() calls, so the host continues from the same place.
template <typename settings> struct RoundTable {
    using Fn = void (*)(waffle::ProverBase<settings>*);
    Fn real[7];
};
template <typename settings> struct ProofImpl : ResidentCache::ProofIface {
    bbg_shim::ResidentProof<settings> pr;
    ProofImpl(waffle::ProverBase<settings>& p, ResidentKey& rk, const bbg_shim::ResidentOptions& opt)
        : pr(p, rk, opt)
    {}
    void step(int k) override { pr.step(k); }
};

template <typename settings> void resident_round(waffle::ProverBase<settings>* self, int round, const RoundTable<settings>& table)
{
    ResidentCache& c = cache();
    if (t_in_reference) return table.real[round](self);
    std::unique_lock<std::mutex> lk(c.mu);
    auto reference = [&]() {
        lk.unlock();
        ReferenceScope rs;
        table.real[round](self);
    };
    if (round == 0) { // a new proof begins on this prover (first use, or after ProverBase::reset())
        c.in_progress.erase(self);
        if (!c.enabled || !self->key || !bbg_shim::resident_supported(*self)) return reference();
        bool fresh = false;
        ResidentCache::Entry* entry = acquire_key(c, self, "execute_preamble_round()", &fresh);
        if (!entry) {
            c.in_progress[self].touched = c.clock; // reference mode for the rest of this proof
            return reference();
        }
        ResidentCache::Progress& pg = c.in_progress[self];
        pg.rk = entry->rk;
        pg.id = self->key.get();
        pg.touched = c.clock;
        pg.content = entry->content;
        // a cached device copy: the host key is re-hashed while the rounds run and looked at before the last one (below)
        if (!fresh && bbg_shim_verify::full_mode()) pg.check = std::async(std::launch::async, [key = self->key]() { return key_content_hash(*key); });
        bbg_shim::ResidentOptions opt;
        if (c.draw) opt.random = &draw_adapter;
        pg.proof = std::make_unique<ProofImpl<settings>>(*self, *pg.rk, opt);
        pg.next = 0;
        c.proofs++;
    }
    auto it = c.in_progress.find(self);
    if (it == c.in_progress.end() || !it->second.proof) return reference(); // not begun here, or on the reference rounds already
    ResidentCache::Progress& pg = it->second;
    // out of order, or this prover object is not the one the proof began on (an abandoned proof whose prover was destroyed and another
    // allocated at the same address, over another key): the resident state is abandoned, the reference does whatever it does with such a call
    if (round != pg.next || self->key.get() != pg.id) {
        pg.proof.reset();
        pg.rk.reset();
        return reference();
    }
    pg.touched = ++c.clock;
    try {
        if (round == 6 && pg.check.valid() && pg.check.get() != pg.content) {
            c.reuploads++;
            throw std::runtime_error("the proving key's host polynomials changed under its cached device copy");
        }
        pg.proof->step(round);
        pg.next = round + 1;
        if (round == 6) c.in_progress.erase(it); // the proof is in the transcript: export_proof() is the reference's
        return;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "bbg_shim: resident round %d failed (%s); the proof is repeated on the reference rounds\n", round, e.what());
        c.entries.erase(pg.id);
        pg.proof.reset();
        pg.rk.reset(); // reference mode from here on
        c.fallbacks++;
    }
    lk.unlock();
    ReferenceScope rs;
    self->reset();
    for (int k = 0; k <= round; k++) {
        table.real[k](self);
        if (k < round) self->queue.process_queue(); // the host's own process_queue() follows round `round`
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
    cache().in_progress.clear();
    cache().entries.clear();
}
// proofs in progress round by round (begun with execute_preamble_round(), not yet through execute_sixth_round())
size_t bbg_shim_resident_in_progress(void)
{
    std::lock_guard<std::mutex> lk(cache().mu);
    size_t live = 0;
    for (const auto& kv : cache().in_progress) live += kv.second.proof ? 1 : 0;
    return live;
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

// the seven rounds, four instantiations each (shim/wrap_flags_prover.txt lists the same 28 names)
#define BBG_WRAP_ROUND(settings, tag, index, mangled)                                                                  \
    void bbg_real_##tag##_##settings(ProverBase<settings>* self) asm("__real_" mangled);                              \
    void bbg_wrap_##tag##_##settings(ProverBase<settings>* self) asm("__wrap_" mangled);
#define BBG_WRAP_ROUNDS(settings, len, mangled_settings)                                                               \
    BBG_WRAP_ROUND(settings, preamble, 0, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE22execute_preamble_roundEv")   \
    BBG_WRAP_ROUND(settings, first, 1, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE19execute_first_roundEv")         \
    BBG_WRAP_ROUND(settings, second, 2, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE20execute_second_roundEv")       \
    BBG_WRAP_ROUND(settings, third, 3, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE19execute_third_roundEv")         \
    BBG_WRAP_ROUND(settings, fourth, 4, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE20execute_fourth_roundEv")       \
    BBG_WRAP_ROUND(settings, fifth, 5, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE19execute_fifth_roundEv")         \
    BBG_WRAP_ROUND(settings, sixth, 6, "_ZN6waffle10ProverBaseINS_" #len mangled_settings "EE19execute_sixth_roundEv")         \
    static const RoundTable<settings> bbg_round_table_##settings = { { &bbg_real_preamble_##settings, &bbg_real_first_##settings,        \
                                                                       &bbg_real_second_##settings, &bbg_real_third_##settings,         \
                                                                       &bbg_real_fourth_##settings, &bbg_real_fifth_##settings,         \
                                                                       &bbg_real_sixth_##settings } };                                  \
    void bbg_wrap_preamble_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 0, bbg_round_table_##settings); }    \
    void bbg_wrap_first_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 1, bbg_round_table_##settings); }       \
    void bbg_wrap_second_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 2, bbg_round_table_##settings); }      \
    void bbg_wrap_third_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 3, bbg_round_table_##settings); }       \
    void bbg_wrap_fourth_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 4, bbg_round_table_##settings); }      \
    void bbg_wrap_fifth_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 5, bbg_round_table_##settings); }       \
    void bbg_wrap_sixth_##settings(ProverBase<settings>* self) { resident_round<settings>(self, 6, bbg_round_table_##settings); }
BBG_WRAP_ROUNDS(turbo_settings, 14, "turbo_settings")
BBG_WRAP_ROUNDS(standard_settings, 17, "standard_settings")
BBG_WRAP_ROUNDS(unrolled_turbo_settings, 23, "unrolled_turbo_settings")
BBG_WRAP_ROUNDS(unrolled_standard_settings, 26, "unrolled_standard_settings")
#undef BBG_WRAP_ROUNDS
#undef BBG_WRAP_ROUND
} // namespace waffle
