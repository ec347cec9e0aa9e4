// bbg_shim_verify.hpp -- full-content verification of the HOST memory a cached device copy was made from.
//
// The shim caches device copies of two kinds of host data it does not own: Pippenger point tables (bbg_barretenberg_shim.cpp) and the
// polynomials of a proving key (bbg_prover_wrap.cpp).  Both caches are keyed by the host address, and the reference has no hook that says
// "this memory changed" (pippenger.cpp:33-36 frees the table, proving_key.cpp:24 is a plain struct).  Rounds 4-5 validated an entry with
// SAMPLED contents -- a tripwire: a host that poked one unsampled point or coefficient got a result over the stale copy with no error.
// Round 6: every use of a cached entry re-hashes ALL the host bytes the device copy stands for and compares with what was recorded when
// the copy was made.  The hashing runs on host threads WHILE the GPU works on the call it guards (the calling thread would otherwise wait);
// the result is looked at before anything is returned, and a mismatch drops the entry, uploads again and repeats the call -- the caller
// sees a correct result or an exception, never a stale one.  BBG_SHIM_VERIFY=sample restores the sampled tripwire (for hosts that treat
// their tables and keys as immutable and want the host cores for themselves); BBG_SHIM_VERIFY_THREADS bounds the threads (default 8).
//
// The hash is a 64-bit non-cryptographic one (xxHash64's round structure, four lanes): it guards against accidents -- a host rewriting
// memory it has handed out --, not against an adversary who can choose the new contents.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace bbg_shim_verify {

inline bool full_mode()
{
    static const bool full = [] {
        const char* e = std::getenv("BBG_SHIM_VERIFY");
        return !(e && std::strcmp(e, "sample") == 0);
    }();
    return full;
}
inline unsigned threads()
{
    static const unsigned t = [] {
        unsigned want = 8;
        if (const char* e = std::getenv("BBG_SHIM_VERIFY_THREADS")) want = (unsigned)std::strtoul(e, nullptr, 10);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && want > hw) want = hw;
        return want ? want : 1u;
    }();
    return t;
}

constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t round1(uint64_t acc, uint64_t w) { return rotl(acc + w * P2, 31) * P1; }
inline uint64_t avalanche(uint64_t h)
{
    h ^= h >> 33;
    h *= P2;
    h ^= h >> 29;
    h *= P3;
    h ^= h >> 32;
    return h;
}
// `words` 64-bit words at w (8-byte aligned: field elements and points are), seeded
inline uint64_t hash_words(const uint64_t* w, size_t words, uint64_t seed)
{
    uint64_t a = seed + P1 + P2, b = seed + P2, c = seed, d = seed - P1;
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
        a = round1(a, w[i]);
        b = round1(b, w[i + 1]);
        c = round1(c, w[i + 2]);
        d = round1(d, w[i + 3]);
    }
    uint64_t h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18) + (uint64_t)words * 8;
    for (; i < words; i++) h = rotl(h ^ round1(0, w[i]), 27) * P1 + P4;
    return avalanche(h ^ P5);
}

// fn(k) for k in [0, count) on up to threads() host threads (the caller's included); fn must be thread-safe
template <typename F> void parallel_chunks(size_t count, F fn)
{
    const unsigned t = (unsigned)std::min<size_t>(threads(), count);
    if (t <= 1) {
        for (size_t k = 0; k < count; k++) fn(k);
        return;
    }
    std::atomic<size_t> next{ 0 };
    auto work = [&]() {
        for (size_t k = next.fetch_add(1); k < count; k = next.fetch_add(1)) fn(k);
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < t; i++) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

// One digest over a list of spans (a key's polynomials): spans are cut into 256-KiB pieces hashed in parallel; a piece's hash is mixed with
// (span, piece) and the mixes are summed, so the digest does not depend on which thread took which piece.
struct Span {
    const uint64_t* words;
    size_t count;
};
inline uint64_t hash_spans(const std::vector<Span>& spans)
{
    constexpr size_t PIECE = 32768; // words
    struct Piece {
        uint32_t span;
        uint32_t index;
    };
    std::vector<Piece> pieces;
    for (size_t s = 0; s < spans.size(); s++)
        for (size_t k = 0; k * PIECE < spans[s].count || k == 0; k++) pieces.push_back({ (uint32_t)s, (uint32_t)k });
    std::vector<uint64_t> out(pieces.size());
    parallel_chunks(pieces.size(), [&](size_t p) {
        const Span& sp = spans[pieces[p].span];
        const size_t lo = (size_t)pieces[p].index * PIECE, n = lo < sp.count ? std::min(PIECE, sp.count - lo) : 0;
        out[p] = avalanche(hash_words(sp.words + lo, n, ((uint64_t)pieces[p].span << 32) | pieces[p].index) + P3 * (p + 1));
    });
    uint64_t h = (uint64_t)spans.size() * P4;
    for (size_t s = 0; s < spans.size(); s++) h += avalanche(spans[s].count + P5 * (s + 1));
    for (uint64_t v : out) h += v;
    return avalanche(h);
}

} // namespace bbg_shim_verify
