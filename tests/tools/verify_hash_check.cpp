// Host-only check of shim/bbg_shim_verify.hpp (the content hash behind the shim's cache verification): built and run by
// tests/test_abi_cpu.py::test_shim_content_hash_host_logic with g++ -- no GPU, no reference headers.
#include "../../shim/bbg_shim_verify.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bbg_shim_verify;

static int fails = 0;
static void expect(bool ok, const char* what)
{
    std::printf("%s  %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) fails++;
}

int main()
{
    // three "polynomials" of different lengths (one shorter than a piece, one spanning several, one empty)
    std::vector<uint64_t> a(4 * 1000), b(4 * 70001), c;
    uint64_t s = 0x243F6A8885A308D3ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& w : a) w = next();
    for (auto& w : b) w = next();
    std::vector<Span> spans = { { a.data(), a.size() }, { b.data(), b.size() }, { c.data(), 0 } };
    const uint64_t h0 = hash_spans(spans);
    std::printf("digest %016llx\n", (unsigned long long)h0); // the test compares it across thread counts
    expect(hash_spans(spans) == h0, "the digest is a function of the contents (two runs, pieces taken by whichever thread is free)");
    // every single-bit change of a sampled set of positions changes the digest -- first word, last word, a word in the middle of a piece, piece borders
    const size_t probes[] = { 0, 1, 32767, 32768, 32769, 65535, 65536, b.size() / 2, b.size() - 2, b.size() - 1 };
    bool all = true;
    for (size_t pidx : probes)
        for (int bit : { 0, 17, 63 }) {
            b[pidx] ^= 1ull << bit;
            all = all && hash_spans(spans) != h0;
            b[pidx] ^= 1ull << bit;
        }
    expect(all, "one flipped bit anywhere (piece borders included) changes the digest");
    expect(hash_spans(spans) == h0, "... and flipping it back restores it");
    a[999 * 4 + 3] += 1;
    expect(hash_spans(spans) != h0, "the last coefficient of a short polynomial counts");
    a[999 * 4 + 3] -= 1;
    // the same words moved between two polynomials, or two polynomials swapped, are different keys
    std::vector<Span> swapped = { { b.data(), b.size() }, { a.data(), a.size() }, { c.data(), 0 } };
    expect(hash_spans(swapped) != h0, "polynomial order is part of the digest");
    std::vector<Span> shifted = { { a.data(), a.size() - 4 }, { b.data(), b.size() }, { c.data(), 0 } };
    expect(hash_spans(shifted) != h0, "a polynomial's length is part of the digest");
    // per-point hashes (the table cache): position-seeded, so two equal points at different indices hash differently
    uint64_t pt[8] = { 1, 2, 3, 4, 5, 6, 7, 8 };
    expect(hash_words(pt, 8, 5) != hash_words(pt, 8, 6), "a point's hash depends on its index");
    uint64_t pt2[8] = { 1, 2, 3, 4, 5, 6, 7, 9 };
    expect(hash_words(pt, 8, 5) != hash_words(pt2, 8, 5), "... and on every word");
    // parallel_chunks visits every chunk exactly once whatever the thread count
    std::vector<std::atomic<int>> seen(1000);
    for (auto& v : seen) v = 0;
    parallel_chunks(seen.size(), [&](size_t k) { seen[k]++; });
    bool once = true;
    for (auto& v : seen) once = once && v == 1;
    expect(once, "parallel_chunks: every chunk exactly once");
    std::printf(fails ? "verify_hash_check FAILED (%d)\n" : "verify_hash_check PASS (threads %u)\n", fails ? fails : threads());
    return fails ? 1 : 0;
}
