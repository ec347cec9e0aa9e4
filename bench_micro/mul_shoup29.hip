// bench_micro/mul_shoup29.hip -- round 5: a product by a CONSTANT on 9 x 29-bit limbs without Montgomery's m digits (Shoup / Barrett with a
// precomputed quotient multiplier), against f29_mul.  Every product of an NTT step has a table twiddle as one operand.
//   w < p, wq = floor(w 2^261 / p) (both exact 29-bit limbs, precomputed).  For a lazily reduced x (carried limbs, x < V p):
//     q^ = floor(sum_{k >= 7} col_k(x, wq) 2^(29k) / 2^261)            53 mads   (columns 0..6 dropped: q^ in {q - 1, q})
//     r  = low 261 bits of x w + q^ (2^261 - p)                        45 + 45 mads, nine exact limbs, r = x w - q^ p in [0, (2 + V / 169) p)
//   143 mads, no v_mul_lo, no m-digit dependency between columns; Montgomery: 162 mads + 9 (mul_lo + and).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mul_shoup29.hip -o mul_shoup29
#include "../aztec-2.0_amd/csrc/field29.hip.h"
#include <cstdio>
#include <vector>
using namespace bbg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2048;

// limb J of 2^261 - p
template <class P> constexpr uint32_t pbar_limb(int j)
{
    uint32_t borrow = 0, out = 0;
    for (int i = 0; i <= j; i++) {
        const uint32_t pl = k29_limb(P::MOD, i);
        const int64_t d = (int64_t)0 - pl - borrow; // 2^261 has zero limbs below limb 9
        out = (uint32_t)(d & M29);
        borrow = d < 0 ? 1 : 0;
    }
    return out;
}
template <class P, int J> struct PB29 {
    static constexpr uint32_t value = pbar_limb<P>(J);
};
template <class P, int N, int J> __device__ __forceinline__ void mad_col_pbar(uint64_t& acc, const uint32_t* x)
{
#define BBG_PL(I) PB29<P, (J - (I) >= 0 && J - (I) <= 8) ? J - (I) : 0>::value
    if constexpr (N == 1) mad1_s(acc, x[0], BBG_PL(0));
    else if constexpr (N == 2) mad2_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1));
    else if constexpr (N == 3) mad3_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2));
    else if constexpr (N == 4) mad4_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3));
    else if constexpr (N == 5) mad5_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4));
    else if constexpr (N == 6) mad6_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5));
    else if constexpr (N == 7) mad7_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6));
    else if constexpr (N == 8) mad8_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6), x[7], BBG_PL(7));
    else if constexpr (N == 9) mad9_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6), x[7], BBG_PL(7), x[8], BBG_PL(8));
#undef BBG_PL
}
__device__ __forceinline__ void shr29(uint64_t& acc) { asm("v_lshrrev_b64 %0, 29, %0" : "+v"(acc)); }

template <class P> __device__ __forceinline__ F29<P> f29_mul_shoup(const F29<P>& x, const uint32_t* w, const uint32_t* wq)
{
    uint64_t acc = 0;
    uint32_t q[9];
    f29_ab_terms<7>(acc, x.v, wq); shr29(acc);
    f29_ab_terms<8>(acc, x.v, wq); shr29(acc);
#define BBG_Q(K) f29_ab_terms<K>(acc, x.v, wq); q[K - 9] = (uint32_t)acc & M29; shr29(acc);
    BBG_Q(9) BBG_Q(10) BBG_Q(11) BBG_Q(12) BBG_Q(13) BBG_Q(14) BBG_Q(15) BBG_Q(16)
#undef BBG_Q
    q[8] = (uint32_t)acc;
    F29<P> r;
    acc = 0;
#define BBG_R(K) f29_ab_terms<K>(acc, x.v, w); mad_col_pbar<P, K + 1, K>(acc, q); r.v[K] = (uint32_t)acc & M29; if (K < 8) shr29(acc);
    BBG_R(0) BBG_R(1) BBG_R(2) BBG_R(3) BBG_R(4) BBG_R(5) BBG_R(6) BBG_R(7) BBG_R(8)
#undef BBG_R
    return r;
}

__global__ void __launch_bounds__(256) k_check(const uint32_t* in, const uint32_t* wtab, int* bad, uint32_t* maxtop, int n)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const Fr x = fe_load<FrP>(in + (size_t)tid * 8);
    const uint32_t* row = wtab + (size_t)(tid & 255) * 32; // [0,9) w limbs, [9,18) wq limbs, [18,26) w in R-form words
    uint32_t w[9], wq[9];
    for (int i = 0; i < 9; i++) { w[i] = row[i]; wq[i] = row[9 + i]; }
    Fr wR;
    for (int i = 0; i < 8; i++) wR.v[i] = row[18 + i];
    // lazily reduced operand: x + x + x (V < 6), carried
    F29<FrP> X = f29_from_fe<FrP, 0>(x);
    X = f29_carry(f29_add(f29_add(X, X), X));
    const F29<FrP> r = f29_mul_shoup(X, w, wq);
    const Fr want = fe_canon(fe_mul(fe_add(fe_add(x, x), x), wR)); // (3x) * w
    const Fr got = fe_canon(f29_to_fe(r));
    if (!fe_eq(got, want)) atomicOr(bad, 1);
    for (int i = 0; i < 8; i++) if (r.v[i] > M29) atomicOr(bad, 2);
    atomicMax(maxtop, r.v[8]);
}

template <int V> __global__ void __launch_bounds__(256) mul_kernel(uint32_t* out, const uint32_t* in, const uint32_t* wtab)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = fe_load<FrP>(in + (size_t)tid * 8);
    const uint32_t* row = wtab + (size_t)(tid & 255) * 32;
    uint32_t w[9], wq[9];
    for (int i = 0; i < 9; i++) { w[i] = row[i]; wq[i] = row[9 + i]; }
    F29<FrP> X = f29_from_fe<FrP, 0>(x), Y = X, W;
    for (int i = 0; i < 9; i++) W.v[i] = w[i];
    for (int it = 0; it < ITERS / 4; it++) {
        if (V == 0) { X = f29_mul(X, W); Y = f29_mul(Y, W); }
        if (V == 1) { X = f29_mul_shoup(X, w, wq); Y = f29_mul_shoup(Y, w, wq); }
    }
    fe_store<FrP>(out + (size_t)tid * 8, f29_to_fe(f29_carry(f29_add(X, Y))));
}

template <class F> double time_it(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

// host big integers: 5 x 64-bit little-endian
struct U320 { uint64_t v[5]; };
static bool ge(const U320& a, const U320& b) { for (int i = 4; i >= 0; i--) { if (a.v[i] != b.v[i]) return a.v[i] > b.v[i]; } return true; }
static void sub(U320& a, const U320& b) { unsigned __int128 br = 0; for (int i = 0; i < 5; i++) { unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - br; a.v[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void shl1(U320& a) { for (int i = 4; i > 0; i--) a.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 63); a.v[0] <<= 1; }
static uint32_t limb29(const U320& a, int j) { const int b = 29 * j, i = b >> 6, o = b & 63; unsigned __int128 x = a.v[i]; if (i + 1 < 5) x |= (unsigned __int128)a.v[i + 1] << 64; return (uint32_t)((x >> o) & M29); }

int main()
{
    const int blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    U320 p{};
    for (int i = 0; i < 4; i++) p.v[i] = (uint64_t)FrP::MOD[2 * i] | ((uint64_t)FrP::MOD[2 * i + 1] << 32);
    std::vector<uint32_t> wt(256 * 32, 0);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int t = 0; t < 256; t++) {
        U320 w{};
        for (int i = 0; i < 4; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w.v[i] = s; }
        w.v[3] &= 0x0fffffffffffffffull; // < 2^252 < p
        if (t == 0) { w = U320{}; }                 // 0
        if (t == 1) { w = p; U320 one{}; one.v[0] = 1; sub(w, one); } // p - 1
        if (t == 2) { w = U320{}; w.v[0] = 1; }     // 1
        // wq = floor(w 2^261 / p), and w R mod p with R = 2^256 (bitwise long division)
        U320 rem = w, q{}, remR{};
        for (int b = 0; b < 261; b++) {
            shl1(rem); shl1(q);
            if (ge(rem, p)) { sub(rem, p); q.v[0] |= 1; }
            if (b == 255) remR = rem;
        }
        for (int j = 0; j < 9; j++) { wt[t * 32 + j] = limb29(w, j); wt[t * 32 + 9 + j] = limb29(q, j); }
        for (int i = 0; i < 8; i++) wt[t * 32 + 18 + i] = (uint32_t)(remR.v[i / 2] >> (32 * (i & 1)));
    }
    uint32_t *out, *in, *dw, *d_top;
    int* d_bad;
    CK(hipMalloc(&out, n * 32)); CK(hipMalloc(&in, n * 32)); CK(hipMalloc(&d_bad, 4)); CK(hipMalloc(&d_top, 4)); CK(hipMalloc(&dw, wt.size() * 4));
    std::vector<uint32_t> h(n * 8);
    s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)s; }
    for (size_t i = 0; i < n; i++) h[i * 8 + 7] &= 0x3fffffffu; // < 2^254 < 2p
    for (int k = 0; k < 8; k++) { h[k] = 0; h[8 + k] = FrP::MOD[k]; h[16 + k] = k ? FrP::MOD[k] : FrP::MOD[0] - 1; h[24 + k] = k ? 0 : 1; }
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
    int bad = 0; uint32_t top = 0;
    CK(hipMemcpy(d_bad, &bad, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_top, &top, 4, hipMemcpyHostToDevice));
    k_check<<<blocks, threads>>>(in, dw, d_bad, d_top, (int)n);
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&top, d_top, 4, hipMemcpyDeviceToHost));
    printf("f29_mul_shoup check over %zu operands x 256 constants: failure mask 0x%x (%s); largest top limb 0x%x (p's top limb 0x%x: value < %.2f p)\n", n, bad,
           bad ? "FAIL" : "PASS", top, k29_limb(FrP::MOD, 8), (double)(top + 1) / k29_limb(FrP::MOD, 8));
    const char* names[] = { "f29_mul (Montgomery, 162 mads)", "f29_mul_shoup (constant operand, 143 mads)" };
#define MUL(V) { double t = time_it([&] { mul_kernel<V><<<blocks, threads>>>(out, in, dw); }); double cnt = (double)n * (ITERS / 4) * 2; \
        printf("%-46s %8.3f ms  %8.2f Gop/s\n", names[V], t * 1e3, cnt / t / 1e9); }
    MUL(0) MUL(1) MUL(0) MUL(1)
    return bad ? 1 : 0;
}
