// One radix-8 NTT step (butterfly + seven step twiddles) on 8 register-resident elements: 8 x 32-bit limbs (what k_ntt_pass8 runs:
// p8_butterfly<3> + fe_mul) against lazily reduced 9 x 29-bit limbs (ntt29.hip.h: n29_step8).  Checks that both give the same residues and
// times ITERS dependent steps per thread.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 ntt_step29.hip -o ntt_step29
#include "../aztec-2.0_amd/csrc/ntt29.hip.h"
namespace bbg {
struct PassParams;
}
#include <cstdio>
#include <vector>
using namespace bbg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 256;

__device__ __forceinline__ void b32(Fr& a, Fr& b) { const Fr u = fe_add(a, b); b = fe_sub(a, b); a = u; }
__device__ __forceinline__ void b32w(Fr& a, Fr& b, const Fr& w) { const Fr u = fe_add(a, b); b = fe_mul(fe_sub(a, b), w); a = u; }
__device__ __forceinline__ void step32(Fr (&x)[8], const Fr& w1, const Fr& w2, const Fr& w3, const Fr (&tw)[8])
{
    b32(x[0], x[4]); b32w(x[1], x[5], w1); b32w(x[2], x[6], w2); b32w(x[3], x[7], w3);
    b32(x[0], x[2]); b32w(x[1], x[3], w2); b32(x[4], x[6]); b32w(x[5], x[7], w2);
    b32(x[0], x[1]); b32(x[2], x[3]); b32(x[4], x[5]); b32(x[6], x[7]);
#pragma unroll
    for (int j = 1; j < 8; j++) x[j] = fe_mul(x[j], tw[j]);
}
// w R -> w R' mod p = w R * 32: Montgomery product with 32 R
__device__ __forceinline__ Fr to_rprime(const Fr& w)
{
    Fr c = Fr::zero();
    c.v[0] = 32;
    return fe_reduce_once(fe_reduce_once(fe_mul(w, fe_to_mont(c))));
}

template <int V> __global__ void __launch_bounds__(256) k_step(uint32_t* out, const uint32_t* in, int* bad)
{
    __shared__ __attribute__((aligned(16))) uint32_t red[NTT29_TABLE_WORDS];
    if (threadIdx.x < NTT29_RED_ROWS) ntt29_fill_reduce_table(red, threadIdx.x);
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x[8], tw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        x[j] = fe_load<FrP>(in + ((size_t)tid * 8 + j) * 8);
        tw[j] = fe_reduce_once(fe_load<FrP>(in + ((size_t)(tid ^ 1) * 8 + j) * 8)); // "twiddles": arbitrary field elements, < p
    }
    const Fr w1 = tw[1], w2 = tw[2], w3 = tw[3];
    if (V == 0) {
        for (int it = 0; it < ITERS; it++) step32(x, w1, w2, w3, tw);
#pragma unroll
        for (int j = 0; j < 8; j++) fe_store<FrP>(out + ((size_t)tid * 8 + j) * 8, fe_canon(x[j]));
    } else {
        Fr29 y[8], t29[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            y[j] = f29_from_fe<FrP, 0>(x[j]);             // the 256 bits re-limbed: the value x R = (x / 32) R'
            t29[j] = f29_from_fe<FrP, 0>(to_rprime(tw[j])); // w R' mod p, exact limbs
        }
        const Fr29 W1 = t29[1], W2 = t29[2], W3 = t29[3];
        for (int it = 0; it < ITERS; it++) n29_step8<0, true>(y, W1, W2, W3, [&](int j) { return t29[j]; }, red);
        int fails = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const Fr back = fe_canon(f29_to_fe(y[j]));
            const Fr want = fe_load<FrP>(out + ((size_t)tid * 8 + j) * 8); // the 32-bit kernel ran first
            if (!fe_eq(back, want)) fails |= 1 << j;
        }
        if (fails) atomicOr(bad, fails);
        if (tid == 0xffffffffu) fe_store<FrP>(out, f29_to_fe(y[0]));
    }
}

int main()
{
    const int blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads * 8;
    uint32_t *out, *in;
    int* d_bad;
    CK(hipMalloc(&out, n * 32)); CK(hipMalloc(&in, n * 32)); CK(hipMalloc(&d_bad, 4));
    std::vector<uint32_t> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s; }
    for (size_t i = 0; i < n; i++) h[i * 8 + 7] &= 0x3fffffffu; // < 2^254 < 2p
    for (int k = 0; k < 8; k++) { h[k] = 0; h[8 + k] = FrP::MOD[k]; h[16 + k] = k ? FrP::MOD[k] : FrP::MOD[0] - 1; h[24 + k] = k ? 0 : 1; } // 0, p, p-1, 1
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));
    int bad = 0;
    CK(hipMemcpy(d_bad, &bad, 4, hipMemcpyHostToDevice));
    auto time_it = [&](auto launch) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best;
    };
    const float t32 = time_it([&] { k_step<0><<<blocks, threads>>>(out, in, d_bad); });
    const float t29 = time_it([&] { k_step<1><<<blocks, threads>>>(out, in, d_bad); });
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    const double steps = (double)blocks * threads * ITERS;
    printf("radix-8 step x %d on %d threads: identical residues: %s (mask 0x%x)\n", ITERS, blocks * threads, bad ? "NO" : "yes", bad);
    printf("  8 x 32-bit limbs (k_ntt_pass8's arithmetic) %8.3f ms  %7.2f G steps/s  (%.1f G products/s)\n", t32, steps / t32 / 1e6, steps * 12 / t32 / 1e6);
    printf("  9 x 29-bit limbs, lazy (ntt29.hip.h)         %8.3f ms  %7.2f G steps/s  (%.1f G products/s)   ratio %.3f\n", t29, steps / t29 / 1e6,
           steps * 12 / t29 / 1e6, t29 / t32);
    return bad ? 1 : 0;
}
