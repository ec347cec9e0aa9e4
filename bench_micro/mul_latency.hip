// LATENCY of one Montgomery product on gfx950 when a wave is ALONE on its SIMD -- the regime of the MSM reduce chains (k_combine*, k_rowcol_q,
// k_final_planes_q, k_final_sum_q: a few waves running a chain of dependent EC operations; a third of a small MSM, msm_kernels.hip.h).
// bench_micro/mul29.hip and mulbench.hip measure THROUGHPUT (many waves per SIMD); here every product waits for the one before it.
//   fe_mul        8 x 32-bit limbs, one mad + one addc per limb product, columns strictly one after the other (field.hip.h)
//   fe_mul29      the same product through the 29-bit multiplier, columns as asm chains (field29.hip.h)
//   fe_mul29_ilp  29-bit limbs with every column in an accumulator of its own: the 81 a*b products do not wait for each other, the reduction's
//                 nine digit steps each add nine independent m*p products (field29.hip.h, round 6)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mul_latency.hip -o mul_latency
#include "../aztec-2.0_amd/csrc/field29.hip.h"
#include <cstdio>
#include <vector>
using namespace bbg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 4096;

template <int V> __device__ __forceinline__ Fq mul_variant(const Fq& a, const Fq& b)
{
    if constexpr (V == 0) return fe_mul(a, b);
    else if constexpr (V == 1) return fe_mul29(a, b);
    else return fe_mul29_ilp(a, b);
}
template <int V> __global__ void __launch_bounds__(64) k_chain(const uint32_t* in, uint32_t* out)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = fe_load<FqP>(in + (size_t)(tid & 1023) * 8);
    const Fq y = fe_load<FqP>(in + (size_t)((tid + 7) & 1023) * 8);
    for (int i = 0; i < ITERS; i++) x = mul_variant<V>(x, y);
    fe_store<FqP>(out + (size_t)tid * 8, x);
}
__global__ void __launch_bounds__(256) k_check(const uint32_t* in, int* bad, int n)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    const Fq x = fe_load<FqP>(in + (size_t)tid * 8), y = fe_load<FqP>(in + (size_t)(tid ^ 1) * 8);
    const Fq z = fe_canon(fe_mul(x, y));
    int fails = 0;
    if (!fe_eq(fe_canon(fe_mul29(x, y)), z)) fails |= 1;
    if (!fe_eq(fe_canon(fe_mul29_ilp(x, y)), z)) fails |= 2;
    // coarse operands (the callers pass values < 2p, sometimes < 4p with the other < p)
    const Fq x2 = fe_add(x, x);
    if (!fe_eq(fe_canon(fe_mul29_ilp(x2, y)), fe_canon(fe_mul(x2, y)))) fails |= 4;
    if (fails) atomicOr(bad, fails);
}

template <int V> static double run(const char* name, const uint32_t* d_in, uint32_t* d_out, int blocks)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(64), 0, 0, d_in, d_out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double ns = best * 1e6 / ITERS;
    printf("%-14s %5d waves (%s)   %8.1f ns per dependent product\n", name, blocks, blocks <= 1024 ? "<= 1 per SIMD" : "several per SIMD", ns);
    return ns;
}

int main()
{
    const int n = 4096;
    std::vector<uint32_t> h((size_t)n * 8);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& w : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        w = (uint32_t)(s >> 16);
    }
    for (int i = 0; i < n; i++) h[(size_t)i * 8 + 7] &= 0x1fffffffu; // < 2^253 < p
    uint32_t *d_in, *d_out;
    int* d_bad;
    CK(hipMalloc(&d_in, h.size() * 4));
    CK(hipMalloc(&d_out, (size_t)8192 * 64 * 32));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(n / 256), dim3(256), 0, 0, d_in, d_bad, n);
    int bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("products agree with fe_mul on %d random pairs: %s (mask %d)\n", n, bad ? "NO" : "yes", bad);
    for (int blocks : { 256, 1024, 4096 }) {
        const double a = run<0>("fe_mul", d_in, d_out, blocks);
        const double b = run<1>("fe_mul29", d_in, d_out, blocks);
        const double c = run<2>("fe_mul29_ilp", d_in, d_out, blocks);
        printf("    -> fe_mul29 %.2fx, fe_mul29_ilp %.2fx the speed of fe_mul\n", a / b, a / c);
    }
    return bad ? 1 : 0;
}
