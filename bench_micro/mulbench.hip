// Micro-benchmarks that size the integer-ALU roofline for the BN254 kernels on gfx950:
// raw issue rates of v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 / v_add / v_fma_f64 and
// the throughput of complete Montgomery multiplications (several formulations).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mulbench.hip -o mulbench
#include "../aztec-2.0_amd/csrc/field.hip.h"
#include <cstdio>
#include <vector>
using namespace bbg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;

// ---- raw instruction rates: 8 independent chains per lane
template <int OP> __global__ void __launch_bounds__(256) raw_kernel(uint32_t* out, uint32_t seed)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[8];
    uint32_t m = seed * 2654435761u + tid;
    double d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = ((uint64_t)(tid + k) << 32) | (seed + k); d[k] = (double)(tid + k); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (OP == 0) a[k] = (uint64_t)(uint32_t)a[k] * m + a[k];               // v_mad_u64_u32
            if (OP == 1) a[k] = (uint32_t)a[k] * m + 1;                            // v_mul_lo_u32 (+add)
            if (OP == 2) a[k] = __umulhi((uint32_t)a[k], m) ^ (uint32_t)a[k];     // v_mul_hi_u32 (+xor)
            if (OP == 3) a[k] = (uint32_t)a[k] + m + (uint32_t)(a[k] >> 3);        // adds
            if (OP == 4) d[k] = __fma_rn(d[k], 1.0000001, 0.5);                    // v_fma_f64
            if (OP == 5) a[k] = a[k] + (((uint64_t)m << 32) | m);                  // 64-bit add (add_co+addc)
            if (OP == 6) a[k] = __umul24((uint32_t)a[k], m) + (uint32_t)a[k]; // v_mad_u32_u24
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k] + (uint64_t)d[k];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <int V> __global__ void __launch_bounds__(256) mul_kernel(uint32_t* out, const uint32_t* in)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = fe_load<FrP>(in + (size_t)tid * 8);
    Fr y = fe_load<FrP>(in + (size_t)(tid ^ 1) * 8);
    for (int it = 0; it < ITERS / 4; it++) {
        if (V == 0) { x = fe_mul_cios(x, y); y = fe_mul_cios(y, x); }
        if (V == 1) { x = fe_mul(x, y); y = fe_mul(y, x); }
        if (V == 2) { x = fe_add(x, y); y = fe_sub(y, x); }
        if (V == 4) { x = fe_mul_v1(x, y); y = fe_mul_v1(y, x); }
        if (V == 3) { // butterfly-shaped: 1 mul + add + sub
            Fr t = fe_mul(x, y);
            Fr s = fe_add(x, t);
            y = fe_sub(y, t);
            x = s;
        }
    }
    fe_store<FrP>(out + (size_t)tid * 8, fe_add(x, y));
}

// occupancy probe: same dependent-mul kernel, but dynamic LDS limits resident blocks per CU (1 block = 1 wave per SIMD)
__global__ void __launch_bounds__(256) occ_kernel(uint32_t* out, const uint32_t* in)
{
    extern __shared__ uint32_t lds_dummy[];
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = fe_load<FrP>(in + (size_t)tid * 8);
    Fr y = fe_load<FrP>(in + (size_t)(tid ^ 1) * 8);
    if (in[0] == 0x12345678u) lds_dummy[threadIdx.x] = 1;
    for (int it = 0; it < ITERS / 4; it++) { x = fe_mul(x, y); y = fe_mul(y, x); }
    fe_store<FrP>(out + (size_t)tid * 8, fe_add(x, y));
}
// 4 independent chains per thread (ILP) at limited occupancy
__global__ void __launch_bounds__(256) occ_kernel_ilp4(uint32_t* out, const uint32_t* in)
{
    extern __shared__ uint32_t lds_dummy[];
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x[4], y[4];
    for (int k = 0; k < 4; k++) { x[k] = fe_load<FrP>(in + (size_t)((tid + k * 64) % (256 * 8 * 256)) * 8); y[k] = fe_load<FrP>(in + (size_t)(((tid ^ 1) + k * 64) % (256 * 8 * 256)) * 8); }
    if (in[0] == 0x12345678u) lds_dummy[threadIdx.x] = 1;
    for (int it = 0; it < ITERS / 16; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = fe_mul(x[k], y[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) y[k] = fe_mul(y[k], x[k]);
    }
    Fr s = fe_add(x[0], y[0]);
    for (int k = 1; k < 4; k++) s = fe_add(s, fe_add(x[k], y[k]));
    fe_store<FrP>(out + (size_t)tid * 8, s);
}

template <class F> double time_it(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e-3;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    uint32_t *out, *in;
    CK(hipMalloc(&out, n * 32)); CK(hipMalloc(&in, n * 32));
    std::vector<uint32_t> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s; }
    for (size_t i = 0; i < n; i++) h[i * 8 + 7] &= 0x0fffffffu;
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));

    const char* names[] = { "v_mad_u64_u32", "v_mul_lo_u32+add", "v_mul_hi_u32+xor", "add x2", "v_fma_f64", "add64", "mad_u32_u24" };
    double ops = (double)n * ITERS * 8;
#define RAW(OP) { double t = time_it([&] { raw_kernel<OP><<<blocks, threads>>>(out, 12345u); }); \
        printf("raw %-18s %8.3f ms  %8.2f Gop/s  (%.2f lanes/clk/CU @2.4GHz)\n", names[OP], t * 1e3, ops / t / 1e9, ops / t / 2.4e9 / 256); }
    RAW(0) RAW(1) RAW(2) RAW(3) RAW(4) RAW(5) RAW(6)

    const char* mnames[] = { "fe_mul CIOS (C++)", "fe_mul FIPS column-asm (shipping)", "add+sub", "butterfly mul+add+sub", "fe_mul FIPS v1 (asm per product)" };
    double per[] = { 2, 2, 2, 1, 2 };
#define MUL(V) { double t = time_it([&] { mul_kernel<V><<<blocks, threads>>>(out, in); }); \
        double cnt = (double)n * (ITERS / 4) * per[V]; \
        printf("%-28s %8.3f ms  %8.2f Gop/s\n", mnames[V], t * 1e3, cnt / t / 1e9); }
    MUL(0) MUL(1) MUL(2) MUL(3) MUL(4)
    CK(hipFuncSetAttribute((const void*)occ_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)occ_kernel_ilp4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int k : { 1, 2, 3, 4, 5, 8 }) {
        size_t lds = (160 * 1024) / k - 512;
        if (k == 8) lds = 16 * 1024;
        double t = time_it([&] { occ_kernel<<<blocks, threads, lds>>>(out, in); });
        double cnt = (double)n * (ITERS / 4) * 2;
        double t4 = time_it([&] { occ_kernel_ilp4<<<blocks, threads, lds>>>(out, in); });
        double cnt4 = (double)n * (ITERS / 16) * 8;
        printf("occupancy %d waves/SIMD: 1 chain/thread %8.2f Gmul/s   4 chains/thread %8.2f Gmul/s\n", k, cnt / t / 1e9, cnt4 / t4 / 1e9);
    }
    return 0;
}
