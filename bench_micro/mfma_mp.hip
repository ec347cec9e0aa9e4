// Time-boxed micro-benchmark: can the CONSTANT-operand half of a Montgomery product -- U = m * p, 256 x 256 -> 512
// bits, p fixed -- run on the int8 matrix cores instead of the 64 (v_mad_u64_u32, v_addc_co_u32) pairs it costs on the VALU?
//
// Formulation: m * p is a product with a fixed Toeplitz matrix.  Bytes of m for 16 elements form the B operand (K = 32 bytes, N = 16
// elements) of v_mfma_i32_16x16x32_i8, rows of the byte Toeplitz matrix of p the A operand: one MFMA yields 16 of the 64 byte-column sums
// (each < 32 * 255^2 < 2^21) for 16 elements; four MFMAs per 16 elements, sixteen per wave of 64 elements.  The A rows are ordered so that
// lane (g, j) -- g = lane / 16, j = lane % 16 -- ends up with the sixteen columns 16g .. 16g+15 of element j: four u32 limbs' worth.
// What the VALU still has to do per wave and 64 elements:
//   (1) recombine:   limb = c0 + (c1 << 8) + (c2 << 16) + (c3 << 24) + carry     -> 3 x v_mad_u64_u32 + one 64-bit add, x 16 limbs per lane
//   (2) propagate the limb carries across the four lanes that share an element    -> 3 steps of (cross-lane move + 4-limb add with carry)
//   (3) move m from "one element per lane" into the MFMA B layout and U_hi back    -> LDS round trips (NOT included below)
// Kernel B below runs the sixteen MFMAs plus (1) and (2) -- an UPPER BOUND on what the matrix-core path can deliver, since (3) and the
// second constant product (T_lo * N' mod 2^256) are left out -- against kernel A, the 64 mad + 64 addc pairs it would replace.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_mp.hip -o mfma_mp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../aztec-2.0_amd/csrc/mac_chains.hip.h"
using namespace bbg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 1024;
typedef int v4i __attribute__((ext_vector_type(4)));

// ---- A: U = m * p on the VALU, product scanning, every limb product one (mad, addc) pair; m <- U_hi (dependency between iterations)
__global__ void __launch_bounds__(256) k_valu_mp(uint32_t* out, const uint32_t* in)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m[8], p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        m[i] = in[(size_t)tid * 8 + i];
        p[i] = in[i] | 1u; // "the modulus": any fixed limbs (rates do not depend on the values)
    }
    for (int it = 0; it < ITERS; it++) {
        uint32_t u[16];
        uint64_t acc = 0;
        uint32_t c2 = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = k - i;
                if (j >= 0 && j < 8) mac1_v(acc, c2, m[i], p[j]);
            }
            u[k] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)c2 << 32);
            c2 = 0;
        }
        u[15] = (uint32_t)acc;
#pragma unroll
        for (int i = 0; i < 8; i++) m[i] = u[8 + i] ^ u[i];
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= m[i];
    out[tid] = s;
}

// ---- B: sixteen int8 MFMAs + column recombination + cross-lane carries per wave and 64 elements (layout moves excluded)
__device__ __forceinline__ void recombine(const v4i& c, uint64_t& carry, uint32_t& limb)
{
    // limb = c0 + c1 2^8 + c2 2^16 + c3 2^24 + carry_in; carry_out = the bits above 32
    uint64_t acc = (uint64_t)(uint32_t)c[0] + carry;
    acc = (uint64_t)(uint32_t)c[1] * 256u + acc;       // v_mad_u64_u32
    acc = (uint64_t)(uint32_t)c[2] * 65536u + acc;     // v_mad_u64_u32
    acc = (uint64_t)(uint32_t)c[3] * 16777216u + acc;  // v_mad_u64_u32
    limb = (uint32_t)acc;
    carry = acc >> 32;
}
__global__ void __launch_bounds__(256) k_mfma_mp(uint32_t* out, const uint32_t* in)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long a[4], b[4]; // A: four row blocks of the Toeplitz matrix of p (constant); B: m bytes of the four element groups
#pragma unroll
    for (int r = 0; r < 4; r++) {
        a[r] = (long)(((uint64_t)in[(size_t)tid * 8 + r] << 32) | in[r + 8]);
        b[r] = (long)(((uint64_t)in[(size_t)tid * 8 + 4 + r] << 32) | in[(size_t)tid * 8 + r]);
    }
    for (int it = 0; it < ITERS; it++) {
        uint32_t lim[4][4];
#pragma unroll
        for (int t = 0; t < 4; t++) { // element group t: 16 elements, this lane = (g, j) holds columns 16g .. 16g+15 of element 16t + j
            v4i c[4];
#pragma unroll
            for (int r = 0; r < 4; r++) c[r] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[r], b[t], v4i{0, 0, 0, 0}, 0, 0, 0);
            uint64_t carry = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) recombine(c[r], carry, lim[t][r]);
            // carries travel g -> g + 1 (three steps; the fourth lane's carry-out is the top of the 512-bit product)
#pragma unroll
            for (int step = 0; step < 3; step++) {
                uint32_t cin = __shfl_up((uint32_t)carry, 16);
                if (lane < 16) cin = 0;
                uint64_t s = (uint64_t)lim[t][0] + cin;
                lim[t][0] = (uint32_t)s;
#pragma unroll
                for (int r = 1; r < 4; r++) {
                    s = (uint64_t)lim[t][r] + (s >> 32);
                    lim[t][r] = (uint32_t)s;
                }
                carry = s >> 32;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) b[t] = (long)(((uint64_t)lim[t][3] << 32) | lim[t][2]) ^ (long)(((uint64_t)lim[t][1] << 32) | lim[t][0]);
    }
    uint32_t s = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) s ^= (uint32_t)b[t] ^ (uint32_t)(b[t] >> 32);
    out[tid] = s;
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
    CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&in, n * 32));
    std::vector<uint32_t> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s; }
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));
    const double products = (double)n * ITERS; // both kernels: one m * p per lane and iteration (B: 64 elements per wave and iteration)
    const double ta = time_it([&] { k_valu_mp<<<blocks, threads>>>(out, in); });
    const double tb = time_it([&] { k_mfma_mp<<<blocks, threads>>>(out, in); });
    printf("A  m*p on the VALU (64 mad + 64 addc)                              %8.3f ms  %8.2f G (m*p)/s\n", ta * 1e3, products / ta / 1e9);
    printf("B  16 x v_mfma_i32_16x16x32_i8 + recombination + carries (no moves) %8.3f ms  %8.2f G (m*p)/s   B/A = %.2f\n", tb * 1e3, products / tb / 1e9,
           ta / tb);
    printf("a full Montgomery product is a*b (64 pairs) + T_lo*N' (8 mul) + m*p (64 pairs): replacing m*p at ratio r changes the product rate by\n"
           "1 / (0.53 + 0.47 / r); >= 15 %% faster needs r >= 1.4 BEFORE the layout moves are paid for.\n");
    return 0;
}
