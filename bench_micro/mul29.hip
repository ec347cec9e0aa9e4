// 9 x 29-bit limbs against 8 x 32-bit limbs for the BN254 Montgomery product on gfx950.
// With 29-bit limbs a column of 9 a*b and 9 m*p products fits a 64-bit accumulator (18 * 2^58 < 2^63), so every limb product is ONE
// v_mad_u64_u32 and the 136 v_addc_co_u32 of the 32-bit formulation (field.hip.h fe_mul) disappear: 162 mads + ~45 shifts / masks
// against 136 mads + 136 addc + 8 mul_lo.  R' = 2^261; R-form values convert by a 5-bit shift folded into the limb split.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mul29.hip -o mul29
#include "../aztec-2.0_amd/csrc/field29.hip.h"
#include <cstdio>
#include <vector>
using namespace bbg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2048;

__global__ void __launch_bounds__(256) k_check(const uint32_t* in, int* bad, int n)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n) return;
    Fq x = fe_load<FqP>(in + (size_t)tid * 8), y = fe_load<FqP>(in + (size_t)(tid ^ 1) * 8);
    int fails = 0;
    const Fq z = fe_canon(fe_mul(x, y));
    { // product: (x * 2^5) * y / 2^261 = x * y / 2^256
        const F29<FqP> r = f29_mul(f29_from_fe<FqP, 5>(x), f29_from_fe<FqP, 0>(y));
        const Fq back = fe_canon(f29_to_fe(f29_carry(r)));
        if (!fe_eq(back, z)) fails |= 1;
    }
    { // square
        const F29<FqP> r = f29_sqr(f29_from_fe<FqP, 0>(x));
        // x^2 / 2^261 = (x^2 / 2^256) / 32: compare 32 * result against fe_sqr(x)
        Fq w = fe_canon(f29_to_fe(f29_carry(r)));
        for (int k = 0; k < 5; k++) w = fe_dbl(w);
        if (!fe_eq(fe_canon(w), fe_canon(fe_mul(x, x)))) fails |= 2;
    }
    { // lazy add / sub feeding a product: ((x + y) - 2y + ...) shapes
        const F29<FqP> X = f29_from_fe<FqP, 5>(x), Y = f29_from_fe<FqP, 5>(y);
        const F29<FqP> s = f29_add(f29_from_fe<FqP, 0>(x), f29_from_fe<FqP, 0>(y)), d = f29_carry(f29_sub<64>(X, Y)); // d = X - Y + 64p
        const F29<FqP> r = f29_mul(d, f29_carry(s));                    // 32 (x - y)(x + y) / 2^261 = (x^2 - y^2) / 2^256
        const Fq w = fe_sub(fe_mul(x, x), fe_mul(y, y));
        if (!fe_eq(fe_canon(f29_to_fe(f29_carry(r))), fe_canon(w))) fails |= 4;
    }
    { // a*b - c*d with one reduction
        const F29<FqP> X = f29_from_fe<FqP, 5>(x), Y = f29_from_fe<FqP, 0>(y);
        const F29<FqP> r = f29_mul_sub2(X, Y, f29_from_fe<FqP, 5>(y), f29_from_fe<FqP, 0>(x)); // x*y - y*x = 0 (mod p)
        if (!fe_eq(fe_canon(f29_to_fe(f29_carry(r))), Fq::zero())) fails |= 8;
        const F29<FqP> r2 = f29_mul_sub2(X, Y, f29_from_fe<FqP, 5>(y), f29_from_fe<FqP, 0>(y)); // x*y - y*y
        const Fq w = fe_sub(fe_mul(x, y), fe_mul(y, y));
        if (!fe_eq(fe_canon(f29_to_fe(f29_carry(r2))), fe_canon(w))) fails |= 16;
    }
    { // the interleaved pairs (what xyzz29_madd calls) against the single products: same limbs, not only the same residues
        const F29<FqP> X = f29_from_fe<FqP, 5>(x), Y = f29_from_fe<FqP, 0>(y), Z = f29_from_fe<FqP, 0>(x);
        const F29<FqP> s1 = f29_mul(X, Y), s2 = f29_mul(Y, Z);
        bool same = true;
        F29<FqP> q1, q2;
        f29_mul2(X, Y, Y, Z, q1, q2);
        for (int i = 0; i < 9; i++) same = same && q1.v[i] == s1.v[i] && q2.v[i] == s2.v[i];
        if (!same) fails |= 32;
        F29<FqP> t1, t2;
        f29_sqr2(Y, Z, t1, t2);
        const F29<FqP> u1 = f29_sqr(Y), u2 = f29_sqr(Z);
        same = true;
        for (int i = 0; i < 9; i++) same = same && t1.v[i] == u1.v[i] && t2.v[i] == u2.v[i];
        if (!same) fails |= 64;
        F29<FqP> w1, w2;
        f29_mul_sub2_mul(X, Y, Z, Y, Y, Z, w1, w2);
        const F29<FqP> v1 = f29_mul_sub2(X, Y, Z, Y), v2 = f29_mul(Y, Z);
        same = true;
        for (int i = 0; i < 9; i++) same = same && w1.v[i] == v1.v[i] && w2.v[i] == v2.v[i];
        if (!same) fails |= 128;
    }
    if (fails) atomicOr(bad, fails);
}

template <int V> __global__ void __launch_bounds__(256) mul_kernel(uint32_t* out, const uint32_t* in)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = fe_load<FqP>(in + (size_t)tid * 8);
    Fq y = fe_load<FqP>(in + (size_t)(tid ^ 1) * 8);
    if (V == 0) {
        for (int it = 0; it < ITERS / 4; it++) { x = fe_mul(x, y); y = fe_mul(y, x); }
        fe_store<FqP>(out + (size_t)tid * 8, fe_add(x, y));
    } else {
        F29<FqP> X = f29_from_fe<FqP, 0>(x), Y = f29_from_fe<FqP, 0>(y);
        for (int it = 0; it < ITERS / 4; it++) {
            if (V == 1) { X = f29_mul(X, Y); Y = f29_mul(Y, X); }
            if (V == 2) { X = f29_sqr(X); Y = f29_sqr(Y); }
            if (V == 3) { X = f29_mul(f29_carry(f29_sub<64>(X, Y)), Y); Y = f29_mul(f29_carry(f29_add(Y, X)), X); } // mul + add / sub + carry pass
        }
        fe_store<FqP>(out + (size_t)tid * 8, f29_to_fe(f29_carry(f29_add(X, Y))));
    }
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
    int* d_bad;
    CK(hipMalloc(&out, n * 32)); CK(hipMalloc(&in, n * 32)); CK(hipMalloc(&d_bad, 4));
    std::vector<uint32_t> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)s; }
    for (size_t i = 0; i < n; i++) h[i * 8 + 7] &= 0x3fffffffu; // < 2^254 < 2p
    for (int k = 0; k < 8; k++) { h[k] = 0; h[8 + k] = FqP::MOD[k]; h[16 + k] = k ? FqP::MOD[k] : FqP::MOD[0] - 1; h[24 + k] = k ? 0 : 1; } // 0, p, p-1, 1
    CK(hipMemcpy(in, h.data(), n * 32, hipMemcpyHostToDevice));
    int bad = 0;
    CK(hipMemcpy(d_bad, &bad, 4, hipMemcpyHostToDevice));
    k_check<<<blocks, threads>>>(in, d_bad, (int)n);
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("f29 check over %zu pairs: failure mask 0x%x (%s)\n", n, bad, bad ? "FAIL" : "PASS");

    const char* mnames[] = { "fe_mul 8x32 FIPS mad+addc (shipping)", "f29_mul 9x29 mad only", "f29_sqr 9x29 (doubled cross terms)", "f29 mul + add/sub + carry pass" };
#define MUL(V) { double t = time_it([&] { mul_kernel<V><<<blocks, threads>>>(out, in); }); \
        double cnt = (double)n * (ITERS / 4) * 2; \
        printf("%-40s %8.3f ms  %8.2f Gop/s\n", mnames[V], t * 1e3, cnt / t / 1e9); }
    MUL(0) MUL(1) MUL(2) MUL(3)
    return bad ? 1 : 0;
}
