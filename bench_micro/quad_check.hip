// Checks curve_quad.hip.h against curve.hip.h on the device: for points k1*G, k2*G (serial double-and-add) the quad-cooperative
// add / dbl must denote the same group element as the serial formulas (compared projectively), incl. P + P, P + (-P), inf cases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../aztec-2.0_amd/csrc/curve_quad.hip.h"
using namespace bbg;

__device__ Xyzz mul_small(uint64_t k)
{
    Affine g;
    g.x = fe_to_mont(Fq::zero()); // placeholder, overwritten below
    Fq one = Fq::zero(); one.v[0] = 1;
    Fq two = Fq::zero(); two.v[0] = 2;
    g.x = fe_to_mont(one);
    g.y = fe_to_mont(two);
    Xyzz acc = xyzz_inf();
    for (int i = 63; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((k >> i) & 1) acc = xyzz_madd(acc, g);
    }
    return acc;
}
__device__ bool same_point(const Xyzz& a, const Xyzz& b)
{
    if (xyzz_is_inf(a) || xyzz_is_inf(b)) return xyzz_is_inf(a) && xyzz_is_inf(b);
    const Fq l1 = fe_reduce_once(fe_reduce_once(fe_mul(a.x, b.zz))), r1 = fe_reduce_once(fe_reduce_once(fe_mul(b.x, a.zz)));
    const Fq l2 = fe_reduce_once(fe_reduce_once(fe_mul(a.y, b.zzz))), r2 = fe_reduce_once(fe_reduce_once(fe_mul(b.y, a.zzz)));
    bool ok = true;
    for (int i = 0; i < 8; i++) ok = ok && l1.v[i] == r1.v[i] && l2.v[i] == r2.v[i];
    // ZZ^3 == ZZZ^2 must hold for the result as well
    const Fq c1 = fe_reduce_once(fe_reduce_once(fe_mul(fe_sqr(a.zz), a.zz))), c2 = fe_reduce_once(fe_reduce_once(fe_sqr(a.zzz)));
    for (int i = 0; i < 8; i++) ok = ok && c1.v[i] == c2.v[i];
    return ok;
}
__global__ void k_check(int* bad)
{
    const int lt = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, q = threadIdx.x & 3;
    const uint64_t k1 = 0x9E3779B97F4A7C15ULL * (lt + 1), k2 = 0xBF58476D1CE4E5B9ULL * (lt + 7);
    const Xyzz a = mul_small(k1), b = mul_small(k2);
    int fails = 0;
    if (!same_point(xyzz_add_q4(a, b, q), xyzz_add(a, b))) fails |= 1;
    if (!same_point(xyzz_dbl_q4(a, q), xyzz_dbl(a))) fails |= 2;
    if (!same_point(xyzz_add_q4(a, a, q), xyzz_dbl(a))) fails |= 4;
    Xyzz na = a;
    na.y = fe_neg(a.y);
    if (!xyzz_is_inf(xyzz_add_q4(a, na, q))) fails |= 8;
    if (!same_point(xyzz_add_q4(a, xyzz_inf(), q), a) || !same_point(xyzz_add_q4(xyzz_inf(), b, q), b)) fails |= 16;
    // a chain: ((a + b) + a) doubled twice
    Xyzz c = xyzz_add_q4(xyzz_add_q4(a, b, q), a, q), cs = xyzz_add(xyzz_add(a, b), a);
    c = xyzz_dbl_q4(xyzz_dbl_q4(c, q), q);
    cs = xyzz_dbl(xyzz_dbl(cs));
    if (!same_point(c, cs)) fails |= 32;
    { // mixed additions (xyzz_madd_q4, the small-MSM accumulation): general case, acc = inf, P = acc (doubling), P = -acc, P = inf, a chain
        Affine pb;
        {
            const Fq izz = fq_invert(fe_mul(b.zz, b.zzz));
            pb.x = fe_reduce_once(fe_mul(b.x, fe_mul(izz, b.zzz)));
            pb.y = fe_reduce_once(fe_mul(b.y, fe_mul(izz, b.zz)));
        }
        if (!same_point(xyzz_madd_q4(a, pb, q), xyzz_madd(a, pb))) fails |= 0x1000;
        if (!same_point(xyzz_madd_q4(xyzz_inf(), pb, q), xyzz_from_affine(pb))) fails |= 0x2000;
        if (!same_point(xyzz_madd_q4(b, pb, q), xyzz_dbl(b))) fails |= 0x4000;
        Affine nb = pb;
        nb.y = fe_neg(pb.y);
        if (!xyzz_is_inf(xyzz_madd_q4(b, nb, q))) fails |= 0x8000;
        if (!same_point(xyzz_madd_q4(a, aff_inf(), q), a)) fails |= 0x10000;
        Xyzz c2 = xyzz_inf(), c2s = xyzz_inf();
        for (int k = 0; k < 5; k++) {
            c2 = xyzz_madd_q4(c2, k & 1 ? nb : pb, q);
            c2s = xyzz_madd(c2s, k & 1 ? nb : pb);
            c2 = xyzz_madd_q4(xyzz_add_q4(c2, a, q), pb, q);
            c2s = xyzz_madd(xyzz_add(c2s, a), pb);
        }
        if (!same_point(c2, c2s)) fails |= 0x20000;
    }
    if (fails) atomicOr(bad, fails);
}
// the shapes msm.hip's reduce kernels use: quads of one wave taking different branches, a runtime-length doubling chain run by quad 0 only
__global__ void __launch_bounds__(512) k_check_shapes(int* bad, int t, Xyzz* out)
{
    const int lt = threadIdx.x >> 2, q = threadIdx.x & 3;
    Xyzz v = xyzz_inf(), vs = xyzz_inf();
    int fails = 0;
    for (int lo = lt; lo < 512; lo += 128)
        if (((lo + 1) >> t) & 1) {
            const Xyzz c = mul_small(0x9E3779B97F4A7C15ULL * (lo + 1));
            v = xyzz_add_q4(v, c, q);
            vs = xyzz_add(vs, c);
        }
    if (!same_point(v, vs)) fails |= 64;
    if (lt == 0) {
        for (int k = 0; k < t; k++) v = xyzz_dbl_q4(v, q);
        for (int k = 0; k < t; k++) vs = xyzz_dbl(vs);
        if (!same_point(v, vs)) fails |= 128;
        if (q == 0) xyzz_store(out, v);
    }
    if (fails) atomicOr(bad, fails);
}
__global__ void __launch_bounds__(512) k_check_variants(int* bad, int t)
{
    const int lt = threadIdx.x >> 2, q = threadIdx.x & 3;
    const Xyzz a = mul_small(0x9E3779B97F4A7C15ULL * (lt + 1));
    int fails = 0;
    { // once, quad 0 only
        Xyzz v = a, vs = a;
        if (lt == 0) { v = xyzz_dbl_q4(v, q); vs = xyzz_dbl(vs); if (!same_point(v, vs)) fails |= 0x100; }
    }
    { // loop, all lanes
        Xyzz v = a, vs = a;
        for (int k = 0; k < t; k++) v = xyzz_dbl_q4(v, q);
        for (int k = 0; k < t; k++) vs = xyzz_dbl(vs);
        if (!same_point(v, vs)) fails |= 0x200;
    }
    { // loop, first wave
        Xyzz v = a, vs = a;
        if (threadIdx.x < 64) {
            for (int k = 0; k < t; k++) v = xyzz_dbl_q4(v, q);
            for (int k = 0; k < t; k++) vs = xyzz_dbl(vs);
            if (!same_point(v, vs)) fails |= 0x400;
        }
    }
    { // loop, quad 0 only, point not from a q4 op
        Xyzz v = a, vs = a;
        if (lt == 0) {
            for (int k = 0; k < t; k++) v = xyzz_dbl_q4(v, q);
            for (int k = 0; k < t; k++) vs = xyzz_dbl(vs);
            if (!same_point(v, vs)) fails |= 0x800;
        }
    }
    if (fails) atomicOr(bad, fails);
}
int main()
{
    int* d_bad;
    int bad = 0;
    hipMalloc(&d_bad, 4);
    hipMemcpy(d_bad, &bad, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(8), dim3(256), 0, 0, d_bad);
    hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost); // (the per-t passes below reuse the flag word)
    printf("k_check (add / dbl / madd, all lanes) mask 0x%x\n", bad);
    Xyzz* d_out;
    hipMalloc(&d_out, sizeof(Xyzz));
    for (int t = 0; t < 15; t++) {
        int z = 0, r = 0;
        hipMemcpy(d_bad, &z, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check_variants, dim3(1), dim3(512), 0, 0, d_bad, t);
        hipMemcpy(&r, d_bad, 4, hipMemcpyDeviceToHost);
        printf("t=%d variants mask 0x%x\n", t, r);
        bad |= r;
    }
    hipMemcpy(d_bad, &bad, 4, hipMemcpyHostToDevice);
    for (int t = 0; t < 15; t++) hipLaunchKernelGGL(k_check_shapes, dim3(1), dim3(512), 0, 0, d_bad, t, d_out);
    hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
    printf("quad_check: failure mask 0x%x (%s)\n", bad, bad ? "FAIL" : "PASS");
    return bad ? 1 : 0;
}
