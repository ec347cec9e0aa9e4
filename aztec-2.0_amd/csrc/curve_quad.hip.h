// Quad-cooperative XYZZ group operations: FOUR adjacent lanes (a DPP quad) carry the SAME operands and share one addition / doubling,
// each lane computing a different field product of the formula in the same instruction slots.
//
// Why: the reduce phase of an MSM (k_combine*, k_rowcol, k_final_planes, k_final_sum in msm.hip) is a CHAIN of ~40 dependent EC
// operations executed by a handful of waves -- pure latency, 8-12 us per operation on one lane (14 dependent Montgomery products of
// ~264 instructions each), 0.44 ms per MSM, and everything a small MSM costs (profiles/r02_msm_small_n.txt).  The lanes beside the
// working one are idle anyway.  add-2008-s has depth 4 when its products run side by side:
//     {U1, U2, S1, S2}  ->  {P^2, R^2, ZZ1 ZZ2, ZZZ1 ZZZ2}  ->  {P PP, U1 PP, ZZp PP, ZZZp P}  ->  {R (Q - X3), S1 PPP, -, T PP}
// i.e. 4 product times + ~13 quad broadcasts (8 DPP moves each) instead of 13.5 product times.  dbl-2008-s-1 likewise: depth 4
// instead of 8.5.  Results are the same group elements as curve.hip.h's serial formulas (possibly another coarse representative).
#pragma once
#include "curve.hip.h"
#include "field29.hip.h"

// The products of the quad operations.  These kernels are chains of dependent products run by a wave that is alone on its SIMD: what counts is
// the time of ONE product, not products per second.  BBG_QUAD_MUL29 = 1 (round 6): the product through the 29-bit multiplier with a column
// accumulator per column (field29.hip.h fe_mul29_ilp: the a*b products wait for nothing, the digit steps add nine independent products each) --
// 473 against 557 ns per dependent product (bench_micro/mul_latency.hip, profiles/r06_mul_latency.txt); same residues, same contract as fe_mul
// (operands < 2p, or < 4p and < p; result < 1.76p).  0: field.hip.h fe_mul (A/B).
#ifndef BBG_QUAD_MUL29
#define BBG_QUAD_MUL29 0
#endif

namespace bbg {

__device__ __forceinline__ Fq quad_mul(const Fq& a, const Fq& b)
{
#if BBG_QUAD_MUL29
    return fe_mul29_ilp(a, b);
#else
    return fe_mul(a, b);
#endif
}

// value held by lane K of the caller's quad, in every lane of the quad (v_mov_b32 with DPP quad_perm:[K,K,K,K])
template <int K> __device__ __forceinline__ Fq quad_bcast(const Fq& v)
{
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.v[i], K * 0x55, 0xf, 0xf, true);
    return r;
}
// Lane role q in 0..3: a0 for lane 0, a1 for lane 1, ...  Written as bit-field inserts under two lane masks, with the masks made opaque
// to the optimiser: as `q == 0 ? a0 : q == 1 ? ...` the compiler builds a switch over q, i.e. DIVERGENT branches inside every quad, keeps
// the operands in scratch to select them by pointer (432-560 B per lane), and -- measured with bench_micro/quad_check.hip -- inside a loop
// produced a wrong selection for lane 1.  v_bfi_b32 x 3 per limb has no control flow at all.
struct QuadRole {
    uint32_t odd, high; // all-ones when q & 1 / q & 2
};
__device__ __forceinline__ QuadRole quad_role(int q)
{
    QuadRole r{0u - (uint32_t)(q & 1), 0u - (uint32_t)((q >> 1) & 1)};
    asm volatile("" : "+v"(r.odd), "+v"(r.high));
    return r;
}
__device__ __forceinline__ Fq quad_pick2(uint32_t m, const Fq& a, const Fq& b) // a where m = 0, b where m = ~0
{
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}
__device__ __forceinline__ Fq quad_pick(const QuadRole& w, const Fq& a0, const Fq& a1, const Fq& a2, const Fq& a3)
{
    return quad_pick2(w.high, quad_pick2(w.odd, a0, a1), quad_pick2(w.odd, a2, a3));
}

// the point held by lane K of the quad, in every lane of the quad
template <int K> __device__ __forceinline__ Xyzz quad_bcast_point(const Xyzz& p)
{
    Xyzz r;
    r.x = quad_bcast<K>(p.x);
    r.y = quad_bcast<K>(p.y);
    r.zz = quad_bcast<K>(p.zz);
    r.zzz = quad_bcast<K>(p.zzz);
    return r;
}
// 2P; all four lanes of the quad pass the same p and receive the same result.  q = lane index within the quad.
__device__ __forceinline__ Xyzz xyzz_dbl_q4(const Xyzz& p, int q)
{
    if (xyzz_is_inf(p)) return p; // uniform across the quad
    const QuadRole w = quad_role(q);
    const Fq U = fe_dbl(p.y);
    // step 1: U^2 | x^2
    const Fq ux = quad_pick2(w.odd, U, p.x);
    Fq m = quad_mul(ux, ux);
    const Fq V = quad_bcast<0>(m), xx = quad_bcast<1>(m);
    const Fq M = fe_add(fe_dbl(xx), xx);
    // step 2: U V | x V | M^2 | V zz
    m = quad_mul(quad_pick(w, U, p.x, M, V), quad_pick(w, V, V, M, p.zz));
    const Fq W = quad_bcast<0>(m), S = quad_bcast<1>(m), MM = quad_bcast<2>(m), ZZ3 = quad_bcast<3>(m);
    Xyzz r;
    r.x = fe_sub(MM, fe_dbl(S));
    // step 3: M (S - X3) | W y | W zzz | -
    m = quad_mul(quad_pick(w, M, W, W, W), quad_pick(w, fe_sub(S, r.x), p.y, p.zzz, p.zzz));
    r.y = fe_sub(quad_bcast<0>(m), quad_bcast<1>(m));
    r.zz = ZZ3;
    r.zzz = quad_bcast<2>(m);
    return r;
}

// a + b, complete (identity / doubling / inverse cases as xyzz_add); same contract as xyzz_dbl_q4.
__device__ __forceinline__ Xyzz xyzz_add_q4(const Xyzz& a, const Xyzz& b, int q)
{
    if (xyzz_is_inf(b)) return a; // the operands are identical in the four lanes, so every branch is uniform across the quad
    if (xyzz_is_inf(a)) return b;
    const QuadRole w = quad_role(q);
    // step 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    Fq m = quad_mul(quad_pick(w, a.x, b.x, a.y, b.y), quad_pick(w, b.zz, a.zz, b.zzz, a.zzz));
    const Fq U1 = quad_bcast<0>(m), U2 = quad_bcast<1>(m), S1 = quad_bcast<2>(m), S2 = quad_bcast<3>(m);
    const Fq P = fe_sub(U2, U1), R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return xyzz_dbl_q4(a, q);
        return xyzz_inf();
    }
    // step 2: P^2 | R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
    m = quad_mul(quad_pick(w, P, R, a.zz, a.zzz), quad_pick(w, P, R, b.zz, b.zzz));
    const Fq PP = quad_bcast<0>(m), RR = quad_bcast<1>(m);
    // step 3: PPP = P PP | Q = U1 PP | ZZ3 = (ZZ1 ZZ2) PP | T = (ZZZ1 ZZZ2) P      (lanes 2, 3 continue from their own step-2 product)
    m = quad_mul(quad_pick(w, P, U1, m, m), quad_pick(w, PP, PP, PP, P));
    const Fq PPP = quad_bcast<0>(m), Q = quad_bcast<1>(m), ZZ3 = quad_bcast<2>(m);
    Xyzz r;
    r.x = fe_sub(fe_sub(RR, PPP), fe_dbl(Q));
    // step 4: R (Q - X3) | S1 PPP | - | ZZZ3 = T PP
    m = quad_mul(quad_pick(w, R, S1, S1, m), quad_pick(w, fe_sub(Q, r.x), PPP, PPP, PP));
    r.y = fe_sub(quad_bcast<0>(m), quad_bcast<1>(m));
    r.zz = ZZ3;
    r.zzz = quad_bcast<3>(m);
    return r;
}

// a + p, p affine (madd-2008-s), complete like xyzz_madd; same contract as xyzz_add_q4 (all four lanes pass the same operands).  Depth 4:
//     {U2 = x2 ZZ1, S2 = y2 ZZZ1}  ->  {P^2, R^2}  ->  {P PP, X1 PP, ZZ1 PP}  ->  {R (Q - X3), Y1 PPP, ZZZ1 PPP}
// (two or three of the four lanes carry a product per step; the others repeat a neighbour's so that no lane diverges).  Used by the
// accumulation of SMALL MSMs, whose run time is the latency of one lane's chain of mixed additions (msm_kernels.hip.h).
__device__ __forceinline__ Xyzz xyzz_madd_q4(const Xyzz& a, const Affine& p, int q)
{
    if (aff_is_inf(p)) return a; // operands are identical in the four lanes: every branch is uniform across the quad
    if (xyzz_is_inf(a)) return xyzz_from_affine(p);
    const QuadRole w = quad_role(q);
    // step 1: U2 | S2 | U2 | S2
    Fq m = quad_mul(quad_pick2(w.odd, p.x, p.y), quad_pick2(w.odd, a.zz, a.zzz));
    const Fq U2 = quad_bcast<0>(m), S2 = quad_bcast<1>(m);
    const Fq P = fe_sub(U2, a.x), R = fe_sub(S2, a.y);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return xyzz_dbl_q4(xyzz_from_affine(p), q);
        return xyzz_inf();
    }
    // step 2: P^2 | R^2 | P^2 | R^2
    const Fq pr = quad_pick2(w.odd, P, R);
    m = quad_mul(pr, pr);
    const Fq PP = quad_bcast<0>(m), RR = quad_bcast<1>(m);
    // step 3: PPP = P PP | Q = X1 PP | ZZ3 = ZZ1 PP | (Q again)
    m = quad_mul(quad_pick(w, P, a.x, a.zz, a.x), PP);
    const Fq PPP = quad_bcast<0>(m), Q = quad_bcast<1>(m), ZZ3 = quad_bcast<2>(m);
    Xyzz r;
    r.x = fe_sub(fe_sub(RR, PPP), fe_dbl(Q));
    // step 4: R (Q - X3) | Y1 PPP | ZZZ3 = ZZZ1 PPP | (Y1 PPP again)
    m = quad_mul(quad_pick(w, R, a.y, a.zzz, a.y), quad_pick(w, fe_sub(Q, r.x), PPP, PPP, PPP));
    r.y = fe_sub(quad_bcast<0>(m), quad_bcast<1>(m));
    r.zz = ZZ3;
    r.zzz = quad_bcast<2>(m);
    return r;
}

// (p0 + p1) + (p2 + p3) for the four DIFFERENT points p held by the lanes of a quad; every lane receives the sum
__device__ __forceinline__ Xyzz quad_sum4(const Xyzz& p, int q)
{
    const Xyzz a = xyzz_add_q4(quad_bcast_point<0>(p), quad_bcast_point<1>(p), q);
    const Xyzz b = xyzz_add_q4(quad_bcast_point<2>(p), quad_bcast_point<3>(p), q);
    return xyzz_add_q4(a, b, q);
}

} // namespace bbg
