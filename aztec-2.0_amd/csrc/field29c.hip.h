// A product by a CONSTANT on 9 x 29-bit limbs without Montgomery's m digits (round 5) -- the multiplier of the NTT's table twiddles.
//
// Every product inside a radix-8 NTT step has a table value as one operand.  For a constant w < p keep, beside its limbs, the quotient
// multiplier wq = floor(w 2^261 / p) (Shoup's trick, Barrett's reduction with the constant folded in).  For a lazily reduced x < V p:
//
//   q^ = floor( sum_{k >= 7} col_k(x, wq) 2^(29 k) / 2^261 )            columns 7 .. 16 of x * wq: 53 limb products
//        -- the true q = floor(x wq / 2^261) or q - 1: the dropped columns 0 .. 6 sum to < 2^238
//   r  = (x w + q^ (2^261 - p)) mod 2^261                               columns 0 .. 8 of both products: 45 + 45 limb products
//      = x w - q^ p exactly, because 0 <= x w - q^ p < (2 + V / 169) p < 2^261:
//        x w / p - x wq / 2^261 = x eps / (p 2^261) < V / 169 with eps = w 2^261 mod p, the floor costs < 1, q^ = q - 1 costs 1.
//
// 143 v_mad_u64_u32 against Montgomery's 162 + 9 (v_mul_lo + v_and) for the m digits, and no dependency of a column on the digit of the one
// before it: 200 against 180 G products/s (bench_micro/mul_shoup29.hip, profiles/r05_mul_shoup29.txt).  The result is x w mod p -- the same
// residue f29_mul(x, w R') leaves -- with EXACT limbs (all nine below 2^29) and value below (2 + V / 169) p where Montgomery leaves 1 + V / 169.
//
// Operand bounds: x limbs <= 2^31 + 2^29 (a column holds at most 9 x-limb * 29-bit products + 9 q * pbar products + the carried-in value:
// 9 (2^31.33 + 2^29) 2^29 < 2^64), x < 2^261 - slack (any lazily reduced value: V < 64); w, wq exact limbs.
#pragma once
#include "field29.hip.h"

namespace bbg {

// a table constant: limbs of w (< p) and of wq = floor(w 2^261 / p); 20-word rows in the tables (18 used)
constexpr int C29_ROW = 20;
template <class P> struct C29 {
    uint32_t w[9];
    uint32_t q[9];
};

// limb J of pbar = 2^261 - p
template <class P> constexpr uint32_t pbar29_limb(int j)
{
    uint32_t borrow = 0, out = 0;
    for (int i = 0; i <= j; i++) {
        const int64_t d = (int64_t)0 - (int64_t)k29_limb(P::MOD, i) - borrow; // the limbs of 2^261 below limb 9 are zero
        out = (uint32_t)(d & M29);
        borrow = d < 0 ? 1 : 0;
    }
    return out;
}
template <class P, int J> struct PB29 {
    static constexpr uint32_t value = pbar29_limb<P>(J);
};
// N products x[i] * pbar[J - i], i = 0 .. N - 1, the constant limbs as SGPR operands
template <class P, int N, int J> __device__ __forceinline__ void mad_col_pbar(uint64_t& acc, const uint32_t* x)
{
#define BBG_PL(I) PB29<P, (J - (I) >= 0 && J - (I) <= 8) ? J - (I) : 0>::value
    if constexpr (N == 1) mad1_s(acc, x[0], BBG_PL(0));
    else if constexpr (N == 2) mad2_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1));
    else if constexpr (N == 3) mad3_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2));
    else if constexpr (N == 4) mad4_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3));
    else if constexpr (N == 5) mad5_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4));
    else if constexpr (N == 6) mad6_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5));
    else if constexpr (N == 7)
        mad7_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6));
    else if constexpr (N == 8)
        mad8_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6), x[7],
               BBG_PL(7));
    else if constexpr (N == 9)
        mad9_s(acc, x[0], BBG_PL(0), x[1], BBG_PL(1), x[2], BBG_PL(2), x[3], BBG_PL(3), x[4], BBG_PL(4), x[5], BBG_PL(5), x[6], BBG_PL(6), x[7],
               BBG_PL(7), x[8], BBG_PL(8));
#undef BBG_PL
}
// N products x[i] * y[-i] with the y in SGPRs (a wave-uniform constant: the butterfly's own multipliers)
template <int N> __device__ __forceinline__ void mad_col_s(uint64_t& acc, const uint32_t* x, const uint32_t* y)
{
    if constexpr (N == 1) mad1_s(acc, x[0], y[0]);
    else if constexpr (N == 2) mad2_s(acc, x[0], y[0], x[1], y[-1]);
    else if constexpr (N == 3) mad3_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2]);
    else if constexpr (N == 4) mad4_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3]);
    else if constexpr (N == 5) mad5_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4]);
    else if constexpr (N == 6) mad6_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5]);
    else if constexpr (N == 7) mad7_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6]);
    else if constexpr (N == 8)
        mad8_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7]);
    else if constexpr (N == 9)
        mad9_s(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7], x[8], y[-8]);
}
// column K of a * b (both nine limbs): SCALAR = the b limbs live in SGPRs
template <int K, bool SCALAR> __device__ __forceinline__ void f29c_terms(uint64_t& acc, const uint32_t* a, const uint32_t* b)
{
    constexpr int lo = K > 8 ? K - 8 : 0, hi = K > 8 ? 8 : K;
    if constexpr (SCALAR) mad_col_s<hi - lo + 1>(acc, a + lo, b + (K - lo));
    else mad_col_v<hi - lo + 1>(acc, a + lo, b + (K - lo));
}
__device__ __forceinline__ void f29c_shr(uint64_t& acc) { asm("v_lshrrev_b64 %0, 29, %0" : "+v"(acc)); }

template <class P, bool SCALAR> struct F29CJob {
    const uint32_t *x, *w, *wq;
    uint64_t acc;
    uint32_t q[9];
    uint32_t* r;
};
// one column of the quotient phase (K = 7 .. 16) and of the remainder phase (K = 0 .. 8) of a job
template <class P, bool S, int K> __device__ __forceinline__ void f29c_qcol(F29CJob<P, S>& j)
{
    f29c_terms<K, S>(j.acc, j.x, j.wq);
    if constexpr (K >= 9) j.q[K - 9] = (uint32_t)j.acc & M29;
    f29c_shr(j.acc);
}
template <class P, bool S, int K> __device__ __forceinline__ void f29c_rcol(F29CJob<P, S>& j)
{
    f29c_terms<K, S>(j.acc, j.x, j.w);
    mad_col_pbar<P, K + 1, K>(j.acc, j.q);
    j.r[K] = (uint32_t)j.acc & M29;
    if constexpr (K < 8) f29c_shr(j.acc);
}
template <class P, bool S> __device__ __forceinline__ void f29c_run(F29CJob<P, S>& j)
{
    j.acc = 0;
#define BBG_X(K) f29c_qcol<P, S, K>(j);
    BBG_X(7) BBG_X(8) BBG_X(9) BBG_X(10) BBG_X(11) BBG_X(12) BBG_X(13) BBG_X(14) BBG_X(15) BBG_X(16)
#undef BBG_X
    j.q[8] = (uint32_t)j.acc;
    j.acc = 0;
#define BBG_X(K) f29c_rcol<P, S, K>(j);
    BBG_X(0) BBG_X(1) BBG_X(2) BBG_X(3) BBG_X(4) BBG_X(5) BBG_X(6) BBG_X(7) BBG_X(8)
#undef BBG_X
}
// two independent products, columns interleaved (the reason of f29_mul2: no s_nop between dependent asm statements, each chain hides the other's latency)
template <class P, bool S1, bool S2> __device__ __forceinline__ void f29c_run2(F29CJob<P, S1>& a, F29CJob<P, S2>& b)
{
    a.acc = 0;
    b.acc = 0;
#define BBG_X(K) f29c_qcol<P, S1, K>(a); f29c_qcol<P, S2, K>(b);
    BBG_X(7) BBG_X(8) BBG_X(9) BBG_X(10) BBG_X(11) BBG_X(12) BBG_X(13) BBG_X(14) BBG_X(15) BBG_X(16)
#undef BBG_X
    a.q[8] = (uint32_t)a.acc;
    b.q[8] = (uint32_t)b.acc;
    a.acc = 0;
    b.acc = 0;
#define BBG_X(K) f29c_rcol<P, S1, K>(a); f29c_rcol<P, S2, K>(b);
    BBG_X(0) BBG_X(1) BBG_X(2) BBG_X(3) BBG_X(4) BBG_X(5) BBG_X(6) BBG_X(7) BBG_X(8)
#undef BBG_X
}

// x * c mod p: exact limbs, value < (2 + V_x / 169) p.  SCALAR: c is wave-uniform and sits in SGPRs.
template <bool SCALAR = false, class P> __device__ __forceinline__ F29<P> f29_mulc(const F29<P>& x, const C29<P>& c)
{
    F29<P> r;
    F29CJob<P, SCALAR> j;
    j.x = x.v, j.w = c.w, j.wq = c.q, j.r = r.v;
    f29c_run(j);
    return r;
}
// r1 = x1 * c1, r2 = x2 * c2 (r may alias x)
template <bool S1 = false, bool S2 = false, class P>
__device__ __forceinline__ void f29_mulc2(const F29<P>& x1, const C29<P>& c1, const F29<P>& x2, const C29<P>& c2, F29<P>& r1, F29<P>& r2)
{
    F29<P> t1, t2;
    F29CJob<P, S1> a;
    F29CJob<P, S2> b;
    a.x = x1.v, a.w = c1.w, a.wq = c1.q, a.r = t1.v;
    b.x = x2.v, b.w = c2.w, b.wq = c2.q, b.r = t2.v;
    f29c_run2(a, b);
    r1 = t1;
    r2 = t2;
}

} // namespace bbg
