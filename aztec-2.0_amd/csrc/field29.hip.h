// BN254 field arithmetic on 9 x 29-bit limbs (gfx950): the multiplier of the MSM bucket accumulation.
//
// Why a second limb format.  v_mad_u64_u32 computes a 32 x 32 product plus a 64-bit addend in one VALU instruction, but with full 32-bit
// limbs the running column sum of a product-scanning Montgomery multiplication overflows 64 bits after two products, so field.hip.h's
// fe_mul pays one v_addc_co_u32 per limb product (136 mads + 136 addc + 8 mul_lo).  With 29-bit limbs a column holds at most 9 a*b and
// 9 m*p products of < 2^58 each: 18 * 2^58 < 2^63, the accumulator never overflows and a limb product is ONE instruction --
// 81 + 81 mads, 9 (mul_lo + and) for the m digits and 17 column shifts.  Same algorithm as the reference's montgomery_mul
// (ecc/fields/field_impl_generic.hpp:392-442), radix 2^29 instead of 2^64, Montgomery radix R' = 2^261.
//
// Ranges.  A value is 9 limbs v[0..8], value = sum v[i] 2^(29 i) < 2^261 + slack; limbs may exceed 29 bits ("lazy"): additions are
// limbwise without carries, f29_carry() is one parallel carry pass (limbs < 2^29 + 8 afterwards).  f29_mul needs
// 9 * maxlimb(a) * maxlimb(b) + 9 * 2^58 < 2^64, i.e. bitlen(a limbs) + bitlen(b limbs) <= 60, and returns
// (a * b + m * p) / R' < a * b / R' + p with limbs < 2^29 (the top one unmasked).  p / R' = 2^-7.4: operands of 32p and 2p give < 1.4p.
//
// Conversion.  Device buffers hold the reference's residues x * 2^256 (8 x u32).  (x * 2^256) << 5 = x * 2^261 as an integer < 2^261
// when x * 2^256 mod p is stored below 2^256, so R-form -> R'-form is the limb split at a 5-bit offset (f29_from_fe<P, 5>), free.  Back
// is a division by 32 mod p: one 5-bit Montgomery step against a table of the multiples of p (f29_div32_to_fe).
#pragma once
#include "field.hip.h"
#include "mad_chains29.hip.h"

namespace bbg {

constexpr uint32_t M29 = 0x1fffffffu;

template <class P> struct F29 {
    uint32_t v[9];
};

// bits [29 j, 29 j + 29) of the 256-bit constant w (limb j of w in radix 2^29)
constexpr uint32_t k29_limb(const uint32_t* w, int j)
{
    const int b = 29 * j, i = b >> 5, o = b & 31;
    const uint64_t lo = i < 8 ? w[i] : 0, hi = i + 1 < 8 ? w[i + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> o) & M29);
}
// limb j of M * p (M < 2^7: M p < 2^261)
constexpr uint32_t k29_limb_times(const uint32_t* w, int j, uint32_t mult)
{
    uint32_t t[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    uint64_t carry = 0;
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)w[i] * mult + carry;
        t[i] = (uint32_t)x;
        carry = x >> 32;
    }
    t[8] = (uint32_t)carry;
    const int b = 29 * j, i = b >> 5, o = b & 31;
    const uint64_t lo = t[i], hi = t[i + 1];
    return (uint32_t)(((lo | (hi << 32)) >> o) & M29);
}
template <class P, int J> struct P29 {
    static constexpr uint32_t value = k29_limb(P::MOD, J);
};
// M p written with every limb below the top raised by 2^E (borrowed from the limb above): limbwise a + spread - b never goes negative
// for b limbs <= 2^E - 2 and a top limb of b <= that of M p minus 2^(E - 29); the value added is exactly M p.
template <class P, int J, int M, int E> struct Spread29 {
    static constexpr uint32_t q = k29_limb_times(P::MOD, J, M);
    static constexpr uint32_t up = 1u << E, down = 1u << (E - 29);
    static constexpr uint32_t value = J == 0 ? q + up : (J < 8 ? q + up - down : q - down);
};

// limb J of (x << S) for a 256-bit x in 8 x u32
template <int S, int J> __device__ __forceinline__ uint32_t f29_split_limb(const uint32_t* x)
{
    constexpr int b = 29 * J - S;
    if constexpr (b < 0) return (x[0] << (-b)) & M29;
    else {
        constexpr int i = b >> 5, o = b & 31;
        if constexpr (i >= 8) return 0;
        else if constexpr (o == 0) return x[i] & M29;
        else if constexpr (o <= 3) return (x[i] >> o) & M29;
        else if constexpr (i == 7) return x[i] >> o;
        else return __builtin_amdgcn_alignbit(x[i + 1], x[i], o) & M29;
    }
}
template <class P, int S> __device__ __forceinline__ F29<P> f29_from_fe(const Fe<P>& x)
{
    F29<P> r;
    r.v[0] = f29_split_limb<S, 0>(x.v);
    r.v[1] = f29_split_limb<S, 1>(x.v);
    r.v[2] = f29_split_limb<S, 2>(x.v);
    r.v[3] = f29_split_limb<S, 3>(x.v);
    r.v[4] = f29_split_limb<S, 4>(x.v);
    r.v[5] = f29_split_limb<S, 5>(x.v);
    r.v[6] = f29_split_limb<S, 6>(x.v);
    r.v[7] = f29_split_limb<S, 7>(x.v);
    r.v[8] = f29_split_limb<S, 8>(x.v);
    return r;
}

// one parallel carry pass: limbs < 2^32 in, limbs < 2^29 + 8 out (top limb: whatever the value needs)
template <class P> __device__ __forceinline__ F29<P> f29_carry(const F29<P>& a)
{
    F29<P> r;
    r.v[0] = a.v[0] & M29;
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = (a.v[i] & M29) + (a.v[i - 1] >> 29);
    r.v[8] = a.v[8] + (a.v[7] >> 29);
    return r;
}
// exact limbs (< 2^29 each below the top): sequential carries
template <class P> __device__ __forceinline__ F29<P> f29_norm(const F29<P>& a)
{
    F29<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = a.v[i] + c;
        r.v[i] = t & M29;
        c = t >> 29;
    }
    r.v[8] = a.v[8] + c;
    return r;
}
// 8 x u32 words of an exactly normalised value < 2^256
template <int I> __device__ __forceinline__ uint32_t f29_join_word(const uint32_t* a)
{
    constexpr int j = (32 * I) / 29, o = 32 * I - 29 * j;
    uint32_t w = a[j] >> o;
    if constexpr (j + 1 < 9) w |= a[j + 1] << (29 - o);
    if constexpr (j + 2 < 9 && 58 - o < 32) w |= a[j + 2] << (58 - o);
    return w;
}
template <class P> __device__ __forceinline__ Fe<P> f29_to_fe(const F29<P>& a0)
{
    const F29<P> a = f29_norm(a0);
    Fe<P> r;
    r.v[0] = f29_join_word<0>(a.v);
    r.v[1] = f29_join_word<1>(a.v);
    r.v[2] = f29_join_word<2>(a.v);
    r.v[3] = f29_join_word<3>(a.v);
    r.v[4] = f29_join_word<4>(a.v);
    r.v[5] = f29_join_word<5>(a.v);
    r.v[6] = f29_join_word<6>(a.v);
    r.v[7] = f29_join_word<7>(a.v);
    return r;
}

// ---- R'-form -> R-form without a multiplication: x R' / 32 = x R.  Division by 32 mod p = one Montgomery step with a 5-bit digit:
// m = -x p^-1 mod 32 makes x + m p divisible by 32.  The 32 multiples m p come from a table (LDS, filled by f29_fill_div32_table: rows of 12 words,
// 9 used); the shift by 5 bits is folded into the word join.  Input < 2^261 - 31p (any lazily reduced value), output < (x + 31p) / 32.
constexpr int DIV32_ROW = 12;
template <class P> __device__ __forceinline__ void f29_fill_div32_table(uint32_t* tbl, int m)
{
    // row m = m * p in radix 2^29 (exact limbs, the top one < 2^27)
    uint64_t carry = 0;
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)P::MOD[i] * (uint32_t)m + carry;
        w[i] = (uint32_t)x;
        carry = x >> 32;
    }
    w[8] = (uint32_t)carry;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int b = 29 * j, i = b >> 5, o = b & 31;
        const uint64_t lo = w[i], hi = i + 1 < 9 ? w[i + 1] : 0;
        tbl[m * DIV32_ROW + j] = (uint32_t)(((lo | (hi << 32)) >> o) & M29);
    }
}
template <int I> __device__ __forceinline__ uint32_t f29_join_word_div32(const uint32_t* a)
{
    constexpr int b = 32 * I + 5, j = b / 29, o = b - 29 * j;
    uint32_t w = a[j] >> o;
    if constexpr (j + 1 < 9) w |= a[j + 1] << (29 - o);
    if constexpr (j + 2 < 9 && 58 - o < 32) w |= a[j + 2] << (58 - o);
    return w;
}
template <class P> __device__ __forceinline__ Fe<P> f29_div32_to_fe(const F29<P>& x, const uint32_t* tbl)
{
    constexpr uint32_t NINV5 = (32u - ((P::MOD[0] * P::MOD[0] * P::MOD[0] * P::MOD[0] * P::MOD[0] * P::MOD[0] * P::MOD[0]) & 31u)) & 31u; // -p^-1 mod 32 (p^8 = 1 mod 32)
    const uint32_t m = (x.v[0] * NINV5) & 31u;
    const uint4* row = reinterpret_cast<const uint4*>(tbl + m * DIV32_ROW);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2];
    F29<P> t;
    t.v[0] = x.v[0] + r0.x; t.v[1] = x.v[1] + r0.y; t.v[2] = x.v[2] + r0.z; t.v[3] = x.v[3] + r0.w;
    t.v[4] = x.v[4] + r1.x; t.v[5] = x.v[5] + r1.y; t.v[6] = x.v[6] + r1.z; t.v[7] = x.v[7] + r1.w;
    t.v[8] = x.v[8] + r2.x;
    const F29<P> a = f29_norm(t);
    Fe<P> r;
    r.v[0] = f29_join_word_div32<0>(a.v);
    r.v[1] = f29_join_word_div32<1>(a.v);
    r.v[2] = f29_join_word_div32<2>(a.v);
    r.v[3] = f29_join_word_div32<3>(a.v);
    r.v[4] = f29_join_word_div32<4>(a.v);
    r.v[5] = f29_join_word_div32<5>(a.v);
    r.v[6] = f29_join_word_div32<6>(a.v);
    r.v[7] = f29_join_word_div32<7>(a.v);
    return r;
}

template <class P> __device__ __forceinline__ F29<P> f29_add(const F29<P>& a, const F29<P>& b)
{
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a - b + M p, limbwise (b limbs <= 2^E - 2, b < M p with the top-limb margin above); result limbs < a limb + 2^E + 2^29
template <class P, int J, int K, int E> __device__ __forceinline__ uint32_t f29_sub_limb(uint32_t a, uint32_t b)
{
    return a + (Spread29<P, J, K, E>::value - b);
}
template <int K, int E = 30, class P> __device__ __forceinline__ F29<P> f29_sub(const F29<P>& a, const F29<P>& b)
{
    F29<P> r;
    r.v[0] = f29_sub_limb<P, 0, K, E>(a.v[0], b.v[0]);
    r.v[1] = f29_sub_limb<P, 1, K, E>(a.v[1], b.v[1]);
    r.v[2] = f29_sub_limb<P, 2, K, E>(a.v[2], b.v[2]);
    r.v[3] = f29_sub_limb<P, 3, K, E>(a.v[3], b.v[3]);
    r.v[4] = f29_sub_limb<P, 4, K, E>(a.v[4], b.v[4]);
    r.v[5] = f29_sub_limb<P, 5, K, E>(a.v[5], b.v[5]);
    r.v[6] = f29_sub_limb<P, 6, K, E>(a.v[6], b.v[6]);
    r.v[7] = f29_sub_limb<P, 7, K, E>(a.v[7], b.v[7]);
    r.v[8] = f29_sub_limb<P, 8, K, E>(a.v[8], b.v[8]);
    return r;
}

// ---- Montgomery product, product scanning over 29-bit limbs
// N products x[i] * y[-i] into acc (mad_chains29.hip.h: one asm statement)
template <int N> __device__ __forceinline__ void mad_col_v(uint64_t& acc, const uint32_t* x, const uint32_t* y)
{
    if constexpr (N == 1) mad1_v(acc, x[0], y[0]);
    else if constexpr (N == 2) mad2_v(acc, x[0], y[0], x[1], y[-1]);
    else if constexpr (N == 3) mad3_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2]);
    else if constexpr (N == 4) mad4_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3]);
    else if constexpr (N == 5) mad5_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4]);
    else if constexpr (N == 6) mad6_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5]);
    else if constexpr (N == 7) mad7_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6]);
    else if constexpr (N == 8)
        mad8_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7]);
    else if constexpr (N == 9)
        mad9_v(acc, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7], x[8], y[-8]);
}
// N products x[i] * p[J - i], modulus limbs as SGPR operands
template <class P, int N, int J> __device__ __forceinline__ void mad_col_mod(uint64_t& acc, const uint32_t* x)
{
#define BBG_PL(I) P29<P, (J - (I) >= 0 && J - (I) <= 8) ? J - (I) : 0>::value
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
template <class P, int K> __device__ __forceinline__ void f29_mp_terms(uint64_t& acc, const uint32_t* m)
{
    // sum over i of m[i] * p[K - i], i in [max(0, K - 8), min(K, 8)] except the i = K term of the low columns (added after m[K] exists)
    constexpr int lo = K > 8 ? K - 8 : 0, hi = K > 8 ? 8 : K - 1;
    if constexpr (lo <= hi) mad_col_mod<P, hi - lo + 1, K - lo>(acc, m + lo);
}
template <int K> __device__ __forceinline__ void f29_ab_terms(uint64_t& acc, const uint32_t* a, const uint32_t* b)
{
    constexpr int lo = K > 8 ? K - 8 : 0, hi = K > 8 ? 8 : K;
    mad_col_v<hi - lo + 1>(acc, a + lo, b + (K - lo));
}
// the reduction half shared by every product shape: closes column K after its a*b terms were added
template <class P, int K> __device__ __forceinline__ void f29_close_column(uint64_t& acc, uint32_t* m, uint32_t* r)
{
    f29_mp_terms<P, K>(acc, m);
    if constexpr (K <= 8) {
        m[K] = ((uint32_t)acc * (P::INV & M29)) & M29;
        mad1_s(acc, m[K], P29<P, 0>::value);
    } else {
        r[K - 9] = (uint32_t)acc & M29;
    }
    asm("v_lshrrev_b64 %0, 29, %0" : "+v"(acc)); // (as asm like the chains: the compiler must not split the column sums off the carried-in value)
}
#define BBG_F29_COLUMNS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

template <class P> __device__ __forceinline__ F29<P> f29_mul(const F29<P>& a, const F29<P>& b)
{
    uint64_t acc = 0;
    uint32_t m[9];
    F29<P> r;
#define BBG_X(K) f29_ab_terms<K>(acc, a.v, b.v); f29_close_column<P, K>(acc, m, r.v);
    BBG_F29_COLUMNS(BBG_X)
#undef BBG_X
    r.v[8] = (uint32_t)acc;
    return r;
}

// a^2: cross terms once against the doubled operand (limbs < 2^30: still inside the column bound)
template <int K> __device__ __forceinline__ void f29_sq_terms(uint64_t& acc, const uint32_t* a, const uint32_t* d)
{
    constexpr int lo = K > 8 ? K - 8 : 0, n = (K + 1) / 2 - lo; // i = lo .. (K - 1) / 2
    if constexpr (n > 0) mad_col_v<n>(acc, d + lo, a + (K - lo));
    if constexpr (K % 2 == 0) mad1_v(acc, a[K / 2], a[K / 2]);
}
template <class P> __device__ __forceinline__ F29<P> f29_sqr(const F29<P>& a)
{
    uint64_t acc = 0;
    uint32_t m[9], d[9];
    F29<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
#define BBG_X(K) f29_sq_terms<K>(acc, a.v, d); f29_close_column<P, K>(acc, m, r.v);
    BBG_F29_COLUMNS(BBG_X)
#undef BBG_X
    r.v[8] = (uint32_t)acc;
    return r;
}

// (a * b - c * d) / R' + multiple of p with ONE reduction (the Y3 shape of the XYZZ group laws; field.hip.h fe_mul_sub2):
// c is replaced by 64p - c limbwise (limbs < 2^31 for c limbs < 2^30), d must be carried (limbs < 2^29 + 8):
// columns stay below 9 * 2^58 + 9 * 2^60 + 9 * 2^58 < 2^64.  c < 63p.
template <class P> __device__ __forceinline__ F29<P> f29_mul_sub2(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d)
{
    F29<P> z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.v[i] = 0;
    const F29<P> nc = f29_sub<64, 30>(z, c);
    uint64_t acc = 0;
    uint32_t m[9];
    F29<P> r;
#define BBG_X(K) f29_ab_terms<K>(acc, a.v, b.v); f29_ab_terms<K>(acc, nc.v, d.v); f29_close_column<P, K>(acc, m, r.v);
    BBG_F29_COLUMNS(BBG_X)
#undef BBG_X
    r.v[8] = (uint32_t)acc;
    return r;
}

// a * b / 2^256 for 8 x u32 operands through the 29-bit multiplier: (a << 5) * b / 2^261.  Same contract as field.hip.h fe_mul: operands < 2p (or
// < 4p and < p) give a result < 128 p^2 / 2^261 + p = 1.76p.
template <class P> __device__ __forceinline__ Fe<P> fe_mul29(const Fe<P>& a, const Fe<P>& b)
{
    return f29_to_fe(f29_mul(f29_from_fe<P, 5>(a), f29_from_fe<P, 0>(b)));
}

// ---- the product for LATENCY-bound callers (a wave alone on its SIMD running a chain of dependent products: the MSM reduce chains).  f29_mul
// above is written for throughput: ONE accumulator runs through all seventeen columns, so each of its 162 + 26 instructions waits for the one
// before it (other waves fill the gaps where there are any).  Here every column sums in an accumulator of its own -- the 81 a*b products depend on
// nothing but the operands -- and each of the nine digit steps adds nine m*p products to nine different columns; only the digit chain
// (column sum -> digit -> carry -> next column) is serial.  Plain C++, so that the compiler may interleave the columns.  Same result as f29_mul
// (same columns, same digits).  Measured: bench_micro/mul_latency.hip.
template <class P> __device__ __forceinline__ F29<P> f29_mul_ilp(const F29<P>& a, const F29<P>& b)
{
    uint64_t t[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t s = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j <= 8) s += (uint64_t)a.v[i] * b.v[j];
        }
        t[k] = s;
    }
    constexpr uint32_t pl[9] = { P29<P, 0>::value, P29<P, 1>::value, P29<P, 2>::value, P29<P, 3>::value, P29<P, 4>::value,
                                 P29<P, 5>::value, P29<P, 6>::value, P29<P, 7>::value, P29<P, 8>::value };
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint64_t ti = t[i] + carry;
        const uint32_t m = ((uint32_t)ti * (P::INV & M29)) & M29;
        carry = (ti + (uint64_t)m * pl[0]) >> 29; // the low 29 bits are zero by the choice of m
#pragma unroll
        for (int j = 1; j < 9; j++) t[i + j] += (uint64_t)m * pl[j];
    }
    F29<P> r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t v = t[9 + k] + carry;
        r.v[k] = (uint32_t)v & M29;
        carry = v >> 29;
    }
    r.v[8] = (uint32_t)carry;
    return r;
}
// fe_mul29 with it: a * b / 2^256 for 8 x u32 operands, same contract as field.hip.h fe_mul
template <class P> __device__ __forceinline__ Fe<P> fe_mul29_ilp(const Fe<P>& a, const Fe<P>& b)
{
    return f29_to_fe(f29_mul_ilp(f29_from_fe<P, 5>(a), f29_from_fe<P, 0>(b)));
}

// ---- two independent products, columns interleaved.  hipcc cannot see inside an asm statement and puts an `s_nop 0` behind every one whose
// result the NEXT instruction reads (gfx940's trans-use hazard, assumed for any inline asm): ~45 per multiplication, 1.2 clocks each at three
// waves per SIMD (bench_micro/issue_rates.hip) -- 5 % of the bucket accumulation.  With two products in flight every statement is followed by
// the other product's, and the independent chains also hide each other's latency.
enum F29Shape { F29_MUL, F29_SQR, F29_MULSUB2 };
template <class P, int SHAPE> struct F29Job {
    const uint32_t *a, *b, *c, *d; // MUL: a * b; SQR: a * a (b = doubled a); MULSUB2: a * b + c * d (c already negated)
    uint64_t acc;
    uint32_t m[9];
    uint32_t* r;
};
template <class P, int SHAPE, int K> __device__ __forceinline__ void f29_job_terms(F29Job<P, SHAPE>& j)
{
    if constexpr (SHAPE == F29_SQR) f29_sq_terms<K>(j.acc, j.a, j.b);
    else {
        f29_ab_terms<K>(j.acc, j.a, j.b);
        if constexpr (SHAPE == F29_MULSUB2) f29_ab_terms<K>(j.acc, j.c, j.d);
    }
}
template <class P, int S1, int S2, int K> __device__ __forceinline__ void f29_pair_column(F29Job<P, S1>& x, F29Job<P, S2>& y)
{
    f29_job_terms<P, S1, K>(x);
    f29_job_terms<P, S2, K>(y);
    f29_mp_terms<P, K>(x.acc, x.m);
    f29_mp_terms<P, K>(y.acc, y.m);
    if constexpr (K <= 8) {
        x.m[K] = ((uint32_t)x.acc * (P::INV & M29)) & M29;
        y.m[K] = ((uint32_t)y.acc * (P::INV & M29)) & M29;
        mad1_s(x.acc, x.m[K], P29<P, 0>::value);
        mad1_s(y.acc, y.m[K], P29<P, 0>::value);
    } else {
        x.r[K - 9] = (uint32_t)x.acc & M29;
        y.r[K - 9] = (uint32_t)y.acc & M29;
    }
    asm("v_lshrrev_b64 %0, 29, %0" : "+v"(x.acc));
    asm("v_lshrrev_b64 %0, 29, %0" : "+v"(y.acc));
}
template <class P, int S1, int S2> __device__ __forceinline__ void f29_pair_run(F29Job<P, S1>& x, F29Job<P, S2>& y)
{
    x.acc = 0;
    y.acc = 0;
#define BBG_X(K) f29_pair_column<P, S1, S2, K>(x, y);
    BBG_F29_COLUMNS(BBG_X)
#undef BBG_X
    x.r[8] = (uint32_t)x.acc;
    y.r[8] = (uint32_t)y.acc;
}
// r1 = a1 * b1, r2 = a2 * b2
template <class P> __device__ __forceinline__ void f29_mul2(const F29<P>& a1, const F29<P>& b1, const F29<P>& a2, const F29<P>& b2, F29<P>& r1, F29<P>& r2)
{
    F29Job<P, F29_MUL> x, y; // (no aggregate initialisation: it would zero the m digits, 18 v_mov per pair that nothing reads)
    x.a = a1.v, x.b = b1.v, x.r = r1.v;
    y.a = a2.v, y.b = b2.v, y.r = r2.v;
    f29_pair_run(x, y);
}
// r1 = a1^2, r2 = a2^2
template <class P> __device__ __forceinline__ void f29_sqr2(const F29<P>& a1, const F29<P>& a2, F29<P>& r1, F29<P>& r2)
{
    uint32_t d1[9], d2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d1[i] = a1.v[i] << 1;
        d2[i] = a2.v[i] << 1;
    }
    F29Job<P, F29_SQR> x, y;
    x.a = a1.v, x.b = d1, x.r = r1.v;
    y.a = a2.v, y.b = d2, y.r = r2.v;
    f29_pair_run(x, y);
}
// r1 = a * b - c * d (as f29_mul_sub2), r2 = e * f
template <class P>
__device__ __forceinline__ void f29_mul_sub2_mul(const F29<P>& a, const F29<P>& b, const F29<P>& c, const F29<P>& d, const F29<P>& e, const F29<P>& f,
                                                 F29<P>& r1, F29<P>& r2)
{
    F29<P> z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.v[i] = 0;
    const F29<P> nc = f29_sub<64, 30>(z, c);
    F29Job<P, F29_MULSUB2> x;
    F29Job<P, F29_MUL> y;
    x.a = a.v, x.b = b.v, x.c = nc.v, x.d = d.v, x.r = r1.v;
    y.a = e.v, y.b = f.v, y.r = r2.v;
    f29_pair_run(x, y);
}

} // namespace bbg
