// Lazily reduced 9 x 29-bit-limb Fr arithmetic (field29.hip.h) for POINTWISE polynomial identities -- the quotient widgets of quotient.hip --
// with every bound carried in the TYPE and checked by the compiler.
//
// Why.  A widget evaluates a fixed polynomial expression of ~15 loaded values per domain point: 20-50 products, most of them terms of a
// sum.  On 8 x 32-bit limbs (field.hip.h) every product is reduced on its own (136 mads + 136 carry instructions) and every addition
// carries, compares and corrects.  On 29-bit limbs a product is 81 + 81 mads without a carry instruction, a SUM of up to six products shares
// ONE reduction (the column accumulator has room for 63 limb products), and additions are nine v_add_u32.  What makes that usable outside a
// hand-checked kernel like the bucket accumulation is the bookkeeping: which operands may meet in a multiplier column, which multiple of
// p a subtraction has to add.  Here that bookkeeping is done by the type system:
//
//   W<C, VQ, LM>   C  = scale class: the limbs hold  x * 2^256 * 32^C  (C = 0: the device arrays' R-form re-limbed as it is; C = 1: the
//                       same words shifted by 5 bits = the R'-form of the 29-bit multiplier, R' = 2^261).  Both conversions are the
//                       limb split of a load (f29_split_limb), neither costs a multiplication.  The multiplier divides by R' = 32 R, so
//                       class(a b) = class(a) + class(b) - 1: an accumulated value of class 0 meets a FRESH operand loaded in class 1 and
//                       stays in class 0; two computed class-0 values meet after up() (a 5-bit limb shift) of one of them.
//                  VQ = bound of value / p in 1/64 units (rounded up).  A product leaves (sum a_t b_t) / R' + p < (sum Va Vb / 169 + 1) p
//                       (R' / p = 169.07).  A class-0 load of a coarse residue (< 2p) has V = 2, a class-1 load V = 64.
//                  LM = bound of the limbs below the top one (the top limb's bound follows from VQ: limbs are non-negative).
//
// mul / dot static_assert the column bound (sum over the terms of 9 La Lb, + 9 2^58 for m p, < 2^64), sub() picks the multiple of p and the
// limb raise (Spread29) from its subtrahend's bounds, finish() (n29_finish: exact signed pass against a row of the 32-row q p table) asserts
// V < 32 and returns the [0, 2p) words a device array holds.  An expression that could overflow for SOME input does not compile.
#pragma once
#include "ntt29.hip.h"

namespace bbg {
namespace w29 {

constexpr uint64_t RP_OVER_P = 169;                     // floor(2^261 / p)
constexpr uint64_t PTOP1 = (uint64_t)NTT29_P_TOP + 1;   // > p / 2^232
constexpr double TWO58 = 288230376151711744.0, TWO64 = 18446744073709551616.0;

constexpr int bitlen64(uint64_t x)
{
    int n = 0;
    while (x) {
        n++;
        x >>= 1;
    }
    return n;
}

template <int C, uint64_t VQ, uint64_t LM> struct W {
    Fr29 f;
    static constexpr int cls = C;
    static constexpr uint64_t vq = VQ, lm = LM;
    static constexpr uint64_t top = (VQ * PTOP1 + 63) / 64 + 2; // limb 8: top * 2^232 <= value < V p
    static constexpr uint64_t any = top > LM ? top : LM;
    static_assert(top < (1ull << 32) && LM < (1ull << 32), "w29: a limb would leave 32 bits");
};

// a coarse residue (< 2p) from a device array or a set-up block, in class C
template <int C> __device__ __forceinline__ W<C, (C ? 64 : 2) * 64, M29> ld(const Fr& x)
{
    static_assert(C == 0 || C == 1, "w29: class 0 or 1");
    return { f29_from_fe<FrP, C ? 5 : 0>(x) };
}

// one product of a dot().  Holds REFERENCES: write t(...) inside the dot(...) call, where temporaries such as ld<1>(x) live until the whole
// expression is done; a Term kept in a variable would outlive them.
template <class A, class B> struct Term {
    const A& a;
    const B& b;
    static constexpr int cls = A::cls + B::cls - 1;
    static constexpr double col = 9.0 * (double)A::any * (double)B::any;
    static constexpr uint64_t vv = A::vq * B::vq;
};
template <class A, class B> __device__ __forceinline__ Term<A, B> t(const A& a, const B& b) { return { a, b }; }

// sum of the terms' products with ONE Montgomery reduction
template <class T0, class... T> __device__ __forceinline__ auto dot(const T0& t0, const T&... ts)
{
    constexpr int C = T0::cls;
    static_assert(((T::cls == C) && ... && true), "w29::dot: terms of different scale classes");
    static_assert(C == 0 || C == 1, "w29::dot: the product would leave classes 0 / 1 (two computed class-0 values: up() one of them)");
    constexpr double col = (T0::col + ... + T::col) + 9.0 * TWO58 + 68719476736.0; // a b terms + m p terms + the carried-in column
    static_assert(col < TWO64 * 0.9999, "w29::dot: a multiplier column could overflow 64 bits (carry() an operand or split the sum)");
    constexpr uint64_t vv = (T0::vv + ... + T::vv);
    constexpr uint64_t VQ = (vv + 64 * RP_OVER_P - 1) / (64 * RP_OVER_P) + 64;
    W<C, VQ, M29> r;
    uint64_t acc = 0;
    uint32_t m[9];
#define BBG_X(K)                                                                                                                                \
    f29_ab_terms<K>(acc, t0.a.f.v, t0.b.f.v);                                                                                                   \
    (f29_ab_terms<K>(acc, ts.a.f.v, ts.b.f.v), ...);                                                                                            \
    f29_close_column<FrP, K>(acc, m, r.f.v);
    BBG_F29_COLUMNS(BBG_X)
#undef BBG_X
    r.f.v[8] = (uint32_t)acc;
    return r;
}
template <class A, class B> __device__ __forceinline__ auto mul(const A& a, const B& b) { return dot(t(a, b)); }

// a^2 of a class-1 value (45 + 81 mads)
template <class A> __device__ __forceinline__ auto sqr(const A& a)
{
    static_assert(A::cls == 1, "w29::sqr: only a class-1 value squares into a class (use mul(ld<1>(x), ld<0>(x)) for a class-0 result)");
    static_assert(A::any < (1ull << 31), "w29::sqr: the doubled operand would leave 32 bits");
    static_assert(9.0 * (double)A::any * (double)A::any + 9.0 * TWO58 + 68719476736.0 < TWO64 * 0.9999, "w29::sqr: column overflow");
    constexpr uint64_t VQ = (A::vq * A::vq + 64 * RP_OVER_P - 1) / (64 * RP_OVER_P) + 64;
    return W<1, VQ, M29>{ f29_sqr(a.f) };
}

template <class A, class B> __device__ __forceinline__ auto add(const A& a, const B& b)
{
    static_assert(A::cls == B::cls, "w29::add: different scale classes");
    return W<A::cls, A::vq + B::vq, A::lm + B::lm>{ f29_add(a.f, b.f) };
}
template <class A> __device__ __forceinline__ auto dbl(const A& a) { return add(a, a); }

// a - b + M p: M and the limb raise 2^E follow from b's bounds (Spread29: b limbs <= 2^E - 2, b's top limb <= that of M p minus 2^(E - 29))
template <class A, class B> __device__ __forceinline__ auto sub(const A& a, const B& b)
{
    static_assert(A::cls == B::cls, "w29::sub: different scale classes");
    constexpr int M = (int)((B::vq + 63) / 64) + 1;
    constexpr int E0 = bitlen64(B::lm + 1), E = E0 < 30 ? 30 : E0;
    static_assert(E <= 31, "w29::sub: subtrahend limbs too large (carry() it first)");
    static_assert(M <= 168, "w29::sub: subtrahend too large for a 9-limb multiple of p");
    static_assert((uint64_t)M * NTT29_P_TOP >= B::top + (1ull << (E - 29)), "w29::sub: top-limb margin");
    return W<A::cls, A::vq + 64ull * M, A::lm + (1ull << E) + (1ull << 29)>{ f29_sub<M, E>(a.f, b.f) };
}
// M p - b
template <class B> __device__ __forceinline__ auto neg(const B& b)
{
    W<B::cls, 0, 0> z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.f.v[i] = 0;
    return sub(z, b);
}

// one parallel carry pass: limbs below the top back under 2^29 + 8
template <class A> __device__ __forceinline__ auto carry(const A& a) { return W<A::cls, A::vq, M29 + 8>{ f29_carry(a.f) }; }

// class 0 -> class 1: the value times 32, a 5-bit shift across the limbs (exact for lazy limbs: a_i 32 = ((a_i << 5) mod 2^29) + 2^29 (a_i >> 24))
template <class A> __device__ __forceinline__ auto up(const A& a)
{
    static_assert(A::cls == 0, "w29::up: class 0 only");
    static_assert(A::top < (1ull << 27), "w29::up: the top limb would leave 32 bits");
    W<1, A::vq * 32, M29 + 256> r;
    r.f.v[0] = (a.f.v[0] << 5) & M29;
#pragma unroll
    for (int i = 1; i < 8; i++) r.f.v[i] = ((a.f.v[i] << 5) & M29) + (a.f.v[i - 1] >> 24);
    r.f.v[8] = (a.f.v[8] << 5) + (a.f.v[7] >> 24);
    return r;
}

// the [0, 2p) R-form words of a class-0 value below 32p (n29_finish: x - q p with the quotient estimate, exact limbs in one signed pass)
template <class A> __device__ __forceinline__ Fr finish(const A& a, const uint32_t* red)
{
    static_assert(A::cls == 0, "w29::finish: device arrays hold class 0");
    static_assert(A::vq <= 31 * 64, "w29::finish: the q p table has 32 rows");
    return n29_finish(a.f, red);
}

// the table finish() reads: NTT29_TABLE_WORDS words of LDS, filled by the first 32 threads of the block (a __syncthreads() follows at the caller)
__device__ __forceinline__ void fill_table(uint32_t* red)
{
    if (threadIdx.x < NTT29_RED_ROWS) ntt29_fill_reduce_table(red, (int)threadIdx.x);
}

} // namespace w29
} // namespace bbg
