// BN254 Fr / Fq arithmetic for gfx950 (CDNA4).
//
// Restates the arithmetic of the reference's field<Params> template
// (barretenberg/src/aztec/ecc/fields/field_impl.hpp:34-157, generic bodies
// field_impl_generic.hpp:171-442) for a 32-bit-multiplier GPU: 8 x u32 limbs,
// little-endian, Montgomery form with R = 2^256 -- the SAME residues as the
// reference's 4 x u64 limbs, so device buffers are byte-identical to the host
// `fr` / `fq` arrays.  Like the reference ("small modulus" branches,
// field_impl_generic.hpp:213-271) values are kept coarsely reduced in [0, 2p);
// canonicalisation happens at the boundary (reduce_once()).
//
// All multiply work is v_mad_u64_u32 (32x32+64 -> 64); there is no MFMA use:
// this is 256-bit modular integer arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "addsub_chains.hip.h"
#include "mac_chains.hip.h"

namespace bbg {

// ---------------------------------------------------------------- parameters
// Moduli / constants: reference ecc/curves/bn254/fr.hpp:12-42, fq.hpp:11-41
// (re-derived with Python big-ints; 32-bit limb split of the same numbers).
struct FrP {
    static constexpr uint32_t MOD[8] = { 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                         0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u };
    static constexpr uint32_t MOD2[8] = { 0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                                          0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u };
    static constexpr uint32_t ONE[8] = { 0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                         0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u };
    static constexpr uint32_t R2[8] = { 0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                        0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u };
    static constexpr uint32_t INV = 0xefffffffu; // -p^-1 mod 2^32
    static constexpr uint32_t NEG2P[8] = { 0x1ffffffeu, 0x783c14d8u, 0x0c8d1eddu, 0xaf982f6fu,
                                           0xfcfd4f45u, 0x8f5f7492u, 0x3d9cbfacu, 0x9f37631au }; // 2^256 - 2p
    static constexpr uint32_t NEGP[8] = { 0x0fffffffu, 0xbc1e0a6cu, 0x86468f6eu, 0xd7cc17b7u,
                                          0x7e7ea7a2u, 0x47afba49u, 0x1ece5fd6u, 0xcf9bb18du };  // 2^256 - p
};
struct FqP {
    static constexpr uint32_t MOD[8] = { 0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                         0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u };
    static constexpr uint32_t MOD2[8] = { 0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u,
                                          0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u };
    static constexpr uint32_t ONE[8] = { 0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                         0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u };
    static constexpr uint32_t R2[8] = { 0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                        0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u };
    static constexpr uint32_t INV = 0xe4866389u;
    static constexpr uint32_t NEG2P[8] = { 0x4f060572u, 0x87bee7d2u, 0x2f1c6ae5u, 0xd0fd2addu,
                                           0xfcfd4f44u, 0x8f5f7492u, 0x3d9cbfacu, 0x9f37631au };
    static constexpr uint32_t NEGP[8] = { 0x278302b9u, 0xc3df73e9u, 0x978e3572u, 0x687e956eu,
                                          0x7e7ea7a2u, 0x47afba49u, 0x1ece5fd6u, 0xcf9bb18du };
};

// ------------------------------------------------------------------- element
template <class P> struct alignas(16) Fe {
    uint32_t v[8];

    __device__ __forceinline__ static Fe zero()
    {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0;
        return r;
    }
    __device__ __forceinline__ static Fe one()
    {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = P::ONE[i];
        return r;
    }
    __device__ __forceinline__ bool is_zero_raw() const
    {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i];
        return o == 0;
    }
};

// r = a - m, returns borrow (1 if a < m)
template <class P> __device__ __forceinline__ uint32_t sub_limbs(uint32_t* r, const uint32_t* a, const uint32_t* m)
{
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a[i] - m[i] - br;
        r[i] = (uint32_t)d;
        br = (d >> 32) & 1u;
    }
    return (uint32_t)br;
}

// The four carry-chain primitives are single asm statements (addsub_chains.hip.h): hipcc lowers the portable
// `uint64_t carry` idiom to ~5 VALU instructions per limb; the native v_add_co / v_addc_co chain is 1 per limb
// (measured on the NTT pass kernel: 1950 -> see profiles/ for the instruction mix).
#define BBG_K8(A) A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7]

// coarse add: inputs in [0,2p) -> output in [0,2p)   (field_impl_generic.hpp:196-234)
template <class P> __device__ __forceinline__ Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b)
{
    Fe<P> r = a;
    asm_add_coarse<BBG_K8(P::NEG2P)>(r.v, b.v);
    return r;
}

// coarse sub: inputs in [0,2p) -> output in [0,2p)   (field_impl_generic.hpp:254-271)
template <class P> __device__ __forceinline__ Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b)
{
    Fe<P> r = a;
    asm_sub_coarse<BBG_K8(P::MOD2)>(r.v, b.v);
    return r;
}

// 2p - a (a in [0,2p]); maps 0 -> 2p which reduce_once() canonicalises (field_impl.hpp:148-157)
template <class P> __device__ __forceinline__ Fe<P> fe_neg(const Fe<P>& a)
{
    Fe<P> r = a;
    asm_neg<BBG_K8(P::MOD2)>(r.v);
    return r;
}

template <class P> __device__ __forceinline__ Fe<P> fe_dbl(const Fe<P>& a) { return fe_add(a, a); }

// one conditional subtraction of p: any 256-bit a -> a - p if a >= p   (field_impl.hpp:100-112 reduce_once)
template <class P> __device__ __forceinline__ Fe<P> fe_reduce_once(const Fe<P>& a)
{
    Fe<P> r = a;
    asm_reduce_once<BBG_K8(P::NEGP)>(r.v);
    return r;
}

// full canonicalisation of any 256-bit value < 4p
template <class P> __device__ __forceinline__ Fe<P> fe_canon(const Fe<P>& a)
{
    return fe_reduce_once(fe_reduce_once(fe_reduce_once(a)));
}

// Montgomery product a*b*R^-1.  For inputs < 2p (even a < 4p with b < p) the result is < 2p
// with no final subtraction because 4p < 2^256 -- the same bound the reference relies on
// (field_impl_generic.hpp:392-442 montgomery_mul, "coarse" form).
//
// Formulation: product scanning (FIPS).  Column k accumulates a_i*b_(k-i) and m_i*p_(k-i) into a
// 96-bit accumulator {c2:acc}; v_mad_u64_u32 adds the 64-bit product into acc and its carry-out
// is counted into c2 by one v_addc -- 2 VALU ops per limb product, which measured 138 Gmul/s on
// MI355X against 104 Gmul/s for the compiler's CIOS lowering (bench_micro/mulbench.hip;
// v_mad_u64_u32 issues at half rate, ~28 T/s chip-wide, so 136 mads alone bound a multiplication
// at ~206 G/s).
#define BBG_MAC(acc, c2, x, y)                                                                                       \
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"                                     \
        : "+v"(acc), "+v"(c2)                                                                                        \
        : "v"(x), "v"(y)                                                                                             \
        : "vcc")
template <class P> __device__ __forceinline__ Fe<P> fe_mul_v1(const Fe<P>& a, const Fe<P>& b)
{
    uint64_t acc = 0;
    uint32_t c2 = 0;
    uint32_t m[8];
    Fe<P> r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) BBG_MAC(acc, c2, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) BBG_MAC(acc, c2, m[i], P::MOD[k - i]);
        m[k] = (uint32_t)acc * P::INV;
        BBG_MAC(acc, c2, m[k], P::MOD[0]);
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
        c2 = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) BBG_MAC(acc, c2, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) BBG_MAC(acc, c2, m[i], P::MOD[k - i]);
        r.v[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)c2 << 32);
        c2 = 0;
    }
    return r;
}

// ---- column helpers: N products x[i] * y[-i] chained in ONE asm statement (mac_chains.hip.h)
template <int N> __device__ __forceinline__ void mac_col_v(uint64_t& acc, uint32_t& c2, const uint32_t* x, const uint32_t* y)
{
    if constexpr (N == 1) mac1_v(acc, c2, x[0], y[0]);
    else if constexpr (N == 2) mac2_v(acc, c2, x[0], y[0], x[1], y[-1]);
    else if constexpr (N == 3) mac3_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2]);
    else if constexpr (N == 4) mac4_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3]);
    else if constexpr (N == 5) mac5_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4]);
    else if constexpr (N == 6) mac6_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5]);
    else if constexpr (N == 7) mac7_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6]);
    else if constexpr (N == 8)
        mac8_v(acc, c2, x[0], y[0], x[1], y[-1], x[2], y[-2], x[3], y[-3], x[4], y[-4], x[5], y[-5], x[6], y[-6], x[7], y[-7]);
}
// same with the second factors = modulus limbs MOD[J], MOD[J-1], ... held in SGPRs
template <class P, int N, int J> __device__ __forceinline__ void mac_col_mod(uint64_t& acc, uint32_t& c2, const uint32_t* x)
{
    if constexpr (N == 1) mac1_s(acc, c2, x[0], P::MOD[J]);
    else if constexpr (N == 2) mac2_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1]);
    else if constexpr (N == 3) mac3_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2]);
    else if constexpr (N == 4) mac4_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2], x[3], P::MOD[J - 3]);
    else if constexpr (N == 5)
        mac5_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2], x[3], P::MOD[J - 3], x[4], P::MOD[J - 4]);
    else if constexpr (N == 6)
        mac6_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2], x[3], P::MOD[J - 3], x[4], P::MOD[J - 4], x[5],
               P::MOD[J - 5]);
    else if constexpr (N == 7)
        mac7_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2], x[3], P::MOD[J - 3], x[4], P::MOD[J - 4], x[5],
               P::MOD[J - 5], x[6], P::MOD[J - 6]);
    else if constexpr (N == 8)
        mac8_s(acc, c2, x[0], P::MOD[J], x[1], P::MOD[J - 1], x[2], P::MOD[J - 2], x[3], P::MOD[J - 3], x[4], P::MOD[J - 4], x[5],
               P::MOD[J - 5], x[6], P::MOD[J - 6], x[7], P::MOD[J - 7]);
}
template <class P, int K> __device__ __forceinline__ void fips_low_column(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b,
                                                                         uint32_t* m)
{
    mac_col_v<K + 1>(acc, c2, a, b + K);
    if constexpr (K > 0) mac_col_mod<P, K, K>(acc, c2, m);
    m[K] = (uint32_t)acc * P::INV;
    mac1_s(acc, c2, m[K], P::MOD[0]);
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
}
template <class P, int K> __device__ __forceinline__ void fips_high_column(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b,
                                                                          const uint32_t* m, uint32_t* r)
{
    if constexpr (K < 15) {
        mac_col_v<15 - K>(acc, c2, a + (K - 7), b + 7);
        mac_col_mod<P, 15 - K, 7>(acc, c2, m + (K - 7));
    }
    r[K - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
}
// Shipping multiplier: same FIPS schedule as fe_mul_v1, but each column is at most three asm statements and the
// modulus limbs are SGPR operands.
template <class P> __device__ __forceinline__ Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b)
{
    uint64_t acc = 0;
    uint32_t c2 = 0;
    uint32_t m[8];
    Fe<P> r;
    fips_low_column<P, 0>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 1>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 2>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 3>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 4>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 5>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 6>(acc, c2, a.v, b.v, m);
    fips_low_column<P, 7>(acc, c2, a.v, b.v, m);
    fips_high_column<P, 8>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 9>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 10>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 11>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 12>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 13>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 14>(acc, c2, a.v, b.v, m, r.v);
    fips_high_column<P, 15>(acc, c2, a.v, b.v, m, r.v);
    return r;
}

// a*b - c*d with ONE Montgomery reduction (the Y3 = R(Q - X3) - Y1*PPP shape of every XYZZ group law): both products
// are accumulated column by column before the m*p terms, saving the 64 + 8 multiply-adds of a second reduction.
// c is negated first (2p - c <= 2p), so T = a*b + (2p - c)*d <= 8p^2 and (T + m*p)/R < p*(8p/R + 1) = 2.52p < 2^256:
// one conditional subtraction of 2p restores the coarse range [0,2p).
template <class P, int K> __device__ __forceinline__ void fips2_low_column(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b,
                                                                          const uint32_t* c, const uint32_t* d, uint32_t* m)
{
    mac_col_v<K + 1>(acc, c2, a, b + K);
    mac_col_v<K + 1>(acc, c2, c, d + K);
    if constexpr (K > 0) mac_col_mod<P, K, K>(acc, c2, m);
    m[K] = (uint32_t)acc * P::INV;
    mac1_s(acc, c2, m[K], P::MOD[0]);
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
}
template <class P, int K> __device__ __forceinline__ void fips2_high_column(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b,
                                                                           const uint32_t* c, const uint32_t* d, const uint32_t* m, uint32_t* r)
{
    if constexpr (K < 15) {
        mac_col_v<15 - K>(acc, c2, a + (K - 7), b + 7);
        mac_col_v<15 - K>(acc, c2, c + (K - 7), d + 7);
        mac_col_mod<P, 15 - K, 7>(acc, c2, m + (K - 7));
    }
    r[K - 8] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)c2 << 32);
    c2 = 0;
}
template <class P> __device__ __forceinline__ Fe<P> fe_mul_sub2(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d)
{
    const Fe<P> nc = fe_neg(c);
    uint64_t acc = 0;
    uint32_t c2 = 0;
    uint32_t m[8];
    Fe<P> r;
    fips2_low_column<P, 0>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 1>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 2>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 3>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 4>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 5>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 6>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_low_column<P, 7>(acc, c2, a.v, b.v, nc.v, d.v, m);
    fips2_high_column<P, 8>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 9>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 10>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 11>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 12>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 13>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 14>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    fips2_high_column<P, 15>(acc, c2, a.v, b.v, nc.v, d.v, m, r.v);
    asm_reduce_once<BBG_K8(P::NEG2P)>(r.v); // r - 2p if r >= 2p
    return r;
}

// Reference formulation kept for cross-checking the asm path in tests (CIOS over 32-bit limbs;
// running value T < 3p < 2^256 after each row, so 9 words suffice).
template <class P> __device__ __forceinline__ Fe<P> fe_mul_cios(const Fe<P>& a, const Fe<P>& b)
{
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t x = (uint64_t)a.v[j] * bi + t[j] + c;
            t[j] = (uint32_t)x;
            c = x >> 32;
        }
        t[8] = (uint32_t)c;
        const uint32_t m = t[0] * P::INV;
        uint64_t x = (uint64_t)m * P::MOD[0] + t[0];
        c = x >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            x = (uint64_t)m * P::MOD[j] + t[j] + c;
            t[j - 1] = (uint32_t)x;
            c = x >> 32;
        }
        t[7] = t[8] + (uint32_t)c;
    }
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}

template <class P> __device__ __forceinline__ Fe<P> fe_sqr(const Fe<P>& a) { return fe_mul(a, a); }

template <class P> __device__ __forceinline__ Fe<P> fe_from_mont(const Fe<P>& a)
{
    Fe<P> o = Fe<P>::zero();
    o.v[0] = 1;
    return fe_reduce_once(fe_mul(a, o));
}

template <class P> __device__ __forceinline__ Fe<P> fe_to_mont(const Fe<P>& a)
{
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = P::R2[i];
    return fe_mul(a, r2);
}

template <class P> __device__ __forceinline__ bool fe_eq(const Fe<P>& a, const Fe<P>& b)
{
    Fe<P> x = fe_reduce_once(a), y = fe_reduce_once(b);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= x.v[i] ^ y.v[i];
    return o == 0;
}

template <class P> __device__ __forceinline__ bool fe_is_zero(const Fe<P>& a)
{
    return fe_reduce_once(a).is_zero_raw();
}

// 16-byte vector loads/stores: two dwordx4 per element.
template <class P> __device__ __forceinline__ Fe<P> fe_load(const void* p)
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fe<P> r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <class P> __device__ __forceinline__ void fe_store(void* p, const Fe<P>& a)
{
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// ---- inversion on ONE lane's dependency chain: the binary extended Euclid of Kaliski's almost-inverse (256-bit shifts, additions,
// subtractions; no multiplication), then one multiplication by a power of two.  With A = a R the Montgomery residue, the loop keeps
//   A r = -u 2^k,   A s = v 2^k   (mod p),   u, v <= p,   r, s < 2p
// (u = p, v = A, r = 0, s = 1, k = 0 at the start): trailing zeros of u are shifted out while s is doubled as often (k += c), the same for v
// and r; with both odd the larger loses the smaller (u -= v, r += s, or v -= u, s += r).  v = 0 leaves u = 1 and r = -A^-1 2^k, 253 <= k <= 507;
// A^-1 2^k 2^(512 - k) = A^-1 R^2 = a^-1 R is the residue of the inverse.  ~180 subtraction steps of ~100 instructions against the 254
// squarings + ~126 multiplications (~100 000 instructions) of a^(p-2): the chain of the permutation grand product's single inversion -- part
// of every proof's round 3, exposed when the polynomials are short -- drops from 0.32 ms to a third.  0 -> 0 like the power.
// UNIFORM = every lane that calls holds the SAME input and calls in uniform control flow: the chain runs on the scalar unit (inline
// s_addc_u32 / s_subb_u32 / s_lshr_b32 blocks -- left to itself the compiler selects vector instructions for carry chains and funnel shifts
// even when every operand is uniform) and leaves the vector unit to whatever else is resident.  The result is canonical (< p).
struct U256 {
    uint32_t w[8];
};
__device__ __forceinline__ bool u256_is_zero(const U256& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3] | a.w[4] | a.w[5] | a.w[6] | a.w[7]) == 0; }
template <bool UNIFORM> __device__ __forceinline__ bool x256_sub(U256& out, const U256& a, const U256& b) // out = a - b mod 2^256, returns the borrow
{
    U256 r;
    uint32_t bo = 0;
    if constexpr (UNIFORM) {
        asm("s_sub_u32 %0, %9, %17\n\ts_subb_u32 %1, %10, %18\n\ts_subb_u32 %2, %11, %19\n\ts_subb_u32 %3, %12, %20\n\ts_subb_u32 %4, %13, %21\n\ts_subb_u32 %5, %14, %22\n\ts_subb_u32 %6, %15, %23\n\ts_subb_u32 %7, %16, %24\n\ts_cselect_b32 %8, 1, 0"
            : "=&s"(r.w[0]), "=&s"(r.w[1]), "=&s"(r.w[2]), "=&s"(r.w[3]), "=&s"(r.w[4]), "=&s"(r.w[5]), "=&s"(r.w[6]), "=&s"(r.w[7]), "=s"(bo)
            : "s"(a.w[0]), "s"(a.w[1]), "s"(a.w[2]), "s"(a.w[3]), "s"(a.w[4]), "s"(a.w[5]), "s"(a.w[6]), "s"(a.w[7]), "s"(b.w[0]), "s"(b.w[1]), "s"(b.w[2]), "s"(b.w[3]), "s"(b.w[4]), "s"(b.w[5]), "s"(b.w[6]), "s"(b.w[7])
            : "scc");
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) r.w[i] = __builtin_subc(a.w[i], b.w[i], bo, &bo);
    }
    out = r;
    return bo != 0;
}
template <bool UNIFORM> __device__ __forceinline__ void x256_add(U256& out, const U256& a, const U256& b)
{
    U256 r;
    if constexpr (UNIFORM) {
        asm("s_add_u32 %0, %8, %16\n\ts_addc_u32 %1, %9, %17\n\ts_addc_u32 %2, %10, %18\n\ts_addc_u32 %3, %11, %19\n\ts_addc_u32 %4, %12, %20\n\ts_addc_u32 %5, %13, %21\n\ts_addc_u32 %6, %14, %22\n\ts_addc_u32 %7, %15, %23"
            : "=&s"(r.w[0]), "=&s"(r.w[1]), "=&s"(r.w[2]), "=&s"(r.w[3]), "=&s"(r.w[4]), "=&s"(r.w[5]), "=&s"(r.w[6]), "=&s"(r.w[7])
            : "s"(a.w[0]), "s"(a.w[1]), "s"(a.w[2]), "s"(a.w[3]), "s"(a.w[4]), "s"(a.w[5]), "s"(a.w[6]), "s"(a.w[7]), "s"(b.w[0]), "s"(b.w[1]), "s"(b.w[2]), "s"(b.w[3]), "s"(b.w[4]), "s"(b.w[5]), "s"(b.w[6]), "s"(b.w[7])
            : "scc");
    } else {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) r.w[i] = __builtin_addc(a.w[i], b.w[i], c, &c);
    }
    out = r;
}
template <bool UNIFORM> __device__ __forceinline__ void x256_shr(U256& a, uint32_t c) // 1 <= c <= 31
{
    U256 r;
    if constexpr (UNIFORM) {
        uint32_t t;
        asm("s_lshr_b32 %0, %9, %17\n\ts_lshl_b32 %8, %10, %18\n\ts_or_b32 %0, %0, %8\n\ts_lshr_b32 %1, %10, %17\n\ts_lshl_b32 %8, %11, %18\n\ts_or_b32 %1, %1, %8\n\ts_lshr_b32 %2, %11, %17\n\ts_lshl_b32 %8, %12, %18\n\ts_or_b32 %2, %2, %8\n\ts_lshr_b32 %3, %12, %17\n\ts_lshl_b32 %8, %13, %18\n\ts_or_b32 %3, %3, %8\n\ts_lshr_b32 %4, %13, %17\n\ts_lshl_b32 %8, %14, %18\n\ts_or_b32 %4, %4, %8\n\ts_lshr_b32 %5, %14, %17\n\ts_lshl_b32 %8, %15, %18\n\ts_or_b32 %5, %5, %8\n\ts_lshr_b32 %6, %15, %17\n\ts_lshl_b32 %8, %16, %18\n\ts_or_b32 %6, %6, %8\n\ts_lshr_b32 %7, %16, %17"
            : "=&s"(r.w[0]), "=&s"(r.w[1]), "=&s"(r.w[2]), "=&s"(r.w[3]), "=&s"(r.w[4]), "=&s"(r.w[5]), "=&s"(r.w[6]), "=&s"(r.w[7]), "=&s"(t)
            : "s"(a.w[0]), "s"(a.w[1]), "s"(a.w[2]), "s"(a.w[3]), "s"(a.w[4]), "s"(a.w[5]), "s"(a.w[6]), "s"(a.w[7]), "s"(c), "s"(32u - c)
            : "scc");
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) r.w[i] = (a.w[i] >> c) | (a.w[i + 1] << (32u - c));
        r.w[7] = a.w[7] >> c;
    }
    a = r;
}
template <bool UNIFORM> __device__ __forceinline__ void x256_shl(U256& a, uint32_t c) // 1 <= c <= 31
{
    U256 r;
    if constexpr (UNIFORM) {
        uint32_t t;
        asm("s_lshl_b32 %7, %16, %17\n\ts_lshr_b32 %8, %15, %18\n\ts_or_b32 %7, %7, %8\n\ts_lshl_b32 %6, %15, %17\n\ts_lshr_b32 %8, %14, %18\n\ts_or_b32 %6, %6, %8\n\ts_lshl_b32 %5, %14, %17\n\ts_lshr_b32 %8, %13, %18\n\ts_or_b32 %5, %5, %8\n\ts_lshl_b32 %4, %13, %17\n\ts_lshr_b32 %8, %12, %18\n\ts_or_b32 %4, %4, %8\n\ts_lshl_b32 %3, %12, %17\n\ts_lshr_b32 %8, %11, %18\n\ts_or_b32 %3, %3, %8\n\ts_lshl_b32 %2, %11, %17\n\ts_lshr_b32 %8, %10, %18\n\ts_or_b32 %2, %2, %8\n\ts_lshl_b32 %1, %10, %17\n\ts_lshr_b32 %8, %9, %18\n\ts_or_b32 %1, %1, %8\n\ts_lshl_b32 %0, %9, %17"
            : "=&s"(r.w[0]), "=&s"(r.w[1]), "=&s"(r.w[2]), "=&s"(r.w[3]), "=&s"(r.w[4]), "=&s"(r.w[5]), "=&s"(r.w[6]), "=&s"(r.w[7]), "=&s"(t)
            : "s"(a.w[0]), "s"(a.w[1]), "s"(a.w[2]), "s"(a.w[3]), "s"(a.w[4]), "s"(a.w[5]), "s"(a.w[6]), "s"(a.w[7]), "s"(c), "s"(32u - c)
            : "scc");
    } else {
#pragma unroll
        for (int i = 7; i > 0; i--) r.w[i] = (a.w[i] << c) | (a.w[i - 1] >> (32u - c));
        r.w[0] = a.w[0] << c;
    }
    a = r;
}
__device__ __forceinline__ uint32_t u256_low_zeros(const U256& a) // trailing zero bits, at most 31 per step
{
    const uint32_t c = a.w[0] ? (uint32_t)__builtin_ctz(a.w[0]) : 31u;
    return c > 31u ? 31u : c;
}
template <class P, bool UNIFORM = false> __device__ inline Fe<P> fe_inverse_gcd(const Fe<P>& a_in)
{
    Fe<P> a = fe_reduce_once(fe_reduce_once(a_in)); // [0, 4p) -> canonical
    if constexpr (UNIFORM) {
#pragma unroll
        for (int i = 0; i < 8; i++) a.v[i] = __builtin_amdgcn_readfirstlane(a.v[i]);
    }
    U256 p, u, v, r, s;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        p.w[i] = P::MOD[i];
        v.w[i] = a.v[i];
        r.w[i] = 0;
        s.w[i] = 0;
    }
    if (u256_is_zero(v)) return Fe<P>::zero();
    u = p;
    s.w[0] = 1;
    uint32_t k = 0;
    for (;;) {
        if (!(u.w[0] & 1)) { // (u is never zero)
            const uint32_t c = u256_low_zeros(u);
            x256_shr<UNIFORM>(u, c);
            x256_shl<UNIFORM>(s, c);
            k += c;
            continue;
        }
        if (!(v.w[0] & 1)) { // (v is not zero here: the loop left when it became zero)
            const uint32_t c = u256_low_zeros(v);
            x256_shr<UNIFORM>(v, c);
            x256_shl<UNIFORM>(r, c);
            k += c;
            continue;
        }
        U256 d;
        const bool below = x256_sub<UNIFORM>(d, u, v); // u < v
        if (!below && !u256_is_zero(d)) {             // u > v
            u = d;
            x256_add<UNIFORM>(r, r, s);
        } else {                                      // v >= u (equal only at the end: both 1)
            (void)x256_sub<UNIFORM>(v, v, u);
            x256_add<UNIFORM>(s, s, r);
            if (u256_is_zero(v)) break;
        }
    }
    U256 t;
    if (!x256_sub<UNIFORM>(t, r, p)) r = t; // r < 2p -> < p
    (void)x256_sub<UNIFORM>(r, p, r);       // -A^-1 2^k -> A^-1 2^k
    Fe<P> x;
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = r.w[i];
    // times 2^(512 - k), 5 <= 512 - k <= 259: Montgomery products with the residues of 2^c, c <= 253 (2^c < p as plain words)
    for (uint32_t e = 512u - k; e > 0;) {
        const uint32_t c = e > 253u ? 253u : e;
        Fe<P> pw = Fe<P>::zero();
#pragma unroll
        for (int i = 0; i < 8; i++) pw.v[i] = (uint32_t)i == (c >> 5) ? 1u << (c & 31u) : 0u;
        x = fe_reduce_once(fe_mul(x, fe_to_mont(pw)));
        e -= c;
    }
    return fe_reduce_once(x);
}

using Fr = Fe<FrP>;
using Fq = Fe<FqP>;

} // namespace bbg
