// The radix-8 NTT step on LAZILY REDUCED 9 x 29-bit limbs (field29.hip.h) -- the arithmetic of k_ntt_pass29 (ntt_pass8.hip.h).
//
// Why.  A radix-8 decimation-in-frequency step on 8 x 32-bit limbs (p8_butterfly + the step twiddles) costs 12 Montgomery products of
// 136 v_mad_u64_u32 + 136 v_addc_co_u32 and 24 butterflies whose add and sub each carry, compare and conditionally correct (8 x u32 with
// carries: ~24 instructions per operation).  On 29-bit limbs a product is 162 mads and NO carry instruction (field29.hip.h: 168-177 against
// 139 G products/s), an addition is 9 v_add_u32, a subtraction 9 v_sub + 9 v_add of a constant -- no carries, no comparisons: values are
// allowed to grow and only the twelve products (which reduce whatever they are given) and one table-driven reduction of the single
// un-multiplied output bring them back.
//
// Representation.  A device array holds x R mod p (R = 2^256), coarsely reduced (< 2p), as 8 x u32.  Re-limbing those 256 bits into
// 9 x 29-bit limbs WITHOUT the 5-bit shift of the MSM's conversion gives the integer x R = (x / 32) R' with R' = 2^261: the whole transform is
// carried out on the values x / 32 in R'-Montgomery form -- a uniform scaling that a linear transform hands through, so the outputs are
// again y R, i.e. already what the array must hold; no conversion multiplication on either side.  Twiddles multiply as w R' mod p: the
// small per-radix tables exist in that form (NttDomain::tw_radix29, built with one multiplication by 32 per entry), the big per-element
// tables (inter-pass twiddles, coset factors) are the R-form ones shifted on the fly by 5 bits (w R << 5 = w R' as an integer < 64 p).
//
// Bounds (V = value / p; L = largest limb).  Products: (a b + m p) / R' < a b / R' + p, p / R' = 2^-7.4 = 1 / 169, limbs < 2^29 (field29.hip.h).
// f29_mul needs 9 La Lb + 9 2^58 < 2^64: Lb < 2^29 (a table value) allows La < 2^31 + 2^29.  f29_sub<K> needs b < K p (top-limb margin: V_b + 1 <= K)
// and b limbs <= 2^30 - 2.  "carried" = after f29_carry: limbs < 2^29 + 8.  Every bound below is asserted on a Python big-integer model
// of exactly these operations (tests/test_ntt29_model.py).
#pragma once
#include "field29.hip.h"
#include "field29c.hip.h"

namespace bbg {

using Fr29 = F29<FrP>;

// ---- the multiplier of the TABLE twiddles (r5): a per-kernel choice, template argument SH of the steps below.
//   SH = 1, 2  : every product by a per-radix table value -- the butterfly's own w8 powers and the step twiddles, twelve of the 13.5 products per
//               eight elements and step -- is the constant-operand product of field29c.hip.h (143 mads, no m digits): a table row holds w and
//               wq = floor(w 2^261 / p).  The same residue with exact limbs but a looser value bound: below (2 + V / 169) p where Montgomery
//               leaves (1 + V / 169) p, so a subtraction whose subtrahend is a product adds one more multiple of p (KP / KPP) and the values
//               downstream grow accordingly; every bound is restated in the comments and asserted on the big-integer model
//               (tests/test_ntt29_model.py, both variants).  Costs 9 more registers per live twiddle: the kernels that run two waves per SIMD
//               anyway take it (log-radix >= 9: ntt_pass29.hip.h p29_shoup), measured -4 % at 2^20 (profiles/r05_ntt_attempts.txt).
//   SH = 0     : Montgomery products against w R' (round 4): the kernels that fit three waves per SIMD (log-radix <= 8).
constexpr int NTT29_TW_ROW = C29_ROW; // words per row of a per-radix table, either format (9 limbs of w R' in the first 12 words, or w and wq in 18)
// SH: 0 = Montgomery, 1 = constant-operand product, 2 = the same with the butterfly's own (wave-uniform) multipliers as SGPR operands (no VGPRs for
// them: the variant for kernels that must stay within 168 VGPRs; slower than 1 where registers do not matter: 0.1092 vs 0.1073 ms at 2^20)
template <int SH> struct N29M {
    using Tw = C29<FrP>;
    static constexpr int KP = 4;  // a - b + KP p for b a product (V < 2.16)
    static constexpr int KPP = 6; // ... for b a sum of two products (V < 4.1)
    static __device__ __forceinline__ void mul2(Fr29& a, const Tw& wa, Fr29& b, const Tw& wb) { f29_mulc2(a, wa, b, wb, a, b); }
    static __device__ __forceinline__ void mul(Fr29& a, const Tw& w) { a = f29_mulc(a, w); }
    // u = the multiplier is the same for every lane of the wave
    static __device__ __forceinline__ void mul2u(Fr29& a, const Tw& wa, Fr29& b, const Tw& wb) { f29_mulc2<SH == 2, SH == 2>(a, wa, b, wb, a, b); }
    static __device__ __forceinline__ void mulu(Fr29& a, const Tw& w) { a = f29_mulc<SH == 2>(a, w); }
};
template <> struct N29M<0> {
    using Tw = Fr29;
    static constexpr int KP = 3, KPP = 4;
    static __device__ __forceinline__ void mul2(Fr29& a, const Tw& wa, Fr29& b, const Tw& wb) { f29_mul2(a, wa, b, wb, a, b); }
    static __device__ __forceinline__ void mul(Fr29& a, const Tw& w) { a = f29_mul(a, w); }
    static __device__ __forceinline__ void mul2u(Fr29& a, const Tw& wa, Fr29& b, const Tw& wb) { f29_mul2(a, wa, b, wb, a, b); }
    static __device__ __forceinline__ void mulu(Fr29& a, const Tw& w) { a = f29_mul(a, w); }
};

// ---- reduction of a value < 32 p to < 3 p without a multiplication: estimate q = floor(x / p) from the top limb (never too large, at
// most 1 too small), go one lower (so that the difference keeps a top limb of its own: the row's borrowed limbs need it) and add row
// q' of a table holding -q' p with every limb below the top raised by 2^30 (the raise borrowed from the limb above: Spread29 with q = 0).
constexpr int NTT29_RED_ROWS = 32, NTT29_RED_ROW = 12; // 12 words per row: three ds_read_b128
constexpr int NTT29_TABLE_WORDS = 2 * NTT29_RED_ROWS * NTT29_RED_ROW; // the borrowed rows (ntt29_reduce), then the plain rows q p (n29_finish)
constexpr uint32_t NTT29_P_TOP = (FrP::MOD[7] >> 8);                            // p >> 232 (22 bits)
constexpr uint32_t NTT29_INV_TOP = (uint32_t)((1ull << 32) / (NTT29_P_TOP + 1)); // floor(2^32 / (p_top + 1))
__device__ __forceinline__ void ntt29_fill_reduce_table(uint32_t* tbl, int k) // one row per calling thread, k < 32
{
    uint64_t carry = 0;
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)FrP::MOD[i] * (uint32_t)k + carry;
        w[i] = (uint32_t)x;
        carry = x >> 32;
    }
    w[8] = (uint32_t)carry;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int b = 29 * j, i = b >> 5, o = b & 31;
        const uint64_t lo = w[i], hi = i + 1 < 9 ? w[i + 1] : 0;
        const uint32_t kp = (uint32_t)(((lo | (hi << 32)) >> o) & M29);
        const uint32_t zero_spread = j == 0 ? (1u << 30) : (j < 8 ? (1u << 30) - 2u : 0u - 2u); // value 0: 2^30 borrowed limb by limb
        // row 0 is all zeros (x < 2p stays as it is: its top limb may be too small to lend); rows k >= 1 are used for x >= (k + 1) p only,
        // whose top limb exceeds that of k p by p's own: top limb -2 - kp (mod 2^32) added to a larger one
        tbl[k * NTT29_RED_ROW + j] = k == 0 ? 0u : zero_spread - kp;
        tbl[(NTT29_RED_ROWS + k) * NTT29_RED_ROW + j] = kp; // plain k p, exact limbs: the way out of a pass subtracts it with signed limbs
    }
}
// x: carried or not (limbs < 2^32 - 2^30), value < 32 p.  Result: carried, value in [p, 3p) for x >= 2p, x itself below.
__device__ __forceinline__ Fr29 ntt29_reduce(const Fr29& x, const uint32_t* tbl)
{
    const Fr29 c = f29_carry(x);
    const uint32_t q = __umulhi(c.v[8], NTT29_INV_TOP);
    const uint32_t k = q > 1u ? q - 1u : 0u;
    const uint4* row = reinterpret_cast<const uint4*>(tbl + k * NTT29_RED_ROW);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2];
    Fr29 t;
    t.v[0] = c.v[0] + r0.x; t.v[1] = c.v[1] + r0.y; t.v[2] = c.v[2] + r0.z; t.v[3] = c.v[3] + r0.w;
    t.v[4] = c.v[4] + r1.x; t.v[5] = c.v[5] + r1.y; t.v[6] = c.v[6] + r1.z; t.v[7] = c.v[7] + r1.w;
    t.v[8] = c.v[8] + r2.x;
    return f29_carry(t);
}

// ---- butterflies.  KB = the multiple of p added by the subtraction: must exceed V_b by one.
template <int KB, int E = 30> __device__ __forceinline__ void n29_bfly(Fr29& a, Fr29& b) // (a, b) <- (a + b, a - b + KB p); b limbs <= 2^E - 2
{
    const Fr29 u = f29_add(a, b);
    b = f29_sub<KB, E>(a, b);
    a = u;
}
__device__ __forceinline__ void n29_carry(Fr29& a) { a = f29_carry(a); }

// S = 3: the radix-8 butterfly of p8_butterfly<3> (same pairs, same twiddle places) followed by the seven step twiddles tw[1..7] (tw[j] is
// the multiplier of register j; pass nullptr-like `have_tw = false` for a step without them) and the reduction of register 0.
//   in : x[j] carried, V < 3 (a reduced or freshly loaded value), w1 = w8, w2 = w8^2 = w4, w3 = w8^3 in reduced R'-form (< p, exact limbs)
//   out: x[0] carried, V < 3; x[1..7] products, V < 1.2 (with step twiddles) or as listed below (without)
// `tw(j)` delivers the step twiddle of register j when the product is about to use it (a table load for the pass kernel: seven multipliers
// held at once are 63 registers the kernel does not have at three waves per SIMD)
// Bounds with SH (a product leaves V < 2 + V_in / 169 <= 2.16, exact limbs): level 2 (4,6) d = a - b + 4p V < 11, u V < 9.04;
// (5,7) u V < 4.08 (L < 2^30), d (+4p) V < 6.04; products of level 2 V < 2.08; level 3 (2,3) d (+4p) V < 17, u V < 15.1; (4,5) a V < 9.04,
// b V < 4.08: d (+6p) V < 15.04, u V < 13.1; (6,7) a V < 11: d (+4p) V < 15, u V < 13.04; (0,1) unchanged (24 / 25).  Step twiddle products V < 2.15.
template <int SH, bool HAVE_TW, class TW>
__device__ __forceinline__ void n29_step8(Fr29 (&x)[8], const typename N29M<SH>::Tw& w1, const typename N29M<SH>::Tw& w2, const typename N29M<SH>::Tw& w3,
                                          TW tw, const uint32_t* red)
{
    using M = N29M<SH>;
    using Tw29 = typename M::Tw;
    // level 1: inputs V < 3 carried.  u: V < 6, L < 2^30 + 16.  d = a - b + 4p: V < 7, L < 2^31 + 8.  NOTHING is carried here (r4b): the
    // sums go into level 2 as they are (its subtractions take subtrahend limbs up to 2^31 - 2: E = 31), x[4] = d04 likewise.
    n29_bfly<4>(x[0], x[4]);
    n29_bfly<4>(x[1], x[5]);
    n29_bfly<4>(x[2], x[6]);
    n29_bfly<4>(x[3], x[7]);
    M::mul2u(x[5], w1, x[6], w2); // V < 7/169 + 1 = 1.05 (Shoup: 2.05), L < 2^29
    M::mulu(x[7], w3);
    // level 2.  (0,2), (1,3): a, b V < 6, L < 2^30 + 16: u V < 12, L < 2^31 + 32; d = a - b + 7p (E = 31) V < 13, L < 2^30 + 16 + 2^31 + 2^29.
    // (4,6): a V < 7, L < 2^31 + 8, b V < 1.05 exact: u V < 8.05, L < 2^31 + 2^29 + 8; d = a - b + 3p V < 10, L < 2^31 + 8 + 2^30 + 2^29.
    // (5,7): a, b V < 1.05 exact: u V < 2.1 (L < 2^30), d = a - b + 3p V < 4.05.
    n29_bfly<7, 31>(x[0], x[2]);
    n29_bfly<7, 31>(x[1], x[3]);
    n29_bfly<M::KP>(x[4], x[6]);
    n29_bfly<M::KP>(x[5], x[7]);
    n29_carry(x[3]);                          // a multiplication's operand: limbs back below 2^29 + 8
    M::mul2u(x[3], w2, x[7], w2); // V < 13/169 + 1 = 1.08 ; V < 1.03 (Shoup: 2.08 / 2.04)
    n29_carry(x[0]); n29_carry(x[1]);         // V < 12
    n29_carry(x[2]);                          // V < 13
    n29_carry(x[4]);                          // V < 8.05
    n29_carry(x[6]);                          // V < 10
    // (x[5]: V < 2.1, L < 2^30 - 1: a sum of two exact values -- within f29_sub's limb bound as it is)
    // level 3.  (0,1): V < 12 each: u V < 24, d = a - b + 13p V < 25.  (2,3): a V < 13, b V < 1.08: u V < 14.1, d (+3p) V < 16.
    // (4,5): a V < 8.05, b V < 2.1: u V < 10.2, d (+4p) V < 12.05.  (6,7): a V < 10, b V < 1.03: u V < 11.03, d (+3p) V < 13.
    n29_bfly<13>(x[0], x[1]);
    n29_bfly<M::KP>(x[2], x[3]);
    n29_bfly<M::KPP>(x[4], x[5]);
    n29_bfly<M::KP>(x[6], x[7]);
    if constexpr (HAVE_TW) {
        // the step twiddles: operands V < 25, L < 2^31 + 8 (a difference of carried values) -> products V < 25/169 + 1 = 1.15 (Shoup: 2.15)
        {
            const Tw29 t1 = tw(1), t2 = tw(2);
            M::mul2(x[1], t1, x[2], t2);
        }
        {
            const Tw29 t3 = tw(3), t4 = tw(4);
            M::mul2(x[3], t3, x[4], t4);
        }
        {
            const Tw29 t5 = tw(5), t6 = tw(6);
            M::mul2(x[5], t5, x[6], t6);
        }
        M::mul(x[7], tw(7));
        x[0] = ntt29_reduce(x[0], red); // V < 24 -> < 3
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = ntt29_reduce(x[j], red); // every register leaves below 3p, carried
    }
}

// The LAST step of a pass whose log-radix is not a multiple of 3 transforms two bits or one (p8_butterfly<2> / <1>): no step twiddles, the
// outputs leave for the pass's final multiplication / conversion.  in: x[j] carried, V < 3.
//   S = 2: radix-4 on (b1 b0) -- pairs (0,2), (1,3) w4, (4,6), (5,7) w4, then (0,1), (2,3), (4,5), (6,7).  out: V < 13, L < 2^31 + 8.
//   S = 1: pairs (0,1), (2,3), (4,5), (6,7).                                                              out: V < 7,  L < 2^31 + 8.
template <int SH> __device__ __forceinline__ void n29_step4(Fr29 (&x)[8], const typename N29M<SH>::Tw& w2)
{
    using M = N29M<SH>;
    // level A: u V < 6, d = a - b + 4p V < 7
    n29_bfly<4>(x[0], x[2]);
    n29_bfly<4>(x[1], x[3]);
    n29_bfly<4>(x[4], x[6]);
    n29_bfly<4>(x[5], x[7]);
    M::mul2u(x[3], w2, x[7], w2); // V < 7/169 + 1 = 1.05 (Shoup: 2.05)
    n29_carry(x[0]); n29_carry(x[1]); n29_carry(x[4]); n29_carry(x[5]); // V < 6
    n29_carry(x[2]); n29_carry(x[6]);                                   // V < 7
    // level B: (0,1), (4,5): a, b V < 6: u V < 12, d (+7p) V < 13.  (2,3), (6,7): a V < 7, b V < 1.05 exact: u V < 8.05, d (+3p) V < 10
    // (Shoup: b V < 2.05: u V < 9.05, d (+4p) V < 11).
    n29_bfly<7>(x[0], x[1]);
    n29_bfly<M::KP>(x[2], x[3]);
    n29_bfly<7>(x[4], x[5]);
    n29_bfly<M::KP>(x[6], x[7]);
}
__device__ __forceinline__ void n29_step2(Fr29 (&x)[8])
{
    n29_bfly<4>(x[0], x[1]);
    n29_bfly<4>(x[2], x[3]);
    n29_bfly<4>(x[4], x[5]);
    n29_bfly<4>(x[6], x[7]);
}
// A radix-8 step WITHOUT step twiddles (the last step of a pass whose log-radix is a multiple of 3): n29_step8<false> reduces every register;
// when a final multiplication follows, the reduction can wait for it: out V < 25, L < 2^31 + 8 (see n29_step8's level 3).
template <int SH>
__device__ __forceinline__ void n29_step8_raw(Fr29 (&x)[8], const typename N29M<SH>::Tw& w1, const typename N29M<SH>::Tw& w2, const typename N29M<SH>::Tw& w3)
{
    using M = N29M<SH>;
    n29_bfly<4>(x[0], x[4]);
    n29_bfly<4>(x[1], x[5]);
    n29_bfly<4>(x[2], x[6]);
    n29_bfly<4>(x[3], x[7]);
    M::mul2u(x[5], w1, x[6], w2);
    M::mulu(x[7], w3);
    n29_bfly<7, 31>(x[0], x[2]);
    n29_bfly<7, 31>(x[1], x[3]);
    n29_bfly<M::KP>(x[4], x[6]);
    n29_bfly<M::KP>(x[5], x[7]);
    n29_carry(x[3]);
    M::mul2u(x[3], w2, x[7], w2);
    n29_carry(x[0]); n29_carry(x[1]);
    n29_carry(x[2]);
    n29_carry(x[4]);
    n29_carry(x[6]);
    n29_bfly<13>(x[0], x[1]);
    n29_bfly<M::KP>(x[2], x[3]);
    n29_bfly<M::KPP>(x[4], x[5]);
    n29_bfly<M::KP>(x[6], x[7]);
}

// the second half of n29_step8<SH, true> on its own: the seven step twiddles and the reduction of register 0, applied to n29_step8_raw's outputs
// (V < 25, L < 2^31 + 8) -- for the pass kernel's steps in which some waves have unit twiddles and skip this half (ntt_pass29.hip.h)
template <int SH, class TW> __device__ __forceinline__ void n29_step8_twiddles(Fr29 (&x)[8], TW tw, const uint32_t* red)
{
    using M = N29M<SH>;
    using Tw29 = typename M::Tw;
    {
        const Tw29 t1 = tw(1), t2 = tw(2);
        M::mul2(x[1], t1, x[2], t2);
    }
    {
        const Tw29 t3 = tw(3), t4 = tw(4);
        M::mul2(x[3], t3, x[4], t4);
    }
    {
        const Tw29 t5 = tw(5), t6 = tw(6);
        M::mul2(x[5], t5, x[6], t6);
    }
    M::mul(x[7], tw(7));
    x[0] = ntt29_reduce(x[0], red); // V < 24 -> < 3
}

// ---- the way out of a pass.  x: V < 25, L < 2^31 + 8 (any output of the last step).  With a multiplier (inter-pass twiddle / post-scale table
// entry, the R-form words of the table shifted by 5 bits: w R' as an integer < 64 p, exact limbs): product V < 25 * 64 / 169 + 1 = 10.5, then the
// table reduction (V < 3), exact limbs, the 8 words, one conditional subtraction -> the coarse [0, 2p) residue the device arrays hold.
// x - q p with q = the quotient estimate itself (x / p - 1 < q <= x / p): the result is in [0, 2p) -- what a device array holds -- and is produced
// with EXACT limbs by one sequential pass over signed limb differences (a limb of x - q p may be negative; the value is not), so neither a
// second carry pass nor a conditional subtraction follows; the 8 words are a join of the limbs.
__device__ __forceinline__ Fr n29_finish(const Fr29& x, const uint32_t* red)
{
    const Fr29 c = f29_carry(x);
    const uint32_t q = __umulhi(c.v[8], NTT29_INV_TOP);
    const uint4* row = reinterpret_cast<const uint4*>(red + (NTT29_RED_ROWS + q) * NTT29_RED_ROW);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2];
    const uint32_t qp[9] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x };
    Fr29 t;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int32_t v = (int32_t)(c.v[i] - qp[i]) + carry; // |difference| < 2^30: no wrap
        t.v[i] = (uint32_t)v & M29;
        carry = v >> 29; // arithmetic shift: borrows travel upwards as -1
    }
    t.v[8] = (uint32_t)((int32_t)(c.v[8] - qp[8]) + carry);
    Fr r;
    r.v[0] = f29_join_word<0>(t.v);
    r.v[1] = f29_join_word<1>(t.v);
    r.v[2] = f29_join_word<2>(t.v);
    r.v[3] = f29_join_word<3>(t.v);
    r.v[4] = f29_join_word<4>(t.v);
    r.v[5] = f29_join_word<5>(t.v);
    r.v[6] = f29_join_word<6>(t.v);
    r.v[7] = f29_join_word<7>(t.v);
    return r;
}
__device__ __forceinline__ Fr n29_finish_mul(const Fr29& x, const Fr& mult_rform, const uint32_t* red)
{
    return n29_finish(f29_mul(x, f29_from_fe<FrP, 5>(mult_rform)), red);
}

} // namespace bbg
