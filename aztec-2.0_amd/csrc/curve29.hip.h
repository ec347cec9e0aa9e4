// The mixed addition of the bucket accumulation on 29-bit limbs (field29.hip.h): acc += P with acc in XYZZ coordinates
// (madd-2008-s, the same 8M + 2S as curve.hip.h xyzz_madd; element::operator+=(affine_element), element_impl.hpp:243-330).
//
// All four coordinates live in R'-form (x * 2^261 mod p) as lazily reduced 9 x 29-bit values:
//   at entry   X, Y < 32p, ZZ, ZZZ < 1.4p, limbs < 2^29 + 8    (a run starts from a table point: canonical, shifted by 5 bits)
//   at exit    X < 20.4p, Y < 7.7p, ZZ, ZZZ < 1.1p, same limb bound
// (bounds: a product is < a * b / 2^261 + p and p / 2^261 = 2^-7.4; the inline comments carry them through).
// The special cases of a complete addition (P = +-acc) are NOT tested per addition: both make PP = (U2 - X1)^2 a multiple of p, so
// ZZ becomes a multiple of p and stays one under every later addition -- the caller tests ZZ once at the end of a run
// (xyzz29_finish) and, if it is, queues the run's bucket: k_redo recomputes that bucket with the complete 32-bit formulas (msm_kernels.hip.h).
#pragma once
#include "curve.hip.h"
#include "field29.hip.h"

namespace bbg {

using Fq29 = F29<FqP>;
struct Xyzz29 {
    Fq29 x, y, zz, zzz;
};

// constants in radix 2^29
__device__ __forceinline__ Fq29 fq29_const(const uint32_t (&w)[8])
{
    Fq29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.v[j] = k29_limb(w, j);
    return r;
}
// 2^261 mod p as 8 x u32 (python: pow(2, 261, p))
__device__ constexpr uint32_t FQ_R261[8] = { 0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u, 0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u };

// A table point as the accumulation uses it: R'-form by the 5-bit split, the sign of the digit applied as p - y.  Table points are canonical
// (k_precompute_tables), so both coordinates are < 32p with exact 29-bit limbs.
struct Aff29 {
    Fq29 x, y;
};
__device__ __forceinline__ Aff29 aff29_from_table(const Affine& p, bool neg)
{
    Fq ny = p.y;
    asm_neg<BBG_K8(FqP::MOD)>(ny.v); // p - y (y != 0 on BN254 G1)
    Fq y;
#pragma unroll
    for (int i = 0; i < 8; i++) y.v[i] = neg ? ny.v[i] : p.y.v[i];
    Aff29 r;
    r.x = f29_from_fe<FqP, 5>(p.x);
    r.y = f29_from_fe<FqP, 5>(y);
    return r;
}
// first point of a run
__device__ __forceinline__ Xyzz29 xyzz29_from_affine(const Aff29& p)
{
    Xyzz29 r;
    r.x = p.x;
    r.y = p.y;
    r.zz = fq29_const(FQ_R261);
    r.zzz = r.zz;
    return r;
}

__device__ __forceinline__ Xyzz29 xyzz29_madd(const Xyzz29& a, const Aff29& p)
{
    // products in independent pairs (f29_mul2 ...: columns interleaved, see field29.hip.h)
    Fq29 U2, S2, PP, RR, PPP, Q;
    f29_mul2(p.x, a.zz, p.y, a.zzz, U2, S2);             // < 32 * 1.4 p * 2^-7.4 + p = 1.3p
    const Fq29 P = f29_carry(f29_sub<34>(U2, a.x));      // U2 - X1 + 34p < 35.3p
    const Fq29 R = f29_carry(f29_sub<34>(S2, a.y));      // < 35.3p
    f29_sqr2(P, R, PP, RR);                              // < 35.3^2 * 2^-7.4 p + p = 8.4p
    f29_mul2(P, PP, a.x, PP, PPP, Q);                    // PPP < 2.8p; Q < 32 * 8.4 * 2^-7.4 p + p = 2.6p
    Xyzz29 r;
    // X3 = R^2 - PPP - 2Q + 12p: the subtrahend (< 8p) has limbs < 3 * 2^29, so this spread constant raises every limb by 2^31
    {
        Fq29 s;
#pragma unroll
        for (int i = 0; i < 9; i++) s.v[i] = PPP.v[i] + 2 * Q.v[i];
        r.x = f29_carry(f29_sub<12, 31>(RR, s));         // < 20.4p
    }
    const Fq29 T = f29_carry(f29_sub<24>(Q, r.x));       // Q - X3 + 24p < 26.6p
    f29_mul_sub2_mul(R, T, a.y, PPP, a.zzz, PPP, r.y, r.zzz); // Y3 < (35.3 * 26.6 + 64 * 2.8) * 2^-7.4 p + p < 7.7p; ZZZ3 = ZZZ1 PPP
    r.zz = f29_mul(a.zz, PP);                            // < 1.4 * 8.4 * 2^-7.4 p + p = 1.1p
    return r;
}

// End of a run: back to the 8 x u32 R-form (x R' / 32 = x R: f29_div32_to_fe), coarse < 2p.
// Returns false when ZZ is a multiple of p: an addition inside the run hit P = +-acc, the values are meaningless.
__device__ __forceinline__ bool xyzz29_finish(const Xyzz29& a, Xyzz& out, const uint32_t* div32_table)
{
    out.x = f29_div32_to_fe(a.x, div32_table);   // < (20.4 + 31) p / 32
    out.y = f29_div32_to_fe(a.y, div32_table);
    out.zz = f29_div32_to_fe(a.zz, div32_table);
    out.zzz = f29_div32_to_fe(a.zzz, div32_table);
    return !fe_is_zero(out.zz);
}

} // namespace bbg
