// BN254 G1 (y^2 = x^3 + 3) group arithmetic for gfx950, on top of field.hip.h.
//
// The reference works in Jacobian coordinates (ecc/groups/element_impl.hpp:70-441: self_dbl, operator+=(affine),
// operator+=(element)).  On the GPU the bucket sums use extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed addition is 8M + 2S instead of 7M + 4S, with no field inversion,
// and -- unlike the reference's affine trick (scalar_multiplication.cpp:305-340) -- no batch inversion pass.
// The group element computed is the same; only the boundary converts back to the reference's 96-byte Jacobian.
//
// Points at infinity: affine / Jacobian use the reference's convention, bit 63 of x.data[3] == bit 31 of x.v[7]
// (element_impl.hpp:497-516, affine_element_impl.hpp:74-93); XYZZ uses ZZ == 0.
#pragma once
#include "field.hip.h"

namespace bbg {

struct alignas(16) Affine {
    Fq x, y;
};
struct alignas(16) Jacobian {
    Fq x, y, z;
};
struct alignas(16) Xyzz {
    Fq x, y, zz, zzz;
};

__device__ __forceinline__ bool aff_is_inf(const Affine& p) { return (p.x.v[7] >> 31) != 0; }
__device__ __forceinline__ Affine aff_inf()
{
    Affine r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.x.v[7] = 0x80000000u;
    return r;
}
__device__ __forceinline__ Xyzz xyzz_inf()
{
    Xyzz r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    r.zz = Fq::zero();
    r.zzz = Fq::zero();
    return r;
}
__device__ __forceinline__ bool xyzz_is_inf(const Xyzz& p) { return p.zz.is_zero_raw(); }

__device__ __forceinline__ Affine aff_load(const void* p)
{
    Affine r;
    r.x = fe_load<FqP>(p);
    r.y = fe_load<FqP>(reinterpret_cast<const char*>(p) + 32);
    return r;
}
__device__ __forceinline__ void aff_store(void* p, const Affine& a)
{
    fe_store<FqP>(p, a.x);
    fe_store<FqP>(reinterpret_cast<char*>(p) + 32, a.y);
}
__device__ __forceinline__ Xyzz xyzz_load(const void* p)
{
    const char* c = reinterpret_cast<const char*>(p);
    Xyzz r;
    r.x = fe_load<FqP>(c);
    r.y = fe_load<FqP>(c + 32);
    r.zz = fe_load<FqP>(c + 64);
    r.zzz = fe_load<FqP>(c + 96);
    return r;
}
__device__ __forceinline__ void xyzz_store(void* p, const Xyzz& a)
{
    char* c = reinterpret_cast<char*>(p);
    fe_store<FqP>(c, a.x);
    fe_store<FqP>(c + 32, a.y);
    fe_store<FqP>(c + 64, a.zz);
    fe_store<FqP>(c + 96, a.zzz);
}

__device__ __forceinline__ Xyzz xyzz_from_affine(const Affine& p)
{
    Xyzz r;
    r.x = p.x;
    r.y = p.y;
    r.zz = Fq::one();
    r.zzz = Fq::one();
    return r;
}

// 2P for an affine P (mdbl-2008-s-1).  P must not be infinity; y = 0 cannot occur on BN254 G1 (odd prime order).
__device__ __forceinline__ Xyzz xyzz_dbl_affine(const Affine& p)
{
    Fq U = fe_dbl(p.y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(p.x, V);
    Fq xx = fe_sqr(p.x);
    Fq M = fe_add(fe_dbl(xx), xx);
    Xyzz r;
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_mul_sub2(M, fe_sub(S, r.x), W, p.y);
    r.zz = V;
    r.zzz = W;
    return r;
}
// 2P (dbl-2008-s-1), restating element::self_dbl (element_impl.hpp:70-139) in XYZZ form.
__device__ __forceinline__ Xyzz xyzz_dbl(const Xyzz& p)
{
    if (xyzz_is_inf(p)) return p;
    Fq U = fe_dbl(p.y);
    Fq V = fe_sqr(U);
    Fq W = fe_mul(U, V);
    Fq S = fe_mul(p.x, V);
    Fq xx = fe_sqr(p.x);
    Fq M = fe_add(fe_dbl(xx), xx);
    Xyzz r;
    r.x = fe_sub(fe_sqr(M), fe_dbl(S));
    r.y = fe_mul_sub2(M, fe_sub(S, r.x), W, p.y);
    r.zz = fe_mul(V, p.zz);
    r.zzz = fe_mul(W, p.zzz);
    return r;
}
// acc + P, P affine (madd-2008-s), complete: handles acc = inf, P = inf, P = +-acc like
// element::operator+=(affine_element) does (element_impl.hpp:243-330).
__device__ __forceinline__ Xyzz xyzz_madd(const Xyzz& a, const Affine& p)
{
    if (aff_is_inf(p)) return a;
    if (xyzz_is_inf(a)) return xyzz_from_affine(p);
    Fq U2 = fe_mul(p.x, a.zz);
    Fq S2 = fe_mul(p.y, a.zzz);
    Fq P = fe_sub(U2, a.x);
    Fq R = fe_sub(S2, a.y);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return xyzz_dbl_affine(p);
        return xyzz_inf();
    }
    Fq PP = fe_sqr(P);
    Fq PPP = fe_mul(P, PP);
    Fq Q = fe_mul(a.x, PP);
    Xyzz r;
    r.x = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    r.y = fe_mul_sub2(R, fe_sub(Q, r.x), a.y, PPP);
    r.zz = fe_mul(a.zz, PP);
    r.zzz = fe_mul(a.zzz, PPP);
    return r;
}
// a + b (add-2008-s), complete like element::operator+=(element) (element_impl.hpp:354-441).
__device__ __forceinline__ Xyzz xyzz_add(const Xyzz& a, const Xyzz& b)
{
    if (xyzz_is_inf(b)) return a;
    if (xyzz_is_inf(a)) return b;
    Fq U1 = fe_mul(a.x, b.zz);
    Fq U2 = fe_mul(b.x, a.zz);
    Fq S1 = fe_mul(a.y, b.zzz);
    Fq S2 = fe_mul(b.y, a.zzz);
    Fq P = fe_sub(U2, U1);
    Fq R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) return xyzz_dbl(a);
        return xyzz_inf();
    }
    Fq PP = fe_sqr(P);
    Fq PPP = fe_mul(P, PP);
    Fq Q = fe_mul(U1, PP);
    Xyzz r;
    r.x = fe_sub(fe_sub(fe_sqr(R), PPP), fe_dbl(Q));
    r.y = fe_mul_sub2(R, fe_sub(Q, r.x), S1, PPP);
    r.zz = fe_mul(fe_mul(a.zz, b.zz), PP);
    r.zzz = fe_mul(fe_mul(a.zzz, b.zzz), PPP);
    return r;
}
// XYZZ -> the reference's Jacobian (X', Y', Z') with x = X'/Z'^2, y = Y'/Z'^3: take Z' = ZZ*ZZZ.
__device__ __forceinline__ Jacobian xyzz_to_jacobian(const Xyzz& p)
{
    Jacobian r;
    if (xyzz_is_inf(p)) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        r.z = Fq::zero();
        r.x.v[7] = 0x80000000u;
        return r;
    }
    Fq z = fe_mul(p.zz, p.zzz);          // Z'
    Fq z2 = fe_sqr(z);                   // Z'^2 = ZZ^2 ZZZ^2
    Fq t = fe_mul(p.zz, fe_sqr(p.zzz));  // ZZ ZZZ^2
    r.x = fe_mul(p.x, t);                // x Z'^2 = X ZZ ZZZ^2
    r.y = fe_mul(fe_mul(p.y, z2), p.zz); // y Z'^3 = Y ZZ^3 ZZZ^2 = Y * Z'^2 * ZZ
    r.z = z;
    return r;
}
__device__ __forceinline__ Xyzz xyzz_from_jacobian(const Jacobian& p)
{
    Xyzz r;
    if ((p.x.v[7] >> 31) != 0) return xyzz_inf();
    r.x = p.x;
    r.y = p.y;
    r.zz = fe_sqr(p.z);
    r.zzz = fe_mul(r.zz, p.z);
    return r;
}
__device__ __forceinline__ Affine aff_neg_if(const Affine& p, bool neg)
{
    Affine r = p;
    Fq ny = fe_neg(p.y);
#pragma unroll
    for (int i = 0; i < 8; i++) r.y.v[i] = neg ? ny.v[i] : p.y.v[i];
    return r;
}

// a^(q-2) in Fq (field::invert, field_impl.hpp:323-329)
__device__ inline Fq fq_invert(const Fq& a)
{
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = FqP::MOD[i];
    e[0] -= 2; // 0xd87cfd47 - 2, no borrow
    Fq acc = Fq::one();
    for (int i = 255; i >= 0; i--) {
        acc = fe_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul(acc, a);
    }
    return acc;
}
// XYZZ -> canonical affine (one inversion): x = X/ZZ, y = Y/ZZZ
__device__ inline Affine xyzz_to_affine(const Xyzz& p)
{
    if (xyzz_is_inf(p)) return aff_inf();
    Fq iz = fq_invert(fe_mul(p.zz, p.zzz)); // 1/(ZZ*ZZZ)
    Affine r;
    r.x = fe_reduce_once(fe_mul(p.x, fe_mul(iz, p.zzz)));
    r.y = fe_reduce_once(fe_mul(p.y, fe_mul(iz, p.zz)));
    return r;
}

} // namespace bbg
