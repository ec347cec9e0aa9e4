// Linear combinations and evaluations of coefficient arrays on lazily reduced 29-bit limbs (w29.hip.h) -- option "poly_limbs29" (default 1).
// Both are sums of products in which one operand is a per-launch CONSTANT (a transcript challenge, a power of the evaluation point): the
// 32-bit kernels reduce every product on its own (fe_mul + fe_add per term, ~300 instructions); here four terms share ONE reduction
// (4 x 81 + 90 mads) and the partial sums are nine v_add_u32, which moves both kernels from the multiplier to HBM.  Same [0, 2p) residues
// in and out.  Included by poly.hip after LinCombArgs / PolyScratch / block_sum.
#pragma once
#include "w29.hip.h"

namespace bbg {
namespace p29 {
using namespace w29;

// A constant c as a multiplier: the limbs of the CANONICAL R-form words of 32 c -- what ld<1>(c) denotes (c 2^256 32), without the factor 32 in
// its bound (ld<1> is a limb split of c's own words, V < 64; this is V < 1), so that dozens of terms fit one finish().  The host computes 32 c
// (five modular doublings: poly.hip times32); the kernels only split the words into limbs, once per block, and keep them in LDS.
using Mult = W<1, 64, M29>;
constexpr int MULT_ROW = 12; // LDS words per multiplier (nine used)
__device__ __forceinline__ void mult_store(uint32_t* dst, const Fr& c32)
{
    const Fr29 m = f29_from_fe<FrP, 0>(c32);
#pragma unroll
    for (int i = 0; i < 9; i++) dst[i] = m.v[i];
}
__device__ __forceinline__ Mult mult_load(const uint32_t* src)
{
    Mult m;
#pragma unroll
    for (int i = 0; i < 9; i++) m.f.v[i] = src[i];
    return m;
}
// a 4-term dot of coarse loads against multipliers: V < 1.1 (what w29::dot computes for it; asserted where the dots are)
constexpr uint64_t GROUP_VQ = (4ull * (2 * 64) * Mult::vq + 64 * RP_OVER_P - 1) / (64 * RP_OVER_P) + 64;
static_assert((LC_MAX / 4) * GROUP_VQ + 2 * 64 <= 31 * 64, "p29: LC_MAX terms and a base value must fit n29_finish");

#define BBG_P29_TERM(K) t(ld<0>(fe_load<FrP>(a.polys[k + (K)] + i)), mult_load(sc + MULT_ROW * (k + (K))))
// out[i] = base[i] + sum_k polys[k][i] * scalars[k] (k_poly_lincomb)
__global__ void __launch_bounds__(256) k_poly_lincomb29(LinCombArgs a, const Fr* __restrict__ base, Fr* out, size_t n)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    __shared__ uint32_t sc[LC_MAX * MULT_ROW];
    fill_table(red);
    for (int k = threadIdx.x >> 6; k < a.count; k += 4) { // wave-uniform k: the scalars (32 c_k, canonical) come through the scalar cache
        uint32_t m[9];
        mult_store(m, a.scalars[k]);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int j = 0; j < 9; j++) sc[MULT_ROW * k + j] = m[j];
        }
    }
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr29 acc = f29_from_fe<FrP, 0>(base ? fe_load<FrP>(base + i) : Fr::zero());
        int k = 0;
        for (; k + 4 <= a.count; k += 4) {
            const auto g = dot(BBG_P29_TERM(0), BBG_P29_TERM(1), BBG_P29_TERM(2), BBG_P29_TERM(3));
            static_assert(decltype(g)::cls == 0 && decltype(g)::vq == GROUP_VQ, "p29: the group bound");
            acc = f29_carry(f29_add(acc, g.f));
        }
        if (a.count - k == 3) acc = f29_carry(f29_add(acc, dot(BBG_P29_TERM(0), BBG_P29_TERM(1), BBG_P29_TERM(2)).f));
        else if (a.count - k == 2) acc = f29_carry(f29_add(acc, dot(BBG_P29_TERM(0), BBG_P29_TERM(1)).f));
        else if (a.count - k == 1) acc = f29_carry(f29_add(acc, dot(BBG_P29_TERM(0)).f));
        fe_store<FrP>(out + i, n29_finish(acc, red));
    }
}
#undef BBG_P29_TERM

// chunk_eval on 29-bit limbs: lane t sums c[base + t + 256 e] z^(256 e), e < 16, as four 4-term dots against the sixteen powers (multipliers in
// PolyScratch, their limbs in LDS) instead of a sixteen-step Horner chain of dependent products; then z^t and the block sum as before.
__device__ __forceinline__ Fr chunk_eval29(const Fr* __restrict__ c, size_t n, size_t base, const PolyScratch* ps, Fr* sm, uint32_t* zl, const uint32_t* red)
{
    static_assert(EV_CHUNK == 4096, "chunk_eval29: sixteen coefficients per lane");
    const int tid = threadIdx.x;
    if (tid < 16) mult_store(zl + MULT_ROW * tid, ps->zmult[tid]); // 32 z^(256 e), canonical (k_poly_pow2)
    __syncthreads();
    Fr29 acc;
#define BBG_P29_C(E) t(ld<0>(base + (size_t)(E) * 256 + tid < n ? fe_load<FrP>(c + base + (size_t)(E) * 256 + tid) : Fr::zero()), mult_load(zl + MULT_ROW * (E)))
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const auto d = dot(BBG_P29_C(4 * g), BBG_P29_C(4 * g + 1), BBG_P29_C(4 * g + 2), BBG_P29_C(4 * g + 3));
        acc = g ? f29_carry(f29_add(acc, d.f)) : d.f;
    }
#undef BBG_P29_C
    const W<0, 4 * GROUP_VQ, M29 + 8> s16{ acc };
    Fr s = finish(mul(s16, ld<1>(ps->ztid[tid])), red);
    return block_sum(s, sm); // z^base: the final kernels
}
__global__ void __launch_bounds__(256) k_eval_partial29(const Fr* __restrict__ c, size_t n, const PolyScratch* ps, Fr* partials)
{
    __shared__ Fr sm[128];
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    __shared__ uint32_t zl[16 * MULT_ROW];
    fill_table(red);
    const Fr s = chunk_eval29(c, n, (size_t)blockIdx.x * EV_CHUNK, ps, sm, zl, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_multi_eval_partial29(MultiEvalArgs a, const PolyScratch* ps, Fr* partials)
{
    __shared__ Fr sm[128];
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    __shared__ uint32_t zl[16 * MULT_ROW];
    const int k = blockIdx.y;
    const size_t base = (size_t)blockIdx.x * EV_CHUNK;
    if (base >= a.len[k]) { // uniform per block
        if (threadIdx.x == 0) partials[k * a.stride + blockIdx.x] = Fr::zero();
        return;
    }
    fill_table(red);
    const Fr s = chunk_eval29(a.poly[k], a.len[k], base, ps + a.point[k], sm, zl, red);
    if (threadIdx.x == 0) partials[k * a.stride + blockIdx.x] = s;
}

} // namespace p29
} // namespace bbg
