// Polynomial helpers of the prover that sit between the NTTs and the MSMs, on device-resident
// coefficient arrays.  Reference: barretenberg/src/aztec/polynomials/polynomial_arithmetic.cpp
//   add / sub / mul                      :486-505   pointwise over a domain                       (HBM-bound: 96 B per element)
//   evaluate                             :507-538   sum_i c_i z^i (Horner in per-thread slices)
//   compute_kate_opening_coefficients    :727-750   W(X) = (F(X) - F(z)) / (X - z), returns F(z)
//   divide_by_pseudo_vanishing_polynomial:628-725   pointwise division by Z*_H on the coset of the target domain
// All results are the same field elements the reference computes (compared on canonical values).
#include "bbg_internal.h"

#include <cstring>
#include "ntt_consts.hip.h"

namespace bbg {

static int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// ---------------------------------------------------------------------------------------------- pointwise
template <int OP> __global__ void __launch_bounds__(256) k_poly_binop(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* r, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Fr x = fe_load<FrP>(a + i), y = fe_load<FrP>(b + i);
        Fr z;
        if (OP == 0) z = fe_add(x, y);
        else if (OP == 1) z = fe_sub(x, y);
        else z = fe_mul(x, y);
        fe_store<FrP>(r + i, z);
    }
}
int poly_binop(int op, const void* a, const void* b, void* r, size_t n, hipStream_t st)
{
    if (op < 0 || op > 2) { set_error("bbg_poly_op: op must be 0 (add), 1 (sub) or 2 (mul)"); return BBG_E_INVALID; }
    if (n == 0) return BBG_OK;
    int grid = grid_for(n, 256);
    if (grid > 256 * 16) grid = 256 * 16; // grid-stride beyond 16 blocks per CU
    if (op == 0) hipLaunchKernelGGL(k_poly_binop<0>, dim3(grid), dim3(256), 0, st, (const Fr*)a, (const Fr*)b, (Fr*)r, n);
    else if (op == 1) hipLaunchKernelGGL(k_poly_binop<1>, dim3(grid), dim3(256), 0, st, (const Fr*)a, (const Fr*)b, (Fr*)r, n);
    else hipLaunchKernelGGL(k_poly_binop<2>, dim3(grid), dim3(256), 0, st, (const Fr*)a, (const Fr*)b, (Fr*)r, n);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// out[i] = base[i] + sum_k polys[k][i] * scalars[k]   (base may be null = 0; out may alias base)
// KateCommitmentScheme::batch_open's accumulation of the opening polynomials (kate_commitment_scheme.cpp:216-226): ~25
// polynomials x one transcript challenge each.  One product per term against 32 B of traffic: balanced between the
// multiplier and HBM.
constexpr int LC_MAX = 32;
struct LinCombArgs {
    const Fr* polys[LC_MAX];
    Fr scalars[LC_MAX];
    int count;
};
__global__ void __launch_bounds__(256) k_poly_lincomb(LinCombArgs a, const Fr* __restrict__ base, Fr* out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr acc = base ? fe_load<FrP>(base + i) : Fr::zero();
        for (int k = 0; k < a.count; k++) acc = fe_add(acc, fe_mul(fe_load<FrP>(a.polys[k] + i), a.scalars[k]));
        fe_store<FrP>(out + i, acc);
    }
}
// (poly_lincomb, the host side, follows the kernels of poly29.hip.h below)

// ---------------------------------------------------------------------------------------------- shared pieces
constexpr int PV_E = 16;     // consecutive evaluations per thread of the division by Z*_H
// Consecutive coefficients per thread of the Kate quotient, 2^LOG_E: 16 from 2^19 coefficients up, 8 from 2^17, 4 below -- a 2^16-coefficient
// polynomial in slices of 16 is 64 waves for 1 024 SIMDs, each with a chain of 16 + 16 dependent products (r5: rounds 6 of a 2^16-gate proof
// 0.747 -> 0.697 ms with slices of 4; at 2^20 the shorter slices LOSE, 92 -> 110 / 158 us per quotient: twice / four times the scan work)
static int kate_log_e(size_t n) { return n >= ((size_t)1 << 19) ? 4 : n >= ((size_t)1 << 17) ? 3 : 2; }
constexpr int EV_CHUNK = 4096; // coefficients per block of the evaluation kernels (256 threads x 16, lane-interleaved)
constexpr int MEV_MAX = 32;    // polynomials per multi-evaluation call
struct PolyScratch {
    Fr z;          // evaluation point
    Fr pow2z[48];  // z^(2^b)
    Fr result;     // F(z)
    Fr tmp[8];
    Fr ztid[256];  // z^t, t < 256
    Fr zmult[16];  // 32 z^(256 e), e < 16, canonical: the multipliers of chunk_eval29 (poly29.hip.h)
};
constexpr int DPV_MAX_EXT = 16, DPV_MAX_CUT = 8;
struct DpvConsts {
    Fr inv_sub[DPV_MAX_EXT]; // 1 / ((g w_ext^j)^n - 1)
    Fr numer[DPV_MAX_CUT];   // -w_src^-(k+1)
};
// context scratch: two evaluation points (zeta, zeta * w), MEV_MAX results, then the partial sums
struct PolyHeader {
    PolyScratch ps[2];
    Fr results[MEV_MAX];
};
static int poly_scratch(bbg_ctx* ctx, size_t partials, PolyHeader** hdr, Fr** part)
{
    const size_t head = (sizeof(PolyHeader) + 255) / 256 * 256;
    int rc = ensure_buffer(&ctx->poly_scratch, &ctx->poly_scratch_bytes, head + (partials + 2) * sizeof(Fr));
    if (rc) return rc;
    *hdr = (PolyHeader*)ctx->poly_scratch;
    if (part) *part = (Fr*)((char*)ctx->poly_scratch + head);
    return BBG_OK;
}

// ---- the evaluation point's powers z^(2^b): a chain of dependent squarings.  On the device that chain is ONE lane's latency -- 45 us per
// evaluation point, six points per proof (round 5's zeta and zeta w, r(zeta), two Kate quotients): 0.27 ms of a proof that takes 3-5 ms at
// 2^12 .. 2^16 gates (profile of round 4).  On a host core it is 48 Montgomery squarings of 4 x u64 limbs: a few microseconds.  So the host
// computes the table (hostfr below: CIOS with unsigned __int128, field_impl_generic.hpp:392-442 restated for this one purpose) and hands
// it to the kernel BY VALUE (1.5 KB of kernel arguments: no pageable-memory copy whose source the caller could reuse before it ran).
namespace hostfr {
struct H {
    uint64_t v[4];
};
static const uint64_t MOD[4] = { ((uint64_t)FrP::MOD[1] << 32) | FrP::MOD[0], ((uint64_t)FrP::MOD[3] << 32) | FrP::MOD[2],
                                 ((uint64_t)FrP::MOD[5] << 32) | FrP::MOD[4], ((uint64_t)FrP::MOD[7] << 32) | FrP::MOD[6] };
static const uint64_t INV64 = 0xc2e1f593efffffffULL; // -p^-1 mod 2^64 (fr.hpp:42)
static_assert((uint32_t)INV64 == FrP::INV, "the 64-bit Montgomery constant extends the 32-bit one the device uses");
static bool geq_mod(const H& a)
{
    for (int i = 3; i >= 0; i--)
        if (a.v[i] != MOD[i]) return a.v[i] > MOD[i];
    return true;
}
static H sub_mod(const H& a)
{
    H r;
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned __int128 d = (unsigned __int128)a.v[i] - MOD[i] - borrow;
        r.v[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
    return r;
}
static H canon(H a) // any 256-bit representative -> [0, p)
{
    while (geq_mod(a)) a = sub_mod(a);
    return a;
}
static H mul(const H& a, const H& b) // a b / 2^256 mod p, canonical; a, b < 2^256 with a b < 2^256 p (one canonical operand suffices)
{
    uint64_t t[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (unsigned __int128)a.v[j] * b.v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV64;
        c = ((unsigned __int128)m * MOD[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (unsigned __int128)m * MOD[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    H r = { { t[0], t[1], t[2], t[3] } };
    if (t[4] || geq_mod(r)) r = sub_mod(r); // t < 2p
    return canon(r);
}
} // namespace hostfr

struct Pow2Arg {
    Fr z;
    Fr pow2z[48];
    Fr zmult[16];
};
// 32 a mod p, canonical, of a canonical a: five modular doublings (a < p < 2^254: 2 a fits the words)
static hostfr::H times32(hostfr::H a)
{
    for (int d = 0; d < 5; d++) {
        for (int i = 3; i > 0; i--) a.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 63);
        a.v[0] <<= 1;
        a = hostfr::canon(a);
    }
    return a;
}
static Pow2Arg host_pow2(const uint64_t* z_limbs, const uint64_t* mul_root) // mul_root (canonical, Montgomery form) or null
{
    hostfr::H a;
    memcpy(&a, z_limbs, 32);
    a = hostfr::canon(a);
    if (mul_root) {
        hostfr::H r;
        memcpy(&r, mul_root, 32);
        a = hostfr::mul(a, hostfr::canon(r));
    }
    Pow2Arg t;
    memcpy(&t.z, &a, 32);
    hostfr::H z256 = a;
    for (int i = 0; i < 48; i++) {
        memcpy(&t.pow2z[i], &a, 32);
        if (i == 8) z256 = a;
        a = hostfr::mul(a, a);
    }
    // 32 z^(256 e): e = 0 is 32 in Montgomery form = 32 * (2^256 mod p); FrP::ONE holds the latter
    hostfr::H one;
    memcpy(&one, FrP::ONE, 32);
    hostfr::H m = times32(hostfr::canon(one));
    for (int e = 0; e < 16; e++) {
        memcpy(&t.zmult[e], &m, 32);
        m = hostfr::mul(m, z256);
    }
    return t;
}
__global__ void __launch_bounds__(256) k_poly_pow2(PolyScratch* s, const Pow2Arg t)
{
    if (threadIdx.x == 0) s->z = t.z;
    if (threadIdx.x < 48) s->pow2z[threadIdx.x] = t.pow2z[threadIdx.x];
    if (threadIdx.x < 16) s->zmult[threadIdx.x] = t.zmult[threadIdx.x];
    s->ztid[threadIdx.x] = fe_reduce_once(pow_from_table(t.pow2z, (uint64_t)threadIdx.x));
}
// LDS tree sum of one field element per thread (256 threads); result valid in thread 0
__device__ Fr block_sum(Fr v, Fr* sm)
{
    const int tid = threadIdx.x;
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (tid >= stride && tid < 2 * stride) sm[tid - stride] = v;
        __syncthreads();
        if (tid < stride) v = fe_add(v, sm[tid]);
        __syncthreads();
    }
    return v;
}
// S = sum_{e < E} c[i0 + e] z^e  (Horner from the top of the slice; coefficients beyond n count as zero)
template <int E> __device__ __forceinline__ Fr slice_horner(const Fr* __restrict__ c, size_t i0, size_t n, const Fr& z)
{
    Fr s = Fr::zero();
#pragma unroll 4
    for (int e = E - 1; e >= 0; e--) {
        s = fe_mul(s, z);
        if (i0 + e < n) s = fe_add(s, fe_load<FrP>(c + i0 + e));
    }
    return s;
}

// ---------------------------------------------------------------------------------------------- evaluate
// A block owns EV_CHUNK consecutive coefficients; lane t takes c[base + t + 256 e], e < 16 -- every load instruction of a wave
// reads 2 KiB of consecutive memory -- and sums them by Horner in z^256; the lane results are weighted by z^t (table) and
// tree-summed; the block results are weighted by z^base where they are added up (k_eval_final / k_multi_eval_final).  One product per coefficient: multiplier and HBM are balanced.
__device__ __forceinline__ Fr chunk_eval(const Fr* __restrict__ c, size_t n, size_t base, const PolyScratch* ps, Fr* sm)
{
    const int tid = threadIdx.x;
    const Fr z256 = ps->pow2z[8];
    Fr s = Fr::zero();
#pragma unroll 4
    for (int e = EV_CHUNK / 256 - 1; e >= 0; e--) {
        s = fe_mul(s, z256);
        const size_t i = base + (size_t)e * 256 + tid;
        if (i < n) s = fe_add(s, fe_load<FrP>(c + i));
    }
    s = fe_mul(s, ps->ztid[tid]);
    return block_sum(s, sm); // without z^base: the final kernels weight the blocks' sums, all at once, instead of one lane's chain of products per block

}
__global__ void __launch_bounds__(256) k_eval_partial(const Fr* __restrict__ c, size_t n, const PolyScratch* ps, Fr* partials)
{
    __shared__ Fr sm[128];
    const Fr s = chunk_eval(c, n, (size_t)blockIdx.x * EV_CHUNK, ps, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_eval_final(const Fr* __restrict__ partials, size_t count, const PolyScratch* ps, Fr* result)
{
    __shared__ Fr sm[128];
    Fr s = Fr::zero();
    for (size_t i = threadIdx.x; i < count; i += 256) s = fe_add(s, fe_mul(partials[i], pow_from_table(ps->pow2z, i * EV_CHUNK)));
    s = block_sum(s, sm);
    if (threadIdx.x == 0) *result = fe_reduce_once(s);
}
// several polynomials, each at one of two points (zeta, zeta * w), in one launch: grid = (blocks of the longest, count)
struct MultiEvalArgs {
    const Fr* poly[MEV_MAX];
    size_t len[MEV_MAX];
    int point[MEV_MAX];
    size_t stride; // partials per polynomial
};
__global__ void __launch_bounds__(256) k_multi_eval_partial(MultiEvalArgs a, const PolyScratch* ps, Fr* partials)
{
    __shared__ Fr sm[128];
    const int k = blockIdx.y;
    const size_t base = (size_t)blockIdx.x * EV_CHUNK;
    if (base >= a.len[k]) { // uniform per block
        if (threadIdx.x == 0) partials[k * a.stride + blockIdx.x] = Fr::zero();
        return;
    }
    const Fr s = chunk_eval(a.poly[k], a.len[k], base, ps + a.point[k], sm);
    if (threadIdx.x == 0) partials[k * a.stride + blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_multi_eval_final(MultiEvalArgs a, const Fr* __restrict__ partials, const PolyScratch* ps, Fr* results)
{
    __shared__ Fr sm[128];
    const Fr* p = partials + (size_t)blockIdx.x * a.stride;
    const Fr* pow2z = ps[a.point[blockIdx.x]].pow2z;
    const size_t blocks = (a.len[blockIdx.x] + EV_CHUNK - 1) / EV_CHUNK; // the partial sums beyond the polynomial's own blocks are zero
    Fr s = Fr::zero();
    for (size_t i = threadIdx.x; i < blocks; i += 256) s = fe_add(s, fe_mul(p[i], pow_from_table(pow2z, i * EV_CHUNK)));
    s = block_sum(s, sm);
    if (threadIdx.x == 0) results[blockIdx.x] = fe_reduce_once(s);
}

} // namespace bbg
#include "poly29.hip.h"
namespace bbg {
static void poly_lincomb_launch(bool limbs29, int grid, const LinCombArgs& a, const Fr* base, Fr* out, size_t n, hipStream_t st)
{
    if (limbs29) hipLaunchKernelGGL(p29::k_poly_lincomb29, dim3(grid), dim3(256), 0, st, a, base, out, n);
    else hipLaunchKernelGGL(k_poly_lincomb, dim3(grid), dim3(256), 0, st, a, base, out, n);
}
int poly_lincomb(bbg_ctx* ctx, const void* const* d_polys, const uint64_t* scalars, size_t count, const void* d_base, void* d_out, size_t n, hipStream_t st)
{
    if (count > (size_t)LC_MAX) { set_error("bbg_poly_linear_combination_device: at most 32 terms per call"); return BBG_E_INVALID; }
    if ((count && (!d_polys || !scalars)) || !d_out) { set_error("bbg_poly_linear_combination_device: null argument"); return BBG_E_INVALID; }
    if (n == 0) return BBG_OK;
    LinCombArgs a;
    a.count = (int)count;
    for (size_t k = 0; k < count; k++) {
        if (!d_polys[k]) { set_error("bbg_poly_linear_combination_device: null polynomial"); return BBG_E_INVALID; }
        a.polys[k] = (const Fr*)d_polys[k];
        memcpy(&a.scalars[k], scalars + 4 * k, 32);
        if (ctx->poly_limbs29) { // k_poly_lincomb29 takes 32 c_k, canonical (poly29.hip.h: Mult)
            hostfr::H c;
            memcpy(&c, scalars + 4 * k, 32);
            c = times32(hostfr::canon(c));
            memcpy(&a.scalars[k], &c, 32);
        }
    }
    int grid = grid_for(n, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    if (ctx->poly_limbs29 && grid > 1024) grid = 1024; // four blocks per CU, each element loop amortising the block's table and multiplier set-up
    poly_lincomb_launch(ctx->poly_limbs29 != 0, grid, a, (const Fr*)d_base, (Fr*)d_out, n, st);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}


static Fr fr_from_host(const uint64_t* limbs)
{
    Fr z;
    memcpy(&z, limbs, 32);
    return z;
}
static int poly_setup(bbg_ctx* ctx, size_t n, const uint64_t* z, PolyHeader** hdr, Fr** partials, size_t* nblocks, hipStream_t st)
{
    // partial sums: one per EV_CHUNK coefficients (evaluation) and one per 256 slices (Kate block totals + carries)
    const size_t slice = (size_t)1 << kate_log_e(n);
    *nblocks = ((n + slice - 1) / slice + 255) / 256;
    int rc = poly_scratch(ctx, 2 * (*nblocks + 1), hdr, partials);
    if (rc) return rc;
    hipLaunchKernelGGL(k_poly_pow2, dim3(1), dim3(256), 0, st, &(*hdr)->ps[0], host_pow2(z, nullptr));
    return BBG_OK;
}

// F(z) into *d_result (device), asynchronous
static void eval_async(bbg_ctx* ctx, const Fr* d_coeffs, size_t n, const PolyScratch* ps, Fr* partials, Fr* d_result, hipStream_t st)
{
    const size_t blocks = (n + EV_CHUNK - 1) / EV_CHUNK;
    if (ctx->poly_limbs29) hipLaunchKernelGGL(p29::k_eval_partial29, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, st, d_coeffs, n, ps, partials);
    else hipLaunchKernelGGL(k_eval_partial, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, st, d_coeffs, n, ps, partials);
    hipLaunchKernelGGL(k_eval_final, dim3(1), dim3(256), 0, st, (const Fr*)partials, blocks ? blocks : 1, ps, d_result);
}

int poly_evaluate(bbg_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t* z, uint64_t* out, hipStream_t st)
{
    if ((!d_coeffs && n) || !z || !out) { set_error("bbg_poly_evaluate: null argument"); return BBG_E_INVALID; }
    PolyHeader* hdr;
    Fr* partials;
    size_t nblocks;
    int rc = poly_setup(ctx, n ? n : 1, z, &hdr, &partials, &nblocks, st);
    if (rc) return rc;
    eval_async(ctx, (const Fr*)d_coeffs, n, &hdr->ps[0], partials, &hdr->ps[0].result, st);
    BBG_HIP(hipMemcpyAsync(out, &hdr->ps[0].result, 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    return BBG_OK;
}

// count polynomials (device arrays of len[k] coefficients), polynomial k at zeta (shifted[k] == 0) or at zeta * w_n (the root of
// the 2^log2n domain; the shifted evaluations of add_opening_evaluations_to_transcript, kate_commitment_scheme.cpp:375-420);
// results (canonical Montgomery) to d_results[count] on the device, asynchronous.
int poly_multi_evaluate(bbg_ctx* ctx, const void* const* d_polys, const size_t* lens, const int* shifted, size_t count, unsigned log2n,
                        const uint64_t* zeta, void** d_results, hipStream_t st)
{
    if (count == 0 || count > (size_t)MEV_MAX || !d_polys || !lens || !zeta || !d_results) {
        set_error("poly_multi_evaluate: bad argument (1..32 polynomials)");
        return BBG_E_INVALID;
    }
    size_t longest = 1;
    MultiEvalArgs a;
    for (size_t k = 0; k < count; k++) {
        if (!d_polys[k]) { set_error("poly_multi_evaluate: null polynomial"); return BBG_E_INVALID; }
        a.poly[k] = (const Fr*)d_polys[k];
        a.len[k] = lens[k];
        a.point[k] = shifted && shifted[k] ? 1 : 0;
        if (lens[k] > longest) longest = lens[k];
    }
    a.stride = (longest + EV_CHUNK - 1) / EV_CHUNK;
    void* dc = nullptr;
    int rc = ntt_domain_consts(ctx, log2n, &dc);
    if (rc) return rc;
    PolyHeader* hdr;
    Fr* partials;
    rc = poly_scratch(ctx, a.stride * count, &hdr, &partials);
    if (rc) return rc;
    uint64_t root[4];
    rc = ntt_domain_root_host(ctx, log2n, root); // the small domain's generator w (host copy kept with the domain)
    if (rc) return rc;
    hipLaunchKernelGGL(k_poly_pow2, dim3(1), dim3(256), 0, st, &hdr->ps[0], host_pow2(zeta, nullptr));
    hipLaunchKernelGGL(k_poly_pow2, dim3(1), dim3(256), 0, st, &hdr->ps[1], host_pow2(zeta, root));
    if (ctx->poly_limbs29) hipLaunchKernelGGL(p29::k_multi_eval_partial29, dim3((unsigned)a.stride, (unsigned)count), dim3(256), 0, st, a, (const PolyScratch*)hdr->ps, partials);
    else hipLaunchKernelGGL(k_multi_eval_partial, dim3((unsigned)a.stride, (unsigned)count), dim3(256), 0, st, a, (const PolyScratch*)hdr->ps, partials);
    hipLaunchKernelGGL(k_multi_eval_final, dim3((unsigned)count), dim3(256), 0, st, a, (const Fr*)partials, (const PolyScratch*)hdr->ps, hdr->results);
    BBG_HIP(hipGetLastError());
    *d_results = hdr->results;
    return BBG_OK;
}

// ---------------------------------------------------------------------------------------------- Kate opening quotient
// The reference runs the first-order recurrence dest[i] = (src[i] - dest[i-1]) * (-1/z) from i = 0 (one thread).  The
// same polynomial W(X) = (F(X) - F(z)) / (X - z) has the closed form  w_i = sum_{j > i} f_j z^(j-i-1)  (synthetic division
// from the top), which is a suffix scan: T_t = S_t + z^E T_{t+1} over the per-thread slice sums S_t, first inside a
// block (Hillis-Steele with multipliers z^(E 2^k)), then across blocks; each thread then unrolls its slice downwards.
template <int LOG_E> __global__ void __launch_bounds__(256) k_kate_block_totals(const Fr* __restrict__ f, size_t n, const PolyScratch* ps, Fr* totals)
{
    constexpr int E = 1 << LOG_E;
    __shared__ Fr sm[128];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i0 = t * E;
    Fr s = Fr::zero();
    if (i0 < n) s = fe_mul(slice_horner<E>(f, i0, n, ps->pow2z[0]), pow_from_table(ps->pow2z, (uint64_t)threadIdx.x * E));
    s = block_sum(s, sm);
    if (threadIdx.x == 0) totals[blockIdx.x] = s; // suffix evaluation of the block's B = 256 E coefficients from its first one
}
// carry[b] = sum_{u > b} totals[u] z^(B (u - b - 1))   (one block, blocks processed from the top in rounds of 256)
template <int LOG_E> __global__ void __launch_bounds__(256) k_kate_block_scan(const Fr* __restrict__ totals, size_t nblocks, const PolyScratch* ps, Fr* carry)
{
    __shared__ Fr sm[256];
    __shared__ Fr incoming;
    const int tid = threadIdx.x;
    if (tid == 0) incoming = Fr::zero();
    __syncthreads();
    const size_t rounds = (nblocks + 255) / 256;
    for (size_t r = rounds; r-- > 0;) {
        const size_t b = r * 256 + tid;
        Fr v = b < nblocks ? totals[b] : Fr::zero();
        // inclusive suffix scan inside the round: G_b = v_b + z^B G_{b+1}
        sm[tid] = v;
        __syncthreads();
        const size_t live = nblocks - r * 256 < 256 ? nblocks - r * 256 : 256; // totals of this round; the rest are zeros: steps beyond them add nothing
        for (int k = 0; k < 8 && ((size_t)1 << k) < live; k++) {
            const int d = 1 << k;
            Fr add = Fr::zero();
            if (tid + d < 256) add = fe_mul(sm[tid + d], ps->pow2z[8 + LOG_E + k]);
            __syncthreads();
            v = fe_add(v, add);
            sm[tid] = v;
            __syncthreads();
        }
        // plus the carry from the rounds above: z^(B (256 - tid)) * incoming
        const Fr inc = incoming;
        Fr g = fe_add(v, fe_mul(inc, pow_from_table(ps->pow2z, (uint64_t)(256 - tid) << (8 + LOG_E))));
        // carry INTO block b is the suffix starting at block b+1
        __syncthreads();
        sm[tid] = g;
        __syncthreads();
        if (b < nblocks) carry[b] = (tid + 1 < 256) ? sm[tid + 1] : inc;
        __syncthreads();
        if (tid == 0) incoming = g;
        __syncthreads();
    }
}
template <int LOG_E> __global__ void __launch_bounds__(256) k_kate_finish(const Fr* __restrict__ f, Fr* dest, size_t n, const PolyScratch* ps, const Fr* __restrict__ carry)
{
    constexpr int E = 1 << LOG_E;
    __shared__ Fr sm[256];
    const int tid = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * 256 + tid;
    const size_t i0 = t * E;
    const Fr z = ps->pow2z[0];
    Fr v = i0 < n ? slice_horner<E>(f, i0, n, z) : Fr::zero();
    sm[tid] = v;
    __syncthreads();
    for (int k = 0; k < 8; k++) { // T_t = S_t + z^E T_{t+1} within the block
        const int d = 1 << k;
        Fr add = Fr::zero();
        if (tid + d < 256) add = fe_mul(sm[tid + d], ps->pow2z[LOG_E + k]);
        __syncthreads();
        v = fe_add(v, add);
        sm[tid] = v;
        __syncthreads();
    }
    if (i0 >= n) return;
    // suffix evaluation starting at the NEXT slice: in-block part + the blocks above
    Fr T = (tid + 1 < 256) ? sm[tid + 1] : Fr::zero();
    T = fe_add(T, fe_mul(carry[blockIdx.x], pow_from_table(ps->pow2z, (uint64_t)(255 - tid) * E)));
    // w_{i0+E-1} = T ; w_{i-1} = f_i + z w_i
    Fr w = T;
    for (int e = E - 1; e >= 0; e--) {
        const size_t i = i0 + e;
        if (i < n) {
            fe_store<FrP>(dest + i, fe_reduce_once(w));
            w = fe_add(fe_mul(w, z), fe_load<FrP>(f + i));
        }
    }
}

// asynchronous core: W(X) into d_dest, F(z) into *d_f (device; may be null).  d_dest must not alias d_src.
int poly_kate_opening_async(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t* z, void* d_f, hipStream_t st)
{
    if (!d_src || !d_dest || !z || n == 0) { set_error("bbg_kate_opening: bad argument"); return BBG_E_INVALID; }
    if (d_src == d_dest) { set_error("bbg_kate_opening: src and dest must differ"); return BBG_E_INVALID; }
    PolyHeader* hdr;
    Fr* partials;
    size_t nblocks;
    int rc = poly_setup(ctx, n, z, &hdr, &partials, &nblocks, st);
    if (rc) return rc;
    PolyScratch* ps = &hdr->ps[0];
    Fr* carry = partials + nblocks + 1;
    // F(z)
    eval_async(ctx, (const Fr*)d_src, n, ps, partials, &ps->result, st);
    if (d_f) BBG_HIP(hipMemcpyAsync(d_f, &ps->result, 32, hipMemcpyDeviceToDevice, st));
    // W(X)
    const Fr* src = (const Fr*)d_src;
    Fr* dest = (Fr*)d_dest;
    const dim3 g((unsigned)nblocks), one(1), b(256);
#define BBG_KATE(L)                                                                                \
    hipLaunchKernelGGL(k_kate_block_totals<L>, g, b, 0, st, src, n, (const PolyScratch*)ps, partials);     \
    hipLaunchKernelGGL(k_kate_block_scan<L>, one, b, 0, st, (const Fr*)partials, nblocks, (const PolyScratch*)ps, carry); \
    hipLaunchKernelGGL(k_kate_finish<L>, g, b, 0, st, src, dest, n, (const PolyScratch*)ps, (const Fr*)carry);
    switch (kate_log_e(n)) {
    case 2: BBG_KATE(2) break;
    case 3: BBG_KATE(3) break;
    default: BBG_KATE(4) break;
    }
#undef BBG_KATE
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}
int poly_kate_opening(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t* z, uint64_t* f_out, hipStream_t st)
{
    if (!f_out) { set_error("bbg_kate_opening: bad argument"); return BBG_E_INVALID; }
    int rc = poly_kate_opening_async(ctx, d_src, d_dest, n, z, nullptr, st);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(f_out, &((PolyHeader*)ctx->poly_scratch)->ps[0].result, 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    return BBG_OK;
}

// ---------------------------------------------------------------------------------------------- divide by Z*_H
__device__ Fr fr_inv_fermat(Fr a) { return fe_inverse_gcd(a); } // (the name is history: field.hip.h's binary extended Euclid since r4)
// compute_multiplicative_subgroup (:119-138) + the "- 1", invert, numerator constants (:680-697)
__global__ void k_dpv_setup(DpvConsts* c, const DomainConsts* src, const DomainConsts* ext_dom, int log2_src, int ext, int cut)
{
    const int j = threadIdx.x;
    if (j < ext) {
        Fr acc = src->gen; // g^n
        for (int i = 0; i < log2_src; i++) acc = fe_sqr(acc);
        // * w_ext^j, w_ext = root of the size-ext domain
        acc = fe_mul(acc, pow_from_table(ext_dom->pow2_root, (uint64_t)j));
        acc = fe_sub(acc, Fr::one());
        c->inv_sub[j] = fe_reduce_once(fr_inv_fermat(acc));
    }
    if (j == 0) {
        Fr k = fe_neg(src->root_inv);
        for (int i = 0; i < cut; i++) {
            c->numer[i] = fe_reduce_once(fe_reduce_once(k));
            k = fe_mul(k, src->root_inv);
        }
    }
}
__global__ void __launch_bounds__(256) k_dpv_apply(Fr* evals, size_t n, const DpvConsts* c, const DomainConsts* target, int ext_mask, int cut)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t i0 = t * PV_E;
    if (i0 >= n) return;
    Fr x = fe_mul(target->gen, pow_from_table(target->pow2_root, i0)); // g * w_T^i
    const Fr w = target->root;
    for (int e = 0; e < PV_E && i0 + e < n; e++) {
        const size_t i = i0 + e;
        Fr v = fe_mul(fe_load<FrP>(evals + i), c->inv_sub[i & ext_mask]);
        for (int k = 0; k < cut; k++) v = fe_mul(v, fe_add(x, c->numer[k]));
        fe_store<FrP>(evals + i, v);
        x = fe_mul(x, w);
    }
}
// the Z*_H division constants of (source domain, target domain, roots cut): `ext` inversions (0.34 ms of single-lane latency), computed once
// per context and kept
static int dpv_constants(bbg_ctx* ctx, unsigned log2_src, unsigned log2_target, size_t roots_cut, DpvConsts** out, void** target_consts, hipStream_t st)
{
    if (log2_target < log2_src || log2_target > 28 || log2_target - log2_src > 4 || roots_cut > DPV_MAX_CUT) {
        set_error("bbg_divide_by_pseudo_vanishing: need src <= target <= 2^28, target/src <= 16, roots_cut <= 8");
        return BBG_E_INVALID;
    }
    void *csrc, *cext;
    int rc = ntt_domain_consts(ctx, log2_src, &csrc);
    if (!rc) rc = ntt_domain_consts(ctx, log2_target, target_consts);
    if (!rc) rc = ntt_domain_consts(ctx, log2_target - log2_src, &cext);
    if (rc) return rc;
    const uint32_t key = (log2_src << 16) | (log2_target << 8) | (uint32_t)roots_cut;
    const int ext = 1 << (log2_target - log2_src);
    auto it = ctx->dpv_consts.find(key);
    if (it != ctx->dpv_consts.end()) {
        *out = (DpvConsts*)it->second;
        return BBG_OK;
    }
    // built on the stream of THIS call and completed before the pointer is published: a later call on any other stream can use the
    // cached constants without an ordering of its own, and a failed set-up leaves no entry behind (one-off, once per key)
    DpvConsts* dc = nullptr;
    BBG_HIP(hipMalloc((void**)&dc, sizeof(DpvConsts)));
    hipLaunchKernelGGL(k_dpv_setup, dim3(1), dim3(64), 0, st, dc, (const DomainConsts*)csrc, (const DomainConsts*)cext, (int)log2_src, ext, (int)roots_cut);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(dc);
        return hip_fail(e, "k_dpv_setup", __FILE__, __LINE__);
    }
    ctx->dpv_consts[key] = dc;
    *out = dc;
    return BBG_OK;
}
int poly_divide_pseudo_vanishing(bbg_ctx* ctx, void* d_evals, unsigned log2_src, unsigned log2_target, size_t roots_cut, hipStream_t st)
{
    if (!d_evals) { set_error("bbg_divide_by_pseudo_vanishing: null evaluations"); return BBG_E_INVALID; }
    DpvConsts* dc = nullptr;
    void* ctgt = nullptr;
    int rc = dpv_constants(ctx, log2_src, log2_target, roots_cut, &dc, &ctgt, st);
    if (rc) return rc;
    const int ext = 1 << (log2_target - log2_src);
    const size_t n = (size_t)1 << log2_target;
    hipLaunchKernelGGL(k_dpv_apply, dim3(grid_for((n + PV_E - 1) / PV_E, 256)), dim3(256), 0, st, (Fr*)d_evals, n, dc, (const DomainConsts*)ctgt, ext - 1,
                       (int)roots_cut);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// The divisor itself as a table: entry i = inv_sub[i mod ext] * prod_k (g w^i + numer_k), what k_dpv_apply multiplies evaluation i by -- the
// kernel run over an array of ones.  A prover divides the quotient's 4n evaluations and then takes them through a coset iFFT; with the table
// the division rides on the transform's first load (ntt_coset_ifft_scaled) instead of a read-modify-write pass of its own: 2^22 evaluations,
// 0.28 ms -> one more product per element inside an issue-bound pass.  32 bytes per evaluation (128 MiB for a 2^20-gate circuit), per
// (source, target, cut), kept for the life of the context like the twiddle tables; bbg_memory_trim releases it.
__global__ void __launch_bounds__(256) k_fill_one(Fr* out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) fe_store<FrP>(out + i, Fr::one());
}
int poly_dpv_table(bbg_ctx* ctx, unsigned log2_src, unsigned log2_target, size_t roots_cut, const void** table, hipStream_t st)
{
    const uint32_t key = (log2_src << 16) | (log2_target << 8) | (uint32_t)roots_cut;
    auto it = ctx->dpv_tables.find(key);
    if (it != ctx->dpv_tables.end()) {
        *table = it->second;
        return BBG_OK;
    }
    DpvConsts* dc = nullptr;
    void* ctgt = nullptr;
    int rc = dpv_constants(ctx, log2_src, log2_target, roots_cut, &dc, &ctgt, st);
    if (rc) return rc;
    const size_t n = (size_t)1 << log2_target;
    Fr* t = nullptr;
    const hipError_t me = hipMalloc((void**)&t, n * sizeof(Fr));
    if (me == hipErrorOutOfMemory) { // not an error of the proof: the caller divides in a pass of its own
        (void)hipGetLastError();
        return BBG_E_NOMEM;
    }
    BBG_HIP(me);
    hipLaunchKernelGGL(k_fill_one, dim3(grid_for(n, 256)), dim3(256), 0, st, t, n);
    hipLaunchKernelGGL(k_dpv_apply, dim3(grid_for((n + PV_E - 1) / PV_E, 256)), dim3(256), 0, st, t, n, dc, (const DomainConsts*)ctgt,
                       (1 << (log2_target - log2_src)) - 1, (int)roots_cut);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st); // complete before the pointer is published (see dpv_constants)
    if (e != hipSuccess) {
        (void)hipFree(t);
        return hip_fail(e, "poly_dpv_table", __FILE__, __LINE__);
    }
    ctx->dpv_tables[key] = t;
    *table = t;
    return BBG_OK;
}

} // namespace bbg
