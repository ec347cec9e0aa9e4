// Radix-2^r multi-pass NTT over BN254 Fr for gfx950.
//
// Computes what the reference's polynomial_arithmetic::fft family computes
// (polynomials/polynomial_arithmetic.cpp:140-255 fft_inner_parallel, :374-484 the public variants):
// A_i = sum_j a_j w^(ij), natural order in, natural order out, in place from the caller's view.
//
// The reference runs log2(n) radix-2 stages over the whole array (one OpenMP barrier per stage).  Here the
// transform of size n = R_1 * R_2 * ... * R_p is split into p passes (p <= 4); pass q performs R_q-point
// sub-transforms entirely inside LDS, so the array crosses HBM p times instead of log2(n) times:
//
//   storage position = d_1*(n/R_1) + d_2*(n/(R_1 R_2)) + ...      (d_q = digit owned by pass q)
//   pass q < p ("column" pass):  for fixed higher digits and W consecutive lower positions `lo`, transform over d_q
//        (stride S_q = n / (R_1..R_q)), multiply by the inter-pass twiddle w_{R_q S_q}^(i_q * lo), write back in place.
//   pass p   ("row" pass):       contiguous R_p-point transforms; result i_p of row (d_1..d_{p-1}) goes to natural
//        index  d_1 + R_1*(d_2 + R_2*(... + R_{p-1}*i_p)), i.e. the digit reversal is folded into the last store.
//        A tile holds W rows with consecutive d_1 so the stores are W*32-byte runs.
//
// Inside a tile the R-point transform is decimation-in-frequency on an LDS tile laid out [column][k] in two
// 16-byte planes (conflict-free ds_read_b128 along k); the bit-reversed result order is undone when the tile is
// written out.  Butterfly: u = a + b, v = (a - b) * w  (1 Montgomery mul, skipped in the last stage where w = 1).
//
// Roofline note: one Fr multiplication is ~136 v_mad_u64_u32; the pass kernels are integer-ALU bound,
// not HBM bound -- the 64n algorithmic bytes cross HBM p (2..3) times plus one twiddle-table read.
#include "bbg_internal.h"

#include <cstring>
#include "field.hip.h"
#ifdef BBG_NTT_MUL29 // A/B build: every product of the transform kernels through the 29-bit multiplier (field29.hip.h fe_mul29)
#include "field29.hip.h"
#define fe_mul fe_mul29
#endif
#include "ntt_consts.hip.h"

namespace bbg {

// Primitive 2^28-th root of unity, Montgomery form (reference ecc/curves/bn254/fr.hpp:27-30).
__device__ __constant__ uint32_t FR_PRIMITIVE_ROOT_28[8] = { 0x80d13d9cu, 0x636e7355u, 0x2445ffd6u, 0xa22bf374u,
                                                             0x1eb203d8u, 0x56452ac0u, 0x2963f9e7u, 0x1860ef94u };

__device__ Fr fr_invert(Fr a) { return fe_inverse_gcd<FrP, true>(a); } // (field.hip.h: binary extended Euclid, a fifth of the a^(p-2) chain)

// evaluation_domain constructor restated on device (polynomials/evaluation_domain.cpp:57-76;
// get_root_of_unity: ecc/fields/field_impl.hpp:496-503; coset_generator(0) = 5: fr.hpp:44-59).
__global__ void k_domain_init(DomainConsts* c, unsigned log2n)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr root;
#pragma unroll
    for (int i = 0; i < 8; i++) root.v[i] = FR_PRIMITIVE_ROOT_28[i];
    for (unsigned i = 28; i > log2n; i--) root = fe_sqr(root);
    Fr root_inv = fr_invert(root);
    Fr nn = Fr::zero();
    // n = 2^log2n as a plain integer (log2n <= 28 fits limb 0), then to Montgomery form
    nn.v[0] = 1u << log2n;
    Fr n_inv = fr_invert(fe_to_mont(nn));
    Fr five = Fr::zero();
    five.v[0] = 5;
    Fr gen = fe_to_mont(five);
    c->root = fe_reduce_once(root);
    c->root_inv = fe_reduce_once(root_inv);
    c->n_inv = fe_reduce_once(n_inv);
    c->gen = fe_reduce_once(gen);
    c->gen_inv = fe_reduce_once(fr_invert(gen));
    Fr a = c->root, b = c->root_inv;
    for (int i = 0; i < 32; i++) {
        c->pow2_root[i] = a;
        c->pow2_root_inv[i] = b;
        a = fe_reduce_once(fe_sqr(a));
        b = fe_reduce_once(fe_sqr(b));
    }
}

// *dst = v: small host constants reach the device as kernel arguments, never through an asynchronous copy out of the caller's
// (pageable) memory, which the caller could reuse or free before the copy has run
__global__ void k_set_fr(Fr* dst, Fr v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}
static Fr fr_arg(const uint64_t* limbs)
{
    Fr v;
    memcpy(&v, limbs, 32);
    return v;
}

// pow2[b] = base^(2^b)
__global__ void k_pow2_table(Fr* pow2, const Fr* base_a, const Fr* base_b)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr a = *base_a;
    if (base_b) a = fe_mul(a, *base_b);
    a = fe_reduce_once(a);
    for (int i = 0; i < 32; i++) {
        pow2[i] = a;
        a = fe_reduce_once(fe_sqr(a));
    }
}

// out[(i << logS) + lo] = w^(i * lo)   (inter-pass twiddles, canonical so that a < 4p operand bound holds)
__global__ void k_twiddle_2d(Fr* out, const Fr* pow2, int logR, int logS, const Fr* scale)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)1 << (logR + logS);
    if (idx >= total) return;
    uint64_t i = idx >> logS, lo = idx & (((size_t)1 << logS) - 1);
    Fr w = pow_from_table(pow2, i * lo);
    if (scale) w = fe_mul(w, *scale);
    fe_store<FrP>(out + idx, fe_reduce_once(w));
}
// out[j] = w^(j * step), j < count   (radix twiddles: step = n / R)
__global__ void k_twiddle_1d(Fr* out, const Fr* pow2, size_t count, uint64_t step)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    fe_store<FrP>(out + idx, fe_reduce_once(pow_from_table(pow2, idx * step)));
}
// out[j] = start * base^j, j < count, `pow2` = table of base^(2^b).  Each thread does E consecutive entries.
constexpr int POW_E = 16;
__global__ void k_powers(Fr* out, const Fr* pow2, const Fr* start, size_t count)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t j0 = t * POW_E;
    if (j0 >= count) return;
    Fr g = pow_from_table(pow2, j0);
    if (start) g = fe_mul(g, *start);
    const Fr base = pow2[0];
    for (int e = 0; e < POW_E && j0 + e < count; e++) {
        fe_store<FrP>(out + j0 + e, fe_reduce_once(g));
        g = fe_mul(g, base);
    }
}
// a[j] *= table[j], j < count
__global__ void k_scale_table(Fr* a, const Fr* __restrict__ table, size_t count)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    fe_store<FrP>(a + j, fe_mul(fe_load<FrP>(a + j), fe_load<FrP>(table + j)));
}
// a[j] *= c
__global__ void k_scale_const(Fr* a, const Fr* c, size_t count)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    fe_store<FrP>(a + j, fe_mul(fe_load<FrP>(a + j), *c));
}
// a[j] *= start * base^j computed on the fly (arbitrary generator: coset_fft_with_generator_shift / _with_constant)
__global__ void k_scale_powers(Fr* a, const Fr* pow2, const Fr* start, size_t count)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t j0 = t * POW_E;
    if (j0 >= count) return;
    Fr g = pow_from_table(pow2, j0);
    if (start) g = fe_mul(g, *start);
    const Fr base = pow2[0];
    for (int e = 0; e < POW_E && j0 + e < count; e++) {
        fe_store<FrP>(a + j0 + e, fe_mul(fe_load<FrP>(a + j0 + e), g));
        g = fe_mul(g, base);
    }
}
// d[j] = reduce_once(a[j]): canonical output like the reference's operator== / serialisation sees it
__global__ void k_canon(Fr* a, size_t count)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    fe_store<FrP>(a + j, fe_reduce_once(fe_load<FrP>(a + j)));
}
// interleave ext sub-results: out[ext*i + k] = in[k*n + i]   (coset_fft 4-way split, polynomial_arithmetic.cpp:432-455)
__global__ void k_interleave(const Fr* in, Fr* out, int log2n, int logext)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)1 << (log2n + logext);
    if (idx >= total) return;
    size_t k = idx & (((size_t)1 << logext) - 1), i = idx >> logext;
    fe_store<FrP>(out + idx, fe_load<FrP>(in + (k << log2n) + i));
}
__global__ void k_replicate(Fr* buf, int log2n, int ext)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << log2n;
    if (idx >= n) return;
    Fr v = fe_load<FrP>(buf + idx);
    for (int k = 1; k < ext; k++) fe_store<FrP>(buf + (size_t)k * n + idx, v);
}

// ------------------------------------------------------------------------------------------ pass kernel
struct PassParams {
    const Fr* in;
    Fr* out;
    const Fr* tw_inter; // [R][S] table or nullptr
    const Fr* tw_radix; // R/2 entries
    const uint32_t* tw_radix29; // the same R entries as w R' mod p, canonical, 9 x 29-bit limbs in 12-word rows (k_ntt_pass29: ntt29.hip.h)
    const Fr* post;     // optional per-output-element multiplier table indexed by natural output index (row pass only)
    const Fr* pre;      // first (column) pass of k_ntt_pass8: input element g < pre_count is multiplied by pre[g] as it is loaded
    size_t pre_count;   //   (coset_fft's a_j *= g^j for j < generator_size, polynomial_arithmetic.cpp:395-399, fused into the load)
    size_t in_count;    // first (column) pass of k_ntt_pass8: input elements g >= in_count are ZERO and are not read
    int logR, logW, logS; // logS = log2(stride) (0 for the row pass)
    int row_pass;
    int log2n;
    int logR1;   // row pass: log2 of the first digit's radix (rows of a tile differ in d_1)
    int nmid;    // row pass: number of middle digits (0..2)
    int logMid[2];
    // a BATCH of independent transforms of the same domain through one launch (r5: the wires of a small circuit -- at n <= 2^18 one transform
    // has 32 .. 128 tiles for 256 CUs): grid.y = batch, transform y reads in_b[y] and writes out_b[y]; batch <= 1: in / out as they are
    // (scalar fields and a select chain, NOT arrays indexed by blockIdx.y: a dynamically indexed kernel-argument array makes the compiler
    // keep the whole parameter block in scratch memory -- every p.field access a scratch load: the pass kernels ran 12-19 % slower)
    int batch;
    const Fr *in_b1, *in_b2, *in_b3; // transform 0 is in / out
    Fr *out_b1, *out_b2, *out_b3;
};
#define BBG_NTT_SELECT_BATCH(p)                                                                                      \
    do {                                                                                                             \
        const unsigned y_ = blockIdx.y;                                                                              \
        if (y_ != 0) {                                                                                               \
            (p).in = y_ == 1 ? (p).in_b1 : y_ == 2 ? (p).in_b2 : (p).in_b3;                                          \
            (p).out = y_ == 1 ? (p).out_b1 : y_ == 2 ? (p).out_b2 : (p).out_b3;                                      \
        }                                                                                                            \
    } while (0)

__device__ __forceinline__ uint32_t bitrev(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

__global__ void __launch_bounds__(1024) k_ntt_pass(PassParams p)
{
    extern __shared__ uint4 lds[];
    BBG_NTT_SELECT_BATCH(p);
    const int R = 1 << p.logR, W = 1 << p.logW;
    const int pitch = R + 1;
    uint4* plo = lds;
    uint4* phi = lds + W * pitch;
    uint4* tlo = phi + W * pitch;
    uint4* thi = tlo + (R >> 1);
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t tile = blockIdx.x;
    const int tile_elems = R * W;

    for (int i = tid; i < (R >> 1); i += nt) {
        const uint4* q = reinterpret_cast<const uint4*>(p.tw_radix + i);
        tlo[i] = q[0];
        thi[i] = q[1];
    }

    // ---- load tile: LDS element (k, c) lives at plane[c * pitch + k]
    size_t base = 0;      // column pass: position of (k=0, c=0)
    size_t lo0 = 0;       // column pass: first low position of the tile
    size_t d1_0 = 0, rest = 0; // row pass
    if (!p.row_pass) {
        const int tiles_per_hi_log = p.logS - p.logW;
        const size_t hi = tile >> tiles_per_hi_log;
        lo0 = (tile & (((size_t)1 << tiles_per_hi_log) - 1)) << p.logW;
        base = (hi << (p.logR + p.logS)) + lo0;
        for (int e = tid; e < tile_elems; e += nt) {
            const int c = e & (W - 1), k = e >> p.logW;
            const uint4* q = reinterpret_cast<const uint4*>(p.in + base + ((size_t)k << p.logS) + c);
            plo[c * pitch + k] = q[0];
            phi[c * pitch + k] = q[1];
        }
    } else {
        // rows are indexed by hi = d_1 * (rows / R_1) + rest ; a tile takes W consecutive d_1 for one `rest`
        const int logRows = p.log2n - p.logR;         // total rows = n / R
        const int logRestCount = logRows - p.logR1;   // rows per d_1
        rest = tile & (((size_t)1 << logRestCount) - 1);
        d1_0 = (tile >> logRestCount) << p.logW;
        for (int e = tid; e < tile_elems; e += nt) {
            const int k = e & (R - 1), c = e >> p.logR;
            const size_t row = ((d1_0 + c) << logRestCount) + rest;
            const uint4* q = reinterpret_cast<const uint4*>(p.in + (row << p.logR) + k);
            plo[c * pitch + k] = q[0];
            phi[c * pitch + k] = q[1];
        }
    }
    __syncthreads();

    // ---- decimation-in-frequency stages, half-size m = R/2 ... 1
    const int half = R >> 1;
    const int nbf = half * W;
    for (int s = p.logR - 1; s >= 0; s--) {
        const int m = 1 << s;
        for (int b = tid; b < nbf; b += nt) {
            const int c = b >> (p.logR - 1);
            const int bb = b & (half - 1);
            const int j = bb & (m - 1);
            const int k = ((bb >> s) << (s + 1)) + j;
            const int ia = c * pitch + k, ib = ia + m;
            uint4 alo = plo[ia], ahi = phi[ia], blo = plo[ib], bhi = phi[ib];
            Fr a, bv;
            a.v[0] = alo.x; a.v[1] = alo.y; a.v[2] = alo.z; a.v[3] = alo.w;
            a.v[4] = ahi.x; a.v[5] = ahi.y; a.v[6] = ahi.z; a.v[7] = ahi.w;
            bv.v[0] = blo.x; bv.v[1] = blo.y; bv.v[2] = blo.z; bv.v[3] = blo.w;
            bv.v[4] = bhi.x; bv.v[5] = bhi.y; bv.v[6] = bhi.z; bv.v[7] = bhi.w;
            Fr u = fe_add(a, bv);
            Fr v = fe_sub(a, bv);
            if (s > 0) {
                const int ti = j << (p.logR - 1 - s);
                uint4 wlo = tlo[ti], whi = thi[ti];
                Fr w;
                w.v[0] = wlo.x; w.v[1] = wlo.y; w.v[2] = wlo.z; w.v[3] = wlo.w;
                w.v[4] = whi.x; w.v[5] = whi.y; w.v[6] = whi.z; w.v[7] = whi.w;
                v = fe_mul(v, w);
            }
            plo[ia] = make_uint4(u.v[0], u.v[1], u.v[2], u.v[3]);
            phi[ia] = make_uint4(u.v[4], u.v[5], u.v[6], u.v[7]);
            plo[ib] = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            phi[ib] = make_uint4(v.v[4], v.v[5], v.v[6], v.v[7]);
        }
        __syncthreads();
    }

    // ---- store: LDS row k holds result index i = bitrev(k); lanes run along c (consecutive global positions)
    for (int e = tid; e < tile_elems; e += nt) {
        const int c = e & (W - 1), k = e >> p.logW;
        const uint32_t i = bitrev((uint32_t)k, p.logR);
        uint4 xlo = plo[c * pitch + k], xhi = phi[c * pitch + k];
        Fr x;
        x.v[0] = xlo.x; x.v[1] = xlo.y; x.v[2] = xlo.z; x.v[3] = xlo.w;
        x.v[4] = xhi.x; x.v[5] = xhi.y; x.v[6] = xhi.z; x.v[7] = xhi.w;
        size_t dst;
        if (!p.row_pass) {
            if (p.tw_inter) x = fe_mul(x, fe_load<FrP>(p.tw_inter + ((size_t)i << p.logS) + lo0 + c));
            dst = base + ((size_t)i << p.logS) + c;
        } else {
            // natural index = d_1 + R_1 * (mid digits, least significant = first middle digit) + (n / R) * i
            size_t acc = 0;
            int shift = 0;
            size_t r = rest;
            // rest = d_2 * R_3 + d_3 (most significant first); output wants d_2 + R_2 * d_3
            if (p.nmid == 1) {
                acc = r;
                shift = p.logMid[0];
            } else if (p.nmid == 2) {
                const size_t d3 = r & (((size_t)1 << p.logMid[1]) - 1), d2 = r >> p.logMid[1];
                acc = d2 + (d3 << p.logMid[0]);
                shift = p.logMid[0] + p.logMid[1];
            }
            dst = (d1_0 + c) + (acc << p.logR1) + ((size_t)i << (p.logR1 + shift));
            if (p.post) x = fe_mul(x, fe_load<FrP>(p.post + dst));
        }
        fe_store<FrP>(p.out + dst, x);
    }
}

} // namespace bbg
#include "ntt_pass8.hip.h"
#include "ntt_pass29.hip.h"
namespace bbg {
// The per-radix table of the 29-bit-limb pass kernel from the R-form one (ntt29.hip.h), rows of NTT29_TW_ROW words.  shoup (the pass's kernel takes
// the constant-operand product: p29_shoup(log-radix)): row j = the limbs of the PLAIN twiddle
// w (canonical) and of wq = floor(w 2^261 / p), 18 words in a 20-word row (field29c.hip.h: the constant-operand product) -- wq by binary long
// division, 261 steps on a 9-word remainder, once per table entry (a table has at most 2048).  Otherwise: row j = in[j] * 32 (canonical) as
// 9 x 29-bit limbs in a 12-word row: w R -> w R' for R' = 2^261 = 32 R, the operand of the Montgomery product.
__global__ void k_to_rprime(uint32_t* out, const Fr* in, size_t count, int shoup)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    uint32_t* row = out + idx * NTT29_TW_ROW;
    if (shoup) {
    const Fr w = fe_canon(fe_from_mont(fe_load<FrP>(in + idx)));
    uint32_t rem[9], q[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        rem[i] = i < 8 ? w.v[i] : 0u;
        q[i] = 0;
    }
    for (int b = 0; b < 261; b++) {
        // rem <<= 1 (rem < p < 2^254 before: no overflow of 9 words), q <<= 1
#pragma unroll
        for (int i = 8; i > 0; i--) {
            rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
            q[i] = (q[i] << 1) | (q[i - 1] >> 31);
        }
        rem[0] <<= 1;
        q[0] <<= 1;
        // rem >= p ?  (compare from the top; rem[8] can only be 0 or 1 here and p has 8 words)
        bool ge = rem[8] != 0;
        if (!ge) {
            ge = true;
            for (int i = 7; i >= 0; i--) {
                if (rem[i] != FrP::MOD[i]) {
                    ge = rem[i] > FrP::MOD[i];
                    break;
                }
            }
        }
        if (ge) {
            uint32_t borrow = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const uint64_t d = (uint64_t)rem[i] - (i < 8 ? FrP::MOD[i] : 0u) - borrow;
                rem[i] = (uint32_t)d;
                borrow = (uint32_t)(d >> 63);
            }
            q[0] |= 1u;
        }
    }
    // 29-bit limbs of w (8 words) and of q (261 bits in 9 words)
    const Fr29 wl = f29_from_fe<FrP, 0>(w);
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, i = bit >> 5, o = bit & 31;
        const uint64_t lo = q[i], hi = i + 1 < 9 ? q[i + 1] : 0;
        row[j] = wl.v[j];
        row[9 + j] = (uint32_t)(((lo | (hi << 32)) >> o) & M29);
    }
    row[18] = row[19] = 0;
    return;
    }
    Fr c = Fr::zero();
    c.v[0] = 32;
    const Fr29 l = f29_from_fe<FrP, 0>(fe_canon(fe_mul(fe_load<FrP>(in + idx), fe_to_mont(c))));
#pragma unroll
    for (int i = 0; i < NTT29_TW_ROW; i++) row[i] = i < 9 ? l.v[i] : 0u;
}

// ------------------------------------------------------------------------------------------ host side
static int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// sizes at which the 29-bit-limb pass kernel is the automatic choice (option ntt_limbs29 = -1).  Measured, isolated fft, ms
// (profiles/r04_ntt29_ab.txt, last series; best 32-bit kernel -> k_ntt_pass29): 2^16 0.0523 -> 0.0507, 2^19 0.0773 -> 0.0748,
// 2^20 0.1225 -> 0.1107, 2^21 0.2540 -> 0.2329, 2^22 0.4788 -> 0.4447, 2^23 0.9477 -> 0.8851, 2^24 1.9194 -> 1.7706.
// r5, with the constant-operand product in the radix >= 2^9 kernels (isolated fft, ms, 32-bit -> 29-bit kernel): 2^14 0.0485 -> 0.0477, 2^16 0.0523 -> 0.0504,
// 2^17 0.0573 -> 0.0542, 2^18 0.0646 -> 0.0583 (2^9 x 2^9: both passes take it), 2^19 0.0776 -> 0.0705: the threshold moves to 2^18.  Below it the gain is
// 1-5 % in isolation and NEGATIVE inside small proofs (2^16 gates 3.86 -> 3.88 ms, 2^14 2.71 -> 2.73 with the 29-bit kernel forced: its 180+ VGPRs
// leave less room beside the reduce chains it runs next to).
#define NTT_LIMBS29_AUTO(log2n) ((log2n) >= 18)

static void plan_passes(bbg_ctx* ctx, NttDomain& d)
{
    const int L = (int)d.log2n;
    d.use_pass8 = false;
    d.tile_log8 = P8_TILE_LOG;
    if (ctx->ntt_kernel == 2 && ((ctx->ntt_big_tile >= 1 && L == 21) || (ctx->ntt_big_tile >= 2 && L == 22) || (ctx->ntt_big_tile >= 3 && L == 20))) {
        // 4096-element tiles (512 threads): radix 2^11 x 2^10 in TWO passes, W = 2 / 4 columns per tile -- one whole pass over the array (and
        // its inter-pass twiddle table) less than the 2048-tile plan's three passes, paid for with 64-byte instead of 256-byte global runs.
        // Measured (HIP events, 50 back-to-back transforms): 2^21 0.2537 vs 0.2604 ms (kept, default); 2^22 (2^11 x 2^11, W = 2 in both
        // passes) 0.518 vs 0.502 ms (option value 2 only).
        d.passes = 2;
        d.logR[0] = L == 20 ? 10 : 11; // (2^20 with 4096-element tiles: round-5 experiment, option value 3 -- 128-byte runs in the column pass, one block of eight waves per CU)
        d.logR[1] = L - d.logR[0];
        d.logW[0] = P8_TILE_LOG_BIG - d.logR[0];
        d.logW[1] = P8_TILE_LOG_BIG - d.logR[1];
        d.tile_log8 = P8_TILE_LOG_BIG;
        d.use_pass8 = true;
        return;
    }
    if (ctx->ntt_kernel == 2 && L >= P8_TILE_LOG) {
        // register-resident radix-8 kernel: 2048-element tiles, log-radix 3..11 per pass, balanced split
        int maxr = ctx->ntt_max_logr8;
        if (maxr > 11) maxr = 11;
        if (maxr < 6) maxr = 6;
        int p = (L + maxr - 1) / maxr;
        if (p <= NTT_MAX_PASSES) {
            d.passes = p;
            int rem = L;
            bool ok = true;
            for (int q = 0; q < p; q++) {
                int r = (rem + (p - q) - 1) / (p - q);
                d.logR[q] = r;
                d.logW[q] = P8_TILE_LOG - r;
                rem -= r;
                if (r < 3 || r > 11) ok = false;
            }
            int logS = L;
            for (int q = 0; q < p && ok; q++) {
                logS -= d.logR[q];
                if (q < p - 1 && d.logW[q] > logS) ok = false;
                if (q == p - 1 && p > 1 && d.logW[q] > d.logR[0]) ok = false;
            }
            if (ok) {
                d.use_pass8 = true;
                return;
            }
        }
    }
    const int tile = ctx->ntt_tile_log;
    int maxr = ctx->ntt_max_logr;
    if (maxr > tile) maxr = tile;
    if (maxr > 10) maxr = 10; // LDS: 2 planes + R/2 radix twiddles must fit 160 KiB
    if (L <= 11) {
        d.passes = 1;
        d.logR[0] = L;
        d.logW[0] = 0;
        return;
    }
    int p = (L + maxr - 1) / maxr;
    if (p < 2) p = 2;
    d.passes = p;
    int rem = L;
    for (int q = 0; q < p; q++) {
        int r = (rem + (p - q) - 1) / (p - q);
        d.logR[q] = r;
        rem -= r;
    }
    for (int q = 0; q < p; q++) {
        int w = tile - d.logR[q];
        // column pass q: W <= S_q ; row pass: W <= R_1
        int logS = L;
        for (int k = 0; k <= q; k++) logS -= d.logR[k];
        if (q < p - 1 && w > logS) w = logS;
        if (q == p - 1 && w > d.logR[0]) w = d.logR[0];
        d.logW[q] = w;
    }
}

void ntt_free_domain(NttDomain& d)
{
    if (d.consts) (void)hipFree(d.consts);
    for (int inv = 0; inv < 2; inv++)
        for (int q = 0; q < NTT_MAX_PASSES; q++) {
            if (d.tw_inter[inv][q]) (void)hipFree(d.tw_inter[inv][q]);
            if (d.tw_radix[inv][q]) (void)hipFree(d.tw_radix[inv][q]);
            if (d.tw_radix29[inv][q]) (void)hipFree(d.tw_radix29[inv][q]);
        }
    if (d.coset_fwd) (void)hipFree(d.coset_fwd);
    if (d.coset_inv) (void)hipFree(d.coset_inv);
    d = NttDomain();
}

static int build_domain(bbg_ctx* ctx, unsigned log2n, NttDomain** out)
{
    auto it = ctx->domains.find(log2n);
    if (it != ctx->domains.end()) {
        *out = &it->second;
        return BBG_OK;
    }
    NttDomain d;
    d.log2n = log2n;
    plan_passes(ctx, d);
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)1 << log2n;
    BBG_HIP(hipMalloc(&d.consts, sizeof(DomainConsts)));
    d.bytes += sizeof(DomainConsts);
    hipLaunchKernelGGL(k_domain_init, dim3(1), dim3(64), 0, st, (DomainConsts*)d.consts, log2n);
    DomainConsts* dc = (DomainConsts*)d.consts;
    for (int inv = 0; inv < 2; inv++) {
        const Fr* pow2 = inv ? dc->pow2_root_inv : dc->pow2_root;
        int logS = (int)log2n;
        for (int q = 0; q < d.passes; q++) {
            const int logR = d.logR[q];
            logS -= logR;
            // radix twiddles w_R^j = w_n^(j * n/R)
            const size_t half = (size_t)1 << logR; // w_R^x for x < R (k_ntt_pass uses the first half, k_ntt_pass8 all of it)
            BBG_HIP(hipMalloc(&d.tw_radix[inv][q], half * sizeof(Fr)));
            d.bytes += half * sizeof(Fr);
            hipLaunchKernelGGL(k_twiddle_1d, dim3(grid_for(half, 256)), dim3(256), 0, st, (Fr*)d.tw_radix[inv][q], pow2, half,
                               (uint64_t)(n >> logR));
            if (d.use_pass8) { // the same table in R'-form, as 9-limb rows, for k_ntt_pass29 (R <= 2048 entries of 48 or 80 bytes)
                BBG_HIP(hipMalloc(&d.tw_radix29[inv][q], half * NTT29_TW_ROW * 4));
                d.bytes += half * NTT29_TW_ROW * 4;
                hipLaunchKernelGGL(k_to_rprime, dim3(grid_for(half, 256)), dim3(256), 0, st, (uint32_t*)d.tw_radix29[inv][q], (const Fr*)d.tw_radix[inv][q], half,
                                   p29_shoup(logR) != 0 ? 1 : 0);
            }
            if (q < d.passes - 1) {
                // inter-pass twiddles w_{N_q}^(i*lo), N_q = R*S ; w_{N_q} = w_n^(n/N_q): use the pow2 table shifted
                const int logN = logR + logS;
                const size_t cnt = (size_t)1 << logN;
                BBG_HIP(hipMalloc(&d.tw_inter[inv][q], cnt * sizeof(Fr)));
                d.bytes += cnt * sizeof(Fr);
                // w_{N_q}^(2^b) = w_n^(2^(b + log2n - logN))
                // inverse direction: n^-1 rides on the first pass's inter-pass twiddles, so ifft costs exactly what fft costs
                // (the reference scales in a separate sweep, polynomial_arithmetic.cpp:379-385)
                const Fr* scale = (inv && q == 0) ? &dc->n_inv : nullptr;
                hipLaunchKernelGGL(k_twiddle_2d, dim3(grid_for(cnt, 256)), dim3(256), 0, st, (Fr*)d.tw_inter[inv][q],
                                   pow2 + (log2n - logN), logR, logS, scale);
            }
        }
    }
    // coset tables: g^j and n^-1 * g^-j
    BBG_HIP(hipMalloc(&d.coset_fwd, n * sizeof(Fr)));
    BBG_HIP(hipMalloc(&d.coset_inv, n * sizeof(Fr)));
    d.bytes += 2 * n * sizeof(Fr);
    hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dc->pow2_tmp, &dc->gen, (const Fr*)nullptr);
    hipLaunchKernelGGL(k_powers, dim3(grid_for((n + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, (Fr*)d.coset_fwd,
                       dc->pow2_tmp, (const Fr*)nullptr, n);
    hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dc->pow2_tmp, &dc->gen_inv, (const Fr*)nullptr);
    d.inv_scaled = d.passes > 1; // the inverse core already delivers n^-1 * (...) when it has an inter-pass twiddle table
    hipLaunchKernelGGL(k_powers, dim3(grid_for((n + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, (Fr*)d.coset_inv,
                       dc->pow2_tmp, d.inv_scaled ? (const Fr*)nullptr : (const Fr*)&dc->n_inv, n);
    BBG_HIP(hipGetLastError());
    {
        Fr root;
        BBG_HIP(hipMemcpyAsync(&root, &dc->root, sizeof(Fr), hipMemcpyDeviceToHost, st));
        BBG_HIP(hipStreamSynchronize(st));
        memcpy(d.root_host, &root, 32);
    }
    auto ins = ctx->domains.emplace(log2n, d);
    *out = &ins.first->second;
    return BBG_OK;
}

static int launch_pass(bbg_ctx* ctx, const NttDomain& d, int q, int inverse, const Fr* in, Fr* out, const Fr* post, hipStream_t st,
                       const Fr* pre = nullptr, size_t pre_count = 0, size_t in_count = ~(size_t)0, int batch = 1, const Fr* const* in_b = nullptr,
                       Fr* const* out_b = nullptr)
{
    PassParams p;
    p.in = in;
    p.out = out;
    p.batch = batch;
    if (batch > 1) {
        p.in = in_b[0];
        p.out = out_b[0];
    }
    p.in_b1 = batch > 1 ? in_b[1] : nullptr;
    p.out_b1 = batch > 1 ? out_b[1] : nullptr;
    p.in_b2 = batch > 2 ? in_b[2] : nullptr;
    p.out_b2 = batch > 2 ? out_b[2] : nullptr;
    p.in_b3 = batch > 3 ? in_b[3] : nullptr;
    p.out_b3 = batch > 3 ? out_b[3] : nullptr;
    p.pre = pre;
    p.pre_count = pre_count;
    p.in_count = in_count;
    p.logR = d.logR[q];
    p.logW = d.logW[q];
    p.log2n = (int)d.log2n;
    p.tw_radix = (const Fr*)d.tw_radix[inverse][q];
    p.tw_radix29 = (const uint32_t*)d.tw_radix29[inverse][q];
    p.post = post;
    p.row_pass = (q == d.passes - 1);
    int logS = (int)d.log2n;
    for (int k = 0; k <= q; k++) logS -= d.logR[k];
    p.logS = logS;
    p.tw_inter = p.row_pass ? nullptr : (const Fr*)d.tw_inter[inverse][q];
    p.logR1 = d.passes > 1 ? d.logR[0] : 0;
    p.nmid = d.passes >= 2 ? d.passes - 2 : 0;
    p.logMid[0] = d.passes >= 3 ? d.logR[1] : 0;
    p.logMid[1] = d.passes >= 4 ? d.logR[2] : 0;
    const int R = 1 << p.logR, W = 1 << p.logW;
    const size_t lds_bytes = ((size_t)2 * W * (R + 1) + R) * 16 + 64;
    const size_t tiles = ((size_t)1 << d.log2n) >> (p.logR + p.logW);
#ifdef BBG_NTT_FAST_AB // experiment builds only (build_ab/): k_ntt_pass29 at radix 2^7, 2^8, 2^10 alone -- n = 2^20, 2^22, 2^24 -- compiles in a fraction of the time
    if (d.use_pass8) {
        bool& attr29 = ctx->ntt_attr29_set;
        if (!attr29) {
            BBG_HIP(p29_attr<7>()); BBG_HIP(p29_attr<8>()); BBG_HIP(p29_attr<10>()); BBG_HIP((p29_attr<10, P8_TILE_LOG_BIG>()));
            attr29 = true;
        }
        ProfScope ps(ctx, "ntt_pass", st);
        if (d.tile_log8 == P8_TILE_LOG && p.logR == 10) p29_launch<10>(p, tiles, st);
        else if (d.tile_log8 == P8_TILE_LOG_BIG && p.logR == 10) p29_launch<10, P8_TILE_LOG_BIG>(p, tiles, st);
        else if (d.tile_log8 == P8_TILE_LOG && p.logR == 8) p29_launch<8>(p, tiles, st);
        else if (d.tile_log8 == P8_TILE_LOG && p.logR == 7) p29_launch<7>(p, tiles, st);
        else { set_error("ntt: BBG_NTT_FAST_AB build (radix 2^7, 2^8, 2^10 passes only: 2^20, 2^22, 2^24)"); return BBG_E_INVALID; }
        return BBG_OK;
    }
#else
    if (d.use_pass8) {
        bool& attr8 = ctx->ntt_attr8_set; // hipFuncSetAttribute is per device: the flag lives in the context, not in a process-wide static
        if (!attr8) {
            BBG_HIP(p8_attr<3>()); BBG_HIP(p8_attr<4>()); BBG_HIP(p8_attr<5>()); BBG_HIP(p8_attr<6>()); BBG_HIP(p8_attr<7>());
            BBG_HIP(p8_attr<8>()); BBG_HIP(p8_attr<9>()); BBG_HIP(p8_attr<10>()); BBG_HIP(p8_attr<11>());
            BBG_HIP((p8_attr<10, P8_TILE_LOG_BIG>())); BBG_HIP((p8_attr<11, P8_TILE_LOG_BIG>()));
            attr8 = true;
        }
        ProfScope ps(ctx, "ntt_pass", st);
        if (ctx->ntt_limbs29 == 1 || (ctx->ntt_limbs29 == -1 && NTT_LIMBS29_AUTO(d.log2n))) { // the pass on lazily reduced 29-bit limbs (ntt_pass29.hip.h)
            bool& attr29 = ctx->ntt_attr29_set;
            if (!attr29) {
                BBG_HIP(p29_attr<3>()); BBG_HIP(p29_attr<4>()); BBG_HIP(p29_attr<5>()); BBG_HIP(p29_attr<6>()); BBG_HIP(p29_attr<7>());
                BBG_HIP(p29_attr<8>()); BBG_HIP(p29_attr<9>()); BBG_HIP(p29_attr<10>()); BBG_HIP(p29_attr<11>());
                BBG_HIP((p29_attr<10, P8_TILE_LOG_BIG>())); BBG_HIP((p29_attr<11, P8_TILE_LOG_BIG>()));
                attr29 = true;
            }
            if (d.tile_log8 == P8_TILE_LOG_BIG) {
                if (p.logR == 11) p29_launch<11, P8_TILE_LOG_BIG>(p, tiles, st);
                else if (p.logR == 10) p29_launch<10, P8_TILE_LOG_BIG>(p, tiles, st);
                else { set_error("ntt: bad pass8 radix for the 4096-element tile"); return BBG_E_INVALID; }
                return BBG_OK;
            }
            switch (p.logR) {
            case 3: p29_launch<3>(p, tiles, st); break;
            case 4: p29_launch<4>(p, tiles, st); break;
            case 5: p29_launch<5>(p, tiles, st); break;
            case 6: p29_launch<6>(p, tiles, st); break;
            case 7: p29_launch<7>(p, tiles, st); break;
            case 8: p29_launch<8>(p, tiles, st); break;
            case 9: p29_launch<9>(p, tiles, st); break;
            case 10: p29_launch<10>(p, tiles, st); break;
            case 11: p29_launch<11>(p, tiles, st); break;
            default: set_error("ntt: bad pass8 radix"); return BBG_E_INVALID;
            }
            return BBG_OK;
        }
        // one-plane exchange (ntt_pass8.hip.h: k_ntt_pass8s: half the LDS, three waves per SIMD): automatic from 2^22 (measured there)
        if (ctx->ntt_lds_planes == 1 || (ctx->ntt_lds_planes == 0 && d.log2n >= 22)) {
            bool& attr8s = ctx->ntt_attr8s_set;
            if (!attr8s) {
                BBG_HIP(p8s_attr<3>()); BBG_HIP(p8s_attr<4>()); BBG_HIP(p8s_attr<5>()); BBG_HIP(p8s_attr<6>()); BBG_HIP(p8s_attr<7>());
                BBG_HIP(p8s_attr<8>()); BBG_HIP(p8s_attr<9>()); BBG_HIP(p8s_attr<10>()); BBG_HIP(p8s_attr<11>());
                BBG_HIP((p8s_attr<10, P8_TILE_LOG_BIG>())); BBG_HIP((p8s_attr<11, P8_TILE_LOG_BIG>()));
                attr8s = true;
            }
            if (d.tile_log8 == P8_TILE_LOG_BIG) {
                if (p.logR == 11) p8s_launch<11, P8_TILE_LOG_BIG>(p, tiles, st);
                else if (p.logR == 10) p8s_launch<10, P8_TILE_LOG_BIG>(p, tiles, st);
                else { set_error("ntt: bad pass8 radix for the 4096-element tile"); return BBG_E_INVALID; }
                return BBG_OK;
            }
            switch (p.logR) {
            case 3: p8s_launch<3>(p, tiles, st); break;
            case 4: p8s_launch<4>(p, tiles, st); break;
            case 5: p8s_launch<5>(p, tiles, st); break;
            case 6: p8s_launch<6>(p, tiles, st); break;
            case 7: p8s_launch<7>(p, tiles, st); break;
            case 8: p8s_launch<8>(p, tiles, st); break;
            case 9: p8s_launch<9>(p, tiles, st); break;
            case 10: p8s_launch<10>(p, tiles, st); break;
            case 11: p8s_launch<11>(p, tiles, st); break;
            default: set_error("ntt: bad pass8 radix"); return BBG_E_INVALID;
            }
            return BBG_OK;
        }
        if (d.tile_log8 == P8_TILE_LOG_BIG) {
            if (p.logR == 11) p8_launch<11, P8_TILE_LOG_BIG>(p, tiles, st);
            else if (p.logR == 10) p8_launch<10, P8_TILE_LOG_BIG>(p, tiles, st);
            else { set_error("ntt: bad pass8 radix for the 4096-element tile"); return BBG_E_INVALID; }
            return BBG_OK;
        }
        switch (p.logR) {
        case 3: p8_launch<3>(p, tiles, st); break;
        case 4: p8_launch<4>(p, tiles, st); break;
        case 5: p8_launch<5>(p, tiles, st); break;
        case 6: p8_launch<6>(p, tiles, st); break;
        case 7: p8_launch<7>(p, tiles, st); break;
        case 8: p8_launch<8>(p, tiles, st); break;
        case 9: p8_launch<9>(p, tiles, st); break;
        case 10: p8_launch<10>(p, tiles, st); break;
        case 11: p8_launch<11>(p, tiles, st); break;
        default: set_error("ntt: bad pass8 radix"); return BBG_E_INVALID;
        }
        return BBG_OK;
    }
#endif
    bool& attr_set = ctx->ntt_attr_set;
    if (!attr_set) {
        BBG_HIP(hipFuncSetAttribute((const void*)k_ntt_pass, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    int threads = (R * W) / 2;
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    ProfScope ps(ctx, "ntt_pass", st);
    hipLaunchKernelGGL(k_ntt_pass, dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(threads), lds_bytes, st, p);
    return BBG_OK;
}

// core transform: `in` -> `out` (may be the same array); `post` (optional) multiplies natural-index outputs.
// Multi-pass pass8 plans also take, fused into the first pass's load: a pre-scale table (pre[g], g < pre_count) and a zero-extended
// input (only in[0 .. in_count) exists; the rest of the domain is zero).  can_fuse(d) says whether the plan supports that.
static bool can_fuse(const NttDomain& d) { return d.use_pass8 && d.passes > 1; }
static int ntt_core(bbg_ctx* ctx, NttDomain& d, const Fr* in, Fr* out, int inverse, const Fr* post, hipStream_t st, const Fr* pre = nullptr,
                    size_t pre_count = 0, size_t in_count = ~(size_t)0)
{
    const size_t n = (size_t)1 << d.log2n;
    if (d.log2n == 0) {
        if (in != out) BBG_HIP(hipMemcpyAsync(out, in, sizeof(Fr), hipMemcpyDeviceToDevice, st));
        if (post) hipLaunchKernelGGL(k_scale_table, dim3(1), dim3(64), 0, st, out, post, (size_t)1);
        return BBG_OK;
    }
    if (d.passes == 1) {
        // a single-pass plan has no fused pre-scale and no zero-extended input: callers ask can_fuse() first; one that did not gets an error,
        // not an unscaled transform that reads past in_count
        if (pre != nullptr || in_count != ~(size_t)0) { set_error("ntt_core: this plan does not fuse a pre-scale table or a zero-extended input (can_fuse)"); return BBG_E_INVALID; }
        return launch_pass(ctx, d, 0, inverse, in, out, post, st);
    }
    int rc = ensure_buffer(&ctx->ntt_scratch, &ctx->ntt_scratch_bytes, n * sizeof(Fr));
    if (rc) return rc;
    Fr* scratch = (Fr*)ctx->ntt_scratch;
    // pass 0: in -> scratch (same positions); middle passes in place on scratch; last pass scratch -> out (transposing)
    rc = launch_pass(ctx, d, 0, inverse, in, scratch, nullptr, st, pre, pre_count, in_count);
    for (int q = 1; q < d.passes - 1 && !rc; q++) rc = launch_pass(ctx, d, q, inverse, scratch, scratch, nullptr, st);
    if (!rc) rc = launch_pass(ctx, d, d.passes - 1, inverse, scratch, out, post, st);
    return rc;
}

// `count` (<= 4) independent transforms of ONE domain through the same launches (grid.y = count): what ntt_core does, for each k, with in[k]
// -> out[k].  For domains whose single transform leaves most of the chip idle (2^18: 128 tiles for 256 CUs).
static int ntt_core_batch(bbg_ctx* ctx, NttDomain& d, int count, const Fr* const* in, Fr* const* out, int inverse, const Fr* post, hipStream_t st,
                          const Fr* pre = nullptr, size_t pre_count = 0, size_t in_count = ~(size_t)0)
{
    if (count == 1) return ntt_core(ctx, d, in[0], out[0], inverse, post, st, pre, pre_count, in_count);
    if (count < 1 || count > 4 || d.log2n == 0) { set_error("ntt_core_batch: 1 .. 4 transforms of a domain of at least two points"); return BBG_E_INVALID; }
    const size_t n = (size_t)1 << d.log2n;
    if (d.passes == 1) {
        if (pre != nullptr || in_count != ~(size_t)0) { set_error("ntt_core_batch: this plan does not fuse a pre-scale table or a zero-extended input (can_fuse)"); return BBG_E_INVALID; }
        return launch_pass(ctx, d, 0, inverse, nullptr, nullptr, post, st, nullptr, 0, ~(size_t)0, count, in, out);
    }
    int rc = ensure_buffer(&ctx->ntt_scratch, &ctx->ntt_scratch_bytes, (size_t)count * n * sizeof(Fr));
    if (rc) return rc;
    Fr* scratch[4] = { nullptr, nullptr, nullptr, nullptr };
    for (int k = 0; k < count; k++) scratch[k] = (Fr*)ctx->ntt_scratch + (size_t)k * n;
    rc = launch_pass(ctx, d, 0, inverse, nullptr, nullptr, nullptr, st, pre, pre_count, in_count, count, in, scratch);
    for (int q = 1; q < d.passes - 1 && !rc; q++) rc = launch_pass(ctx, d, q, inverse, nullptr, nullptr, nullptr, st, nullptr, 0, ~(size_t)0, count, scratch, scratch);
    if (!rc) rc = launch_pass(ctx, d, d.passes - 1, inverse, nullptr, nullptr, post, st, nullptr, 0, ~(size_t)0, count, scratch, out);
    return rc;
}

// The prover's FFT work item without its copies (work_queue.hpp:252-264): the n_in coefficients at d_in, zero-extended to the 2^log2n
// domain, coset FFT with generator_size n_in, result to d_out (2^log2n elements; must not overlap d_in).  No staging copy, no
// zero fill, no separate scaling sweep when the plan fuses (any domain of >= 2^12 points).
int ntt_coset_extend(bbg_ctx* ctx, const void* d_in, size_t n_in, void* d_out, unsigned log2n, hipStream_t st)
{
    if (!d_in || !d_out || log2n > 28 || n_in > ((size_t)1 << log2n)) { set_error("ntt_coset_extend: bad argument"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    const size_t n = (size_t)1 << log2n;
    if (can_fuse(*dp)) return ntt_core(ctx, *dp, (const Fr*)d_in, (Fr*)d_out, 0, nullptr, st, (const Fr*)dp->coset_fwd, n_in, n_in);
    BBG_HIP(hipMemcpyAsync(d_out, d_in, n_in * 32, hipMemcpyDeviceToDevice, st));
    if (n > n_in) BBG_HIP(hipMemsetAsync((char*)d_out + n_in * 32, 0, (n - n_in) * 32, st));
    return ntt_run(ctx, d_out, log2n, BBG_COSET_FFT, n_in, nullptr, st);
}

// ifft out of place: the values at d_in stay as they are, the coefficients go to d_out (no overlap) -- a wire of the resident prover keeps
// its Lagrange form for the grand product and gets its coefficient form without a staging copy (r5: 15-60 us per wire at 2^20)
int ntt_ifft_to(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2n, hipStream_t st)
{
    if (!d_in || !d_out || log2n > 28) { set_error("ntt_ifft_to: bad argument"); return BBG_E_INVALID; }
    const size_t n = (size_t)1 << log2n;
    if (log2n < 12) { // the small plans run in place
        BBG_HIP(hipMemcpyAsync(d_out, d_in, n * 32, hipMemcpyDeviceToDevice, st));
        return ntt_run(ctx, d_out, log2n, BBG_IFFT, 0, nullptr, st);
    }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    rc = ntt_core(ctx, *dp, (const Fr*)d_in, (Fr*)d_out, 1, nullptr, st);
    if (!rc && !dp->inv_scaled)
        hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, (Fr*)d_out, &((DomainConsts*)dp->consts)->n_inv, n);
    return rc;
}

// the same for `count` (<= 4) arrays at once -- the wires of a round (r5): one launch set instead of `count`
int ntt_ifft_to_batch(bbg_ctx* ctx, int count, const void* const* d_in, void* const* d_out, unsigned log2n, hipStream_t st)
{
    if (count < 1 || count > 4) { set_error("ntt_ifft_to_batch: 1 .. 4 arrays"); return BBG_E_INVALID; }
    if (count == 1 || log2n < 12) {
        for (int k = 0; k < count; k++) {
            int rc = ntt_ifft_to(ctx, d_in[k], d_out[k], log2n, st);
            if (rc) return rc;
        }
        return BBG_OK;
    }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    const size_t n = (size_t)1 << log2n;
    rc = ntt_core_batch(ctx, *dp, count, (const Fr* const*)d_in, (Fr* const*)d_out, 1, nullptr, st);
    for (int k = 0; k < count && !rc && !dp->inv_scaled; k++)
        hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, (Fr*)d_out[k], &((DomainConsts*)dp->consts)->n_inv, n);
    return rc;
}
// ntt_coset_extend for `count` (<= 4) inputs of the same length at once
int ntt_coset_extend_batch(bbg_ctx* ctx, int count, const void* const* d_in, size_t n_in, void* const* d_out, unsigned log2n, hipStream_t st)
{
    if (count < 1 || count > 4) { set_error("ntt_coset_extend_batch: 1 .. 4 arrays"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    if (count > 1 && can_fuse(*dp) && n_in <= ((size_t)1 << log2n))
        return ntt_core_batch(ctx, *dp, count, (const Fr* const*)d_in, (Fr* const*)d_out, 0, nullptr, st, (const Fr*)dp->coset_fwd, n_in, n_in);
    for (int k = 0; k < count; k++) {
        rc = ntt_coset_extend(ctx, d_in[k], n_in, d_out[k], log2n, st);
        if (rc) return rc;
    }
    return BBG_OK;
}

int ntt_coset_ifft_scaled(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, const void* d_scale, hipStream_t st)
{
    if (!d_coeffs || !d_scale || log2n > 28) { set_error("ntt_coset_ifft_scaled: bad argument"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    if (!can_fuse(*dp)) return BBG_E_NOFUSE;
    return ntt_core(ctx, *dp, (Fr*)d_coeffs, (Fr*)d_coeffs, 1, (const Fr*)dp->coset_inv, st, (const Fr*)d_scale, (size_t)1 << log2n);
}

int ntt_run(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant,
            hipStream_t st)
{
    if (!d_coeffs) { set_error("bbg_ntt: null coeffs"); return BBG_E_INVALID; }
    if (log2n > 28) { set_error("bbg_ntt: log2n > 28 exceeds the 2-adicity of BN254 Fr (fr.hpp:27-30)"); return BBG_E_INVALID; }
    if (op < 0 || op > 7) { set_error("bbg_ntt: unknown op"); return BBG_E_INVALID; }
    if (op >= 4 && !constant) { set_error("bbg_ntt: op needs a constant"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    NttDomain& d = *dp;
    DomainConsts* dc = (DomainConsts*)d.consts;
    const size_t n = (size_t)1 << log2n;
    if (generator_size == 0 || generator_size > n) generator_size = n;
    Fr* a = (Fr*)d_coeffs;
    Fr* cst = nullptr;
    if (constant) {
        cst = &dc->constant;
        hipLaunchKernelGGL(k_set_fr, dim3(1), dim3(1), 0, st, cst, fr_arg(constant));
    }
    switch (op) {
    case BBG_FFT:
        rc = ntt_core(ctx, d, a, a, 0, nullptr, st);
        break;
    case BBG_IFFT:
        rc = ntt_core(ctx, d, a, a, 1, nullptr, st);
        if (!rc && !d.inv_scaled) hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, a, &dc->n_inv, n);
        break;
    case BBG_COSET_FFT:
        if (can_fuse(d)) { // a_j *= g^j (j < generator_size) as the first pass loads a_j
            rc = ntt_core(ctx, d, a, a, 0, nullptr, st, (const Fr*)d.coset_fwd, generator_size);
            break;
        }
        hipLaunchKernelGGL(k_scale_table, dim3(grid_for(generator_size, 256)), dim3(256), 0, st, a, (const Fr*)d.coset_fwd,
                           generator_size);
        rc = ntt_core(ctx, d, a, a, 0, nullptr, st);
        break;
    case BBG_COSET_IFFT:
        rc = ntt_core(ctx, d, a, a, 1, (const Fr*)d.coset_inv, st); // the table carries n^-1 g^-j, or g^-j when the core delivers n^-1
        break;
    case BBG_FFT_WITH_CONSTANT:
        rc = ntt_core(ctx, d, a, a, 0, nullptr, st);
        if (!rc) hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, a, cst, n);
        break;
    case BBG_COSET_FFT_WITH_CONSTANT:
        // a[j] *= constant * g^j, j < generator_size
        hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dc->pow2_tmp, &dc->gen, (const Fr*)nullptr);
        hipLaunchKernelGGL(k_scale_powers, dim3(grid_for((generator_size + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, a,
                           dc->pow2_tmp, cst, generator_size);
        rc = ntt_core(ctx, d, a, a, 0, nullptr, st);
        break;
    case BBG_COSET_FFT_WITH_GENERATOR_SHIFT:
        // generator = g * constant
        hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dc->pow2_tmp, &dc->gen, (const Fr*)cst);
        hipLaunchKernelGGL(k_scale_powers, dim3(grid_for((generator_size + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, a,
                           dc->pow2_tmp, (const Fr*)nullptr, generator_size);
        rc = ntt_core(ctx, d, a, a, 0, nullptr, st);
        break;
    case BBG_IFFT_WITH_CONSTANT:
        rc = ntt_core(ctx, d, a, a, 1, nullptr, st);
        if (!rc) {
            if (!d.inv_scaled) hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, a, &dc->n_inv, n);
            hipLaunchKernelGGL(k_scale_const, dim3(grid_for(n, 256)), dim3(256), 0, st, a, cst, n);
        }
        break;
    }
    if (rc) return rc;
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

int ntt_coset_split(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, size_t ext, hipStream_t st)
{
    int logext = 0;
    while (((size_t)1 << logext) < ext) logext++;
    if (((size_t)1 << logext) != ext || ext == 0 || log2n + logext > 28) {
        set_error("bbg_coset_fft_split: ext must be a power of two with n*ext <= 2^28");
        return BBG_E_INVALID;
    }
    NttDomain *dsmall = nullptr, *dlarge = nullptr;
    int rc = build_domain(ctx, log2n, &dsmall);
    if (rc) return rc;
    rc = build_domain(ctx, log2n + logext, &dlarge);
    if (rc) return rc;
    const size_t n = (size_t)1 << log2n;
    Fr* a = (Fr*)d_coeffs;
    DomainConsts* dcs = (DomainConsts*)dsmall->consts;
    DomainConsts* dcl = (DomainConsts*)dlarge->consts;
    // replicate the n coefficients into ext slots, scale slot k by (g * w_{ext n}^k)^j, transform each, interleave
    hipLaunchKernelGGL(k_replicate, dim3(grid_for(n, 256)), dim3(256), 0, st, a, (int)log2n, (int)ext);
    // generator of slot k: g * root_large^k ; build incrementally in pow2_tmp[30] of the small domain
    Fr* gk = &dcs->gk;
    BBG_HIP(hipMemcpyAsync(gk, &dcs->gen, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    for (size_t k = 0; k < ext; k++) {
        hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dcs->pow2_tmp, (const Fr*)gk, (const Fr*)nullptr);
        hipLaunchKernelGGL(k_scale_powers, dim3(grid_for((n + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, a + k * n,
                           dcs->pow2_tmp, (const Fr*)nullptr, n);
        rc = ntt_core(ctx, *dsmall, a + k * n, a + k * n, 0, nullptr, st);
        if (rc) return rc;
        // gk *= root_large  (k_pow2_table with two bases writes pow2[0] = gk*root; copy back)
        hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dcs->pow2_tmp, (const Fr*)gk, (const Fr*)&dcl->root);
        BBG_HIP(hipMemcpyAsync(gk, &dcs->pow2_tmp[0], sizeof(Fr), hipMemcpyDeviceToDevice, st));
    }
    // interleave through the scratch buffer (ntt_core is done with it by now on this stream)
    rc = ensure_buffer(&ctx->ntt_scratch, &ctx->ntt_scratch_bytes, n * ext * sizeof(Fr));
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(ctx->ntt_scratch, a, n * ext * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_interleave, dim3(grid_for(n * ext, 256)), dim3(256), 0, st, (const Fr*)ctx->ntt_scratch, a, (int)log2n,
                       logext);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// ---------------------------------------------------------------------------------- building blocks of the sharded NTT
// (aztec-2.0_amd/parallel.py: residue-class decomposition across GPUs)

// out = base^e for a device-resident or staged base; single lane (setup-time helper)
__global__ void k_fr_pow(Fr* out, const Fr* base, uint64_t e)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr acc = Fr::one(), b = *base;
    while (e) {
        if (e & 1) acc = fe_mul(acc, b);
        b = fe_sqr(b);
        e >>= 1;
    }
    *out = fe_reduce_once(acc);
}

// out[t][q] = sum_s w_G^(s*t) in[s][q], s,t < G = 2^LOGG, q < len: the size-G DFT across the chunks received from the G ranks.
template <int LOGG> __global__ void __launch_bounds__(256) k_cross_dft(const Fr* __restrict__ in, Fr* out, size_t len, const Fr* wtab /* w_G^j, j < G */)
{
    constexpr int G = 1 << LOGG;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= len) return;
    Fr x[G];
#pragma unroll
    for (int s = 0; s < G; s++) x[s] = fe_load<FrP>(in + (size_t)s * len + q);
    // radix-2 DIF in registers, natural order in, bit-reversed out
#pragma unroll
    for (int st = LOGG - 1; st >= 0; st--) {
        const int m = 1 << st;
#pragma unroll
        for (int b = 0; b < G / 2; b++) {
            const int j = b & (m - 1);
            const int k = ((b >> st) << (st + 1)) + j;
            const Fr u = fe_add(x[k], x[k + m]);
            Fr v = fe_sub(x[k], x[k + m]);
            if (j != 0) v = fe_mul(v, wtab[j << (LOGG - 1 - st)]);
            x[k] = u;
            x[k + m] = v;
        }
    }
#pragma unroll
    for (int s = 0; s < G; s++) {
        const int t = (int)(__brev((uint32_t)s) >> (32 - (LOGG > 0 ? LOGG : 1))) & (G - 1);
        fe_store<FrP>(out + (size_t)(LOGG > 0 ? t : 0) * len + q, x[s]);
    }
}

// a[j] *= start * base^j (j < count); start may be null.  base / start: host limbs (Montgomery).
int ntt_scale_powers(bbg_ctx* ctx, void* d_a, size_t count, const uint64_t* start, const uint64_t* base, hipStream_t st)
{
    if (!d_a || !base) { set_error("bbg_scale_powers: null argument"); return BBG_E_INVALID; }
    if (count == 0) return BBG_OK;
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, 0, &dp); // the size-1 domain only lends its scratch constants block
    if (rc) return rc;
    DomainConsts* dc = (DomainConsts*)dp->consts;
    hipLaunchKernelGGL(k_set_fr, dim3(1), dim3(1), 0, st, &dc->gk, fr_arg(base));
    if (start) hipLaunchKernelGGL(k_set_fr, dim3(1), dim3(1), 0, st, &dc->constant, fr_arg(start));
    hipLaunchKernelGGL(k_pow2_table, dim3(1), dim3(64), 0, st, dc->pow2_tmp, (const Fr*)&dc->gk, (const Fr*)nullptr);
    hipLaunchKernelGGL(k_scale_powers, dim3(grid_for((count + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, (Fr*)d_a, dc->pow2_tmp,
                       start ? (const Fr*)&dc->constant : (const Fr*)nullptr, count);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// pow2[b] = (base^step)^(2^b) and *start = (mul ? *mul : 1) * base^e0: set-up of a geometric scaling whose base and offsets are
// device-resident domain constants (the sharded NTT's coset factors g^(r + G j) and twiddles w_n^(r q), multi.hip)
__global__ void k_geometric_setup(Fr* pow2, Fr* start, const Fr* base, uint64_t step, uint64_t e0, const Fr* mul)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Fr b = *base;
    auto pw = [&](uint64_t e) {
        Fr acc = Fr::one(), x = b;
        while (e) {
            if (e & 1) acc = fe_mul(acc, x);
            x = fe_sqr(x);
            e >>= 1;
        }
        return acc;
    };
    Fr s = pw(e0);
    if (mul) s = fe_mul(s, *mul);
    *start = fe_reduce_once(s);
    Fr a = fe_reduce_once(pw(step));
    for (int i = 0; i < 32; i++) {
        pow2[i] = a;
        a = fe_reduce_once(fe_sqr(a));
    }
}
// a[j] *= mul * base^(e0 + step * j), j < count; which_base: 0 = coset generator g, 1 = g^-1, 2 = w_n (root of the 2^log2n domain),
// 3 = w_n^-1; mul_inv_log2 >= 0: mul = (2^mul_inv_log2)^-1, else 1.  Asynchronous on st.
int ntt_scale_geometric(bbg_ctx* ctx, void* d_a, size_t count, unsigned log2n, int which_base, uint64_t step, uint64_t e0, int mul_inv_log2,
                        hipStream_t st)
{
    if (!d_a || which_base < 0 || which_base > 3) { set_error("ntt_scale_geometric: bad argument"); return BBG_E_INVALID; }
    if (count == 0) return BBG_OK;
    NttDomain *dp = nullptr, *dm = nullptr, *d0 = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (!rc) rc = build_domain(ctx, 0, &d0); // lends its scratch table
    if (!rc && mul_inv_log2 >= 0) rc = build_domain(ctx, (unsigned)mul_inv_log2, &dm);
    if (rc) return rc;
    DomainConsts* dc = (DomainConsts*)dp->consts;
    DomainConsts* sc = (DomainConsts*)d0->consts;
    const Fr* base = which_base == 0 ? &dc->gen : which_base == 1 ? &dc->gen_inv : which_base == 2 ? &dc->root : &dc->root_inv;
    const Fr* mul = dm ? &((DomainConsts*)dm->consts)->n_inv : nullptr;
    hipLaunchKernelGGL(k_geometric_setup, dim3(1), dim3(64), 0, st, sc->pow2_tmp, &sc->gk, base, step, e0, mul);
    hipLaunchKernelGGL(k_scale_powers, dim3(grid_for((count + POW_E - 1) / POW_E, 256)), dim3(256), 0, st, (Fr*)d_a, sc->pow2_tmp, (const Fr*)&sc->gk, count);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// out = w_n^e (forward root of the 2^log2n domain; inverse != 0 -> its inverse), returned to the host
int ntt_root_pow(bbg_ctx* ctx, unsigned log2n, uint64_t e, int inverse, uint64_t* out, hipStream_t st)
{
    if (log2n > 28 || !out) { set_error("bbg_fr_root_pow: bad argument"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    DomainConsts* dc = (DomainConsts*)dp->consts;
    hipLaunchKernelGGL(k_fr_pow, dim3(1), dim3(64), 0, st, &dc->gk, inverse ? &dc->root_inv : &dc->root, e);
    BBG_HIP(hipMemcpyAsync(out, &dc->gk, 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    return BBG_OK;
}
int ntt_fr_pow(bbg_ctx* ctx, const uint64_t* base, uint64_t e, uint64_t* out, hipStream_t st)
{
    if (!base || !out) { set_error("bbg_fr_pow: null argument"); return BBG_E_INVALID; }
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, 0, &dp);
    if (rc) return rc;
    DomainConsts* dc = (DomainConsts*)dp->consts;
    hipLaunchKernelGGL(k_set_fr, dim3(1), dim3(1), 0, st, &dc->constant, fr_arg(base));
    hipLaunchKernelGGL(k_fr_pow, dim3(1), dim3(64), 0, st, &dc->gk, (const Fr*)&dc->constant, e);
    BBG_HIP(hipMemcpyAsync(out, &dc->gk, 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    return BBG_OK;
}

// out[t*len + q] = sum_s w_G^(s t) in[s*len + q]; w_G = w_n^(n/G) of the 2^log2n domain (inverse: its inverse)
int ntt_cross_dft(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2G, size_t len, unsigned log2n, int inverse, hipStream_t st)
{
    if (!d_in || !d_out || log2G > 3 || log2G > log2n || log2n > 28) { set_error("bbg_cross_dft: bad argument (G <= 8)"); return BBG_E_INVALID; }
    if (len == 0) return BBG_OK;
    NttDomain* dp = nullptr;
    int rc = build_domain(ctx, log2n, &dp);
    if (rc) return rc;
    DomainConsts* dc = (DomainConsts*)dp->consts;
    // w_G^j = w_n^(j * n/G): reuse the twiddle generator into pow2_tmp[0..G)
    const Fr* pow2 = inverse ? dc->pow2_root_inv : dc->pow2_root;
    hipLaunchKernelGGL(k_twiddle_1d, dim3(1), dim3(64), 0, st, dc->pow2_tmp, pow2, (size_t)1 << log2G, (uint64_t)(((size_t)1 << log2n) >> log2G));
    const dim3 grid(grid_for(len, 256)), block(256);
    switch (log2G) {
    case 0: hipLaunchKernelGGL(k_cross_dft<0>, grid, block, 0, st, (const Fr*)d_in, (Fr*)d_out, len, (const Fr*)dc->pow2_tmp); break;
    case 1: hipLaunchKernelGGL(k_cross_dft<1>, grid, block, 0, st, (const Fr*)d_in, (Fr*)d_out, len, (const Fr*)dc->pow2_tmp); break;
    case 2: hipLaunchKernelGGL(k_cross_dft<2>, grid, block, 0, st, (const Fr*)d_in, (Fr*)d_out, len, (const Fr*)dc->pow2_tmp); break;
    default: hipLaunchKernelGGL(k_cross_dft<3>, grid, block, 0, st, (const Fr*)d_in, (Fr*)d_out, len, (const Fr*)dc->pow2_tmp); break;
    }
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

int ntt_domain_consts(bbg_ctx* ctx, unsigned log2n, void** consts)
{
    if (log2n > 28) { set_error("domain: log2n > 28"); return BBG_E_INVALID; }
    NttDomain* d = nullptr;
    int rc = build_domain(ctx, log2n, &d);
    if (rc) return rc;
    *consts = d->consts;
    return BBG_OK;
}

int ntt_domain_root_host(bbg_ctx* ctx, unsigned log2n, uint64_t out[4])
{
    if (log2n > 28) { set_error("domain: log2n > 28"); return BBG_E_INVALID; }
    NttDomain* d = nullptr;
    int rc = build_domain(ctx, log2n, &d);
    if (rc) return rc;
    memcpy(out, d->root_host, 32);
    return BBG_OK;
}

// bbg_ntt_plan: the passes a 2^log2n transform runs as NOW (options included) and which pass kernel executes them -- the selection
// logic of launch_pass restated on the plan, without building tables.  kernel: 0 = k_ntt_pass (radix 2 in LDS), 8 = k_ntt_pass8,
// 81 = k_ntt_pass8s (one-plane exchange), 29 = k_ntt_pass29 (lazily reduced 9 x 29-bit limbs)
int ntt_plan(bbg_ctx* ctx, unsigned log2n, int* passes, int* log_radix, int* kernel, int* tile_log)
{
    if (log2n > 28) { set_error("bbg_ntt_plan: log2n > 28"); return BBG_E_INVALID; }
    NttDomain d;
    auto it = ctx->domains.find(log2n);
    if (it != ctx->domains.end()) { // a built domain keeps the plan it was built with
        d.passes = it->second.passes;
        d.use_pass8 = it->second.use_pass8;
        d.tile_log8 = it->second.tile_log8;
        for (int q = 0; q < NTT_MAX_PASSES; q++) d.logR[q] = it->second.logR[q];
    } else {
        d.log2n = log2n;
        plan_passes(ctx, d);
    }
    *passes = d.passes;
    for (int q = 0; q < NTT_MAX_PASSES; q++) log_radix[q] = q < d.passes ? d.logR[q] : 0;
    *tile_log = d.use_pass8 ? d.tile_log8 : ctx->ntt_tile_log;
    if (!d.use_pass8) *kernel = 0;
    else if (ctx->ntt_limbs29 == 1 || (ctx->ntt_limbs29 == -1 && NTT_LIMBS29_AUTO(log2n))) *kernel = 29;
    else if (ctx->ntt_lds_planes == 1 || (ctx->ntt_lds_planes == 0 && log2n >= 22)) *kernel = 81;
    else *kernel = 8;
    return BBG_OK;
}

int ntt_prepare(bbg_ctx* ctx, unsigned log2n)
{
    if (log2n > 28) { set_error("bbg_ntt_prepare: log2n > 28"); return BBG_E_INVALID; }
    NttDomain* d = nullptr;
    return build_domain(ctx, log2n, &d);
}

// ---- field self-test kernels (bbg_field_op)
template <class P> __global__ void k_field_op(int op, const Fe<P>* a, const Fe<P>* b, Fe<P>* out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<P> x = fe_load<P>(a + i), y = b ? fe_load<P>(b + i) : Fe<P>::zero(), z;
    if (op == 6 || op == 7) { // raw: no pre-reduction (inputs promised < 2p): exercises the coarse representation
        z = op == 6 ? fe_mul(x, y) : fe_mul_cios(x, y);
        fe_store<P>(out + i, fe_reduce_once(z));
        return;
    }
    // inputs may be any 256-bit value: bring into [0,2p) the way the reference assumes its inputs are
    for (int k = 0; k < 5; k++) { // 2^256 < 6p
        x = fe_reduce_once(x);
        y = fe_reduce_once(y);
    }
    switch (op) {
    case 0: z = fe_mul(x, y); break;
    case 1: z = fe_add(x, y); break;
    case 2: z = fe_sub(x, y); break;
    case 3: z = fe_mul_cios(x, y); break;
    case 4: z = fe_from_mont(x); break;
    case 5: z = fe_to_mont(x); break;
    case 8: z = fe_mul_sub2(x, x, y, y); break;                       // x^2 - y^2, single reduction
    case 9: z = fe_mul_sub2(fe_neg(x), fe_neg(y), x, fe_neg(y)); break; // extreme operands (2p - x ...): = 2xy
    case 10: z = fe_inverse_gcd<P, false>(x); break;                  // the Montgomery residue of 1 / x (0 -> 0), one lane per element
    default: z = x; break;
    }
    fe_store<P>(out + i, fe_reduce_once(z));
}
// op 11: the same inverse through the scalar-unit variant (every lane of a wave holds the same input): one wave per element
template <class P> __global__ void k_field_inverse_uniform(const Fe<P>* a, Fe<P>* out)
{
    Fe<P> x = fe_load<P>(a + blockIdx.x);
    for (int k = 0; k < 5; k++) x = fe_reduce_once(x);
    const Fe<P> z = fe_inverse_gcd<P, true>(x);
    if (threadIdx.x == 0) fe_store<P>(out + blockIdx.x, z);
}
int field_op_device(int which, int op, const void* a, const void* b, void* out, size_t n, hipStream_t st)
{
    if (n == 0) return BBG_OK;
    if (op == 11) {
        if (which == 0) hipLaunchKernelGGL(k_field_inverse_uniform<FrP>, dim3((unsigned)n), dim3(64), 0, st, (const Fr*)a, (Fr*)out);
        else hipLaunchKernelGGL(k_field_inverse_uniform<FqP>, dim3((unsigned)n), dim3(64), 0, st, (const Fq*)a, (Fq*)out);
        BBG_HIP(hipGetLastError());
        return BBG_OK;
    }
    if (which == 0)
        hipLaunchKernelGGL(k_field_op<FrP>, dim3(grid_for(n, 256)), dim3(256), 0, st, op, (const Fr*)a, (const Fr*)b, (Fr*)out, n);
    else
        hipLaunchKernelGGL(k_field_op<FqP>, dim3(grid_for(n, 256)), dim3(256), 0, st, op, (const Fq*)a, (const Fq*)b, (Fq*)out, n);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

} // namespace bbg
