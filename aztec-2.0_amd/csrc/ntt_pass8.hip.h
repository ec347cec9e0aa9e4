// k_ntt_pass8: the production NTT pass kernel (included by ntt.hip).
//
// Same contract as k_ntt_pass (one pass = R-point decimation-in-frequency sub-transforms of a 2048-element tile, column or
// row flavour, see ntt.hip) but every thread keeps 8 elements in registers and does up to three radix-2 stages
// (a radix-8 DIF butterfly) per visit: a radix-2^r pass touches LDS ceil(r/3) - 1 times instead of r times and has
// as many barriers, which is what bounded the first version (profiles/r01_kernel_stats_v1.txt: 61 us per pass against
// ~35 us of multiplier time).  The arithmetic (number of Montgomery products) is unchanged: a radix-8 step spends
// 5 products inside the butterfly (w8^1, w8^2, w8^3, w4 twice) and 7 on the step twiddles w_R^(J' * Lo).
//
// Step t owns a 3-bit field [f+2 : f] of the in-tile index p; the thread holds the 8 elements that differ in that field.
// The top s bits of the field are the digit being transformed (s = 3, or 1..2 in the last step when 3 does not divide r).
// Results stay in place, so after all steps position p holds output bitrev_r(p) -- undone by the final store, exactly as
// in k_ntt_pass.
#pragma once

namespace bbg {

constexpr int P8_TILE_LOG = 11;                       // 2048 elements per tile = 256 threads x 8 (the default tile)
constexpr int P8_TILE_LOG_BIG = 12;                   // 4096 elements = 512 threads x 8: 2^21 / 2^22 in TWO passes of >= 64-byte runs (r2)
template <int TL> constexpr int p8_plane() { return (1 << TL) + ((1 << TL) >> 4) * 2; } // uint4 slots per plane incl. padding (2 per 16)
template <int TL> constexpr size_t p8_lds_bytes() { return (size_t)2 * p8_plane<TL>() * 16; }

__device__ __forceinline__ int p8_addr(int p, int c, int logW) // padded LDS slot of tile element (p, c)
{
    const int q = (p << logW) + c;
    return q + ((q >> 4) << 1);
}
__device__ __forceinline__ Fr p8_lds_load(const uint4* plo, const uint4* phi, int a)
{
    const uint4 l = plo[a], h = phi[a];
    Fr r;
    r.v[0] = l.x; r.v[1] = l.y; r.v[2] = l.z; r.v[3] = l.w;
    r.v[4] = h.x; r.v[5] = h.y; r.v[6] = h.z; r.v[7] = h.w;
    return r;
}
__device__ __forceinline__ void p8_lds_store(uint4* plo, uint4* phi, int a, const Fr& x)
{
    plo[a] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    phi[a] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ void p8_bfly(Fr& a, Fr& b) // (a, b) <- (a + b, a - b)
{
    const Fr u = fe_add(a, b);
    b = fe_sub(a, b);
    a = u;
}
__device__ __forceinline__ void p8_bfly_w(Fr& a, Fr& b, const Fr& w) // (a, b) <- (a + b, (a - b) * w)
{
    const Fr u = fe_add(a, b);
    b = fe_mul(fe_sub(a, b), w);
    a = u;
}

// S = 3: radix-8 DIF over the whole 3-bit register index j; register j ends up holding digit J' = bitrev3(j).
// S = 2 / 1 (only the LAST step of a pass whose log-radix is not a multiple of 3): the digit is the LOW S bits of j, the
// upper bits of the field are digits transformed earlier; nothing lies below, so there are no step twiddles.
template <int S> __device__ __forceinline__ void p8_butterfly(Fr (&x)[8], const Fr& w8_1, const Fr& w8_2, const Fr& w8_3)
{
    if constexpr (S == 3) {
        p8_bfly(x[0], x[4]);
        p8_bfly_w(x[1], x[5], w8_1);
        p8_bfly_w(x[2], x[6], w8_2);
        p8_bfly_w(x[3], x[7], w8_3);
        p8_bfly(x[0], x[2]);
        p8_bfly_w(x[1], x[3], w8_2);
        p8_bfly(x[4], x[6]);
        p8_bfly_w(x[5], x[7], w8_2);
        p8_bfly(x[0], x[1]);
        p8_bfly(x[2], x[3]);
        p8_bfly(x[4], x[5]);
        p8_bfly(x[6], x[7]);
    } else if constexpr (S == 2) { // radix-4 on (b1 b0), w4 = w8_2
        p8_bfly(x[0], x[2]);
        p8_bfly_w(x[1], x[3], w8_2);
        p8_bfly(x[4], x[6]);
        p8_bfly_w(x[5], x[7], w8_2);
        p8_bfly(x[0], x[1]);
        p8_bfly(x[2], x[3]);
        p8_bfly(x[4], x[5]);
        p8_bfly(x[6], x[7]);
    } else { // S == 1: radix-2 on b0
        p8_bfly(x[0], x[1]);
        p8_bfly(x[2], x[3]);
        p8_bfly(x[4], x[5]);
        p8_bfly(x[6], x[7]);
    }
}
__device__ __forceinline__ constexpr int p8_brev3(int j) { return ((j & 1) << 2) | (j & 2) | ((j >> 2) & 1); }

// one step: FIRST loads from global, LAST stores to global, otherwise through LDS
// where the LAST step of a pass delivers register j of this thread: the in-tile position pj (bit-reversed on the way out) and column c
template <int LOGR, int TL> __device__ __forceinline__ void p8_last_coords(int tid, int j, int& pj, int& c)
{
    constexpr int NSTEPS = (LOGR + 2) / 3, T = NSTEPS - 1;
    constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int LOGW = TL - LOGR, W = 1 << LOGW;
    c = tid & (W - 1);
    const int q = tid >> LOGW;
    const int qlo = q & ((1 << F) - 1);
    pj = (((q >> F) << (F + 3)) | qlo) | (j << F);
}
// global index of that output (column pass: also the index into the inter-pass twiddle table; row pass: the natural output index)
template <int LOGR, bool ROW>
__device__ __forceinline__ size_t p8_out_index(const PassParams& p, int pj, int c, size_t base, size_t lo0, size_t d1_0, size_t rest, bool twiddle)
{
    const uint32_t i = __brev((uint32_t)pj) >> (32 - LOGR);
    if (!ROW) return (twiddle ? lo0 : base) + ((size_t)i << p.logS) + c;
    size_t acc = 0;
    int shift = 0;
    if (p.nmid == 1) {
        acc = rest;
        shift = p.logMid[0];
    } else if (p.nmid == 2) {
        const size_t d3 = rest & (((size_t)1 << p.logMid[1]) - 1), d2 = rest >> p.logMid[1];
        acc = d2 + (d3 << p.logMid[0]);
        shift = p.logMid[0] + p.logMid[1];
    }
    return (d1_0 + c) + (acc << p.logR1) + ((size_t)i << (p.logR1 + shift));
}

template <int LOGR, bool ROW, int T, int TL>
__device__ __forceinline__ void p8_step(Fr (&x)[8], const PassParams& p, uint4* plo, uint4* phi, const Fr* __restrict__ tw, size_t base,
                                        size_t lo0, size_t d1_0, size_t rest, int logRestCount, const Fr (&outmul)[8], bool have_outmul)
{
    constexpr int NSTEPS = (LOGR + 2) / 3;
    constexpr bool FIRST = (T == 0), LAST = (T == NSTEPS - 1);
    constexpr int S = (LOGR - 3 * T >= 3) ? 3 : (LOGR - 3 * T);
    constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int LOGW = TL - LOGR, W = 1 << LOGW;
    constexpr int QBITS = LOGR - 3;
    const int tid = threadIdx.x;
    int c, q;
    if (ROW && FIRST) { // lanes run along the row (k) so that the global loads coalesce
        q = tid & ((1 << QBITS) - 1);
        c = tid >> QBITS;
    } else {
        c = tid & (W - 1);
        q = tid >> LOGW;
    }
    const int qlo = q & ((1 << F) - 1);
    const int pbase = ((q >> F) << (F + 3)) | qlo; // field bits zero
    // ---- fetch
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pj = pbase | (j << F);
        if (FIRST) {
            size_t g;
            if (!ROW) {
                g = base + ((size_t)pj << p.logS) + c; // first pass of a multi-pass plan: g is the natural input index
                if (g >= p.in_count) { // zero-extended input (the prover's n coefficients on the 4n domain): nothing to read
                    x[j] = Fr::zero();
                    continue;
                }
                x[j] = fe_load<FrP>(p.in + g);
                if (p.pre && g < p.pre_count) x[j] = fe_mul(x[j], fe_load<FrP>(p.pre + g)); // coset_fft's g^j, fused into the load
                continue;
            }
            g = (((((d1_0 + c) << logRestCount) + rest)) << LOGR) + pj;
            x[j] = fe_load<FrP>(p.in + g);
        } else {
            x[j] = p8_lds_load(plo, phi, p8_addr(pj, c, LOGW));
        }
    }
    // ---- butterfly on the top S bits of the field
    {
        Fr w1, w2, w3;
        if constexpr (S == 3) {
            w1 = fe_load<FrP>(tw + (1 << (LOGR - 3)));
            w3 = fe_load<FrP>(tw + (3 << (LOGR - 3)));
        }
        if constexpr (S >= 2) w2 = fe_load<FrP>(tw + (1 << (LOGR - 2)));
        else w2 = Fr::zero();
        if constexpr (S != 3) {
            w1 = Fr::zero();
            w3 = Fr::zero();
        }
        p8_butterfly<S>(x, w1, w2, w3);
    }
    // ---- step twiddle w_R^((J' * Lo) << done): Lo = the F bits below the field, done = bits above it (S = 3 steps only;
    //      a partial last step has nothing below its digit)
    if constexpr (S == 3 && F > 0) {
        constexpr int DONE = LOGR - 3 - F;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            const int e = (p8_brev3(j) * qlo) << DONE;
            x[j] = fe_mul(x[j], fe_load<FrP>(tw + e));
        }
    }
    // ---- deliver
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int pj = pbase | (j << F);
        if (!LAST) {
            p8_lds_store(plo, phi, p8_addr(pj, c, LOGW), x[j]);
        } else {
            Fr v = x[j];
            if (have_outmul) v = fe_mul(v, outmul[j]); // inter-pass twiddle (column pass) / post-scale table (row pass), fetched steps ago
            const size_t dst = p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, false);
            fe_store<FrP>(p.out + dst, v);
        }
    }
}

template <int LOGR, bool ROW, int TL = P8_TILE_LOG> __global__ void __launch_bounds__(1 << (TL - 3)) k_ntt_pass8(PassParams p)
{
    extern __shared__ uint4 lds[];
    uint4* plo = lds;
    uint4* phi = lds + p8_plane<TL>();
    constexpr int NSTEPS = (LOGR + 2) / 3;
    constexpr int LOGW = TL - LOGR;
    const size_t tile = blockIdx.x;
    size_t base = 0, lo0 = 0, d1_0 = 0, rest = 0;
    int logRestCount = 0;
    if (!ROW) {
        const int tiles_per_hi_log = p.logS - LOGW;
        const size_t hi = tile >> tiles_per_hi_log;
        lo0 = (tile & (((size_t)1 << tiles_per_hi_log) - 1)) << LOGW;
        base = (hi << (LOGR + p.logS)) + lo0;
    } else {
        const int logRows = p.log2n - LOGR;
        logRestCount = logRows - p.logR1;
        rest = tile & (((size_t)1 << logRestCount) - 1);
        d1_0 = (tile >> logRestCount) << LOGW;
    }
    const Fr* tw = p.tw_radix; // w_R^x, x < R
    Fr x[8], outmul[8];
    // The multiplier each output takes on its way out (inter-pass twiddle / post table) is a 32-byte HBM read per element whose address
    // is known from the start.  Fetched here -- behind the data loads, ahead of all the arithmetic -- instead of one at a time in the
    // last step, where with two waves per SIMD nothing covers the latency.  64 VGPRs; occupancy is bounded by LDS (2 blocks per CU).
    const Fr* mul_table = ROW ? p.post : p.tw_inter;
    const bool have_outmul = mul_table != nullptr;
    auto fetch_outmul = [&]() {
        if (have_outmul) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int pj, c;
                p8_last_coords<LOGR, TL>(threadIdx.x, j, pj, c);
                outmul[j] = fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, true));
            }
        }
    };
    if constexpr (NSTEPS == 1) fetch_outmul();
    p8_step<LOGR, ROW, 0, TL>(x, p, plo, phi, tw, base, lo0, d1_0, rest, logRestCount, outmul, NSTEPS == 1 && have_outmul);
    if constexpr (NSTEPS > 1) {
        fetch_outmul();
        __syncthreads();
        p8_step<LOGR, ROW, 1, TL>(x, p, plo, phi, tw, base, lo0, d1_0, rest, logRestCount, outmul, NSTEPS == 2 && have_outmul);
    }
    if constexpr (NSTEPS > 2) {
        __syncthreads();
        p8_step<LOGR, ROW, 2, TL>(x, p, plo, phi, tw, base, lo0, d1_0, rest, logRestCount, outmul, NSTEPS == 3 && have_outmul);
    }
    if constexpr (NSTEPS > 3) {
        __syncthreads();
        p8_step<LOGR, ROW, 3, TL>(x, p, plo, phi, tw, base, lo0, d1_0, rest, logRestCount, outmul, NSTEPS == 4 && have_outmul);
    }
}

template <int LOGR, int TL = P8_TILE_LOG> static void p8_launch(const PassParams& p, size_t tiles, hipStream_t st)
{
    if (p.row_pass) hipLaunchKernelGGL((k_ntt_pass8<LOGR, true, TL>), dim3((unsigned)tiles), dim3(1 << (TL - 3)), p8_lds_bytes<TL>(), st, p);
    else hipLaunchKernelGGL((k_ntt_pass8<LOGR, false, TL>), dim3((unsigned)tiles), dim3(1 << (TL - 3)), p8_lds_bytes<TL>(), st, p);
}
template <int LOGR, int TL = P8_TILE_LOG> static hipError_t p8_attr()
{
    hipError_t e = hipFuncSetAttribute((const void*)k_ntt_pass8<LOGR, true, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8_lds_bytes<TL>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_ntt_pass8<LOGR, false, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8_lds_bytes<TL>());
}

} // namespace bbg
