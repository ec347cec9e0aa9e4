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
// padded slot of register j of a thread whose register 0 sits at slot base_slot = p8_addr(pb, c): registers are 2^(F + LOGW) elements apart; where
// that is a multiple of 16 the padding is a constant per register and the slot is base + j * step -- ONE address register and immediate offsets for
// the eight accesses (r5: the compiler does not see that floor((q + 16 k) / 16) = floor(q / 16) + k and keeps eight addresses alive)
template <int F, int LOGW> __device__ __forceinline__ int p8_reg_slot(int base_slot, int pb, int c, int j)
{
    constexpr int STRIDE = 1 << (F + LOGW);
    if constexpr (STRIDE % 16 == 0) return base_slot + j * (STRIDE + (STRIDE >> 4) * 2);
    else return p8_addr(pb | (j << F), c, LOGW);
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
    const int slot0 = p8_addr(pbase, c, LOGW);
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
            x[j] = p8_lds_load(plo, phi, p8_reg_slot<F, LOGW>(slot0, pbase, c, j));
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
            p8_lds_store(plo, phi, p8_reg_slot<F, LOGW>(slot0, pbase, c, j), x[j]);
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
    BBG_NTT_SELECT_BATCH(p);
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

// ------------------------------------------------------------------------------------ one-plane exchange (r4)
// k_ntt_pass8 keeps the whole tile in LDS between two steps: two 16-byte planes, 73 728 B per 256-thread block -> two blocks per CU, two
// waves per SIMD, although its 104-116 VGPRs would allow four (profiles/r03_kernel_stats_v4.txt).  k_ntt_pass8s moves the tile between two
// steps ONE PLANE AT A TIME through a single plane-sized buffer (36 864 B -> four blocks per CU): low halves out, barrier, low halves of the
// next step's elements in, barrier, high halves out, barrier, high halves in.  No more registers than before (a thread holds the high
// halves of its old elements beside the low halves of its new ones: 8 x 8 words, what x[8] takes anyway); three barriers per exchange
// instead of one, covered by the other blocks of the CU.  Same arithmetic, same positions, same outputs -- bit-identical by construction,
// and checked against the reference digests at every size (tests/test_gpu_parity.py).
template <int LOGR, bool ROW, int T, int TL> __device__ __forceinline__ void p8s_coords(int tid, int& c, int& pbase, int& qlo)
{
    constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int LOGW = TL - LOGR, W = 1 << LOGW;
    constexpr int QBITS = LOGR - 3;
    int q;
    if (ROW && T == 0) { // lanes run along the row (k) so that the global loads coalesce
        q = tid & ((1 << QBITS) - 1);
        c = tid >> QBITS;
    } else {
        c = tid & (W - 1);
        q = tid >> LOGW;
    }
    qlo = q & ((1 << F) - 1);
    pbase = ((q >> F) << (F + 3)) | qlo; // field bits zero
}
// butterfly on the top S bits of step T's field + the step twiddles (the arithmetic of p8_step, nothing else)
template <int LOGR, int T> __device__ __forceinline__ void p8s_compute(Fr (&x)[8], const Fr* __restrict__ tw, int qlo)
{
    constexpr int S = (LOGR - 3 * T >= 3) ? 3 : (LOGR - 3 * T);
    constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
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
    if constexpr (S == 3 && F > 0) {
        constexpr int DONE = LOGR - 3 - F;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            const int e = (p8_brev3(j) * qlo) << DONE;
            x[j] = fe_mul(x[j], fe_load<FrP>(tw + e));
        }
    }
}
// x: the 8 elements of step T (in place) -> the 8 elements of step T + 1, one plane at a time through `buf`
template <int LOGR, bool ROW, int T, int TL> __device__ __forceinline__ void p8s_exchange(Fr (&x)[8], uint4* buf)
{
    constexpr int LOGW = TL - LOGR;
    constexpr int F0 = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int F1 = (LOGR - 3 * (T + 1) >= 3) ? (LOGR - 3 * (T + 2)) : 0;
    int c0, pb0, ql0, c1, pb1, ql1;
    p8s_coords<LOGR, ROW, T, TL>(threadIdx.x, c0, pb0, ql0);
    p8s_coords<LOGR, ROW, T + 1, TL>(threadIdx.x, c1, pb1, ql1);
    const int s0 = p8_addr(pb0, c0, LOGW), s1 = p8_addr(pb1, c1, LOGW);
    if (T > 0) __syncthreads(); // everybody has taken the high halves of the previous exchange out of the buffer
#pragma unroll
    for (int j = 0; j < 8; j++) buf[p8_reg_slot<F0, LOGW>(s0, pb0, c0, j)] = make_uint4(x[j].v[0], x[j].v[1], x[j].v[2], x[j].v[3]);
    __syncthreads();
    uint4 lo[8];
#pragma unroll
    for (int j = 0; j < 8; j++) lo[j] = buf[p8_reg_slot<F1, LOGW>(s1, pb1, c1, j)];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) buf[p8_reg_slot<F0, LOGW>(s0, pb0, c0, j)] = make_uint4(x[j].v[4], x[j].v[5], x[j].v[6], x[j].v[7]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 h = buf[p8_reg_slot<F1, LOGW>(s1, pb1, c1, j)];
        x[j].v[0] = lo[j].x; x[j].v[1] = lo[j].y; x[j].v[2] = lo[j].z; x[j].v[3] = lo[j].w;
        x[j].v[4] = h.x; x[j].v[5] = h.y; x[j].v[6] = h.z; x[j].v[7] = h.w;
    }
}

// The exchange in front of a LAST step that transforms ONE bit (log-radix = 1 mod 3: 2^10, 2^7): the thread holds the elements whose
// in-tile position p differs in bits [3:1] and needs those that differ in bits [2:0] -- a swap of p's bit 3 (a register-index bit) with
// p's bit 0 (a thread-index bit), i.e. half of the registers change places with ONE partner lane (tid ^ W).  Done with lane shuffles: no
// LDS buffer, no barrier -- an exchange less per 2^10 pass (three barriers and a tile's round trip through LDS).
template <int LOGR, int TL> __device__ __forceinline__ void p8s_exchange_last1(Fr (&x)[8])
{
    constexpr int LOGW = TL - LOGR;
    const uint32_t ql = (threadIdx.x >> LOGW) & 1u; // p's bit 0 of what this thread holds; p's bit 3 of what it is going to hold
    Fr y[8];
#pragma unroll
    for (int m = 0; m < 4; m++) { // m = (b2 b1) of the new position's low bits
        const int b1 = m & 1, b2 = m >> 1;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            // to the partner: my register with j2 = 1 - ql (the partner's ql), low bits m; from it: its register with j2 = my ql, low bits m
            const uint32_t send = ql ? x[m].v[w] : x[4 + m].v[w];
            const uint32_t recv = (uint32_t)__shfl_xor((int)send, 1 << LOGW);
            const uint32_t own = ql ? x[4 + m].v[w] : x[m].v[w]; // my register with j2 = ql, low bits m
            // new register j' = (b2 b1 b0): b0 = ql -> own, b0 = 1 - ql -> received
            y[(b2 << 2) | (b1 << 1) | 0].v[w] = ql ? recv : own;
            y[(b2 << 2) | (b1 << 1) | 1].v[w] = ql ? own : recv;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = y[j];
}

template <int TL> constexpr size_t p8s_lds_bytes() { return (size_t)p8_plane<TL>() * 16; }

// BBG_NTT_OCC = minimum waves per SIMD the compiler must fit the kernel into (3 -> <= 168 VGPRs, 4 -> <= 128); BBG_NTT_LATE_OUTMUL = 1
// fetches the outputs' multipliers in front of the LAST step instead of behind the first loads (their 64 registers are then free for most
// of the kernel -- what makes 168 fit without spills).  Measured, isolated, fft (profiles/r04_ntt_planes_ab.txt; two-plane kernel = 1.00):
//   OCC 2, early: 2^20 1.05, 2^22 1.00, 2^24 1.00      OCC 3, early: 2^20 1.17, 2^22 1.03, 2^24 1.00
//   OCC 2, late : 2^20 1.02, 2^22 0.99, 2^24 0.99      OCC 3, late : 2^20 1.04, 2^22 0.94, 2^24 0.96   <- the defaults
//   OCC 4 (128 VGPRs, spills): 1.13 - 1.23 at 2^20.
// So the one-plane kernel is the automatic choice from 2^22 (radix 2^7 / 2^8 passes, three per transform), the two-plane kernel below
// (2^20 = two radix-2^10 passes of four steps: the two extra barriers per exchange cost more than the third wave per SIMD hides).
#ifndef BBG_NTT_OCC
#define BBG_NTT_OCC 3
#endif
#ifndef BBG_NTT_LATE_OUTMUL
#define BBG_NTT_LATE_OUTMUL 1
#endif
#ifndef BBG_NTT_SHFL_LAST
#define BBG_NTT_SHFL_LAST 1 // a one-bit last step takes its elements by lane shuffles (p8s_exchange_last1) instead of through LDS
#endif
template <int LOGR, bool ROW, int TL = P8_TILE_LOG> __global__ void __launch_bounds__(1 << (TL - 3), BBG_NTT_OCC) k_ntt_pass8s(PassParams p)
{
    extern __shared__ uint4 lds[];
    BBG_NTT_SELECT_BATCH(p);
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
    const Fr* tw = p.tw_radix;
    Fr x[8], outmul[8];
    const Fr* mul_table = ROW ? p.post : p.tw_inter;
    const bool have_outmul = mul_table != nullptr;
    // ---- step 0: the tile from global memory (same addressing as p8_step<.., 0, ..>)
    {
        int c, pbase, qlo;
        p8s_coords<LOGR, ROW, 0, TL>(threadIdx.x, c, pbase, qlo);
        constexpr int F = (LOGR >= 3) ? (LOGR - 3) : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pj = pbase | (j << F);
            size_t g;
            if (!ROW) {
                g = base + ((size_t)pj << p.logS) + c;
                if (g >= p.in_count) {
                    x[j] = Fr::zero();
                    continue;
                }
                x[j] = fe_load<FrP>(p.in + g);
                if (p.pre && g < p.pre_count) x[j] = fe_mul(x[j], fe_load<FrP>(p.pre + g));
                continue;
            }
            g = (((((d1_0 + c) << logRestCount) + rest)) << LOGR) + pj;
            x[j] = fe_load<FrP>(p.in + g);
        }
        // the multipliers of the outputs: behind the data loads, ahead of all the arithmetic (see k_ntt_pass8)
        if (have_outmul && (!BBG_NTT_LATE_OUTMUL || NSTEPS == 1)) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int pj, cc;
                p8_last_coords<LOGR, TL>(threadIdx.x, j, pj, cc);
                outmul[j] = fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, cc, base, lo0, d1_0, rest, true));
            }
        }
        p8s_compute<LOGR, 0>(x, tw, qlo);
    }
    auto late_outmul = [&](bool last) { // in front of the last step's arithmetic: 12+ products cover the latency, the registers are free until then
        if (BBG_NTT_LATE_OUTMUL && last && have_outmul) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                int pj, cc;
                p8_last_coords<LOGR, TL>(threadIdx.x, j, pj, cc);
                outmul[j] = fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, cc, base, lo0, d1_0, rest, true));
            }
        }
    };
    if constexpr (NSTEPS > 1) {
        p8s_exchange<LOGR, ROW, 0, TL>(x, lds);
        late_outmul(NSTEPS == 2);
        int c, pbase, qlo;
        p8s_coords<LOGR, ROW, 1, TL>(threadIdx.x, c, pbase, qlo);
        p8s_compute<LOGR, 1>(x, tw, qlo);
    }
    if constexpr (NSTEPS > 2) {
        if constexpr (NSTEPS == 3 && LOGR - 3 * 2 == 1 && (TL - LOGR) <= 5 && BBG_NTT_SHFL_LAST) p8s_exchange_last1<LOGR, TL>(x);
        else p8s_exchange<LOGR, ROW, 1, TL>(x, lds);
        late_outmul(NSTEPS == 3);
        int c, pbase, qlo;
        p8s_coords<LOGR, ROW, 2, TL>(threadIdx.x, c, pbase, qlo);
        p8s_compute<LOGR, 2>(x, tw, qlo);
    }
    if constexpr (NSTEPS > 3) {
        if constexpr (NSTEPS == 4 && LOGR - 3 * 3 == 1 && (TL - LOGR) <= 5 && BBG_NTT_SHFL_LAST) p8s_exchange_last1<LOGR, TL>(x);
        else p8s_exchange<LOGR, ROW, 2, TL>(x, lds);
        late_outmul(NSTEPS == 4);
        int c, pbase, qlo;
        p8s_coords<LOGR, ROW, 3, TL>(threadIdx.x, c, pbase, qlo);
        p8s_compute<LOGR, 3>(x, tw, qlo);
    }
    // ---- the last step's elements leave for global memory (bit reversal folded into the index)
    {
        constexpr int T = NSTEPS - 1;
        constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
        int c, pbase, qlo;
        p8s_coords<LOGR, ROW, T, TL>(threadIdx.x, c, pbase, qlo);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pj = pbase | (j << F);
            Fr v = x[j];
            if (have_outmul) v = fe_mul(v, outmul[j]);
            fe_store<FrP>(p.out + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, false), v);
        }
    }
}

template <int LOGR, int TL = P8_TILE_LOG> static void p8s_launch(const PassParams& p, size_t tiles, hipStream_t st)
{
    if (p.row_pass) hipLaunchKernelGGL((k_ntt_pass8s<LOGR, true, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), p8s_lds_bytes<TL>(), st, p);
    else hipLaunchKernelGGL((k_ntt_pass8s<LOGR, false, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), p8s_lds_bytes<TL>(), st, p);
}
template <int LOGR, int TL = P8_TILE_LOG> static hipError_t p8s_attr()
{
    hipError_t e = hipFuncSetAttribute((const void*)k_ntt_pass8s<LOGR, true, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8s_lds_bytes<TL>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_ntt_pass8s<LOGR, false, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8s_lds_bytes<TL>());
}

template <int LOGR, int TL = P8_TILE_LOG> static void p8_launch(const PassParams& p, size_t tiles, hipStream_t st)
{
    if (p.row_pass) hipLaunchKernelGGL((k_ntt_pass8<LOGR, true, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), p8_lds_bytes<TL>(), st, p);
    else hipLaunchKernelGGL((k_ntt_pass8<LOGR, false, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), p8_lds_bytes<TL>(), st, p);
}
template <int LOGR, int TL = P8_TILE_LOG> static hipError_t p8_attr()
{
    hipError_t e = hipFuncSetAttribute((const void*)k_ntt_pass8<LOGR, true, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8_lds_bytes<TL>());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_ntt_pass8<LOGR, false, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p8_lds_bytes<TL>());
}

} // namespace bbg
