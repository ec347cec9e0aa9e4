// k_ntt_pass29: the NTT pass of k_ntt_pass8s (ntt_pass8.hip.h: 2048- / 4096-element tiles, 8 elements per thread in registers, radix-8 steps,
// one-plane-at-a-time exchange, three waves per SIMD) with every step's arithmetic on LAZILY REDUCED 9 x 29-bit limbs (ntt29.hip.h).
//
// Same contract, same positions, same global addressing as k_ntt_pass8 / k_ntt_pass8s; what differs is what a register holds between the
// load and the store: F29<FrP> values of (x / 32) in R'-Montgomery form (the 256 stored bits re-limbed -- ntt29.hip.h), 9 words each.  An
// exchange therefore moves 36 bytes per element: the low four limbs through the plane buffer, limb 8 through a buffer of its own IN THE SAME
// ROUND (separate LDS arrays, no extra barrier), then limbs 4..7 through the plane buffer.  LDS: 36 864 + 9 216 B of buffers + 3 072 B of
// reduction tables = 49.2 KB per 256-thread block, three blocks per CU.
//
// Results are the same residues as the 32-bit kernels give (the representative may differ: both are coarse, < 2p); the parity tests compare
// canonical values against the oracle and the reference digests with this kernel selected (option "ntt_limbs29").
#pragma once
#include "ntt29.hip.h"

namespace bbg {

// BBG_NTT29_EXCH1 (round 5): the exchange moves all nine words of an element in ONE round -- write, barrier, read: two barriers per exchange
// instead of four -- through a buffer that holds the whole tile: limbs 0..3 and 4..7 as two 16-byte planes, limb 8 as a 4-byte plane, 36 bytes
// per element and NO padding (a 2048-element tile: 73 728 B + 3 072 B of tables = 76 800 B, two blocks per CU in 153.6 of the 160 KB).  Bank
// conflicts are kept down by an XOR swizzle of the slot index instead of padding (p29_slot; chosen on scripts/model/lds_conflicts.py, the
// lane groups of MI355X_MICROARCH.md's LDS table: cheaper than the two-per-sixteen padding of the 16-byte planes on every step pattern).
#ifndef BBG_NTT29_EXCH1
#define BBG_NTT29_EXCH1 0
#endif
#ifndef BBG_NTT29_EXCH1_MIN_LOGR
#define BBG_NTT29_EXCH1_MIN_LOGR 0 // with BBG_NTT29_EXCH1: only the kernels of at least this log-radix (those that run two blocks per CU anyway)
#endif
constexpr bool p29_exch1(int logR) { return BBG_NTT29_EXCH1 && logR >= BBG_NTT29_EXCH1_MIN_LOGR; }
template <int TL, int LOGR> constexpr size_t p29_lds_bytes()
{
    return p29_exch1(LOGR) ? ((size_t)36 << TL) + NTT29_TABLE_WORDS * 4 : (size_t)p8_plane<TL>() * 16 + (size_t)p8_plane<TL>() * 4 + NTT29_TABLE_WORDS * 4;
}
__device__ __forceinline__ int p29_slot(int p, int c, int logW) // swizzled slot of tile element (p, c): a bijection of [0, tile)
{
    const int q = (p << logW) + c;
    return q ^ ((q >> 3) & 15);
}

// Which multiplier a pass kernel takes for its table twiddles (ntt29.hip.h N29M<SH>), r5.  Log-radix >= 9 (two waves per SIMD anyway: 182 - 198
// VGPRs with Montgomery): SH = 1, the constant-operand product with every twiddle in VGPRs (223 - 233).  Log-radix <= 8 (the three-pass plans of
// 2^22 .. 2^24, three waves per SIMD in 144 - 168 VGPRs): SH = 2, the same product with the butterfly's own, wave-uniform multipliers as SGPR operands,
// compiled for three waves (168 VGPRs: radix 2^8 fits, radix 2^7 spills 12 - 14 registers and still wins).  With SH = 1 everywhere those kernels need
// 225 - 240 VGPRs, lose their third wave and 2^22 / 2^24 get 2 - 5 % SLOWER; with SH = 2: 2^24 1.745 -> 1.670 ms, 2^22 0.456 -> 0.444
// (profiles/r05_ntt_attempts.txt).  BBG_NTT_SHOUP = 0: Montgomery everywhere; BBG_NTT_SHOUP_SMALL = 0: Montgomery in the radix <= 2^8 kernels (A/B).
#ifndef BBG_NTT29_OCC
#define BBG_NTT29_OCC 2
#endif
#ifndef BBG_NTT_SHOUP
#define BBG_NTT_SHOUP 1
#endif
#ifndef BBG_NTT_SHOUP_SMALL
#define BBG_NTT_SHOUP_SMALL 2
#endif
constexpr int p29_shoup(int logR) { return !BBG_NTT_SHOUP ? 0 : logR >= 9 ? 1 : BBG_NTT_SHOUP_SMALL; }
constexpr int p29_occ(int logR) { return (BBG_NTT_SHOUP && BBG_NTT_SHOUP_SMALL && logR <= 8) ? 3 : BBG_NTT29_OCC; }

template <int SH> __device__ __forceinline__ typename N29M<SH>::Tw p29_load_tw(const uint32_t* __restrict__ tw29, int idx) // a table row
{
    const uint4* row = reinterpret_cast<const uint4*>(tw29 + (size_t)idx * NTT29_TW_ROW);
    typename N29M<SH>::Tw r;
    if constexpr (SH != 0) {
        const uint4 a = row[0], b = row[1], c = row[2], d = row[3];
        const uint2 e = *reinterpret_cast<const uint2*>(row + 4);
        r.w[0] = a.x; r.w[1] = a.y; r.w[2] = a.z; r.w[3] = a.w;
        r.w[4] = b.x; r.w[5] = b.y; r.w[6] = b.z; r.w[7] = b.w;
        r.w[8] = c.x; r.q[0] = c.y; r.q[1] = c.z; r.q[2] = c.w;
        r.q[3] = d.x; r.q[4] = d.y; r.q[5] = d.z; r.q[6] = d.w;
        r.q[7] = e.x; r.q[8] = e.y;
    } else {
        const uint4 a = row[0], b = row[1];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        r.v[8] = tw29[(size_t)idx * NTT29_TW_ROW + 8];
    }
    return r;
}

// butterfly of step T + its step twiddles (p8s_compute's structure on F29 values).  `raw_last`: a LAST step with S = 3 leaves its outputs
// un-reduced for the pass's final multiplication / reduction (n29_step8_raw)
// ---- unit step twiddles (r5).  The step twiddle of register j is w_R^((brev3(j) * qlo) << DONE): for qlo = 0 all seven are ONE.  In the radix-8
// step whose qlo has one or two bits (F = 1, 2: the step in front of a pass's last) that is every second / fourth thread -- 7 of its 12 products
// multiply by one.  With the standard thread -> element assignment qlo sits in the low bits of the thread index (lanes of every wave differ); in
// such a step k_ntt_pass29 takes qlo from the TOP bits of the thread index instead (a relabelling of which thread holds which eight elements of
// the tile: the exchanges on either side use the same function), so that it is the same for a whole wave, and a wave with qlo = 0 runs the step
// without twiddles (n29_step8<SH, false>: a table reduction per register instead of a product).  Radix 2^10: 3.5 of a thread's 44 products per
// pass on average, 2^7: 3.5 of 32, 2^8: 1.75 of 34.  BBG_NTT29_UNIT = 0: the standard assignment (A/B).
#ifndef BBG_NTT29_UNIT
#define BBG_NTT29_UNIT 1
#endif
#ifndef BBG_NTT29_UNIT_MIN_LOGR
#define BBG_NTT29_UNIT_MIN_LOGR 9 // the radix <= 2^8 kernels (168 VGPRs for three waves) spill 29 registers around the branch and get 5 % slower
#endif
constexpr bool p29_unit_step(int logR, int T)
{
    const int S = (logR - 3 * T >= 3) ? 3 : (logR - 3 * T), F = (logR - 3 * T >= 3) ? (logR - 3 * (T + 1)) : 0;
    return BBG_NTT29_UNIT && logR >= BBG_NTT29_UNIT_MIN_LOGR && T >= 1 && S == 3 && F >= 1 && F <= 2 && logR - 3 > F;
}
template <int LOGR, bool ROW, int T, int TL> __device__ __forceinline__ void p29_coords(int tid, int& c, int& pbase, int& qlo)
{
    if constexpr (p29_unit_step(LOGR, T)) {
        constexpr int F = LOGR - 3 * (T + 1), LOGW = TL - LOGR, QBITS = LOGR - 3;
        c = tid & ((1 << LOGW) - 1);
        const int q = tid >> LOGW;
        qlo = q >> (QBITS - F); // thread-index bits [TL - 3 - F, TL - 3): above the lane bits, one value per wave
        // which waves of a block have qlo = 0 alternates with the block: the waves of one block sit on four different SIMDs, and two blocks whose
        // unit waves share SIMDs leave the other SIMDs with all the work (measured at 2^20: 0.1074 ms without the alternation, 0.1055 with it,
        // 0.1083 without the unit steps at all)
        qlo ^= (blockIdx.x & 1) ? ((1 << F) - 1) : 0;
        pbase = ((q & ((1 << (QBITS - F)) - 1)) << (F + 3)) | qlo;
    } else {
        p8s_coords<LOGR, ROW, T, TL>(tid, c, pbase, qlo);
    }
}

template <int LOGR, int T> __device__ __forceinline__ void p29_compute(Fr29 (&x)[8], const uint32_t* __restrict__ tw29, int qlo, const uint32_t* red)
{
    constexpr int S = (LOGR - 3 * T >= 3) ? 3 : (LOGR - 3 * T);
    constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int SH = p29_shoup(LOGR);
    using Tw29 = typename N29M<SH>::Tw;
    if constexpr (S == 3) {
        const Tw29 w1 = p29_load_tw<SH>(tw29, 1 << (LOGR - 3)), w2 = p29_load_tw<SH>(tw29, 1 << (LOGR - 2)), w3 = p29_load_tw<SH>(tw29, 3 << (LOGR - 3));
        if constexpr (F > 0) {
            constexpr int DONE = LOGR - 3 - F;
            if constexpr (p29_unit_step(LOGR, T)) {
                // the butterfly is the same for both kinds of wave; only what follows it differs
                n29_step8_raw<SH>(x, w1, w2, w3);
                if (__builtin_amdgcn_readfirstlane(qlo) == 0) { // the whole wave: every step twiddle is one -> a table reduction per register
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        x[j] = ntt29_reduce(x[j], red);
                        asm volatile("" ::: "memory"); // one table row in flight at a time: eight hoisted rows are 96 registers
                    }
                } else {
                    n29_step8_twiddles<SH>(x, [&](int j) { return p29_load_tw<SH>(tw29, (p8_brev3(j) * qlo) << DONE); }, red);
                }
            } else {
                n29_step8<SH, true>(x, w1, w2, w3, [&](int j) { return p29_load_tw<SH>(tw29, (p8_brev3(j) * qlo) << DONE); }, red);
            }
        } else {
            n29_step8_raw<SH>(x, w1, w2, w3);
        }
    } else if constexpr (S == 2) {
        n29_step4<SH>(x, p29_load_tw<SH>(tw29, 1 << (LOGR - 2)));
    } else {
        n29_step2(x);
    }
}

// x: the 8 elements of step T (in place) -> the 8 elements of step T + 1.  Every value that crosses is a valid step input (V < 3, limbs below
// 2^29 + 8): a product, or a reduced register 0.
template <int LOGR, bool ROW, int T, int TL> __device__ __forceinline__ void p29_exchange(Fr29 (&x)[8], uint4* buf, uint32_t* buf8)
{
    constexpr int LOGW = TL - LOGR;
    constexpr int F0 = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int F1 = (LOGR - 3 * (T + 1) >= 3) ? (LOGR - 3 * (T + 2)) : 0;
    int c0, pb0, ql0, c1, pb1, ql1;
    p29_coords<LOGR, ROW, T, TL>(threadIdx.x, c0, pb0, ql0);
    p29_coords<LOGR, ROW, T + 1, TL>(threadIdx.x, c1, pb1, ql1);
#if defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 1) // timing experiment only (wrong results): no exchange at all
    return;
#endif
    const int s0 = p8_addr(pb0, c0, LOGW), s1 = p8_addr(pb1, c1, LOGW); // slots of register 0 on the way out / on the way in
    if (T > 0) __syncthreads(); // everybody has taken limbs 4..7 of the previous exchange out of the buffer
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int a = p8_reg_slot<F0, LOGW>(s0, pb0, c0, j);
        buf[a] = make_uint4(x[j].v[0], x[j].v[1], x[j].v[2], x[j].v[3]);
        buf8[a] = x[j].v[8];
    }
    __syncthreads();
    uint4 lo[8];
    uint32_t top[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int a = p8_reg_slot<F1, LOGW>(s1, pb1, c1, j);
        lo[j] = buf[a];
        top[j] = buf8[a];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) buf[p8_reg_slot<F0, LOGW>(s0, pb0, c0, j)] = make_uint4(x[j].v[4], x[j].v[5], x[j].v[6], x[j].v[7]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint4 h = buf[p8_reg_slot<F1, LOGW>(s1, pb1, c1, j)];
        x[j].v[0] = lo[j].x; x[j].v[1] = lo[j].y; x[j].v[2] = lo[j].z; x[j].v[3] = lo[j].w;
        x[j].v[4] = h.x; x[j].v[5] = h.y; x[j].v[6] = h.z; x[j].v[7] = h.w;
        x[j].v[8] = top[j];
    }
}

// single-round form of p29_exchange (BBG_NTT29_EXCH1): lo = limbs 0..3, hi = limbs 4..7, top = limb 8, one slot index for all three
template <int LOGR, bool ROW, int T, int TL> __device__ __forceinline__ void p29_exchange1(Fr29 (&x)[8], uint4* lo, uint4* hi, uint32_t* top)
{
#if defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 1)
    return;
#endif
    constexpr int LOGW = TL - LOGR;
    constexpr int F0 = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
    constexpr int F1 = (LOGR - 3 * (T + 1) >= 3) ? (LOGR - 3 * (T + 2)) : 0;
    int c0, pb0, ql0, c1, pb1, ql1;
    p29_coords<LOGR, ROW, T, TL>(threadIdx.x, c0, pb0, ql0);
    p29_coords<LOGR, ROW, T + 1, TL>(threadIdx.x, c1, pb1, ql1);
    if (T > 0) __syncthreads(); // everybody has read the previous exchange's elements
    const int e0 = p29_slot(pb0, c0, LOGW), e1 = p29_slot(pb1, c1, LOGW);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int a = ((1 << (F0 + LOGW)) % 128 == 0) ? e0 + (j << (F0 + LOGW)) : p29_slot(pb0 | (j << F0), c0, LOGW); // the swizzle sees bits 3..6 only
        lo[a] = make_uint4(x[j].v[0], x[j].v[1], x[j].v[2], x[j].v[3]);
        hi[a] = make_uint4(x[j].v[4], x[j].v[5], x[j].v[6], x[j].v[7]);
        top[a] = x[j].v[8];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int a = ((1 << (F1 + LOGW)) % 128 == 0) ? e1 + (j << (F1 + LOGW)) : p29_slot(pb1 | (j << F1), c1, LOGW);
        const uint4 l = lo[a], h = hi[a];
        x[j].v[0] = l.x; x[j].v[1] = l.y; x[j].v[2] = l.z; x[j].v[3] = l.w;
        x[j].v[4] = h.x; x[j].v[5] = h.y; x[j].v[6] = h.z; x[j].v[7] = h.w;
        x[j].v[8] = top[a];
    }
}

// Two waves per SIMD (no register cap); the eight output multipliers are fetched where they are used.  Measured (profiles/r04_ntt29_ab.txt,
// isolated fft, ms): capping the kernel at 168 VGPRs for three waves gives the same times with 23-46 spills (2^20 0.1193 vs 0.1190); fetching
// the multipliers behind the data loads as k_ntt_pass8 does -- 64 registers held through the whole pass -- is SLOWER (2^20 0.1236 vs 0.1187,
// 2^22 0.508 vs 0.455): with 9-word elements the registers are worth more than the latency they would hide.
#ifndef BBG_NTT29_PREFETCH
#define BBG_NTT29_PREFETCH 0 // 1 = multipliers fetched at the start (64 registers held through the pass), 0 = fetched where they are used
#endif
template <int LOGR, bool ROW, int TL = P8_TILE_LOG> __global__ void __launch_bounds__(1 << (TL - 3), p29_occ(LOGR)) k_ntt_pass29(PassParams p)
{
    extern __shared__ uint4 lds[];
    BBG_NTT_SELECT_BATCH(p);
    constexpr bool EX1 = p29_exch1(LOGR);
    uint4* buf = lds;                                                                      // limbs 0..3 (single round) / the plane buffer
    uint4* bufhi = lds + (1 << TL);                                                        // limbs 4..7 (single round only)
    uint32_t* buf8 = reinterpret_cast<uint32_t*>(lds + (EX1 ? (2 << TL) : p8_plane<TL>())); // limb 8
    uint32_t* red = buf8 + (EX1 ? (1 << TL) : p8_plane<TL>());                             // 16-byte aligned either way
#define P29_EXCHANGE(T)                                                                      \
    do {                                                                                   \
        if constexpr (EX1) p29_exchange1<LOGR, ROW, T, TL>(x, buf, bufhi, buf8);           \
        else p29_exchange<LOGR, ROW, T, TL>(x, buf, buf8);                                 \
    } while (0)
    constexpr int NSTEPS = (LOGR + 2) / 3;
    constexpr int LOGW = TL - LOGR;
    if (threadIdx.x < NTT29_RED_ROWS) ntt29_fill_reduce_table(red, threadIdx.x);
#if defined(BBG_NTT29_STAGGER_TICKS) // timing experiment: half of the blocks start late (ticks of the 100 MHz constant clock)
    {
#if BBG_NTT29_STAGGER_MODE == 1
        const bool late = blockIdx.x & 1;
#elif BBG_NTT29_STAGGER_MODE == 3 // the second half of the FIRST round of blocks only: later rounds inherit the phase shift
        const bool late = blockIdx.x >= 256 && blockIdx.x < 512;
#else
        const bool late = (blockIdx.x >> 8) & 1;
#endif
        if (late) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < BBG_NTT29_STAGGER_TICKS) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif
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
    const uint32_t* tw29 = p.tw_radix29; // w_R^x R' mod p, x < R: rows of 9 limbs
    Fr29 x[8];
    const Fr* mul_table = ROW ? p.post : p.tw_inter;
    const bool have_outmul = mul_table != nullptr;
    // ---- step 0: the tile from global memory (the addressing of k_ntt_pass8), re-limbed; the first pass's coset factor as a product
    int c, pbase, qlo;
    p29_coords<LOGR, ROW, 0, TL>(threadIdx.x, c, pbase, qlo);
    {
        constexpr int F = (LOGR >= 3) ? (LOGR - 3) : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pj = pbase | (j << F);
            size_t g;
            if (!ROW) {
                g = base + ((size_t)pj << p.logS) + c;
                if (g >= p.in_count) {
#pragma unroll
                    for (int i = 0; i < 9; i++) x[j].v[i] = 0;
                    continue;
                }
#if defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 2) // timing experiment only: no global loads
                { Fr t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)g * 2654435761u + i; t.v[7] &= 0x0fffffffu; x[j] = f29_from_fe<FrP, 0>(t); }
#else
                x[j] = f29_from_fe<FrP, 0>(fe_load<FrP>(p.in + g));
#endif
                if (p.pre && g < p.pre_count) x[j] = f29_mul(x[j], f29_from_fe<FrP, 5>(fe_load<FrP>(p.pre + g))); // V < 2 * 64 / 169 + 1
                continue;
            }
            g = (((((d1_0 + c) << logRestCount) + rest)) << LOGR) + pj;
#if defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 2)
            { Fr t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)g * 2654435761u + i; t.v[7] &= 0x0fffffffu; x[j] = f29_from_fe<FrP, 0>(t); }
#else
            x[j] = f29_from_fe<FrP, 0>(fe_load<FrP>(p.in + g));
#endif
        }
    }
    Fr outmul[8];
    if (BBG_NTT29_PREFETCH == 1 && have_outmul) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int pj, cc;
            p8_last_coords<LOGR, TL>(threadIdx.x, j, pj, cc);
            outmul[j] = fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, cc, base, lo0, d1_0, rest, true));
        }
    }
    // BBG_NTT29_PREFETCH == 2 (round 5): the multipliers are requested in front of the LAST exchange where the last step is a cheap one
    // (radix 2 or 4: log-radix 10, 8, 7 ...): 64 registers held through an exchange and a few butterflies, not through the whole pass
    constexpr bool LATE_PREFETCH = BBG_NTT29_PREFETCH == 2 && NSTEPS > 1 && (LOGR % 3) != 0;
    auto fetch_outmul = [&]() {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int pj, cc;
            p8_last_coords<LOGR, TL>(threadIdx.x, j, pj, cc);
            outmul[j] = fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, cc, base, lo0, d1_0, rest, true));
        }
    };
    __syncthreads(); // the reduction table is complete
    p29_compute<LOGR, 0>(x, tw29, qlo, red);
    if constexpr (NSTEPS > 1) {
        if constexpr (LATE_PREFETCH && NSTEPS == 2) { if (have_outmul) fetch_outmul(); }
        P29_EXCHANGE(0);
        p29_coords<LOGR, ROW, 1, TL>(threadIdx.x, c, pbase, qlo);
        p29_compute<LOGR, 1>(x, tw29, qlo, red);
    }
    if constexpr (NSTEPS > 2) {
        if constexpr (LATE_PREFETCH && NSTEPS == 3) { if (have_outmul) fetch_outmul(); }
        P29_EXCHANGE(1);
        p29_coords<LOGR, ROW, 2, TL>(threadIdx.x, c, pbase, qlo);
        p29_compute<LOGR, 2>(x, tw29, qlo, red);
    }
    if constexpr (NSTEPS > 3) {
        if constexpr (LATE_PREFETCH && NSTEPS == 4) { if (have_outmul) fetch_outmul(); }
        P29_EXCHANGE(2);
        p29_coords<LOGR, ROW, 3, TL>(threadIdx.x, c, pbase, qlo);
        p29_compute<LOGR, 3>(x, tw29, qlo, red);
    }
    // ---- the last step's elements: final multiplier (inter-pass twiddle / post table), reduction, back to 8 words, out (bit reversal in the index)
    {
        constexpr int T = NSTEPS - 1;
        constexpr int F = (LOGR - 3 * T >= 3) ? (LOGR - 3 * (T + 1)) : 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int pj = pbase | (j << F);
            Fr v;
#if defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 2) // timing experiment only: no twiddle loads, a store nobody executes
            if (have_outmul) { Fr t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)pj * 40503u + i + c; t.v[7] &= 0x0fffffffu; v = n29_finish_mul(x[j], t, red); }
            else v = n29_finish(x[j], red);
            if (v.v[0] == 0x12345678u && v.v[1] == 0x9abcdef0u && v.v[5] == 77u) fe_store<FrP>(p.out + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, false), v);
#elif defined(BBG_NTT29_EXP) && (BBG_NTT29_EXP & 4) // timing experiment only (round 6, wrong results): the output multipliers made up in registers -- what the
            // inter-pass twiddle / post tables cost a transform, i.e. the most ANY on-chip generation could gain before its own products are paid for
            if (have_outmul) { Fr t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)pj * 40503u + i + c; t.v[7] &= 0x0fffffffu; v = n29_finish_mul(x[j], t, red); }
            else v = n29_finish(x[j], red);
            fe_store<FrP>(p.out + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, false), v);
#else
            if (have_outmul)
                v = n29_finish_mul(x[j], (BBG_NTT29_PREFETCH == 1 || LATE_PREFETCH) ? outmul[j] : fe_load<FrP>(mul_table + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, true)), red);
            else v = n29_finish(x[j], red);
            fe_store<FrP>(p.out + p8_out_index<LOGR, ROW>(p, pj, c, base, lo0, d1_0, rest, false), v);
#endif
        }
    }
}

#undef P29_EXCHANGE

template <int LOGR, int TL = P8_TILE_LOG> static void p29_launch(const PassParams& p, size_t tiles, hipStream_t st)
{
    constexpr size_t lds = p29_lds_bytes<TL, LOGR>();
    if (p.row_pass) hipLaunchKernelGGL((k_ntt_pass29<LOGR, true, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), lds, st, p);
    else hipLaunchKernelGGL((k_ntt_pass29<LOGR, false, TL>), dim3((unsigned)tiles, (unsigned)(p.batch > 1 ? p.batch : 1)), dim3(1 << (TL - 3)), lds, st, p);
}
template <int LOGR, int TL = P8_TILE_LOG> static hipError_t p29_attr()
{
    hipError_t e = hipFuncSetAttribute((const void*)k_ntt_pass29<LOGR, true, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(p29_lds_bytes<TL, LOGR>()));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_ntt_pass29<LOGR, false, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(p29_lds_bytes<TL, LOGR>()));
}

} // namespace bbg
