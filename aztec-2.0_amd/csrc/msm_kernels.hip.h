// Templated kernels and the per-width driver of the bucket MSM (see msm.hip for the overview).  Included by msm_wNN.hip only: every
// window width is its own translation unit.
#pragma once
#include "msm_cfg.h"
#include "curve.hip.h"
#include "curve_quad.hip.h"
#include "curve29.hip.h"

#include <algorithm>
#include <cstring>
#include <type_traits>
#ifdef BBG_ROCPRIM_SORT // A/B build only (make ROCPRIM_SORT=1): k_recode + rocPRIM radix sort + k_offsets instead of the partition sort
#include <rocprim/device/device_radix_sort.hpp>
#endif

namespace bbg {

static int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// ---------------------------------------------------------------------------------- SRS precomputation
// table[w * n + i] = 2^(table_offset(w)) * P_i (affine, canonical).  One thread per point: doublings in XYZZ from one window to
// the next, then one shared inversion (Montgomery's trick over the Z-products) to normalise.
template <int C> __global__ void __launch_bounds__(128) k_precompute_tables(const Affine* __restrict__ points, Affine* table, size_t n)
{
    using K = MsmCfg<C>;
    constexpr int MSM_WINDOWS = K::windows;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine p = aff_load(points + i);
    if (aff_is_inf(p)) {
        for (int w = 0; w < MSM_WINDOWS; w++) aff_store(table + (size_t)w * n + i, aff_inf());
        return;
    }
    p.x = fe_reduce_once(p.x);
    p.y = fe_reduce_once(p.y);
    aff_store(table + i, p);
    // the multiples live in per-thread scratch (one-off kernel: simplicity over registers)
    Xyzz pts[MSM_WINDOWS - 1];
    Fq prod[MSM_WINDOWS - 1];
    Xyzz q = xyzz_dbl_affine(p); // 2^1 P
    int at = 1;
    Fq acc = Fq::one();
    for (int w = 1; w < MSM_WINDOWS; w++) {
        for (; at < K::table_offset(w); at++) q = xyzz_dbl(q);
        pts[w - 1] = q;
        prod[w - 1] = acc;
        acc = fe_mul(acc, fe_mul(q.zz, q.zzz));
    }
    Fq inv = fq_invert(acc);
    for (int w = MSM_WINDOWS - 1; w >= 1; w--) {
        const Xyzz& t = pts[w - 1];
        Fq iz = fe_mul(inv, prod[w - 1]); // 1 / (ZZ * ZZZ) of point w
        inv = fe_mul(inv, fe_mul(t.zz, t.zzz));
        Affine a;
        a.x = fe_reduce_once(fe_mul(t.x, fe_mul(iz, t.zzz))); // X / ZZ
        a.y = fe_reduce_once(fe_mul(t.y, fe_mul(iz, t.zz)));  // Y / ZZZ
        aff_store(table + (size_t)w * n + i, a);
    }
}

// ---------------------------------------------------------------------------------- scalar recoding
// scalar (Montgomery, any rep < 2^256) -> canonical integer k -> signed digits d_w, |d_w| <= 2^(width(w) - 1), k = sum d_w 2^(offset(w)).
// Replaces compute_wnaf_states + fixed_wnaf_with_counts (scalar_multiplication.cpp:188-252, wnaf.hpp:230-283):
// same idea (signed windows halve the bucket count), but plain signed digits with carry instead of the
// odd-digit + skew form, and zero digits produce no work.
// from_montgomery = Montgomery product with the integer 1: (s + m*r) / 2^256 <= r for ANY 256-bit s, so one conditional
// subtraction canonicalises (from_montgomery_form, field_impl.hpp:245-255).  Top window: k < 2^254 and the windows cover 255 bits, so
// its digit plus the incoming carry is at most 2^(width - 1): never negative, never a carry out.
template <int C>
__device__ __forceinline__ void recode_digits(const Fr* __restrict__ scalars, size_t i, uint32_t (&mag)[MSM_MAX_WINDOWS], uint32_t& signs)
{
    using K = MsmCfg<C>;
    const Fr k = fe_from_mont(fe_load<FrP>(scalars + i)); // canonical integer < r < 2^254
    uint32_t carry = 0;
    signs = 0;
#pragma unroll
    for (int w = 0; w < K::windows; w++) {
        const uint32_t FULL = 1u << K::width(w), HALF = FULL >> 1;
        const int bit = K::offset(w), limb = bit >> 5, sh = bit & 31;
        uint64_t two = k.v[limb];
        if (limb + 1 < 8) two |= (uint64_t)k.v[limb + 1] << 32;
        const uint32_t d = ((uint32_t)(two >> sh) & (FULL - 1)) + carry; // 0 .. 2^width
        const uint32_t neg = d > HALF;
        mag[w] = (neg ? (FULL - d) : d) << K::scale(w); // bucket: |digit| in [0, 2^(width-1)], doubled for the narrow windows (msm_cfg.h)
        carry = neg;
        signs |= neg << w;
    }
}
#ifdef BBG_ROCPRIM_SORT
template <int C>
__global__ void __launch_bounds__(256) k_recode(const Fr* __restrict__ scalars, size_t n, size_t from, uint32_t* keys, uint32_t* vals)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t mag[MSM_MAX_WINDOWS], signs;
    recode_digits<C>(scalars, i, mag, signs);
#pragma unroll
    for (int w = 0; w < MsmCfg<C>::windows; w++) {
        keys[(size_t)w * n + i] = mag[w];
        vals[(size_t)w * n + i] = (((signs >> w) & 1u) << 31) | ((uint32_t)w << MsmCfg<C>::idx_bits) | (uint32_t)(from + i);
    }
}

#endif

// ---------------------------------------------------------------------------------- fused recode + two-level partition sort
// Replaces k_recode + the 2-pass library radix sort + k_offsets (0.42 ms at 2^20) with an MSD partition that exploits what
// the accumulation needs: runs per bucket in ANY order.  Bucket id mag in [0, 2^15] splits into hi = mag >> 5 (1025
// partitions) and lo = mag & 31.
//   k_sortA_count   : digits of 1024 scalars per block, LDS histogram of hi, one global add per (block, partition)
//   k_sortA_scan    : exclusive scan of the partition sizes
//   k_sortA_scatter : digits again; each block reserves a contiguous range in every partition with ONE global atomic,
//                     groups its entries by partition in LDS and writes them out coalesced; entry = (lo << 32) | value
//   k_sortB         : one block per partition: LDS histogram of lo, writes the bucket offsets of its 32 buckets and
//                     moves the 32-bit values to their final position (through LDS when the partition fits)
// Digits are recomputed instead of stored (one Montgomery product per scalar is cheaper than 2 x 34 MiB of traffic).
// Order inside a bucket is arbitrary (atomics), so the Jacobian REPRESENTATIVE of an MSM result may differ between runs;
// the point it denotes does not (the reference's representative likewise depends on its thread count).
constexpr int SORT_PAD = 2048;   // partition table size (power of two >= MsmCfg::parts = 1025)
constexpr int SORT_BLOCK = 1024; // scalars per block in the A kernels

// A BATCH of MSMs over one SRS that travels through ONE launch set (bbg_msm_batch*, the reference's unit of work is a round's queue of
// commitments: prover.cpp:66-74, :120-135, work_queue.hpp:208-282).  MSM k's entries are filed under bucket set k: global bucket number
// k * 2^(C-1) + |digit|, partition tables / rows / columns / planes per set.  blockIdx.y selects the set in the kernels that work per MSM
// (the sort's first level, row / column sums, bit planes); the accumulation and the combine kernels see one long bucket array.  A single
// MSM is a batch of one: the same kernels, the same code path.
constexpr int MSM_BATCH_MAX = BBG_MSM_BATCH_MAX;
struct MsmBatch {
    const Fr* scalars[MSM_BATCH_MAX];
    uint32_t n[MSM_BATCH_MAX];
    uint32_t from[MSM_BATCH_MAX];
};

// LDS counter bump that stays fast when a whole wave hits one counter (all-equal scalars): one atomic per wave then.
// Inactive lanes are masked off (no traffic); returns the lane's rank within the counter.
__device__ __forceinline__ uint32_t lds_take(uint32_t* ctr, uint32_t key, bool active)
{
    uint32_t r = 0;
    if (active) {
        const uint64_t act = __ballot(1);
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(key);
        if (__ballot(key == k0) == act) {
            const int lane = threadIdx.x & 63;
            const uint32_t below = (uint32_t)__popcll(act & ((1ull << lane) - 1));
            uint32_t b = 0;
            if (below == 0) b = atomicAdd(&ctr[k0], (uint32_t)__popcll(act));
            r = __builtin_amdgcn_readfirstlane(b) + below;
        } else {
            r = atomicAdd(&ctr[key], 1u);
        }
    }
    return r;
}

// The same for a counter array of a FEW bins (the second sort level of the 13-bit configuration: 4 bins per partition): with a thousand
// entries on four counters the per-lane atomics of lds_take serialise (measured: a 2^16-term MSM 0.81 ms against 0.34 with 16-bit windows).
// Here every bin is counted with one ballot per wave and reserved with ONE atomic per wave and bin; the lane's rank is its position among
// the wave's lanes of the same bin.
template <int BINS> __device__ __forceinline__ uint32_t lds_take_few(uint32_t* ctr, uint32_t key, bool active)
{
    const int lane = threadIdx.x & 63;
    const uint64_t below_mask = (1ull << lane) - 1;
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < BINS; b++) {
        const uint64_t m = __ballot(active && key == (uint32_t)b);
        if (m) { // wave-uniform
            uint32_t base = 0;
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&ctr[b], (uint32_t)__popcll(m));
            base = __shfl(base, (int)__builtin_ctzll(m));
            if (active && key == (uint32_t)b) r = base + (uint32_t)__popcll(m & below_mask);
        }
    }
    return r;
}
template <int BINS> __device__ __forceinline__ uint32_t lds_take_bins(uint32_t* ctr, uint32_t key, bool active)
{
    if constexpr (BINS <= 8) return lds_take_few<BINS>(ctr, key, active);
    else return lds_take(ctr, key, active);
}

// exclusive scan of tbl[0 .. SORT_PAD) by a 1024-thread block, two adjacent entries per thread; returns the pair's
// exclusive prefixes.  wsum = 16 words of LDS scratch.
__device__ __forceinline__ void block_scan_pairs(const uint32_t* tbl, uint32_t* wsum, uint32_t& excl0, uint32_t& c0, uint32_t& c1)
{
    const int tid = threadIdx.x;
    c0 = tbl[2 * tid];
    c1 = tbl[2 * tid + 1];
    uint32_t incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < (tid >> 6); k++) before += wsum[k];
    excl0 = before + incl - (c0 + c1);
}

// 256-thread blocks (four scalars per thread), not SORT_BLOCK-sized ones: this kernel starts while the previous MSM's row / column sums still
// hold three waves per SIMD, and a 1024-thread block needs four free wave slots on EVERY SIMD of a CU at once -- it waited for them
// (134-160 us per launch in the step's timeline against 29 us alone, profiles/r03_timeline_a.txt); one 49-VGPR wave per SIMD fits beside them.
constexpr int COUNT_THREADS = 256;
constexpr int COUNT_MAX_BLOCKS = 2048; // 8 blocks per CU
// STRIDE = false (n <= 2^21: one group per block) keeps the kernel at 56 VGPRs -- a wave of it then fits beside the three 152-VGPR waves per SIMD of
// the previous MSM's row / column sums (3 x 152 + 56 = 512); the loop costs four more registers, which is 0.1 ms of waiting per bench step.
template <int C, bool STRIDE> __global__ void __launch_bounds__(COUNT_THREADS) k_sortA_count(const MsmBatch batch, uint32_t* part_count)
{
    constexpr int MSM_WINDOWS = MsmCfg<C>::windows, SORT_LO_BITS = MsmCfg<C>::lo_bits, SORT_PARTS = MsmCfg<C>::parts;
    __shared__ uint32_t hist[SORT_PAD];
    const int tid = threadIdx.x;
    const int set = blockIdx.y;
    const Fr* __restrict__ scalars = batch.scalars[set];
    const size_t n = batch.n[set];
    if (!STRIDE && (size_t)blockIdx.x * SORT_BLOCK >= n) return; // a shorter MSM of the batch: whole blocks leave before any barrier
    part_count += set * SORT_PAD;
    for (int h = tid; h < SORT_PAD; h += COUNT_THREADS) hist[h] = 0;
    __syncthreads();
    // grid-stride over groups of SORT_BLOCK scalars: a large MSM is counted by COUNT_MAX_BLOCKS blocks, each flushing its histogram ONCE
    // (one block per group meant 16 384 x 1 025 global atomics onto the same 1 025 words at n = 2^24: 0.37 ms)
    auto count_group = [&](size_t g) {
        for (int r = 0; r < SORT_BLOCK / COUNT_THREADS; r++) {
            const size_t i = g * SORT_BLOCK + (size_t)r * COUNT_THREADS + tid;
            uint32_t mag[MSM_MAX_WINDOWS], signs;
            if (i < n) recode_digits<C>(scalars, i, mag, signs);
#pragma unroll
            for (int w = 0; w < MSM_WINDOWS; w++) {
                const bool on = i < n && mag[w] != 0; // zero digits contribute nothing: never sorted
                lds_take(hist, on ? mag[w] >> SORT_LO_BITS : 0u, on);
            }
        }
    };
    if constexpr (STRIDE) {
        const size_t groups = (n + SORT_BLOCK - 1) / SORT_BLOCK;
        for (size_t g = blockIdx.x; g < groups; g += gridDim.x) count_group(g);
    } else {
        count_group(blockIdx.x);
    }
    __syncthreads();
    for (int h = tid; h < SORT_PARTS; h += COUNT_THREADS)
        if (hist[h]) atomicAdd(&part_count[h], hist[h]);
}
// part_base[h] = sum_{h' < h} count[h'] ; cursor[h] = part_base[h] ; offsets[MSM_BUCKETS + 1] = total
// The counters are cleared again once they have been read (and the call's long-bucket counter with them), so that no MSM needs a memset of
// its own in front of the counting pass: the runtime's fill kernel cost 20-34 us of the main stream's critical path per MSM (r03 timeline).
template <int C>
__global__ void __launch_bounds__(1024) k_sortA_scan(uint32_t* part_count, uint32_t* part_base, uint32_t* cursor, uint32_t* offsets, uint32_t* long_count, int sets)
{
    constexpr int SORT_PARTS = MsmCfg<C>::parts, MSM_BUCKETS = MsmCfg<C>::buckets;
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x;
    const int h0 = 2 * tid, h1 = 2 * tid + 1;
    if (tid == 0) long_count[0] = long_count[1] = 0; // the long-bucket queue and the redo queue (its count is the word behind) start empty
    uint32_t carry = 0; // entries of the sets before this one: set k's partitions follow set k-1's in the entry array
    for (int k = 0; k < sets; k++) {
        uint32_t excl, c0, c1;
        block_scan_pairs(part_count + k * SORT_PAD, wsum, excl, c0, c1); // part_count[SORT_PARTS ..) is zero
        excl += carry;
        part_count[k * SORT_PAD + h0] = 0;
        part_count[k * SORT_PAD + h1] = 0;
        if (h0 <= SORT_PARTS) part_base[k * SORT_PAD + h0] = excl; // part_base[SORT_PARTS] = entries up to and including this set
        if (h1 <= SORT_PARTS) part_base[k * SORT_PAD + h1] = excl + c0;
        if (h0 < SORT_PARTS) cursor[k * SORT_PAD + h0] = excl;
        if (h1 < SORT_PARTS) cursor[k * SORT_PAD + h1] = excl + c0;
        if (h0 == SORT_PARTS) carry_s = excl;
        if (h1 == SORT_PARTS) carry_s = excl + c0;
        __syncthreads(); // also: wsum is free again
        carry = carry_s;
    }
    if (tid == 0) offsets[(size_t)sets * MSM_BUCKETS + 1] = carry;
}
// The block's <= 16 Ki entries are first grouped by partition in LDS (values + bucket ids, 128 KiB) and then written
// out with consecutive threads on consecutive addresses: every (block, partition) chunk is one contiguous burst instead
// of independent 8-byte stores issued at random times.
template <int C> __global__ void __launch_bounds__(SORT_BLOCK)
k_sortA_scatter(const MsmBatch batch, uint32_t* cursor, uint64_t* entries)
{
    constexpr int MSM_WINDOWS = MsmCfg<C>::windows, SORT_LO_BITS = MsmCfg<C>::lo_bits, SORT_PARTS = MsmCfg<C>::parts;
    constexpr int CAP = SORT_BLOCK * MSM_WINDOWS;
    const int set = blockIdx.y;
    const Fr* __restrict__ scalars = batch.scalars[set];
    const size_t n = batch.n[set], from = batch.from[set];
    if ((size_t)blockIdx.x * SORT_BLOCK >= n) return; // a shorter MSM of the batch
    cursor += set * SORT_PAD;
    __shared__ uint32_t hist[SORT_PAD];   // per-partition count, then rank counter
    __shared__ uint32_t lstart[SORT_PAD]; // first LDS slot of each partition
    __shared__ uint32_t gbase[SORT_PAD];  // this block's first global slot in each partition, minus lstart
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t st_val[CAP];
    // bucket ids fit 16 bits where there are more than 16 windows (C <= 15: ids <= 2^14), and must: 20 windows x 1024 scalars x 8 bytes would be
    // the whole LDS of a CU
    using StMag = typename std::conditional<(MSM_WINDOWS > 16), uint16_t, uint32_t>::type;
    static_assert(MSM_WINDOWS <= 16 || MsmCfg<C>::buckets < 65536, "16-bit staging of bucket ids");
    __shared__ StMag st_mag[CAP];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    hist[tid + 1024] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * SORT_BLOCK + tid;
    uint32_t mag[MSM_MAX_WINDOWS], signs = 0;
    if (i < n) recode_digits<C>(scalars, i, mag, signs);
    // ONE round of LDS atomics (r4): the value a counting atomic returns IS the entry's rank inside its partition; it waits in a register while
    // the counts are scanned (a second, ranking round of 14 atomics per scalar cost a quarter of the kernel)
    uint32_t rank[MSM_MAX_WINDOWS];
#pragma unroll
    for (int w = 0; w < MSM_WINDOWS; w++) {
        const bool on = i < n && mag[w] != 0;
        rank[w] = lds_take(hist, on ? mag[w] >> SORT_LO_BITS : 0u, on);
    }
    __syncthreads();
    uint32_t excl, c0, c1;
    block_scan_pairs(hist, wsum, excl, c0, c1);
    __syncthreads();
    {
        const int h0 = 2 * tid, h1 = 2 * tid + 1;
        lstart[h0] = excl;
        lstart[h1] = excl + c0;
        gbase[h0] = (c0 ? atomicAdd(&cursor[h0], c0) : 0u) - excl; // one global reservation per (block, partition)
        gbase[h1] = (c1 ? atomicAdd(&cursor[h1], c1) : 0u) - (excl + c0);
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < MSM_WINDOWS; w++) {
        const bool on = i < n && mag[w] != 0;
        const uint32_t h = on ? mag[w] >> SORT_LO_BITS : 0u;
        if (on) {
            const uint32_t slot = lstart[h] + rank[w];
            st_val[slot] = (((signs >> w) & 1u) << 31) | ((uint32_t)w << MsmCfg<C>::idx_bits) | (uint32_t)(from + i);
            st_mag[slot] = (StMag)mag[w];
        }
    }
    __syncthreads();
    const uint32_t total = lstart[SORT_PARTS - 1] + hist[SORT_PARTS - 1];
    for (uint32_t slot = tid; slot < total; slot += SORT_BLOCK) {
        const uint32_t m = st_mag[slot];
        entries[gbase[m >> SORT_LO_BITS] + slot] = ((uint64_t)(m & ((1u << SORT_LO_BITS) - 1)) << 32) | st_val[slot];
    }
}

// One block per partition.  Fast path (partition <= SORTB_CAP entries, the normal case up to n = 2^20): entries are
// read once into registers, ranked with LDS counters, staged in LDS in bucket order and written out coalesced.  Larger
// partitions (bigger n, skewed digits) go through LDS in chunks and leave as one run per bucket and chunk.
// BINS = 2^lo_bits buckets per partition: up to 1024 one per thread, above (C = 22: 2048) BPT consecutive bins per thread.
constexpr int SORTB_PER_THREAD = 18;
constexpr int SORTB_CAP = SORTB_PER_THREAD * 1024; // 72 KiB of staging (+ the counters): two blocks per CU up to 1024 bins
constexpr int SORTB_UNROLL = 8;

// exclusive scan of cnt[0 .. BINS) into excl[0 .. BINS) by a 1024-thread block (thread t owns bins t*BPT .. t*BPT + BPT - 1); wsum = 16 words
template <int BINS, int TPB> __device__ __forceinline__ void block_scan_bins(const uint32_t* cnt, uint32_t* excl, uint32_t* wsum)
{
    constexpr int BPT = BINS > TPB ? BINS / TPB : 1;
    const int tid = threadIdx.x;
    uint32_t c[BPT], tot = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        const int b = tid * BPT + j;
        c[j] = b < BINS ? cnt[b] : 0u;
        tot += c[j];
    }
    uint32_t incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - tot;
    for (int k = 0; k < (tid >> 6); k++) run += wsum[k];
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        const int b = tid * BPT + j;
        if (b < BINS) excl[b] = run;
        run += c[j];
    }
}

// TPB threads per block: 1024, or 256 for small MSMs (partitions of a few hundred entries: a 1024-thread block per partition spends its time
// in barriers and scans -- 25 us of a 2^12-term MSM's 270).
template <int C, int TPB> __global__ void __launch_bounds__(TPB)
k_sortB(const uint64_t* __restrict__ entries, const uint32_t* __restrict__ part_base, uint32_t* offsets, uint32_t* svals)
{
    constexpr int SORT_LO_BITS = MsmCfg<C>::lo_bits, MSM_BUCKETS = MsmCfg<C>::buckets, BINS = 1 << SORT_LO_BITS;
    constexpr uint32_t SORT_LO_MASK = BINS - 1;
    // LDS budget: TWO blocks per CU (<= 80 KB each) whatever the bin count -- in the chunked path a block alternates between global loads and
    // LDS work, and a second block on the CU is what covers either.  Stage = PER entries per thread: 18 up to 1024 bins (72 KB + 8 KB of counters),
    // 15 for 2048 bins (60 KB + 16 KB); the chunked path's per-chunk bin starts live in the stage's tail instead of an array of their own.
    constexpr int PER = BINS > 1024 ? 15 : SORTB_PER_THREAD;
    constexpr int CAP = PER * TPB; // fast-path capacity and LDS stage
    constexpr int UNROLL = (CAP - BINS) / (2 * TPB) < SORTB_UNROLL ? (CAP - BINS) / (2 * TPB) : SORTB_UNROLL; // entries per thread and chunk
    static_assert(BINS <= 2048 && BINS <= 8 * TPB, "LDS: three counter arrays of BINS words beside the stage; at most 8 bins per thread");
    __shared__ uint32_t hist[BINS];
    __shared__ uint32_t off[BINS];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t stage[CAP];
    const int tid = threadIdx.x;
    const uint32_t h = blockIdx.x, set = blockIdx.y;
    const uint32_t pb = part_base[set * SORT_PAD + h], pe = part_base[set * SORT_PAD + h + 1];
    const uint32_t len = pe - pb;
    const bool fast = len <= (uint32_t)CAP;
    for (int b = tid; b < BINS; b += TPB) hist[b] = 0;
    __syncthreads();
    uint64_t e[PER];
    uint32_t rk[PER]; // fast path: the entry's rank inside its bucket, as the counting atomic returned it
    constexpr uint32_t CHUNK = TPB * UNROLL;
    const uint32_t span = (len + CHUNK - 1) / CHUNK * CHUNK; // whole waves stay in the loops (lds_take is wave-cooperative)
    if (fast) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t q = u * TPB + tid;
            const uint64_t v = entries[pb + (q < len ? q : 0u)]; // unconditional: the loads batch up
            e[u] = q < len ? v : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < PER; u++) rk[u] = lds_take_bins<BINS>(hist, (uint32_t)(e[u] >> 32) & SORT_LO_MASK, e[u] != ~0ull); // count AND rank (r4)
    } else {
        for (uint32_t q0 = 0; q0 < span; q0 += CHUNK) {
            uint32_t key[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint32_t q = q0 + u * TPB + tid;
                const uint32_t k = (uint32_t)(entries[pb + (q < len ? q : 0u)] >> 32);
                key[u] = q < len ? k : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) lds_take_bins<BINS>(hist, key[u] & SORT_LO_MASK, key[u] != 0xffffffffu);
        }
    }
    __syncthreads();
    block_scan_bins<BINS, TPB>(hist, off, wsum); // off[b] = entries of this partition in bins below b
    __syncthreads();
    for (int b = tid; b < BINS; b += TPB) {
        const uint32_t bucket = (h << SORT_LO_BITS) + (uint32_t)b;
        // global bucket number = set * 2^(C-1) + bucket; a later set's bucket 0 (zero digits: never sorted, always empty) would be the
        // previous set's LAST bucket, whose offset that set's own top partition writes
        if (bucket <= (uint32_t)MSM_BUCKETS && (bucket != 0 || set == 0)) offsets[(size_t)set * MSM_BUCKETS + bucket] = pb + off[b];
    }
    __syncthreads();
    if (fast) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (e[u] != ~0ull) stage[off[(uint32_t)(e[u] >> 32) & SORT_LO_MASK] + rk[u]] = (uint32_t)e[u];
        }
        __syncthreads();
        for (uint32_t q = tid; q < len; q += TPB) svals[pb + q] = stage[q];
    } else {
        // Large partition: chunks of 8192 entries are grouped by bucket in LDS and written out as runs (one run per bucket
        // and chunk, on consecutive addresses) instead of 8192 independent 4-byte stores.  off[] = running global position
        // of every bucket; cnt[] / cstart[] = this chunk's counts and their exclusive scan (reusing hist[] and wsum[]).
        uint32_t* cnt = hist;
        uint32_t* cstart = stage + 2 * CHUNK; // the stage's tail
        static_assert(2 * CHUNK + BINS <= (uint32_t)CAP, "chunk staging + per-chunk bin starts must fit the stage");
        uint32_t* stage_val = stage;
        uint32_t* stage_bin = stage + CHUNK;
        for (uint32_t q0 = 0; q0 < span; q0 += CHUNK) {
            uint64_t x[UNROLL];
            uint32_t crk[UNROLL];
            for (int b = tid; b < BINS; b += TPB) cnt[b] = 0;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint32_t q = q0 + u * TPB + tid;
                const uint64_t v = entries[pb + (q < len ? q : 0u)];
                x[u] = q < len ? v : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) crk[u] = lds_take_bins<BINS>(cnt, (uint32_t)(x[u] >> 32) & SORT_LO_MASK, x[u] != ~0ull);
            __syncthreads();
            block_scan_bins<BINS, TPB>(cnt, cstart, wsum); // exclusive scan of this chunk's counts
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (x[u] != ~0ull) {
                    const uint32_t b = (uint32_t)(x[u] >> 32) & SORT_LO_MASK;
                    const uint32_t slot = cstart[b] + crk[u];
                    stage_val[slot] = (uint32_t)x[u];
                    stage_bin[slot] = b;
                }
            }
            __syncthreads();
            const uint32_t chunk_len = len - q0 < CHUNK ? len - q0 : CHUNK;
            for (uint32_t slot = tid; slot < chunk_len; slot += TPB) {
                const uint32_t b = stage_bin[slot];
                svals[pb + off[b] + (slot - cstart[b])] = stage_val[slot];
            }
            __syncthreads();
            for (int b = tid; b < BINS; b += TPB) off[b] += cnt[b];
            __syncthreads();
        }
    }
}

#ifdef BBG_ROCPRIM_SORT
// offsets[b] = first sorted position with key >= b, for b = 0 .. MSM_BUCKETS + 1
template <int C> __global__ void k_offsets(const uint32_t* __restrict__ keys, size_t total, uint32_t* offsets)
{
    constexpr uint32_t MSM_BUCKETS = MsmCfg<C>::buckets;
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > total) return;
    if (e == total) {
        uint32_t last = total ? keys[total - 1] : 0;
        if (total == 0) offsets[0] = 0;
        for (uint32_t b = last + 1; b <= MSM_BUCKETS + 1; b++) offsets[b] = (uint32_t)total;
        return;
    }
    uint32_t k = keys[e];
    if (e == 0) {
        for (uint32_t b = 0; b <= k; b++) offsets[b] = 0;
    } else {
        uint32_t kp = keys[e - 1];
        for (uint32_t b = kp + 1; b <= k; b++) offsets[b] = (uint32_t)e;
    }
}

#endif

// ---------------------------------------------------------------------------------- bucket accumulation
// Load-balanced: the sorted entry array (without the key-0 prefix) is cut into segments of MSM_SEG entries, one lane
// per segment, regardless of bucket boundaries -- every lane does the same number of mixed additions (a per-bucket
// split leaves a wave waiting for its fullest bucket: ~77 % lane efficiency at Poisson(64)).  A lane walks its
// segment; the bucket it is in comes from the offsets table (binary search once, then sequential).  Runs are emitted as
//   head[lane]  : the run containing the segment's first entry (may continue from the previous lane)
//   tail[lane]  : the run containing the segment's last entry, if different from the head run
//   buckets[b]  : runs that start and end strictly inside the segment (complete buckets)
// and k_combine adds head/tail pieces per bucket.  Values address the window tables: point = table[w*n_srs + idx],
// negated when bit 31 is set.
constexpr uint32_t MSM_SEG_MIN = 8, MSM_SEG_DEFAULT = 64; // segment length is chosen per call (msm_seg_len)
constexpr size_t MSM_QUAD_ACC_MAX_LANES = 32768; // k_accumulate_q4 up to this many lane segments (two waves per SIMD of quads)
constexpr int MSM_LONG_SPAN = 48; // buckets spanning more lanes than this are summed by a whole block

template <int C> __device__ __forceinline__ Affine load_entry_point(const Affine* __restrict__ table, size_t n_srs, uint32_t v)
{
    using K = MsmCfg<C>;
    return aff_load(table + (size_t)((v >> K::idx_bits) & ((1u << K::win_bits) - 1)) * n_srs + (v & ((1u << K::idx_bits) - 1)));
}

template <int C> __global__ void __launch_bounds__(256)
k_accumulate(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ offsets, const Affine* __restrict__ table,
             size_t n_srs, uint32_t seg, Xyzz* head, Xyzz* tail, Xyzz* buckets, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    const uint32_t total = offsets[MSM_BUCKETS + 1]; // the partition sort drops zero digits: the count lives on the device
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t base = offsets[1]; // entries with key 0 (zero digits) sort first and are skipped
    const uint64_t s64 = (uint64_t)base + (uint64_t)lane * seg;
    if (s64 >= total) return;
    const uint32_t s = (uint32_t)s64;
    const uint32_t e = (total - s > seg) ? s + seg : total;
    // bucket containing position s: largest b in [1, 2^15] with offsets[b] <= s
    uint32_t lo = 1, hi = MSM_BUCKETS;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= s) lo = mid;
        else hi = mid - 1;
    }
    uint32_t cur = lo;
    uint32_t cur_end = offsets[cur + 1];
    bool first_run = true;
    Xyzz acc = xyzz_inf();
    uint32_t v = vals[s];
    Affine p = load_entry_point<C>(table, n_srs, v);
    for (uint32_t q = s; q < e; q++) {
        if (q == cur_end) { // the run of bucket `cur` ended inside this segment
            if (first_run) xyzz_store(head + lane, acc);
            else xyzz_store(buckets + (cur - 1), acc);
            first_run = false;
            acc = xyzz_inf();
            do {
                cur++;
                cur_end = offsets[cur + 1];
            } while (cur_end <= q); // skip empty buckets
        }
        const uint32_t vc = v;
        const Affine pc = p;
        if (q + 1 < e) { // software prefetch of the next gather
            v = vals[q + 1];
            p = load_entry_point<C>(table, n_srs, v);
        }
        acc = xyzz_madd(acc, aff_neg_if(pc, (vc >> 31) != 0));
    }
    if (first_run) xyzz_store(head + lane, acc);
    else xyzz_store(tail + lane, acc);
}

// The same accumulation with FOUR threads per lane segment (xyzz_madd_q4): identical control flow and results, the chain of dependent mixed
// additions ~2x shorter in time.  Only for small MSMs (few lanes: the chip is idle anyway and the kernel's run time is that chain); a
// quad-cooperative addition costs ~1.5x the instructions of a one-lane one.
template <int C> __global__ void __launch_bounds__(256)
k_accumulate_q4(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ offsets, const Affine* __restrict__ table,
                size_t n_srs, uint32_t seg, Xyzz* head, Xyzz* tail, Xyzz* buckets, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t lane = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int qd = threadIdx.x & 3;
    const uint32_t base = offsets[1];
    const uint64_t s64 = (uint64_t)base + (uint64_t)lane * seg;
    if (s64 >= total) return; // whole quads leave together
    const uint32_t s = (uint32_t)s64;
    const uint32_t e = (total - s > seg) ? s + seg : total;
    uint32_t lo = 1, hi = MSM_BUCKETS;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= s) lo = mid;
        else hi = mid - 1;
    }
    uint32_t cur = lo;
    uint32_t cur_end = offsets[cur + 1];
    bool first_run = true;
    Xyzz acc = xyzz_inf();
    uint32_t v = vals[s];
    Affine p = load_entry_point<C>(table, n_srs, v);
    for (uint32_t q = s; q < e; q++) {
        if (q == cur_end) {
            if (qd == 0) {
                if (first_run) xyzz_store(head + lane, acc);
                else xyzz_store(buckets + (cur - 1), acc);
            }
            first_run = false;
            acc = xyzz_inf();
            do {
                cur++;
                cur_end = offsets[cur + 1];
            } while (cur_end <= q);
        }
        const uint32_t vc = v;
        const Affine pc = p;
        if (q + 1 < e) {
            v = vals[q + 1];
            p = load_entry_point<C>(table, n_srs, v);
        }
        acc = xyzz_madd_q4(acc, aff_neg_if(pc, (vc >> 31) != 0), qd);
    }
    if (qd == 0) {
        if (first_run) xyzz_store(head + lane, acc);
        else xyzz_store(tail + lane, acc);
    }
}

// ---- the accumulation on 29-bit limbs (field29.hip.h, curve29.hip.h): the same segments, runs and outputs as k_accumulate with ~25 % fewer VALU
// instructions per mixed addition.  The complete-addition special cases (P = +-acc) are not tested per addition; a run that met one ends with
// ZZ = 0 (mod p) and its bucket is queued for k_redo, which recomputes the whole bucket from its entries after the combine kernels.
struct RedoQueue {
    uint32_t* count; // cleared per MSM (k_sortA_scan)
    uint32_t* list;  // bucket numbers
    uint32_t* flags; // one word per bucket, all zero between MSMs: a bucket is queued once
};
__device__ __forceinline__ void redo_push(const RedoQueue& rq, uint32_t b)
{
    if (atomicExch(rq.flags + (b - 1), 1u) == 0) rq.list[atomicAdd(rq.count, 1u)] = b;
}

// Finished runs are not converted where they end (some lane of a wave ends a run in most iterations: the ~270 instructions of the R'-form ->
// R-form step and the store would be issued for a handful of active lanes nearly every time): a lane whose run ends PARKS the raw 4 x 9 limbs
// and the destination in a per-wave LDS queue and goes on; when the queue cannot take the next round of runs -- and once at the end of the
// segment -- the WHOLE wave converts and stores one queued run per lane (acc29_drain).  A segment's first entry starts its first run without an
// addition (all lanes: no selects), its last run goes out from registers with every lane active.
constexpr int ACC29_QCAP = 64; // queue entries per wave = lanes of one drain
struct Acc29Queue {
    uint4 limbs[9][ACC29_QCAP]; // planes: x 0-3, x 4-7, y 0-3, y 4-7, zz 0-3, zz 4-7, zzz 0-3, zzz 4-7, limb 8 of x | y | zz | zzz
    uint4 meta[ACC29_QCAP];     // destination (64-bit address) | bucket number | 1 = the run holds points at infinity only
};
__device__ __forceinline__ void acc29_park(Acc29Queue& qu, uint32_t slot, const Xyzz29& a, Xyzz* dst, uint32_t bucket, bool empty)
{
    qu.limbs[0][slot] = make_uint4(a.x.v[0], a.x.v[1], a.x.v[2], a.x.v[3]);
    qu.limbs[1][slot] = make_uint4(a.x.v[4], a.x.v[5], a.x.v[6], a.x.v[7]);
    qu.limbs[2][slot] = make_uint4(a.y.v[0], a.y.v[1], a.y.v[2], a.y.v[3]);
    qu.limbs[3][slot] = make_uint4(a.y.v[4], a.y.v[5], a.y.v[6], a.y.v[7]);
    qu.limbs[4][slot] = make_uint4(a.zz.v[0], a.zz.v[1], a.zz.v[2], a.zz.v[3]);
    qu.limbs[5][slot] = make_uint4(a.zz.v[4], a.zz.v[5], a.zz.v[6], a.zz.v[7]);
    qu.limbs[6][slot] = make_uint4(a.zzz.v[0], a.zzz.v[1], a.zzz.v[2], a.zzz.v[3]);
    qu.limbs[7][slot] = make_uint4(a.zzz.v[4], a.zzz.v[5], a.zzz.v[6], a.zzz.v[7]);
    qu.limbs[8][slot] = make_uint4(a.x.v[8], a.y.v[8], a.zz.v[8], a.zzz.v[8]);
    const uint64_t d = reinterpret_cast<uint64_t>(dst);
    qu.meta[slot] = make_uint4((uint32_t)d, (uint32_t)(d >> 32), bucket, empty ? 1u : 0u);
}
__device__ __forceinline__ Fq29 acc29_coord(const uint4& lo, const uint4& hi, uint32_t top)
{
    Fq29 r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    r.v[8] = top;
    return r;
}
// lanes [0, count) of the wave: one queued run each -> R-form words at its destination; a run that met P = +-acc (ZZ = 0 mod p) queues its bucket
__device__ __forceinline__ void acc29_drain(const Acc29Queue& qu, uint32_t count, const uint32_t* div32, const RedoQueue& redo)
{
    const uint32_t l = threadIdx.x & 63;
    if (l < count) {
        const uint4 m = qu.meta[l];
        const uint4 top = qu.limbs[8][l];
        char* dst = reinterpret_cast<char*>((uint64_t)m.x | ((uint64_t)m.y << 32));
        const bool empty = m.w != 0;
        Fq o = f29_div32_to_fe(acc29_coord(qu.limbs[4][l], qu.limbs[5][l], top.z), div32);
        const bool bad = fe_is_zero(o) && !empty;
        if (empty) o = Fq::zero();
        fe_store<FqP>(dst + 64, o);
        o = f29_div32_to_fe(acc29_coord(qu.limbs[0][l], qu.limbs[1][l], top.x), div32);
        if (empty) o = Fq::zero();
        fe_store<FqP>(dst, o);
        o = f29_div32_to_fe(acc29_coord(qu.limbs[2][l], qu.limbs[3][l], top.y), div32);
        if (empty) o = Fq::zero();
        fe_store<FqP>(dst + 32, o);
        o = f29_div32_to_fe(acc29_coord(qu.limbs[6][l], qu.limbs[7][l], top.w), div32);
        if (empty) o = Fq::zero();
        fe_store<FqP>(dst + 96, o);
        if (bad) redo_push(redo, m.z);
    }
}

template <int C> __global__ void __launch_bounds__(256)
k_accumulate29(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ offsets, const Affine* __restrict__ table,
               size_t n_srs, uint32_t seg, Xyzz* head, Xyzz* tail, Xyzz* buckets, RedoQueue redo, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    __shared__ __attribute__((aligned(16))) uint32_t div32[32 * DIV32_ROW]; // multiples of p for the R'-form -> R-form step at the end of a run
    __shared__ Acc29Queue queues[4];
    if (threadIdx.x < 32) f29_fill_div32_table<FqP>(div32, threadIdx.x);
    __syncthreads();
    Acc29Queue& qu = queues[threadIdx.x >> 6];
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t base = offsets[1];
    const uint64_t s64 = (uint64_t)base + (uint64_t)lane * seg;
    const bool active = s64 < total;
    if (__ballot(active) == 0) return; // whole waves leave; a partly filled wave keeps its idle lanes for the drains
    const uint32_t s = active ? (uint32_t)s64 : total - 1; // idle lanes: valid addresses, no iterations
    const uint32_t len = !active ? 0u : (total - s > seg) ? seg : total - s;
    uint32_t lo = 1, hi = MSM_BUCKETS;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= s) lo = mid;
        else hi = mid - 1;
    }
    uint32_t cur = lo;
    uint32_t cur_end = offsets[cur + 1]; // > s: no run ends at the segment's first entry
    bool first_run = true;
    uint32_t v = vals[s];
    Affine p = load_entry_point<C>(table, n_srs, v);
    // the first entry starts the first run
    bool empty = aff_is_inf(p);
    Xyzz29 acc = xyzz29_from_affine(aff29_from_table(p, (v >> 31) != 0));
    if (len > 1) {
        v = vals[s + 1];
        p = load_entry_point<C>(table, n_srs, v);
    }
    const uint64_t below = (1ull << (threadIdx.x & 63)) - 1;
    uint32_t parked = 0; // wave-uniform: runs waiting in the queue
    for (uint32_t it = 1;; it++) {
        const uint32_t q = s + it;
        const bool last = it >= seg; // wave-uniform
        const bool ends = !last && it < len && q == cur_end; // the run of bucket `cur` ended inside this segment
        const uint64_t enders = __ballot(ends);
        const uint32_t ne = (uint32_t)__popcll(enders);
        if (last || parked + ne > (uint32_t)ACC29_QCAP) {
            acc29_drain(qu, parked, div32, redo);
            parked = 0;
            if (last) break;
        }
        if (ends) {
            acc29_park(qu, parked + (uint32_t)__popcll(enders & below), acc, first_run ? head + lane : buckets + (cur - 1), cur, empty);
            first_run = false;
            empty = true;
            do {
                cur++;
                cur_end = offsets[cur + 1];
            } while (cur_end <= q); // skip empty buckets
        }
        parked += ne;
        if (it < len) {
            const uint32_t vc = v;
            const Affine pc = p;
            if (it + 1 < len) { // software prefetch of the next gather
                v = vals[q + 1];
                p = load_entry_point<C>(table, n_srs, v);
            }
            if (!aff_is_inf(pc)) {
                const Aff29 pt = aff29_from_table(pc, (vc >> 31) != 0);
                const Xyzz29 sum = xyzz29_madd(acc, pt);
                const Xyzz29 start = xyzz29_from_affine(pt);
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    acc.x.v[i] = empty ? start.x.v[i] : sum.x.v[i];
                    acc.y.v[i] = empty ? start.y.v[i] : sum.y.v[i];
                    acc.zz.v[i] = empty ? start.zz.v[i] : sum.zz.v[i];
                    acc.zzz.v[i] = empty ? start.zzz.v[i] : sum.zzz.v[i];
                }
                empty = false;
            }
        }
    }
    // the segment's last run, straight from registers
    if (active) {
        Xyzz out;
        if (empty) out = xyzz_inf(); // only points at infinity
        else if (!xyzz29_finish(acc, out, div32)) redo_push(redo, cur);
        xyzz_store(first_run ? head + lane : tail + lane, out);
    }
}

// one block per queued bucket: the bucket's sum from its entries, complete formulas (grid-stride over the queue; empty in all but degenerate inputs)
static __device__ Xyzz block_reduce(Xyzz v, Xyzz* sm, int nthreads);
template <int C> __global__ void __launch_bounds__(256)
k_redo(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ offsets, const Affine* __restrict__ table, size_t n_srs, RedoQueue rq, Xyzz* buckets)
{
    __shared__ Xyzz sm[128];
    const uint32_t cnt = *rq.count;
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
        const uint32_t b = rq.list[i];
        const uint32_t sb = offsets[b], eb = offsets[b + 1];
        Xyzz acc = xyzz_inf();
        for (uint32_t q = sb + threadIdx.x; q < eb; q += 256) {
            const uint32_t v = vals[q];
            acc = xyzz_madd(acc, aff_neg_if(load_entry_point<C>(table, n_srs, v), (v >> 31) != 0));
        }
        acc = block_reduce(acc, sm, 256);
        if (threadIdx.x == 0) {
            xyzz_store(buckets + (b - 1), acc);
            rq.flags[b - 1] = 0;
        }
        __syncthreads();
    }
}

// piece of bucket b held by lane l (see k_accumulate): head if the bucket starts at or before the lane's segment start
__device__ __forceinline__ Xyzz bucket_piece(const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail, uint32_t l, uint32_t l0,
                                             bool starts_at_seg_start)
{
    if (l == l0 && !starts_at_seg_start) return xyzz_load(tail + l);
    return xyzz_load(head + l);
}

// buckets[b-1] = sum of the pieces of bucket b; complete ("middle") runs were already written by k_accumulate.
// Buckets spanning more than MSM_LONG_SPAN lanes (skewed scalar distributions) are queued for k_combine_long.
template <int C> __global__ void __launch_bounds__(256, 1)
k_combine(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail,
          Xyzz* buckets, uint32_t* long_count, uint32_t* long_list, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    __shared__ uint32_t work[256]; // buckets of this block whose pieces have to be ADDED
    __shared__ uint32_t nwork;
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t base = offsets[1];
    if (threadIdx.x == 0) nwork = 0;
    __syncthreads();
    // Even buckets first, then odd ones: with narrow windows filed under doubled bucket numbers (msm_cfg.h) even buckets hold several times the
    // entries of odd ones and span more lane segments; a wave that mixes both runs the long loop for everybody (2^24, C = 22: 0.83 vs 0.61 ms).
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < MSM_BUCKETS) {
        constexpr uint32_t PER_SET = MsmCfg<C>::buckets;
        const uint32_t set = t / PER_SET, r = t % PER_SET;
        const uint32_t b = set * PER_SET + (r < PER_SET / 2 ? 2 * (r + 1) : 2 * (r - PER_SET / 2) + 1);
        const uint32_t sb = offsets[b], eb = offsets[b + 1];
        if (sb == eb) {
            xyzz_store(buckets + (b - 1), xyzz_inf());
        } else {
            const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
            const bool at_start = (sb == base + l0 * seg);
            if (l0 == l1) {
                uint32_t seg_end = base + (l0 + 1) * seg;
                if (seg_end > total || seg_end < base) seg_end = total;
                if (at_start) xyzz_store(buckets + (b - 1), xyzz_load(head + l0));
                else if (eb == seg_end) xyzz_store(buckets + (b - 1), xyzz_load(tail + l0));
                // else: complete run, already stored
            } else if (l1 - l0 > (uint32_t)MSM_LONG_SPAN) {
                const uint32_t slot = atomicAdd(long_count, 1u);
                long_list[slot] = b;
            } else {
                work[atomicAdd(&nwork, 1u)] = b;
            }
        }
    }
    __syncthreads();
    // About half of the buckets straddle a segment boundary and need one addition (rarely two): compacted, the additions fill whole waves
    // instead of half of every wave (37 M -> ~20 M VALU instructions per launch at n = 2^20).
    if (threadIdx.x < nwork) {
        const uint32_t b = work[threadIdx.x];
        const uint32_t sb = offsets[b], eb = offsets[b + 1];
        const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
        Xyzz acc = bucket_piece(head, tail, l0, l0, sb == base + l0 * seg);
        for (uint32_t l = l0 + 1; l <= l1; l++) acc = xyzz_add(acc, xyzz_load(head + l));
        xyzz_store(buckets + (b - 1), acc);
    }
}

static __device__ Xyzz block_reduce(Xyzz v, Xyzz* sm, int nthreads);

__device__ __forceinline__ Xyzz xyzz_shfl_xor(const Xyzz& v, int mask)
{
    Xyzz r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        r.x.v[i] = __shfl_xor(v.x.v[i], mask);
        r.y.v[i] = __shfl_xor(v.y.v[i], mask);
        r.zz.v[i] = __shfl_xor(v.zz.v[i], mask);
        r.zzz.v[i] = __shfl_xor(v.zzz.v[i], mask);
    }
    return r;
}

// Latency-oriented variant of k_combine: MSM_COMBINE_LANES adjacent lanes share a bucket, each sums every 4th piece, then
// a butterfly over the lane group (a bucket of ~512 entries has ~9 pieces: the serial chain drops from 8 additions to
// 5).  4 lanes, not 8: the butterfly runs in lock-step on every lane, and with 8 the reduce phase (which shares the chip
// with the next MSM's accumulation) cost 40 lane-additions per bucket instead of 17 -- 5 % of the pipelined step time.
constexpr int MSM_COMBINE_LANES = 4;
template <int C> __global__ void __launch_bounds__(256, 1)
k_combine_lanes(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail,
           Xyzz* buckets, uint32_t* long_count, uint32_t* long_list, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid / MSM_COMBINE_LANES + 1;
    const uint32_t r = gid % MSM_COMBINE_LANES;
    if (b > MSM_BUCKETS) return; // whole lane groups drop out together (grid is a multiple of 8)
    const uint32_t base = offsets[1];
    const uint32_t sb = offsets[b], eb = offsets[b + 1];
    bool store = true, reduce = false;
    Xyzz acc = xyzz_inf();
    if (sb != eb) {
        const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
        const bool at_start = (sb == base + l0 * seg);
        if (l0 == l1) {
            uint32_t seg_end = base + (l0 + 1) * seg;
            if (seg_end > total || seg_end < base) seg_end = total;
            if (at_start) acc = xyzz_load(head + l0);
            else if (eb == seg_end) acc = xyzz_load(tail + l0);
            else store = false; // complete run, already written by k_accumulate
        } else if (l1 - l0 > (uint32_t)MSM_LONG_SPAN) {
            if (r == 0) {
                const uint32_t slot = atomicAdd(long_count, 1u);
                long_list[slot] = b;
            }
            store = false;
        } else {
            reduce = true;
            for (uint32_t l = l0 + r; l <= l1; l += MSM_COMBINE_LANES) acc = xyzz_add(acc, bucket_piece(head, tail, l, l0, at_start));
        }
    }
    // the butterfly is executed by all lanes of the wave (shuffles need every lane); groups that do not reduce carry infinities
    if (!reduce && !(sb != eb && store)) acc = xyzz_inf();
    const unsigned long long any = __ballot(reduce);
    if (any) {
        Xyzz part = reduce ? acc : xyzz_inf();
        for (int m = MSM_COMBINE_LANES >> 1; m >= 1; m >>= 1) part = xyzz_add(part, xyzz_shfl_xor(part, m));
        if (reduce) acc = part;
    }
    if (store && r == 0) xyzz_store(buckets + (b - 1), acc);
}


// one block per queued long bucket (grid-stride over the queue)
template <int C> __global__ void __launch_bounds__(256, 1)
k_combine_long(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail, Xyzz* buckets,
               const uint32_t* __restrict__ long_count, const uint32_t* __restrict__ long_list)
{
    __shared__ Xyzz sm[128];
    const uint32_t cnt = *long_count;
    const uint32_t base = offsets[1];
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
        const uint32_t b = long_list[i];
        const uint32_t sb = offsets[b], eb = offsets[b + 1];
        const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
        const bool at_start = (sb == base + l0 * seg);
        Xyzz acc = xyzz_inf();
        for (uint32_t l = l0 + threadIdx.x; l <= l1; l += 256) acc = xyzz_add(acc, bucket_piece(head, tail, l, l0, at_start));
        acc = block_reduce(acc, sm, 256);
        if (threadIdx.x == 0) xyzz_store(buckets + (b - 1), acc);
        __syncthreads();
    }
}

// LDS tree reduction of one point per thread; result valid in thread 0.
static __device__ Xyzz block_reduce(Xyzz v, Xyzz* sm, int nthreads)
{
    const int tid = threadIdx.x;
    for (int stride = nthreads >> 1; stride >= 1; stride >>= 1) {
        if (tid >= stride && tid < 2 * stride) sm[tid - stride] = v;
        __syncthreads();
        if (tid < stride) v = xyzz_add(v, sm[tid]);
        __syncthreads();
    }
    return v;
}

// weight of bucket index idx (0-based) is idx + 1 = hi*COLS + lo + 1  (COLS = 2^log_cols, ROWS = 2^log_rows).
// blocks 0..ROWS-1: Row_hi = sum_lo B[hi][lo] ; blocks ROWS..ROWS+COLS-1: Col_lo = sum_hi B[hi][lo].
template <int C> __global__ void __launch_bounds__(256, 1) k_rowcol(const Xyzz* __restrict__ buckets, Xyzz* rows, Xyzz* cols)
{
    constexpr int ROWS = 1 << MsmCfg<C>::log_rows, COLS = 1 << MsmCfg<C>::log_cols;
    __shared__ Xyzz sm[128];
    const int tid = threadIdx.x;
    buckets += (size_t)blockIdx.y * MsmCfg<C>::buckets; // bucket set of MSM blockIdx.y of the batch
    rows += (size_t)blockIdx.y * ROWS;
    cols += (size_t)blockIdx.y * COLS;
    if (blockIdx.x < ROWS) {
        const int hi = blockIdx.x;
        Xyzz v = tid < COLS ? xyzz_load(buckets + (size_t)hi * COLS + tid) : xyzz_inf(); // (C = 13: 64 columns)
        for (int lo = tid + 256; lo < COLS; lo += 256) v = xyzz_add(v, xyzz_load(buckets + (size_t)hi * COLS + lo));
        v = block_reduce(v, sm, 256);
        if (tid == 0) xyzz_store(rows + hi, v);
    } else {
        const int lo = blockIdx.x - ROWS;
        Xyzz v = xyzz_inf();
        for (int hi = tid; hi < ROWS; hi += 256) v = xyzz_add(v, xyzz_load(buckets + (size_t)hi * COLS + lo));
        v = block_reduce(v, sm, 256);
        if (tid == 0) xyzz_store(cols + lo, v);
    }
}

// sum_b b*B_b = COLS * sum_hi hi*Row_hi + sum_lo (lo+1)*Col_lo = sum_t 2^t H_t with the bit planes
//   H_t = sum_{lo : bit t of (lo+1)} Col_lo  +  sum_{hi : bit (t - log_cols) of hi} Row_hi        (t = 0 .. C-2).
// One block per bit plane: tree-sum the selected rows/columns, then t doublings by one lane -- the planes run side by
// side, so the serial depth is ~9 additions + t doublings instead of a double-and-add on top of a tree plus log_cols more
// doublings.  k_final_sum adds the planes and converts to the reference's Jacobian layout.
template <int C> __global__ void __launch_bounds__(256, 1) k_final_planes(const Xyzz* __restrict__ rows, const Xyzz* __restrict__ cols, Xyzz* planes)
{
    constexpr int ROWS = 1 << MsmCfg<C>::log_rows, COLS = 1 << MsmCfg<C>::log_cols, LOGC = MsmCfg<C>::log_cols;
    __shared__ Xyzz sm[128];
    const int t = blockIdx.x, tid = threadIdx.x;
    rows += (size_t)blockIdx.y * ROWS;
    cols += (size_t)blockIdx.y * COLS;
    planes += (size_t)blockIdx.y * MSM_MAX_PLANES;
    Xyzz v = xyzz_inf();
    if (t <= LOGC)
        for (int lo = tid; lo < COLS; lo += 256)
            if (((lo + 1) >> t) & 1) v = xyzz_add(v, xyzz_load(cols + lo));
    if (t >= LOGC)
        for (int hi = tid; hi < ROWS; hi += 256)
            if ((hi >> (t - LOGC)) & 1) v = xyzz_add(v, xyzz_load(rows + hi));
    v = block_reduce(v, sm, 256);
    if (tid == 0) {
        for (int k = 0; k < t; k++) v = xyzz_dbl(v);
        xyzz_store(planes + t, v);
    }
}

// ------------------------------------------------------------------------------------ quad-cooperative reduce phase
// The same reduce phase with every EC operation shared by FOUR adjacent lanes (curve_quad.hip.h): a logical lane lt = thread / 4,
// q = thread % 4.  Identical structure and results; the dependency chain is ~3x shorter in time.  Blocks carry 4x the threads for the
// same number of logical lanes (512 threads = 128 logical lanes, so that a thread may keep up to 256 VGPRs).
__device__ __forceinline__ Xyzz block_reduce_q4(Xyzz v, Xyzz* sm, int nlogical)
{
    const int lt = threadIdx.x >> 2, q = threadIdx.x & 3;
    for (int stride = nlogical >> 1; stride >= 1; stride >>= 1) {
        if (q == 0 && lt >= stride && lt < 2 * stride) sm[lt - stride] = v;
        __syncthreads();
        if (lt < stride) v = xyzz_add_q4(v, sm[lt], q);
        __syncthreads();
    }
    return v;
}
constexpr int Q_LOGICAL = 128, Q_THREADS = 4 * Q_LOGICAL;

template <int C> __global__ void __launch_bounds__(Q_THREADS)
k_combine_q(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail,
            Xyzz* buckets, uint32_t* long_count, uint32_t* long_list, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t gl = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    const uint32_t b = gl + 1;
    if (b > MSM_BUCKETS) return;
    const uint32_t base = offsets[1];
    const uint32_t sb = offsets[b], eb = offsets[b + 1];
    if (sb == eb) {
        if (q == 0) xyzz_store(buckets + (b - 1), xyzz_inf());
        return;
    }
    const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
    const bool at_start = (sb == base + l0 * seg);
    if (l0 == l1) {
        uint32_t seg_end = base + (l0 + 1) * seg;
        if (seg_end > total || seg_end < base) seg_end = total;
        if (at_start) { if (q == 0) xyzz_store(buckets + (b - 1), xyzz_load(head + l0)); }
        else if (eb == seg_end) { if (q == 0) xyzz_store(buckets + (b - 1), xyzz_load(tail + l0)); }
        return; // else: complete run, already stored
    }
    if (l1 - l0 > (uint32_t)MSM_LONG_SPAN) {
        if (q == 0) {
            const uint32_t slot = atomicAdd(long_count, 1u);
            long_list[slot] = b;
        }
        return;
    }
    Xyzz acc = bucket_piece(head, tail, l0, l0, at_start);
    for (uint32_t l = l0 + 1; l <= l1; l++) acc = xyzz_add_q4(acc, xyzz_load(head + l), q);
    if (q == 0) xyzz_store(buckets + (b - 1), acc);
}

// MSM_COMBINE_LANES logical lanes (16 threads) per bucket, butterfly across the logical lanes (shuffle distances 8 and 4)
template <int C> __global__ void __launch_bounds__(Q_THREADS)
k_combine_lanes_q(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail,
                  Xyzz* buckets, uint32_t* long_count, uint32_t* long_list, uint32_t nb)
{
    const uint32_t MSM_BUCKETS = nb; // sets x 2^(C-1): every MSM of the batch has its own bucket set
    const uint32_t total = offsets[MSM_BUCKETS + 1];
    const uint32_t gl = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    const uint32_t b = gl / MSM_COMBINE_LANES + 1;
    const uint32_t r = gl % MSM_COMBINE_LANES;
    if (b > MSM_BUCKETS) return; // whole waves drop out together (64 threads = 4 buckets)
    const uint32_t base = offsets[1];
    const uint32_t sb = offsets[b], eb = offsets[b + 1];
    bool store = true, reduce = false;
    Xyzz acc = xyzz_inf();
    if (sb != eb) {
        const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
        const bool at_start = (sb == base + l0 * seg);
        if (l0 == l1) {
            uint32_t seg_end = base + (l0 + 1) * seg;
            if (seg_end > total || seg_end < base) seg_end = total;
            if (at_start) acc = xyzz_load(head + l0);
            else if (eb == seg_end) acc = xyzz_load(tail + l0);
            else store = false; // complete run, already written by k_accumulate
        } else if (l1 - l0 > (uint32_t)MSM_LONG_SPAN) {
            if (r == 0 && q == 0) {
                const uint32_t slot = atomicAdd(long_count, 1u);
                long_list[slot] = b;
            }
            store = false;
        } else {
            reduce = true;
            for (uint32_t l = l0 + r; l <= l1; l += MSM_COMBINE_LANES) acc = xyzz_add_q4(acc, bucket_piece(head, tail, l, l0, at_start), q);
        }
    }
    if (!reduce && !(sb != eb && store)) acc = xyzz_inf();
    const unsigned long long any = __ballot(reduce);
    if (any) {
        Xyzz part = reduce ? acc : xyzz_inf();
        for (int m = MSM_COMBINE_LANES >> 1; m >= 1; m >>= 1) part = xyzz_add_q4(part, xyzz_shfl_xor(part, 4 * m), q);
        if (reduce) acc = part;
    }
    if (store && r == 0 && q == 0) xyzz_store(buckets + (b - 1), acc);
}

template <int C> __global__ void __launch_bounds__(Q_THREADS)
k_combine_long_q(const uint32_t* __restrict__ offsets, uint32_t seg, const Xyzz* __restrict__ head, const Xyzz* __restrict__ tail, Xyzz* buckets,
                 const uint32_t* __restrict__ long_count, const uint32_t* __restrict__ long_list)
{
    __shared__ Xyzz sm[Q_LOGICAL / 2];
    const int lt = threadIdx.x >> 2, q = threadIdx.x & 3;
    const uint32_t cnt = *long_count;
    const uint32_t base = offsets[1];
    for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
        const uint32_t b = long_list[i];
        const uint32_t sb = offsets[b], eb = offsets[b + 1];
        const uint32_t l0 = (sb - base) / seg, l1 = (eb - 1 - base) / seg;
        const bool at_start = (sb == base + l0 * seg);
        Xyzz acc = xyzz_inf();
        for (uint32_t l = l0 + lt; l <= l1; l += Q_LOGICAL) acc = xyzz_add_q4(acc, bucket_piece(head, tail, l, l0, at_start), q);
        acc = block_reduce_q4(acc, sm, Q_LOGICAL);
        if (threadIdx.x == 0) xyzz_store(buckets + (b - 1), acc);
        __syncthreads();
    }
}

// QL logical lanes (4 QL threads) per row / column: each lane first sums its share serially, then a tree of log2 QL levels.  Fewer lanes =
// more serial additions but fewer idle tree slots and fewer waves per SIMD (a tree level costs ~2.7 us with one wave per SIMD, ~7 with three).
#ifndef BBG_ROWCOL_QL
#define BBG_ROWCOL_QL 64 // logical lanes (quads) per row / column block
#endif
template <int C, int QL> __global__ void __launch_bounds__(4 * QL) k_rowcol_q(const Xyzz* __restrict__ buckets, Xyzz* rows, Xyzz* cols)
{
    constexpr int ROWS = 1 << MsmCfg<C>::log_rows, COLS = 1 << MsmCfg<C>::log_cols;
    __shared__ Xyzz sm[QL / 2];
    const int q = threadIdx.x & 3;
    buckets += (size_t)blockIdx.y * MsmCfg<C>::buckets; // bucket set of MSM blockIdx.y of the batch
    rows += (size_t)blockIdx.y * ROWS;
    cols += (size_t)blockIdx.y * COLS;
    // serial phase: every THREAD sums its own share with one-lane additions (a quad-cooperative addition costs 1.55x the instructions of a
    // one-lane one -- it buys latency, and there is none to buy while all four lanes have items of their own); then the four partial sums of
    // a quad are added with four-lane operations, then the tree over the QL quads.  A thread's FIRST item is loaded, not added to an
    // infinity (r4): with 4 QL = 256 threads a row of 2^8 .. 2^9 buckets gives a thread one or two items, so starting from infinity was
    // half of the serial additions (C = 19: 262 144 -> 65 536 one-lane additions per launch) and 40 % of a small MSM's row / column kernel.
    constexpr int T = 4 * QL;
    Xyzz s = xyzz_inf();
    if (blockIdx.x < ROWS) {
        const int hi = blockIdx.x;
        const Xyzz* src = buckets + (size_t)hi * COLS;
        if ((int)threadIdx.x < COLS) s = xyzz_load(src + threadIdx.x);
        for (int lo = T; lo < COLS; lo += T) s = xyzz_add(s, xyzz_load(src + lo + threadIdx.x)); // COLS is a multiple of T beyond the first item
        Xyzz v = block_reduce_q4(quad_sum4(s, q), sm, QL);
        if (threadIdx.x == 0) xyzz_store(rows + hi, v);
    } else {
        const int lo = blockIdx.x - ROWS;
        if ((int)threadIdx.x < ROWS) s = xyzz_load(buckets + (size_t)threadIdx.x * COLS + lo);
        for (int hi = T; hi < ROWS; hi += T) s = xyzz_add(s, xyzz_load(buckets + (size_t)(hi + threadIdx.x) * COLS + lo));
        Xyzz v = block_reduce_q4(quad_sum4(s, q), sm, QL);
        if (threadIdx.x == 0) xyzz_store(cols + lo, v);
    }
}

template <int C> __global__ void __launch_bounds__(Q_THREADS) k_final_planes_q(const Xyzz* __restrict__ rows, const Xyzz* __restrict__ cols, Xyzz* planes)
{
    constexpr int ROWS = 1 << MsmCfg<C>::log_rows, COLS = 1 << MsmCfg<C>::log_cols, LOGC = MsmCfg<C>::log_cols;
    __shared__ Xyzz sm[Q_LOGICAL / 2];
    const int t = blockIdx.x, lt = threadIdx.x >> 2, q = threadIdx.x & 3;
    rows += (size_t)blockIdx.y * ROWS;
    cols += (size_t)blockIdx.y * COLS;
    planes += (size_t)blockIdx.y * MSM_MAX_PLANES;
    Xyzz v = xyzz_inf();
    if (t <= LOGC)
        for (int lo = lt; lo < COLS; lo += Q_LOGICAL)
            if (((lo + 1) >> t) & 1) v = xyzz_add_q4(v, xyzz_load(cols + lo), q);
    if (t >= LOGC)
        for (int hi = lt; hi < ROWS; hi += Q_LOGICAL)
            if ((hi >> (t - LOGC)) & 1) v = xyzz_add_q4(v, xyzz_load(rows + hi), q);
    v = block_reduce_q4(v, sm, Q_LOGICAL);
    if (lt == 0) {
        for (int k = 0; k < t; k++) v = xyzz_dbl_q4(v, q);
        if (q == 0) xyzz_store(planes + t, v);
    }
}

// ---------------------------------------------------------------------------------- host side
struct MsmLayout {
    size_t entries, lanes;
    size_t off_keys0, off_keys1, off_vals0[bbg_ctx::MSM_SLOTS], off_vals1, off_sort, off_parts;
    uint32_t seg;
    // reduce-phase working set, double buffered so that the reduce of MSM i (aux stream) overlaps MSM i+1
    size_t off_offsets[bbg_ctx::MSM_SLOTS], off_head[bbg_ctx::MSM_SLOTS], off_tail[bbg_ctx::MSM_SLOTS], off_buckets[bbg_ctx::MSM_SLOTS], off_rows[bbg_ctx::MSM_SLOTS],
        off_cols[bbg_ctx::MSM_SLOTS], off_long[bbg_ctx::MSM_SLOTS], off_redo[bbg_ctx::MSM_SLOTS];
    size_t sort_bytes;
    size_t zero_bytes; // the leading part of the arena that holds the zero-initialised regions
    size_t total;
};
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Segment length (entries per accumulation lane).  Around 64 at the headline size; shorter for small MSMs, whose run time
// is the latency of one lane's serial chain of mixed additions (64 additions = 0.43 ms regardless of n), and longer for
// very large ones so that a bucket spans <= ~32 lanes and the lane-group combine (not the block-per-bucket fallback) sums
// its pieces.  When one round of waves covers the whole job the lane count is made a multiple of the chip's 65536 SIMD
// lanes: with 3.25 waves per SIMD the kernel takes as long as with 4, because the busiest SIMD sets the time.
static uint32_t msm_seg_len(size_t entries, size_t buckets, int waves_override)
{
    constexpr size_t CHIP_LANES = 65536; // 256 CUs x 4 SIMDs x 64
    const size_t MAX_WAVES = waves_override > 0 ? (size_t)waves_override : 6; // lane segments per SIMD lane and round
    if (entries <= MSM_SEG_MIN * 4 * CHIP_LANES) return MSM_SEG_MIN; // small: at most 4 waves per SIMD of 8 entries
    size_t seg;
    if (entries <= MSM_SEG_DEFAULT * MAX_WAVES * CHIP_LANES) { // one round of k = 4..6 full waves per SIMD, <= 64 entries each
        size_t k = (entries + MSM_SEG_DEFAULT * CHIP_LANES - 1) / (MSM_SEG_DEFAULT * CHIP_LANES);
        if (k < 4) k = 4;
        if (waves_override > 0) k = (size_t)waves_override;
        seg = (entries + k * CHIP_LANES - 1) / (k * CHIP_LANES);
    } else { // whole rounds of MAX_WAVES waves per SIMD, ~64 entries each
        const size_t per_round = MSM_SEG_DEFAULT * MAX_WAVES * CHIP_LANES;
        size_t rounds = (entries + per_round / 2) / per_round;
        if (rounds < 1) rounds = 1;
        seg = (entries + rounds * MAX_WAVES * CHIP_LANES - 1) / (rounds * MAX_WAVES * CHIP_LANES);
    }
    while (entries / seg > buckets * 32 && entries / seg > (size_t)1048576) seg *= 2; // <= 32 pieces per average bucket
    return (uint32_t)seg;
}

// total_n = scalars of all `sets` MSMs of the batch together.  The two regions that must be ZERO when an MSM starts -- the sort's partition
// counters and the redo queues' per-bucket flags -- come first and are sized for `cap_sets` (the largest batch this context has run at this
// width): they keep their place when n or the batch size change, so a prover's rounds (batches of 4, 1, 4, 2 over the same n) neither move
// nor clear them (round 4: twelve memsets per proof).
template <int C> static int msm_layout(size_t total_n, int sets, int cap_sets, bool library_sort, MsmLayout& L, int acc_waves = 0)
{
    using K = MsmCfg<C>;
    const size_t nbuckets = (size_t)sets * K::buckets;
    L.entries = total_n * K::windows;
    L.seg = msm_seg_len(L.entries, nbuckets, acc_waves);
    L.lanes = (L.entries + L.seg - 1) / L.seg;
    size_t tmp = 0;
#ifdef BBG_ROCPRIM_SORT
    if (library_sort) { // only the A/B path (msm_sort = 0) needs rocPRIM's temporary storage: the default path neither queries nor reserves it
        rocprim::double_buffer<uint32_t> dk(nullptr, nullptr), dv(nullptr, nullptr);
        hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, dk, dv, L.entries, 0u, (unsigned)C);
        if (e != hipSuccess) return hip_fail(e, "rocprim::radix_sort_pairs(size query)", __FILE__, __LINE__);
    }
#else
    (void)library_sort;
#endif
    L.sort_bytes = tmp;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.off_parts = take((size_t)3 * cap_sets * SORT_PAD * 4); // per set: partition counts | bases | cursors (strides of cap_sets tables)
    for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) L.off_redo[k] = take((size_t)2 * cap_sets * K::buckets * 4); // RedoQueue: flags (zero between MSMs), then the list
    L.zero_bytes = o;
    L.off_keys0 = take(L.entries * 4);
    L.off_keys1 = take(L.entries * 4);
    // sorted values, one copy per slot: a bucket queued for k_redo (reduce phase, auxiliary stream) is recomputed from its entries while the
    // next MSM already sorts
    for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) L.off_vals0[k] = take(L.entries * 4);
    L.off_vals1 = take(library_sort ? L.entries * 4 : 0); // second value buffer of the library sort's double buffer (A/B build only)
    L.off_sort = take(L.sort_bytes);
    for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) {
        L.off_offsets[k] = take((nbuckets + 2) * 4);
        L.off_head[k] = take(L.lanes * sizeof(Xyzz));
        L.off_tail[k] = take(L.lanes * sizeof(Xyzz));
        L.off_buckets[k] = take(nbuckets * sizeof(Xyzz));
        L.off_rows[k] = take((size_t)sets * ((size_t)(1 << K::log_rows) + MSM_MAX_PLANES) * sizeof(Xyzz)); // row sums of every set, then their bit planes
        L.off_cols[k] = take((size_t)sets * (size_t)(1 << K::log_cols) * sizeof(Xyzz));
        L.off_long[k] = take((nbuckets + 2) * 4); // long-bucket count, redo count, long-bucket list
    }
    L.total = o;
    return BBG_OK;
}

static inline char* base_of(bbg_ctx* ctx) { return (char*)ctx->msm.buf; }
// the auxiliary streams the reduce phases run on (one per reduce slot) and their events; created with the context's first MSM
static inline int msm_ensure_aux_streams(bbg_ctx* ctx)
{
    if (ctx->aux_stream) return BBG_OK;
    // the reduce phase is latency work that only has to finish before its result is consumed: a LOW-priority stream, so that what
    // the caller queues next on the main stream (the following MSM's sort / accumulation, an NTT) is dispatched first
    int least = 0, greatest = 0;
    BBG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) {
        BBG_HIP(hipStreamCreateWithPriority(&ctx->aux_streams[k], hipStreamNonBlocking, ctx->msm_reduce_low_priority ? least : (least + greatest) / 2));
        BBG_HIP(hipEventCreateWithFlags(&ctx->ev_acc[k], hipEventDisableTiming));
        BBG_HIP(hipEventCreateWithFlags(&ctx->ev_done[k], hipEventDisableTiming));
    }
    ctx->aux_stream = ctx->aux_streams[0];
    return BBG_OK;
}

template <int C>
int msm_run_c(bbg_ctx* ctx, const Srs& srs, const void* table_v, int sets, const void* const* d_scalars_v, const size_t* from_v, const size_t* n_v,
              void* d_out_jac, hipStream_t st, const void* h_scalars)
{
    const Affine* table = (const Affine*)table_v;
    using K = MsmCfg<C>;
    MsmBatch batch;
    size_t total_n = 0, max_n = 0;
    for (int k = 0; k < MSM_BATCH_MAX; k++) {
        batch.scalars[k] = k < sets ? (const Fr*)d_scalars_v[k] : nullptr;
        batch.n[k] = k < sets ? (uint32_t)n_v[k] : 0u;
        batch.from[k] = k < sets ? (uint32_t)from_v[k] : 0u;
        if (k < sets) {
            total_n += n_v[k];
            max_n = std::max(max_n, n_v[k]);
        }
    }
    if (total_n * K::windows >= ((size_t)1 << 32)) {
        set_error("bbg_msm: more than 2^32 sorted entries in one launch set (split the batch)");
        return BBG_E_INVALID;
    }
    const size_t n = max_n; // grid extent of the per-scalar kernels (blockIdx.y = MSM of the batch; shorter ones leave early)
    const uint32_t nbuckets = (uint32_t)sets * K::buckets;
    MsmLayout L;
    const int cap_sets = (ctx->msm_zero_c == C && ctx->msm_zero_sets > sets) ? ctx->msm_zero_sets : sets;
    int rc = msm_layout<C>(total_n, sets, cap_sets, ctx->msm_sort == 0, L, ctx->msm_acc_waves);
    if (rc) return rc;
    // The "already cleared" record (msm_zero_*) describes ONE allocation, not an address: a regrown arena may come back at the same base
    // with arbitrary contents, and a call that fails between the counting kernel and the scan that clears the counters again leaves them
    // non-zero.  So the record is dropped whenever the arena is (re)allocated and whenever this function leaves early (round-4 advisor).
    struct ZeroRecordGuard {
        bbg_ctx* c;
        bool done = false;
        ~ZeroRecordGuard() { if (!done) c->msm_zero_buf = nullptr; }
    } zero_guard{ctx};
    const size_t arena_had = ctx->msm.bytes;
    rc = ensure_buffer(&ctx->msm.buf, &ctx->msm.bytes, L.total);
    if (rc) return rc;
    if (ctx->msm.bytes != arena_had) ctx->msm_zero_buf = nullptr;
    rc = msm_ensure_aux_streams(ctx);
    if (rc) return rc;
    if (ctx->msm_layout_n != total_n || ctx->msm_layout_sets != sets || ctx->msm_layout_c != C || ctx->msm_layout_sort != ctx->msm_sort) {
        // a different (n, sets, C) lays the arena out differently: a reduce phase still running on the auxiliary stream reads
        // regions this call is about to overwrite, so the main stream first waits for both slots (no host sync)
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++)
            if (ctx->ev_done_valid[k]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[k], 0));
        ctx->msm_layout_n = total_n;
        ctx->msm_layout_sets = sets;
        ctx->msm_layout_c = C;
        ctx->msm_layout_sort = ctx->msm_sort;
    }
    if (ctx->msm_zero_buf != ctx->msm.buf || ctx->msm_zero_c != C || ctx->msm_zero_sets != cap_sets) {
        // the zero-initialised regions are new (fresh arena), belong to another width, or grew: cleared once here -- then the partition
        // counters are kept clear by k_sortA_scan and the redo flags by k_redo, whatever n and batch size the following calls have
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++)
            if (ctx->ev_done_valid[k]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[k], 0));
        BBG_HIP(hipMemsetAsync(base_of(ctx), 0, L.zero_bytes, st));
        ctx->msm_zero_buf = ctx->msm.buf;
        ctx->msm_zero_c = C;
        ctx->msm_zero_sets = cap_sets;
    }
    const int slot = (int)(ctx->msm_seq++ % bbg_ctx::MSM_SLOTS);
    char* base = (char*)ctx->msm.buf;
    uint32_t* keys0 = (uint32_t*)(base + L.off_keys0);
    uint32_t* vals0 = (uint32_t*)(base + L.off_vals0[slot]);
#ifdef BBG_ROCPRIM_SORT
    uint32_t* keys1 = (uint32_t*)(base + L.off_keys1);
    uint32_t* vals1 = (uint32_t*)(base + L.off_vals1);
#endif
    uint32_t* offsets = (uint32_t*)(base + L.off_offsets[slot]);
    Xyzz* head = (Xyzz*)(base + L.off_head[slot]);
    Xyzz* tail = (Xyzz*)(base + L.off_tail[slot]);
    Xyzz* buckets = (Xyzz*)(base + L.off_buckets[slot]);
    Xyzz* rows = (Xyzz*)(base + L.off_rows[slot]);
    Xyzz* cols = (Xyzz*)(base + L.off_cols[slot]);
    Xyzz* planes = rows + (size_t)sets * (1 << K::log_rows);
    uint32_t* long_count = (uint32_t*)(base + L.off_long[slot]);
    uint32_t* long_list = long_count + 2; // word 1: the redo queue's count
    const bool overlap = ctx->msm_async_reduce;
    hipStream_t rst = overlap ? ctx->aux_streams[slot] : st; // stream of the reduce phase

    // this slot's offsets / head / tail / buckets were last read by the reduce phase of the MSM MSM_SLOTS calls ago
    if (ctx->ev_done_valid[slot]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[slot], 0));
    const uint32_t* svals;
    const void* d_scalars = d_scalars_v[0]; // host-buffer uploads exist for a batch of one only (bbg_msm)
    const int pieces = (h_scalars && sets == 1 && ctx->msm_sort == 1 && n >= ((size_t)1 << 16)) ? ctx->msm_upload_pieces : 1;
    if (h_scalars && pieces <= 1) // small n / library sort / option: one copy in front of everything
        BBG_HIP(hipMemcpyAsync((void*)d_scalars, h_scalars, n * 32, hipMemcpyHostToDevice, st));
    if (ctx->msm_sort == 1) {
        // fused recode + MSD partition sort (keys0 area = 64-bit entries, vals0 = final values, keys1 head = partition tables)
        uint64_t* entries = (uint64_t*)keys0; // keys0 and keys1 are adjacent: 2 x 4 x 16n bytes = 8 x 16n
        uint32_t* part_count = (uint32_t*)(base + L.off_parts);
        uint32_t* part_base = part_count + (size_t)cap_sets * SORT_PAD;
        uint32_t* cursor = part_count + (size_t)2 * cap_sets * SORT_PAD;
        const int nblk = grid_for(n, SORT_BLOCK);
        {
            ProfScope ps(ctx, "msm_recode", st);
            if (pieces > 1) {
                // The counting pass is a histogram (global atomics): it does not care in which order, or in how many launches, it
                // sees the scalars.  So the 32n bytes travel in pieces on their own stream and each piece is counted as soon as it has
                // landed -- the only part of the MSM that can start before ALL scalars are there (the scatter needs the totals).
                if (!ctx->upload_stream) {
                    BBG_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
                    BBG_HIP(hipEventCreateWithFlags(&ctx->ev_upload_go, hipEventDisableTiming));
                    for (int k = 0; k < bbg_ctx::UPLOAD_PIECES; k++) BBG_HIP(hipEventCreateWithFlags(&ctx->ev_upload[k], hipEventDisableTiming));
                }
                BBG_HIP(hipEventRecord(ctx->ev_upload_go, st)); // whatever `st` still does with the staging area comes first
                BBG_HIP(hipStreamWaitEvent(ctx->upload_stream, ctx->ev_upload_go, 0));
                const size_t blocks_per_piece = ((size_t)nblk + pieces - 1) / pieces;
                for (int k = 0; k < pieces; k++) {
                    const size_t lo = (size_t)k * blocks_per_piece * SORT_BLOCK;
                    if (lo >= n) break;
                    const size_t len = n - lo < blocks_per_piece * SORT_BLOCK ? n - lo : blocks_per_piece * SORT_BLOCK;
                    BBG_HIP(hipMemcpyAsync((char*)d_scalars + lo * 32, (const char*)h_scalars + lo * 32, len * 32, hipMemcpyHostToDevice,
                                           ctx->upload_stream));
                    BBG_HIP(hipEventRecord(ctx->ev_upload[k], ctx->upload_stream));
                    BBG_HIP(hipStreamWaitEvent(st, ctx->ev_upload[k], 0));
                    MsmBatch piece = batch;
                    piece.scalars[0] = (const Fr*)d_scalars + lo;
                    piece.n[0] = (uint32_t)len;
                    hipLaunchKernelGGL((k_sortA_count<C, true>), dim3(std::min(grid_for(len, SORT_BLOCK), COUNT_MAX_BLOCKS)), dim3(COUNT_THREADS), 0, st, piece,
                                       part_count);
                }
            } else {
                if (nblk > COUNT_MAX_BLOCKS)
                    hipLaunchKernelGGL((k_sortA_count<C, true>), dim3(COUNT_MAX_BLOCKS, sets), dim3(COUNT_THREADS), 0, st, batch, part_count);
                else
                    hipLaunchKernelGGL((k_sortA_count<C, false>), dim3(nblk, sets), dim3(COUNT_THREADS), 0, st, batch, part_count);
            }
            hipLaunchKernelGGL(k_sortA_scan<C>, dim3(1), dim3(1024), 0, st, part_count, part_base, cursor, offsets, long_count, sets);
        }
        {
            ProfScope ps(ctx, "msm_sort", st);
            hipLaunchKernelGGL(k_sortA_scatter<C>, dim3(nblk, sets), dim3(SORT_BLOCK), 0, st, batch, cursor, entries);
            // partitions of a few hundred entries (n <= 2^17 at 16 windows): 256-thread blocks; skewed input is still handled (chunked path)
            if (L.entries / ((size_t)1024 * sets) <= (size_t)SORTB_PER_THREAD * 256 / 2)
                hipLaunchKernelGGL((k_sortB<C, 256>), dim3(K::parts, sets), dim3(256), 0, st, entries, part_base, offsets, vals0);
            else
                hipLaunchKernelGGL((k_sortB<C, 1024>), dim3(K::parts, sets), dim3(1024), 0, st, entries, part_base, offsets, vals0);
        }
        svals = vals0;
    } else {
#ifdef BBG_ROCPRIM_SORT
        if (sets != 1) {
            set_error("msm_sort = 0 (library sort, A/B only) runs single MSMs only");
            return BBG_E_INVALID;
        }
        {
            ProfScope ps(ctx, "msm_recode", st);
            hipLaunchKernelGGL(k_recode<C>, dim3(grid_for(n, 256)), dim3(256), 0, st, (const Fr*)d_scalars, n, from_v[0], keys0, vals0);
        }
        rocprim::double_buffer<uint32_t> dk(keys0, keys1), dv(vals0, vals1);
        {
            ProfScope ps(ctx, "msm_sort", st);
            size_t tmp = L.sort_bytes;
            hipError_t e = rocprim::radix_sort_pairs(base + L.off_sort, tmp, dk, dv, L.entries, 0u, (unsigned)C, st);
            if (e != hipSuccess) return hip_fail(e, "rocprim::radix_sort_pairs", __FILE__, __LINE__);
        }
        const uint32_t* skeys = dk.current();
        if (dv.current() != vals0) // k_redo (reduce stream) re-reads the sorted values while the next MSM sorts: they live in this slot's buffer
            BBG_HIP(hipMemcpyAsync(vals0, dv.current(), L.entries * 4, hipMemcpyDeviceToDevice, st));
        svals = vals0;
        {
            ProfScope ps(ctx, "msm_offsets", st);
            hipLaunchKernelGGL(k_offsets<C>, dim3(grid_for(L.entries + 1, 256)), dim3(256), 0, st, skeys, L.entries, offsets);
        }
#else
        set_error("msm_sort = 0 needs a library built with ROCPRIM_SORT=1");
        return BBG_E_INVALID;
#endif
    }
    bool redo_pending = false; // k_accumulate29 ran: buckets it could not sum are queued
    uint32_t* redo_words = reinterpret_cast<uint32_t*>(base + L.off_redo[slot]);
    const RedoQueue redo{ long_count + 1, redo_words + (size_t)cap_sets * K::buckets, redo_words }; // flags first: their place does not depend on the batch
    {
        ProfScope ps(ctx, "msm_accumulate", st);
        if (ctx->msm_sort != 1) BBG_HIP(hipMemsetAsync(long_count, 0, 8, st)); // the partition sort's scan kernel clears both counts
        // few lanes (n <= 2^14 or so): the kernel's time is one lane's chain of dependent mixed additions -- four threads per lane shorten it
        if (ctx->msm_accumulate_quad && L.lanes <= MSM_QUAD_ACC_MAX_LANES)
            hipLaunchKernelGGL(k_accumulate_q4<C>, dim3(grid_for(L.lanes * 4, 256)), dim3(256), 0, st, svals, offsets, table, srs.n, L.seg, head, tail,
                               buckets, nbuckets);
        else if (ctx->msm_limbs29) {
            redo_pending = true;
            hipLaunchKernelGGL(k_accumulate29<C>, dim3(grid_for(L.lanes, 256)), dim3(256), 0, st, svals, offsets, table, srs.n, L.seg, head, tail, buckets,
                               redo, nbuckets);
        } else
            hipLaunchKernelGGL(k_accumulate<C>, dim3(grid_for(L.lanes, 256)), dim3(256), 0, st, svals, offsets, table, srs.n, L.seg, head, tail, buckets,
                               nbuckets);
    }
    if (overlap) {
        BBG_HIP(hipEventRecord(ctx->ev_acc[slot], st));
        BBG_HIP(hipStreamWaitEvent(rst, ctx->ev_acc[slot], 0));
    }
    {
        ProfScope ps(ctx, "msm_reduce", rst);
        // msm_reduce_quad: bit 0 combine, bit 1 row/column sums, bit 2 bit planes, bit 3 plane sum -- each stage either with four lanes per EC
        // operation (curve_quad.hip.h: the same chain, ~3x shorter in time) or with one (the round-1 kernels, kept for A/B)
        const int quad = ctx->msm_reduce_quad;
        if (quad & 1) {
            if (L.lanes > (size_t)2 * nbuckets)
                hipLaunchKernelGGL(k_combine_lanes_q<C>, dim3(grid_for((size_t)nbuckets * MSM_COMBINE_LANES * 4, Q_THREADS)), dim3(Q_THREADS), 0, rst,
                                   offsets, L.seg, head, tail, buckets, long_count, long_list, nbuckets);
            else
                hipLaunchKernelGGL(k_combine_q<C>, dim3(grid_for((size_t)nbuckets * 4, Q_THREADS)), dim3(Q_THREADS), 0, rst, offsets, L.seg, head, tail,
                                   buckets, long_count, long_list, nbuckets);
            hipLaunchKernelGGL(k_combine_long_q<C>, dim3(256), dim3(Q_THREADS), 0, rst, offsets, L.seg, head, tail, buckets, long_count, long_list);
        } else {
            if (L.lanes > (size_t)2 * nbuckets) // several pieces per bucket: lane groups + butterfly; else one lane per bucket
                hipLaunchKernelGGL(k_combine_lanes<C>, dim3(grid_for((size_t)nbuckets * MSM_COMBINE_LANES, 256)), dim3(256), 0, rst, offsets,
                                   L.seg, head, tail, buckets, long_count, long_list, nbuckets);
            else
                hipLaunchKernelGGL(k_combine<C>, dim3(grid_for((size_t)nbuckets, 256)), dim3(256), 0, rst, offsets, L.seg, head, tail,
                                   buckets, long_count, long_list, nbuckets);
            hipLaunchKernelGGL(k_combine_long<C>, dim3(256), dim3(256), 0, rst, offsets, L.seg, head, tail, buckets, long_count, long_list);
        }
        if (redo_pending) hipLaunchKernelGGL(k_redo<C>, dim3(128), dim3(256), 0, rst, svals, offsets, table, srs.n, redo, buckets);
        // 64 logical lanes per row / column (measured against 128 / 32 / 16: reduce phase 0.458 / 0.49 / 0.53 / 0.55 ms at 2^20 stand-alone,
        // 0.154 / 0.165 / 0.164 / 0.184 at 2^10; bench step equal for 64 and 128, worse below)
        const dim3 rc_grid((1 << K::log_rows) + (1 << K::log_cols), sets);
        if (quad & 2) hipLaunchKernelGGL((k_rowcol_q<C, BBG_ROWCOL_QL>), rc_grid, dim3(4 * BBG_ROWCOL_QL), 0, rst, buckets, rows, cols);
        else hipLaunchKernelGGL(k_rowcol<C>, rc_grid, dim3(256), 0, rst, buckets, rows, cols);
        if (quad & 4) hipLaunchKernelGGL(k_final_planes_q<C>, dim3(K::planes, sets), dim3(Q_THREADS), 0, rst, rows, cols, planes);
        else hipLaunchKernelGGL(k_final_planes<C>, dim3(K::planes, sets), dim3(256), 0, rst, rows, cols, planes);
        // (the plane sum stays a launch of its own: folded into the planes kernel as "the last block to finish adds the planes", its
        // device-scope fences write back and invalidate the L2 of every XCD a block runs on -- the accumulation running beside it lost
        // 2 % (1.120 vs 1.098 ms, bench step 1.672 vs 1.647), and a small MSM gained nothing: 87-98 us against 64 + 20)
        rc = msm_launch_final_sum((quad & 8) != 0, planes, (int)K::planes, d_out_jac, rst, sets);
        if (rc) return rc;
    }
    if (overlap) {
        BBG_HIP(hipEventRecord(ctx->ev_done[slot], rst));
        ctx->ev_done_valid[slot] = true;
    }
    BBG_HIP(hipGetLastError());
    zero_guard.done = true;
    return BBG_OK;
}

template <int C> int srs_build_tables_c(const void* d_points, size_t n, void* d_table, hipStream_t st)
{
    if (n == 0) return BBG_OK;
    hipLaunchKernelGGL(k_precompute_tables<C>, dim3(grid_for(n, 128)), dim3(128), 0, st, (const Affine*)d_points, (Affine*)d_table, n);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

} // namespace bbg
