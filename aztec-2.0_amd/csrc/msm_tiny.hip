// The MSM of SMALL circuits (automatic up to 2^13 terms per MSM, msm.hip MSM_TINY_MAX_LOG2N): 8-bit windows and three launches.
//
// The bucket pipeline of msm_kernels.hip.h (count -> scan -> scatter -> second sort level -> accumulate -> combine -> long buckets -> redo ->
// row / column sums -> bit planes -> plane sum: eleven launches) is built for millions of entries.  At n = 2^12 every one of its stages runs for
// 5 - 50 us on a chip that is idle around it, and the MSM takes 0.24 ms whatever n is (profiles/r06_small_timeline_12.txt): the reduce chain
// alone -- a tree over 2^12 buckets, up to 12 serial doublings for the bit planes, the plane sum -- is 0.11 ms.  The reference meets the same
// effect from the other side: its bucket width shrinks with n (get_optimal_bucket_width, runtime_states.hpp:9-63) because buckets that outnumber
// their entries cost more to reduce than to fill.
//
// Here: MsmCfg<8> -- 32 windows of 8 bits (31 x 8 + 1 x 7 = 255), signed digits, 2^7 = 128 buckets shared by all windows thanks to the window
// tables T[w][i] = 2^(offset w) P_i (32 n points: 8 MiB at n = 2^12) -- and NO sort:
//   k_tiny_recode   one thread per scalar: from_montgomery, 32 signed digits -> one byte per digit (|d| <= 128 ... filed as the bucket number)
//                   in window-major order, the 32 signs as one word.
//   k_tiny_buckets  one block per (bucket, slice of the digit array, MSM of the batch): the block scans its slice of the digit bytes for ITS
//                   bucket number (the whole array is 32 n bytes: L2-resident, read by every block), lists the hits in LDS, and its 64 quads sum the
//                   listed table points with the complete quad-cooperative mixed addition (curve_quad.hip.h: every special case handled, no redo
//                   queue); a tree over the 64 quads leaves ONE point per block.  Dependent operations: entries / (128 S 64) additions + 6.
//   k_tiny_final    one block per MSM: slices summed, then sum_b b B_b WITHOUT bit planes and their doublings: it equals the sum of all suffix
//                   sums sum_j (sum_{b >= j} B_b) -- a parallel suffix scan (6 levels) and a tree (6 levels) over 64 quads, two buckets per quad
//                   -> the reference's Jacobian.
// Measured (profiles/r06_tiny_msm.txt; a level of these chains = one quad addition, ~4 us): 2^12 terms 0.159 ms stand-alone against 0.251 through the 13-bit configuration, 2^10 0.127 vs 0.233, a batch of four 2^12-term MSMs 0.249 vs 0.315.
//
// Results are the same group elements as the bucket pipeline's (tests: every golden and oracle case at the sizes this path takes, batches,
// ragged n, `from`, all-equal scalars, points at infinity, P and -P).  Selected by msm_auto_window (msm.hip) as window width 8.
#include "msm_kernels.hip.h"

namespace bbg {

namespace {
constexpr int TC = 8;                       // the widest window
using TK = MsmCfg<TC>;
constexpr int T_WINDOWS = TK::windows;      // 32
constexpr int T_BUCKETS = TK::buckets;      // 128
static_assert(T_WINDOWS == 32 && T_BUCKETS == 128, "the tiny path is written for 8-bit windows");
static_assert(MSM_MAX_WINDOWS >= T_WINDOWS, "recode_digits' digit array");
constexpr int T_MAX_SLICES = 16;
constexpr int T_LANE_MIN = 384;             // listed entries from which a block sums them one LANE per entry instead of one quad per entry
constexpr int T_LIST = 8192;                // hits a block may list before it sums them (the scan stops while a step of 4096 digits still fits)

// mags[set][w * n_pad + i] = bucket number of digit w of scalar i (0 = no contribution); signs[set][i] bit w = digit negative
__global__ void __launch_bounds__(256) k_tiny_recode(const MsmBatch batch, uint32_t n_pad, uint8_t* mags, uint32_t* signs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int set = blockIdx.y;
    if (i >= n_pad) return;
    mags += (size_t)set * T_WINDOWS * n_pad;
    signs += (size_t)set * n_pad;
    uint32_t mag[MSM_MAX_WINDOWS], sg = 0;
    const bool live = i < batch.n[set];
    if (live) recode_digits<TC>(batch.scalars[set], i, mag, sg);
#pragma unroll
    for (int w = 0; w < T_WINDOWS; w++) mags[(size_t)w * n_pad + i] = live ? (uint8_t)mag[w] : (uint8_t)0; // (bucket numbers <= 128 fit a byte)
    signs[i] = sg;
}

// One block: bucket b = blockIdx.x / S + 1, slice blockIdx.x % S of the digit array (in 16-byte words), MSM blockIdx.y.
// The scan goes on, 256 words at a time, until the list could overflow with the next 4096 digits -- normally the whole slice is listed before
// the first addition, so the quads then run their additions back to back (a round per 256 words made every round pay the first gather's
// latency: 98 us instead of 40 at n = 2^12).  A listed entry carries everything the gather needs: sign | window | index.
__global__ void __launch_bounds__(256) k_tiny_buckets(const MsmBatch batch, uint32_t n_pad, uint32_t S, const uint8_t* __restrict__ mags,
                                                      const uint32_t* __restrict__ signs, const Affine* __restrict__ table, size_t n_srs, Xyzz* parts)
{
    __shared__ uint32_t list[T_LIST];
    __shared__ uint32_t count;
    __shared__ Xyzz sm[32];
    const int tid = threadIdx.x, lt = tid >> 2, qd = tid & 3;
    const int set = blockIdx.y;
    const uint32_t b = blockIdx.x / S + 1, slice = blockIdx.x % S;
    mags += (size_t)set * T_WINDOWS * n_pad;
    signs += (size_t)set * n_pad;
    const size_t from = batch.from[set];
    const uint32_t words = (uint32_t)(((size_t)T_WINDOWS * n_pad) >> 4); // n_pad is a multiple of 16: a word never straddles two windows
    const uint32_t per = (words + S - 1) / S;
    const uint32_t w0 = slice * per, w1 = (w0 + per < words) ? w0 + per : words;
    const uint32_t wpw = n_pad >> 4; // words per window
    const uint4* mw = reinterpret_cast<const uint4*>(mags);
    Xyzz acc = xyzz_inf(), lane_acc = xyzz_inf();
    bool lane_used = false;
    if (tid == 0) count = 0;
    __syncthreads();
    uint32_t r0 = w0;
    while (r0 < w1) {
        // ---- list: 256 words (4096 digits) per step while the list has room for a step in which every digit hits
        uint32_t m = 0;
        do {
            const uint32_t wi = r0 + tid;
            if (wi < w1) {
                const uint4 v = mw[wi];
                const uint32_t x[4] = { v.x, v.y, v.z, v.w };
                const uint32_t w = wi / wpw, i0 = (wi - w * wpw) << 4; // window and first scalar of this word
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // bytes equal to b = zero bytes of x ^ (b, b, b, b); a borrow can flag a byte above a zero byte: every flagged byte is compared
                    const uint32_t t = x[k] ^ (b * 0x01010101u);
                    uint32_t hit = (t - 0x01010101u) & ~t & 0x80808080u;
                    while (hit) {
                        const int byte = __builtin_ctz(hit) >> 3;
                        hit &= hit - 1;
                        if (((x[k] >> (8 * byte)) & 0xffu) == b) {
                            const uint32_t i = i0 + (uint32_t)(4 * k + byte);
                            const uint32_t neg = (signs[i] >> w) & 1u;
                            list[atomicAdd(&count, 1u)] = (neg << 31) | (w << 24) | i; // i < 2^20 (msm_run_tiny), w < 32
                        }
                    }
                }
            }
            r0 += 256;
            __syncthreads();
            m = count;
            __syncthreads(); // everybody has read the count before the next step adds to it: the loop condition is the same for all
        } while (r0 < w1 && m + 4096 <= (uint32_t)T_LIST);
        // ---- sum: the next entry's table point is requested before the current addition
        auto fetch = [&](uint32_t idx) {
            const uint32_t g = list[idx];
            return aff_neg_if(aff_load(table + (size_t)((g >> 24) & 31u) * n_srs + from + (g & 0xffffffu)), (g >> 31) != 0);
        };
        if (m >= (uint32_t)T_LANE_MIN) {
            // plenty of entries: ONE LANE per entry in turn (a quad-cooperative addition costs ~1.5x the instructions of a one-lane one spread
            // over four lanes -- right when the lanes beside it would idle, wrong when every lane has work: a batch of four 2^12-term MSMs
            // has 2^19 entries for the chip's 2^16 lanes); the four lanes of a quad are summed once, at the end
            uint32_t e = tid;
            Affine p = aff_inf();
            bool have = e < m;
            if (have) p = fetch(e);
            while (have) {
                const Affine pc = p;
                e += 256;
                have = e < m;
                if (have) p = fetch(e);
                lane_acc = xyzz_madd(lane_acc, pc);
            }
            lane_used = true; // (uniform: m is)
        } else {
            uint32_t e = lt;
            Affine p = aff_inf();
            bool have = e < m;
            if (have) p = fetch(e);
            while (have) {
                const Affine pc = p;
                e += 64;
                have = e < m;
                if (have) p = fetch(e);
                acc = xyzz_madd_q4(acc, pc, qd);
            }
        }
        __syncthreads();
        if (tid == 0) count = 0;
        __syncthreads();
    }
    if (lane_used) acc = xyzz_add_q4(acc, quad_sum4(lane_acc, qd), qd);
    acc = block_reduce_q4(acc, sm, 64);
    if (tid == 0) xyzz_store(parts + ((size_t)set * T_BUCKETS + (b - 1)) * S + slice, acc);
}

// One block of 64 quads (one wave per SIMD: a level costs one quad addition, not two) per MSM.  v_b = sum of bucket b's slices;
// sum_b b v_b = sum_j (sum_{b >= j} v_b), the sum of all suffix sums.  Quad t owns buckets 2t + 1 and 2t + 2 (x = v_{2t+1}, y = v_{2t+2}):
//   pair sum p_t = x + y -> suffix scan of the pair sums over the quads (6 levels) -> the two suffix sums of the quad are
//   (tail + x + y) and (tail + y) with tail = the scan's value of the NEXT quad; their sum is what the quad contributes -> tree (6 levels).
__global__ void __launch_bounds__(2 * T_BUCKETS) k_tiny_final(const Xyzz* __restrict__ parts, uint32_t S, Jacobian* out)
{
    constexpr int NQ = T_BUCKETS / 2;
    __shared__ Xyzz sm[2][NQ];
    const int tid = threadIdx.x, lt = tid >> 2, q = tid & 3;
    parts += ((size_t)blockIdx.x * T_BUCKETS + 2 * lt) * S;
    out += blockIdx.x;
    Xyzz x = xyzz_load(parts), y = xyzz_load(parts + S);
    for (uint32_t s = 1; s < S; s++) {
        x = xyzz_add_q4(x, xyzz_load(parts + s), q);
        y = xyzz_add_q4(y, xyzz_load(parts + S + s), q);
    }
    const Xyzz pair = xyzz_add_q4(x, y, q);
    // inclusive suffix sums of the pair sums (Hillis-Steele, double-buffered in LDS)
    Xyzz v = pair;
    int cur = 0;
    for (int d = 1; d < NQ; d <<= 1) {
        if (q == 0) sm[cur][lt] = v;
        __syncthreads();
        if (lt + d < NQ) v = xyzz_add_q4(v, sm[cur][lt + d], q);
        cur ^= 1;
    }
    // v = sum over quads >= lt of their pairs = the suffix sum at bucket 2 lt + 1; the one at bucket 2 lt + 2 is v - x: instead of a
    // subtraction, contribution = (v) + (v - x) = 2 (tail + y) + x with tail = the NEXT quad's v
    if (q == 0) sm[cur][lt] = v;
    __syncthreads();
    const Xyzz tail = lt + 1 < NQ ? sm[cur][lt + 1] : xyzz_inf();
    const Xyzz ty = xyzz_add_q4(tail, y, q);
    Xyzz c = xyzz_add_q4(xyzz_dbl_q4(ty, q), x, q);
    __syncthreads();
    c = block_reduce_q4(c, sm[0], NQ);
    if (tid == 0) {
        const Jacobian j = xyzz_to_jacobian(c);
        fe_store<FqP>(&out->x, j.x);
        fe_store<FqP>(&out->y, j.y);
        fe_store<FqP>(&out->z, j.z);
    }
}
} // namespace

template int srs_build_tables_c<TC>(const void*, size_t, void*, hipStream_t);

// slices per bucket: enough blocks to put a wave on most SIMDs, and a chain of about a dozen additions per quad
static uint32_t tiny_slices(size_t max_n, int sets)
{
    size_t s = (max_n + 3071) / 3072; // entries per (bucket, slice) = 32 n / (128 S); 64 quads -> n / (256 S) additions per quad
    while ((size_t)T_BUCKETS * s * sets < 256 && s < 4) s++;
    while (s > 1 && (size_t)T_BUCKETS * s * sets > 512) s >>= 1; // a batch fills the chip with one or two blocks per CU: fuller blocks, summed one lane per entry
    uint32_t p = 1;
    while (p < s) p <<= 1;
    return p > (uint32_t)T_MAX_SLICES ? (uint32_t)T_MAX_SLICES : p;
}

int msm_run_tiny(bbg_ctx* ctx, const Srs& srs, const void* table, int sets, const void* const* d_scalars, const size_t* from, const size_t* n,
                 void* d_out_jac, hipStream_t st, const void* h_scalars)
{
    size_t max_n = 0;
    MsmBatch batch;
    for (int k = 0; k < MSM_BATCH_MAX; k++) {
        batch.scalars[k] = k < sets ? (const Fr*)d_scalars[k] : nullptr;
        batch.n[k] = k < sets ? (uint32_t)n[k] : 0u;
        batch.from[k] = k < sets ? (uint32_t)from[k] : 0u;
        if (k < sets && n[k] > max_n) max_n = n[k];
    }
    if (max_n > ((size_t)1 << 20)) { set_error("msm_run_tiny: 8-bit windows are for small MSMs (n <= 2^20)"); return BBG_E_INVALID; }
    const uint32_t n_pad = (uint32_t)((max_n + 15) & ~(size_t)15);
    const uint32_t S = tiny_slices(max_n, sets);
    const size_t mags_bytes = align_up((size_t)sets * T_WINDOWS * n_pad, 256), signs_bytes = align_up((size_t)sets * n_pad * 4, 256);
    const size_t parts_bytes = align_up((size_t)sets * T_BUCKETS * S * sizeof(Xyzz), 256); // per reduce slot
    int rc = msm_ensure_aux_streams(ctx);
    if (rc) return rc;
    // (a growing buffer is released with hipFree, which waits for whatever still reads it)
    rc = ensure_buffer(&ctx->msm_tiny.buf, &ctx->msm_tiny.bytes, mags_bytes + signs_bytes + bbg_ctx::MSM_SLOTS * parts_bytes);
    if (rc) return rc;
    // The last kernel -- one block per MSM: the chip is idle beside it -- runs on this call's reduce stream like the bucket pipeline's reduce
    // phase (msm_async_reduce; bbg_join / msm_join wait for it), so that the NEXT MSM's recode and bucket kernels run beside it; the
    // per-block bucket sums it reads are double-buffered by reduce slot.
    const int slot = (int)(ctx->msm_seq++ % bbg_ctx::MSM_SLOTS);
    const bool overlap = ctx->msm_async_reduce;
    hipStream_t rst = overlap ? ctx->aux_streams[slot] : st;
    const uint64_t layout = ((uint64_t)n_pad << 16) | ((uint64_t)sets << 8) | S;
    if (ctx->msm_tiny_layout != layout) {
        // another shape puts the bucket sums elsewhere: a last kernel still running on a reduce stream reads where this call is about to write
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++)
            if (ctx->ev_done_valid[k]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[k], 0));
        ctx->msm_tiny_layout = layout;
    }
    if (ctx->ev_done_valid[slot]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[slot], 0)); // this slot's bucket sums were last read two MSMs ago
    uint8_t* mags = (uint8_t*)ctx->msm_tiny.buf;
    uint32_t* signs = (uint32_t*)(mags + mags_bytes);
    Xyzz* parts = (Xyzz*)(mags + mags_bytes + signs_bytes + (size_t)slot * parts_bytes);
    if (h_scalars) BBG_HIP(hipMemcpyAsync((void*)d_scalars[0], h_scalars, n[0] * 32, hipMemcpyHostToDevice, st)); // bbg_msm: a batch of one
    {
        ProfScope ps(ctx, "msm_recode", st);
        hipLaunchKernelGGL(k_tiny_recode, dim3(grid_for(n_pad, 256), sets), dim3(256), 0, st, batch, n_pad, mags, signs);
    }
    {
        ProfScope ps(ctx, "msm_accumulate", st);
        hipLaunchKernelGGL(k_tiny_buckets, dim3(T_BUCKETS * S, sets), dim3(256), 0, st, batch, n_pad, S, mags, signs, (const Affine*)table, srs.n, parts);
    }
    if (overlap) {
        BBG_HIP(hipEventRecord(ctx->ev_acc[slot], st));
        BBG_HIP(hipStreamWaitEvent(rst, ctx->ev_acc[slot], 0));
    }
    {
        ProfScope ps(ctx, "msm_reduce", rst);
        hipLaunchKernelGGL(k_tiny_final, dim3(sets), dim3(2 * T_BUCKETS), 0, rst, parts, S, (Jacobian*)d_out_jac);
    }
    if (overlap) {
        BBG_HIP(hipEventRecord(ctx->ev_done[slot], rst));
        ctx->ev_done_valid[slot] = true;
    }
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

} // namespace bbg
