// Bucket-method multi-scalar multiplication over BN254 G1 for gfx950.
//
// Computes what scalar_multiplication::pippenger / pippenger_unsafe compute (reference
// ecc/curves/bn254/scalar_multiplication/scalar_multiplication.cpp:853-929): sum_i s_i * P_i.
//
// The reference pipeline is: GLV split + signed-window recode (compute_wnaf_states :188-252) -> per-round radix
// sort by bucket (organize_buckets :260-271, process_buckets.cpp:7-62) -> affine-trick bucket accumulation
// (reduce_buckets :445-521, add_affine_points :305-340) -> running-sum + `w` doublings per round
// (evaluate_pippenger_rounds :720-838).  The GPU pipeline keeps the bucket method but is re-designed around what
// the SRS being FIXED and resident in 288 GB of HBM allows:
//
//   * registration (bbg_srs_register, the Pippenger-constructor hook) precomputes the window tables
//     T[w][i] = 2^(16 w) * P_i, w = 0..15, in affine form.  They play the role of the reference's endomorphism
//     table (generate_pippenger_point_table :104-112): more precomputation instead of the GLV split, and -- because
//     every window's weight is already in the table -- all 16 windows share ONE set of 2^15 buckets: no per-round
//     running sums, no inter-round doublings.
//   * k_recode: scalar -> from_montgomery -> 16 signed 16-bit digits d in [-2^15, 2^15]; entry key = |d|
//     (0 = no contribution), value = sign | window | point index.
//   * radix sort of the (key, value) pairs by key: bucket b's contributions become one contiguous run.
//   * k_accumulate: each bucket's run is split between T lanes; every lane gathers its table points (64-byte
//     loads) and sums them with complete mixed XYZZ additions (curve.hip.h) -> T partial sums per bucket.
//   * k_bucket_sum, k_rowcol, k_final: sum_b b * B_b evaluated as 256 * sum_hi hi*Row_hi + sum_lo (lo+1)*Col_lo
//     with LDS tree reductions (short dependency chains; a single EC addition is ~6 us of latency on one wave).
//
// All exceptional cases of the group law are handled (the reference's handle_edge_cases = true behaviour), so
// pippenger_unsafe's "attempted to invert zero" failure mode (:317-318) does not exist here.
#include "msm_cfg.h"
#include "curve.hip.h"
#include "curve_quad.hip.h"

#include <algorithm>
#include <cstring>

namespace bbg {

static int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// ---------------------------------------------------------------------------------- synthetic SRS
// P_i = (a + i*s) * G : thread t owns CH consecutive points; start by double-and-add, then madd steps, normalise
// with one inversion per thread.
constexpr int SYNTH_CH = 16;
__device__ Xyzz g1_mul_u128(const Affine& g, unsigned __int128 k)
{
    Xyzz acc = xyzz_inf();
    for (int i = 127; i >= 0; i--) {
        acc = xyzz_dbl(acc);
        if ((uint64_t)(k >> i) & 1) acc = xyzz_madd(acc, g);
    }
    return acc;
}
__device__ Xyzz g1_mul_u64(const Affine& g, uint64_t k) { return g1_mul_u128(g, (unsigned __int128)k); }
__global__ void __launch_bounds__(128) k_srs_synth(Affine* out, size_t n, uint64_t a, uint64_t s)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * SYNTH_CH;
    if (i0 >= n) return;
    Affine G;
    G.x = Fq::one();
    Fq two = Fq::zero();
    two.v[0] = 2;
    G.y = fe_reduce_once(fe_to_mont(two));
    Affine S = xyzz_to_affine(g1_mul_u64(G, s));
    Xyzz q = g1_mul_u128(G, (unsigned __int128)a + (unsigned __int128)i0 * s); // no 64-bit wrap: P_i = (a + i*s) G over the integers
    Xyzz pts[SYNTH_CH];
    Fq prod[SYNTH_CH];
    Fq acc = Fq::one();
    int cnt = 0;
    for (int e = 0; e < SYNTH_CH && i0 + e < n; e++) {
        pts[e] = q;
        prod[e] = acc;
        acc = fe_mul(acc, fe_mul(q.zz, q.zzz));
        q = xyzz_madd(q, S);
        cnt++;
    }
    Fq inv = fq_invert(acc);
    for (int e = cnt - 1; e >= 0; e--) {
        Fq iz = fe_mul(inv, prod[e]);
        inv = fe_mul(inv, fe_mul(pts[e].zz, pts[e].zzz));
        Affine o;
        o.x = fe_reduce_once(fe_mul(pts[e].x, fe_mul(iz, pts[e].zzz)));
        o.y = fe_reduce_once(fe_mul(pts[e].y, fe_mul(iz, pts[e].zz)));
        aff_store(out + i0 + e, o);
    }
}

// P_i = k_i * G with k_i = mix64(seed + i) | 1: a synthetic SRS without small linear relations (the A + i*S
// family has P_0 + P_3 = P_1 + P_2, which the reference's pippenger_unsafe cannot digest -- see oracle_srs_hashed).
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
constexpr int HASH_CH = 4;
__global__ void __launch_bounds__(128) k_srs_hashed(Affine* out, size_t n, uint64_t seed)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i0 = t * HASH_CH;
    if (i0 >= n) return;
    Affine G;
    G.x = Fq::one();
    Fq two = Fq::zero();
    two.v[0] = 2;
    G.y = fe_reduce_once(fe_to_mont(two));
    Xyzz pts[HASH_CH];
    Fq prod[HASH_CH];
    Fq acc = Fq::one();
    int cnt = 0;
    for (int e = 0; e < HASH_CH && i0 + e < n; e++) {
        Xyzz q = g1_mul_u64(G, mix64(seed + (uint64_t)(i0 + e)) | 1ULL);
        pts[e] = q;
        prod[e] = acc;
        acc = fe_mul(acc, fe_mul(q.zz, q.zzz));
        cnt++;
    }
    Fq inv = fq_invert(acc);
    for (int e = cnt - 1; e >= 0; e--) {
        Fq iz = fe_mul(inv, prod[e]);
        inv = fe_mul(inv, fe_mul(pts[e].zz, pts[e].zzz));
        Affine o;
        o.x = fe_reduce_once(fe_mul(pts[e].x, fe_mul(iz, pts[e].zzz)));
        o.y = fe_reduce_once(fe_mul(pts[e].y, fe_mul(iz, pts[e].zz)));
        aff_store(out + i0 + e, o);
    }
}

// ---------------------------------------------------------------------------------- last reduce stage, g1 helpers
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
static __device__ __forceinline__ Xyzz block_reduce_q4(Xyzz v, Xyzz* sm, int nlogical)
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
__global__ void __launch_bounds__(64, 1) k_final_sum(const Xyzz* __restrict__ planes, int nplanes, Jacobian* out)
{
    __shared__ Xyzz sm[16];
    planes += (size_t)blockIdx.x * MSM_MAX_PLANES; // one block per MSM of the batch
    out += blockIdx.x;
    Xyzz v = (int)threadIdx.x < nplanes ? xyzz_load(planes + threadIdx.x) : xyzz_inf();
    v = block_reduce(v, sm, MSM_MAX_PLANES);
    if (threadIdx.x == 0) {
        Jacobian j = xyzz_to_jacobian(v);
        fe_store<FqP>(&out->x, j.x);
        fe_store<FqP>(&out->y, j.y);
        fe_store<FqP>(&out->z, j.z);
    }
}
__global__ void __launch_bounds__(4 * MSM_MAX_PLANES) k_final_sum_q(const Xyzz* __restrict__ planes, int nplanes, Jacobian* out)
{
    __shared__ Xyzz sm[MSM_MAX_PLANES / 2];
    const int lt = threadIdx.x >> 2;
    planes += (size_t)blockIdx.x * MSM_MAX_PLANES; // one block per MSM of the batch
    out += blockIdx.x;
    Xyzz v = lt < nplanes ? xyzz_load(planes + lt) : xyzz_inf();
    int width = 1;
    while (width < nplanes) width <<= 1; // 15 planes -> 4 levels, not 5
    v = block_reduce_q4(v, sm, width);
    if (threadIdx.x == 0) {
        Jacobian j = xyzz_to_jacobian(v);
        fe_store<FqP>(&out->x, j.x);
        fe_store<FqP>(&out->y, j.y);
        fe_store<FqP>(&out->z, j.z);
    }
}
int msm_launch_final_sum(bool quad, const void* d_planes, int nplanes, void* d_out_jac, hipStream_t st, int sets)
{
    if (quad) hipLaunchKernelGGL(k_final_sum_q, dim3(sets), dim3(4 * MSM_MAX_PLANES), 0, st, (const Xyzz*)d_planes, nplanes, (Jacobian*)d_out_jac);
    else hipLaunchKernelGGL(k_final_sum, dim3(sets), dim3(64), 0, st, (const Xyzz*)d_planes, nplanes, (Jacobian*)d_out_jac);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// sum of n Jacobian points (g1_sum, reference c_bind.cpp:39-46): one block, serial per thread then tree.
__global__ void __launch_bounds__(256, 1) k_g1_sum(const Jacobian* __restrict__ pts, size_t n, Jacobian* out)
{
    __shared__ Xyzz sm[128];
    Xyzz acc = xyzz_inf();
    for (size_t i = threadIdx.x; i < n; i += 256) {
        Jacobian j;
        j.x = fe_load<FqP>(&pts[i].x);
        j.y = fe_load<FqP>(&pts[i].y);
        j.z = fe_load<FqP>(&pts[i].z);
        acc = xyzz_add(acc, xyzz_from_jacobian(j));
    }
    // tree depth follows the input count: combining the 8 partials of an 8-GPU MSM is 3 levels, not 8
    int width = 256;
    while (width > 1 && (size_t)(width >> 1) >= n) width >>= 1;
    acc = block_reduce(acc, sm, width);
    if (threadIdx.x == 0) {
        Jacobian j = xyzz_to_jacobian(acc);
        fe_store<FqP>(&out->x, j.x);
        fe_store<FqP>(&out->y, j.y);
        fe_store<FqP>(&out->z, j.z);
    }
}
// g1::affine_element(element) (element_impl.hpp:51-68) for n points
__global__ void __launch_bounds__(128) k_normalize(const Jacobian* __restrict__ pts, size_t n, Affine* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jacobian j;
    j.x = fe_load<FqP>(&pts[i].x);
    j.y = fe_load<FqP>(&pts[i].y);
    j.z = fe_load<FqP>(&pts[i].z);
    if ((j.x.v[7] >> 31) != 0) {
        aff_store(out + i, aff_inf());
        return;
    }
    // bring arbitrary 256-bit representatives into range first
    j.x = fe_reduce_once(fe_reduce_once(j.x));
    j.y = fe_reduce_once(fe_reduce_once(j.y));
    j.z = fe_reduce_once(fe_reduce_once(j.z));
    aff_store(out + i, xyzz_to_affine(xyzz_from_jacobian(j)));
}

// ---------------------------------------------------------------------------------- host side: width dispatch
static const int MSM_WIDTHS[] = {
#define X(c) c,
    BBG_MSM_TABLE_WIDTHS(X)
#undef X
};
constexpr int MSM_NUM_WIDTHS = (int)(sizeof(MSM_WIDTHS) / sizeof(MSM_WIDTHS[0]));
static_assert(MSM_NUM_WIDTHS == Srs::MAX_WIDTHS, "Srs::tables has one slot per compiled window width");

int msm_width_slot(int c)
{
    for (int k = 0; k < MSM_NUM_WIDTHS; k++)
        if (MSM_WIDTHS[k] == c) return k;
    return -1;
}
int msm_width_of_slot(int slot) { return slot >= 0 && slot < MSM_NUM_WIDTHS ? MSM_WIDTHS[slot] : 0; }
int msm_windows_for(int c)
{
    switch (c) {
#define X(c) case c: return MsmCfg<c>::windows;
        BBG_MSM_TABLE_WIDTHS(X)
#undef X
    }
    return 0;
}
int srs_build_tables(const void* d_points, size_t n, void* d_table, int c, hipStream_t st)
{
    switch (c) {
#define X(c) case c: return srs_build_tables_c<c>(d_points, n, d_table, st);
        BBG_MSM_TABLE_WIDTHS(X)
#undef X
    }
    set_error("srs_build_tables: window width not compiled");
    return BBG_E_INVALID;
}

// Window width for an n-term MSM.  The trade: windows x n mixed additions against a bucket reduction (combine, row / column sums: two
// full additions per bucket) that grows with 2^(C-1) and runs beside the NEXT call's sort -- the reference widens its buckets with n
// for the same reason (get_optimal_bucket_width, runtime_states.hpp:9-63).  The thresholds are measured (profiles/r03_window_sweep.txt,
// tests/tools/msm_window_sweep.py: pipelined MSMs, every compiled width, interleaved); msm_window = 0 selects them, a compiled width forces one.
#ifndef MSM_SMALL_WINDOW_MAX_LOG2N
#define MSM_SMALL_WINDOW_MAX_LOG2N 14 // the 13-bit configuration is the automatic choice up to 2^14 terms (profiles/r04_window13_sweep.txt: it wins by 4-10 % there, ties at 2^13-2^14, loses from 2^15 -- its four-bin second sort level serialises on LDS counters once partitions hold a thousand entries)
#endif
#ifndef MSM_TINY_MAX_LOG2N
#define MSM_TINY_MAX_LOG2N 13 // the 8-bit-window path (msm_tiny.hip) is the automatic choice up to this many terms (profiles/r06_tiny_msm.txt: at 2^13 it wins
                              // stand-alone, 0.204 vs 0.256 ms, and ties in a burst of independent MSMs, 0.156 vs 0.153; a 2^13-gate proof 2.15 -> 2.05 ms;
                              // at 2^14 it still wins stand-alone, 0.255 vs 0.274, but loses the burst, 0.222 vs 0.171)
#endif
int msm_auto_window(size_t n)
{
    if (n >= ((size_t)1 << 23)) return 22; // 2^23: 9.8 vs 10.1 ms, 2^24: 18.3 vs 19.1 ms (22 vs 20 bits, pipelined); 2^22: 5.25 vs 4.94
    if (n >= ((size_t)1 << 21)) return 20; // 2^21: 2.72 (20) / 2.73 (19) / 2.79 (17); 2^22: 5.13 (20) / 5.56 (19)
    if (n >= ((size_t)1 << 20)) return 19; // 2^20: MSM + NTT step 1.48 (19) / 1.49 (17) / 1.54 (20) / 1.53 (16): the 29-bit-limb accumulation made
                                           // an entry cheaper, so one more window and half the buckets pay (20 bits won with the 32-bit limbs)
    // (+ 1024: StandardPLONK commits to n + 1 coefficients over an SRS of n + 1 points -- the same configuration as its n-term MSMs)
    if (n > ((size_t)1 << MSM_SMALL_WINDOW_MAX_LOG2N) + 1024) return 16; // 2^19: 0.80 (16) / 0.85 (19); 2^18: 0.51 vs 0.55 (17); 2^16: 0.233 vs 0.235 (17)
    if (n <= ((size_t)1 << MSM_TINY_MAX_LOG2N) + 1024) return MSM_TINY_WIDTH; // r6: 8-bit windows, three launches, no sort (msm_tiny.hip; profiles/r06_tiny_msm.txt)
    return 13;                             // small circuits: 2^12 buckets instead of 2^15 (r4; profiles/r04_window13_sweep.txt)
}
int msm_pick_window(const bbg_ctx* ctx, size_t n)
{
    if (ctx->msm_window && msm_width_slot(ctx->msm_window) >= 0) return ctx->msm_window;
    return msm_auto_window(n);
}

int srs_synth_linear(bbg_ctx*, uint64_t a, uint64_t s, size_t n, void* d_points, hipStream_t st)
{
    if (n == 0) return BBG_OK;
    hipLaunchKernelGGL(k_srs_synth, dim3(grid_for((n + SYNTH_CH - 1) / SYNTH_CH, 128)), dim3(128), 0, st, (Affine*)d_points, n, a,
                       s);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

int srs_synth_hashed(bbg_ctx*, uint64_t seed, size_t n, void* d_points, hipStream_t st)
{
    if (n == 0) return BBG_OK;
    hipLaunchKernelGGL(k_srs_hashed, dim3(grid_for((n + HASH_CH - 1) / HASH_CH, 128)), dim3(128), 0, st, (Affine*)d_points, n,
                       seed);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// window width and table for MSMs of at most max_n terms over `srs`; build = false only answers (bbg_msm_plan)
static int msm_choose(bbg_ctx* ctx, Srs& srs, size_t max_n, bool build, hipStream_t st, int* c_out)
{
    int c = msm_pick_window(ctx, max_n);
    if (!ctx->msm_window && !srs.tables[msm_width_slot(c)] && max_n * 4 <= srs.n) {
        // a short MSM over a long SRS (automatic width only): the width chosen for n has no table yet and building one costs
        // windows x srs.n x 64 bytes for a call that touches a fraction of it -- use the resident table whose width is nearest instead
        int best = -1;
        for (int k = 0; k < MSM_NUM_WIDTHS; k++)
            if (srs.tables[k] && (best < 0 || abs(MSM_WIDTHS[k] - c) < abs(MSM_WIDTHS[best] - c))) best = k;
        if (best >= 0) c = MSM_WIDTHS[best];
    }
    if (msm_windows_for(c) > 16 && srs.n > ((size_t)1 << 26)) { // the 20-window configuration indexes 2^26 points (msm_cfg.h)
        if (ctx->msm_window == c) {
            set_error("bbg_msm: msm_window = 13 indexes at most 2^26 points per device");
            return BBG_E_INVALID;
        }
        c = 16;
    }
    *c_out = c;
    if (!build) return BBG_OK;
    void*& table = srs.tables[msm_width_slot(c)];
    if (!table) { // first MSM of this width on this SRS: build its window tables from the plain points (one-off)
        hipError_t e = hipMalloc(&table, srs.n * (size_t)msm_windows_for(c) * sizeof(Affine));
        if (e != hipSuccess) {
            table = nullptr;
            return hip_fail(e, "hipMalloc(SRS window tables)", __FILE__, __LINE__);
        }
        int rc = srs_build_tables(srs.points, srs.n, table, c, st);
        if (rc) { // never keep a table whose contents were not built: later MSMs of this width would read garbage
            (void)hipFree(table);
            table = nullptr;
            return rc;
        }
    }
    return BBG_OK;
}
int msm_plan(bbg_ctx* ctx, const Srs* srs, size_t n, int* c_out)
{
    if (!srs) {
        *c_out = msm_pick_window(ctx, n);
        return BBG_OK;
    }
    std::lock_guard<std::mutex> srs_lk(const_cast<Srs*>(srs)->mu);
    return msm_choose(ctx, const_cast<Srs&>(*srs), n, false, nullptr, c_out);
}

int msm_run_batch(bbg_ctx* ctx, Srs& srs, int sets, const void* const* d_scalars, const size_t* from, const size_t* n, void* d_out_jac, hipStream_t st,
                  const void* h_scalars)
{
    if (sets < 1 || sets > BBG_MSM_BATCH_MAX) {
        set_error("bbg_msm_batch: 1 .. BBG_MSM_BATCH_MAX MSMs per batch");
        return BBG_E_INVALID;
    }
    if (srs.n > ((size_t)1 << MSM_IDX_BITS)) {
        set_error("bbg_msm: SRS larger than 2^27 points per device is not supported (shard it by point range across devices: bbg_multi_*)");
        return BBG_E_INVALID;
    }
    size_t max_n = 0;
    for (int k = 0; k < sets; k++) {
        if (from[k] > srs.n || n[k] > srs.n - from[k]) {
            set_error("bbg_msm: range [from, from+n) exceeds the registered SRS");
            return BBG_E_INVALID;
        }
        if (n[k] && !d_scalars[k]) {
            set_error("bbg_msm: null scalars");
            return BBG_E_INVALID;
        }
        max_n = std::max(max_n, n[k]);
    }
    if (max_n == 0) {
        // pippenger(): n == 0 -> point at infinity (scalar_multiplication.cpp:868-872)
        uint64_t inf[12 * BBG_MSM_BATCH_MAX] = { 0 };
        for (int k = 0; k < sets; k++) inf[12 * k + 3] = 1ULL << 63;
        BBG_HIP(hipMemcpyAsync(d_out_jac, inf, (size_t)96 * sets, hipMemcpyHostToDevice, st));
        BBG_HIP(hipStreamSynchronize(st));
        return BBG_OK;
    }
    int c = 0;
    std::lock_guard<std::mutex> srs_lk(srs.mu); // until the accumulation that gathers from the table is queued (Srs::mu)
    int rc = msm_choose(ctx, srs, max_n, true, st, &c);
    if (rc) return rc;
    const void* table = srs.tables[msm_width_slot(c)];
    if (c == MSM_TINY_WIDTH) return msm_run_tiny(ctx, srs, table, sets, d_scalars, from, n, d_out_jac, st, h_scalars);
    switch (c) {
#define X(c) case c: return msm_run_c<c>(ctx, srs, table, sets, d_scalars, from, n, d_out_jac, st, h_scalars);
        BBG_MSM_WIDTHS(X)
#undef X
    }
    set_error("bbg_msm: window width not compiled");
    return BBG_E_INVALID;
}

int msm_run(bbg_ctx* ctx, Srs& srs, const void* d_scalars, size_t from, size_t n, void* d_out_jac, hipStream_t st, const void* h_scalars)
{
    return msm_run_batch(ctx, srs, 1, &d_scalars, &from, &n, d_out_jac, st, h_scalars);
}

// makes the context stream wait for every reduce phase queued on the auxiliary stream (no host sync)
int msm_join(bbg_ctx* ctx, hipStream_t st)
{
    for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++)
        if (ctx->ev_done_valid[k]) BBG_HIP(hipStreamWaitEvent(st, ctx->ev_done[k], 0));
    return BBG_OK;
}

int g1_sum_device(bbg_ctx*, const void* d_jacs, size_t n, void* d_out, hipStream_t st)
{
    hipLaunchKernelGGL(k_g1_sum, dim3(1), dim3(256), 0, st, (const Jacobian*)d_jacs, n, (Jacobian*)d_out);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}
int g1_normalize_device(const void* d_jacs, size_t n, void* d_out, hipStream_t st)
{
    if (n == 0) return BBG_OK;
    hipLaunchKernelGGL(k_normalize, dim3(grid_for(n, 128)), dim3(128), 0, st, (const Jacobian*)d_jacs, n, (Affine*)d_out);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

} // namespace bbg
