// Compile-time configuration of the bucket MSM (msm_kernels.hip.h) and the entry points every window width instantiates.
#pragma once
#include "bbg_internal.h"

namespace bbg {

// A configuration is named by its widest window C (the bucket count is 2^(C-1)) and is a per-call choice among the compiled ones
// (msm_pick_window, BBG_MSM_WIDTHS below): C = 16 (16 windows, 2^15 buckets) ... C = 22 (12 windows, 2^21 buckets).  Wider windows trade mixed
// additions (windows x n of them) for a bucket reduction that grows with 2^(C-1); the automatic rule follows n the way the reference's
// bucket width does (runtime_states.hpp:9-63).
//
// BALANCED windows: the scalar has 254 bits and the signed recoding needs one more for its carry, so W = ceil(255 / C) windows must cover
// exactly 255 bits: the first `nwide` windows are C bits wide, the others C - 1 (C = 20: 8 x 20 + 5 x 19; C = 22: 3 x 22 + 9 x 21; C = 17:
// 15 x 17).  Cutting W windows of C bits from the bottom instead leaves a top window with a handful of bits (C = 19: 7, C = 22: 12) whose n
// digits all land in the first few buckets -- one partition of the sort then holds n entries and its single block runs for a millisecond
// (measured, profiles/r03_window_sweep_a.txt).  Widths whose W equals a narrower width's (18 -> 17, 21 -> 20) are pointless and not compiled.
// C = 13 (r4: 20 windows, 2^12 buckets) is the small-circuit configuration: at n = 2^12 .. 2^16 the 2^15 buckets of C = 16 hold one to thirty
// entries each and the bucket reduction (two additions per bucket, per MSM of a batch) costs as much as the accumulation.
// Each configuration has its own window tables T[w][i] = 2^(table_offset(w)) P_i, built the first time it is used on an SRS, and its own
// translation unit (msm_wNN.hip) so that the configurations compile side by side.
template <int C> struct MsmCfg {
    static constexpr int c = C;
    static constexpr int windows = (254 + C) / C;           // C * windows >= 255
    static constexpr int nwide = 255 - windows * (C - 1);   // windows [0, nwide) have C bits, the rest C - 1
    static_assert(nwide >= 1 && nwide <= windows, "use the narrower configuration with the same number of windows");
    // A sorted value = sign (bit 31) | window | point index.  Up to 16 windows the window field has 4 bits and the index 27 (2^27 points per device);
    // narrower configurations (C = 13: 20 windows -- the small-circuit end, where 2^(C-1) buckets must not outnumber the entries) take a 5-bit
    // window field and a 26-bit index.  The reference's schedule word: 32 index bits (scalar_multiplication.hpp:24-29).
    static_assert(windows <= 32, "the window field of a sorted value has at most five bits");
    static constexpr int win_bits = windows > 16 ? 5 : 4;
    static constexpr int idx_bits = 31 - win_bits;
    static constexpr int width(int w) { return w < nwide ? C : C - 1; }
    static constexpr int offset(int w) { return w * (C - 1) + (w < nwide ? w : nwide); } // first scalar bit of window w; offset(windows) = 255
    // A narrow window's digits d fill only the lower half of the bucket range; filed under bucket d they would double the load of the lower
    // half of the sort's partitions (at n = 2^20, C = 20 exactly up to the second level's single-pass capacity).  They are filed under bucket
    // 2d against the table point of HALF the weight instead -- d 2^offset P = (2d) 2^(offset - 1) P --: every partition sees every window.
    static constexpr int scale(int w) { return w < nwide ? 0 : 1; }                 // bucket = |digit| << scale(w)
    static constexpr int table_offset(int w) { return offset(w) - scale(w); }       // table[w][i] = 2^table_offset(w) P_i
    static constexpr int buckets = 1 << (C - 1);       // |digit| in [1, 2^(C-1)]
    static constexpr int lo_bits = C > 11 ? C - 11 : 0; // sort partitions = buckets >> lo_bits (+1) = 1025; bins per partition = 2^lo_bits (C = 8, msm_tiny.hip, has no sort)
    static constexpr int parts = (buckets >> lo_bits) + 1;
    static constexpr int log_cols = C / 2;             // bucket index (0-based) = hi * cols + lo
    static constexpr int log_rows = C - 1 - log_cols;
    static constexpr int planes = C - 1;               // bit planes of the weight idx + 1 <= 2^(C-1)
};
constexpr int MSM_MAX_WINDOWS = 32; // (C = 8, the small-circuit path msm_tiny.hip: 32 windows)
// The widest point index a sorted value carries (MsmCfg<C>::idx_bits of the configurations with up to 16 windows): up to 2^27 points per
// device -- the whole 100.8 M-point Ignition SRS (2^26.6) fits one device's format, as it fits its HBM (12 windows x 64 B x 2^27 = 103 GB).
// Beyond 2^27 points an SRS is sharded by point range across devices (bbg_multi_*).  (Round 3: 26 bits -- a bit was left unused between the
// window field and the sign.)  The 20-window configuration C = 13 indexes 2^26 points; it is chosen for small MSMs only.
constexpr int MSM_IDX_BITS = 27;
constexpr int MSM_MAX_PLANES = 32;

// every width with a translation unit msm_wNN.hip (X-macro: dispatch tables in msm.hip, option parsing, table slots in Srs)
#define BBG_MSM_WIDTHS(X) X(13) X(16) X(17) X(19) X(20) X(22)
// every width an SRS may hold window tables for: the bucket pipeline's widths and the 8-bit windows of the small-circuit path (msm_tiny.hip:
// three launches, no sort -- its own entry point msm_run_tiny, not an instantiation of msm_run_c)
constexpr int MSM_TINY_WIDTH = 8;
#define BBG_MSM_TABLE_WIDTHS(X) X(8) BBG_MSM_WIDTHS(X)

// a batch of `sets` MSMs (1 .. BBG_MSM_BATCH_MAX; MSM k: n[k] terms from point from[k], result at d_out_jac + 96 k) with C-bit windows over
// `table` (window tables of this width) through one launch set; defined in msm_kernels.hip.h, instantiated in msm_wNN.hip
template <int C>
int msm_run_c(bbg_ctx* ctx, const Srs& srs, const void* table, int sets, const void* const* d_scalars, const size_t* from, const size_t* n,
              void* d_out_jac, hipStream_t st, const void* h_scalars);
// the same contract on 8-bit windows without a sort (msm_tiny.hip); `table` = window tables of width MSM_TINY_WIDTH
int msm_run_tiny(bbg_ctx* ctx, const Srs& srs, const void* table, int sets, const void* const* d_scalars, const size_t* from, const size_t* n,
                 void* d_out_jac, hipStream_t st, const void* h_scalars);
// table[w * n + i] = 2^(MsmCfg<C>::table_offset(w)) P_i
template <int C> int srs_build_tables_c(const void* d_points, size_t n, void* d_table, hipStream_t st);
// the last reduce stage (sum of the bit planes -> Jacobian), shared by all widths (msm.hip)
int msm_launch_final_sum(bool quad, const void* d_planes, int nplanes, void* d_out_jac, hipStream_t st, int sets);

} // namespace bbg
