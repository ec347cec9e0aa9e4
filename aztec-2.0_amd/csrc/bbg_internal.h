// Internal declarations shared by the HIP translation units of libbbg.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bbg.h"

namespace bbg {

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BBG_HIP(expr)                                                                                                \
    do {                                                                                                             \
        hipError_t _e = (expr);                                                                                      \
        if (_e != hipSuccess) return ::bbg::hip_fail(_e, #expr, __FILE__, __LINE__);                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------- NTT
constexpr int NTT_MAX_PASSES = 4;

// Per-domain-size device tables: the GPU counterpart of evaluation_domain::compute_lookup_table()
// (reference polynomials/evaluation_domain.cpp:33-55,169-175).  Built once per log2(n), cached in the context.
struct NttDomain {
    unsigned log2n = 0;
    int passes = 0;
    bool use_pass8 = false; // register-resident radix-8 pass kernel (ntt_pass8.hip.h) vs the radix-2-in-LDS one
    int tile_log8 = 11;     // its tile: 2^11 elements (256 threads) or 2^12 (512 threads: 2^21 / 2^22 in two passes)
    bool inv_scaled = false; // the inverse inter-pass twiddles of pass 0 carry n^-1: ifft needs no scaling sweep
    int logR[NTT_MAX_PASSES] = { 0, 0, 0, 0 };
    int logW[NTT_MAX_PASSES] = { 0, 0, 0, 0 };
    void* consts = nullptr;                          // DomainConsts (device)
    void* tw_inter[2][NTT_MAX_PASSES] = {};          // [inverse][pass]: omega_{N_q}^{i*lo}, N_q entries (passes 0..p-2)
    void* tw_radix[2][NTT_MAX_PASSES] = {};          // [inverse][pass]: omega_{R_q}^j, R_q/2 entries
    void* tw_radix29[2][NTT_MAX_PASSES] = {};        // the same entries times 32 (R'-form, canonical): multipliers of the 29-bit-limb pass kernel
    uint64_t root_host[4] = { 0, 0, 0, 0 };          // host copy of the domain's root of unity (Montgomery form, canonical): poly.hip builds evaluation-point tables on the host
    void* coset_fwd = nullptr;                       // g^j, j < n
    void* coset_inv = nullptr;                       // n^-1 * g^-j, j < n
    size_t bytes = 0;
};

// ---------------------------------------------------------------------------------------------- MSM
struct Srs {
    size_t n = 0;          // number of (plain) points
    void* points = nullptr; // device: n affine points, 64 B each, Montgomery, canonical (= window 0 of a table below)
    static constexpr int MAX_WIDTHS = 7; // BBG_MSM_TABLE_WIDTHS (msm_cfg.h)
    void* tables[MAX_WIDTHS] = {}; // window tables T[w][i] = 2^(MsmCfg<C>::table_offset(w)) P_i (balanced windows, halved weight for the narrow ones: msm_cfg.h) per compiled width C (slot = msm_width_slot(C), msm.hip); built on first use
    int home_slot = -1;    // the table built at registration: its window 0 IS `points`, so it is never released before the handle
    int device = 0;
    // tables[] may be read, built (first MSM of a width) and released (bbg_memory_trim of the owning context) from DIFFERENT contexts'
    // threads: held from the choice of a table until every kernel that reads it is queued, and by the trim around its device
    // synchronisation + free (lock order: bbg_ctx::mu -> g_srs_mu -> Srs::mu)
    std::mutex mu;
};

struct MsmScratch {
    void* buf = nullptr;
    size_t bytes = 0;
};

} // namespace bbg

namespace bbg {
// Optional per-kernel timing with HIP events on the launch stream (bbg_profile_*): bench.py reads the average
// duration of the dominant kernels live from here, so the roofline numbers do not depend on an external profiler.
struct ProfEntry {
    std::vector<hipEvent_t> start, stop;
};
} // namespace bbg

struct bbg_ctx {
    bool prof_on = false;
    std::map<std::string, bbg::ProfEntry> prof;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::map<unsigned, bbg::NttDomain> domains;
    void* ntt_scratch = nullptr;
    size_t ntt_scratch_bytes = 0;
    void* staging = nullptr; // device staging for host-pointer entry points
    size_t staging_bytes = 0;
    bbg::MsmScratch msm;
    bbg::MsmScratch msm_tiny; // digit bytes, sign words and per-block bucket sums of the small-circuit MSM path (msm_tiny.hip)
    uint64_t msm_tiny_layout = 0; // (n_pad, sets, slices) the buffer was last laid out for: another shape moves the double-buffered bucket sums
    void* poly_scratch = nullptr; // evaluate / kate partial sums, pow tables
    size_t poly_scratch_bytes = 0;
    // MSM reduce phase may run on an auxiliary stream so that it overlaps the next MSM's sort / accumulation
    // MSM_SLOTS reduce phases may be in flight at once, each on its own auxiliary stream with its own working set, so the reduce
    // phases of consecutive MSMs (latency chains that use a sliver of the chip) overlap each other as well as the next accumulation.
    // Two, not more: with four (main + copy + 4 auxiliary streams = 6 > the runtime's 4 hardware queues) streams share queues and
    // pick up false dependencies -- measured 459 instead of 585 Mscalar-mul/s on the headline step, small proofs 10-20 % slower.
    static constexpr int MSM_SLOTS = 2;
    hipStream_t aux_stream = nullptr; // = aux_streams[0] (non-null once the streams exist)
    hipStream_t aux_streams[MSM_SLOTS] = {};
    hipEvent_t ev_acc[MSM_SLOTS] = {}, ev_done[MSM_SLOTS] = {};
    // host-buffer MSM (bbg_msm): the scalars travel in pieces on their own stream, the counting pass of a piece starts when it has landed
    static constexpr int UPLOAD_PIECES = 4;
    hipStream_t upload_stream = nullptr;
    hipEvent_t ev_upload[UPLOAD_PIECES] = {}, ev_upload_go = nullptr;
    int msm_upload_pieces = 1; // option "msm_upload_pieces": 1 = one copy in front of the MSM (default: measured FASTER, profiles/r02_host_msm_ab.txt)
    bool ev_done_valid[MSM_SLOTS] = {};
    unsigned long msm_seq = 0;
    size_t msm_layout_n = 0; // (scalars of the batch, MSMs in it, window width) of the layout the scratch arena currently holds: a change of any
    int msm_layout_sets = 0; // moves every region, so pending reduce phases are joined first (msm_run_c)
    int msm_layout_c = 0;
    void* msm_zero_buf = nullptr; // the arena whose leading zero-initialised regions (partition counters, redo flags) were cleared ...
    int msm_zero_c = 0;           // ... for this window width ...
    int msm_zero_sets = 0;        // ... and this many bucket sets (msm_layout's cap_sets)
    int msm_layout_sort = 1; // (the sort path is part of the layout: the library-sort path reserves rocPRIM's temporary storage)
    bool msm_async_reduce = false;
    int msm_reduce_quad = 14;   // reduce stages with four lanes per EC operation (curve_quad.hip.h): bit 0 combine (a THROUGHPUT kernel over all buckets: one lane per operation is cheaper, measured), 1 row/col, 2 planes, 3 sum
    int msm_acc_waves = 0;           // option "msm_acc_waves": lane segments per SIMD lane of the accumulation (0 = automatic), A/B
    bool msm_limbs29 = true;         // option "msm_limbs29": the accumulation's field arithmetic on 9 x 29-bit limbs (0 = 8 x 32, A/B)
    bool msm_accumulate_quad = true; // option "msm_accumulate_quad": small MSMs accumulate with four threads per lane segment (0 = one, A/B)
    bool msm_reduce_low_priority = true; // auxiliary stream created with the lowest priority (option "msm_reduce_priority" = 0 undoes it)
    std::map<uint32_t, void*> dpv_consts; // poly.hip: Z*_H division constants per (src, target, roots cut)
    std::map<uint32_t, void*> dpv_tables; // poly.hip: the Z*_H divisor per target-domain point, same key (poly_dpv_table; the prover's round 4)
    void* gp_totals = nullptr;  // quotient.hip: grand-product thread totals
    size_t gp_totals_bytes = 0;
    void* quot_setup = nullptr; // quotient.hip: derived challenges / constants
    size_t quot_setup_bytes = 0;
    int quotient_setup_plan = 1;     // option "quotient_setup_plan": the widgets' set-up blocks by the lanes of one wave side by side (quotient.hip k_quotient_setup_plan); 0 = the one-lane chain (A/B)
    int poly_limbs29 = 1;            // option "poly_limbs29": linear combinations and evaluations of coefficient arrays on 9 x 29-bit limbs, four terms per reduction (poly29.hip.h); 0 = the 32-bit kernels (A/B)
    int prover_fused_divide = 1;     // option "prover_fused_divide": round 4 divides by Z*_H inside the coset iFFT's first load (poly_dpv_table + ntt_coset_ifft_scaled) instead of a pass of its own; 0 = the separate pass (A/B)
    int prover_tail_window = 0;      // option "prover_tail_window" (A/B): window width of the commitments whose reduce phase ends a round (rounds 4 and 6: the host waits for them with the chip idle) -- fewer buckets, shorter tail, more windows; 0 = the automatic width
    bool prover_ntt_batch = true;    // option "prover_ntt_batch": the wires' iFFTs (round 1) and 4n coset forms of circuits up to 2^17 gates go through ONE launch set each (grid.y = wires) instead of one per wire (A/B)
    int prover_fail_round = 0;       // option "prover_fail_round" (tests only): the next bbg_prover_round<k> returns BBG_E_HIP once -- how the shim's fallback to the reference body is exercised
    int prover_early_cosets = -1; // -1 = from 2^18 gates (default), 0 = never, 1 = always. option "prover_early_cosets": the wires' 4n coset forms are queued behind round 1's last commitment (beside its reduce phase) instead of in front of round 3's grand product
    int prover_msm_batch = 4; // option "prover_msm_batch": commitments of a prover round per launch set (0 / 1 = one each; prover.hip commit())
    bool quotient_limbs29 = true; // option "quotient_limbs29": permutation / fixed-base / fused arithmetic + range + logic widgets on lazily reduced 29-bit limbs (quotient29.hip.h; 0 = the 32-bit kernels, A/B)
    bool quotient_fuse = true; // option "quotient_fuse": arithmetic + range + logic widgets of a chain in one pass over the wires (0 = one kernel each, A/B)
    int msm_window = 0; // 0 = automatic (msm_auto_window), or one of the compiled widths (BBG_MSM_WIDTHS)
    int msm_sort = 1; // 1 = fused recode + MSD partition sort (msm.hip), 0 = k_recode + rocPRIM radix sort + k_offsets
    int ntt_tile_log = 10; // log2(elements per LDS tile); 10/7 measured best on MI355X (profiles/r01_ntt_plan_sweep.txt)
    int ntt_max_logr = 7;
    int ntt_kernel = 2;      // 2 = k_ntt_pass8 where applicable (n >= 2^11), 1 = k_ntt_pass only
    int ntt_max_logr8 = 10;  // max log-radix per pass for k_ntt_pass8
    int ntt_big_tile = 1;    // option "ntt_big_tile": 1 = 2^21 as TWO passes over 4096-element tiles instead of three over 2048; 2 = 2^22 as well; 0 = never
    int ntt_lds_planes = 0;  // option "ntt_lds_planes": 2 = the tile stays in LDS between two steps (k_ntt_pass8), 1 = it moves one 16-byte plane at a time (k_ntt_pass8s: half the LDS, three blocks per CU), 0 = automatic (1 from 2^22)
    int ntt_limbs29 = -1;    // option "ntt_limbs29": 1 = pass arithmetic on lazily reduced 9 x 29-bit limbs (k_ntt_pass29), 0 = 8 x 32-bit (k_ntt_pass8 / 8s), -1 = automatic
    bool ntt_attr29_set = false;
    bool ntt_attr8s_set = false;
    bool ntt_attr8_set = false, ntt_attr_set = false; // dynamic-LDS attributes of the pass kernels set on this context's device
};

struct bbg_srs {
    bbg::Srs s;
    bbg_ctx* ctx = nullptr;
    // owners of the handle: the creator plus every bbg_prover built on it (bbg_srs_retain); bbg_srs_free drops one and releases the device
    // memory with the last -- a cache that replaces an entry (shim/bbg_barretenberg_shim.cpp) cannot pull the SRS from under a live prover
    std::atomic<int> refs{ 1 };
};

namespace bbg {
struct ProfScope {
    bbg_ctx* ctx;
    hipStream_t st;
    hipEvent_t stop = nullptr;
    ProfScope(bbg_ctx* c, const char* name, hipStream_t s) : ctx(c), st(s)
    {
        if (!c || !c->prof_on) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ProfEntry& e = c->prof[name];
        e.start.push_back(a);
        e.stop.push_back(b);
        (void)hipEventRecord(a, s);
        stop = b;
    }
    ~ProfScope()
    {
        if (stop) (void)hipEventRecord(stop, st);
    }
};
int ensure_buffer(void** buf, size_t* have, size_t need);
int ntt_run(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant,
            hipStream_t stream);
int ntt_ifft_to(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2n, hipStream_t stream);
int ntt_ifft_to_batch(bbg_ctx* ctx, int count, const void* const* d_in, void* const* d_out, unsigned log2n, hipStream_t stream);
constexpr int BBG_E_NOFUSE = -100; // internal: this domain's plan has no fused load; never returned through the C ABI
// coset iFFT of a[i] * scale[i] (scale: 2^log2n Montgomery-form values), the product taken as the first pass loads a[i]; BBG_E_NOFUSE
// when the domain's plan has no fused load (single-pass domains) -- the caller scales first and calls ntt_run
int ntt_coset_ifft_scaled(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, const void* d_scale, hipStream_t stream);
int ntt_coset_extend_batch(bbg_ctx* ctx, int count, const void* const* d_in, size_t n_in, void* const* d_out, unsigned log2n, hipStream_t stream);
void ntt_free_domain(NttDomain& d);
int ntt_coset_extend(bbg_ctx* ctx, const void* d_in, size_t n_in, void* d_out, unsigned log2n, hipStream_t stream);
int ntt_coset_split(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, size_t ext, hipStream_t stream);
int ntt_prepare(bbg_ctx* ctx, unsigned log2n);
int ntt_plan(bbg_ctx* ctx, unsigned log2n, int* passes, int* log_radix, int* kernel, int* tile_log);
int ntt_domain_consts(bbg_ctx* ctx, unsigned log2n, void** consts);
int ntt_domain_root_host(bbg_ctx* ctx, unsigned log2n, uint64_t out[4]);
int poly_binop(int op, const void* a, const void* b, void* r, size_t n, hipStream_t st);
int poly_evaluate(bbg_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t* z, uint64_t* out, hipStream_t st);
int poly_kate_opening(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t* z, uint64_t* f_out, hipStream_t st);
int poly_divide_pseudo_vanishing(bbg_ctx* ctx, void* d_evals, unsigned log2_src, unsigned log2_target, size_t roots_cut, hipStream_t st);
// the divisor as a per-point table (cached per context); *table stays valid until bbg_memory_trim / bbg_destroy
int poly_dpv_table(bbg_ctx* ctx, unsigned log2_src, unsigned log2_target, size_t roots_cut, const void** table, hipStream_t st);
int ntt_scale_powers(bbg_ctx* ctx, void* d_a, size_t count, const uint64_t* start, const uint64_t* base, hipStream_t stream);
int ntt_scale_geometric(bbg_ctx* ctx, void* d_a, size_t count, unsigned log2n, int which_base, uint64_t step, uint64_t e0, int mul_inv_log2,
                        hipStream_t st);
int ntt_root_pow(bbg_ctx* ctx, unsigned log2n, uint64_t e, int inverse, uint64_t* out, hipStream_t stream);
int ntt_fr_pow(bbg_ctx* ctx, const uint64_t* base, uint64_t e, uint64_t* out, hipStream_t stream);
int ntt_cross_dft(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2G, size_t len, unsigned log2n, int inverse, hipStream_t stream);
int field_op_device(int which, int op, const void* a, const void* b, void* out, size_t n, hipStream_t stream);
// h_scalars (optional): the n scalars are still on the HOST and d_scalars is where they belong on the device -- msm_run uploads them
// itself, piece by piece, overlapped with its first pass
int msm_run(bbg_ctx* ctx, Srs& srs, const void* d_scalars, size_t from, size_t n, void* d_out_jac,
            hipStream_t stream, const void* h_scalars = nullptr);
// `sets` MSMs over one SRS through ONE launch set (msm_cfg.h); result k at d_out_jac + 96 k
int msm_run_batch(bbg_ctx* ctx, Srs& srs, int sets, const void* const* d_scalars, const size_t* from, const size_t* n, void* d_out_jac, hipStream_t stream,
                  const void* h_scalars = nullptr);
// widest window an n-term MSM over srs (may be null) would use now (no table is built)
int msm_plan(bbg_ctx* ctx, const Srs* srs, size_t n, int* c_out);
int srs_synth_linear(bbg_ctx* ctx, uint64_t a, uint64_t s, size_t n, void* d_points, hipStream_t stream);
int msm_join(bbg_ctx* ctx, hipStream_t stream);
int srs_synth_hashed(bbg_ctx* ctx, uint64_t seed, size_t n, void* d_points, hipStream_t stream);
int g1_sum_device(bbg_ctx* ctx, const void* d_jacs, size_t n, void* d_out, hipStream_t stream);
} // namespace bbg
