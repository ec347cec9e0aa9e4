// C ABI of libbbg.so (declared in include/bbg.h).  Host-side plumbing only: contexts, device buffers, the SRS
// registry and the Ignition-transcript reader.  All arithmetic runs in the HIP kernels of ntt.hip / msm.hip.
#include "bbg_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <set>

namespace bbg {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    g_last_error = buf;
    return e == hipErrorOutOfMemory ? BBG_E_NOMEM : BBG_E_HIP;
}
int ensure_buffer(void** buf, size_t* have, size_t need)
{
    if (*have >= need && *buf) return BBG_OK;
    if (*buf) {
        BBG_HIP(hipDeviceSynchronize());
        BBG_HIP(hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    BBG_HIP(hipMalloc(buf, need));
    *have = need;
    return BBG_OK;
}
int srs_build_tables(const void* d_points, size_t n, void* d_table, int c, hipStream_t st);
int msm_pick_window(const bbg_ctx* ctx, size_t n);
void prover_report(const bbg_ctx* ctx, size_t* bytes, unsigned* count); // prover.hip: live bbg_prover handles of a context

// every live SRS handle of the process (bbg_memory_report totals the ones of a context; a handle may outlive its context, so the
// registry is not a member of bbg_ctx)
static std::mutex g_srs_mu;
static std::set<bbg_srs*> g_live_srs;
int permutation_grand_product(bbg_ctx* ctx, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n, const uint64_t* challenges,
                              void* d_z, hipStream_t st);
int poly_lincomb(bbg_ctx* ctx, const void* const* d_polys, const uint64_t* scalars, size_t count, const void* d_base, void* d_out, size_t n, hipStream_t st);
int quotient_widget(bbg_ctx* ctx, int widget, const void* const* d_polys, unsigned log2_large, const uint64_t* challenges, void* d_quotient,
                    uint64_t* alpha_out, hipStream_t st);
int msm_windows_for(int c);
int msm_width_slot(int c);
int msm_width_of_slot(int slot);
constexpr size_t DPV_CONSTS_BYTES = 4096; // poly.hip: one block of Z*_H division constants per (src, target, roots cut)
int g1_normalize_device(const void* d_jacs, size_t n, void* d_out, hipStream_t st);

static int make_srs(bbg_ctx* ctx, const void* d_plain_points, size_t n, bbg_srs** out)
{
    bbg_srs* s = new bbg_srs;
    s->ctx = ctx;
    s->s.n = n;
    s->s.device = ctx->device;
    if (n) {
        // window tables of the width a full-size MSM over this SRS will use; the other width is built on demand (msm_run)
        const int c = msm_pick_window(ctx, n);
        void* table = nullptr;
        hipError_t e = hipMalloc(&table, n * (size_t)msm_windows_for(c) * 64);
        if (e != hipSuccess) {
            delete s;
            return hip_fail(e, "hipMalloc(SRS window tables)", __FILE__, __LINE__);
        }
        int rc = srs_build_tables(d_plain_points, n, table, c, ctx->stream);
        if (rc == BBG_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = hip_fail(hipGetLastError(), "SRS table build", __FILE__, __LINE__);
        if (rc) {
            (void)hipFree(table);
            delete s;
            return rc;
        }
        s->s.points = table; // window 0 = the plain points
        s->s.tables[msm_width_slot(c)] = table;
        s->s.home_slot = msm_width_slot(c);
    }
    {
        std::lock_guard<std::mutex> lk(g_srs_mu);
        g_live_srs.insert(s);
    }
    *out = s;
    return BBG_OK;
}
} // namespace bbg

using namespace bbg;

#define CHECK_CTX(ctx)                                                                                               \
    do {                                                                                                             \
        if (!(ctx)) { set_error("null bbg_ctx"); return BBG_E_INVALID; }                                             \
        hipError_t _e = hipSetDevice((ctx)->device);                                                                 \
        if (_e != hipSuccess) return hip_fail(_e, "hipSetDevice", __FILE__, __LINE__);                               \
    } while (0)

extern "C" {

const char* bbg_last_error(void) { return g_last_error.c_str(); }

int bbg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bbg_init(int device, bbg_ctx** out)
{
    if (!out) { set_error("bbg_init: null out"); return BBG_E_INVALID; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error("bbg_init: no HIP device visible (this library has no CPU fallback)");
        return BBG_E_NODEVICE;
    }
    if (device < 0 || device >= n) { set_error("bbg_init: device index out of range"); return BBG_E_INVALID; }
    BBG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BBG_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("bbg_init: device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only");
        return BBG_E_NODEVICE;
    }
    bbg_ctx* c = new bbg_ctx;
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    c->own_stream = true;
    *out = c;
    return BBG_OK;
}

void bbg_destroy(bbg_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : ctx->domains) ntt_free_domain(kv.second);
    for (auto& kv : ctx->prof) {
        for (auto e : kv.second.start) (void)hipEventDestroy(e);
        for (auto e : kv.second.stop) (void)hipEventDestroy(e);
    }
    for (auto& kv : ctx->dpv_consts) (void)hipFree(kv.second);
    for (auto& kv : ctx->dpv_tables) (void)hipFree(kv.second);
    if (ctx->ntt_scratch) (void)hipFree(ctx->ntt_scratch);
    if (ctx->quot_setup) (void)hipFree(ctx->quot_setup);
    if (ctx->gp_totals) (void)hipFree(ctx->gp_totals);
    if (ctx->staging) (void)hipFree(ctx->staging);
    if (ctx->msm.buf) (void)hipFree(ctx->msm.buf);
    if (ctx->msm_tiny.buf) (void)hipFree(ctx->msm_tiny.buf);
    if (ctx->poly_scratch) (void)hipFree(ctx->poly_scratch);
    if (ctx->aux_stream) {
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) {
            (void)hipStreamDestroy(ctx->aux_streams[k]);
            (void)hipEventDestroy(ctx->ev_acc[k]);
            (void)hipEventDestroy(ctx->ev_done[k]);
        }
    }
    if (ctx->upload_stream) {
        (void)hipStreamDestroy(ctx->upload_stream);
        (void)hipEventDestroy(ctx->ev_upload_go);
        for (int k = 0; k < bbg_ctx::UPLOAD_PIECES; k++) (void)hipEventDestroy(ctx->ev_upload[k]);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int bbg_sync(bbg_ctx* ctx)
{
    CHECK_CTX(ctx);
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream)
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) BBG_HIP(hipStreamSynchronize(ctx->aux_streams[k]));
    return BBG_OK;
}

int bbg_join(bbg_ctx* ctx)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return msm_join(ctx, ctx->stream);
}

int bbg_join_lag(bbg_ctx* ctx, int lag)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (lag <= 0) return msm_join(ctx, ctx->stream);
    // wait for every outstanding reduction except the `lag` most recent ones (slots are used round robin)
    for (int back = lag; back < bbg_ctx::MSM_SLOTS; back++) {
        if (ctx->msm_seq < (unsigned long)back + 1) break;
        const int slot = (int)((ctx->msm_seq - 1 - (unsigned long)back) % bbg_ctx::MSM_SLOTS);
        if (ctx->ev_done_valid[slot]) BBG_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_done[slot], 0));
    }
    return BBG_OK;
}

int bbg_set_stream(bbg_ctx* ctx, void* hip_stream)
{
    CHECK_CTX(ctx);
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return BBG_OK;
}

int bbg_set_option(bbg_ctx* ctx, const char* key, long value)
{
    CHECK_CTX(ctx);
    if (!key) { set_error("bbg_set_option: null key"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!strcmp(key, "msm_async_reduce")) {
        BBG_HIP(hipDeviceSynchronize());
        ctx->msm_async_reduce = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_reduce_priority")) { // 1 = low-priority auxiliary stream (default), 0 = normal; takes effect when the stream is (re)created
        BBG_HIP(hipDeviceSynchronize());
        ctx->msm_reduce_low_priority = value != 0;
        if (ctx->aux_stream) {
            for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) {
                (void)hipStreamDestroy(ctx->aux_streams[k]);
                ctx->aux_streams[k] = nullptr;
                (void)hipEventDestroy(ctx->ev_acc[k]);
                (void)hipEventDestroy(ctx->ev_done[k]);
                ctx->ev_done_valid[k] = false;
            }
            ctx->aux_stream = nullptr;
        }
        return BBG_OK;
    }
    if (!strcmp(key, "msm_upload_pieces")) { // bbg_msm: pieces the host scalars travel in (1 = one copy in front of the MSM)
        if (value < 1 || value > bbg_ctx::UPLOAD_PIECES) { set_error("bbg_set_option: msm_upload_pieces must be 1..4"); return BBG_E_INVALID; }
        ctx->msm_upload_pieces = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_reduce_quad")) {
        BBG_HIP(hipDeviceSynchronize());
        ctx->msm_reduce_quad = (int)value & 15;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_acc_waves")) {
        BBG_HIP(hipDeviceSynchronize());
        ctx->msm_acc_waves = (int)value;
        ctx->msm_layout_n = 0;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_limbs29")) {
        ctx->msm_limbs29 = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_accumulate_quad")) {
        ctx->msm_accumulate_quad = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_msm_batch")) {
        if (value < 0 || value > BBG_MSM_BATCH_MAX) { set_error("prover_msm_batch must be 0 .. BBG_MSM_BATCH_MAX"); return BBG_E_INVALID; }
        ctx->prover_msm_batch = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_early_cosets")) {
        if (value < -1 || value > 1) { set_error("prover_early_cosets: -1 (automatic), 0 or 1"); return BBG_E_INVALID; }
        ctx->prover_early_cosets = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "quotient_setup_plan")) {
        if (value != 0 && value != 1) { set_error("quotient_setup_plan: 0 or 1"); return BBG_E_INVALID; }
        ctx->quotient_setup_plan = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "poly_limbs29")) {
        if (value != 0 && value != 1) { set_error("poly_limbs29: 0 or 1"); return BBG_E_INVALID; }
        ctx->poly_limbs29 = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_fused_divide")) {
        if (value != 0 && value != 1) { set_error("prover_fused_divide: 0 or 1"); return BBG_E_INVALID; }
        ctx->prover_fused_divide = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_tail_window")) {
        if (value != 0 && msm_width_slot((int)value) < 0) { set_error("prover_tail_window: 0 or a compiled window width"); return BBG_E_INVALID; }
        ctx->prover_tail_window = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_ntt_batch")) {
        ctx->prover_ntt_batch = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "prover_fail_round")) { // tests only: the next call of this prover round (1, 3, 4, 5, 6) fails once, as a device error would
        // fault injection is not an option of a production process: it exists only where the environment asked for test hooks when the
        // process started (BBG_TEST_HOOKS=1, read once) -- a stray option string cannot make a proof fail (round-5 advisor finding)
        static const bool hooks = [] { const char* e = getenv("BBG_TEST_HOOKS"); return e && e[0] == '1' && e[1] == 0; }();
        if (!hooks) { set_error("bbg_set_option: prover_fail_round is a test hook (start the process with BBG_TEST_HOOKS=1)"); return BBG_E_INVALID; }
        ctx->prover_fail_round = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "quotient_limbs29")) {
        ctx->quotient_limbs29 = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "quotient_fuse")) {
        ctx->quotient_fuse = value != 0;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_window")) {
        if (value != 0 && msm_width_slot((int)value) < 0) { set_error("msm_window must be 0 (automatic) or a compiled width: 13, 16, 17, 19, 20, 22"); return BBG_E_INVALID; }
        ctx->msm_window = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "msm_sort")) {
        if (value != 0 && value != 1) { set_error("msm_sort must be 0 or 1"); return BBG_E_INVALID; }
#ifndef BBG_ROCPRIM_SORT
        if (value == 0) { set_error("msm_sort = 0 (rocPRIM radix sort, A/B only) needs a library built with `make ROCPRIM_SORT=1`"); return BBG_E_INVALID; }
#endif
        BBG_HIP(hipDeviceSynchronize());
        ctx->msm_sort = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "ntt_limbs29")) { // a launch-time choice between kernels over the same plan and tables
        if (value < -1 || value > 1) { set_error("ntt_limbs29 must be -1 (automatic), 0 or 1"); return BBG_E_INVALID; }
        ctx->ntt_limbs29 = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "ntt_lds_planes")) { // a launch-time choice between two kernels over the same plan: no domain is rebuilt
        if (value < 0 || value > 2) { set_error("ntt_lds_planes must be 0 (automatic), 1 or 2"); return BBG_E_INVALID; }
        ctx->ntt_lds_planes = (int)value;
        return BBG_OK;
    }
    if (!strcmp(key, "ntt_tile_log")) {
        if (value < 9 || value > 12) { set_error("ntt_tile_log must be 9..12"); return BBG_E_INVALID; }
        ctx->ntt_tile_log = (int)value;
    } else if (!strcmp(key, "ntt_kernel")) {
        if (value != 1 && value != 2) { set_error("ntt_kernel must be 1 or 2"); return BBG_E_INVALID; }
        ctx->ntt_kernel = (int)value;
    } else if (!strcmp(key, "ntt_big_tile")) {
        if (value < 0 || value > 3) { set_error("ntt_big_tile must be 0 .. 3"); return BBG_E_INVALID; }
        ctx->ntt_big_tile = (int)value;
    } else if (!strcmp(key, "ntt_max_logr8")) {
        if (value < 6 || value > 11) { set_error("ntt_max_logr8 must be 6..11"); return BBG_E_INVALID; }
        ctx->ntt_max_logr8 = (int)value;
    } else if (!strcmp(key, "ntt_max_logr")) {
        if (value < 4 || value > 10) { set_error("ntt_max_logr must be 4..10"); return BBG_E_INVALID; }
        ctx->ntt_max_logr = (int)value;
    } else {
        set_error(std::string("bbg_set_option: unknown key ") + key);
        return BBG_E_INVALID;
    }
    // plans are per domain: drop cached domains so the new plan takes effect
    BBG_HIP(hipDeviceSynchronize());
    for (auto& kv : ctx->domains) ntt_free_domain(kv.second);
    ctx->domains.clear();
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ HBM budget
int bbg_memory_report(bbg_ctx* ctx, bbg_memory_info* out)
{
    CHECK_CTX(ctx);
    if (!out) { set_error("bbg_memory_report: null out"); return BBG_E_INVALID; }
    memset(out, 0, sizeof(*out));
    {
        std::lock_guard<std::mutex> lk(g_srs_mu);
        for (const bbg_srs* s : g_live_srs) {
            if (s->ctx != ctx) continue;
            out->live_srs++;
            std::lock_guard<std::mutex> lk2(const_cast<bbg_srs*>(s)->s.mu);
            for (int k = 0; k < Srs::MAX_WIDTHS; k++)
                if (s->s.tables[k]) out->srs_tables += s->s.n * (size_t)msm_windows_for(msm_width_of_slot(k)) * 64;
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (const auto& kv : ctx->domains) {
        out->ntt_tables += kv.second.bytes;
        out->ntt_domains++;
    }
    for (const auto& kv : ctx->dpv_tables) out->ntt_tables += (size_t)32 << ((kv.first >> 8) & 0xff); // poly_dpv_table: one Fr per target-domain point
    out->msm_arena = ctx->msm.bytes + ctx->msm_tiny.bytes;
    out->scratch = ctx->ntt_scratch_bytes + ctx->staging_bytes + ctx->poly_scratch_bytes + ctx->gp_totals_bytes + ctx->quot_setup_bytes +
                   ctx->dpv_consts.size() * (size_t)DPV_CONSTS_BYTES;
    prover_report(ctx, &out->prover_keys, &out->live_provers);
    out->total = out->srs_points + out->srs_tables + out->ntt_tables + out->msm_arena + out->scratch + out->prover_keys;
    BBG_HIP(hipMemGetInfo(&out->device_free, &out->device_total));
    return BBG_OK;
}

int bbg_memory_trim(bbg_ctx* ctx, int tables, size_t* released)
{
    CHECK_CTX(ctx);
    bbg_memory_info before, after;
    int rc = bbg_memory_report(ctx, &before);
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        BBG_HIP(hipDeviceSynchronize()); // nothing queued may still read what goes away
        for (auto& kv : ctx->domains) ntt_free_domain(kv.second);
        ctx->domains.clear();
        for (auto& kv : ctx->dpv_consts) (void)hipFree(kv.second);
        ctx->dpv_consts.clear();
        for (auto& kv : ctx->dpv_tables) (void)hipFree(kv.second);
        ctx->dpv_tables.clear();
        auto drop = [](void** buf, size_t* bytes) {
            if (*buf) (void)hipFree(*buf);
            *buf = nullptr;
            *bytes = 0;
        };
        drop(&ctx->ntt_scratch, &ctx->ntt_scratch_bytes);
        drop(&ctx->staging, &ctx->staging_bytes);
        drop(&ctx->poly_scratch, &ctx->poly_scratch_bytes);
        drop(&ctx->gp_totals, &ctx->gp_totals_bytes);
        drop(&ctx->quot_setup, &ctx->quot_setup_bytes);
        drop(&ctx->msm.buf, &ctx->msm.bytes);
        drop(&ctx->msm_tiny.buf, &ctx->msm_tiny.bytes);
        ctx->msm_tiny_layout = 0;
        ctx->msm_layout_n = 0; // the arena's counters are re-initialised with the next layout
        ctx->msm_zero_buf = nullptr;
        ctx->msm_zero_c = 0;
        ctx->msm_zero_sets = 0;
        for (int k = 0; k < bbg_ctx::MSM_SLOTS; k++) ctx->ev_done_valid[k] = false;
        if (tables) {
            std::lock_guard<std::mutex> lk2(g_srs_mu);
            for (bbg_srs* s : g_live_srs) {
                if (s->ctx != ctx) continue;
                // an MSM issued through ANOTHER context on this SRS holds Srs::mu while it picks a table and queues the kernels that read
                // it: with the lock, whatever was queued before is on the device, and the synchronisation below waits for it
                std::lock_guard<std::mutex> lk3(s->s.mu);
                BBG_HIP(hipDeviceSynchronize());
                for (int k = 0; k < Srs::MAX_WIDTHS; k++)
                    if (s->s.tables[k] && k != s->s.home_slot) { // the registration table holds the plain points: it stays
                        (void)hipFree(s->s.tables[k]);
                        s->s.tables[k] = nullptr;
                    }
            }
        }
    }
    rc = bbg_memory_report(ctx, &after);
    if (rc) return rc;
    if (released) *released = before.total > after.total ? before.total - after.total : 0; // another thread may have allocated in between
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ kernel timing
int bbg_profile_enable(bbg_ctx* ctx, int on)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->prof) {
        for (auto e : kv.second.start) (void)hipEventDestroy(e);
        for (auto e : kv.second.stop) (void)hipEventDestroy(e);
    }
    ctx->prof.clear();
    ctx->prof_on = on != 0;
    return BBG_OK;
}
int bbg_profile_get(bbg_ctx* ctx, const char* name, double* total_ms, size_t* launches)
{
    CHECK_CTX(ctx);
    if (!name || !total_ms || !launches) { set_error("bbg_profile_get: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    *total_ms = 0;
    *launches = 0;
    auto it = ctx->prof.find(name);
    if (it == ctx->prof.end()) return BBG_OK;
    for (size_t i = 0; i < it->second.start.size(); i++) {
        float ms = 0;
        BBG_HIP(hipEventElapsedTime(&ms, it->second.start[i], it->second.stop[i]));
        *total_ms += ms;
    }
    *launches = it->second.start.size();
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ memory helpers
int bbg_dev_alloc(bbg_ctx* ctx, size_t bytes, void** d_ptr)
{
    CHECK_CTX(ctx);
    if (!d_ptr) { set_error("bbg_dev_alloc: null out"); return BBG_E_INVALID; }
    BBG_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return BBG_OK;
}
int bbg_dev_free(bbg_ctx* ctx, void* d_ptr)
{
    CHECK_CTX(ctx);
    BBG_HIP(hipFree(d_ptr));
    return BBG_OK;
}
int bbg_dev_upload(bbg_ctx* ctx, void* d_dst, const void* src, size_t bytes)
{
    CHECK_CTX(ctx);
    BBG_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}
int bbg_dev_download(bbg_ctx* ctx, void* dst, const void* d_src, size_t bytes)
{
    CHECK_CTX(ctx);
    BBG_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ SRS
int bbg_srs_register(bbg_ctx* ctx, const uint64_t* points, size_t n, size_t stride_bytes, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!out || (!points && n)) { set_error("bbg_srs_register: null argument"); return BBG_E_INVALID; }
    if (stride_bytes != 64 && stride_bytes != 128) {
        set_error("bbg_srs_register: stride_bytes must be 64 (plain points) or 128 (interleaved endomorphism table)");
        return BBG_E_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    void* d_plain = nullptr;
    BBG_HIP(hipMalloc(&d_plain, n ? n * 64 : 64));
    hipError_t e;
    if (stride_bytes == 64)
        e = hipMemcpyAsync(d_plain, points, n * 64, hipMemcpyHostToDevice, ctx->stream);
    else
        e = hipMemcpy2DAsync(d_plain, 64, points, 128, 64, n, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_plain); return hip_fail(e, "SRS upload", __FILE__, __LINE__); }
    int rc = make_srs(ctx, d_plain, n, out);
    (void)hipFree(d_plain);
    return rc;
}

int bbg_srs_register_device(bbg_ctx* ctx, const void* d_points, size_t n, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!out || (!d_points && n)) { set_error("bbg_srs_register_device: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return make_srs(ctx, d_points, n, out);
}

int bbg_srs_synth_linear(bbg_ctx* ctx, uint64_t a, uint64_t s, size_t n, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!out) { set_error("bbg_srs_synth_linear: null out"); return BBG_E_INVALID; }
    if (s == 0 || a == 0) { set_error("bbg_srs_synth_linear: a and s must be non-zero"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    void* d_plain = nullptr;
    BBG_HIP(hipMalloc(&d_plain, n ? n * 64 : 64));
    int rc = srs_synth_linear(ctx, a, s, n, d_plain, ctx->stream);
    if (rc == BBG_OK) rc = make_srs(ctx, d_plain, n, out);
    (void)hipFree(d_plain);
    return rc;
}

int bbg_srs_synth_hashed(bbg_ctx* ctx, uint64_t seed, size_t n, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!out) { set_error("bbg_srs_synth_hashed: null out"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    void* d_plain = nullptr;
    BBG_HIP(hipMalloc(&d_plain, n ? n * 64 : 64));
    int rc = srs_synth_hashed(ctx, seed, n, d_plain, ctx->stream);
    if (rc == BBG_OK) rc = make_srs(ctx, d_plain, n, out);
    (void)hipFree(d_plain);
    return rc;
}

// pts: num_points x 8 limbs in STANDARD (non-Montgomery) form, slot 0 free: sets monomials[0] = G = (1, 2), converts to Montgomery form
// on the device and registers the SRS (the tail of read_transcript_g1 / Pippenger's constructors)
static int srs_from_plain_points(bbg_ctx* ctx, std::vector<uint64_t>& pts, size_t num_points, bbg_srs** out)
{
    for (int i = 0; i < 8; i++) pts[i] = 0;
    pts[0] = 1;
    pts[4] = 2;
    std::lock_guard<std::mutex> lk(ctx->mu);
    void* d_plain = nullptr;
    BBG_HIP(hipMalloc(&d_plain, num_points * 64));
    hipError_t e = hipMemcpyAsync(d_plain, pts.data(), num_points * 64, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_plain); return hip_fail(e, "transcript upload", __FILE__, __LINE__); }
    int rc = field_op_device(1, 5 /* to_montgomery */, d_plain, nullptr, d_plain, num_points * 2, ctx->stream);
    if (rc == BBG_OK) rc = make_srs(ctx, d_plain, num_points, out);
    (void)hipFree(d_plain);
    return rc;
}

// Ignition transcript reader: restates io::read_transcript_g1 (reference srs/io.cpp:134-162): manifest of seven
// big-endian u32 (:11-19,31-45), then num_g1_points x 64 B, every 8-byte limb big-endian, limbs least-significant
// first, values NOT in Montgomery form (:47-67).  monomials[0] = G, file points follow; files transcript00.dat,
// transcript01.dat, ... are consumed until num_points are read (:123-126).  Conversion to Montgomery form runs
// on the device.
int bbg_srs_load_transcript(bbg_ctx* ctx, const char* dir, size_t num_points, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!dir || !out || num_points == 0) { set_error("bbg_srs_load_transcript: bad argument"); return BBG_E_INVALID; }
    std::vector<uint64_t> pts(num_points * 8);
    size_t num_read = 1;
    for (size_t num = 0; num_read < num_points; num++) {
        char name[64];
        snprintf(name, sizeof(name), "/transcript%02zu.dat", num);
        std::string path = std::string(dir) + name;
        std::ifstream f(path, std::ifstream::binary);
        if (!f.good()) break;
        unsigned char m[28];
        f.read((char*)m, 28);
        if (f.gcount() != 28) { set_error("bbg_srs_load_transcript: short manifest in " + path); return BBG_E_INVALID; }
        auto be32 = [&](int k) { return ((uint32_t)m[4 * k] << 24) | ((uint32_t)m[4 * k + 1] << 16) | ((uint32_t)m[4 * k + 2] << 8) | m[4 * k + 3]; };
        const size_t num_g1 = be32(4);
        const size_t take = std::min(num_g1, num_points - num_read);
        f.read((char*)&pts[num_read * 8], (std::streamsize)(take * 64));
        if ((size_t)f.gcount() != take * 64) { set_error("bbg_srs_load_transcript: short read in " + path); return BBG_E_INVALID; }
        for (size_t i = num_read * 8; i < (num_read + take) * 8; i++) pts[i] = __builtin_bswap64(pts[i]);
        num_read += take;
    }
    if (num_read < num_points) {
        char buf[160];
        snprintf(buf, sizeof(buf), "Only read %zu points but require %zu. Is your srs large enough?", num_read, num_points);
        set_error(buf);
        return BBG_E_INVALID;
    }
    return srs_from_plain_points(ctx, pts, num_points, out);
}

// Pippenger(uint8_t const* points, size_t num_points) (pippenger.cpp:7-17; the C binding new_pippenger, c_bind.cpp:21-24): the points of a
// transcript already in memory -- (num_points - 1) x 64 bytes in the file encoding (io::read_g1_elements_from_buffer, srs/io.cpp:47-67);
// monomials[0] = G.
int bbg_srs_register_transcript_buffer(bbg_ctx* ctx, const uint8_t* points, size_t num_points, bbg_srs** out)
{
    CHECK_CTX(ctx);
    if (!out || num_points == 0 || (!points && num_points > 1)) { set_error("bbg_srs_register_transcript_buffer: bad argument"); return BBG_E_INVALID; }
    std::vector<uint64_t> pts(num_points * 8);
    if (num_points > 1) memcpy(&pts[8], points, (num_points - 1) * 64);
    for (size_t i = 8; i < num_points * 8; i++) pts[i] = __builtin_bswap64(pts[i]);
    return srs_from_plain_points(ctx, pts, num_points, out);
}

// BLAKE2b-512 (RFC 7693, unkeyed), the checksum an Ignition transcript carries after its points (srs/io.cpp:21-29 accounts for its
// 64 bytes; the reference reader skips it).  Host-side byte hashing, no field arithmetic.
namespace {
struct Blake2b {
    uint64_t h[8], t = 0;
    uint8_t buf[128];
    size_t fill = 0;
    static uint64_t rotr(uint64_t x, int r) { return (x >> r) | (x << (64 - r)); }
    Blake2b()
    {
        static const uint64_t IV[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010040ULL; // digest length 64, no key, fanout 1, depth 1
    }
    void compress(const uint8_t* block, bool last)
    {
        static const uint64_t IV[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                        0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
        static const uint8_t SIGMA[12][16] = {
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
            { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
            { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
            { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
            { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 } };
        uint64_t m[16], v[16];
        memcpy(m, block, 128); // little-endian host
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t* sg = SIGMA[r];
            G(0, 4, 8, 12, m[sg[0]], m[sg[1]]);   G(1, 5, 9, 13, m[sg[2]], m[sg[3]]);
            G(2, 6, 10, 14, m[sg[4]], m[sg[5]]);  G(3, 7, 11, 15, m[sg[6]], m[sg[7]]);
            G(0, 5, 10, 15, m[sg[8]], m[sg[9]]);  G(1, 6, 11, 12, m[sg[10]], m[sg[11]]);
            G(2, 7, 8, 13, m[sg[12]], m[sg[13]]); G(3, 4, 9, 14, m[sg[14]], m[sg[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const void* data, size_t len)
    {
        const uint8_t* p = (const uint8_t*)data;
        while (len) {
            if (fill == 128) { // only compress a full buffer when more input follows: the last block is flagged
                t += 128;
                compress(buf, false);
                fill = 0;
            }
            const size_t take = std::min(len, (size_t)128 - fill);
            memcpy(buf + fill, p, take);
            fill += take;
            p += take;
            len -= take;
        }
    }
    void final(uint8_t out[64])
    {
        t += fill;
        memset(buf + fill, 0, 128 - fill);
        compress(buf, true);
        memcpy(out, h, 64);
    }
};
} // namespace

// the checksum of a transcript file body (host only: usable without a GPU, e.g. to validate a downloaded transcript)
int bbg_transcript_checksum(const void* data, size_t len, uint8_t out[64])
{
    if ((!data && len) || !out) { set_error("bbg_transcript_checksum: null argument"); return BBG_E_INVALID; }
    Blake2b hash;
    hash.update(data, len);
    hash.final(out);
    return BBG_OK;
}

// Ignition transcript WRITER, the inverse of bbg_srs_load_transcript / io::read_transcript_g1 (srs/io.cpp:11-45 manifest, :47-67 point
// encoding, :123-162 file sequence): files dir/transcript00.dat, 01, ... hold points 1 .. n-1 of the SRS (point 0 is the generator,
// which every reader supplies itself, :137), points_per_file each (0 = all in one file).  Each point is x || y, every 8-byte limb
// big-endian, limbs least-significant first, values in standard (non-Montgomery) form -- the conversion runs on the device.
// g2_x_raw (may be NULL): 128 bytes appended to file 00 as its single G2 point, written as given.  The 64-byte BLAKE2b-512 checksum
// of everything before it closes each file.
int bbg_srs_write_transcript(bbg_srs* srs, const char* dir, size_t points_per_file, const uint8_t* g2_x_raw)
{
    if (!srs || !dir) { set_error("bbg_srs_write_transcript: null argument"); return BBG_E_INVALID; }
    CHECK_CTX(srs->ctx);
    const size_t n = srs->s.n;
    if (n < 2) { set_error("bbg_srs_write_transcript: nothing to write (an SRS of fewer than 2 points)"); return BBG_E_INVALID; }
    const size_t total = n - 1;
    if (points_per_file == 0 || points_per_file > total) points_per_file = total;
    const size_t files = (total + points_per_file - 1) / points_per_file;
    if (files > 100) { set_error("bbg_srs_write_transcript: more than 100 files (transcriptNN.dat)"); return BBG_E_INVALID; }
    std::vector<uint64_t> plain(total * 8);
    {
        std::lock_guard<std::mutex> lk(srs->ctx->mu);
        void* d_plain = nullptr;
        BBG_HIP(hipMalloc(&d_plain, total * 64));
        int rc = field_op_device(1, 4 /* from_montgomery */, (const char*)srs->s.points + 64, nullptr, d_plain, total * 2, srs->ctx->stream);
        hipError_t e = hipSuccess;
        if (rc == BBG_OK) e = hipMemcpyAsync(plain.data(), d_plain, total * 64, hipMemcpyDeviceToHost, srs->ctx->stream);
        if (rc == BBG_OK && e == hipSuccess) e = hipStreamSynchronize(srs->ctx->stream);
        (void)hipFree(d_plain);
        if (rc) return rc;
        if (e != hipSuccess) return hip_fail(e, "transcript download", __FILE__, __LINE__);
    }
    for (auto& limb : plain) limb = __builtin_bswap64(limb);
    auto be32 = [](uint8_t* out, uint32_t v) { out[0] = (uint8_t)(v >> 24); out[1] = (uint8_t)(v >> 16); out[2] = (uint8_t)(v >> 8); out[3] = (uint8_t)v; };
    for (size_t k = 0; k < files; k++) {
        const size_t first = k * points_per_file, count = std::min(points_per_file, total - first);
        const uint32_t num_g2 = (k == 0 && g2_x_raw) ? 1 : 0;
        uint8_t manifest[28];
        const uint32_t fields[7] = { (uint32_t)k, (uint32_t)files, (uint32_t)total, g2_x_raw ? 1u : 0u, (uint32_t)count, num_g2, (uint32_t)first };
        for (int i = 0; i < 7; i++) be32(manifest + 4 * i, fields[i]);
        Blake2b hash;
        hash.update(manifest, 28);
        hash.update(&plain[first * 8], count * 64);
        if (num_g2) hash.update(g2_x_raw, 128);
        uint8_t digest[64];
        hash.final(digest);
        char name[64];
        snprintf(name, sizeof(name), "/transcript%02zu.dat", k);
        const std::string path = std::string(dir) + name;
        std::ofstream f(path, std::ofstream::binary | std::ofstream::trunc);
        if (!f.good()) { set_error("bbg_srs_write_transcript: cannot open " + path); return BBG_E_INVALID; }
        f.write((const char*)manifest, 28);
        f.write((const char*)&plain[first * 8], (std::streamsize)(count * 64));
        if (num_g2) f.write((const char*)g2_x_raw, 128);
        f.write((const char*)digest, 64);
        if (!f.good()) { set_error("bbg_srs_write_transcript: short write to " + path); return BBG_E_INVALID; }
    }
    return BBG_OK;
}

size_t bbg_srs_num_points(const bbg_srs* srs) { return srs ? srs->s.n : 0; }

int bbg_srs_read(bbg_srs* srs, size_t from, size_t count, uint64_t* out_points)
{
    if (!srs || !out_points) { set_error("bbg_srs_read: null argument"); return BBG_E_INVALID; }
    CHECK_CTX(srs->ctx);
    if (from > srs->s.n || count > srs->s.n - from) { set_error("bbg_srs_read: range out of bounds"); return BBG_E_INVALID; }
    BBG_HIP(hipMemcpy(out_points, (const char*)srs->s.points + from * 64, count * 64, hipMemcpyDeviceToHost));
    return BBG_OK;
}

int bbg_srs_retain(bbg_srs* srs)
{
    if (!srs) { set_error("bbg_srs_retain: null handle"); return BBG_E_INVALID; }
    srs->refs.fetch_add(1);
    return BBG_OK;
}

void bbg_srs_free(bbg_srs* srs)
{
    if (!srs) return;
    if (srs->refs.fetch_sub(1) > 1) return; // another owner (a bbg_prover, a second cache entry) still uses it
    {
        std::lock_guard<std::mutex> lk(g_srs_mu);
        g_live_srs.erase(srs);
    }
    (void)hipSetDevice(srs->s.device); // the handle's own record: the context may already be gone (bbg_destroy before bbg_srs_free)
    (void)hipDeviceSynchronize();
    for (void* t : srs->s.tables)
        if (t) (void)hipFree(t);
    delete srs;
}

// ------------------------------------------------------------------------------------------------ MSM
int bbg_msm_device(bbg_ctx* ctx, bbg_srs* srs, const void* d_scalars, size_t from, size_t n, void* d_out_jacobian)
{
    CHECK_CTX(ctx);
    if (!srs || (!d_scalars && n) || !d_out_jacobian) { set_error("bbg_msm_device: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return msm_run(ctx, srs->s, d_scalars, from, n, d_out_jacobian, ctx->stream);
}

int bbg_msm_batch_device(bbg_ctx* ctx, bbg_srs* srs, size_t count, const void* const* d_scalars, const size_t* from, const size_t* n,
                         void* d_out_jacobians)
{
    CHECK_CTX(ctx);
    if (!srs || !d_scalars || !n || !d_out_jacobians) { set_error("bbg_msm_batch_device: null argument"); return BBG_E_INVALID; }
    if (count < 1 || count > BBG_MSM_BATCH_MAX) { set_error("bbg_msm_batch_device: 1 .. BBG_MSM_BATCH_MAX MSMs per batch"); return BBG_E_INVALID; }
    size_t zero[BBG_MSM_BATCH_MAX] = { 0 };
    std::lock_guard<std::mutex> lk(ctx->mu);
    return msm_run_batch(ctx, srs->s, (int)count, d_scalars, from ? from : zero, n, d_out_jacobians, ctx->stream);
}

int bbg_msm_batch(bbg_ctx* ctx, bbg_srs* srs, size_t count, const uint64_t* const* scalars, const size_t* from, const size_t* n, uint64_t* out_jacobians)
{
    CHECK_CTX(ctx);
    if (!srs || !scalars || !n || !out_jacobians) { set_error("bbg_msm_batch: null argument"); return BBG_E_INVALID; }
    if (count < 1 || count > BBG_MSM_BATCH_MAX) { set_error("bbg_msm_batch: 1 .. BBG_MSM_BATCH_MAX MSMs per batch"); return BBG_E_INVALID; }
    size_t zero[BBG_MSM_BATCH_MAX] = { 0 }, total = 0;
    for (size_t k = 0; k < count; k++) {
        if (n[k] && !scalars[k]) { set_error("bbg_msm_batch: null scalars"); return BBG_E_INVALID; }
        total += n[k];
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, total * 32 + 1024);
    if (rc) return rc;
    char* st = (char*)ctx->staging; // results (count x 96 B <= 768 B) | scalars of MSM 0 | MSM 1 | ...
    const void* d_ptrs[BBG_MSM_BATCH_MAX];
    size_t at = 1024;
    for (size_t k = 0; k < count; k++) {
        d_ptrs[k] = st + at;
        if (n[k]) BBG_HIP(hipMemcpyAsync(st + at, scalars[k], n[k] * 32, hipMemcpyHostToDevice, ctx->stream));
        at += n[k] * 32;
    }
    rc = msm_run_batch(ctx, srs->s, (int)count, d_ptrs, from ? from : zero, n, st, ctx->stream);
    if (rc) return rc;
    rc = msm_join(ctx, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out_jacobians, st, count * 96, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

int bbg_msm_plan(bbg_ctx* ctx, const bbg_srs* srs, size_t n, int* window_bits, int* windows)
{
    CHECK_CTX(ctx);
    if (!window_bits || !windows) { set_error("bbg_msm_plan: null out"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int c = 0;
    int rc = msm_plan(ctx, srs ? &srs->s : nullptr, n, &c);
    if (rc) return rc;
    *window_bits = c;
    *windows = msm_windows_for(c);
    return BBG_OK;
}

int bbg_msm(bbg_ctx* ctx, bbg_srs* srs, const uint64_t* scalars, size_t from, size_t n, uint64_t out_jacobian[12])
{
    CHECK_CTX(ctx);
    if (!srs || (!scalars && n) || !out_jacobian) { set_error("bbg_msm: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * 32 + 256);
    if (rc) return rc;
    char* st = (char*)ctx->staging;
    rc = msm_run(ctx, srs->s, st + 256, from, n, st, ctx->stream, scalars); // uploads the scalars itself, in pieces, under its first pass
    if (rc) return rc;
    rc = msm_join(ctx, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out_jacobian, st, 96, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

int bbg_g1_sum(bbg_ctx* ctx, const uint64_t* jacobians, size_t n, uint64_t out_jacobian[12])
{
    CHECK_CTX(ctx);
    if ((!jacobians && n) || !out_jacobian) { set_error("bbg_g1_sum: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * 96 + 256);
    if (rc) return rc;
    char* st = (char*)ctx->staging;
    if (n) BBG_HIP(hipMemcpyAsync(st + 256, jacobians, n * 96, hipMemcpyHostToDevice, ctx->stream));
    rc = g1_sum_device(ctx, st + 256, n, st, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out_jacobian, st, 96, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

int bbg_g1_sum_device(bbg_ctx* ctx, const void* d_jacobians, size_t n, void* d_out_jacobian)
{
    CHECK_CTX(ctx);
    if ((!d_jacobians && n) || !d_out_jacobian) { set_error("bbg_g1_sum_device: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return g1_sum_device(ctx, d_jacobians, n, d_out_jacobian, ctx->stream);
}

int bbg_g1_normalize(bbg_ctx* ctx, const uint64_t* jacobians, size_t n, uint64_t* out_affine)
{
    CHECK_CTX(ctx);
    if ((!jacobians || !out_affine) && n) { set_error("bbg_g1_normalize: null argument"); return BBG_E_INVALID; }
    if (n == 0) return BBG_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * 160);
    if (rc) return rc;
    char* st = (char*)ctx->staging;
    BBG_HIP(hipMemcpyAsync(st, jacobians, n * 96, hipMemcpyHostToDevice, ctx->stream));
    rc = g1_normalize_device(st, n, st + n * 96, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out_affine, st + n * 96, n * 64, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ NTT
int bbg_ntt_prepare(bbg_ctx* ctx, unsigned log2n)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_prepare(ctx, log2n);
}

int bbg_ntt_plan(bbg_ctx* ctx, unsigned log2n, int* passes, int log_radix[4], int* kernel, int* tile_log)
{
    CHECK_CTX(ctx);
    if (!passes || !log_radix || !kernel || !tile_log) { set_error("bbg_ntt_plan: null out"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_plan(ctx, log2n, passes, log_radix, kernel, tile_log);
}

int bbg_ntt_device(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_run(ctx, d_coeffs, log2n, op, generator_size, constant, ctx->stream);
}

int bbg_ntt(bbg_ctx* ctx, uint64_t* coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant)
{
    CHECK_CTX(ctx);
    if (!coeffs) { set_error("bbg_ntt: null coeffs"); return BBG_E_INVALID; }
    if (log2n > 28) { set_error("bbg_ntt: log2n > 28 exceeds the 2-adicity of BN254 Fr (fr.hpp:27-30)"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t bytes = ((size_t)1 << log2n) * 32;
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, bytes);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(ctx->staging, coeffs, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ntt_run(ctx, ctx->staging, log2n, op, generator_size, constant, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(coeffs, ctx->staging, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

int bbg_scale_powers_device(bbg_ctx* ctx, void* d_a, size_t count, const uint64_t* start, const uint64_t* base)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_scale_powers(ctx, d_a, count, start, base, ctx->stream);
}
int bbg_fr_root_pow(bbg_ctx* ctx, unsigned log2n, uint64_t e, int inverse, uint64_t out[4])
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_root_pow(ctx, log2n, e, inverse, out, ctx->stream);
}
int bbg_fr_pow(bbg_ctx* ctx, const uint64_t base[4], uint64_t e, uint64_t out[4])
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_fr_pow(ctx, base, e, out, ctx->stream);
}
int bbg_cross_dft_device(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2G, size_t len, unsigned log2n, int inverse)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_cross_dft(ctx, d_in, d_out, log2G, len, log2n, inverse, ctx->stream);
}

int bbg_coset_fft_split_device(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, size_t ext)
{
    CHECK_CTX(ctx);
    if (!d_coeffs) { set_error("bbg_coset_fft_split: null coeffs"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return ntt_coset_split(ctx, d_coeffs, log2n, ext, ctx->stream);
}

int bbg_poly_linear_combination_device(bbg_ctx* ctx, const void* const* d_polys, const uint64_t* scalars, size_t count, const void* d_base,
                                       void* d_out, size_t n)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_lincomb(ctx, d_polys, scalars, count, d_base, d_out, n, ctx->stream);
}

int bbg_permutation_grand_product_device(bbg_ctx* ctx, const void* const d_wires[4], const void* const d_sigmas[4], unsigned log2n,
                                         const uint64_t* challenges, void* d_z)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return permutation_grand_product(ctx, d_wires, d_sigmas, log2n, challenges, d_z, ctx->stream);
}

int bbg_quotient_widget_device(bbg_ctx* ctx, int widget, const void* const d_polys[BBG_QP_COUNT], unsigned log2_large_domain,
                               const uint64_t* challenges, void* d_quotient, uint64_t* alpha_base_out)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return quotient_widget(ctx, widget, d_polys, log2_large_domain, challenges, d_quotient, alpha_base_out, ctx->stream);
}

int bbg_coset_fft_extend(bbg_ctx* ctx, const uint64_t* coeffs, unsigned log2n, unsigned log2_domain, uint64_t* out)
{
    CHECK_CTX(ctx);
    if (!coeffs || !out) { set_error("bbg_coset_fft_extend: null argument"); return BBG_E_INVALID; }
    if (log2_domain > 28 || log2_domain < 2 || log2n > log2_domain) {
        set_error("bbg_coset_fft_extend: need log2n <= log2_domain and 2 <= log2_domain <= 28");
        return BBG_E_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t n = (size_t)1 << log2n, m = (size_t)1 << log2_domain;
    // n coefficients land behind the m-element result area; the transform reads them zero-extended and writes the result area
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, (m + n) * 32);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync((char*)ctx->staging + m * 32, coeffs, n * 32, hipMemcpyHostToDevice, ctx->stream));
    rc = ntt_coset_extend(ctx, (char*)ctx->staging + m * 32, n, ctx->staging, log2_domain, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out, ctx->staging, m * 32, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out + m * 4, out, 4 * 32); // add_lagrange_base_coefficient(out[0..3])
    return BBG_OK;
}

int bbg_coset_fft_split(bbg_ctx* ctx, uint64_t* coeffs, unsigned log2n, size_t ext)
{
    CHECK_CTX(ctx);
    if (!coeffs) { set_error("bbg_coset_fft_split: null coeffs"); return BBG_E_INVALID; }
    if (ext == 0 || log2n > 28 || ext > ((size_t)1 << 28)) { set_error("bbg_coset_fft_split: bad size"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t n = (size_t)1 << log2n;
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * ext * 32);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(ctx->staging, coeffs, n * 32, hipMemcpyHostToDevice, ctx->stream));
    rc = ntt_coset_split(ctx, ctx->staging, log2n, ext, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(coeffs, ctx->staging, n * ext * 32, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

// ------------------------------------------------------------------------------------------------ polynomial helpers
int bbg_poly_op_device(bbg_ctx* ctx, int op, const void* d_a, const void* d_b, void* d_r, size_t n)
{
    CHECK_CTX(ctx);
    if ((!d_a || !d_b || !d_r) && n) { set_error("bbg_poly_op_device: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_binop(op, d_a, d_b, d_r, n, ctx->stream);
}
int bbg_poly_evaluate_device(bbg_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4])
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_evaluate(ctx, d_coeffs, n, z, out, ctx->stream);
}
int bbg_kate_opening_device(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t z[4], uint64_t f_out[4])
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_kate_opening(ctx, d_src, d_dest, n, z, f_out, ctx->stream);
}
int bbg_poly_evaluate(bbg_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4])
{
    CHECK_CTX(ctx);
    if ((!coeffs && n) || !z || !out) { set_error("bbg_poly_evaluate: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * 32 + 32);
    if (rc) return rc;
    if (n) BBG_HIP(hipMemcpyAsync(ctx->staging, coeffs, n * 32, hipMemcpyHostToDevice, ctx->stream));
    return poly_evaluate(ctx, ctx->staging, n, z, out, ctx->stream);
}
int bbg_kate_opening(bbg_ctx* ctx, const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_out[4])
{
    CHECK_CTX(ctx);
    if ((n && (!src || !dest)) || !z || !f_out) { set_error("bbg_kate_opening: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, 2 * n * 32 + 64);
    if (rc) return rc;
    char* d_src = (char*)ctx->staging;
    char* d_dest = d_src + n * 32 + 32;
    if (n) BBG_HIP(hipMemcpyAsync(d_src, src, n * 32, hipMemcpyHostToDevice, ctx->stream));
    rc = poly_kate_opening(ctx, d_src, d_dest, n, z, f_out, ctx->stream);
    if (rc) return rc;
    if (n) BBG_HIP(hipMemcpyAsync(dest, d_dest, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}
int bbg_divide_by_pseudo_vanishing(bbg_ctx* ctx, uint64_t* evals, unsigned log2_src, unsigned log2_target, size_t num_roots_cut)
{
    CHECK_CTX(ctx);
    if (!evals) { set_error("bbg_divide_by_pseudo_vanishing: null argument"); return BBG_E_INVALID; }
    if (log2_target > 28) { set_error("bbg_divide_by_pseudo_vanishing: log2_target > 28"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t bytes = ((size_t)1 << log2_target) * 32;
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, bytes);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(ctx->staging, evals, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = poly_divide_pseudo_vanishing(ctx, ctx->staging, log2_src, log2_target, num_roots_cut, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(evals, ctx->staging, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}
int bbg_divide_by_pseudo_vanishing_device(bbg_ctx* ctx, void* d_evals, unsigned log2_src, unsigned log2_target, size_t num_roots_cut)
{
    CHECK_CTX(ctx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return poly_divide_pseudo_vanishing(ctx, d_evals, log2_src, log2_target, num_roots_cut, ctx->stream);
}

int bbg_field_op(bbg_ctx* ctx, int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n)
{
    CHECK_CTX(ctx);
    if (!a || !out) { set_error("bbg_field_op: null argument"); return BBG_E_INVALID; }
    if (n == 0) return BBG_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    int rc = ensure_buffer(&ctx->staging, &ctx->staging_bytes, n * 96);
    if (rc) return rc;
    char* st = (char*)ctx->staging;
    BBG_HIP(hipMemcpyAsync(st, a, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (b) BBG_HIP(hipMemcpyAsync(st + n * 32, b, n * 32, hipMemcpyHostToDevice, ctx->stream));
    rc = field_op_device(which, op, st, b ? st + n * 32 : nullptr, st + n * 64, n, ctx->stream);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(out, st + n * 64, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    BBG_HIP(hipStreamSynchronize(ctx->stream));
    return BBG_OK;
}

} // extern "C"
