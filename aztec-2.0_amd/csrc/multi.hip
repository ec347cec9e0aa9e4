// Multi-GPU inside the library (include/bbg.h, "bbg_multi_*"): one MSM or one (coset) NTT spread over several contexts of ONE
// process -- one context per GPU of the node (several contexts may share a device, which is how this is tested on a one-GPU box).
// SURVEY.md 8e:
//
//   MSM   point-range shards, the reference's own decomposition for large MSMs (Pippenger::pippenger_unsafe(scalars, from, range)
//         + g1_sum, pippenger.cpp:27-31, c_bind.cpp:31-46): context g holds SRS points [g*per, (g+1)*per) with their window tables
//         resident, runs the full bucket MSM over its slice of the scalars; the G 96-byte partials are collected and summed on
//         context 0.  No arithmetic exchange: the only traffic between GPUs is G x 96 bytes.
//   NTT   residue classes + ONE all-to-all (the reference's precedent is the 4-way coset split, polynomial_arithmetic.cpp:401-456):
//         context g transforms a_{g + G j} (size m = n / G), multiplies by w_n^(g q); chunk r of every context goes to context r
//         (peer copies over xGMI: (G-1)/G^2 of the data leaves each GPU); context r finishes with size-G DFTs across the chunks.
//
// Two exchange back ends, selected with bbg_multi_set_option("exchange", ...):
//   0  peer copies (hipMemcpyPeerAsync ordered with events, never with host synchronisation between the phases) -- the default
//   1  RCCL from C++ (north_star: "RCCL reduce/all-gather over xGMI"): one communicator per context (ncclCommInitAll over the group's
//      devices), ncclAllGather of the 96-byte partials + the group sum on every context's own GPU, and the all-to-all as grouped
//      ncclSend / ncclRecv; everything is enqueued on the contexts' streams, so stream order alone sequences compute and exchange.
//      librccl.so is loaded on first use (dlopen): a single-GPU user of libbbg.so neither links nor initialises RCCL.
// The process-per-GPU form of the same split (torch.distributed / RCCL, what bench.py --gpus N runs) is aztec-2.0_amd/parallel.py;
// all three call the same kernels.
#include "bbg_internal.h"

#include <dlfcn.h>

// The few RCCL declarations this file needs -- the stable public NCCL ABI (nccl.h: opaque communicator, result / data-type enums, the
// eight entry points below), declared here so that libbbg.so builds on a ROCm install without the RCCL development headers; the entry
// points themselves are resolved with dlsym on first use of the RCCL exchange.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;         // every other value is an error (ncclGetErrorString names it)
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t; // byte transport only
ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char* ncclGetErrorString(ncclResult_t result);
}

#include <cstring>
#include <thread>

using namespace bbg;

struct bbg_multi {
    int G = 0;
    std::vector<bbg_ctx*> ctx;
    std::vector<bbg_srs*> srs;
    std::vector<size_t> shard_from, shard_n;
    size_t srs_n = 0;
    // per context scratch
    std::vector<void*> d_scal;   // MSM: this context's slice of the scalars
    std::vector<size_t> d_scal_bytes;
    std::vector<void*> d_part;   // MSM: 96-byte partial
    std::vector<void*> d_x, d_recv, d_out; // NTT: shard, received chunks, cross-DFT output
    std::vector<size_t> ntt_bytes;
    std::vector<char*> h_stage;  // NTT host path: pinned residue-class staging
    std::vector<size_t> h_stage_bytes;
    std::vector<hipEvent_t> ev_sent, ev_done;
    std::vector<bool> ev_done_valid;
    char* h_parts = nullptr;     // pinned: G x 96 bytes
    // exchange back end 1: RCCL
    bool use_rccl = false;
    std::vector<ncclComm_t> comm;   // one per context (rank g = context g)
    std::vector<void*> d_gather;    // MSM: G x 96 bytes on every context
    void* d_sum = nullptr;          // MSM: the group sum on context 0
    std::mutex mu;
};

namespace {
// librccl.so, resolved once per process
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return BBG_OK;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { set_error(std::string("bbg_multi: cannot load librccl.so: ") + dlerror()); return BBG_E_INVALID; }
    RcclApi a;
    a.lib = lib;
#define BBG_RCCL_SYM(field, name)                                                                              \
    a.field = (decltype(a.field))dlsym(lib, name);                                                             \
    if (!a.field) { set_error("bbg_multi: librccl.so lacks " name); dlclose(lib); return BBG_E_INVALID; }
    BBG_RCCL_SYM(CommInitAll, "ncclCommInitAll")
    BBG_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    BBG_RCCL_SYM(AllGather, "ncclAllGather")
    BBG_RCCL_SYM(Send, "ncclSend")
    BBG_RCCL_SYM(Recv, "ncclRecv")
    BBG_RCCL_SYM(GroupStart, "ncclGroupStart")
    BBG_RCCL_SYM(GroupEnd, "ncclGroupEnd")
    BBG_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef BBG_RCCL_SYM
    g_rccl = a;
    return BBG_OK;
}
int rccl_fail(ncclResult_t r, const char* what)
{
    set_error(std::string("bbg_multi (RCCL): ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    return BBG_E_HIP;
}
#define BBG_RCCL(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) return rccl_fail(_r, #expr);                                                    \
    } while (0)
} // namespace

namespace {

// Every enqueue on a context's stream holds that context's mutex (a caller may drive bbg_multi_ctx(k) from another thread at the same
// time); a grouped RCCL operation enqueues on ALL streams of the group, so it takes all of them -- in index order, the only order used.
struct LockAllContexts {
    bbg_multi* m;
    explicit LockAllContexts(bbg_multi* mm) : m(mm)
    {
        for (int g = 0; g < m->G; g++) m->ctx[(size_t)g]->mu.lock();
    }
    ~LockAllContexts()
    {
        for (int g = m->G - 1; g >= 0; g--) m->ctx[(size_t)g]->mu.unlock();
    }
};

int set_dev(bbg_ctx* c)
{
    BBG_HIP(hipSetDevice(c->device));
    return BBG_OK;
}
// runs fn(g) for every context on its own host thread (uploads from pageable memory block their caller; one thread per GPU keeps
// every PCIe link busy); returns the first error with its message
template <typename F> int for_each_ctx(bbg_multi* m, F fn)
{
    std::vector<int> rc((size_t)m->G, BBG_OK);
    std::vector<std::string> msg((size_t)m->G);
    std::vector<std::thread> th;
    for (int g = 0; g < m->G; g++)
        th.emplace_back([&, g]() {
            rc[(size_t)g] = fn(g);
            if (rc[(size_t)g]) msg[(size_t)g] = bbg_last_error();
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < m->G; g++)
        if (rc[(size_t)g]) {
            set_error("context " + std::to_string(g) + ": " + msg[(size_t)g]);
            return rc[(size_t)g];
        }
    return BBG_OK;
}
int log2_exact(size_t v)
{
    int l = 0;
    while (((size_t)1 << l) < v) l++;
    return ((size_t)1 << l) == v ? l : -1;
}
void free_srs(bbg_multi* m)
{
    for (auto& s : m->srs) {
        if (s) bbg_srs_free(s);
        s = nullptr;
    }
    m->srs_n = 0;
}
int ensure_ntt_buffers(bbg_multi* m, size_t bytes)
{
    bool grow = false;
    for (int g = 0; g < m->G; g++) grow = grow || m->ntt_bytes[(size_t)g] < bytes;
    if (!grow) return BBG_OK;
    // bbg_multi_ntt_device is asynchronous: peer copies queued by OTHER devices' streams for a transform still in flight target this
    // context's d_recv, so every device of the group is drained before any buffer is released (a growth happens once per size)
    for (int g = 0; g < m->G; g++) {
        int rc = set_dev(m->ctx[(size_t)g]);
        if (rc) return rc;
        BBG_HIP(hipDeviceSynchronize());
    }
    for (int g = 0; g < m->G; g++) {
        if (m->ntt_bytes[(size_t)g] >= bytes) continue;
        int rc = set_dev(m->ctx[(size_t)g]);
        if (rc) return rc;
        for (void** b : { &m->d_x[(size_t)g], &m->d_recv[(size_t)g], &m->d_out[(size_t)g] }) {
            if (*b) BBG_HIP(hipFree(*b));
            *b = nullptr;
            BBG_HIP(hipMalloc(b, bytes));
        }
        m->ntt_bytes[(size_t)g] = bytes;
    }
    return BBG_OK;
}

// The device-resident transform: d_shards[g] = residue class g (a_{g + G j}, j < m) on context g, in place.
// After the call shard r holds A[t*m + r*len + q] at index t*len + q (t < G, q < len = m / G): G contiguous runs of the natural-order
// result, run t starting at natural index t*m + r*len.
int multi_ntt_core(bbg_multi* m, void* const* d_shards, unsigned log2n, int op)
{
    const int G = m->G, log2G = log2_exact((size_t)G);
    if (log2G < 0 || log2G > 3) { set_error("bbg_multi_ntt: the number of contexts must be 1, 2, 4 or 8"); return BBG_E_INVALID; }
    if (op < BBG_FFT || op > BBG_COSET_IFFT) { set_error("bbg_multi_ntt: op must be fft, ifft, coset_fft or coset_ifft"); return BBG_E_INVALID; }
    if (log2n > 28 || log2n < (unsigned)(2 * log2G)) { set_error("bbg_multi_ntt: need 2 log2(G) <= log2n <= 28"); return BBG_E_INVALID; }
    const bool inverse = (op == BBG_IFFT || op == BBG_COSET_IFFT), coset = (op == BBG_COSET_FFT || op == BBG_COSET_IFFT);
    const unsigned log2m = log2n - (unsigned)log2G;
    const size_t mm = (size_t)1 << log2m, len = mm >> log2G;
    if (G == 1 && !m->use_rccl) { // one context: the plain transform (with RCCL selected even a group of one goes through the exchange)
        int rc = set_dev(m->ctx[0]);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(m->ctx[0]->mu);
        return ntt_run(m->ctx[0], d_shards[0], log2n, op, 0, nullptr, m->ctx[0]->stream);
    }
    int rc = ensure_ntt_buffers(m, mm * 32);
    if (rc) return rc;
    // phase 1 (every context, asynchronous): coset factors, local transform, twiddles, peer copies of the chunks
    for (int g = 0; g < G; g++) {
        bbg_ctx* c = m->ctx[(size_t)g];
        rc = set_dev(c);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(c->mu);
        hipStream_t st = c->stream;
        void* x = d_shards[g];
        // the receive buffers of the previous transform must have been consumed before anything is sent again
        for (int r = 0; r < G && !m->use_rccl; r++)
            if (m->ev_done_valid[(size_t)r]) BBG_HIP(hipStreamWaitEvent(st, m->ev_done[(size_t)r], 0));
        if (coset && !inverse) rc = ntt_scale_geometric(c, x, mm, log2n, 0 /* g */, (uint64_t)G, (uint64_t)g, -1, st); // a_{g+Gj} *= gen^(g+Gj)
        if (!rc) rc = ntt_run(c, x, log2m, inverse ? BBG_IFFT : BBG_FFT, 0, nullptr, st);
        // x[q] *= w_n^(+-g q); the inverse also owes the factor 1/G (the local ifft divided by m only)
        if (!rc && (g != 0 || inverse)) rc = ntt_scale_geometric(c, x, mm, log2n, inverse ? 3 : 2, (uint64_t)g, 0, inverse ? log2G : -1, st);
        if (rc) return rc;
        if (!m->use_rccl) {
            for (int r = 0; r < G; r++) // chunk r -> context r, slot g
                BBG_HIP(hipMemcpyPeerAsync((char*)m->d_recv[(size_t)r] + (size_t)g * len * 32, m->ctx[(size_t)r]->device, (const char*)x + (size_t)r * len * 32,
                                           c->device, len * 32, st));
            BBG_HIP(hipEventRecord(m->ev_sent[(size_t)g], st));
        }
    }
    if (m->use_rccl) {
        // the all-to-all as ONE group of point-to-point operations (xGMI is point-to-point: every pair has its own links): rank g sends
        // chunk r of its shard to rank r and receives rank r's chunk g into slot r.  Enqueued on the contexts' streams: behind the
        // local transform, ahead of the cross DFT -- and behind the previous transform's cross DFT, which read the same receive buffer.
        LockAllContexts all(m);
        BBG_RCCL(g_rccl.GroupStart());
        for (int g = 0; g < G; g++) {
            hipStream_t st = m->ctx[(size_t)g]->stream;
            for (int r = 0; r < G; r++) {
                BBG_RCCL(g_rccl.Send((const char*)d_shards[g] + (size_t)r * len * 32, len * 32, ncclUint8, r, m->comm[(size_t)g], st));
                BBG_RCCL(g_rccl.Recv((char*)m->d_recv[(size_t)g] + (size_t)r * len * 32, len * 32, ncclUint8, r, m->comm[(size_t)g], st));
            }
        }
        BBG_RCCL(g_rccl.GroupEnd());
    }
    // phase 2: size-G DFT across the received chunks, the coset_ifft post-scaling, result back into the shard
    for (int r = 0; r < G; r++) {
        bbg_ctx* c = m->ctx[(size_t)r];
        rc = set_dev(c);
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(c->mu);
        hipStream_t st = c->stream;
        if (!m->use_rccl)
            for (int g = 0; g < G; g++) BBG_HIP(hipStreamWaitEvent(st, m->ev_sent[(size_t)g], 0));
        rc = ntt_cross_dft(c, m->d_recv[(size_t)r], m->d_out[(size_t)r], (unsigned)log2G, len, log2n, inverse ? 1 : 0, st);
        if (rc) return rc;
        BBG_HIP(hipEventRecord(m->ev_done[(size_t)r], st));
        m->ev_done_valid[(size_t)r] = true;
        if (coset && inverse) // coset_ifft: a_j *= g^-j over the natural index j = t*m + r*len + q (polynomial_arithmetic.cpp:480-484)
            for (int t = 0; t < G && !rc; t++)
                rc = ntt_scale_geometric(c, (char*)m->d_out[(size_t)r] + (size_t)t * len * 32, len, log2n, 1 /* g^-1 */, 1, (uint64_t)t * mm + (uint64_t)r * len, -1, st);
        if (rc) return rc;
        BBG_HIP(hipMemcpyAsync(d_shards[r], m->d_out[(size_t)r], mm * 32, hipMemcpyDeviceToDevice, st));
    }
    return BBG_OK;
}

void multi_release_rccl(bbg_multi* m)
{
    for (size_t g = 0; g < m->comm.size(); g++)
        if (m->comm[g] && g_rccl.CommDestroy) {
            if (m->ctx[g]) {
                (void)hipSetDevice(m->ctx[g]->device);
                (void)hipDeviceSynchronize();
            }
            (void)g_rccl.CommDestroy(m->comm[g]);
        }
    m->comm.clear();
    for (size_t g = 0; g < m->d_gather.size(); g++)
        if (m->d_gather[g] && m->ctx[g]) {
            (void)hipSetDevice(m->ctx[g]->device);
            (void)hipFree(m->d_gather[g]);
        }
    m->d_gather.clear();
    if (m->d_sum && m->ctx[0]) {
        (void)hipSetDevice(m->ctx[0]->device);
        (void)hipFree(m->d_sum);
    }
    m->d_sum = nullptr;
    m->use_rccl = false;
}
// communicators over the group's devices (rank g = context g) and the all-gather buffers
int multi_init_rccl(bbg_multi* m)
{
    if (!m->comm.empty()) return BBG_OK;
    int rc = rccl_load();
    if (rc) return rc;
    std::vector<int> devices((size_t)m->G);
    for (int g = 0; g < m->G; g++) devices[(size_t)g] = m->ctx[(size_t)g]->device;
    for (int a = 0; a < m->G; a++)
        for (int b = a + 1; b < m->G; b++)
            if (devices[(size_t)a] == devices[(size_t)b]) {
                set_error("bbg_multi_set_option(exchange = 1): RCCL needs one DISTINCT device per context (a device appears twice in this group)");
                return BBG_E_INVALID;
            }
    m->comm.assign((size_t)m->G, nullptr);
    ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), m->G, devices.data());
    if (r != ncclSuccess) {
        m->comm.clear();
        return rccl_fail(r, "ncclCommInitAll");
    }
    m->d_gather.assign((size_t)m->G, nullptr);
    for (int g = 0; g < m->G; g++) {
        rc = set_dev(m->ctx[(size_t)g]);
        if (rc) return rc;
        BBG_HIP(hipMalloc(&m->d_gather[(size_t)g], (size_t)m->G * 96));
    }
    rc = set_dev(m->ctx[0]);
    if (rc) return rc;
    BBG_HIP(hipMalloc(&m->d_sum, 96));
    return BBG_OK;
}

} // namespace

extern "C" {

int bbg_multi_set_option(bbg_multi* m, const char* key, long value)
{
    if (!m || !key) { set_error("bbg_multi_set_option: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    if (!strcmp(key, "exchange")) {
        if (value != 0 && value != 1) { set_error("bbg_multi_set_option: exchange must be 0 (peer copies) or 1 (RCCL)"); return BBG_E_INVALID; }
        int rc = bbg_multi_sync(m); // pending work of the other back end first
        if (rc) return rc;
        if (value == 1) {
            rc = multi_init_rccl(m);
            if (rc) {
                multi_release_rccl(m);
                return rc;
            }
        }
        m->use_rccl = value == 1;
        return BBG_OK;
    }
    set_error("bbg_multi_set_option: unknown key");
    return BBG_E_INVALID;
}

int bbg_multi_create(const int* devices, int count, bbg_multi** out)
{
    if (!devices || !out || count < 1 || count > 64) { set_error("bbg_multi_create: bad argument"); return BBG_E_INVALID; }
    bbg_multi* m = new bbg_multi;
    m->G = count;
    const size_t G = (size_t)count;
    m->ctx.assign(G, nullptr);
    m->srs.assign(G, nullptr);
    m->shard_from.assign(G, 0);
    m->shard_n.assign(G, 0);
    m->d_scal.assign(G, nullptr);
    m->d_scal_bytes.assign(G, 0);
    m->d_part.assign(G, nullptr);
    m->d_x.assign(G, nullptr);
    m->d_recv.assign(G, nullptr);
    m->d_out.assign(G, nullptr);
    m->ntt_bytes.assign(G, 0);
    m->h_stage.assign(G, nullptr);
    m->h_stage_bytes.assign(G, 0);
    m->ev_sent.assign(G, nullptr);
    m->ev_done.assign(G, nullptr);
    m->ev_done_valid.assign(G, false);
    int rc = BBG_OK;
    for (int g = 0; g < count && !rc; g++) {
        rc = bbg_init(devices[g], &m->ctx[(size_t)g]);
        hipError_t e = hipSuccess;
        if (!rc) e = hipMalloc(&m->d_part[(size_t)g], 96);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_sent[(size_t)g], hipEventDisableTiming);
        if (!rc && e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_done[(size_t)g], hipEventDisableTiming);
        if (!rc && e != hipSuccess) rc = hip_fail(e, "bbg_multi_create", __FILE__, __LINE__);
    }
    // direct peer access between distinct devices (xGMI); failure is not fatal: peer copies then stage through the host
    for (int a = 0; a < count && !rc; a++)
        for (int b = 0; b < count; b++)
            if (devices[a] != devices[b]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[a]);
                    (void)hipDeviceEnablePeerAccess(devices[b], 0);
                    (void)hipGetLastError(); // "already enabled" is fine
                }
            }
    if (!rc && hipHostMalloc((void**)&m->h_parts, G * 96, hipHostMallocDefault) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipHostMalloc", __FILE__, __LINE__);
    if (rc) {
        bbg_multi_destroy(m);
        return rc;
    }
    *out = m;
    return BBG_OK;
}

void bbg_multi_destroy(bbg_multi* m)
{
    if (!m) return;
    free_srs(m);
    multi_release_rccl(m);
    for (int g = 0; g < m->G; g++) {
        bbg_ctx* c = m->ctx[(size_t)g];
        if (!c) continue;
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        for (void* b : { m->d_scal[(size_t)g], m->d_part[(size_t)g], m->d_x[(size_t)g], m->d_recv[(size_t)g], m->d_out[(size_t)g] })
            if (b) (void)hipFree(b);
        if (m->h_stage[(size_t)g]) (void)hipHostFree(m->h_stage[(size_t)g]);
        if (m->ev_sent[(size_t)g]) (void)hipEventDestroy(m->ev_sent[(size_t)g]);
        if (m->ev_done[(size_t)g]) (void)hipEventDestroy(m->ev_done[(size_t)g]);
        bbg_destroy(c);
    }
    if (m->h_parts) (void)hipHostFree(m->h_parts);
    delete m;
}

int bbg_multi_count(const bbg_multi* m) { return m ? m->G : 0; }
bbg_ctx* bbg_multi_ctx(bbg_multi* m, int k) { return (m && k >= 0 && k < m->G) ? m->ctx[(size_t)k] : nullptr; }

int bbg_multi_srs_register(bbg_multi* m, const uint64_t* points, size_t n, size_t stride_bytes)
{
    if (!m || (!points && n)) { set_error("bbg_multi_srs_register: null argument"); return BBG_E_INVALID; }
    if (stride_bytes != 64 && stride_bytes != 128) { set_error("bbg_multi_srs_register: stride_bytes must be 64 or 128"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    free_srs(m);
    const size_t per = (n + (size_t)m->G - 1) / (size_t)m->G;
    int rc = for_each_ctx(m, [&](int g) {
        const size_t from = std::min(n, (size_t)g * per), cnt = std::min(per, n - from);
        m->shard_from[(size_t)g] = from;
        m->shard_n[(size_t)g] = cnt;
        return bbg_srs_register(m->ctx[(size_t)g], (const uint64_t*)((const char*)points + from * stride_bytes), cnt, stride_bytes, &m->srs[(size_t)g]);
    });
    if (rc) {
        free_srs(m);
        return rc;
    }
    m->srs_n = n;
    return BBG_OK;
}

int bbg_multi_srs_synth_hashed(bbg_multi* m, uint64_t seed, size_t n)
{
    if (!m) { set_error("bbg_multi_srs_synth_hashed: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    free_srs(m);
    const size_t per = (n + (size_t)m->G - 1) / (size_t)m->G;
    int rc = for_each_ctx(m, [&](int g) {
        const size_t from = std::min(n, (size_t)g * per), cnt = std::min(per, n - from);
        m->shard_from[(size_t)g] = from;
        m->shard_n[(size_t)g] = cnt;
        return bbg_srs_synth_hashed(m->ctx[(size_t)g], seed + from, cnt, &m->srs[(size_t)g]); // P_i = mix64(seed + i) G: a shard is the same generator offset
    });
    if (rc) {
        free_srs(m);
        return rc;
    }
    m->srs_n = n;
    return BBG_OK;
}

size_t bbg_multi_srs_num_points(const bbg_multi* m) { return m ? m->srs_n : 0; }

int bbg_multi_msm(bbg_multi* m, const uint64_t* scalars, size_t from, size_t n, uint64_t out_jacobian[12])
{
    if (!m || (!scalars && n) || !out_jacobian) { set_error("bbg_multi_msm: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    if (from > m->srs_n || n > m->srs_n - from) { set_error("bbg_multi_msm: range [from, from+n) exceeds the registered SRS"); return BBG_E_INVALID; }
    int rc = for_each_ctx(m, [&](int g) {
        bbg_ctx* c = m->ctx[(size_t)g];
        int r = set_dev(c);
        if (r) return r;
        const size_t sf = m->shard_from[(size_t)g], sn = m->shard_n[(size_t)g];
        const size_t lo = std::max(from, sf), hi = std::min(from + n, sf + sn);
        const size_t cnt = hi > lo ? hi - lo : 0;
        std::lock_guard<std::mutex> lkc(c->mu);
        hipStream_t st = c->stream;
        if (cnt) {
            r = ensure_buffer(&m->d_scal[(size_t)g], &m->d_scal_bytes[(size_t)g], cnt * 32);
            if (r) return r;
            BBG_HIP(hipMemcpyAsync(m->d_scal[(size_t)g], scalars + (lo - from) * 4, cnt * 32, hipMemcpyHostToDevice, st));
        }
        r = msm_run(c, m->srs[(size_t)g]->s, m->d_scal[(size_t)g], cnt ? lo - sf : 0, cnt, m->d_part[(size_t)g], st); // cnt == 0 -> infinity
        if (!r) r = msm_join(c, st);
        if (r) return r;
        if (m->use_rccl) return (int)BBG_OK; // the partial stays on the device: the all-gather below is ordered behind it on this stream
        BBG_HIP(hipMemcpyAsync(m->h_parts + (size_t)g * 96, m->d_part[(size_t)g], 96, hipMemcpyDeviceToHost, st));
        BBG_HIP(hipStreamSynchronize(st));
        return (int)BBG_OK;
    });
    if (rc) return rc;
    if (m->use_rccl) {
        // "reduce" = all-gather + local group sum (RCCL has no elliptic-curve reduction operator): every context receives the G partials
        {
            LockAllContexts all(m);
            BBG_RCCL(g_rccl.GroupStart());
            for (int g = 0; g < m->G; g++)
                BBG_RCCL(g_rccl.AllGather(m->d_part[(size_t)g], m->d_gather[(size_t)g], 96, ncclUint8, m->comm[(size_t)g], m->ctx[(size_t)g]->stream));
            BBG_RCCL(g_rccl.GroupEnd());
        }
        bbg_ctx* c0 = m->ctx[0];
        rc = set_dev(c0);
        if (rc) return rc;
        {
            std::lock_guard<std::mutex> lkc(c0->mu);
            rc = g1_sum_device(c0, m->d_gather[0], (size_t)m->G, m->d_sum, c0->stream); // g1_sum of the partials (c_bind.cpp:39-46)
            if (rc) return rc;
            BBG_HIP(hipMemcpyAsync(m->h_parts, m->d_sum, 96, hipMemcpyDeviceToHost, c0->stream));
        }
        rc = bbg_multi_sync(m); // every context's all-gather has completed: the partial buffers may be reused
        if (rc) return rc;
        memcpy(out_jacobian, m->h_parts, 96);
        return BBG_OK;
    }
    return bbg_g1_sum(m->ctx[0], (const uint64_t*)m->h_parts, (size_t)m->G, out_jacobian); // g1_sum of the partials (c_bind.cpp:39-46)
}

int bbg_multi_ntt_device(bbg_multi* m, void* const* d_shards, unsigned log2n, int op)
{
    if (!m || !d_shards) { set_error("bbg_multi_ntt_device: null argument"); return BBG_E_INVALID; }
    for (int g = 0; g < m->G; g++)
        if (!d_shards[g]) { set_error("bbg_multi_ntt_device: null shard"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    return multi_ntt_core(m, d_shards, log2n, op);
}

int bbg_multi_sync(bbg_multi* m)
{
    if (!m) { set_error("bbg_multi_sync: null argument"); return BBG_E_INVALID; }
    for (int g = 0; g < m->G; g++) {
        int rc = bbg_sync(m->ctx[(size_t)g]);
        if (rc) return rc;
    }
    return BBG_OK;
}

int bbg_multi_ntt(bbg_multi* m, uint64_t* coeffs, unsigned log2n, int op)
{
    if (!m || !coeffs) { set_error("bbg_multi_ntt: null argument"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(m->mu);
    const int G = m->G, log2G = log2_exact((size_t)G);
    if (log2G < 0 || log2G > 3 || log2n > 28 || log2n < (unsigned)(2 * log2G)) { set_error("bbg_multi_ntt: need 1, 2, 4 or 8 contexts and 2 log2(G) <= log2n <= 28"); return BBG_E_INVALID; }
    const size_t n = (size_t)1 << log2n, mm = n >> log2G, len = mm >> log2G;
    int rc = ensure_ntt_buffers(m, mm * 32);
    if (rc) return rc;
    // residue class g, gathered on the host into pinned staging by context g's thread, then one upload per context
    rc = for_each_ctx(m, [&](int g) {
        bbg_ctx* c = m->ctx[(size_t)g];
        int r = set_dev(c);
        if (r) return r;
        if (m->h_stage_bytes[(size_t)g] < mm * 32) {
            if (m->h_stage[(size_t)g]) BBG_HIP(hipHostFree(m->h_stage[(size_t)g]));
            m->h_stage[(size_t)g] = nullptr;
            BBG_HIP(hipHostMalloc((void**)&m->h_stage[(size_t)g], mm * 32, hipHostMallocDefault));
            m->h_stage_bytes[(size_t)g] = mm * 32;
        }
        uint64_t* stage = (uint64_t*)m->h_stage[(size_t)g];
        for (size_t j = 0; j < mm; j++) memcpy(stage + 4 * j, coeffs + 4 * ((size_t)g + (size_t)G * j), 32);
        BBG_HIP(hipMemcpyAsync(m->d_x[(size_t)g], stage, mm * 32, hipMemcpyHostToDevice, c->stream));
        return (int)BBG_OK;
    });
    if (rc) return rc;
    rc = multi_ntt_core(m, m->d_x.data(), log2n, op);
    if (rc) return rc;
    // G contiguous runs of the natural-order result per context
    return for_each_ctx(m, [&](int r_) {
        bbg_ctx* c = m->ctx[(size_t)r_];
        int r = set_dev(c);
        if (r) return r;
        if (G == 1) {
            BBG_HIP(hipMemcpyAsync(coeffs, m->d_x[0], n * 32, hipMemcpyDeviceToHost, c->stream));
        } else {
            for (int t = 0; t < G; t++)
                BBG_HIP(hipMemcpyAsync(coeffs + 4 * ((size_t)t * mm + (size_t)r_ * len), (const char*)m->d_x[(size_t)r_] + (size_t)t * len * 32, len * 32,
                                       hipMemcpyDeviceToHost, c->stream));
        }
        BBG_HIP(hipStreamSynchronize(c->stream));
        return (int)BBG_OK;
    });
}

} // extern "C"
