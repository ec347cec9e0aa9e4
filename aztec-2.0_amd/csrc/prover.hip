// Resident PLONK prover rounds (include/bbg.h, "bbg_prover_*"): everything O(n) a TurboPLONK / StandardPLONK proof needs between
// the witness and the commitments stays in HBM -- the native form of SURVEY.md 8f-1 (the work_queue replacement) with 8f-2 / 8f-4
// inside it.  The host keeps what is O(1): the transcript (Fiat-Shamir hashing) and the challenge algebra; this file sequences
// the kernels of ntt.hip / msm.hip / quotient.hip / poly.hip per round on the context stream:
//
//   round 1   wires (Lagrange, blinded) up -> ifft -> W_i = MSM                         prover.cpp:139-190 + work_queue IFFT / MSM items
//   round 3   z = grand product, blinding rows, ifft, Z = MSM, coset FFTs (4n) of w_i, z  permutation_widget_impl.hpp:48-312, prover.cpp:239-268
//   round 4   quotient widgets, / Z*_H, coset iFFT (4n), T_i = MSM                        prover.cpp:275-363, :117-137
//   round 5   evaluations at zeta / zeta w, r(X) = sum c_k P_k, r(zeta)                   prover.cpp:388-410, kate_commitment_scheme.cpp:362-420
//   round 6   F = t_low + sum nu_k P_k, F' = sum nu'_k P'_k, Kate quotients, PI_Z, PI_Z_OMEGA = MSM    kate_commitment_scheme.cpp:133-236
//
// Per proving key: the selector / permutation polynomials are registered ONCE in coefficient form under an explicit handle
// (bbg_prover_set_key_poly); their Lagrange (sigma) and 4n-coset forms and L_1 on the coset are derived on the device.
// Per proof: 4 x n wire values go up, 11 commitments (64 B each) and ~25 field elements come down.  The reduce phase of every MSM
// runs on the context's auxiliary stream and overlaps the next kernels; each round ends with ONE host synchronisation.
#include "bbg_internal.h"

#include <algorithm>
#include <cstring>
#include <set>
#include "field.hip.h"
#include "ntt_consts.hip.h"

namespace bbg {
int permutation_grand_product_begin(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n,
                                    const uint64_t* challenges, void* d_z, hipStream_t st, hipStream_t inv_stream, hipEvent_t ev_ready, hipEvent_t ev_inverted);
int permutation_grand_product_finish(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n, void* d_z,
                                     hipStream_t st, hipEvent_t ev_inverted);
int quotient_widgets_chain(bbg_ctx* ctx, const int* widgets, int count, const void* const* d_polys, unsigned log2_large,
                           const uint64_t* challenges, void* d_quotient, uint64_t* alpha_out, hipStream_t st);
int poly_multi_evaluate(bbg_ctx* ctx, const void* const* d_polys, const size_t* lens, const int* shifted, size_t count, unsigned log2n,
                        const uint64_t* zeta, void** d_results, hipStream_t st);
int poly_kate_opening_async(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t* z, void* d_f, hipStream_t st);
int poly_lincomb(bbg_ctx* ctx, const void* const* d_polys, const uint64_t* scalars, size_t count, const void* d_base, void* d_out, size_t n, hipStream_t st);

// out[j] = *c, j < count
__global__ void k_fill_const(Fr* out, const Fr* c, size_t count)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) fe_store<FrP>(out + j, *c);
}
// *out = *a * s  (the (3n+1)-th quotient coefficient of StandardPLONK entering the opening polynomial, kate_commitment_scheme.cpp:196-205)
__global__ void k_mul_one(Fr* out, const Fr* a, Fr s)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) fe_store<FrP>(out, fe_mul(fe_load<FrP>(a), s));
}
} // namespace bbg

using namespace bbg;

namespace {
constexpr int MAX_RESULTS = 16;
constexpr size_t PIN_AFFINE = 0, PIN_EVAL = 2048, PIN_BLIND = 3072, PIN_BYTES = 4096;
}

struct bbg_prover {
    bbg_ctx* ctx = nullptr;
    bbg_srs* srs = nullptr; // retained (bbg_srs_retain) for the lifetime of the handle
    int device = 0;
    unsigned log2n = 0;
    int width = 4;
    int flavour = BBG_FLAVOUR_TURBO;
    size_t n = 0;
    uint64_t gens[16] = { 0 };   // g, k1, k2, k3 (Montgomery)
    uint64_t beta[4] = { 0 }, gamma[4] = { 0 };
    // per proving key
    void* key_coeff[BBG_QP_EXT_COUNT] = {}; // indexed by key_slot(id)
    void* key_coset[BBG_QP_EXT_COUNT] = {};
    void* sigma_lagrange[4] = {};
    // generation of the coefficient form each slot holds, and the generation the other forms were derived from (or supplied at): a form
    // older than its coefficients is stale and finalize re-derives it in place -- re-registering a selector on a live handle replaces
    // EVERY form of it (bbg.h), not only the one rounds 5 / 6 read
    unsigned coeff_gen[BBG_QP_EXT_COUNT] = {};
    unsigned coset_gen[BBG_QP_EXT_COUNT] = {};
    unsigned sigma_lagrange_gen[4] = {};
    bool key_final = false;
    // per proof
    void* wire_lagrange[4] = {};
    void* wire_coeff[4] = {};
    void* z_coeff = nullptr;
    void* coset[5] = {};     // w_1..w_4, z on the 4n coset
    void* quotient = nullptr; // 4n
    void* linear = nullptr;   // n
    void* opening[2] = {};    // n + 1 each
    void* tmp = nullptr;      // n + 1
    void* d_jac = nullptr;    // MAX_RESULTS x 96 B
    char* h_pin = nullptr;    // pinned host staging (results down, blinding rows up)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[4] = {};
    int stage = 0; // rounds completed in the current proof (guards the call order)
    // Round 1 may already have queued the wires' 4n coset forms (option prover_early_cosets); round 3 skips them only if they were made
    // from THIS proof's wire_coeff: round 1 is the only writer of wire_coeff and starts a new proof_seq before it touches them, so
    // coset[0 .. width) are current exactly when wire_cosets_seq == proof_seq (round-4 advisor: the flag alone was not tied to a proof)
    uint64_t proof_seq = 0, wire_cosets_seq = ~0ull;
    std::vector<void*> allocs;
    size_t device_bytes = 0; // sum of `allocs` (bbg_prover_device_bytes, bbg_memory_report)
};

namespace bbg {
// live handles of the process: bbg_memory_report totals the ones of a context
static std::mutex g_prover_mu;
static std::set<bbg_prover*> g_live_provers;
void prover_report(const bbg_ctx* ctx, size_t* bytes, unsigned* count)
{
    std::lock_guard<std::mutex> lk(g_prover_mu);
    for (const bbg_prover* p : g_live_provers)
        if (p->ctx == ctx) {
            *bytes += p->device_bytes;
            (*count)++;
        }
}
} // namespace bbg

namespace {

#define CHECK_P(p)                                                                                                   \
    do {                                                                                                             \
        if (!(p) || !(p)->ctx) { set_error("null bbg_prover"); return BBG_E_INVALID; }                               \
        hipError_t _e = hipSetDevice((p)->ctx->device);                                                              \
        if (_e != hipSuccess) return hip_fail(_e, "hipSetDevice", __FILE__, __LINE__);                               \
    } while (0)

// option "prover_fail_round" (tests only): fail this round once, the way a device error in the middle of a proof would
#define FAIL_INJECT(p, round)                                                                                        \
    do {                                                                                                             \
        if ((p)->ctx->prover_fail_round == (round)) {                                                                \
            (p)->ctx->prover_fail_round = 0;                                                                         \
            set_error("injected failure in prover round " #round " (option prover_fail_round)");                     \
            return BBG_E_HIP;                                                                                        \
        }                                                                                                            \
    } while (0)

int dev_alloc(bbg_prover* p, void** out, size_t bytes)
{
    BBG_HIP(hipMalloc(out, bytes));
    p->allocs.push_back(*out);
    p->device_bytes += bytes;
    return BBG_OK;
}

// The independent commitments of a round -- the reference queues them and processes the queue as one unit (prover.cpp:66-74 the wires,
// :120-135 the quotient parts, work_queue.hpp:208-282) -- go through ONE sort / accumulate / reduce launch set, `max_batch` at a time
// (option "prover_msm_batch": 0 / 1 = one launch set per commitment, the round-3 behaviour, A/B).  Result k lands at d_jac + 96 (first + k).
int commit(bbg_prover* p, int count, const void* const* d_polys, const size_t* lens, int first, hipStream_t st, bool tail = false)
{
    const int max_batch = std::max(1, std::min(p->ctx->prover_msm_batch, BBG_MSM_BATCH_MAX));
    const size_t zero[BBG_MSM_BATCH_MAX] = { 0 };
    // `tail`: the host waits for these commitments with nothing else queued (rounds 4 and 6): option prover_tail_window trades windows for buckets there
    const int saved_window = p->ctx->msm_window;
    if (tail && p->ctx->prover_tail_window && !saved_window) p->ctx->msm_window = p->ctx->prover_tail_window;
    int rc = BBG_OK;
    for (int k = 0; k < count && !rc; k += max_batch) {
        const int c = std::min(max_batch, count - k);
        rc = msm_run_batch(p->ctx, p->srs->s, c, d_polys + k, zero, lens + k, (char*)p->d_jac + (size_t)(first + k) * 96, st);
    }
    p->ctx->msm_window = saved_window;
    return rc;
}
int grid_for(size_t n, int block) { return (int)((n + block - 1) / block); }

// the MSMs of a round keep their reduce phase on the auxiliary stream; RAII so that every exit path restores the option
struct AsyncReduce {
    bbg_ctx* ctx;
    bool saved;
    explicit AsyncReduce(bbg_ctx* c) : ctx(c), saved(c->msm_async_reduce) { c->msm_async_reduce = true; }
    ~AsyncReduce() { ctx->msm_async_reduce = saved; }
};

// commitments of a round: join the reduce phases, one copy down, ONE host synchronisation.  They leave as g1::element (Jacobian, 96
// bytes), what pippenger_unsafe returns; the caller normalises exactly as work_queue::process_queue does with an MSM result
// (g1::affine_element(result), work_queue.hpp:233-239) -- a Fermat inversion is 0.27 ms of single-lane latency on the device (it was
// 1.1 ms of every 2^20-gate proof, profiles/r02_prover_kernel_stats_v1.txt) and microseconds on a host core.
int fetch_commitments(bbg_prover* p, size_t count, uint64_t* out, hipStream_t st)
{
    int rc = msm_join(p->ctx, st);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(p->h_pin + PIN_AFFINE, p->d_jac, count * 96, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    memcpy(out, p->h_pin + PIN_AFFINE, count * 96);
    return BBG_OK;
}

// n coefficients -> their values on the 4n coset (the FFT work item, work_queue.hpp:252-264; the +4 wrap-around entries the host
// prover appends are not needed: the widget kernels index modulo 4n)
int to_coset(bbg_prover* p, const void* d_coeff, void* d_out, hipStream_t st)
{
    return ntt_coset_extend(p->ctx, d_coeff, p->n, d_out, p->log2n + 2, st); // no staging copy, no zero fill, g^j fused into the first load
}

// the wires' 4n coset forms (the FFT work items of round 3, prover.cpp:255-264).  Up to 2^19-point domains the transforms of all wires go
// through one launch set (r5: a 2^18-point transform has 128 tiles for 256 CUs); larger ones fill the chip by themselves.
int wires_to_coset(bbg_prover* p, hipStream_t st)
{
    if (p->log2n + 2 <= 19 && p->ctx->prover_ntt_batch) return ntt_coset_extend_batch(p->ctx, p->width, p->wire_coeff, p->n, p->coset, p->log2n + 2, st);
    for (int k = 0; k < p->width; k++) {
        int rc = to_coset(p, p->wire_coeff[k], p->coset[k], st);
        if (rc) return rc;
    }
    return BBG_OK;
}

// slot of a key polynomial id in key_coeff / key_coset (the widget table's index), -1 for ids that are not key polynomials
int key_slot(int id)
{
    if (id > BBG_QP_Z && id < BBG_QP_COUNT) return id;
    if (id == BBG_PP_Q_MIMC_COEFFICIENT) return BBG_QP_EXT_Q_MIMC_COEFFICIENT;
    if (id == BBG_PP_Q_MIMC_SELECTOR) return BBG_QP_EXT_Q_MIMC_SELECTOR;
    return -1;
}

// device address and length of a polynomial id (bbg_quotient_poly / bbg_prover_poly) in coefficient form
int coeff_poly(const bbg_prover* p, int id, const void** ptr, size_t* len)
{
    const size_t n = p->n;
    *len = n;
    if (id >= BBG_QP_W_1 && id <= BBG_QP_W_4) *ptr = id - BBG_QP_W_1 < p->width ? p->wire_coeff[id - BBG_QP_W_1] : nullptr;
    else if (id == BBG_QP_Z) *ptr = p->z_coeff;
    else if (id > BBG_QP_Z && id < BBG_QP_LAGRANGE_1) *ptr = p->key_coeff[id];
    else if (id == BBG_PP_QUOTIENT) { *ptr = p->quotient; *len = 4 * n; }
    else if (id >= BBG_PP_T_1 && id <= BBG_PP_T_4) {
        *ptr = (const char*)p->quotient + (size_t)(id - BBG_PP_T_1) * n * 32;
        if (p->width == 3 && id == BBG_PP_T_3) *len = n + 1; // t_high of StandardPLONK has n + 1 coefficients (prover.cpp:99-137)
    }
    else if (id == BBG_PP_LINEAR) *ptr = p->linear;
    else if (id == BBG_PP_OPENING) *ptr = p->opening[0];
    else if (id == BBG_PP_SHIFTED_OPENING) *ptr = p->opening[1];
    else if (id == BBG_PP_Q_MIMC_COEFFICIENT || id == BBG_PP_Q_MIMC_SELECTOR) *ptr = p->key_coeff[key_slot(id)];
    else *ptr = nullptr;
    if (!*ptr) { set_error("bbg_prover: polynomial id not available in coefficient form (unknown id, or never registered)"); return BBG_E_INVALID; }
    return BBG_OK;
}

} // namespace

extern "C" {

int bbg_prover_create(bbg_ctx* ctx, bbg_srs* srs, unsigned log2n, int program_width, const uint64_t* generators, bbg_prover** out)
{
    if (program_width != 3 && program_width != 4) { set_error("bbg_prover_create: program_width must be 3 (StandardPLONK) or 4 (TurboPLONK)"); return BBG_E_INVALID; }
    return bbg_prover_create_flavour(ctx, srs, log2n, program_width == 4 ? BBG_FLAVOUR_TURBO : BBG_FLAVOUR_STANDARD, generators, out);
}

int bbg_prover_create_flavour(bbg_ctx* ctx, bbg_srs* srs, unsigned log2n, int flavour, const uint64_t* generators, bbg_prover** out)
{
    if (!ctx || !srs || !generators || !out) { set_error("bbg_prover_create: null argument"); return BBG_E_INVALID; }
    BBG_HIP(hipSetDevice(ctx->device));
    if (flavour != BBG_FLAVOUR_TURBO && flavour != BBG_FLAVOUR_STANDARD && flavour != BBG_FLAVOUR_MIMC) {
        set_error("bbg_prover_create_flavour: unknown flavour");
        return BBG_E_INVALID;
    }
    const int program_width = flavour == BBG_FLAVOUR_TURBO ? 4 : 3;
    if (log2n < 3 || log2n > 26) { set_error("bbg_prover_create: need 3 <= log2n <= 26 (the quotient lives on the 4n domain)"); return BBG_E_INVALID; }
    const size_t n = (size_t)1 << log2n;
    if (srs->s.device != ctx->device) {
        set_error("bbg_prover_create: the SRS lives on another device than the context (register it on this context)");
        return BBG_E_INVALID;
    }
    if (srs->s.n < n + (program_width == 3 ? 1 : 0)) {
        set_error("bbg_prover_create: the SRS needs n points (n + 1 for StandardPLONK: t_high has n + 1 coefficients)");
        return BBG_E_INVALID;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    bbg_prover* p = new bbg_prover;
    p->ctx = ctx;
    p->srs = srs;
    srs->refs.fetch_add(1); // released in bbg_prover_destroy
    p->device = ctx->device;
    p->log2n = log2n;
    p->width = program_width;
    p->flavour = flavour;
    p->n = n;
    memcpy(p->gens, generators, sizeof(p->gens));
    int rc = BBG_OK;
    for (int k = 0; k < program_width && !rc; k++) {
        rc = dev_alloc(p, &p->wire_lagrange[k], n * 32);
        if (!rc) rc = dev_alloc(p, &p->wire_coeff[k], n * 32);
    }
    for (int k = 0; k < 5 && !rc; k++)
        if (k < program_width || k == 4) rc = dev_alloc(p, &p->coset[k], 4 * n * 32);
    if (!rc) rc = dev_alloc(p, &p->z_coeff, n * 32);
    if (!rc) rc = dev_alloc(p, &p->quotient, 4 * n * 32);
    if (!rc) rc = dev_alloc(p, &p->linear, n * 32);
    if (!rc) rc = dev_alloc(p, &p->opening[0], (n + 1) * 32);
    if (!rc) rc = dev_alloc(p, &p->opening[1], (n + 1) * 32);
    if (!rc) rc = dev_alloc(p, &p->tmp, (n + 1) * 32);
    if (!rc) rc = dev_alloc(p, &p->d_jac, MAX_RESULTS * 96);
    hipError_t e = hipSuccess;
    if (!rc) e = hipHostMalloc((void**)&p->h_pin, PIN_BYTES, hipHostMallocDefault);
    if (!rc && e == hipSuccess) e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    for (int k = 0; k < 4 && !rc && e == hipSuccess; k++) e = hipEventCreateWithFlags(&p->ev_up[k], hipEventDisableTiming);
    if (!rc && e != hipSuccess) rc = hip_fail(e, "bbg_prover_create", __FILE__, __LINE__);
    // twiddle / coset tables of the two domains (evaluation_domain::compute_lookup_table of proving_key::init, proving_key.cpp:52-57)
    if (!rc) rc = ntt_prepare(ctx, log2n);
    if (!rc) rc = ntt_prepare(ctx, log2n + 2);
    if (rc) {
        bbg_prover_destroy(p);
        return rc;
    }
    {
        std::lock_guard<std::mutex> lk2(g_prover_mu);
        g_live_provers.insert(p);
    }
    *out = p;
    return BBG_OK;
}

void bbg_prover_destroy(bbg_prover* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_prover_mu);
        g_live_provers.erase(p);
    }
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void* a : p->allocs) (void)hipFree(a);
    if (p->srs) bbg_srs_free(p->srs);
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    for (int k = 0; k < 4; k++)
        if (p->ev_up[k]) (void)hipEventDestroy(p->ev_up[k]);
    delete p;
}

int bbg_prover_set_key_poly(bbg_prover* p, int id, int form, const uint64_t* values)
{
    CHECK_P(p);
    if (!values) { set_error("bbg_prover_set_key_poly: null values"); return BBG_E_INVALID; }
    const int ks = key_slot(id);
    if (ks < 0) { set_error("bbg_prover_set_key_poly: id must be a selector / permutation / L_1 polynomial"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    const size_t n = p->n;
    void** slot = nullptr;
    size_t count = n;
    if (form == BBG_FORM_COEFF && id != BBG_QP_LAGRANGE_1) slot = &p->key_coeff[ks];
    else if (form == BBG_FORM_LAGRANGE && id >= BBG_QP_SIGMA_1 && id <= BBG_QP_SIGMA_4) slot = &p->sigma_lagrange[id - BBG_QP_SIGMA_1];
    else if (form == BBG_FORM_COSET) { slot = &p->key_coset[ks]; count = 4 * n; }
    if (!slot) { set_error("bbg_prover_set_key_poly: this polynomial is not kept in that form"); return BBG_E_INVALID; }
    if (!*slot) {
        int rc = dev_alloc(p, slot, count * 32);
        if (rc) return rc;
    }
    BBG_HIP(hipMemcpyAsync(*slot, values, count * 32, hipMemcpyHostToDevice, p->ctx->stream));
    BBG_HIP(hipStreamSynchronize(p->ctx->stream)); // the caller's array may go away as soon as this returns
    if (form == BBG_FORM_COEFF) p->coeff_gen[ks]++;                                              // every other form of this id is now stale
    else if (form == BBG_FORM_COSET) p->coset_gen[ks] = p->coeff_gen[ks];                        // supplied for the current coefficients
    else p->sigma_lagrange_gen[id - BBG_QP_SIGMA_1] = p->coeff_gen[ks];
    p->key_final = false;
    return BBG_OK;
}

int bbg_prover_finalize_key(bbg_prover* p)
{
    CHECK_P(p);
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    hipStream_t st = p->ctx->stream;
    const size_t n = p->n;
    int rc = BBG_OK;
    for (int id = BBG_QP_SIGMA_1; id < BBG_QP_EXT_COUNT && !rc; id++) { // slots: the widget table's indices
        if (id == BBG_QP_LAGRANGE_1 || !p->key_coeff[id]) continue;
        if (!p->key_coset[id] || p->coset_gen[id] != p->coeff_gen[id]) { // coefficient form -> values on the 4n coset (compute_proving_key's selector FFTs)
            if (!p->key_coset[id]) rc = dev_alloc(p, &p->key_coset[id], 4 * n * 32);
            if (!rc) rc = to_coset(p, p->key_coeff[id], p->key_coset[id], st);
            if (!rc) p->coset_gen[id] = p->coeff_gen[id];
        }
        if (!rc && id <= BBG_QP_SIGMA_4 &&
            (!p->sigma_lagrange[id - BBG_QP_SIGMA_1] || p->sigma_lagrange_gen[id - BBG_QP_SIGMA_1] != p->coeff_gen[id])) { // sigma in Lagrange base, read by the grand product
            void** slot = &p->sigma_lagrange[id - BBG_QP_SIGMA_1];
            if (!*slot) rc = dev_alloc(p, slot, n * 32);
            if (!rc) BBG_HIP(hipMemcpyAsync(*slot, p->key_coeff[id], n * 32, hipMemcpyDeviceToDevice, st));
            if (!rc) rc = ntt_run(p->ctx, *slot, p->log2n, BBG_FFT, 0, nullptr, st);
            if (!rc) p->sigma_lagrange_gen[id - BBG_QP_SIGMA_1] = p->coeff_gen[id];
        }
    }
    if (!rc && !p->key_coset[BBG_QP_LAGRANGE_1]) {
        // L_1(X) = (X^n - 1) / (n (X - 1)) = n^-1 (1 + X + ... + X^(n-1)): n equal coefficients, then the same coset FFT
        // (the reference evaluates the closed form, compute_lagrange_polynomial_fft, proving_key.cpp:62-64; same values)
        rc = dev_alloc(p, &p->key_coset[BBG_QP_LAGRANGE_1], 4 * n * 32);
        void* dc = nullptr;
        if (!rc) rc = ntt_domain_consts(p->ctx, p->log2n, &dc);
        if (!rc) {
            hipLaunchKernelGGL(k_fill_const, dim3(grid_for(n, 256)), dim3(256), 0, st, (Fr*)p->tmp, (const Fr*)&((const DomainConsts*)dc)->n_inv, n);
            rc = to_coset(p, p->tmp, p->key_coset[BBG_QP_LAGRANGE_1], st);
        }
    }
    if (rc) return rc;
    const int need_sigma = p->width;
    for (int k = 0; k < need_sigma; k++)
        if (!p->sigma_lagrange[k] || !p->key_coset[BBG_QP_SIGMA_1 + k] || !p->key_coeff[BBG_QP_SIGMA_1 + k]) {
            set_error("bbg_prover_finalize_key: sigma_1..sigma_width must be registered (coefficient form)");
            return BBG_E_INVALID;
        }
    BBG_HIP(hipGetLastError());
    BBG_HIP(hipStreamSynchronize(st));
    p->key_final = true;
    p->stage = 0;
    return BBG_OK;
}

int bbg_prover_round1(bbg_prover* p, const uint64_t* const* wires_lagrange, uint64_t* commitments)
{
    CHECK_P(p);
    if (!wires_lagrange || !commitments) { set_error("bbg_prover_round1: null argument"); return BBG_E_INVALID; }
    if (!p->key_final) { set_error("bbg_prover_round1: call bbg_prover_finalize_key first"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    FAIL_INJECT(p, 1);
    hipStream_t st = p->ctx->stream;
    AsyncReduce ar(p->ctx);
    const size_t n = p->n;
    for (int k = 0; k < p->width; k++)
        if (!wires_lagrange[k]) { set_error("bbg_prover_round1: null wire"); return BBG_E_INVALID; }
    p->stage = 0;   // a new proof: nothing of the previous one may be taken for this one's if the round fails half way
    p->proof_seq++; // ... including coset forms of the previous wires
    // The wires travel on the copy stream (pageable host memory: each copy call returns when its data has been staged), the commitments go
    // in groups through one launch set each (commit()).  Small circuits (upload negligible) commit all wires at once.  Large ones keep one
    // wire per group: wire k+1 travels (0.65 ms at 2^20 gates) while wire k is transformed and committed (1.5 ms) -- a group of two would
    // start 0.65 ms later to save 0.15 ms of launch-set overhead (measured: profiles/r04_batch_ab.txt).
    const int max_batch = std::max(1, std::min(p->ctx->prover_msm_batch, BBG_MSM_BATCH_MAX));
    const int group = (max_batch == 1 || p->log2n > 17) ? 1 : std::min(max_batch, p->width);
    const size_t lens[4] = { n, n, n, n };
    for (int k0 = 0; k0 < p->width; k0 += group) {
        const int cnt = std::min(group, p->width - k0);
        for (int k = k0; k < k0 + cnt; k++) {
            BBG_HIP(hipMemcpyAsync(p->wire_lagrange[k], wires_lagrange[k], n * 32, hipMemcpyHostToDevice, p->copy_stream));
            BBG_HIP(hipEventRecord(p->ev_up[k], p->copy_stream));
        }
        for (int k = k0; k < k0 + cnt; k++) BBG_HIP(hipStreamWaitEvent(st, p->ev_up[k], 0));
        // out of place (no staging copy); the wires of a group in ONE launch set (r5): at n <= 2^17 a single transform has at most 64 tiles
        int rc0 = BBG_OK;
        if (p->ctx->prover_ntt_batch) rc0 = ntt_ifft_to_batch(p->ctx, cnt, p->wire_lagrange + k0, p->wire_coeff + k0, p->log2n, st);
        else
            for (int k = k0; k < k0 + cnt && !rc0; k++) rc0 = ntt_ifft_to(p->ctx, p->wire_lagrange[k], p->wire_coeff[k], p->log2n, st);
        if (rc0) return rc0;
        int rc = commit(p, cnt, p->wire_coeff + k0, lens, k0, st);
        if (rc) return rc;
    }
    // The wires' values on the 4n coset (the FFT work items of round 3, prover.cpp:255-264) depend on nothing but the wires: queued here they
    // run beside the last commitment's accumulation and fill its reduce phase -- a chain of short kernels that leaves most of the chip idle at
    // the end of the round -- instead of standing in front of round 3's grand product.  From 2^18 gates (option -1, automatic): below, the
    // transforms beside the reduce chain slow that chain by more than they take in front of round 3, where they hide the grand product's
    // single inversion (r5: a 2^16-gate proof 3.59 -> 3.51 ms, 2^17 5.13 -> 5.00; 2^18 and 2^19 level; 2^20 24.13 against 24.23 in front of round 3)
    const int early = p->ctx->prover_early_cosets;
    if (early > 0 || (early < 0 && p->log2n >= 18)) {
        int rc = wires_to_coset(p, st);
        if (rc) return rc;
        p->wire_cosets_seq = p->proof_seq;
    }
    int rc = fetch_commitments(p, (size_t)p->width, commitments, st);
    if (rc) return rc;
    BBG_HIP(hipStreamSynchronize(p->copy_stream));
    p->stage = 1;
    return BBG_OK;
}

int bbg_prover_round3(bbg_prover* p, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* blind, uint64_t z_commitment[12])
{
    CHECK_P(p);
    if (!beta || !gamma || !blind || !z_commitment) { set_error("bbg_prover_round3: null argument"); return BBG_E_INVALID; }
    if (p->stage < 1) { set_error("bbg_prover_round3: round 1 has not run for this proof"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    FAIL_INJECT(p, 3);
    hipStream_t st = p->ctx->stream;
    AsyncReduce ar(p->ctx);
    const size_t n = p->n;
    memcpy(p->beta, beta, 32);
    memcpy(p->gamma, gamma, 32);
    uint64_t ch[20];
    memcpy(ch, beta, 32);
    memcpy(ch + 4, gamma, 32);
    memcpy(ch + 8, p->gens + 4, 96); // k1..k3
    // The grand product needs ONE inversion, a 0.25 ms dependency chain on a single lane: it runs on the (idle) copy stream while the
    // main stream does the part of the round that does not depend on z -- the wires' coset FFTs (the FFT work items, prover.cpp:255-264)
    int rc = permutation_grand_product_begin(p->ctx, p->width, p->wire_lagrange, p->sigma_lagrange, p->log2n, ch, p->z_coeff, st, p->copy_stream,
                                             p->ev_up[0], p->ev_up[1]);
    const bool wire_cosets_current = p->wire_cosets_seq == p->proof_seq;
    if (!rc && !wire_cosets_current) rc = wires_to_coset(p, st);
    if (!rc) p->wire_cosets_seq = p->proof_seq;
    if (!rc) rc = permutation_grand_product_finish(p->ctx, p->width, p->wire_lagrange, p->sigma_lagrange, p->log2n, p->z_coeff, st, p->ev_up[1]);
    if (rc) return rc;
    // rows n-3 .. n-1 carry the zero-knowledge blinding of z (permutation_widget_impl.hpp:283-287); pinned staging, stream ordered
    memcpy(p->h_pin + PIN_BLIND, blind, 96);
    BBG_HIP(hipMemcpyAsync((char*)p->z_coeff + (n - 3) * 32, p->h_pin + PIN_BLIND, 96, hipMemcpyHostToDevice, st));
    rc = ntt_run(p->ctx, p->z_coeff, p->log2n, BBG_IFFT, 0, nullptr, st);
    if (!rc) rc = msm_run(p->ctx, p->srs->s, p->z_coeff, 0, n, p->d_jac, st);
    if (!rc) rc = to_coset(p, p->z_coeff, p->coset[4], st); // z on the 4n coset while its commitment's reduce phase runs beside it
    if (rc) return rc;
    rc = fetch_commitments(p, 1, z_commitment, st);
    if (rc) return rc;
    p->stage = 3;
    return BBG_OK;
}

int bbg_prover_round4(bbg_prover* p, const uint64_t alpha[4], const uint64_t public_input_delta[4], uint64_t* t_commitments)
{
    CHECK_P(p);
    if (!alpha || !public_input_delta || !t_commitments) { set_error("bbg_prover_round4: null argument"); return BBG_E_INVALID; }
    if (p->stage < 3) { set_error("bbg_prover_round4: round 3 has not run for this proof"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    FAIL_INJECT(p, 4);
    hipStream_t st = p->ctx->stream;
    AsyncReduce ar(p->ctx);
    const size_t n = p->n;
    const void* polys[BBG_QP_EXT_COUNT];
    for (int k = 0; k < BBG_QP_EXT_COUNT; k++) polys[k] = p->key_coset[k];
    for (int k = 0; k < 4; k++) polys[BBG_QP_W_1 + k] = k < p->width ? p->coset[k] : nullptr;
    polys[BBG_QP_Z] = p->coset[4];
    uint64_t ch[36];
    memcpy(ch, alpha, 32);               // alpha_base = alpha for the first widget (prover.cpp:304)
    memcpy(ch + 4, alpha, 32);
    memcpy(ch + 8, p->beta, 32);
    memcpy(ch + 12, p->gamma, 32);
    memcpy(ch + 16, public_input_delta, 32);
    memcpy(ch + 20, p->gens, 128);       // g, k1, k2, k3
    static const int TURBO[5] = { BBG_WIDGET_PERMUTATION, BBG_WIDGET_TURBO_ARITHMETIC, BBG_WIDGET_TURBO_FIXED_BASE, BBG_WIDGET_TURBO_RANGE,
                                  BBG_WIDGET_TURBO_LOGIC };                                        // turbo_composer.cpp:735-752
    static const int STANDARD[2] = { BBG_WIDGET_PERMUTATION_3, BBG_WIDGET_ARITHMETIC };            // standard_composer.cpp:569-577
    static const int MIMC[3] = { BBG_WIDGET_PERMUTATION_3, BBG_WIDGET_MIMC, BBG_WIDGET_ARITHMETIC }; // mimc_composer.cpp:285-294
    const int* widgets = p->flavour == BBG_FLAVOUR_TURBO ? TURBO : p->flavour == BBG_FLAVOUR_MIMC ? MIMC : STANDARD;
    const int widget_count = p->flavour == BBG_FLAVOUR_TURBO ? 5 : p->flavour == BBG_FLAVOUR_MIMC ? 3 : 2;
    int rc = quotient_widgets_chain(p->ctx, widgets, widget_count, polys, p->log2n + 2, ch, p->quotient, nullptr, st);
    // divide by Z*_H and back to coefficients (prover.cpp:337-341): the divisor is a fixed function of the point, so it is a table the
    // coset iFFT multiplies by as it first loads each evaluation -- no read-modify-write pass of 4n evaluations for the division
    bool divided = false;
    if (!rc && p->ctx->prover_fused_divide) {
        const void* divisor = nullptr;
        rc = poly_dpv_table(p->ctx, p->log2n, p->log2n + 2, 4, &divisor, st);
        if (rc == BBG_E_NOMEM) rc = BBG_OK; // no room for the table (32 bytes per point of the 4n domain): the separate pass below
        else if (!rc) {
            rc = ntt_coset_ifft_scaled(p->ctx, p->quotient, p->log2n + 2, divisor, st);
            if (rc == BBG_E_NOFUSE) rc = BBG_OK; // a single-pass domain: the separate pass below
            else divided = true;
        }
    }
    if (!rc && !divided) {
        rc = poly_divide_pseudo_vanishing(p->ctx, p->quotient, p->log2n, p->log2n + 2, 4, st);
        if (!rc) rc = ntt_run(p->ctx, p->quotient, p->log2n + 2, BBG_COSET_IFFT, 0, nullptr, st);
    }
    // T_1 .. T_width: n coefficients each; t_high of StandardPLONK has n + 1 (compute_quotient_pre_commitment, prover.cpp:117-137)
    if (!rc) {
        const void* parts[4];
        size_t lens[4];
        for (int k = 0; k < p->width; k++) {
            parts[k] = (char*)p->quotient + (size_t)k * n * 32;
            lens[k] = (p->width == 3 && k == 2) ? n + 1 : n;
        }
        rc = commit(p, p->width, parts, lens, 0, st, true);
    }
    if (rc) return rc;
    rc = fetch_commitments(p, (size_t)p->width, t_commitments, st);
    if (rc) return rc;
    p->stage = 4;
    return BBG_OK;
}

int bbg_prover_evaluate(bbg_prover* p, size_t count, const int* ids, const int* shifted, const uint64_t zeta[4], uint64_t* out)
{
    CHECK_P(p);
    if (!ids || !zeta || !out || count == 0 || count > 32) { set_error("bbg_prover_evaluate: bad argument (1..32 evaluations per call)"); return BBG_E_INVALID; }
    if (p->stage < 4) { set_error("bbg_prover_evaluate: round 4 has not run for this proof (wires / z / quotient are not this proof's yet)"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    FAIL_INJECT(p, 5);
    hipStream_t st = p->ctx->stream;
    const void* ptrs[32];
    size_t lens[32];
    for (size_t k = 0; k < count; k++) {
        int rc = coeff_poly(p, ids[k], &ptrs[k], &lens[k]);
        if (rc) return rc;
    }
    void* d_res = nullptr;
    int rc = poly_multi_evaluate(p->ctx, ptrs, lens, shifted, count, p->log2n, zeta, &d_res, st);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(p->h_pin + PIN_EVAL, d_res, count * 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    memcpy(out, p->h_pin + PIN_EVAL, count * 32);
    return BBG_OK;
}

int bbg_prover_linearise(bbg_prover* p, size_t count, const int* ids, const uint64_t* scalars, const uint64_t zeta[4], uint64_t r_eval[4])
{
    CHECK_P(p);
    if (!ids || !scalars || !zeta || !r_eval || count == 0 || count > 32) { set_error("bbg_prover_linearise: bad argument (1..32 terms)"); return BBG_E_INVALID; }
    if (p->stage < 4) { set_error("bbg_prover_linearise: round 4 has not run for this proof"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    hipStream_t st = p->ctx->stream;
    const void* ptrs[32];
    for (size_t k = 0; k < count; k++) {
        size_t len;
        int rc = coeff_poly(p, ids[k], &ptrs[k], &len);
        if (rc) return rc;
    }
    int rc = poly_lincomb(p->ctx, ptrs, scalars, count, nullptr, p->linear, p->n, st);
    if (rc) return rc;
    const void* lin = p->linear;
    const size_t len = p->n;
    void* d_res = nullptr;
    rc = poly_multi_evaluate(p->ctx, &lin, &len, nullptr, 1, p->log2n, zeta, &d_res, st);
    if (rc) return rc;
    BBG_HIP(hipMemcpyAsync(p->h_pin + PIN_EVAL, d_res, 32, hipMemcpyDeviceToHost, st));
    BBG_HIP(hipStreamSynchronize(st));
    memcpy(r_eval, p->h_pin + PIN_EVAL, 32);
    p->stage = 5;
    return BBG_OK;
}

int bbg_prover_round6(bbg_prover* p, size_t count_zeta, const int* ids_zeta, const uint64_t* scalars_zeta, size_t count_omega,
                      const int* ids_omega, const uint64_t* scalars_omega, const uint64_t zeta[4], const uint64_t zeta_omega[4],
                      const uint64_t* t_high_top_scalar, uint64_t pi_z[12], uint64_t pi_z_omega[12])
{
    CHECK_P(p);
    if (!ids_zeta || !scalars_zeta || !ids_omega || !scalars_omega || !zeta || !zeta_omega || !pi_z || !pi_z_omega || count_zeta > 32 ||
        count_omega > 32 || count_omega == 0) {
        set_error("bbg_prover_round6: bad argument");
        return BBG_E_INVALID;
    }
    if (p->stage < 4) { set_error("bbg_prover_round6: round 4 has not run for this proof"); return BBG_E_INVALID; }
    if (p->width == 3 && !t_high_top_scalar) { set_error("bbg_prover_round6: StandardPLONK needs the scalar of t_high's top coefficient"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    FAIL_INJECT(p, 6);
    hipStream_t st = p->ctx->stream;
    AsyncReduce ar(p->ctx);
    const size_t n = p->n;
    const void* ptrs[32];
    size_t len;
    for (size_t k = 0; k < count_zeta; k++) {
        int rc = coeff_poly(p, ids_zeta[k], &ptrs[k], &len);
        if (rc) return rc;
    }
    // F(X) = t_low(X) + sum_k scalar_k P_k(X)   (kate_commitment_scheme.cpp:216-222)
    int rc = poly_lincomb(p->ctx, ptrs, scalars_zeta, count_zeta, p->quotient, p->tmp, n, st);
    if (rc) return rc;
    size_t f_len = n;
    if (p->width == 3) { // F[n] = zeta^(2n) t[3n]: the opening polynomial of StandardPLONK has n + 1 coefficients (:196-205, :42)
        Fr s;
        memcpy(&s, t_high_top_scalar, 32);
        hipLaunchKernelGGL(k_mul_one, dim3(1), dim3(64), 0, st, (Fr*)p->tmp + n, (const Fr*)p->quotient + 3 * n, s);
        f_len = n + 1;
    }
    rc = poly_kate_opening_async(p->ctx, p->tmp, p->opening[0], f_len, zeta, nullptr, st);
    const bool together = p->ctx->prover_msm_batch >= 2; // PI_Z and PI_Z_OMEGA through one launch set (commit())
    if (!rc && !together) rc = msm_run(p->ctx, p->srs->s, p->opening[0], 0, n, p->d_jac, st);
    if (rc) return rc;
    for (size_t k = 0; k < count_omega; k++) {
        rc = coeff_poly(p, ids_omega[k], &ptrs[k], &len);
        if (rc) return rc;
    }
    rc = poly_lincomb(p->ctx, ptrs, scalars_omega, count_omega, nullptr, p->tmp, n, st);
    if (!rc) rc = poly_kate_opening_async(p->ctx, p->tmp, p->opening[1], n, zeta_omega, nullptr, st);
    if (!rc && !together) rc = msm_run(p->ctx, p->srs->s, p->opening[1], 0, n, (char*)p->d_jac + 96, st);
    if (!rc && together) {
        const size_t lens[2] = { n, n };
        rc = commit(p, 2, p->opening, lens, 0, st, true);
    }
    if (rc) return rc;
    uint64_t both[24];
    rc = fetch_commitments(p, 2, both, st);
    if (rc) return rc;
    memcpy(pi_z, both, 96);
    memcpy(pi_z_omega, both + 12, 96);
    p->stage = 0;
    return BBG_OK;
}

int bbg_prover_device_bytes(const bbg_prover* p, size_t* bytes)
{
    if (!p || !bytes) { set_error("bbg_prover_device_bytes: null argument"); return BBG_E_INVALID; }
    *bytes = p->device_bytes;
    return BBG_OK;
}

int bbg_prover_read_poly(bbg_prover* p, int id, int form, uint64_t* out, size_t count)
{
    CHECK_P(p);
    if (!out) { set_error("bbg_prover_read_poly: null out"); return BBG_E_INVALID; }
    std::lock_guard<std::mutex> lk(p->ctx->mu);
    const void* src = nullptr;
    size_t len = 0;
    if (form == BBG_FORM_COEFF) {
        int rc = coeff_poly(p, id, &src, &len);
        if (rc) return rc;
    } else if (form == BBG_FORM_LAGRANGE) {
        len = p->n;
        if (id >= BBG_QP_W_1 && id <= BBG_QP_W_4) src = p->wire_lagrange[id - BBG_QP_W_1];
        else if (id >= BBG_QP_SIGMA_1 && id <= BBG_QP_SIGMA_4) src = p->sigma_lagrange[id - BBG_QP_SIGMA_1];
    } else if (form == BBG_FORM_COSET) {
        len = 4 * p->n;
        if (id >= BBG_QP_W_1 && id <= BBG_QP_Z) src = p->coset[id];
        else if (key_slot(id) >= 0) src = p->key_coset[key_slot(id)];
    }
    if (!src || count > len) { set_error("bbg_prover_read_poly: polynomial not available in that form, or count too large"); return BBG_E_INVALID; }
    BBG_HIP(hipMemcpyAsync(out, src, count * 32, hipMemcpyDeviceToHost, p->ctx->stream));
    BBG_HIP(hipStreamSynchronize(p->ctx->stream));
    return BBG_OK;
}

} // extern "C"
