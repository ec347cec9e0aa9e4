/*
 * bbg.h -- C ABI of libbbg.so: the MI355X (gfx950) implementation of barretenberg's PLONK-prover
 * hot path -- Pippenger MSM over BN254 G1 and the radix-2 NTT family over BN254 Fr.
 *
 * This is the drop-in boundary.  Every entry point takes plain pointers and sizes in the
 * REFERENCE'S OWN MEMORY LAYOUT (paths relative to barretenberg/src/aztec/):
 *   fr / fq           32 B, 4 x u64 little-endian limbs, Montgomery form R = 2^256, any representative
 *                     in [0, 2p) accepted                        (ecc/fields/field.hpp:24,86)
 *   g1::affine_element 64 B  x || y                               (ecc/groups/affine_element.hpp:70-71)
 *   g1::element        96 B  x || y || z Jacobian, infinity = bit 63 of x.data[3]
 *                                                                 (ecc/groups/element.hpp:88-90, element_impl.hpp:497-516)
 * so a barretenberg build binds them with no marshalling (see INTEGRATION.md for the C++ shim that
 * re-exports scalar_multiplication::pippenger*() and polynomial_arithmetic::fft*() on top of this).
 *
 * All functions return 0 on success and a negative BBG_E* code on failure; bbg_last_error() gives a
 * thread-local message.  The reference has no status codes (throw_or_abort, common/throw_or_abort.hpp:5-13);
 * the C++ shim turns a non-zero return into std::runtime_error.
 *
 * There is no CPU fallback anywhere behind this ABI: without a HIP device bbg_init() fails.
 */
#ifndef BBG_H
#define BBG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBG_OK 0
#define BBG_E_INVALID (-1)  /* bad argument (null pointer, size out of range, n not a power of two ...) */
#define BBG_E_HIP (-2)      /* a HIP runtime call failed; see bbg_last_error() */
#define BBG_E_NODEVICE (-3) /* no gfx950 device visible */
#define BBG_E_NOMEM (-4)

typedef struct bbg_ctx bbg_ctx; /* one per GPU / per process rank */
typedef struct bbg_srs bbg_srs; /* device-resident SRS (the Pippenger point table) */

/* ---- context ---------------------------------------------------------------------------------------- */
int bbg_device_count(void);
/* Binds `device` (hipSetDevice), creates the context and its stream. */
int bbg_init(int device, bbg_ctx** out);
void bbg_destroy(bbg_ctx* ctx);
const char* bbg_last_error(void);
/* Blocks until everything queued on the context's stream has finished. */
int bbg_sync(bbg_ctx* ctx);
/* With the option "msm_async_reduce" = 1 the last phase of bbg_msm_device (bucket reduction) is queued on an auxiliary
 * stream so that it overlaps the next call's sort / accumulation (the prover issues several independent MSMs per round,
 * prover.cpp:66-73).  The result buffer is then complete after bbg_sync(), or, for stream-ordered consumers, after
 * bbg_join(): it makes the context stream wait (on the device, no host sync) for all outstanding reductions. */
int bbg_join(bbg_ctx* ctx);
/* Like bbg_join but leaves the `lag` most recent reductions outstanding (lag = 1: wait for everything except the
 * MSM issued last) -- lets a caller consume result i-1 while MSM i is still reducing. */
int bbg_join_lag(bbg_ctx* ctx, int lag);
/* Use a caller-owned HIP stream (hipStream_t passed as void*; e.g. torch.cuda.current_stream().cuda_stream).  Every *_device
 * entry point enqueues on the context stream and returns; a caller that fills or reads those device buffers on ANOTHER stream
 * (a framework's memset, a copy) must order the two itself -- sharing one stream through this call is the simple way. */
int bbg_set_stream(bbg_ctx* ctx, void* hip_stream);

/* ---- SRS: replaces scalar_multiplication::Pippenger (ecc/curves/bn254/scalar_multiplication/pippenger.hpp:35-52,
 *      pippenger.cpp:7-37) and the C binding new_pippenger/delete_pippenger (.../c_bind.cpp:17-29). --------------
 * points: n affine points in Montgomery form.  stride_bytes = 64 for a plain array P_0..P_{n-1} (what
 * io::read_transcript_g1 yields, srs/io.cpp:134-162) or 128 for the reference's interleaved endomorphism table
 * [P_i, (beta*x_i, -y_i)] (generate_pippenger_point_table, scalar_multiplication.cpp:104-112), whose odd entries
 * are ignored: the (beta*x, -y) twin is one Fq multiplication on chip.  The table is uploaded to HBM once. */
int bbg_srs_register(bbg_ctx* ctx, const uint64_t* points, size_t n, size_t stride_bytes, bbg_srs** out);
/* Same, from a device buffer of n plain 64-byte points (copied). */
int bbg_srs_register_device(bbg_ctx* ctx, const void* d_points, size_t n, bbg_srs** out);
/* Synthetic SRS P_i = (a + i*s)*G generated on the GPU (the Ignition transcript is absent from the
 * reference snapshot).  a, s: plain 64-bit scalars, s != 0. */
int bbg_srs_synth_linear(bbg_ctx* ctx, uint64_t a, uint64_t s, size_t n, bbg_srs** out);
/* Synthetic SRS P_i = k_i*G, k_i = mix64(seed + i) | 1 (splitmix64 finaliser): no small linear relations between
 * the bases, which pippenger_unsafe requires of its inputs (scalar_multiplication.cpp:908-921). */
int bbg_srs_synth_hashed(bbg_ctx* ctx, uint64_t seed, size_t n, bbg_srs** out);
/* Reads an Ignition-format transcript file (manifest + big-endian points; srs/io.cpp:11-162): result is
 * monomials[0] = G followed by the file's points, num_points in total -- exactly read_transcript_g1. */
int bbg_srs_load_transcript(bbg_ctx* ctx, const char* path, size_t num_points, bbg_srs** out);
/* The same from memory: Pippenger(uint8_t const* points, size_t num_points) (pippenger.cpp:7-17, the C binding new_pippenger): `points` =
 * (num_points - 1) x 64 bytes in the transcript encoding (srs/io.cpp:47-67); monomials[0] = G. */
int bbg_srs_register_transcript_buffer(bbg_ctx* ctx, const uint8_t* points, size_t num_points, bbg_srs** out);
/* The inverse: writes the SRS as Ignition-format files dir/transcript00.dat, 01, ... holding points 1 .. n-1 (point 0 is the
 * generator every reader supplies itself, srs/io.cpp:137), points_per_file per file (0 = one file); manifest and point encoding of
 * srs/io.cpp:11-67.  g2_x_raw (may be NULL): 128 bytes stored as file 00's single G2 point, as given.  Each file ends with the
 * BLAKE2b-512 checksum of what precedes it. */
int bbg_srs_write_transcript(bbg_srs* srs, const char* dir, size_t points_per_file, const uint8_t* g2_x_raw);
/* BLAKE2b-512 (RFC 7693) of a byte string: the checksum that closes a transcript file.  Host only, needs no GPU. */
int bbg_transcript_checksum(const void* data, size_t len, uint8_t out[64]);
size_t bbg_srs_num_points(const bbg_srs* srs);
/* Copies points [from, from+count) back to the host (64-byte Montgomery affine each). */
int bbg_srs_read(bbg_srs* srs, size_t from, size_t count, uint64_t* out_points);
/* Shared ownership: bbg_srs_retain adds an owner, bbg_srs_free drops one; the device memory goes with the last owner.  bbg_prover_create
 * retains the SRS it is given (and bbg_prover_destroy releases it), so freeing or replacing a cached SRS never invalidates a live prover. */
int bbg_srs_retain(bbg_srs* srs);
void bbg_srs_free(bbg_srs* srs);

/* ---- MSM: replaces scalar_multiplication::pippenger / pippenger_unsafe
 *      (scalar_multiplication.cpp:853-929) and Pippenger::pippenger_unsafe(scalars, from, range) (pippenger.cpp:27-31),
 *      C binding pippenger_unsafe (c_bind.cpp:31-37). -------------------------------------------------------------
 * result = sum_{i<n} scalars[i] * P_{from+i}.  scalars: n x 4 limbs, Montgomery Fr.  out_jacobian: 12 limbs
 * (g1::element).  Exceptional cases (equal / opposite points, zero scalars, n = 0 -> infinity) are always handled,
 * i.e. the behaviour of handle_edge_cases = true; the "unsafe" entry of the shim maps here too. */
int bbg_msm(bbg_ctx* ctx, bbg_srs* srs, const uint64_t* scalars, size_t from, size_t n, uint64_t out_jacobian[12]);
/* Device-resident variant: d_scalars and d_out_jacobian (96 B) are device pointers; asynchronous on the context stream. */
int bbg_msm_device(bbg_ctx* ctx, bbg_srs* srs, const void* d_scalars, size_t from, size_t n, void* d_out_jacobian);
/* A round's independent commitments as ONE unit of work -- what the reference queues and processes together (the four wire commitments of
 * round 1, prover.cpp:66-74; the quotient parts of round 4, :120-135; work_queue.hpp:208-282): `count` (<= BBG_MSM_BATCH_MAX) MSMs over the SAME
 * SRS go through ONE sort / accumulate / reduce launch set, MSM k's entries filed under bucket set k (count x 2^(C-1) buckets), instead of
 * `count` latency chains.  out_jacobians[k] = sum_{i<n[k]} scalars[k][i] * P_{from[k]+i}; from == NULL means all zero; the n[k] may differ
 * (StandardPLONK's t_high has n + 1 coefficients).  Each result is the same group element bbg_msm gives for that MSM alone. */
#define BBG_MSM_BATCH_MAX 8
int bbg_msm_batch(bbg_ctx* ctx, bbg_srs* srs, size_t count, const uint64_t* const* scalars, const size_t* from, const size_t* n,
                  uint64_t* out_jacobians /* count x 12 limbs */);
/* Device-resident variant: d_scalars[k] are device pointers, d_out_jacobians holds count x 96 B; asynchronous on the context stream. */
int bbg_msm_batch_device(bbg_ctx* ctx, bbg_srs* srs, size_t count, const void* const* d_scalars, const size_t* from, const size_t* n,
                         void* d_out_jacobians);
/* The configuration an n-term MSM over `srs` (may be NULL: a fresh SRS of n points) would run with now: *window_bits = the widest bucket
 * window C (buckets = 2^(C-1)), *windows = the number of windows, i.e. table gathers and mixed additions per scalar.  Follows the option
 * "msm_window" and the resident-table rule for short MSMs over long SRSs. */
int bbg_msm_plan(bbg_ctx* ctx, const bbg_srs* srs, size_t n, int* window_bits, int* windows);
/* g1_sum (c_bind.cpp:39-46): sum of n Jacobian points (host arrays, 96 B each). */
int bbg_g1_sum(bbg_ctx* ctx, const uint64_t* jacobians, size_t n, uint64_t out_jacobian[12]);
/* Same on device buffers, asynchronous on the context stream (the multi-GPU combine after an all-gather of partials). */
int bbg_g1_sum_device(bbg_ctx* ctx, const void* d_jacobians, size_t n, void* d_out_jacobian);
/* g1::affine_element(result) (element_impl.hpp:51-68) for n points; output canonical Montgomery affine (64 B each). */
int bbg_g1_normalize(bbg_ctx* ctx, const uint64_t* jacobians, size_t n, uint64_t* out_affine);

/* ---- NTT family: replaces polynomial_arithmetic::fft/ifft/coset_fft/coset_ifft/... on fr* coeffs
 *      (polynomials/polynomial_arithmetic.cpp:374-484) and the C bindings coset_fft_with_generator_shift / ifft
 *      (plonk/proof_system/prover/c_bind.cpp:59-76). ----------------------------------------------------------- */
enum {
    BBG_FFT = 0,                             /* fft                            :374 */
    BBG_IFFT = 1,                            /* ifft                           :379 */
    BBG_COSET_FFT = 2,                       /* coset_fft(coeffs, domain)      :395 */
    BBG_COSET_IFFT = 3,                      /* coset_ifft                     :480 */
    BBG_FFT_WITH_CONSTANT = 4,               /* fft_with_constant              :387 */
    BBG_COSET_FFT_WITH_CONSTANT = 5,         /* coset_fft_with_constant        :458 */
    BBG_COSET_FFT_WITH_GENERATOR_SHIFT = 6,  /* coset_fft_with_generator_shift :465 */
    BBG_IFFT_WITH_CONSTANT = 7               /* ifft_with_constant             :471 */
};
/* In place on coeffs[2^log2n] (Montgomery Fr, natural order in and out).  generator_size = evaluation_domain::
 * generator_size (0 = whole domain): coset_fft scales only the first generator_size coefficients by g^j
 * (polynomial_arithmetic.cpp:397, proving_key.cpp:21-22).  constant: Montgomery Fr for ops 4-7, else NULL.
 * The per-size twiddle tables (evaluation_domain::compute_lookup_table) are built on first use and cached.
 * log2n <= 28, the 2-adicity of BN254 Fr (fr.hpp:27-30); every size up to 2^28 is checked against the reference's digests
 * (tests/golden/ntt_large.json: 2^25 .. 2^28 in round 6; the tables of a 2^28 domain take 5 x 32 n bytes = 40 GiB of HBM). */
int bbg_ntt(bbg_ctx* ctx, uint64_t* coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant);
int bbg_ntt_device(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, int op, size_t generator_size, const uint64_t* constant);
/* Pre-builds the tables for a domain size (outside any timed region, like compute_lookup_table()). */
int bbg_ntt_prepare(bbg_ctx* ctx, unsigned log2n);
/* How a 2^log2n transform runs NOW (options included), without building anything: *passes = kernel launches per transform, log_radix[q] =
 * log2 of pass q's radix (0 beyond the last pass), *tile_log = log2 elements per tile, *kernel = the pass kernel: 0 k_ntt_pass (radix 2 in LDS,
 * n < 2^11), 8 k_ntt_pass8, 81 k_ntt_pass8s (one-plane exchange), 29 k_ntt_pass29 (9 x 29-bit limbs).  What bench.py labels its NTT roofline with. */
int bbg_ntt_plan(bbg_ctx* ctx, unsigned log2n, int* passes, int log_radix[4], int* kernel, int* tile_log);
/* coset_fft(coeffs, small_domain, large_domain, ext) (:401-456): coeffs holds 2^log2n coefficients in a buffer of
 * ext * 2^log2n elements; result interleaves ext size-n coset FFTs at index ext*i + k. */
int bbg_coset_fft_split(bbg_ctx* ctx, uint64_t* coeffs, unsigned log2n, size_t ext);
int bbg_coset_fft_split_device(bbg_ctx* ctx, void* d_coeffs, unsigned log2n, size_t ext);
/* The prover's FFT work item as ONE call (work_queue.hpp:252-264, WorkType::FFT): copy the 2^log2n coefficients of a wire
 * into a zeroed domain of 2^log2_domain elements, coset_fft over that domain with generator_size = 2^log2n, and append
 * the first four results (polynomial::add_lagrange_base_coefficient x4).  coeffs: 2^log2n elements (read only);
 * out: 2^log2_domain + 4 elements.  Uploads n elements instead of 4n and spares the host the 4n copy / zero fill. */
int bbg_coset_fft_extend(bbg_ctx* ctx, const uint64_t* coeffs, unsigned log2n, unsigned log2_domain, uint64_t* out);

/* ---- building blocks of an NTT sharded across GPUs by residue class (aztec-2.0_amd/parallel.py::ntt_sharded; the
 *      reference's precedent is the 4-way coset split, work_queue.hpp:166-199, polynomial_arithmetic.cpp:401-456) ---- */
/* a[j] *= start * base^j, j < count (start may be NULL = 1); base/start Montgomery Fr on the host. */
int bbg_scale_powers_device(bbg_ctx* ctx, void* d_a, size_t count, const uint64_t* start, const uint64_t* base);
/* out = w_n^e for the 2^log2n domain (inverse != 0: w_n^-e); out = base^e. */
int bbg_fr_root_pow(bbg_ctx* ctx, unsigned log2n, uint64_t e, int inverse, uint64_t out[4]);
int bbg_fr_pow(bbg_ctx* ctx, const uint64_t base[4], uint64_t e, uint64_t out[4]);
/* out[t*len + q] = sum_{s<G} w_G^(s*t) in[s*len + q], G = 2^log2G <= 8, w_G = w_n^(n/G): the cross-rank DFT after the
 * all-to-all exchange. */
int bbg_cross_dft_device(bbg_ctx* ctx, const void* d_in, void* d_out, unsigned log2G, size_t len, unsigned log2n, int inverse);

/* ---- polynomial helpers between the NTTs and the MSMs, on device-resident Montgomery Fr arrays ----
 * polynomial_arithmetic::add / sub / mul (polynomials/polynomial_arithmetic.cpp:486-505): r[i] = a[i] (op) b[i], op = 0 add,
 * 1 sub, 2 mul; r may alias a or b. */
int bbg_poly_op_device(bbg_ctx* ctx, int op, const void* d_a, const void* d_b, void* d_r, size_t n);
/* out[i] = base[i] + sum_k polys[k][i] * scalars[k], k < count <= 32 (base may be NULL = 0; out may alias base): the
 * accumulation of the opening polynomials in KateCommitmentScheme::batch_open (kate_commitment_scheme.cpp:216-226).
 * d_polys: host array of device addresses; scalars: count Montgomery Fr on the host. */
int bbg_poly_linear_combination_device(bbg_ctx* ctx, const void* const* d_polys, const uint64_t* scalars, size_t count, const void* d_base,
                                       void* d_out, size_t n);
/* polynomial_arithmetic::evaluate (:507-538): out = sum_i coeffs[i] z^i (canonical Montgomery).  Synchronous. */
int bbg_poly_evaluate_device(bbg_ctx* ctx, const void* d_coeffs, size_t n, const uint64_t z[4], uint64_t out[4]);
/* polynomial_arithmetic::compute_kate_opening_coefficients (:727-750): dest = coefficients of (F(X) - F(z)) / (X - z),
 * f_out = F(z).  dest must not alias src.  Synchronous. */
int bbg_kate_opening_device(bbg_ctx* ctx, const void* d_src, void* d_dest, size_t n, const uint64_t z[4], uint64_t f_out[4]);
/* polynomial_arithmetic::divide_by_pseudo_vanishing_polynomial (:628-725): in place on the 2^log2_target coset evaluations;
 * src domain 2^log2_src, num_roots_cut roots cut out of the vanishing polynomial (the reference default is 4). */
int bbg_divide_by_pseudo_vanishing_device(bbg_ctx* ctx, void* d_evals, unsigned log2_src, unsigned log2_target, size_t num_roots_cut);

/* Host-buffer forms of the three helpers above (what the C++ shim binds polynomial_arithmetic::evaluate,
 * compute_kate_opening_coefficients and divide_by_pseudo_vanishing_polynomial to): upload, compute, download; synchronous.
 * bbg_kate_opening: dest may alias src, as the prover calls it (kate_commitment_scheme.cpp:231-235). */
int bbg_poly_evaluate(bbg_ctx* ctx, const uint64_t* coeffs, size_t n, const uint64_t z[4], uint64_t out[4]);
int bbg_kate_opening(bbg_ctx* ctx, const uint64_t* src, uint64_t* dest, size_t n, const uint64_t z[4], uint64_t f_out[4]);
int bbg_divide_by_pseudo_vanishing(bbg_ctx* ctx, uint64_t* evals, unsigned log2_src, unsigned log2_target, size_t num_roots_cut);

/* ---- device memory helpers for hosts that do not link HIP (bbmalloc/bbfree analogue, c_bind.cpp:11-15) ---- */
int bbg_dev_alloc(bbg_ctx* ctx, size_t bytes, void** d_ptr);
int bbg_dev_free(bbg_ctx* ctx, void* d_ptr);
int bbg_dev_upload(bbg_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int bbg_dev_download(bbg_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- quotient-polynomial pointwise kernels on the 4n coset domain: the widgets' compute_quotient_contribution of
 *      ProverBase::execute_fourth_round (prover.cpp:304-319).  Device-resident: every polynomial is an array of
 *      2^log2_large_domain (+4) Fr values on the device, as left behind by the coset FFTs ("*_fft" arrays of the proving key). ---- */
enum bbg_quotient_poly { /* index into d_polys[]; entries a widget does not read may be NULL */
    BBG_QP_W_1 = 0, BBG_QP_W_2, BBG_QP_W_3, BBG_QP_W_4, BBG_QP_Z,
    BBG_QP_SIGMA_1, BBG_QP_SIGMA_2, BBG_QP_SIGMA_3, BBG_QP_SIGMA_4,
    BBG_QP_Q_1, BBG_QP_Q_2, BBG_QP_Q_3, BBG_QP_Q_4, BBG_QP_Q_5, BBG_QP_Q_M, BBG_QP_Q_C,
    BBG_QP_Q_ARITH, BBG_QP_Q_FIXED_BASE, BBG_QP_Q_RANGE, BBG_QP_Q_LOGIC, BBG_QP_LAGRANGE_1,
    BBG_QP_COUNT,
    /* extension read only by BBG_WIDGET_MIMC: d_polys then has BBG_QP_EXT_COUNT entries (the first BBG_QP_COUNT as above) */
    BBG_QP_EXT_Q_MIMC_COEFFICIENT = BBG_QP_COUNT, BBG_QP_EXT_Q_MIMC_SELECTOR,
    BBG_QP_EXT_COUNT
};
enum bbg_quotient_widget {
    BBG_WIDGET_PERMUTATION = 0,      /* ProverPermutationWidget<4,false> (permutation_widget_impl.hpp:316-420): ASSIGNS the quotient */
    BBG_WIDGET_TURBO_ARITHMETIC = 1, /* TurboArithmeticKernel (turbo_arithmetic_widget.hpp): the transition widgets ACCUMULATE */
    BBG_WIDGET_TURBO_FIXED_BASE = 2, /* TurboFixedBaseKernel */
    BBG_WIDGET_TURBO_RANGE = 3,      /* TurboRangeKernel */
    BBG_WIDGET_TURBO_LOGIC = 4,      /* TurboLogicKernel */
    BBG_WIDGET_PERMUTATION_3 = 5,    /* StandardPLONK: ProverPermutationWidget<3,false> (w_4 / sigma_4 not read): ASSIGNS */
    BBG_WIDGET_ARITHMETIC = 6,       /* StandardPLONK: ArithmeticKernel (arithmetic_widget.hpp): accumulates */
    BBG_WIDGET_MIMC = 7              /* MiMCComposer: MiMCKernel (mimc_widget.hpp:17-52) over w_1..w_3 and the two BBG_QP_EXT selectors: accumulates */
};
/* challenges: 9 Montgomery Fr (4 limbs each) on the host -- alpha_base (this widget's starting power of alpha), alpha,
 * beta, gamma, public_input_delta, g (the small domain's coset generator), k1, k2, k3 (fr::coset_generator(0..2)).
 * alpha_base_out (may be NULL) receives the alpha_base for the next widget, as compute_quotient_contribution returns it. */
int bbg_quotient_widget_device(bbg_ctx* ctx, int widget, const void* const d_polys[BBG_QP_COUNT], unsigned log2_large_domain,
                               const uint64_t* challenges, void* d_quotient, uint64_t* alpha_base_out);

/* The permutation grand product z of ProverPermutationWidget<4,false>::compute_round_commitments
 * (permutation_widget_impl.hpp:48-268, before the blinding of the last rows and the ifft): d_z[0] = 1,
 * d_z[j+1] = prod_{i<=j} prod_k (w_k[i] + gamma + beta K_k w^i) / (w_k[i] + gamma + beta sigma_k[i]).  d_wires / d_sigmas: 4 device
 * arrays of 2^log2n Lagrange-base values each; challenges: 5 Montgomery Fr on the host -- beta, gamma, k1, k2, k3; d_z: 2^log2n
 * values.  Asynchronous on the context stream. */
int bbg_permutation_grand_product_device(bbg_ctx* ctx, const void* const d_wires[4], const void* const d_sigmas[4], unsigned log2n,
                                         const uint64_t* challenges, void* d_z);

/* ---- resident prover rounds (SURVEY 8f-1 with 8f-2 / 8f-4 inside): the native, handle-based form of a PLONK prover's O(n) work.
 *      The host keeps the transcript (Fiat-Shamir hashing) and the O(1) challenge algebra -- in a barretenberg build that is
 *      shim/bbg_resident_prover.hpp, which drives these entry points from ProverBase's own public members in place of
 *      ProverBase::construct_proof (prover.cpp:420-436) and work_queue::process_queue (work_queue.hpp:208-282).
 *      One handle per proving key; selector / permutation polynomials are registered ONCE (explicit lifetime, no content
 *      fingerprints), wires go up per proof, 11 commitments and the opening evaluations come down.  All values Montgomery Fr; commitments
 *      leave as g1::element (Jacobian, 96 bytes = 12 limbs, what pippenger_unsafe returns) and the caller normalises them the way
 *      work_queue::process_queue does (g1::affine_element(result), work_queue.hpp:233-239).  Rounds must be called in order; each
 *      returns after ONE host sync. ---- */
typedef struct bbg_prover bbg_prover;
enum bbg_poly_form {
    BBG_FORM_COEFF = 0,    /* n coefficients (monomial form)                                   */
    BBG_FORM_LAGRANGE = 1, /* n values on the small domain                                     */
    BBG_FORM_COSET = 2     /* 4n values on the coset g * <w_4n> (the key's "*_fft" arrays)      */
};
enum bbg_prover_poly { /* continues enum bbg_quotient_poly: the polynomials a proof creates */
    BBG_PP_QUOTIENT = BBG_QP_COUNT, /* all 4n coefficients of t(X)                                                      */
    BBG_PP_T_1, BBG_PP_T_2, BBG_PP_T_3, BBG_PP_T_4, /* t_low, t_mid, t_high, t_higher: slices of n coefficients of t(X) */
    BBG_PP_LINEAR,                  /* r(X), the linearisation polynomial                                               */
    BBG_PP_OPENING, BBG_PP_SHIFTED_OPENING, /* W_zeta(X), W_zeta_omega(X)                                               */
    BBG_PP_Q_MIMC_COEFFICIENT, BBG_PP_Q_MIMC_SELECTOR, /* key polynomials of the MiMC flavour (set_key_poly / evaluate / linearise / round 6 ids) */
    BBG_PP_COUNT
};
enum bbg_prover_flavour { /* the widget list of round 4, in the prover's order */
    BBG_FLAVOUR_TURBO = 0,    /* width 4: permutation<4>, turbo arithmetic, fixed base, range, logic (turbo_composer.cpp:735-752) */
    BBG_FLAVOUR_STANDARD = 1, /* width 3: permutation<3>, arithmetic (standard_composer.cpp:562-582)                             */
    BBG_FLAVOUR_MIMC = 2      /* width 3: permutation<3>, MiMC, arithmetic (MiMCComposer::preprocess, mimc_composer.cpp:277-302)  */
};
/* program_width: 4 = TurboPLONK (ProverPermutationWidget<4> + turbo arithmetic / fixed base / range / logic widgets,
 * turbo_composer.cpp:735-752), 3 = StandardPLONK (ProverPermutationWidget<3> + arithmetic widget, standard_composer.cpp:562-582).
 * generators: 4 Montgomery Fr -- the small domain's coset generator g and fr::coset_generator(0..2) = k1, k2, k3.
 * srs must hold n points (n + 1 for StandardPLONK).  The srs and the context must outlive the handle. */
int bbg_prover_create(bbg_ctx* ctx, bbg_srs* srs, unsigned log2n, int program_width, const uint64_t* generators, bbg_prover** out);
/* The same with the flavour named (bbg_prover_create(4) = TURBO, (3) = STANDARD). */
int bbg_prover_create_flavour(bbg_ctx* ctx, bbg_srs* srs, unsigned log2n, int flavour, const uint64_t* generators, bbg_prover** out);
void bbg_prover_destroy(bbg_prover* p);
/* Per proving key.  id: BBG_QP_SIGMA_1 .. BBG_QP_LAGRANGE_1, BBG_PP_Q_MIMC_*.  form BBG_FORM_COEFF (n values; what proving_key::constraint_selectors /
 * permutation_selectors hold) is sufficient: bbg_prover_finalize_key derives sigma's Lagrange form, every 4n-coset form and L_1 on the
 * device.  A caller may instead hand over its own BBG_FORM_LAGRANGE (sigma) / BBG_FORM_COSET arrays.  The array is copied before
 * the call returns.  Re-registering a polynomial replaces it (call finalize again). */
int bbg_prover_set_key_poly(bbg_prover* p, int id, int form, const uint64_t* values);
int bbg_prover_finalize_key(bbg_prover* p);
/* Preamble + round 1 (prover.cpp:139-190, :66-84): wires_lagrange[k], k < program_width: the n values of wire k INCLUDING the blinding
 * rows n-4 .. n-2 the host has drawn.  Uploads them (kept for round 3), iffts to coefficient form (resident), commits.
 * commitments: program_width x 12 limbs, W_1 .. W_w as g1::element. */
int bbg_prover_round1(bbg_prover* p, const uint64_t* const* wires_lagrange, uint64_t* commitments);
/* Round 3 (permutation_widget_impl.hpp:48-312, prover.cpp:239-268): grand product z over the resident wires and sigmas, rows
 * n-3 .. n-1 <- blind[3][4], ifft, commitment Z, and the 4n-coset forms of z and the wires (resident, for round 4). */
int bbg_prover_round3(bbg_prover* p, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t* blind, uint64_t z_commitment[12]);
/* Round 4 (prover.cpp:275-363): the flavour's widgets in the prover's order, division by Z*_H, coset_ifft(4n) -> t(X) resident;
 * commitments T_1 .. T_w, 12 limbs each (the last one over n + 1 coefficients for StandardPLONK, prover.cpp:117-137). */
int bbg_prover_round4(bbg_prover* p, const uint64_t alpha[4], const uint64_t public_input_delta[4], uint64_t* t_commitments);
/* Round 5a (add_opening_evaluations_to_transcript, kate_commitment_scheme.cpp:362-420; quotient_large.evaluate, prover.cpp:397):
 * out[k] = P_ids[k](zeta), or P(zeta * w_n) where shifted[k] != 0 (shifted may be NULL); ids from bbg_quotient_poly /
 * bbg_prover_poly (coefficient forms; BBG_PP_QUOTIENT evaluates all 4n coefficients).  count <= 32. */
int bbg_prover_evaluate(bbg_prover* p, size_t count, const int* ids, const int* shifted, const uint64_t zeta[4], uint64_t* out);
/* Round 5b (compute_linear_contribution of every widget, prover.cpp:399-407): r(X) = sum_k scalars[k] * P_ids[k](X) (resident as
 * BBG_PP_LINEAR), r_eval = r(zeta). */
int bbg_prover_linearise(bbg_prover* p, size_t count, const int* ids, const uint64_t* scalars, const uint64_t zeta[4], uint64_t r_eval[4]);
/* Round 6 (KateCommitmentScheme::batch_open, kate_commitment_scheme.cpp:133-236): F = t_low + sum scalars_zeta[k] P_ids_zeta[k],
 * F' = sum scalars_omega[k] P_ids_omega[k]; W_zeta = (F - F(zeta)) / (X - zeta), W_zeta_omega likewise at zeta_omega; commitments
 * PI_Z and PI_Z_OMEGA.  t_high_top_scalar: StandardPLONK only (else NULL) -- the scalar zeta^(2n) of t's (3n+1)-th coefficient, which
 * enters F as its coefficient n (:196-205). */
int bbg_prover_round6(bbg_prover* p, size_t count_zeta, const int* ids_zeta, const uint64_t* scalars_zeta, size_t count_omega,
                      const int* ids_omega, const uint64_t* scalars_omega, const uint64_t zeta[4], const uint64_t zeta_omega[4],
                      const uint64_t* t_high_top_scalar, uint64_t pi_z[12], uint64_t pi_z_omega[12]);
/* Copies a resident polynomial back (tests; a host that wants the reference's post-proof state): id / form as above. */
int bbg_prover_read_poly(bbg_prover* p, int id, int form, uint64_t* out, size_t count);
/* Device bytes this handle owns (key polynomials in every form + the per-proof working set); what a key cache budgets with. */
int bbg_prover_device_bytes(const bbg_prover* p, size_t* bytes);

/* ---- several GPUs in one process: one MSM / one (coset) NTT spread over a group of contexts, one per device of the node
 *      (device indices may repeat: several contexts on one GPU, which is how the split is tested on a single-GPU box).
 *      MSM: point-range shards as in Pippenger::pippenger_unsafe(scalars, from, range) + g1_sum (pippenger.cpp:27-31, c_bind.cpp:31-46);
 *      only G x 96 bytes cross between GPUs.  NTT: residue classes + ONE all-to-all over peer copies (xGMI) + size-G DFTs.
 *      The process-per-GPU form of the same split (torch.distributed / RCCL) is aztec-2.0_amd/parallel.py. ---- */
typedef struct bbg_multi bbg_multi;
int bbg_multi_create(const int* devices, int count, bbg_multi** out);
void bbg_multi_destroy(bbg_multi* m);
int bbg_multi_count(const bbg_multi* m);
bbg_ctx* bbg_multi_ctx(bbg_multi* m, int k); /* context k of the group (for the *_device entry points on its GPU) */
int bbg_multi_sync(bbg_multi* m);
/* key "exchange": 0 = peer copies ordered with events (default), 1 = RCCL from C++ -- one communicator per context (ncclCommInitAll over
 * the group's devices, which must be distinct), ncclAllGather of the 96-byte MSM partials + the group sum, the NTT's all-to-all as grouped
 * ncclSend / ncclRecv on the contexts' streams.  librccl.so is loaded on first use.  Both back ends give identical results. */
int bbg_multi_set_option(bbg_multi* m, const char* key, long value);
/* Shards the SRS by point range: context g keeps points [g*ceil(n/G), (g+1)*ceil(n/G)) with their window tables resident.
 * points / stride_bytes as bbg_srs_register.  Replaces a previously registered SRS. */
int bbg_multi_srs_register(bbg_multi* m, const uint64_t* points, size_t n, size_t stride_bytes);
/* The hashed synthetic SRS of bbg_srs_synth_hashed(seed, n), every shard generated on its own GPU. */
int bbg_multi_srs_synth_hashed(bbg_multi* m, uint64_t seed, size_t n);
size_t bbg_multi_srs_num_points(const bbg_multi* m);
/* result = sum_{i<n} scalars[i] * P_{from+i}: every context runs the bucket MSM over its part of the range (scalars uploaded in
 * parallel, one host thread per GPU), the 96-byte partials are summed on context 0.  Same contract as bbg_msm. */
int bbg_multi_msm(bbg_multi* m, const uint64_t* scalars, size_t from, size_t n, uint64_t out_jacobian[12]);
/* op: BBG_FFT, BBG_IFFT, BBG_COSET_FFT or BBG_COSET_IFFT over the whole 2^log2n domain; G = 1, 2, 4 or 8 contexts.
 * Device-resident, asynchronous: d_shards[g] (on context g's GPU) holds the residue class a_{g + G j}, j < m = n / G; on return (after
 * bbg_multi_sync) shard r holds A[t*m + r*len + q] at index t*len + q (t < G, q < len = m / G), i.e. G contiguous runs of the
 * natural-order result.  One all-to-all of (G-1)/G^2 of the data per GPU, no host synchronisation between the phases. */
int bbg_multi_ntt_device(bbg_multi* m, void* const* d_shards, unsigned log2n, int op);
/* Host-buffer form, in place on coeffs[2^log2n] in natural order (residue classes gathered on the host by one thread per GPU; bound by
 * that gather and PCIe, not by the GPUs -- the resident form above is the one to build a multi-GPU prover on). */
int bbg_multi_ntt(bbg_multi* m, uint64_t* coeffs, unsigned log2n, int op);

/* ---- HBM budget: what a context holds on its device, by purpose.  Everything here is allocated on first use and then kept (the
 *      pippenger_runtime_state / evaluation_domain analogue: runtime_states.cpp:14-66, evaluation_domain.cpp:57-76); this call is how a
 *      host sees and bounds the total.  bbg_memory_trim releases what can be rebuilt on demand. ---- */
typedef struct bbg_memory_info {
    size_t srs_points;     /* plain SRS points not part of a window table (none today: window 0 of a table IS the plain points) */
    size_t srs_tables;     /* window tables of every live SRS of this context, all widths */
    size_t ntt_tables;     /* per-domain twiddle / coset tables (5 x 32n bytes per prepared size) */
    size_t msm_arena;      /* the MSM scratch arena (entries, sorted values, per-slot bucket sets) */
    size_t scratch;        /* NTT ping-pong buffer, host-entry staging, evaluation partials, widget constants */
    size_t prover_keys;    /* every live bbg_prover handle of this context (bbg_prover_device_bytes) */
    size_t total;          /* sum of the above */
    size_t device_total;   /* hipMemGetInfo: the device's memory ... */
    size_t device_free;    /* ... and what is free right now (other processes and the runtime included) */
    unsigned live_srs, live_provers, ntt_domains;
} bbg_memory_info;
int bbg_memory_report(bbg_ctx* ctx, bbg_memory_info* out);
/* Frees the rebuildable part after a device synchronisation: NTT tables of every size, the MSM arena, scratch and staging buffers, and --
 * with tables != 0 -- every window table except the one each SRS was registered with.  Returns the bytes released in *released (may be NULL). */
int bbg_memory_trim(bbg_ctx* ctx, int tables, size_t* released);

/* ---- tuning / introspection ---- */
/* key: "ntt_tile_log" (log2 elements per LDS tile, 9..12), "ntt_max_logr" (max radix per pass, 4..10),
 * "msm_async_reduce" (0/1, see bbg_join), "msm_window" (widest bucket window: 0 = automatic [8 bits up to 2^13 terms -- the small-circuit path: three launches, no sort --, 13 up to 2^14, 16 below 2^20, 19 at 2^20, 20 from 2^21, 22 from 2^23 -- bbg_msm_plan reports it],
 * or a compiled width 8 / 13 / 16 / 17 / 19 / 20 / 22; windows are BALANCED -- 255 bits split as evenly as the window count allows --; a width's window tables
 * are built the first time it is used on an SRS), "msm_sort" (1 = fused recode + two-level partition sort, default; 0 = recode + rocPRIM radix sort, only
 * in builds made with `make ROCPRIM_SORT=1`),
 * "msm_reduce_quad" (bit mask 0..15, default 14: reduce-phase stages with four lanes per EC operation -- bit 0 combine, 1 row/column sums, 2 bit
 * planes, 3 plane sum; 0 = the one-lane kernels), "msm_accumulate_quad" (1 = small MSMs accumulate with four threads per lane segment, default),
 * "msm_limbs29" (1 = the bucket accumulation's field arithmetic on 9 x 29-bit limbs, default; 0 = on 8 x 32-bit limbs, A/B),
 * "msm_acc_waves" (0 = automatic; N > 0 = lane segments per SIMD lane of the bucket accumulation, A/B), "msm_reduce_priority" (1 = low-priority reduce streams, default), "msm_upload_pieces" (1..4, default 1: pieces the host scalars of bbg_msm travel in),
 * "quotient_fuse" (1 = arithmetic + range + logic widgets of a chain in one pass, default),
 * "quotient_setup_plan" (1 = the challenge powers of a widget chain's set-up blocks are computed by the lanes of one wave side by side, default; 0 = one
 * lane's chain of products, A/B),
 * "poly_limbs29" (1 = bbg_poly_linear_combination*, bbg_poly_evaluate* and the prover's evaluations / linearisation / opening sums on 9 x 29-bit limbs
 * with four terms per reduction, default; 0 = on 8 x 32-bit limbs, A/B),
 * "prover_fused_divide" (1 = a bbg_prover's round 4 divides the quotient by Z*_H inside the coset iFFT's first load, from a per-point divisor table
 * kept per context -- 32 bytes per point of the 4n domain, counted under ntt_tables -- default; 0 = a pass of its own, A/B),
 * "prover_tail_window" (A/B: window width of the commitments that end prover rounds 4 and 6, 0 = automatic, default; measured: no width beats it),
 * "prover_ntt_batch" (1 = the wires' iFFTs of round 1 and their 4n coset forms go through ONE launch set each -- grid.y = wires -- for circuits up to
 * 2^17 gates, where a single transform has at most 128 tiles for 256 CUs, default; 0 = one launch set per wire, A/B),
 * "prover_early_cosets" (1 = a bbg_prover queues the wires' 4n coset forms behind round 1's last commitment, beside its reduce phase; 0 = in
 * front of round 3's grand product; -1 = the first from 2^18 gates, the second below, default),
 * "quotient_limbs29" (1 = the quotient widgets on lazily reduced 9 x 29-bit limbs with sums of products sharing one reduction, default; 0 = on 8 x 32-bit limbs, A/B),
 * "ntt_kernel" (2 = register-resident radix-8 passes, default; 1 = radix-2 in LDS),
 * "ntt_max_logr8" (6..11, max log-radix per radix-8 pass, default 10), "ntt_big_tile" (0 / 1 / 2: 4096-element tiles for 2^21 [default] / also 2^22),
 * "ntt_lds_planes" (2 = a pass keeps its tile in LDS between two radix-8 steps, 1 = the tile moves one 16-byte plane at a time through half the LDS with
 * three waves per SIMD, 0 = automatic [default]: 1 from 2^22), "prover_msm_batch" (0 .. BBG_MSM_BATCH_MAX, default 4: commitments of a bbg_prover round per
 * launch set; 0 / 1 = one launch set each).
 * Every value of every option gives bit-identical results; they exist for A/B measurements. */
int bbg_set_option(bbg_ctx* ctx, const char* key, long value);
/* Per-kernel timing with HIP events recorded on the launch stream.  Names: "msm_recode", "msm_sort", "msm_offsets",
 * "msm_accumulate", "msm_reduce", "ntt_pass", "quotient_widget".  enable(…, 1) clears previous samples. */
int bbg_profile_enable(bbg_ctx* ctx, int on);
int bbg_profile_get(bbg_ctx* ctx, const char* name, double* total_ms, size_t* launches);
/* Field-level self test entry used by tests: out[i] = a[i] (op) b[i] computed by the device field code.
 * which: 0 Fr, 1 Fq.  op: 0 mul, 1 add, 2 sub, 3 mul via the CIOS cross-check path, 4 from_montgomery, 5 to_montgomery,
 * 6/7 mul / CIOS mul without pre-reduction, 8 a*a - b*b and 9 (-a)(-b) - a(-b) through the fused two-product multiplier,
 * 10 / 11 the inverse of a (0 -> 0) by the binary extended Euclid of the set-up kernels, per lane / on the scalar unit (one wave per element). */
int bbg_field_op(bbg_ctx* ctx, int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* BBG_H */
