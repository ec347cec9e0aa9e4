// Quotient-polynomial pointwise kernels on the 4n coset domain: what a TurboPLONK prover's widgets do in
// ProverBase::execute_fourth_round (reference plonk/proof_system/prover/prover.cpp:304-319) --
//
//   ProverPermutationWidget<4,false>::compute_quotient_contribution     widgets/random_widgets/permutation_widget_impl.hpp:316-420
//   TransitionWidget<..., TurboArithmeticKernel / TurboFixedBaseKernel / TurboRangeKernel / TurboLogicKernel>
//       ::compute_quotient_contribution                                  widgets/transition_widgets/transition_widget.hpp:262-290
//       with the kernels of turbo_arithmetic_widget.hpp:17-139, turbo_fixed_base_widget.hpp, turbo_range_widget.hpp,
//       turbo_logic_widget.hpp
//
// Each is a pure pointwise function of ~10 of the key's "*_fft" arrays (values on the 4n coset, index i and the shifted
// index (i + 4) mod 4n = the next row of the circuit) and the transcript challenges; the permutation widget ASSIGNS the
// quotient array, the transition widgets ACCUMULATE into it, exactly as the reference.  The identities are restated
// here from the equations they implement (comments give the algebraic form); values stay coarsely reduced in [0, 2p).
// One thread per domain point (the permutation kernel walks 4 consecutive points to amortise w^i).  All reads are
// coalesced 32-byte elements: this is the one part of the prover that is genuinely HBM-streaming.
#include "bbg_internal.h"

#include <cstring>
#include "field.hip.h"
#include "ntt_consts.hip.h"

namespace bbg {

enum { QP_W1 = 0, QP_W2, QP_W3, QP_W4, QP_Z, QP_S1, QP_S2, QP_S3, QP_S4, QP_Q1, QP_Q2, QP_Q3, QP_Q4, QP_Q5, QP_QM, QP_QC,
       QP_QARITH, QP_QECC, QP_QRANGE, QP_QLOGIC, QP_L1, QP_COUNT,
       QP_QMIMC_C = QP_COUNT, QP_QMIMC_S, QP_EXT_COUNT }; // the MiMC widget's two selectors: the extended table (bbg.h)
static_assert(QP_COUNT == BBG_QP_COUNT && QP_EXT_COUNT == BBG_QP_EXT_COUNT && QP_QMIMC_C == BBG_QP_EXT_Q_MIMC_COEFFICIENT,
              "include/bbg.h and quotient.hip disagree on the polynomial table");
constexpr int WIDGET_COUNT = 8;

struct QuotientSetup {
    Fr ap[7];       // alpha_base * alpha^k
    Fr alpha, beta, gamma, delta;
    Fr alpha2, alpha3x2; // alpha^2, 2 alpha^3: the logic identity's Horner chain written as a sum (quotient29.hip.h)
    Fr alpha_base_sqr;
    Fr beta_g;      // beta * g (g = the small domain's coset generator): beta*g*w^i is the identity-permutation term
    Fr k1, k2, k3;  // coset generators of the wire columns 2..4 (fr::coset_generator(0..2))
    Fr one, c2, c3, c6, c7, c17, c81, c83;
    Fr alpha_out[WIDGET_COUNT]; // per widget: the alpha_base the next widget starts from
};
struct QuotientArgs {
    const Fr* p[QP_EXT_COUNT];
    Fr* quotient;
    uint32_t mask; // 4n - 1
    const QuotientSetup* s;
    const DomainConsts* dc; // large (4n) domain: root and its power-of-two table
};

__device__ __forceinline__ Fr fr_small(uint32_t k) // the Montgomery residue of a small integer: one product with R^2 (r4: was 32 doublings)
{
    Fr v = Fr::zero();
    v.v[0] = k;
    return fe_reduce_once(fe_to_mont(v));
}
// challenges arrive as Montgomery limbs from the host, BY VALUE as a kernel argument (no pageable-memory copy whose source the
// caller could reuse before it ran); everything derived from them is computed here (the product has no CPU field arithmetic).
// alpha_base_dev (optional): take alpha_base from device memory instead -- the alpha_out of the previous widget's set-up block,
// which chains the widgets of a round without a host round trip.
struct QuotientChallenges {
    Fr v[9]; // alpha_base, alpha, beta, gamma, delta, g, k1, k2, k3
};
// consts: a block of the same launch whose challenge-independent constants and alpha powers are already there (the chain kernel fills
// them once and copies: 17 products per further block instead of 30)
__device__ __forceinline__ void quotient_setup_one(QuotientSetup* s, const QuotientChallenges& in, const Fr& alpha_base, const QuotientSetup* consts = nullptr)
{
    const Fr alpha = in.v[1];
    s->alpha = alpha;
    if (consts) {
        s->alpha2 = consts->alpha2;
        s->alpha3x2 = consts->alpha3x2;
        s->beta_g = consts->beta_g;
    } else {
        s->alpha2 = fe_sqr(alpha);
        const Fr a3 = fe_mul(s->alpha2, alpha);
        s->alpha3x2 = fe_add(a3, a3);
        s->beta_g = fe_mul(in.v[2], in.v[5]);
    }
    s->beta = in.v[2];
    s->gamma = in.v[3];
    s->delta = in.v[4];
    s->k1 = in.v[6];
    s->k2 = in.v[7];
    s->k3 = in.v[8];
    Fr a = alpha_base;
    for (int k = 0; k < 7; k++) {
        s->ap[k] = a;
        a = fe_mul(a, alpha);
    }
    s->alpha_base_sqr = fe_sqr(alpha_base);
    s->one = Fr::one();
    if (consts) {
        s->c2 = consts->c2; s->c3 = consts->c3; s->c6 = consts->c6; s->c7 = consts->c7;
        s->c17 = consts->c17; s->c81 = consts->c81; s->c83 = consts->c83;
    } else {
        s->c2 = fr_small(2);
        s->c3 = fr_small(3);
        s->c6 = fr_small(6);
        s->c7 = fr_small(7);
        s->c17 = fr_small(17);
        s->c81 = fr_small(81);
        s->c83 = fr_small(83);
    }
    // update_alpha (transition_widget.hpp:88-94): alpha_powers[num_independent_relations - 1] * alpha
    s->alpha_out[0] = fe_sqr(s->alpha_base_sqr);     // permutation: alpha_base^4 (permutation_widget_impl.hpp:419)
    s->alpha_out[1] = fe_mul(s->ap[1], alpha);       // arithmetic: 2 relations
    s->alpha_out[2] = fe_mul(s->ap[6], alpha);       // fixed base: 7
    s->alpha_out[3] = fe_mul(s->ap[3], alpha);       // range: 4
    s->alpha_out[4] = fe_mul(s->ap[3], alpha);       // logic: 4
    s->alpha_out[5] = s->alpha_out[0];               // permutation, 3 wires
    s->alpha_out[6] = s->ap[1];                      // standard arithmetic: 1 relation
    s->alpha_out[7] = s->ap[2];                      // MiMC: 2 relations
}
__global__ void k_quotient_setup(QuotientSetup* s, QuotientChallenges in, const Fr* alpha_base_dev)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    quotient_setup_one(s, in, alpha_base_dev ? fe_load<FrP>(alpha_base_dev) : in.v[0]);
}
// The set-up blocks of a whole widget chain in ONE launch: block w starts from the alpha_out block w - 1 produced for its widget (r4: a
// launch per widget was a chain of five 31-us single-lane kernels, 0.16 ms of every proof).
struct WidgetChain {
    int count;
    int widget[8];
};
// The same blocks with the lanes of one wave working side by side (r5).  Every value of the chain is alpha_base^(4^j) * alpha^e with (j, e)
// known to the HOST from the widget list alone -- a permutation widget hands on base^4, the others base * alpha^r (update_alpha,
// transition_widget.hpp:88-94) -- so block w's eight powers base_w * alpha^k are eight independent lanes: a shared table alpha^(2^b), one
// power from it, one product.  The serial kernel above is a chain of ~ 90 dependent products on one lane (59 us: 3 % of a 2^12-gate proof);
// here the longest chain is the table's squarings + one lane's power + two squarings: ~ 14 products.
struct WidgetPlan {
    int count;
    uint32_t widgets; // 4 bits per block: the widget
    uint32_t c4;      // 4 bits per block: j, the block's base is alpha_base^(4^j) * alpha^e
    uint64_t e[2];    // 16 bits per block: e
    int table_bits;   // alpha^(2^b) for b < table_bits (>= 2) covers every e + 7
    int max_c4;
};
constexpr int PLAN_MAX_C4 = 4, PLAN_MAX_BITS = 16;
__global__ void __launch_bounds__(64) k_quotient_setup_plan(QuotientSetup* setups, QuotientChallenges in, WidgetPlan plan)
{
    __shared__ Fr T[PLAN_MAX_BITS];  // alpha^(2^b)
    __shared__ Fr C[PLAN_MAX_C4 + 1]; // alpha_base^(4^j)
    const int lane = threadIdx.x;
    const Fr alpha = in.v[1];
    { // the same chains on every lane (uniform: no divergence), stored once
        Fr t = alpha;
        for (int b = 0; b < plan.table_bits; b++) {
            if (lane == 0) T[b] = t;
            t = fe_sqr(t);
        }
        Fr c = in.v[0];
        for (int j = 0; j <= plan.max_c4; j++) {
            if (lane == 0) C[j] = c;
            if (j < plan.max_c4) c = fe_sqr(fe_sqr(c));
        }
    }
    __syncthreads();
    const int w = lane >> 3, k = lane & 7;
    if (w >= plan.count) return;
    QuotientSetup* s = setups + w;
    const uint32_t e = (uint32_t)((w < 4 ? plan.e[0] >> (16 * w) : plan.e[1] >> (16 * (w - 4))) & 0xffff);
    const Fr v = fe_mul(C[(plan.c4 >> (4 * w)) & 15], pow_from_table(T, (uint64_t)(e + k))); // base_w * alpha^k
    if (k < 7) s->ap[k] = v;
    switch (k) {
    case 0: {
        const Fr sq = fe_sqr(v);
        s->alpha_base_sqr = sq;
        const Fr q = fe_sqr(sq); // permutation: alpha_base^4 (permutation_widget_impl.hpp:419)
        s->alpha_out[0] = q;
        s->alpha_out[5] = q;
        break;
    }
    case 1: {
        s->alpha_out[6] = v; // standard arithmetic: 1 relation
        s->alpha2 = T[1];
        const Fr a3 = fe_mul(T[1], alpha);
        s->alpha3x2 = fe_add(a3, a3);
        break;
    }
    case 2:
        s->alpha_out[1] = v; // arithmetic: 2 relations
        s->alpha_out[7] = v; // MiMC: 2
        s->beta_g = fe_mul(in.v[2], in.v[5]);
        break;
    case 3:
        s->alpha = alpha;
        s->beta = in.v[2];
        s->gamma = in.v[3];
        s->delta = in.v[4];
        break;
    case 4:
        s->alpha_out[3] = v; // range: 4 relations
        s->alpha_out[4] = v; // logic: 4
        break;
    case 5:
        s->k1 = in.v[6];
        s->k2 = in.v[7];
        s->k3 = in.v[8];
        s->one = Fr::one();
        break;
    case 6:
        s->c2 = fr_small(2);
        s->c3 = fr_small(3);
        s->c6 = fr_small(6);
        s->c7 = fr_small(7);
        s->c17 = fr_small(17);
        s->c81 = fr_small(81);
        s->c83 = fr_small(83);
        break;
    default:
        s->alpha_out[2] = v; // fixed base: 7 relations
        break;
    }
}
// (j, e) of every block of a widget list; false when a value leaves the plan's fields (the serial kernel then)
static bool widget_plan(const int* widgets, int count, WidgetPlan& plan)
{
    static const int RELATIONS[WIDGET_COUNT] = { 0, 2, 7, 4, 4, 0, 1, 2 }; // alpha_out = base * alpha^r; widgets 0 and 5 hand on base^4
    memset(&plan, 0, sizeof(plan));
    plan.count = count;
    uint64_t e = 0;
    int j = 0, bits = 2;
    for (int w = 0; w < count; w++) {
        if (j > PLAN_MAX_C4 || e + 7 >= ((uint64_t)1 << PLAN_MAX_BITS)) return false;
        plan.widgets |= (uint32_t)widgets[w] << (4 * w);
        plan.c4 |= (uint32_t)j << (4 * w);
        plan.e[w >> 2] |= e << (16 * (w & 3));
        while ((e + 7) >> bits) bits++;
        if (j > plan.max_c4) plan.max_c4 = j;
        if (widgets[w] == 0 || widgets[w] == 5) {
            j++;
            e *= 4;
        } else {
            e += (uint64_t)RELATIONS[widgets[w]];
        }
    }
    plan.table_bits = bits;
    return count <= 8;
}
__global__ void k_quotient_setup_chain(QuotientSetup* setups, QuotientChallenges in, WidgetChain chain)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr alpha_base = in.v[0];
    for (int w = 0; w < chain.count; w++) {
        quotient_setup_one(setups + w, in, alpha_base, w ? setups : nullptr);
        alpha_base = setups[w].alpha_out[chain.widget[w]];
    }
}

#define QLOAD(id, idx) fe_load<FrP>(a.p[id] + (idx))
__device__ __forceinline__ Fr x4(const Fr& v)
{
    const Fr d = fe_add(v, v);
    return fe_add(d, d);
}
// D (D - 1) (D - 2) (D - 3): vanishes exactly on the base-4 digits
__device__ __forceinline__ Fr quad_check(const Fr& d, const QuotientSetup& s)
{
    Fr t = fe_sub(fe_sqr(d), d);
    t = fe_mul(t, fe_sub(d, s.c2));
    return fe_mul(t, fe_sub(d, s.c3));
}

// ---- permutation argument, 4 wire columns, identity permutation given implicitly by X, k1 X, k2 X, k3 X:
//   q = alpha_base * [ z(X) prod_k (w_k + gamma + beta K_k X) - z(wX) prod_k (w_k + gamma + beta sigma_k)
//                      + (z(wX) - delta) alpha_base L_{n-4}(X) + (z(X) - 1) alpha_base^2 L_1(X) ]
// L_{n-4} is read from the L_1 table at the shifted index i + 4 + 4*4 (4 roots cut out of the vanishing polynomial).
constexpr int PERM_CH = 4;
template <int WIDTH> __global__ void __launch_bounds__(256) k_quotient_permutation(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * PERM_CH;
    if (i0 > a.mask) return;
    Fr rb = fe_mul(s.beta_g, pow_from_table(a.dc->pow2_root, (uint64_t)i0)); // beta * g * w^i
    const Fr root = a.dc->root;
#pragma unroll 1
    for (int e = 0; e < PERM_CH; e++) {
        const uint32_t i = i0 + e, ish = (i + 4) & a.mask;
        Fr wpg = fe_add(QLOAD(QP_W1, i), s.gamma);
        Fr num = fe_add(wpg, rb);
        Fr den = fe_add(wpg, fe_mul(QLOAD(QP_S1, i), s.beta));
        wpg = fe_add(QLOAD(QP_W2, i), s.gamma);
        num = fe_mul(num, fe_add(wpg, fe_mul(s.k1, rb)));
        den = fe_mul(den, fe_add(wpg, fe_mul(QLOAD(QP_S2, i), s.beta)));
        wpg = fe_add(QLOAD(QP_W3, i), s.gamma);
        num = fe_mul(num, fe_add(wpg, fe_mul(s.k2, rb)));
        den = fe_mul(den, fe_add(wpg, fe_mul(QLOAD(QP_S3, i), s.beta)));
        if constexpr (WIDTH == 4) { // StandardPLONK has three wire columns (ProverPermutationWidget<3,false>)
            wpg = fe_add(QLOAD(QP_W4, i), s.gamma);
            num = fe_mul(num, fe_add(wpg, fe_mul(s.k3, rb)));
            den = fe_mul(den, fe_add(wpg, fe_mul(QLOAD(QP_S4, i), s.beta)));
        }
        const Fr z = QLOAD(QP_Z, i), zw = QLOAD(QP_Z, ish);
        num = fe_mul(num, z);
        den = fe_mul(den, zw);
        Fr t = fe_mul(fe_mul(fe_sub(zw, s.delta), s.ap[0]), QLOAD(QP_L1, (i + 4 + 16) & a.mask));
        num = fe_add(num, t);
        t = fe_mul(fe_mul(fe_sub(z, s.one), s.alpha_base_sqr), QLOAD(QP_L1, i));
        num = fe_add(num, t);
        fe_store<FrP>(a.quotient + i, fe_mul(fe_sub(num, den), s.ap[0]));
        rb = fe_mul(rb, root);
    }
}

// ---- StandardPLONK arithmetic gate (arithmetic_widget.hpp): alpha_base (q_m w1 w2 + q_1 w1 + q_2 w2 + q_3 w3 + q_c)
__global__ void __launch_bounds__(256) k_quotient_standard_arith(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i);
    Fr gate = fe_mul(fe_mul(w1, w2), QLOAD(QP_QM, i));
    gate = fe_add(gate, fe_mul(w1, QLOAD(QP_Q1, i)));
    gate = fe_add(gate, fe_mul(w2, QLOAD(QP_Q2, i)));
    gate = fe_add(gate, fe_mul(QLOAD(QP_W3, i), QLOAD(QP_Q3, i)));
    gate = fe_add(gate, QLOAD(QP_QC, i));
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(gate, s.ap[0])));
}

// ---- MiMC round gate (mimc_widget.hpp:17-52), T = w1 + w3 + q_mimc_coefficient:
//   q_mimc_selector alpha_base [ (T^3 - w2) + alpha (w2^2 T - w3(wX)) ]         (w2 = the cube, w3(wX) = the next round's input)
__global__ void __launch_bounds__(256) k_quotient_mimc(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w2 = QLOAD(QP_W2, i);
    const Fr t = fe_add(fe_add(QLOAD(QP_W1, i), QLOAD(QP_W3, i)), QLOAD(QP_QMIMC_C, i));
    const Fr cube = fe_sub(fe_mul(fe_sqr(t), t), w2);
    const Fr out = fe_sub(fe_mul(fe_sqr(w2), t), QLOAD(QP_W3, (i + 4) & a.mask));
    const Fr id = fe_add(fe_mul(cube, s.ap[0]), fe_mul(out, s.ap[1]));
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(id, QLOAD(QP_QMIMC_S, i))));
}

// ---- turbo arithmetic gate:
//   alpha_base * [ q_arith (q_m w1 w2 + q_1 w1 + q_2 w2 + q_3 w3 + q_4 w4 + q_c) + alpha q_5 q_arith w4 (w4 - 1)(w4 - 2) ]
//   + alpha_base (q_arith^2 - q_arith) d (9 d - 2 d^2 - 7),   d = w3 - 4 w4   (high-bit extraction, active when q_arith = 2)
__global__ void __launch_bounds__(256) k_quotient_turbo_arith(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const Fr qa = QLOAD(QP_QARITH, i);
    Fr gate = fe_mul(fe_mul(w1, w2), QLOAD(QP_QM, i));
    gate = fe_add(gate, fe_mul(w1, QLOAD(QP_Q1, i)));
    gate = fe_add(gate, fe_mul(w2, QLOAD(QP_Q2, i)));
    gate = fe_add(gate, fe_mul(w3, QLOAD(QP_Q3, i)));
    gate = fe_add(gate, fe_mul(w4, QLOAD(QP_Q4, i)));
    gate = fe_add(gate, QLOAD(QP_QC, i));
    Fr t = fe_mul(fe_sub(fe_sqr(w4), w4), fe_sub(w4, s.c2));
    t = fe_mul(fe_mul(t, s.alpha), QLOAD(QP_Q5, i));
    gate = fe_mul(fe_add(gate, t), qa);
    const Fr d = fe_sub(w3, x4(w4));
    const Fr d2 = fe_sqr(d);
    Fr h = fe_add(x4(d), x4(d));          // 8 d
    h = fe_sub(fe_add(h, d), fe_add(d2, d2)); // 9 d - 2 d^2
    h = fe_mul(fe_sub(h, s.c7), d);
    h = fe_mul(h, fe_sub(fe_sqr(qa), qa));
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(fe_add(gate, h), s.ap[0])));
}

// ---- fixed-base scalar multiplication ladder over Grumpkin (y^2 = x^3 - 17); ap[k] = alpha_base alpha^k.  Two kernels -- the
// selector-weighted terms and the gate identities -- because one kernel holding 8 wire values, 8 selectors and 7 alpha
// powers needs 250 VGPRs (2 waves per SIMD, 5.0 ms at 4n = 2^22); split, each half re-reads the wires and stays near 128.
//   delta = w4' - 4 w4 is the next quad, in {-3, -1, 1, 3}
__global__ void __launch_bounds__(256) k_quotient_turbo_fixed_base_linear(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr qe = QLOAD(QP_QECC, i);
    const Fr w4 = QLOAD(QP_W4, i);
    const Fr delta = fe_sub(QLOAD(QP_W4, ish), x4(w4));
    const Fr w3n = QLOAD(QP_W3, ish);
    Fr lin = fe_mul(fe_mul(fe_sqr(delta), s.ap[1]), QLOAD(QP_Q1, i));                       // q_1: x-coordinate lookup, delta^2 term
    lin = fe_add(lin, fe_mul(s.ap[1], QLOAD(QP_Q2, i)));                                     // q_2: constant term
    {
        const Fr w1 = QLOAD(QP_W1, i);
        Fr t3 = fe_mul(fe_mul(fe_mul(fe_sub(QLOAD(QP_W1, ish), w1), delta), w3n), s.ap[3]);
        const Fr u = fe_mul(fe_mul(fe_mul(delta, w3n), QLOAD(QP_W2, i)), s.ap[2]);
        t3 = fe_add(t3, fe_add(u, u));
        lin = fe_add(lin, fe_mul(t3, QLOAD(QP_Q3, i)));                                      // q_3: y-coordinate lookup
    }
    {
        const Fr w3 = QLOAD(QP_W3, i), qc = QLOAD(QP_QC, i);                                 // q_4, q_5, q_m: initialisation row
        Fr init = fe_mul(fe_mul(w3, s.ap[5]), QLOAD(QP_Q4, i));
        init = fe_add(init, fe_mul(fe_mul(fe_sub(s.one, w4), s.ap[5]), QLOAD(QP_Q5, i)));
        init = fe_add(init, fe_mul(fe_mul(w3, s.ap[6]), QLOAD(QP_QM, i)));
        lin = fe_add(lin, fe_mul(init, qc));
    }
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(lin, qe)));
}
__global__ void __launch_bounds__(256) k_quotient_turbo_fixed_base_gate(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const Fr w1n = QLOAD(QP_W1, ish), w3n = QLOAD(QP_W3, ish);
    const Fr qc = QLOAD(QP_QC, i), qe = QLOAD(QP_QECC, i);
    const Fr delta = fe_sub(QLOAD(QP_W4, ish), x4(w4));
    Fr acc = fe_mul(fe_mul(fe_add(delta, s.one), fe_add(delta, s.c3)), fe_mul(fe_sub(delta, s.one), fe_sub(delta, s.c3)));
    Fr gate = fe_sub(fe_mul(acc, s.ap[0]), fe_mul(w3n, s.ap[1]));                            // accumulator + x_alpha identities
    const Fr dx = fe_sub(w3n, w1);
    {
        Fr xacc = fe_mul(fe_add(fe_add(w1n, w1), w3n), fe_sqr(dx));                          // (x3 + x1 + x_alpha)(x_alpha - x1)^2
        const Fr rhs = fe_sub(fe_add(fe_mul(fe_sqr(w3n), w3n), fe_sqr(w2)), s.c17);          // x_alpha^3 + y1^2 - 17
        Fr two_dy = fe_mul(fe_mul(delta, w2), qe);
        two_dy = fe_add(two_dy, two_dy);
        gate = fe_add(gate, fe_mul(fe_add(fe_sub(xacc, rhs), two_dy), s.ap[2]));
    }
    {
        Fr yacc = fe_mul(fe_add(QLOAD(QP_W2, ish), w2), dx);
        yacc = fe_add(yacc, fe_mul(fe_sub(w1, w1n), fe_sub(w2, fe_mul(qe, delta))));
        gate = fe_add(gate, fe_mul(yacc, s.ap[3]));
    }
    {
        const Fr w4m1 = fe_sub(w4, s.one);
        Fr init = fe_mul(fe_mul(w4m1, fe_sub(w4m1, w3)), s.ap[4]);                           // accumulator / x / y initialisation
        init = fe_sub(init, fe_mul(fe_mul(w1, w3), s.ap[5]));
        init = fe_add(init, fe_mul(fe_sub(fe_mul(fe_sub(s.one, w4), qc), fe_mul(w2, w3)), s.ap[6]));
        gate = fe_add(gate, fe_mul(init, qc));
    }
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(gate, qe)));
}

// ---- base-4 range check ("raster scan" over the 4 wire columns and the next row's 4th column):
//   q_range * sum_k ap[k] f(D_k),  f(D) = D (D-1)(D-2)(D-3),  D_1 = w3 - 4 w4, D_2 = w2 - 4 w3, D_3 = w1 - 4 w2, D_4 = w4' - 4 w1
__global__ void __launch_bounds__(256) k_quotient_turbo_range(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const Fr w4n = QLOAD(QP_W4, (i + 4) & a.mask);
    Fr sum = fe_mul(quad_check(fe_sub(w3, x4(w4)), s), s.ap[0]);
    sum = fe_add(sum, fe_mul(quad_check(fe_sub(w2, x4(w3)), s), s.ap[1]));
    sum = fe_add(sum, fe_mul(quad_check(fe_sub(w1, x4(w2)), s), s.ap[2]));
    sum = fe_add(sum, fe_mul(quad_check(fe_sub(w4n, x4(w1)), s), s.ap[3]));
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(sum, QLOAD(QP_QRANGE, i))));
}

// ---- AND / XOR on base-4 quads a, b (c = the output quad, w3 = a*b):
//   q_logic alpha_base [ ((2 (a b - w3) alpha + f(a)) alpha + f(b)) alpha + 3 (a + b + c) - 2 E + q_c (9 c - 3 (a + b)) ]
//   E = w3 ( w3 (4 w3 - 18 (a + b) + 81) + 18 (a^2 + b^2) - 81 (a + b) + 83 )
//   a = w1' - 4 w1, b = w2' - 4 w2, c = w4' - 4 w4
__global__ void __launch_bounds__(256) k_quotient_turbo_logic(QuotientArgs a)
{
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w3 = QLOAD(QP_W3, i);
    const Fr qa = fe_sub(QLOAD(QP_W1, ish), x4(QLOAD(QP_W1, i)));
    const Fr qb = fe_sub(QLOAD(QP_W2, ish), x4(QLOAD(QP_W2, i)));
    const Fr qcq = fe_sub(QLOAD(QP_W4, ish), x4(QLOAD(QP_W4, i)));
    const Fr sum = fe_add(qa, qb);
    Fr id = fe_sub(fe_mul(qa, qb), w3);
    id = fe_mul(fe_add(id, id), s.alpha);
    id = fe_mul(fe_add(id, quad_check(qa, s)), s.alpha);
    id = fe_mul(fe_add(id, quad_check(qb, s)), s.alpha);
    const Fr sum3 = fe_add(fe_add(sum, sum), sum), sum9 = fe_add(fe_add(sum3, sum3), sum3);
    const Fr sum18 = fe_add(sum9, sum9);
    Fr sum81 = x4(sum18);               // 72
    sum81 = fe_add(sum81, sum9);        // 81
    const Fr sq = fe_add(fe_sqr(qa), fe_sqr(qb));
    const Fr sq3 = fe_add(fe_add(sq, sq), sq), sq9 = fe_add(fe_add(sq3, sq3), sq3);
    const Fr sq18 = fe_add(sq9, sq9);
    Fr e = fe_add(fe_sub(x4(w3), sum18), s.c81);
    e = fe_mul(e, w3);
    e = fe_add(e, fe_add(fe_sub(sq18, sum81), s.c83));
    e = fe_mul(e, w3);
    const Fr c3 = fe_add(fe_add(qcq, qcq), qcq), c9 = fe_add(fe_add(c3, c3), c3);
    Fr tail = fe_sub(fe_add(c3, sum3), fe_add(e, e));
    tail = fe_add(tail, fe_mul(fe_sub(c9, sum3), QLOAD(QP_QC, i)));
    id = fe_mul(fe_add(id, tail), s.ap[0]);
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, fe_mul(id, QLOAD(QP_QLOGIC, i))));
}

// ---- arithmetic + range + logic in ONE pass over the wires (TurboPLONK round 4): the three cheapest transition widgets are memory-bound
// on their own (each reads the four wire columns -- range and logic the shifted rows as well -- and read-modify-writes the quotient:
// 0.57 + 0.66 + 0.64 ms at 4n = 2^22, profiles/r02_prover_kernel_stats_v3.txt); fused, the wires, their shifted rows, q_c and the
// quotient are touched once.  Same identities, same alpha powers: each part takes its own set-up block (the alpha_base chain of
// execute_fourth_round, prover.cpp:304-319, runs over the set-up kernels only), and field addition does not care about the order.
__global__ void __launch_bounds__(256) k_quotient_turbo_arith_range_logic(QuotientArgs a, const QuotientSetup* s_range, const QuotientSetup* s_logic)
{
    const QuotientSetup& s = *a.s; // arithmetic widget's block; alpha and the small constants are the same in all three
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const Fr w4n = QLOAD(QP_W4, ish);
    const Fr qc = QLOAD(QP_QC, i);
    Fr total;
    { // arithmetic (k_quotient_turbo_arith)
        const Fr qa = QLOAD(QP_QARITH, i);
        Fr gate = fe_mul(fe_mul(w1, w2), QLOAD(QP_QM, i));
        gate = fe_add(gate, fe_mul(w1, QLOAD(QP_Q1, i)));
        gate = fe_add(gate, fe_mul(w2, QLOAD(QP_Q2, i)));
        gate = fe_add(gate, fe_mul(w3, QLOAD(QP_Q3, i)));
        gate = fe_add(gate, fe_mul(w4, QLOAD(QP_Q4, i)));
        gate = fe_add(gate, qc);
        Fr t = fe_mul(fe_sub(fe_sqr(w4), w4), fe_sub(w4, s.c2));
        t = fe_mul(fe_mul(t, s.alpha), QLOAD(QP_Q5, i));
        gate = fe_mul(fe_add(gate, t), qa);
        const Fr d = fe_sub(w3, x4(w4));
        const Fr d2 = fe_sqr(d);
        Fr h = fe_add(x4(d), x4(d));
        h = fe_sub(fe_add(h, d), fe_add(d2, d2));
        h = fe_mul(fe_sub(h, s.c7), d);
        h = fe_mul(h, fe_sub(fe_sqr(qa), qa));
        total = fe_mul(fe_add(gate, h), s.ap[0]);
    }
    { // range (k_quotient_turbo_range)
        const QuotientSetup& r = *s_range;
        Fr sum = fe_mul(quad_check(fe_sub(w3, x4(w4)), s), r.ap[0]);
        sum = fe_add(sum, fe_mul(quad_check(fe_sub(w2, x4(w3)), s), r.ap[1]));
        sum = fe_add(sum, fe_mul(quad_check(fe_sub(w1, x4(w2)), s), r.ap[2]));
        sum = fe_add(sum, fe_mul(quad_check(fe_sub(w4n, x4(w1)), s), r.ap[3]));
        total = fe_add(total, fe_mul(sum, QLOAD(QP_QRANGE, i)));
    }
    { // logic (k_quotient_turbo_logic)
        const QuotientSetup& l = *s_logic;
        const Fr qa = fe_sub(QLOAD(QP_W1, ish), x4(w1));
        const Fr qb = fe_sub(QLOAD(QP_W2, ish), x4(w2));
        const Fr qcq = fe_sub(w4n, x4(w4));
        const Fr sum = fe_add(qa, qb);
        Fr id = fe_sub(fe_mul(qa, qb), w3);
        id = fe_mul(fe_add(id, id), s.alpha);
        id = fe_mul(fe_add(id, quad_check(qa, s)), s.alpha);
        id = fe_mul(fe_add(id, quad_check(qb, s)), s.alpha);
        const Fr sum3 = fe_add(fe_add(sum, sum), sum), sum9 = fe_add(fe_add(sum3, sum3), sum3);
        const Fr sum18 = fe_add(sum9, sum9);
        Fr sum81 = x4(sum18);
        sum81 = fe_add(sum81, sum9);
        const Fr sq = fe_add(fe_sqr(qa), fe_sqr(qb));
        const Fr sq3 = fe_add(fe_add(sq, sq), sq), sq9 = fe_add(fe_add(sq3, sq3), sq3);
        const Fr sq18 = fe_add(sq9, sq9);
        Fr e = fe_add(fe_sub(x4(w3), sum18), s.c81);
        e = fe_mul(e, w3);
        e = fe_add(e, fe_add(fe_sub(sq18, sum81), s.c83));
        e = fe_mul(e, w3);
        const Fr c3 = fe_add(fe_add(qcq, qcq), qcq), c9 = fe_add(fe_add(c3, c3), c3);
        Fr tail = fe_sub(fe_add(c3, sum3), fe_add(e, e));
        tail = fe_add(tail, fe_mul(fe_sub(c9, sum3), qc));
        id = fe_mul(fe_add(id, tail), l.ap[0]);
        total = fe_add(total, fe_mul(id, QLOAD(QP_QLOGIC, i)));
    }
    const Fr q = fe_load<FrP>(a.quotient + i);
    fe_store<FrP>(a.quotient + i, fe_add(q, total));
}

} // namespace bbg
#include "quotient29.hip.h"
namespace bbg {

// ---------------------------------------------------------------------------------------------- permutation grand product
// z of ProverPermutationWidget<W,false>::compute_round_commitments (permutation_widget_impl.hpp:48-268, steps 1-3; the blinding of
// the last rows and the ifft stay with the caller):
//   z[0] = 1,   z[j+1] = prod_{i <= j} N_i / D_i,   N_i = prod_k (w_k[i] + gamma + beta K_k w^i),  D_i = prod_k (w_k[i] + gamma + beta sigma_k[i])
// The reference runs 2W serial prefix products and one batched inversion per thread.  Round 1 of this repo inverted once per 16 rows
// (65 536 Fermat chains at n = 2^20: 0.83 ms).  Here the whole polynomial costs ONE inversion:
//   z[j+1] = PN_j * SD_{j+1} * (prod_all D)^-1,     PN_j = prod_{i <= j} N_i (prefix),   SD_{j+1} = prod_{i > j} D_i (suffix)
//   k_gp_terms  : block = 1024 rows (256 threads x 4): N_i, D_i; in-block inclusive prefix of N -> z[j+1], in-block exclusive suffix of D
//                 -> sd[j]; block totals of both
//   k_gp_blocks : one block: exclusive prefix of the N block totals, exclusive suffix of the D block totals, the grand total of D
//   k_gp_invert : one lane: (prod_all D)^-1  (a dependency chain -- 0.32 ms as a^(p-2), r4: binary extended Euclid --: the prover overlaps it with
//                 the wires' coset FFTs)
//   k_gp_apply  : z[j+1] *= (prefix of earlier blocks) * sd[j] * (suffix of later blocks) * inverse
// Rows per thread of k_gp_terms: 4 from 2^18 rows up (fewer block totals, shorter scans per row), 1 below -- a 2^16-row grand product with four
// rows per thread is 256 waves for 1 024 SIMDs, each with a chain of 4 x 15 + 24 dependent products (r5)
static int gp_rows_per_thread(size_t n) { return n >= ((size_t)1 << 18) ? 4 : 1; }
__device__ inline Fr fr_inverse(const Fr& a) { return fe_inverse_gcd<FrP, true>(a); } // field.hip.h: binary extended Euclid on the scalar unit (one lane, one input), canonical result
struct GpArgs {
    const Fr* w[4];
    const Fr* sigma[4];
    Fr* z;      // n entries
    Fr* sd;     // n entries: in-block exclusive suffix products of D
    Fr* bt;     // block tables: [0, B) N totals -> exclusive prefixes; [B, 2B) D totals -> exclusive suffixes; [2B] total of D; [2B+1] its inverse
    size_t n, nblocks;
    size_t block_rows; // 256 x rows per thread
    const QuotientSetup* s; // beta, gamma, k1..k3
    const DomainConsts* dc; // small (n) domain
};
__global__ void k_gp_setup(QuotientSetup* s, QuotientChallenges in)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    s->beta = in.v[2];
    s->gamma = in.v[3];
    s->k1 = in.v[6];
    s->k2 = in.v[7];
    s->k3 = in.v[8];
}
template <int WIDTH, int GP_E> __global__ void __launch_bounds__(256) k_gp_terms(GpArgs a)
{
    __shared__ Fr smn[256], smd[256];
    const int tid = threadIdx.x;
    const size_t j0 = (size_t)blockIdx.x * (256 * GP_E) + (size_t)tid * GP_E;
    const QuotientSetup& s = *a.s;
    Fr num[GP_E], den[GP_E];
    Fr rb = fe_mul(s.beta, pow_from_table(a.dc->pow2_root, (uint64_t)j0)); // beta * w^j
    const Fr root = a.dc->root;
#pragma unroll
    for (int e = 0; e < GP_E; e++) {
        const size_t j = j0 + e;
        if (j < a.n) {
            Fr wpg = fe_add(fe_load<FrP>(a.w[0] + j), s.gamma);
            Fr n_ = fe_add(wpg, rb);
            Fr d_ = fe_add(wpg, fe_mul(fe_load<FrP>(a.sigma[0] + j), s.beta));
            wpg = fe_add(fe_load<FrP>(a.w[1] + j), s.gamma);
            n_ = fe_mul(n_, fe_add(wpg, fe_mul(s.k1, rb)));
            d_ = fe_mul(d_, fe_add(wpg, fe_mul(fe_load<FrP>(a.sigma[1] + j), s.beta)));
            wpg = fe_add(fe_load<FrP>(a.w[2] + j), s.gamma);
            n_ = fe_mul(n_, fe_add(wpg, fe_mul(s.k2, rb)));
            d_ = fe_mul(d_, fe_add(wpg, fe_mul(fe_load<FrP>(a.sigma[2] + j), s.beta)));
            if constexpr (WIDTH == 4) { // StandardPLONK: three wire columns (ProverPermutationWidget<3,false>)
                wpg = fe_add(fe_load<FrP>(a.w[3] + j), s.gamma);
                n_ = fe_mul(n_, fe_add(wpg, fe_mul(s.k3, rb)));
                d_ = fe_mul(d_, fe_add(wpg, fe_mul(fe_load<FrP>(a.sigma[3] + j), s.beta)));
            }
            num[e] = n_;
            den[e] = d_;
        } else { // rows beyond n: neutral
            num[e] = Fr::one();
            den[e] = Fr::one();
        }
        rb = fe_mul(rb, root);
    }
    // thread-local inclusive prefix of N and exclusive suffix of D
    Fr pn[GP_E], sdl[GP_E];
    pn[0] = num[0];
#pragma unroll
    for (int e = 1; e < GP_E; e++) pn[e] = fe_mul(pn[e - 1], num[e]);
    sdl[GP_E - 1] = Fr::one();
#pragma unroll
    for (int e = GP_E - 2; e >= 0; e--) sdl[e] = fe_mul(sdl[e + 1], den[e + 1]);
    const Fr tn = pn[GP_E - 1], td = fe_mul(sdl[0], den[0]);
    // block scans over the 256 thread totals: inclusive prefix of N (Hillis-Steele upwards), inclusive suffix of D (downwards)
    Fr vn = tn, vd = td;
    smn[tid] = vn;
    smd[tid] = vd;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        Fr xn = vn, xd = vd;
        if (tid >= d) xn = fe_mul(smn[tid - d], vn);
        if (tid + d < 256) xd = fe_mul(vd, smd[tid + d]);
        __syncthreads();
        vn = xn;
        vd = xd;
        smn[tid] = vn;
        smd[tid] = vd;
        __syncthreads();
    }
    const Fr before = tid ? smn[tid - 1] : Fr::one();        // product of N over the block's earlier threads
    const Fr after = tid + 1 < 256 ? smd[tid + 1] : Fr::one(); // product of D over the block's later threads
#pragma unroll
    for (int e = 0; e < GP_E; e++) {
        const size_t j = j0 + e;
        if (j < a.n) {
            if (j + 1 < a.n) fe_store<FrP>(a.z + j + 1, fe_mul(before, pn[e])); // in-block PN_j, parked where z[j+1] will be
            fe_store<FrP>(a.sd + j, fe_mul(sdl[e], after));                      // in-block SD_{j+1}
        }
    }
    if (tid == 255) a.bt[blockIdx.x] = vn;           // block total of N (inclusive prefix at the last thread)
    if (tid == 0) a.bt[a.nblocks + blockIdx.x] = vd; // block total of D (inclusive suffix at the first thread)
}
// exclusive prefix products of bt[0 .. B) (N totals), exclusive suffix products of bt[B .. 2B) (D totals), bt[2B] = prod of all D
__global__ void __launch_bounds__(256) k_gp_blocks(Fr* bt, size_t B)
{
    __shared__ Fr smn[256], smd[256];
    const int tid = threadIdx.x;
    const size_t per = (B + 255) / 256;
    const size_t lo = (size_t)tid * per, hi = lo + per < B ? lo + per : B;
    Fr vn = Fr::one(), vd = Fr::one();
    for (size_t i = lo; i < hi; i++) {
        vn = fe_mul(vn, fe_load<FrP>(bt + i));
        vd = fe_mul(vd, fe_load<FrP>(bt + B + i));
    }
    smn[tid] = vn;
    smd[tid] = vd;
    __syncthreads();
    const int active = (int)((B + per - 1) / per); // threads that hold totals; the others hold ones: a scan step beyond them changes nothing
    for (int d = 1; d < active; d <<= 1) {
        Fr xn = vn, xd = vd;
        if (tid >= d) xn = fe_mul(smn[tid - d], vn);
        if (tid + d < 256) xd = fe_mul(vd, smd[tid + d]);
        __syncthreads();
        vn = xn;
        vd = xd;
        smn[tid] = vn;
        smd[tid] = vd;
        __syncthreads();
    }
    if (tid == 0) bt[2 * B] = fe_reduce_once(smd[0]); // product of every D
    Fr pre = tid ? smn[tid - 1] : Fr::one();
    for (size_t i = lo; i < hi; i++) { // exclusive prefix of the N totals
        const Fr v = fe_load<FrP>(bt + i);
        fe_store<FrP>(bt + i, pre);
        pre = fe_mul(pre, v);
    }
    Fr suf = tid + 1 < 256 ? smd[tid + 1] : Fr::one();
    for (size_t i = hi; i-- > lo;) { // exclusive suffix of the D totals
        const Fr v = fe_load<FrP>(bt + B + i);
        fe_store<FrP>(bt + B + i, suf);
        suf = fe_mul(suf, v);
    }
}
__global__ void k_gp_invert(Fr* bt, size_t B)
{
    // every lane of the (single) wave computes the same inverse from the same address: uniform data, no divergent branch around the chain, so it
    // runs on the scalar unit (fe_inverse_gcd<.., true>); one lane stores.  A zero D has probability ~2^-230 (random beta, gamma).
    const Fr inv = fr_inverse(fe_load<FrP>(bt + 2 * B));
    if (threadIdx.x == 0 && blockIdx.x == 0) fe_store<FrP>(bt + 2 * B + 1, inv);
}
__global__ void __launch_bounds__(256) k_gp_apply(GpArgs a)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // row j -> z[j+1]
    if (j == 0) fe_store<FrP>(a.z, Fr::one());
    if (j + 1 >= a.n) return;
    const size_t b = j / a.block_rows;
    // per-block factor: (N of earlier blocks) * (D of later blocks) / (all D); 2 products per row is cheaper than another kernel
    const Fr f = fe_mul(fe_mul(fe_load<FrP>(a.bt + b), fe_load<FrP>(a.bt + a.nblocks + b)), fe_load<FrP>(a.bt + 2 * a.nblocks + 1));
    fe_store<FrP>(a.z + j + 1, fe_mul(fe_mul(fe_load<FrP>(a.z + j + 1), fe_load<FrP>(a.sd + j)), f));
}

int ntt_domain_consts(bbg_ctx* ctx, unsigned log2n, void** consts);

// set-up blocks: one per widget of a round (the chained form keeps all of them alive until the round's kernels have run)
constexpr int QUOT_SETUPS = 8;
static int quot_setup_block(bbg_ctx* ctx, int slot, QuotientSetup** out)
{
    int rc = ensure_buffer(&ctx->quot_setup, &ctx->quot_setup_bytes, QUOT_SETUPS * sizeof(QuotientSetup));
    if (rc) return rc;
    *out = (QuotientSetup*)ctx->quot_setup + slot;
    return BBG_OK;
}

// width = 4 (TurboPLONK, ProverPermutationWidget<4,false>) or 3 (StandardPLONK, <3,false>: w_4 / sigma_4 not read).
// Two halves so that a caller can put independent work between them: _begin queues the row terms, the scans and -- on inv_stream,
// ordered by events -- the single inversion; _finish makes `st` wait for it and applies it.
static int gp_fill(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n, void* d_z, GpArgs& a)
{
    if (log2n > 28) { set_error("bbg_permutation_grand_product_device: log2n > 28"); return BBG_E_INVALID; }
    if (width != 3 && width != 4) { set_error("bbg_permutation_grand_product_device: width must be 3 or 4"); return BBG_E_INVALID; }
    if (!d_wires || !d_sigmas || !d_z) { set_error("bbg_permutation_grand_product_device: null argument"); return BBG_E_INVALID; }
    for (int k = 0; k < 4; k++) {
        if (k < width && (!d_wires[k] || !d_sigmas[k])) { set_error("bbg_permutation_grand_product_device: null polynomial"); return BBG_E_INVALID; }
        a.w[k] = k < width ? (const Fr*)d_wires[k] : nullptr;
        a.sigma[k] = k < width ? (const Fr*)d_sigmas[k] : nullptr;
    }
    a.n = (size_t)1 << log2n;
    a.block_rows = (size_t)256 * gp_rows_per_thread(a.n);
    a.nblocks = (a.n + a.block_rows - 1) / a.block_rows;
    int rc = ensure_buffer(&ctx->gp_totals, &ctx->gp_totals_bytes, (a.n + 2 * a.nblocks + 2) * sizeof(Fr));
    if (rc) return rc;
    a.sd = (Fr*)ctx->gp_totals;
    a.bt = a.sd + a.n;
    a.z = (Fr*)d_z;
    QuotientSetup* setup = nullptr;
    rc = quot_setup_block(ctx, QUOT_SETUPS - 1, &setup); // its own block: the widgets' chain uses blocks 0 ..
    if (rc) return rc;
    a.s = setup;
    void* dc = nullptr;
    rc = ntt_domain_consts(ctx, log2n, &dc);
    if (rc) return rc;
    a.dc = (const DomainConsts*)dc;
    return BBG_OK;
}
int permutation_grand_product_begin(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n,
                                    const uint64_t* challenges, void* d_z, hipStream_t st, hipStream_t inv_stream, hipEvent_t ev_ready, hipEvent_t ev_inverted)
{
    if (!challenges) { set_error("bbg_permutation_grand_product_device: null argument"); return BBG_E_INVALID; }
    GpArgs a;
    int rc = gp_fill(ctx, width, d_wires, d_sigmas, log2n, d_z, a);
    if (rc) return rc;
    // the widgets' set-up block, of which the grand product reads beta, gamma, k1 .. k3 only: stored as they are (r5: the widgets' set-up
    // kernel ran here, 30 products on one lane -- 19 us -- to derive powers of an alpha this round does not have)
    QuotientChallenges ch;
    memset(&ch, 0, sizeof(ch));
    memcpy(&ch.v[2], challenges, 32);          // beta
    memcpy(&ch.v[3], challenges + 4, 32);      // gamma
    memcpy(&ch.v[6], challenges + 8, 3 * 32);  // k1..k3
    hipLaunchKernelGGL(k_gp_setup, dim3(1), dim3(64), 0, st, (QuotientSetup*)a.s, ch);
    {
        ProfScope ps(ctx, "grand_product", st);
        const dim3 g((unsigned)a.nblocks), b(256);
        if (a.block_rows == 256) {
            if (width == 4) hipLaunchKernelGGL((k_gp_terms<4, 1>), g, b, 0, st, a);
            else hipLaunchKernelGGL((k_gp_terms<3, 1>), g, b, 0, st, a);
        } else {
            if (width == 4) hipLaunchKernelGGL((k_gp_terms<4, 4>), g, b, 0, st, a);
            else hipLaunchKernelGGL((k_gp_terms<3, 4>), g, b, 0, st, a);
        }
        hipLaunchKernelGGL(k_gp_blocks, dim3(1), dim3(256), 0, st, a.bt, a.nblocks);
    }
    if (inv_stream != st) {
        BBG_HIP(hipEventRecord(ev_ready, st));
        BBG_HIP(hipStreamWaitEvent(inv_stream, ev_ready, 0));
    }
    hipLaunchKernelGGL(k_gp_invert, dim3(1), dim3(64), 0, inv_stream, a.bt, a.nblocks);
    if (inv_stream != st) BBG_HIP(hipEventRecord(ev_inverted, inv_stream));
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}
int permutation_grand_product_finish(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n, void* d_z,
                                     hipStream_t st, hipEvent_t ev_inverted)
{
    GpArgs a;
    int rc = gp_fill(ctx, width, d_wires, d_sigmas, log2n, d_z, a);
    if (rc) return rc;
    if (ev_inverted) BBG_HIP(hipStreamWaitEvent(st, ev_inverted, 0));
    ProfScope ps(ctx, "grand_product", st);
    hipLaunchKernelGGL(k_gp_apply, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, st, a);
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}
int permutation_grand_product_w(bbg_ctx* ctx, int width, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n,
                                const uint64_t* challenges, void* d_z, hipStream_t st)
{
    int rc = permutation_grand_product_begin(ctx, width, d_wires, d_sigmas, log2n, challenges, d_z, st, st, nullptr, nullptr);
    if (rc) return rc;
    return permutation_grand_product_finish(ctx, width, d_wires, d_sigmas, log2n, d_z, st, nullptr);
}
int permutation_grand_product(bbg_ctx* ctx, const void* const* d_wires, const void* const* d_sigmas, unsigned log2n, const uint64_t* challenges,
                              void* d_z, hipStream_t st)
{
    return permutation_grand_product_w(ctx, 4, d_wires, d_sigmas, log2n, challenges, d_z, st);
}

// which polynomials each widget reads (a null pointer for one of them is an error; the others may be null)
static const uint32_t WIDGET_NEEDS[WIDGET_COUNT] = {
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_W4) | (1u << QP_Z) | (1u << QP_S1) | (1u << QP_S2) | (1u << QP_S3) |
        (1u << QP_S4) | (1u << QP_L1),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_W4) | (1u << QP_Q1) | (1u << QP_Q2) | (1u << QP_Q3) | (1u << QP_Q4) |
        (1u << QP_Q5) | (1u << QP_QM) | (1u << QP_QC) | (1u << QP_QARITH),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_W4) | (1u << QP_Q1) | (1u << QP_Q2) | (1u << QP_Q3) | (1u << QP_Q4) |
        (1u << QP_Q5) | (1u << QP_QM) | (1u << QP_QC) | (1u << QP_QECC),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_W4) | (1u << QP_QRANGE),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_W4) | (1u << QP_QC) | (1u << QP_QLOGIC),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_Z) | (1u << QP_S1) | (1u << QP_S2) | (1u << QP_S3) | (1u << QP_L1),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_Q1) | (1u << QP_Q2) | (1u << QP_Q3) | (1u << QP_QM) | (1u << QP_QC),
    (1u << QP_W1) | (1u << QP_W2) | (1u << QP_W3) | (1u << QP_QMIMC_C) | (1u << QP_QMIMC_S),
};

static int launch_widget(bbg_ctx* ctx, int widget, const QuotientArgs& a, size_t m, hipStream_t st)
{
    ProfScope ps(ctx, "quotient_widget", st);
    if (ctx->quotient_limbs29) { // the 29-bit-limb kernels of quotient29.hip.h
        switch (widget) {
        case 0:
            if (m < ((size_t)1 << 20)) hipLaunchKernelGGL((q29::k_quotient29_permutation<4, 1>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((q29::k_quotient29_permutation<4, PERM_CH>), dim3((unsigned)((m / PERM_CH + 255) / 256)), dim3(256), 0, st, a);
            break;
        case 5:
            if (m < ((size_t)1 << 20)) hipLaunchKernelGGL((q29::k_quotient29_permutation<3, 1>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((q29::k_quotient29_permutation<3, PERM_CH>), dim3((unsigned)((m / PERM_CH + 255) / 256)), dim3(256), 0, st, a);
            break;
        case 2:
            hipLaunchKernelGGL(q29::k_quotient29_turbo_fixed_base_linear, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
            hipLaunchKernelGGL(q29::k_quotient29_turbo_fixed_base_gate, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
            break;
        case 1: hipLaunchKernelGGL(q29::k_quotient29_turbo_arith_range_logic<1>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, a.s, a.s); break;
        case 3: hipLaunchKernelGGL(q29::k_quotient29_turbo_arith_range_logic<2>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, a.s, a.s); break;
        case 4: hipLaunchKernelGGL(q29::k_quotient29_turbo_arith_range_logic<4>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, a.s, a.s); break;
        case 6: hipLaunchKernelGGL(q29::k_quotient29_standard_arith, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
        case 7: hipLaunchKernelGGL(q29::k_quotient29_mimc, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
        default: set_error("bbg_quotient_widget_device: unknown widget"); return BBG_E_INVALID;
        }
        BBG_HIP(hipGetLastError());
        return BBG_OK;
    }
    switch (widget) {
    case 0: hipLaunchKernelGGL(k_quotient_permutation<4>, dim3((unsigned)((m / PERM_CH + 255) / 256)), dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL(k_quotient_permutation<3>, dim3((unsigned)((m / PERM_CH + 255) / 256)), dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL(k_quotient_standard_arith, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
    case 7: hipLaunchKernelGGL(k_quotient_mimc, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL(k_quotient_turbo_arith, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
    case 2:
        hipLaunchKernelGGL(k_quotient_turbo_fixed_base_linear, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_quotient_turbo_fixed_base_gate, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a);
        break;
    case 3: hipLaunchKernelGGL(k_quotient_turbo_range, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_quotient_turbo_logic, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a); break;
    default: set_error("bbg_quotient_widget_device: unknown widget"); return BBG_E_INVALID;
    }
    BBG_HIP(hipGetLastError());
    return BBG_OK;
}

// `count` widgets in the prover's order (execute_fourth_round, prover.cpp:304-319), each starting from the alpha_base the previous
// one returned -- chained on the device (set-up block k reads alpha_out of block k-1), no host round trip between them.
// challenges: 9 Montgomery Fr on the host: alpha_base, alpha, beta, gamma, public_input_delta, g, k1, k2, k3.
int quotient_widgets_chain(bbg_ctx* ctx, const int* widgets, int count, const void* const* d_polys, unsigned log2_large,
                           const uint64_t* challenges, void* d_quotient, uint64_t* alpha_out, hipStream_t st)
{
    if (count < 1 || count >= QUOT_SETUPS) { set_error("bbg_quotient_widget_device: bad widget count"); return BBG_E_INVALID; }
    if (log2_large < 3 || log2_large > 28) { set_error("bbg_quotient_widget_device: need 3 <= log2 of the 4n domain <= 28"); return BBG_E_INVALID; }
    if (!d_polys || !challenges || !d_quotient || !widgets) { set_error("bbg_quotient_widget_device: null argument"); return BBG_E_INVALID; }
    QuotientArgs a;
    bool extended = false; // d_polys has BBG_QP_EXT_COUNT entries only when a widget that reads the extension is asked for
    for (int w = 0; w < count; w++) {
        if (widgets[w] < 0 || widgets[w] >= WIDGET_COUNT) { set_error("bbg_quotient_widget_device: unknown widget"); return BBG_E_INVALID; }
        extended = extended || (WIDGET_NEEDS[widgets[w]] >> QP_COUNT) != 0;
    }
    for (int k = 0; k < QP_EXT_COUNT; k++) a.p[k] = (k < QP_COUNT || extended) ? (const Fr*)d_polys[k] : nullptr;
    for (int w = 0; w < count; w++) {
        for (int k = 0; k < QP_EXT_COUNT; k++)
            if (((WIDGET_NEEDS[widgets[w]] >> k) & 1u) && !a.p[k]) {
                set_error("bbg_quotient_widget_device: a polynomial this widget reads is null");
                return BBG_E_INVALID;
            }
    }
    QuotientSetup* setups = nullptr;
    int rc = quot_setup_block(ctx, 0, &setups);
    if (rc) return rc;
    void* dc = nullptr;
    rc = ntt_domain_consts(ctx, log2_large, &dc);
    if (rc) return rc;
    QuotientChallenges ch;
    memcpy(&ch, challenges, sizeof(ch));
    a.quotient = (Fr*)d_quotient;
    a.mask = (uint32_t)(((size_t)1 << log2_large) - 1);
    a.dc = (const DomainConsts*)dc;
    const size_t m = (size_t)1 << log2_large;
    // the alpha_base chain runs over the set-up kernels alone (block w reads the alpha_out of block w - 1), so the widget kernels behind
    // them may run in any order and share passes: arithmetic + range + logic of a TurboPLONK chain go through the data once
    int pos_arith = -1, pos_range = -1, pos_logic = -1;
    {
        WidgetPlan plan;
        if (ctx->quotient_setup_plan && widget_plan(widgets, count, plan)) {
            hipLaunchKernelGGL(k_quotient_setup_plan, dim3(1), dim3(64), 0, st, setups, ch, plan);
        } else {
            WidgetChain chain;
            chain.count = count;
            for (int w = 0; w < 8; w++) chain.widget[w] = w < count ? widgets[w] : 0;
            hipLaunchKernelGGL(k_quotient_setup_chain, dim3(1), dim3(64), 0, st, setups, ch, chain);
        }
    }
    for (int w = 0; w < count; w++) {
        if (widgets[w] == 1 && pos_arith < 0) pos_arith = w;
        if (widgets[w] == 3 && pos_range < 0) pos_range = w;
        if (widgets[w] == 4 && pos_logic < 0) pos_logic = w;
    }
    // a widget that ASSIGNS the quotient (the permutation widgets) must come first; the prover's chains start with it
    const bool fuse = ctx->quotient_fuse && pos_arith > 0 && pos_range > 0 && pos_logic > 0 && (widgets[0] == 0 || widgets[0] == 5);
    for (int w = 0; w < count; w++) {
        if (fuse && (w == pos_arith || w == pos_range || w == pos_logic)) continue;
        a.s = setups + w;
        rc = launch_widget(ctx, widgets[w], a, m, st);
        if (rc) return rc;
    }
    if (fuse) {
        ProfScope ps(ctx, "quotient_widget", st);
        a.s = setups + pos_arith;
        if (ctx->quotient_limbs29)
            hipLaunchKernelGGL(q29::k_quotient29_turbo_arith_range_logic<7>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, setups + pos_range, setups + pos_logic);
        else
            hipLaunchKernelGGL(k_quotient_turbo_arith_range_logic, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, a, setups + pos_range, setups + pos_logic);
        BBG_HIP(hipGetLastError());
    }
    if (alpha_out) {
        BBG_HIP(hipMemcpyAsync(alpha_out, &setups[count - 1].alpha_out[widgets[count - 1]], sizeof(Fr), hipMemcpyDeviceToHost, st));
        BBG_HIP(hipStreamSynchronize(st));
    }
    return BBG_OK;
}

int quotient_widget(bbg_ctx* ctx, int widget, const void* const* d_polys, unsigned log2_large, const uint64_t* challenges, void* d_quotient,
                    uint64_t* alpha_out, hipStream_t st)
{
    return quotient_widgets_chain(ctx, &widget, 1, d_polys, log2_large, challenges, d_quotient, alpha_out, st);
}

} // namespace bbg
