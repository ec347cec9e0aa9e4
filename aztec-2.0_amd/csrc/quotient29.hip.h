// The TurboPLONK / StandardPLONK quotient widgets of quotient.hip on lazily reduced 29-bit limbs (w29.hip.h) -- option "quotient_limbs29"
// (default 1).  Same identities, same set-up blocks, same [0, 2p) residues in and out as the 32-bit kernels (which stay as the A/B path,
// option 0); the bounds of every intermediate value are template arguments of its type and checked by the compiler.  Included by
// quotient.hip after QuotientArgs / QuotientSetup / QLOAD.
//
// Conventions: linear combinations of LOADED values (d = w3 - 4 w4, 9 c - 3 (a + b), ...) stay on the 8 x u32 words (fe_add / fe_sub: exact,
// [0, 2p)), products and sums of products run on the 29-bit limbs; a value enters them by the limb split of ld<0> (as it is) or ld<1>
// (shifted by 5 bits), whichever class its place in the expression needs (w29.hip.h).
#pragma once
#include "w29.hip.h"

namespace bbg {
namespace q29 {
using namespace w29;

// waves per SIMD the kernels are compiled for (A/B: make EXTRA=-DBBG_Q29_OCC_ARL=3)
#ifndef BBG_Q29_OCC_ARL
#define BBG_Q29_OCC_ARL 2
#endif
#ifndef BBG_Q29_OCC_FBG
#define BBG_Q29_OCC_FBG 2
#endif
#ifndef BBG_Q29_OCC_FBL
#define BBG_Q29_OCC_FBL 2
#endif
#ifndef BBG_Q29_OCC_PERM
#define BBG_Q29_OCC_PERM 2
#endif

#define Q0(id, idx) ld<0>(QLOAD(id, idx))
#define Q1(id, idx) ld<1>(QLOAD(id, idx))

__device__ __forceinline__ Fr x3(const Fr& v) { return fe_add(fe_add(v, v), v); }

// f(D) = D (D - 1)(D - 2)(D - 3) = u^2 + 2 u, u = D^2 - 3 D, class 1, from D^2 (class 1) and D's words
template <class D2> __device__ __forceinline__ auto quad_from(const D2& d2, const Fr& d)
{
    const auto u = carry(sub(d2, ld<1>(x3(d))));
    return carry(add(sqr(u), dbl(u)));
}

// ---- turbo arithmetic, without its alpha_base:
//   q_arith (q_m w1 w2 + q_1 w1 + q_2 w2 + q_3 w3 + q_4 w4 + q_c + alpha q_5 w4 (w4 - 1)(w4 - 2)) + (q_arith^2 - q_arith) d (9 d - 2 d^2 - 7)
// d = w3 - 4 w4, d2 = d^2 in class 1 (shared with the range widget's first quad)
template <class D2> __device__ __forceinline__ auto arith_part(const QuotientArgs& a, const QuotientSetup& s, uint32_t i, const Fr& w1, const Fr& w2,
                                                                const Fr& w3, const Fr& w4, const Fr& qc, const Fr& d, const D2& d2)
{
    const Fr qa = QLOAD(QP_QARITH, i);
    const auto w12 = mul(ld<1>(w1), ld<0>(w2));
    const auto u4 = sub(mul(ld<1>(w4), ld<0>(w4)), ld<0>(w4));        // w4^2 - w4
    const auto t2 = mul(mul(u4, ld<1>(fe_sub(w4, s.c2))), ld<1>(s.alpha));
    const auto gate = dot(t(w12, Q1(QP_QM, i)), t(ld<0>(w1), Q1(QP_Q1, i)), t(ld<0>(w2), Q1(QP_Q2, i)), t(ld<0>(w3), Q1(QP_Q3, i)),
                          t(ld<0>(w4), Q1(QP_Q4, i)), t(t2, Q1(QP_Q5, i)));
    const auto g2 = add(gate, ld<0>(qc));
    const Fr d8 = fe_add(x4(d), x4(d));
    const Fr lin = fe_sub(fe_add(d8, d), s.c7);                        // 9 d - 7
    const auto h1 = mul(sub(ld<1>(lin), dbl(d2)), ld<0>(d));           // (9 d - 7 - 2 d^2) d
    const auto qq = sub(sqr(ld<1>(qa)), ld<1>(qa));                    // q_arith^2 - q_arith
    return dot(t(g2, ld<1>(qa)), t(h1, qq));
}
// ---- range, without q_range: sum_k ap[k] f(D_k), D_1 = d = w3 - 4 w4, D_2 = w2 - 4 w3, D_3 = w1 - 4 w2, D_4 = w4' - 4 w1
template <class D2> __device__ __forceinline__ auto range_part(const QuotientSetup& r, const Fr& w1, const Fr& w2, const Fr& w3, const Fr& w4n, const Fr& d,
                                                                const D2& d2)
{
    const Fr d2_ = fe_sub(w2, x4(w3)), d3_ = fe_sub(w1, x4(w2)), d4_ = fe_sub(w4n, x4(w1));
    const auto f1 = quad_from(d2, d);
    const auto f2 = quad_from(sqr(ld<1>(d2_)), d2_);
    const auto f3 = quad_from(sqr(ld<1>(d3_)), d3_);
    const auto f4 = quad_from(sqr(ld<1>(d4_)), d4_);
    return dot(t(f1, ld<0>(r.ap[0])), t(f2, ld<0>(r.ap[1])), t(f3, ld<0>(r.ap[2])), t(f4, ld<0>(r.ap[3])));
}
// ---- logic, without q_logic: ap0 [ 2 (a b - w3) alpha^3 + f(a) alpha^2 + f(b) alpha + 3 (a + b + c) - 2 E + q_c (9 c - 3 (a + b)) ],
//      E = w3 (w3 (4 w3 - 18 (a + b) + 81) + 18 (a^2 + b^2) - 81 (a + b) + 83),  a = w1' - 4 w1, b = w2' - 4 w2, c = w4' - 4 w4
__device__ __forceinline__ auto logic_part(const QuotientArgs& a, const QuotientSetup& l, uint32_t ish, const Fr& w1, const Fr& w2, const Fr& w3,
                                           const Fr& w4, const Fr& w4n, const Fr& qc)
{
    const Fr qa = fe_sub(QLOAD(QP_W1, ish), x4(w1));
    const Fr qb = fe_sub(QLOAD(QP_W2, ish), x4(w2));
    const Fr qcq = fe_sub(w4n, x4(w4));
    const Fr sum = fe_add(qa, qb);
    const Fr sum3 = x3(sum), sum9 = x3(sum3), sum18 = fe_add(sum9, sum9);
    const Fr sum81 = fe_add(x4(sum18), sum9);
    const Fr c3 = x3(qcq), c9 = x3(c3);
    const auto a2 = sqr(ld<1>(qa)), b2 = sqr(ld<1>(qb));
    const auto fa = quad_from(a2, qa), fb = quad_from(b2, qb);
    const auto abw = carry(sub(mul(ld<1>(qa), ld<0>(qb)), ld<0>(w3)));             // a b - w3
    const Fr in1 = fe_add(fe_sub(x4(w3), sum18), l.c81);                           // 4 w3 - 18 (a + b) + 81
    const Fr lin2 = fe_sub(l.c83, sum81);                                          // 83 - 81 (a + b)
    const Fr w3_9 = x3(x3(w3));
    const auto x = mul(ld<1>(w3), ld<1>(in1));
    const auto e = dot(t(ld<0>(w3), add(x, ld<1>(lin2))), t(ld<0>(fe_add(w3_9, w3_9)), add(a2, b2)));
    const auto id = dot(t(abw, ld<1>(l.alpha3x2)), t(fa, ld<0>(l.alpha2)), t(fb, ld<0>(l.alpha)), t(ld<0>(fe_sub(c9, sum3)), ld<1>(qc)));
    const auto tail = carry(sub(add(id, ld<0>(fe_add(c3, sum3))), dbl(e)));
    return mul(tail, ld<1>(l.ap[0]));
}

// PARTS = 7: arithmetic + range + logic in one pass over the wires (k_quotient_turbo_arith_range_logic) with ONE final reduction for the three
// contributions and the quotient's own value; a.s = the arithmetic widget's set-up block.  PARTS = 1 / 2 / 4: one widget (a.s = its own block).
template <int PARTS> __global__ void __launch_bounds__(256, BBG_Q29_OCC_ARL)
k_quotient29_turbo_arith_range_logic(QuotientArgs a, const QuotientSetup* s_range, const QuotientSetup* s_logic)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const auto q = ld<0>(fe_load<FrP>(a.quotient + i));
    if constexpr (PARTS == 7) {
        const Fr w4n = QLOAD(QP_W4, ish);
        const Fr qc = QLOAD(QP_QC, i);
        const Fr d = fe_sub(w3, x4(w4)); // the arithmetic widget's high-bit term and the range widget's first quad
        const auto d2 = sqr(ld<1>(d));
        const auto inner = arith_part(a, s, i, w1, w2, w3, w4, qc, d, d2);
        const auto range = range_part(*s_range, w1, w2, w3, w4n, d, d2);
        const auto logic = logic_part(a, *s_logic, ish, w1, w2, w3, w4, w4n, qc);
        const auto total = dot(t(inner, ld<1>(s.ap[0])), t(range, Q1(QP_QRANGE, i)), t(logic, Q1(QP_QLOGIC, i)));
        fe_store<FrP>(a.quotient + i, finish(add(total, q), red));
    } else if constexpr (PARTS == 1) {
        const Fr d = fe_sub(w3, x4(w4));
        const auto inner = arith_part(a, s, i, w1, w2, w3, w4, QLOAD(QP_QC, i), d, sqr(ld<1>(d)));
        fe_store<FrP>(a.quotient + i, finish(add(mul(inner, ld<1>(s.ap[0])), q), red));
    } else if constexpr (PARTS == 2) {
        const Fr d = fe_sub(w3, x4(w4));
        const auto range = range_part(s, w1, w2, w3, QLOAD(QP_W4, ish), d, sqr(ld<1>(d)));
        fe_store<FrP>(a.quotient + i, finish(add(mul(range, Q1(QP_QRANGE, i)), q), red));
    } else {
        const auto logic = logic_part(a, s, ish, w1, w2, w3, w4, QLOAD(QP_W4, ish), QLOAD(QP_QC, i));
        fe_store<FrP>(a.quotient + i, finish(add(mul(logic, Q1(QP_QLOGIC, i)), q), red));
    }
}

// StandardPLONK arithmetic gate (k_quotient_standard_arith): alpha_base (q_m w1 w2 + q_1 w1 + q_2 w2 + q_3 w3 + q_c)
__global__ void __launch_bounds__(256) k_quotient29_standard_arith(QuotientArgs a)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i);
    const auto w12 = mul(ld<1>(w1), ld<0>(w2));
    const auto gate = dot(t(w12, Q1(QP_QM, i)), t(ld<0>(w1), Q1(QP_Q1, i)), t(ld<0>(w2), Q1(QP_Q2, i)), t(Q0(QP_W3, i), Q1(QP_Q3, i)));
    const auto out = mul(add(gate, Q0(QP_QC, i)), ld<1>(s.ap[0]));
    fe_store<FrP>(a.quotient + i, finish(add(out, ld<0>(fe_load<FrP>(a.quotient + i))), red));
}

// MiMC round gate (k_quotient_mimc), T = w1 + w3 + q_mimc_coefficient:  q_mimc_selector [ ap0 (T^3 - w2) + ap1 (w2^2 T - w3(wX)) ]
__global__ void __launch_bounds__(256) k_quotient29_mimc(QuotientArgs a)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const Fr w2 = QLOAD(QP_W2, i);
    const Fr tt = fe_add(fe_add(QLOAD(QP_W1, i), QLOAD(QP_W3, i)), QLOAD(QP_QMIMC_C, i));
    const auto cube = carry(sub(mul(sqr(ld<1>(tt)), ld<0>(tt)), ld<0>(w2)));
    const auto nxt = carry(sub(mul(sqr(ld<1>(w2)), ld<0>(tt)), Q0(QP_W3, (i + 4) & a.mask)));
    const auto id = dot(t(cube, ld<1>(s.ap[0])), t(nxt, ld<1>(s.ap[1])));
    const auto out = mul(id, Q1(QP_QMIMC_S, i));
    fe_store<FrP>(a.quotient + i, finish(add(out, ld<0>(fe_load<FrP>(a.quotient + i))), red));
}

// fixed-base ladder, the selector-weighted terms (k_quotient_turbo_fixed_base_linear):
//   q_ecc [ q_1 ap1 delta^2 + q_2 ap1 + q_3 delta w3' (ap3 (w1' - w1) + 2 ap2 w2) + q_c w3 (ap5 q_4 + ap6 q_m) + q_c ap5 (1 - w4) q_5 ]
__global__ void __launch_bounds__(256, BBG_Q29_OCC_FBL) k_quotient29_turbo_fixed_base_linear(QuotientArgs a)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w4 = QLOAD(QP_W4, i);
    const Fr delta = fe_sub(QLOAD(QP_W4, ish), x4(w4));
    const Fr w3n = QLOAD(QP_W3, ish);
    const Fr w1 = QLOAD(QP_W1, i), w3 = QLOAD(QP_W3, i), qc = QLOAD(QP_QC, i);
    const auto dsq = mul(ld<1>(delta), ld<0>(delta));                                     // class 0
    const auto q1t = mul(dsq, Q1(QP_Q1, i));                                              // q_1 delta^2
    const auto dw = mul(ld<1>(delta), ld<0>(w3n));                                        // delta w3'
    const Fr w2x2 = fe_add(QLOAD(QP_W2, i), QLOAD(QP_W2, i));
    const auto y = dot(t(ld<0>(fe_sub(QLOAD(QP_W1, ish), w1)), ld<1>(s.ap[3])), t(ld<0>(w2x2), ld<1>(s.ap[2]))); // ap3 (w1' - w1) + 2 ap2 w2
    const auto t3 = mul(dw, up(y));                                                       // class 0
    const auto sel = dot(t(ld<0>(s.ap[5]), Q1(QP_Q4, i)), t(ld<0>(s.ap[6]), Q1(QP_QM, i)));           // ap5 q_4 + ap6 q_m
    const auto i5 = mul(ld<0>(fe_sub(s.one, w4)), ld<1>(s.ap[5]));                        // ap5 (1 - w4)
    const auto init = dot(t(sel, ld<1>(w3)), t(i5, Q1(QP_Q5, i)));                        // class 0
    const auto lin = dot(t(q1t, ld<1>(s.ap[1])), t(ld<0>(s.ap[1]), Q1(QP_Q2, i)), t(t3, Q1(QP_Q3, i)), t(init, ld<1>(qc)));
    const auto out = mul(lin, Q1(QP_QECC, i));
    fe_store<FrP>(a.quotient + i, finish(add(out, ld<0>(fe_load<FrP>(a.quotient + i))), red));
}

// fixed-base ladder, the gate identities (k_quotient_turbo_fixed_base_gate)
__global__ void __launch_bounds__(256, BBG_Q29_OCC_FBG) k_quotient29_turbo_fixed_base_gate(QuotientArgs a)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > a.mask) return;
    const uint32_t ish = (i + 4) & a.mask;
    const Fr w1 = QLOAD(QP_W1, i), w2 = QLOAD(QP_W2, i), w3 = QLOAD(QP_W3, i), w4 = QLOAD(QP_W4, i);
    const Fr w1n = QLOAD(QP_W1, ish), w3n = QLOAD(QP_W3, ish);
    const Fr qc = QLOAD(QP_QC, i), qe = QLOAD(QP_QECC, i);
    const Fr delta = fe_sub(QLOAD(QP_W4, ish), x4(w4));
    // accumulator identity: (delta + 1)(delta + 3)(delta - 1)(delta - 3) = (delta^2 - 1)(delta^2 - 9)
    const auto dsq = sqr(ld<1>(delta));                                                   // class 1
    const auto acc = mul(carry(sub(dsq, ld<1>(s.one))), carry(sub(dsq, ld<1>(x3(s.c3))))); // class 1
    // x identity: (x3 + x1 + x_alpha)(x_alpha - x1)^2 - x_alpha^3 - y1^2 + 17 + 2 delta y1 q_ecc
    const Fr dx = fe_sub(w3n, w1);
    const auto dx2 = sqr(ld<1>(dx));                                                      // class 1
    const auto xa2 = sqr(ld<1>(w3n));                                                     // class 1
    const auto dy = mul(ld<1>(delta), ld<0>(w2));                                         // class 0: delta y1
    const auto nw3n = carry(neg(ld<0>(w3n))), nw2 = carry(neg(ld<0>(w2)));                // 3p - x_alpha, 3p - y1
    const Fr qe2 = fe_add(qe, qe);
    const auto xid = dot(t(ld<0>(fe_add(fe_add(w1n, w1), w3n)), dx2), t(nw3n, xa2), t(nw2, ld<1>(w2)), t(dy, ld<1>(qe2))); // class 0
    const auto xid17 = add(xid, ld<0>(s.c17));
    // y identity: (y3 + y1)(x_alpha - x1) + (x1 - x3)(y1 - q_ecc delta)
    const auto qd = mul(ld<1>(qe), ld<0>(delta));                                         // class 0
    const auto ym = carry(sub(ld<0>(w2), qd));                                            // y1 - q_ecc delta, class 0
    const auto yid = dot(t(ld<0>(fe_add(QLOAD(QP_W2, ish), w2)), ld<1>(dx)), t(ym, ld<1>(fe_sub(w1, w1n))));   // class 0
    // initialisation row: q_c [ ap4 (w4 - 1)(w4 - 1 - w3) - ap5 w1 w3 + ap6 ((1 - w4) q_c - w2 w3) ]
    const Fr w4m1 = fe_sub(w4, s.one);
    const auto i1 = mul(ld<1>(w4m1), ld<0>(fe_sub(w4m1, w3)));                            // class 0
    const auto i2 = mul(ld<1>(w1), ld<0>(w3));                                            // class 0
    const auto i3 = dot(t(ld<0>(fe_sub(s.one, w4)), ld<1>(qc)), t(nw2, ld<1>(w3)));    // class 0
    const auto init = dot(t(i1, ld<1>(s.ap[4])), t(neg(i2), ld<1>(s.ap[5])), t(i3, ld<1>(s.ap[6])));            // class 0
    // gate = ap0 acc - ap1 w3' + ap2 xid + ap3 yid + q_c init, all times q_ecc
    const auto gate = dot(t(acc, ld<0>(s.ap[0])), t(nw3n, ld<1>(s.ap[1])), t(xid17, ld<1>(s.ap[2])), t(yid, ld<1>(s.ap[3])), t(init, ld<1>(qc)));
    const auto out = mul(carry(gate), ld<1>(qe));
    fe_store<FrP>(a.quotient + i, finish(add(out, ld<0>(fe_load<FrP>(a.quotient + i))), red));
}

// permutation argument (k_quotient_permutation): ASSIGNS the quotient
// CH points per thread: PERM_CH from 2^20 points up; 1 below (a 2^18-point domain at four points per thread is one wave per SIMD with a chain of
// 4 x 22 products; the chip has the lanes for a thread per point, and beta g w^i from the table costs what three more points would)
template <int WIDTH, int CH> __global__ void __launch_bounds__(256, BBG_Q29_OCC_PERM) k_quotient29_permutation(QuotientArgs a)
{
    __shared__ uint32_t red[NTT29_TABLE_WORDS];
    fill_table(red);
    __syncthreads();
    const QuotientSetup& s = *a.s;
    // a block covers 256 * CH consecutive points, 256 at a time: the lanes of a wave read consecutive 32-byte values of each of the
    // thirteen arrays (whole cache lines per instruction), a thread's next point is 256 further on and its beta g w^i is w^256 times the last
    const uint32_t i0 = blockIdx.x * (256u * CH) + threadIdx.x;
    if (i0 > a.mask) return;
    Fr rb = fe_mul(s.beta_g, pow_from_table(a.dc->pow2_root, (uint64_t)i0)); // beta * g * w^i (words: [0, 2p))
    const Fr root = a.dc->pow2_root[8];                                      // w^256
#pragma unroll 1
    for (int e = 0; e < CH; e++) {
        const uint32_t i = i0 + e * 256u, ish = (i + 4) & a.mask;
        if (i > a.mask) break;
        const auto rb1 = ld<1>(rb);
        // factors of the numerator w_k + gamma + beta K_k X and of the denominator w_k + gamma + beta sigma_k: the first of each product
        // chain in class 0, the others in class 1 (a class-0 running product times a class-1 factor stays in class 0)
        const Fr wg1 = fe_add(QLOAD(QP_W1, i), s.gamma);
        auto num = add(ld<0>(wg1), ld<0>(rb));
        auto den = add(ld<0>(wg1), mul(Q0(QP_S1, i), ld<1>(s.beta)));
        {
            const auto wg = ld<1>(fe_add(QLOAD(QP_W2, i), s.gamma));
            const auto nn = mul(num, add(wg, mul(rb1, ld<1>(s.k1))));
            const auto dd = mul(den, add(wg, mul(Q1(QP_S2, i), ld<1>(s.beta))));
            const auto wh = ld<1>(fe_add(QLOAD(QP_W3, i), s.gamma));
            const auto nn2 = mul(nn, add(wh, mul(rb1, ld<1>(s.k2))));
            const auto dd2 = mul(dd, add(wh, mul(Q1(QP_S3, i), ld<1>(s.beta))));
            const Fr z = QLOAD(QP_Z, i), zw = QLOAD(QP_Z, ish);
            const auto t1 = mul(ld<0>(fe_sub(zw, s.delta)), ld<1>(s.ap[0]));          // (z(wX) - delta) alpha_base
            const auto t2 = mul(ld<0>(fe_sub(z, s.one)), ld<1>(s.alpha_base_sqr));    // (z(X) - 1) alpha_base^2
            if constexpr (WIDTH == 4) {
                const auto wi = ld<1>(fe_add(QLOAD(QP_W4, i), s.gamma));
                const auto nn3 = mul(nn2, add(wi, mul(rb1, ld<1>(s.k3))));
                const auto dd3 = mul(dd2, add(wi, mul(Q1(QP_S4, i), ld<1>(s.beta))));
                const auto in = dot(t(nn3, ld<1>(z)), t(neg(dd3), ld<1>(zw)), t(t1, Q1(QP_L1, (i + 4 + 16) & a.mask)), t(t2, Q1(QP_L1, i)));
                fe_store<FrP>(a.quotient + i, finish(mul(in, ld<1>(s.ap[0])), red));
            } else {
                const auto in = dot(t(nn2, ld<1>(z)), t(neg(dd2), ld<1>(zw)), t(t1, Q1(QP_L1, (i + 4 + 16) & a.mask)), t(t2, Q1(QP_L1, i)));
                fe_store<FrP>(a.quotient + i, finish(mul(in, ld<1>(s.ap[0])), red));
            }
        }
        rb = fe_mul(rb, root);
    }
}

#undef Q0
#undef Q1
} // namespace q29
} // namespace bbg
