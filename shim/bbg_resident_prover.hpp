// bbg_resident_prover.hpp -- the C++ host side of the RESIDENT prover: a drop-in for
// waffle::ProverBase<settings>::construct_proof() (plonk/proof_system/prover/prover.cpp:420-436) that keeps every O(n) step of a
// TurboPLONK / StandardPLONK proof on the GPU behind libbbg.so's C ABI (include/bbg.h, bbg_prover_*).
//
// Compiled INSIDE a barretenberg build (it includes barretenberg's own headers; nothing of the reference is copied here).  It uses
// only PUBLIC members of the prover: `transcript` (all Fiat-Shamir hashing and the manifest stay the reference's), `key`, `witness`,
// the widget lists (to recognise the flavour and to take the linearisation scalars from the reference's own kernel templates) and
// export_proof().  What moves to the device, round by round:
//
//   execute_preamble_round + first round   blinded wires up, ifft, W_i                 bbg_prover_round1
//   execute_third_round                    z (grand product), Z, coset FFTs            bbg_prover_round3
//   execute_fourth_round                   quotient widgets, / Z*_H, coset_ifft, T_i   bbg_prover_round4
//   execute_fifth_round                    evaluations, r(X), r(zeta)                  bbg_prover_evaluate / bbg_prover_linearise
//   execute_sixth_round                    opening polynomials, PI_Z, PI_Z_OMEGA       bbg_prover_round6
//
// Lifetime, explicitly: a ResidentKey owns the device copy of ONE proving key (selectors, permutations) and holds the
// std::shared_ptr<proving_key>, so the key cannot be freed or its address reused while the device copy exists.  Many proofs
// (different witnesses) over the same key reuse it.  A maintainer's patch is two lines:
//
//     bbg_shim::ResidentKey rk(prover.key, settings::program_width);     // once per circuit
//     auto& proof = bbg_shim::construct_proof(prover, rk);               // instead of prover.construct_proof()
//
// The host arrays of the key and the witness are left as the composer produced them (wires in Lagrange form with the blinding
// rows written, like execute_preamble_round leaves them); the proof is the only output, as with the reference.
#pragma once
#include <array>
#include <cerrno>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <sys/random.h>

#include <numeric/uintx/uintx.hpp>
#include <plonk/proof_system/prover/prover.hpp>
#include <plonk/proof_system/public_inputs/public_inputs.hpp>
#include <plonk/proof_system/types/program_settings.hpp>
#include <plonk/proof_system/widgets/random_widgets/permutation_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/arithmetic_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/mimc_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/turbo_arithmetic_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/turbo_fixed_base_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/turbo_logic_widget.hpp>
#include <plonk/proof_system/widgets/transition_widgets/turbo_range_widget.hpp>
#include <polynomials/polynomial_arithmetic.hpp>

#include "../include/bbg.h"

// provided by shim/bbg_barretenberg_shim.cpp: the shim's device context and the device copy of a Pippenger point table
extern "C" bbg_ctx* bbg_shim_context(void);
extern "C" bbg_srs* bbg_shim_srs_for(const void* endo_table, size_t num_points);

namespace bbg_shim {

using barretenberg::fr;
using barretenberg::g1;

[[noreturn]] inline void resident_fail(const char* what)
{
    throw std::runtime_error(std::string(what) + ": " + bbg_last_error());
}

// Where the blinding scalars come from.  Default: os_random_fr(), the distribution of fr::random_element() (a uniform 512-bit
// string reduced mod r, field_impl.hpp:505-515) with the 64 bytes taken from the kernel CSPRNG in ONE getrandom(2) call.  The
// reference's engine draws the same bytes through 64 separate std::random_device reads per element (numeric/random/engine.cpp:9-14,
// :103-120), which measured 1.1 ms per element on the GPU box's host: 17 ms for the 15 scalars of a TurboPLONK proof that otherwise
// takes 7-33 ms.  `random` overrides it (tests replay recorded values; a caller may plug its own DRBG).
struct ResidentOptions {
    fr (*random)(void* user) = nullptr;
    void* user = nullptr;
};
inline fr os_random_fr()
{
    uint64_t w[8];
    size_t have = 0;
    while (have < sizeof(w)) {
        const ssize_t got = getrandom(reinterpret_cast<char*>(w) + have, sizeof(w) - have, 0);
        if (got < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error("bbg_shim: getrandom failed");
        }
        have += (size_t)got;
    }
    const uint512_t source(uint256_t(w[0], w[1], w[2], w[3]), uint256_t(w[4], w[5], w[6], w[7]));
    return fr((source % uint512_t(fr::modulus)).lo);
}

// PolynomialIndex (types/polynomial_manifest.hpp:9-49) -> polynomial id of include/bbg.h; -1 = not known to the device
inline int device_poly_id(waffle::PolynomialIndex index)
{
    using waffle::PolynomialIndex;
    switch (index) {
    case PolynomialIndex::W_1: return BBG_QP_W_1;
    case PolynomialIndex::W_2: return BBG_QP_W_2;
    case PolynomialIndex::W_3: return BBG_QP_W_3;
    case PolynomialIndex::W_4: return BBG_QP_W_4;
    case PolynomialIndex::Z: return BBG_QP_Z;
    case PolynomialIndex::SIGMA_1: return BBG_QP_SIGMA_1;
    case PolynomialIndex::SIGMA_2: return BBG_QP_SIGMA_2;
    case PolynomialIndex::SIGMA_3: return BBG_QP_SIGMA_3;
    case PolynomialIndex::SIGMA_4: return BBG_QP_SIGMA_4;
    case PolynomialIndex::Q_1: return BBG_QP_Q_1;
    case PolynomialIndex::Q_2: return BBG_QP_Q_2;
    case PolynomialIndex::Q_3: return BBG_QP_Q_3;
    case PolynomialIndex::Q_4: return BBG_QP_Q_4;
    case PolynomialIndex::Q_5: return BBG_QP_Q_5;
    case PolynomialIndex::Q_M: return BBG_QP_Q_M;
    case PolynomialIndex::Q_C: return BBG_QP_Q_C;
    case PolynomialIndex::Q_ARITHMETIC_SELECTOR: return BBG_QP_Q_ARITH;
    case PolynomialIndex::Q_FIXED_BASE_SELECTOR: return BBG_QP_Q_FIXED_BASE;
    case PolynomialIndex::Q_RANGE_SELECTOR: return BBG_QP_Q_RANGE;
    case PolynomialIndex::Q_LOGIC_SELECTOR: return BBG_QP_Q_LOGIC;
    case PolynomialIndex::Q_MIMC_COEFFICIENT: return BBG_PP_Q_MIMC_COEFFICIENT;
    case PolynomialIndex::Q_MIMC_SELECTOR: return BBG_PP_Q_MIMC_SELECTOR;
    default: return -1;
    }
}

// Device copy of one proving key.  Registers the coefficient forms of the key's selector and permutation polynomials (the arrays
// compute_proving_key fills: proving_key::constraint_selectors / permutation_selectors); Lagrange and 4n-coset forms are derived on
// the device (bbg_prover_finalize_key).
class ResidentKey {
  public:
    ResidentKey(std::shared_ptr<waffle::proving_key> key, size_t program_width)
        : key_(std::move(key))
        , width_(program_width)
    {
        const size_t n = key_->n;
        bbg_srs* srs = bbg_shim_srs_for(key_->reference_string->get_monomials(), n + (program_width == 3 ? 1 : 0));
        const fr gens[4] = { key_->small_domain.generator, fr::coset_generator(0), fr::coset_generator(1), fr::coset_generator(2) };
        // the widget list of round 4: a width-3 key with the MiMC selectors belongs to MiMCComposer (mimc_composer.hpp:10-26)
        const int flavour = program_width == 4 ? BBG_FLAVOUR_TURBO
                            : key_->constraint_selectors.count("q_mimc_selector") ? BBG_FLAVOUR_MIMC : BBG_FLAVOUR_STANDARD;
        if (bbg_prover_create_flavour(bbg_shim_context(), srs, (unsigned)key_->small_domain.log2_size, flavour,
                                      reinterpret_cast<const uint64_t*>(gens), &handle_) != BBG_OK)
            resident_fail("bbg_prover_create_flavour");
        for (const auto& info : key_->polynomial_manifest) {
            if (info.source == waffle::PolynomialSource::WITNESS) continue;
            const int id = device_poly_id(info.index);
            const std::string label(info.polynomial_label);
            if (id < 0) {
                bbg_prover_destroy(handle_);
                throw std::runtime_error("bbg_shim::ResidentKey: polynomial '" + label + "' has no device counterpart");
            }
            const barretenberg::polynomial& poly = info.source == waffle::PolynomialSource::SELECTOR ? key_->constraint_selectors.at(label)
                                                                                                       : key_->permutation_selectors.at(label);
            if (bbg_prover_set_key_poly(handle_, id, BBG_FORM_COEFF, reinterpret_cast<const uint64_t*>(&poly[0])) != BBG_OK) {
                bbg_prover_destroy(handle_);
                resident_fail("bbg_prover_set_key_poly");
            }
        }
        if (bbg_prover_finalize_key(handle_) != BBG_OK) {
            bbg_prover_destroy(handle_);
            resident_fail("bbg_prover_finalize_key");
        }
    }
    ~ResidentKey() { bbg_prover_destroy(handle_); }
    ResidentKey(const ResidentKey&) = delete;
    ResidentKey& operator=(const ResidentKey&) = delete;

    bbg_prover* handle() const { return handle_; }
    const std::shared_ptr<waffle::proving_key>& key() const { return key_; }
    size_t program_width() const { return width_; }

    // The device state of a proof in progress (wires, z, quotient, the round reached) belongs to the KEY's device copy, and a host may begin
    // a second proof over the same key between two rounds of the first (another prover's construct_proof(), another round-by-round proof).
    // Each ResidentProof claims the key when it begins (its preamble) and every later round checks that it still holds the claim: a proof
    // whose device state was overwritten fails with an exception (the wrap then repeats it on the reference rounds) instead of going on
    // over the other proof's polynomials (round-5 advisor finding).
    uint64_t claim() { return ++claim_; }
    bool claimed_by(uint64_t token) const { return claim_ == token; }

  private:
    std::shared_ptr<waffle::proving_key> key_;
    size_t width_;
    bbg_prover* handle_ = nullptr;
    uint64_t claim_ = 0;
};

namespace detail {

// scalars[id] += c.  (barretenberg's field has a user-provided default constructor that leaves the limbs UNINITIALISED, so
// std::map::operator[] must not be used to create an entry.)
inline void accumulate(std::map<int, fr>& scalars, int id, const fr& c)
{
    auto it = scalars.find(id);
    if (it == scalars.end()) scalars.emplace(id, c);
    else it->second += c;
}

template <typename T> const uint64_t* limbs(const T& v)
{
    return reinterpret_cast<const uint64_t*>(&v);
}
inline void add_commitment(transcript::StandardTranscript& transcript, const std::string& tag, const g1::element& point)
{
    // what work_queue::process_queue does with an MSM result: normalise, serialise, add (work_queue.hpp:233-239)
    transcript.add_element(tag, g1::affine_element(point).to_buffer());
}

// One transition widget's share of the linearisation polynomial: r(X) += sum_selectors c_sel * q_sel(X).  The scalars are taken
// from the reference's own kernel templates: linear_terms by EvaluationKernel::compute_linear_terms (transition_widget.hpp:317-326),
// and c_sel by evaluating MonomialKernel::sum_linear_terms -- a linear form in the selector values at one index -- on the unit
// vectors (every selector 0, one selector 1).  No widget formula is restated here.
template <typename Widget, typename Prover>
bool linear_scalars_of(waffle::widget::TransitionWidgetBase<fr>* base, Prover& p, fr& alpha_base, std::map<int, fr>& scalars)
{
    if (dynamic_cast<Widget*>(base) == nullptr) return false;
    using MonomialGetter = typename Widget::MonomialGetter;
    using EvaluationGetter = typename Widget::EvaluationGetter;
    using MonomialKernel = typename Widget::MonomialKernel;
    using EvaluationKernel = typename Widget::EvaluationKernel;
    using FFTKernel = typename Widget::FFTKernel;
    auto challenges = MonomialGetter::get_challenges(p.transcript, alpha_base);
    auto evaluations = EvaluationGetter::get_polynomial_evaluations(p.key->polynomial_manifest, p.transcript);
    waffle::widget::containers::coefficient_array<fr> linear_terms{};
    EvaluationKernel::compute_linear_terms(evaluations, challenges, linear_terms);
    fr zero = fr::zero(), one = fr::one();
    waffle::widget::containers::poly_ptr_array<fr> probe;
    probe.block_mask = 0;
    for (auto& ptr : probe.coefficients) ptr = &zero;
    if (!(MonomialKernel::sum_linear_terms(probe, challenges, linear_terms, 0) == fr::zero()))
        throw std::runtime_error("bbg_shim: a widget's linear contribution is not a linear form in the selectors");
    for (const auto& info : p.key->polynomial_manifest) {
        if (info.source != waffle::PolynomialSource::SELECTOR) continue;
        probe.coefficients[info.index] = &one;
        const fr c = MonomialKernel::sum_linear_terms(probe, challenges, linear_terms, 0);
        probe.coefficients[info.index] = &zero;
        if (c == fr::zero()) continue;
        accumulate(scalars, device_poly_id(info.index), c);
    }
    alpha_base = MonomialGetter::update_alpha(challenges, FFTKernel::num_independent_relations);
    return true;
}

// The permutation widget's share: r(X) = m_z z(X) + m_sigma sigma_last(X); m_z and m_sigma are functions of the opening evaluations
// and the challenges (ProverPermutationWidget::compute_linear_contribution steps 1-4, permutation_widget_impl.hpp:505-586; the O(n)
// step 5 is the device's).  Returns the alpha_base for the next widget (alpha^4, :588).
template <size_t program_width, typename Prover> fr permutation_linear_scalars(Prover& p, const fr& alpha, std::map<int, fr>& scalars)
{
    const auto& t = p.transcript;
    const fr zeta = fr::serialize_from_buffer(t.get_challenge("z").begin());
    const fr beta = fr::serialize_from_buffer(t.get_challenge("beta").begin());
    const fr gamma = fr::serialize_from_buffer(t.get_challenge("beta", 1).begin());
    const auto lagrange = barretenberg::polynomial_arithmetic::get_lagrange_evaluations(zeta, p.key->small_domain);
    const fr zeta_beta = zeta * beta;
    std::array<fr, program_width> w;
    for (size_t i = 0; i < program_width; ++i) w[i] = fr::serialize_from_buffer(&t.get_element("w_" + std::to_string(i + 1))[0]);
    const fr z_omega = fr::serialize_from_buffer(&t.get_element("z_omega")[0]);
    fr z_term = fr::one(); // prod_i (w_i + beta K_i zeta + gamma), K_0 = 1, K_i = coset_generator(i - 1)
    for (size_t i = 0; i < program_width; ++i) z_term *= (w[i] + gamma + zeta_beta * (i == 0 ? fr::one() : fr::coset_generator(i - 1)));
    const fr m_z = z_term * alpha + lagrange.l_start * (alpha.sqr() * alpha);
    fr sigma_term = z_omega; // z(zeta w) prod_{i < width-1} (w_i + beta sigma_i(zeta) + gamma)
    for (size_t i = 0; i + 1 < program_width; ++i)
        sigma_term *= (w[i] + gamma + beta * fr::serialize_from_buffer(&t.get_element("sigma_" + std::to_string(i + 1))[0]));
    const fr m_sigma = -(sigma_term * alpha) * beta;
    accumulate(scalars, BBG_QP_Z, m_z);
    accumulate(scalars, BBG_QP_SIGMA_1 + (int)program_width - 1, m_sigma);
    return alpha.sqr().sqr();
}

} // namespace detail

// Is this prover one of the three flavours the device rounds implement (widget lists of TurboComposer::create_prover,
// turbo_composer.cpp:735-752, StandardComposer::create_prover, standard_composer.cpp:562-582, and MiMCComposer::preprocess,
// mimc_composer.cpp:277-302) -- i.e. every prover the reference's composers build?
template <typename settings> bool resident_supported(waffle::ProverBase<settings>& p)
{
    using namespace waffle;
    constexpr size_t W = settings::program_width;
    if (settings::uses_quotient_mid || p.random_widgets.size() != 1) return false; // linearised and "unrolled" (all polynomials opened) alike
    if constexpr (W == 4) {
        if (!dynamic_cast<ProverPermutationWidget<4, false>*>(p.random_widgets[0].get()) || p.transition_widgets.size() != 4) return false;
        return dynamic_cast<ProverTurboArithmeticWidget<settings>*>(p.transition_widgets[0].get()) &&
               dynamic_cast<ProverTurboFixedBaseWidget<settings>*>(p.transition_widgets[1].get()) &&
               dynamic_cast<ProverTurboRangeWidget<settings>*>(p.transition_widgets[2].get()) &&
               dynamic_cast<ProverTurboLogicWidget<settings>*>(p.transition_widgets[3].get());
    } else if constexpr (W == 3) {
        if (!dynamic_cast<ProverPermutationWidget<3, false>*>(p.random_widgets[0].get())) return false;
        if (p.transition_widgets.size() == 1) return dynamic_cast<ProverArithmeticWidget<settings>*>(p.transition_widgets[0].get()) != nullptr;
        return p.transition_widgets.size() == 2 && dynamic_cast<ProverMiMCWidget<settings>*>(p.transition_widgets[0].get()) &&
               dynamic_cast<ProverArithmeticWidget<settings>*>(p.transition_widgets[1].get()) &&
               p.key->constraint_selectors.count("q_mimc_selector");
    }
    return false;
}

// One proof in progress on the device rounds, ROUND BY ROUND -- the seven steps ProverBase<settings>::construct_proof() is made of
// (prover.cpp:420-436) and a host may also drive one at a time through the reference's C binding (prover_execute_preamble_round ...
// prover_execute_sixth_round + prover_process_queue, plonk/proof_system/prover/c_bind.cpp:59-92).  Each step does what the reference round of
// the same name does to the transcript AND what the following process_queue() would: its commitments are in the transcript when it returns,
// the work queue stays empty.  `p` must have been created over rk.key() (same proving key).  Steps must be called in order, once each.
template <typename settings> class ResidentProof {
  public:
    static constexpr size_t W = settings::program_width;
    static constexpr size_t CUT = settings::num_roots_cut_out_of_vanishing_polynomial;

    ResidentProof(waffle::ProverBase<settings>& prover, ResidentKey& rk, const ResidentOptions& options = ResidentOptions())
        : p(prover)
        , opt(options)
        , rkey(rk)
        , dev(rk.handle())
        , key(prover.key.get())
        , witness(prover.witness.get())
        , n(prover.key->n)
    {
        if (p.key.get() != rk.key().get() || rk.program_width() != W)
            throw std::runtime_error("bbg_shim::construct_proof: prover and ResidentKey belong to different proving keys");
        if (!resident_supported(p))
            throw std::runtime_error("bbg_shim::construct_proof: unsupported widget set (TurboPLONK / StandardPLONK / MiMC composer provers only)");
    }

    // execute_preamble_round (prover.cpp:136-190): sizes, "init", three blinding scalars per wire in rows n-4 .. n-2 of its Lagrange form
    void preamble()
    {
        auto& transcript = p.transcript;
        p.queue.flush_queue();
        transcript.add_element("circuit_size", { static_cast<uint8_t>(n >> 24), static_cast<uint8_t>(n >> 16), static_cast<uint8_t>(n >> 8),
                                                 static_cast<uint8_t>(n) });
        transcript.add_element("public_input_size",
                               { static_cast<uint8_t>(key->num_public_inputs >> 24), static_cast<uint8_t>(key->num_public_inputs >> 16),
                                 static_cast<uint8_t>(key->num_public_inputs >> 8), static_cast<uint8_t>(key->num_public_inputs) });
        transcript.apply_fiat_shamir("init");
        for (size_t i = 0; i < W; ++i) {
            barretenberg::polynomial& wire = witness->wires.at("w_" + std::to_string(i + 1));
            for (size_t k = 0; k < 3; ++k) wire.at(n - CUT + k) = random();
            wire_ptrs[i] = reinterpret_cast<const uint64_t*>(&wire[0]);
        }
    }
    // execute_first_round (:192-222, :66-84): the public inputs are rows of w_2 in Lagrange form; W_1 .. W_w
    void first()
    {
        auto& transcript = p.transcript;
        {
            const barretenberg::polynomial& w_2 = witness->wires.at("w_2");
            std::vector<fr> public_wires;
            for (size_t i = 0; i < key->num_public_inputs; ++i) public_wires.push_back(w_2[i]);
            transcript.add_element("public_inputs", ::to_buffer(public_wires));
        }
        if (bbg_prover_round1(dev, wire_ptrs, reinterpret_cast<uint64_t*>(commitments)) != BBG_OK) resident_fail("bbg_prover_round1");
        for (size_t i = 0; i < W; ++i) detail::add_commitment(transcript, "W_" + std::to_string(i + 1), commitments[i]);
    }
    // execute_second_round (:224-231): nothing to commit for these flavours
    void second() { p.transcript.apply_fiat_shamir("eta"); }
    // execute_third_round (:233-268): z
    void third()
    {
        auto& transcript = p.transcript;
        transcript.apply_fiat_shamir("beta");
        beta = fr::serialize_from_buffer(transcript.get_challenge("beta").begin());
        gamma = fr::serialize_from_buffer(transcript.get_challenge("beta", 1).begin());
        const fr blind[3] = { random(), random(), random() }; // rows n-3 .. n-1 of z (permutation_widget_impl.hpp:283-287)
        if (bbg_prover_round3(dev, detail::limbs(beta), detail::limbs(gamma), reinterpret_cast<const uint64_t*>(blind),
                              reinterpret_cast<uint64_t*>(commitments)) != BBG_OK)
            resident_fail("bbg_prover_round3");
        detail::add_commitment(transcript, "Z", commitments[0]);
    }
    // execute_fourth_round (:270-363): the quotient
    void fourth()
    {
        auto& transcript = p.transcript;
        transcript.apply_fiat_shamir("alpha");
        alpha = fr::serialize_from_buffer(transcript.get_challenge("alpha").begin());
        const std::vector<fr> public_inputs = many_from_buffer<fr>(transcript.get_element("public_inputs"));
        const fr delta = waffle::compute_public_input_delta<fr>(public_inputs, beta, gamma, key->small_domain.root);
        if (bbg_prover_round4(dev, detail::limbs(alpha), detail::limbs(delta), reinterpret_cast<uint64_t*>(commitments)) != BBG_OK)
            resident_fail("bbg_prover_round4");
        for (size_t i = 0; i < W; ++i) detail::add_commitment(transcript, "T_" + std::to_string(i + 1), commitments[i]);
    }
    // execute_fifth_round (:365-410): opening evaluations in manifest order (kate_commitment_scheme.cpp:362-420), t(zeta), r(X), r(zeta)
    void fifth()
    {
        using namespace waffle;
        auto& transcript = p.transcript;
        transcript.apply_fiat_shamir("z");
        zeta = fr::serialize_from_buffer(transcript.get_challenge("z").begin());
        std::vector<int> ids, shifted;
        std::vector<std::string> labels;
        for (const auto& info : key->polynomial_manifest) {
            const std::string label(info.polynomial_label);
            // the unrolled provers (use_linearisation = false: what the rollup circuits use, rollup/proofs/*) evaluate EVERY polynomial
            if (!info.is_linearised || !settings::use_linearisation) { ids.push_back(device_poly_id(info.index)); shifted.push_back(0); labels.push_back(label); }
            if (info.requires_shifted_evaluation) { ids.push_back(device_poly_id(info.index)); shifted.push_back(1); labels.push_back(label + "_omega"); }
        }
        ids.push_back(BBG_PP_QUOTIENT); // t_eval = quotient_large.evaluate(zeta, 4n) (prover.cpp:397)
        shifted.push_back(0);
        std::vector<fr> evals(ids.size());
        if (bbg_prover_evaluate(dev, ids.size(), ids.data(), shifted.data(), detail::limbs(zeta), reinterpret_cast<uint64_t*>(evals.data())) != BBG_OK)
            resident_fail("bbg_prover_evaluate");
        for (size_t k = 0; k < labels.size(); ++k) transcript.add_element(labels[k], evals[k].to_buffer());
        const fr t_eval = evals.back();
        if constexpr (settings::use_linearisation) {
            // r(X): the permutation widget assigns, the transition widgets accumulate (:399-405)
            std::map<int, fr> scalars;
            fr alpha_base = detail::permutation_linear_scalars<W>(p, alpha, scalars);
            for (auto& widget : p.transition_widgets) {
                auto* w = widget.get();
                bool ok;
                if constexpr (W == 4)
                    ok = detail::linear_scalars_of<ProverTurboArithmeticWidget<settings>>(w, p, alpha_base, scalars) ||
                         detail::linear_scalars_of<ProverTurboFixedBaseWidget<settings>>(w, p, alpha_base, scalars) ||
                         detail::linear_scalars_of<ProverTurboRangeWidget<settings>>(w, p, alpha_base, scalars) ||
                         detail::linear_scalars_of<ProverTurboLogicWidget<settings>>(w, p, alpha_base, scalars);
                else
                    ok = detail::linear_scalars_of<ProverArithmeticWidget<settings>>(w, p, alpha_base, scalars) ||
                         detail::linear_scalars_of<ProverMiMCWidget<settings>>(w, p, alpha_base, scalars);
                if (!ok) throw std::runtime_error("bbg_shim::construct_proof: unknown transition widget");
            }
            std::vector<int> lin_ids;
            std::vector<fr> lin_scalars;
            for (const auto& kv : scalars) { lin_ids.push_back(kv.first); lin_scalars.push_back(kv.second); }
            fr r_eval;
            if (bbg_prover_linearise(dev, lin_ids.size(), lin_ids.data(), reinterpret_cast<const uint64_t*>(lin_scalars.data()), detail::limbs(zeta),
                                     reinterpret_cast<uint64_t*>(&r_eval)) != BBG_OK)
                resident_fail("bbg_prover_linearise");
            transcript.add_element("r", r_eval.to_buffer());
        }
        transcript.add_element("t", t_eval.to_buffer());
    }
    // execute_sixth_round (:412-418, KateCommitmentScheme::batch_open kate_commitment_scheme.cpp:133-236)
    void sixth()
    {
        auto& transcript = p.transcript;
        transcript.apply_fiat_shamir("nu");
        std::vector<int> at_zeta, at_omega;
        std::vector<fr> nu_zeta, nu_omega;
        for (const auto& info : key->polynomial_manifest) {
            const std::string label(info.polynomial_label);
            if (!info.is_linearised || !settings::use_linearisation) {
                at_zeta.push_back(device_poly_id(info.index));
                nu_zeta.push_back(transcript.get_challenge_field_element_from_map("nu", label));
            }
            if (info.requires_shifted_evaluation) {
                at_omega.push_back(device_poly_id(info.index));
                nu_omega.push_back(transcript.get_challenge_field_element_from_map("nu", label + "_omega"));
            }
        }
        fr top_scalar = fr::zero();
        for (size_t i = 1; i < W; ++i) { // t_mid, t_high (, t_higher) with zeta^(i n); t_low is the base of the sum
            const fr scalar = zeta.pow(static_cast<uint64_t>(i * n));
            at_zeta.push_back(BBG_PP_T_1 + (int)i);
            nu_zeta.push_back(scalar);
            if (i == W - 1) top_scalar = scalar;
        }
        if constexpr (settings::use_linearisation) { // [r(X), nu_r] only where the proof system has a linearisation polynomial (:211-215)
            at_zeta.push_back(BBG_PP_LINEAR);
            nu_zeta.push_back(transcript.get_challenge_field_element_from_map("nu", "r"));
        }
        const fr zeta_omega = zeta * key->small_domain.root;
        if (bbg_prover_round6(dev, at_zeta.size(), at_zeta.data(), reinterpret_cast<const uint64_t*>(nu_zeta.data()), at_omega.size(), at_omega.data(),
                              reinterpret_cast<const uint64_t*>(nu_omega.data()), detail::limbs(zeta), detail::limbs(zeta_omega),
                              W == 3 ? detail::limbs(top_scalar) : nullptr, reinterpret_cast<uint64_t*>(&commitments[0]),
                              reinterpret_cast<uint64_t*>(&commitments[1])) != BBG_OK)
            resident_fail("bbg_prover_round6");
        detail::add_commitment(transcript, "PI_Z", commitments[0]);
        detail::add_commitment(transcript, "PI_Z_OMEGA", commitments[1]);
    }
    // step k of the seven (0 = preamble ... 6 = sixth round)
    void step(int k)
    {
        if (k == 0) token = rkey.claim();
        else if (!rkey.claimed_by(token))
            throw std::runtime_error("bbg_shim::ResidentProof: another proof over the same proving key began between two rounds of this one");
        switch (k) {
        case 0: preamble(); break;
        case 1: first(); break;
        case 2: second(); break;
        case 3: third(); break;
        case 4: fourth(); break;
        case 5: fifth(); break;
        case 6: sixth(); break;
        default: throw std::runtime_error("bbg_shim::ResidentProof: no such round");
        }
    }

  private:
    fr random() { return opt.random ? opt.random(opt.user) : os_random_fr(); }
    waffle::ProverBase<settings>& p;
    ResidentOptions opt;
    ResidentKey& rkey;
    uint64_t token = 0;
    bbg_prover* dev;
    waffle::proving_key* key;
    waffle::program_witness* witness;
    size_t n;
    const uint64_t* wire_ptrs[4] = { nullptr, nullptr, nullptr, nullptr };
    g1::element commitments[4];
    fr beta, gamma, alpha, zeta;
};

// ProverBase::construct_proof() with every O(n) step on the device.  `p` must have been created over rk.key() (same proving key).
template <typename settings>
waffle::plonk_proof& construct_proof(waffle::ProverBase<settings>& p, ResidentKey& rk, const ResidentOptions& opt = ResidentOptions())
{
    ResidentProof<settings> proof(p, rk, opt);
    for (int k = 0; k <= 6; k++) proof.step(k);
    return p.export_proof();
}

} // namespace bbg_shim
