"""CPU tests: pin the oracle (oracle/bn254_oracle.c) to the reference.

 * against tests/golden/golden.json -- outputs of the REAL reference compiled from /root/reference
   (tests/golden/gen_golden.py), and
 * against the known-answer constants in the reference's own unit tests (tests/golden/reference_kats.json), and
 * where oracle/_ref/libbbref.so is present, live against the reference on fresh inputs.
"""
import numpy as np
import pytest

from conftest import limbs, sha, unhex


# ------------------------------------------------------------------ reference unit-test constants
@pytest.mark.parametrize("name,which,op", [("fr_mul", 0, "mul"), ("fr_sqr", 0, "sqr"), ("fr_add", 0, "add"), ("fr_sub", 0, "sub"),
                                           ("fq_mul", 1, "mul"), ("fq_mul_short", 1, "mul"), ("fq_sqr", 1, "sqr"),
                                           ("fq_add", 1, "add"), ("fq_sub", 1, "sub")])
def test_field_kats(oracle, kats, name, which, op):
    k = kats[name]
    a = limbs(k["a"])
    b = limbs(k["b"]) if "b" in k else a
    fn = {"mul": oracle.fe_mul, "sqr": oracle.fe_mul, "add": oracle.fe_add, "sub": oracle.fe_sub}[op]
    got = fn(which, a, b)[0]
    want = oracle.canon(which, limbs(k["expected"]))[0]
    assert np.array_equal(got, want), k["cite"]


def _jac(oracle, k, p):
    j = np.concatenate([oracle.to_mont(1, limbs(k[p + "_x"]))[0], oracle.to_mont(1, limbs(k[p + "_y"]))[0],
                        oracle.to_mont(1, limbs(k[p + "_z"]))[0]])
    return oracle.jac_to_affine(j)


def test_g1_kats(oracle, kats):
    k = kats["g1_mixed_add"]
    rhs = np.concatenate([oracle.to_mont(1, limbs(k["b_x"]))[0], oracle.to_mont(1, limbs(k["b_y"]))[0]])
    assert np.array_equal(oracle.g1_add(_jac(oracle, k, "a"), rhs), _jac(oracle, k, "expected")), k["cite"]
    k = kats["g1_add"]
    assert np.array_equal(oracle.g1_add(_jac(oracle, k, "a"), _jac(oracle, k, "b")), _jac(oracle, k, "expected")), k["cite"]
    k = kats["g1_dbl_x3"]
    p = _jac(oracle, k, "a")
    for _ in range(3):
        p = oracle.g1_add(p, p)
    assert np.array_equal(p, _jac(oracle, k, "expected")), k["cite"]


# ------------------------------------------------------------------ golden vectors from the compiled reference
def test_field_golden(oracle, golden):
    for which, name in ((0, "fr"), (1, "fq")):
        g = golden["field"][name]
        a, b = unhex(g["a"]), unhex(g["b"])
        assert np.array_equal(oracle.fe_mul(which, a, b), unhex(g["mul"]))
        assert np.array_equal(oracle.fe_add(which, a, b), unhex(g["add"]))
        assert np.array_equal(oracle.fe_sub(which, a, b), unhex(g["sub"]))
        assert np.array_equal(oracle.fe_inv(which, a), unhex(g["inv"]))
        assert np.array_equal(oracle.to_mont(which, a), unhex(g["to_mont"]))
        assert np.array_equal(oracle.from_mont(which, a), unhex(g["from_mont"]))
        assert np.array_equal(oracle.fe_mul(which, a, a), unhex(g["sqr"]))


def test_roots_golden(oracle, golden):
    for k, v in golden["roots_of_unity"].items():
        assert np.array_equal(oracle.root_of_unity(int(k)), unhex(v)[0])
    five = oracle.to_mont(0, np.array([5, 0, 0, 0], dtype=np.uint64))[0]
    assert np.array_equal(five, unhex(golden["coset_generator"])[0])


def test_endo_and_wnaf_golden(pkg, oracle, golden):
    g = golden["endo_split"]
    sc = pkg.synthetic_scalars(g["seed"], g["n"])
    assert np.array_equal(oracle.endo_split(sc), unhex(g["out"]))
    for key, scal in (("wnaf", sc), ("wnaf_mixed", pkg.inputs.mixed_scalars(golden["wnaf_mixed"]["seed"], 64, lambda p: oracle.to_mont(0, p)))):
        w = golden[key]
        sched, skew, counts = oracle.wnaf_schedule(scal, w["wnaf_bits"])
        assert sha(sched) == w["schedule_sha256"]
        assert bytes(skew).hex() == w["skew"]
        assert [int(x) for x in counts] == w["round_counts"]
    assert np.array_equal(oracle.wnaf_schedule(sc, golden["wnaf"]["wnaf_bits"])[0][0, :16], unhex(golden["wnaf"]["first_round"], 16)[0])


def test_endo_identity(pkg, oracle):
    """k = k1 - k2*lambda (mod r): the identity the reference checks in fr.test.cpp:254-296 / scalar_multiplication.test.cpp:502-526."""
    r = int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 16)
    lam = int("b3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd", 16)
    sc = pkg.synthetic_scalars(4242, 50)
    plain = oracle.from_mont(0, sc)
    out = oracle.endo_split(sc)
    for i in range(50):
        k = sum(int(plain[i, j]) << (64 * j) for j in range(4))
        k1 = int(out[i, 0]) | (int(out[i, 1]) << 64)
        k2 = int(out[i, 2]) | (int(out[i, 3]) << 64)
        assert (k1 - k2 * lam - k) % r == 0
        assert k1 < (1 << 128) and k2 < (1 << 128)


def test_group_golden(oracle, golden):
    g = golden["group"]
    gen = unhex(g["generator"], 8)[0]
    assert np.array_equal(oracle.g1_generator(), gen)
    k = unhex(g["k"])
    P, Q = unhex(g["P"], 8)[0], unhex(g["Q"], 8)[0]
    assert np.array_equal(oracle.g1_mul(gen, k[0]), P)
    assert np.array_equal(oracle.g1_mul(gen, k[1]), Q)
    assert np.array_equal(oracle.g1_add(P, Q), unhex(g["P_plus_Q"], 8)[0])
    assert np.array_equal(oracle.g1_add(P, P), unhex(g["two_P"], 8)[0])
    assert np.array_equal(oracle.g1_mul(P, k[2]), unhex(g["kP3"], 8)[0])
    assert oracle.g1_to_buffer(P).hex() == g["P_buffer"]
    assert oracle.g1_on_curve(P) and oracle.g1_on_curve(Q)
    t = golden["point_table"]
    pts = oracle.srs_hashed(t["seed"], t["n"])
    assert sha(pts) == t["points_sha256"]
    assert np.array_equal(oracle.point_table(pts), unhex(t["table"], 8))


def _msm_inputs(pkg, oracle, rec, cache):
    n, frm = rec["n"], rec["from"]
    if rec["srs"] == "hashed":
        key = ("h", rec["srs_seed"])
        need = frm + n
        if key not in cache or cache[key].shape[0] < need:
            cache[key] = oracle.srs_hashed(rec["srs_seed"], need)
        pts = cache[key][frm:frm + n]
    elif rec["srs"] == "linear":
        pts = oracle.srs_linear(rec["a"], rec["s"], frm + n)[frm:]
    else:
        pts = np.tile(oracle.srs_hashed(rec["srs_seed"], 1), (n, 1))
    if rec.get("scalar_kind") == "mixed":
        sc = pkg.inputs.mixed_scalars(rec["scalar_seed"], n, lambda p: oracle.to_mont(0, p))
    else:
        sc = pkg.synthetic_scalars(rec["scalar_seed"], n)  # the generator is index-based, so prefixes agree
    return sc, pts


def test_msm_golden(pkg, oracle, golden):
    """oracle_pippenger == reference pippenger / pippenger_unsafe on every recorded case up to 2^16 (2^20 is in the gpu tests)."""
    cache = {}
    for rec in golden["msm"]:
        if rec["n"] > (1 << 16) + 1:
            continue
        sc, pts = _msm_inputs(pkg, oracle, rec, cache)
        assert np.array_equal(oracle.pippenger(sc, pts), unhex(rec["result"], 8)[0]), rec
    assert sha(cache[("h", golden["seed_base"])][:4096]) == golden["msm_points_sha256"]["hashed_2^12"]
    assert sha(cache[("h", golden["seed_base"])][:1 << 16]) == golden["msm_points_sha256"]["hashed_2^16"]


def test_msm_naive_equals_bucket(pkg, oracle):
    """the reference tests' own expectation: bucket MSM == naive sum_i s_i P_i (scalar_multiplication.test.cpp:655-686)."""
    pts = oracle.srs_hashed(7, 300)
    sc = pkg.synthetic_scalars(8, 300)
    assert np.array_equal(oracle.pippenger(sc, pts), oracle.msm_naive(sc, pts))
    inf = oracle.pippenger(sc[:0], pts[:0])
    assert int(inf[3]) >> 63 == 1  # pippenger_zero_points (:895-908)
    zero = np.zeros((5, 4), dtype=np.uint64)
    assert int(oracle.pippenger(zero, pts[:5])[3]) >> 63 == 1  # pippenger_mul_by_zero (:910-927)


def test_ntt_golden(pkg, oracle, golden):
    kc = unhex(golden["ntt_constant"])[0]
    for rec in golden["ntt"]:
        if rec["log2n"] > 16:
            continue
        c = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2n"])
        out = oracle.ntt(c, rec["op"], rec["generator_size"], kc if rec["op"] >= 4 else None)
        assert sha(out) == rec["sha256"], rec
        if "out" in rec:
            assert np.array_equal(out, unhex(rec["out"]))
    for rec in golden["coset_fft_split"]:
        c = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2n"])
        assert sha(oracle.coset_fft_split(c, rec["ext"])) == rec["sha256"], rec
    pe = golden["poly_eval"]
    c = pkg.synthetic_scalars(pe["seed"], pe["n"])
    assert np.array_equal(oracle.poly_eval(c, unhex(pe["z"])[0]), unhex(pe["value"])[0])


def test_fft_matches_horner(pkg, oracle):
    """fft_with_small_degree (polynomial_arithmetic.test.cpp:45-68): FFT output i == evaluate(poly, omega^i); pins ordering and omega."""
    n = 16
    c = pkg.synthetic_scalars(31337, n)
    out = oracle.ntt(c, 0)
    w = oracle.root_of_unity(4)
    z = oracle.to_mont(0, np.array([1, 0, 0, 0], dtype=np.uint64))[0]
    for i in range(n):
        assert np.array_equal(oracle.poly_eval(c, z), out[i])
        z = oracle.fe_mul(0, z, w)[0]


def test_ntt_roundtrips(pkg, oracle):
    """basic_fft / fft_ifft_consistency / fft_coset_ifft_consistency (polynomial_arithmetic.test.cpp:70-134)."""
    c = pkg.synthetic_scalars(55, 1 << 10)
    canon = oracle.canon(0, c)
    assert np.array_equal(oracle.ntt(oracle.ntt(c, 0), 1), canon)
    assert np.array_equal(oracle.ntt(oracle.ntt(c, 2), 3), canon)


# ------------------------------------------------------------------ live against the reference build (this container)
def test_live_against_reference(pkg, oracle, ref):
    a = oracle.canon(0, pkg.synthetic_scalars(901, 200))
    b = oracle.canon(0, pkg.synthetic_scalars(902, 200))
    for which in (0, 1):
        ca, cb = oracle.canon(which, a), oracle.canon(which, b)
        assert np.array_equal(oracle.fe_mul(which, ca, cb), ref.fe_op(which, 0, ca, cb))
        assert np.array_equal(oracle.fe_sub(which, ca, cb), ref.fe_op(which, 2, ca, cb))
    pts = oracle.srs_hashed(903, 777)
    sc = pkg.synthetic_scalars(904, 777)
    m = ref.msm(pts)
    for nn in (0, 1, 63, 64, 65, 500, 777):
        want, _ = m.run(sc[:nn])
        assert np.array_equal(oracle.pippenger(sc[:nn], pts[:nn]), want), nn
    m.free()
    for lg in (1, 6, 10):
        c = pkg.synthetic_scalars(905 + lg, 1 << lg)
        d = ref.domain(lg, 0)
        for op in range(4):
            want, _ = d.run(c, op)
            assert np.array_equal(oracle.ntt(c, op), want), (lg, op)
        d.free()


def test_poly_helpers_golden(pkg, oracle, golden):
    """add/sub/mul, compute_kate_opening_coefficients, divide_by_pseudo_vanishing_polynomial vs the compiled reference."""
    kc = unhex(golden["ntt_constant"])[0]
    for rec in golden["poly"]["binop"]:
        a = pkg.synthetic_scalars(rec["seed_a"], 1 << rec["log2n"])
        b = pkg.synthetic_scalars(rec["seed_b"], 1 << rec["log2n"])
        assert sha(oracle.poly_binop(rec["op"], a, b)) == rec["sha256"], rec
    for rec in golden["poly"]["kate"]:
        a = pkg.synthetic_scalars(rec["seed"], rec["n"])
        dest, f = oracle.kate_opening(a, kc)
        assert np.array_equal(f, unhex(rec["f"])[0]) and sha(dest) == rec["dest_sha256"], rec
    for rec in golden["poly"]["dpv"]:
        e = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2_target"])
        assert sha(oracle.divide_by_pseudo_vanishing(e, rec["log2_src"], rec["cut"])) == rec["sha256"], rec


# ---------------------------------------------------------------------------------------------- the reference PROVER seam
def test_reference_prover_with_oracle_engine(oracle):
    """The reference's real TurboPLONK prover (TurboComposer circuit -> TurboProver rounds, oracle/ref_prover_driver.cpp) with
    every MSM / coset-FFT / iFFT work item of work_queue::process_queue (work_queue.hpp:208-282) computed by THIS repo's
    CPU oracle instead: each item must equal the reference's CPU result bit for bit, and the reference's TurboVerifier must
    accept the resulting proof.  Pins the oracle on the data a real proof produces (blinded wires, quotient parts, opening
    polynomials), not only on synthetic vectors."""
    from oracle.oracle import RefProver, prover_available
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent (built from /root/reference by `make -C oracle prover`)")
    O = oracle
    x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    pts = O.srs_powers(x, (1 << 9) + 1)

    class Engine:
        def __init__(self, mon):
            self.mon = mon

        def msm(self, s):
            j = np.zeros(12, dtype=np.uint64)
            j[:8] = O.pippenger(s, self.mon[:s.shape[0]])
            j[8:12] = O.to_mont(1, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
            return j

        def coset_fft(self, a, generator_size):
            return O.ntt(a, 2, generator_size)

        def ifft(self, a):
            return O.ntt(a, 1)

    P = RefProver(1 << 8, 7, pts, x)
    assert P.n == 1 << 9
    assert len(P.prove()) > 0 and P.verify() == 1  # baseline: the reference on its own
    P.free()
    P = RefProver(1 << 8, 7, pts, x)
    mon = P.monomials()
    assert np.array_equal(mon, pts[: P.n + 1])
    proof = P.prove(Engine(mon))
    assert P.mismatches == 0
    assert P.counts[0] >= 9 and P.counts[1] >= 4 and P.counts[2] >= 3, P.counts
    assert len(proof) > 0 and P.verify() == 1
    P.free()

    class FusedEngine(Engine):  # the FFT work item as one call: n coefficients -> 4n + 4 values (copy, coset FFT, 4 wrapped)
        def fft_item(self, wire, log2_domain):
            m = 1 << log2_domain
            a = np.zeros((m, 4), dtype=np.uint64)
            a[: wire.shape[0]] = wire
            r = O.ntt(a, 2, wire.shape[0])
            return np.concatenate([r, r[:4]])

    P = RefProver(1 << 8, 7, pts, x)
    proof = P.prove(FusedEngine(P.monomials()))
    assert P.mismatches == 0 and len(proof) > 0 and P.verify() == 1
    P.free()


def test_quotient_widgets_oracle_vs_reference_golden(oracle):
    """The oracle's restatement of the five TurboPLONK quotient widgets (oracle_quotient_widget) against the digests recorded
    from the reference's own widget objects (tests/golden/widgets.json, gen_golden_widgets.py): same seeded inputs, the
    prover's widget order, quotient array and returned alpha_base after each widget."""
    import json
    import os
    from oracle.oracle import RefWidgets
    import __graft_entry__ as ge
    pkg = ge.load_package()
    with open(os.path.join(os.path.dirname(__file__), "golden", "widgets.json")) as f:
        G = json.load(f)
    for case in G["cases"] + G["standard_cases"]:  # TurboPLONK widgets 0..4, then StandardPLONK (three wires) widgets 5, 6
        log2_large = case["log2n"] + 2
        m = 1 << log2_large
        c = case["challenges"]
        ch9 = np.stack([unhex(c[k], 4)[0] for k in ("alpha", "alpha", "beta", "gamma", "public_input_delta", "g", "k1", "k2", "k3")])
        polys = [pkg.synthetic_scalars(G["seed"] + k, m) for k in range(len(RefWidgets.LABELS))]
        quot = np.zeros((m, 4), dtype=np.uint64)
        alpha_base = unhex(c["alpha"], 4)[0]
        for rec in case["widgets"]:
            ch = ch9.copy()
            ch[0] = alpha_base
            alpha_base = oracle.quotient_widget(rec["widget"], polys, log2_large, ch, quot)
            assert np.array_equal(alpha_base, unhex(rec["alpha_base_out"], 4)[0]), rec["widget"]
            qc = oracle.canon(0, quot)
            assert np.array_equal(qc[:2], unhex(rec["quotient_first2"], 4)), rec["widget"]
            assert sha(qc) == rec["quotient_sha256"], rec["widget"]


def test_quotient_widgets_and_grand_product_oracle_vs_reference_live(oracle):
    """Where the reference build is present: the oracle's widget restatement against the reference's widget OBJECTS on fresh
    seeds (TurboPLONK five + StandardPLONK pair + MiMCComposer's three), and oracle_permutation_z against the z of a real proof's round 3."""
    import ctypes
    from oracle.oracle import RefProver, RefWidgets, prover_available
    import __graft_entry__ as ge
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent (built from /root/reference by `make -C oracle prover`)")
    pkg = ge.load_package()
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    n = 1 << 8
    pts = oracle.srs_powers(x, n + 1)
    P = RefProver(n - 24, 21, pts, x)
    for standard, order in ((None, ((0, 0), (1, 1), (2, 2), (3, 3), (4, 4))), ((n - 24, pts, x), ((0, 5), (1, 6)))):
        W = RefWidgets(P, standard=standard)
        m = W.m
        polys = [pkg.synthetic_scalars(31337 + 7 * k, m) for k in range(len(RefWidgets.LABELS))]
        for label, a in zip(RefWidgets.LABELS, polys):
            if W.has_poly(label):
                W.set_poly(label, a)
        ch = W.challenges()
        ch9 = np.stack([ch[0], ch[0], ch[1], ch[2], ch[3], ch[7], ch[4], ch[5], ch[6]])
        quot = np.zeros((m, 4), dtype=np.uint64)
        alpha_ref = alpha_or = ch[0]
        for ref_widget, lib_widget in order:
            alpha_ref = W.run(ref_widget, alpha_ref)
            c = ch9.copy()
            c[0] = alpha_or
            alpha_or = oracle.quotient_widget(lib_widget, polys, m.bit_length() - 1, c, quot)
            assert np.array_equal(alpha_or, oracle.canon(0, alpha_ref.reshape(1, 4))[0]), lib_widget
            assert np.array_equal(oracle.canon(0, quot), oracle.canon(0, W.get_poly("quotient_large", m))), lib_widget
        W.free()
    # MiMCComposer's prover (mimc_composer.cpp:277-302): permutation over three wires, the MiMC widget, the arithmetic widget
    M = RefProver(n - 24, 22, pts, x, flavour=2)
    W = RefWidgets(M, flavour=2)
    m = W.m
    polys = [pkg.synthetic_scalars(41337 + 7 * k, m) for k in range(len(RefWidgets.MIMC_LABELS))]
    for label, a in zip(RefWidgets.MIMC_LABELS, polys):
        if W.has_poly(label):
            W.set_poly(label, a)
    ch = W.challenges()
    ch9 = np.stack([ch[0], ch[0], ch[1], ch[2], ch[3], ch[7], ch[4], ch[5], ch[6]])
    quot = np.zeros((m, 4), dtype=np.uint64)
    alpha_ref = alpha_or = ch[0]
    for ref_widget, lib_widget in ((0, 5), (1, 7), (2, 6)):
        alpha_ref = W.run(ref_widget, alpha_ref)
        c = ch9.copy()
        c[0] = alpha_or
        alpha_or = oracle.quotient_widget(lib_widget, polys, m.bit_length() - 1, c, quot)
        assert np.array_equal(alpha_or, oracle.canon(0, alpha_ref.reshape(1, 4))[0]), lib_widget
        assert np.array_equal(oracle.canon(0, quot), oracle.canon(0, W.get_poly("quotient_large", m))), lib_widget
    W.free()
    M.free()
    # round 3 of a real proof: rows 0 .. n-4 of the reference's z
    for k in range(3):
        P.lib.refp_execute_round(P.h, k)
        P.lib.refp_process_queue_reference(P.h)
    wires = np.zeros((4, n, 4), dtype=np.uint64)
    sigmas = np.zeros((4, n, 4), dtype=np.uint64)
    ch5 = np.zeros((5, 4), dtype=np.uint64)
    zref = np.zeros((n, 4), dtype=np.uint64)
    P.lib.refp_round3_probe.restype = ctypes.c_int
    assert P.lib.refp_round3_probe(ctypes.c_void_p(P.h), ctypes.c_void_p(wires.ctypes.data), ctypes.c_void_p(sigmas.ctypes.data),
                                   ctypes.c_void_p(ch5.ctypes.data), ctypes.c_void_p(zref.ctypes.data)) == 0
    got = oracle.permutation_z(wires, sigmas, ch5[0], ch5[1], ch5[2:5])
    assert np.array_equal(got[: n - 3], oracle.canon(0, zref)[: n - 3])
    P.free()
