"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI (libbbg.so); results are compared
bit-exactly, on canonical values, with the oracle on the same seeded inputs, with the golden vectors recorded from the
compiled reference, and -- at BASELINE.json's full sizes -- through size-independent algebraic properties."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import limbs, sha, unhex
import callback_engines  # tests/tools: Python stand-ins for work-queue callbacks (test tooling)

pytestmark = pytest.mark.gpu

FFT, IFFT, COSET_FFT, COSET_IFFT = 0, 1, 2, 3
MSM_WINDOWS = (8, 13, 16, 17, 19, 20, 22)  # every window width libbbg.so compiles (csrc/msm_cfg.h BBG_MSM_TABLE_WIDTHS; option msm_window); 8 = the small-circuit path without a sort (msm_tiny.hip)


# ---------------------------------------------------------------------------------------------- fields
def test_native_library_loaded(pkg, bbg):
    maps = open("/proc/self/maps").read()
    assert "libbbg.so" in maps, "the HIP extension is not the code that ran"


def test_quad_ec_operations_device_check():
    """curve_quad.hip.h (four lanes per EC operation, the MSM reduce phase) against curve.hip.h's one-lane formulas ON THE DEVICE, in the
    shapes the reduce kernels use: all lanes active, quads of one wave taking different branches, runtime-length doubling chains run by
    one quad, loops (where a compiler-built switch over the lane role once selected the wrong operand: DESIGN 3.6)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_micro", "quad_check")
    if not os.path.exists(exe):
        pytest.fail("bench_micro/quad_check has not been built: run __graft_entry__.build()")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode == 0 and "failure mask 0x0 (PASS)" in out, out[-600:]


def test_constant_operand_product_device_check():
    """field29c.hip.h (the NTT's multiplier for table twiddles: x * w mod p through wq = floor(w 2^261 / p), no Montgomery digits) ON THE DEVICE
    against field.hip.h's fe_mul: 524 288 lazily reduced operands x 256 constants incl. 0, 1 and p - 1, exact limbs asserted
    (bench_micro/mul_shoup29.hip; the bounds themselves are asserted on big integers in tests/test_ntt29_model.py)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_micro", "mul_shoup29")
    if not os.path.exists(exe):
        pytest.fail("bench_micro/mul_shoup29 has not been built: run __graft_entry__.build()")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    assert r.returncode == 0 and "failure mask 0x0 (PASS)" in out, out[-600:]


@pytest.mark.parametrize("which", [0, 1])
def test_field_ops(pkg, oracle, bbg, which):
    a = pkg.synthetic_scalars(11, 20000)
    b = pkg.synthetic_scalars(12, 20000)
    a[0] = 0xFFFFFFFFFFFFFFFF  # un-reduced representatives
    b[1] = 0
    a[2] = oracle.canon(which, a[3:4])[0]
    assert np.array_equal(bbg.field_op(which, 0, a, b), oracle.fe_mul(which, a, b))
    assert np.array_equal(bbg.field_op(which, 3, a, b), oracle.fe_mul(which, a, b))  # CIOS cross-check path
    assert np.array_equal(bbg.field_op(which, 1, a, b), oracle.fe_add(which, a, b))
    assert np.array_equal(bbg.field_op(which, 2, a, b), oracle.fe_sub(which, a, b))
    assert np.array_equal(bbg.field_op(which, 4, a), oracle.from_mont(which, a))
    assert np.array_equal(bbg.field_op(which, 5, a), oracle.to_mont(which, a))
    # fused two-product multiplier a*b - c*d (one Montgomery reduction; used by every XYZZ group law)
    sq = oracle.fe_sub(which, oracle.fe_mul(which, a, a), oracle.fe_mul(which, b, b))
    assert np.array_equal(bbg.field_op(which, 8, a, b), sq)
    ab = oracle.fe_mul(which, a, b)
    assert np.array_equal(bbg.field_op(which, 9, a, b), oracle.fe_add(which, ab, ab))


@pytest.mark.parametrize("which", [0, 1])
def test_field_inverse_gcd(pkg, oracle, bbg, which):
    """The single-lane inversion of the set-up kernels (field.hip.h fe_inverse_gcd: Kaliski's binary extended Euclid, then a product with a
    power of two), per lane (op 10) and on the scalar unit (op 11), against Python's modular inverse: random residues, 0, 1, 2, p - 1,
    powers of two (the shortest and longest runs of the shift loop), un-reduced representatives."""
    p = (0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47)[which]
    R = 1 << 256
    vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 253, (1 << 253) + 1, (1 << 200) - 1, 5, p + 5, 2 * p - 1, (1 << 256) - 1]
    rnd = pkg.synthetic_scalars(77 + which, 600)
    vals += [sum(int(rnd[i, k]) << (64 * k) for k in range(4)) for i in range(rnd.shape[0])]
    a = np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)] for v in vals], dtype=np.uint64)
    want = []
    for v in vals:
        x = v % p  # the residue a R the words stand for
        std = x * pow(R, -1, p) % p
        w = 0 if std == 0 else pow(std, -1, p) * R % p
        want.append([(w >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)])
    want = np.array(want, dtype=np.uint64)
    assert np.array_equal(bbg.field_op(which, 10, a), want)
    assert np.array_equal(bbg.field_op(which, 11, a[:200]), want[:200])


def test_field_reference_kats(bbg, oracle, kats):
    for name, which, op in (("fr_mul", 0, 0), ("fr_add", 0, 1), ("fr_sub", 0, 2), ("fq_mul", 1, 0), ("fq_mul_short", 1, 0),
                            ("fq_add", 1, 1), ("fq_sub", 1, 2), ("fr_sqr", 0, 0), ("fq_sqr", 1, 0)):
        k = kats[name]
        a = limbs(k["a"])
        b = limbs(k["b"]) if "b" in k else a
        assert np.array_equal(bbg.field_op(which, op, a, b)[0], oracle.canon(which, limbs(k["expected"]))[0]), k["cite"]


# ---------------------------------------------------------------------------------------------- NTT family
@pytest.mark.parametrize("lg", [0, 1, 2, 3, 5, 9, 11, 12, 13, 15, 16])
def test_ntt_all_variants_vs_oracle(pkg, oracle, bbg, golden, lg):
    kc = unhex(golden["ntt_constant"])[0]
    c = pkg.synthetic_scalars(0xBB254 + 100 + lg, 1 << lg)
    for op in range(8):
        k = kc if op >= 4 else None
        assert np.array_equal(oracle.canon(0, bbg.ntt(c, op, 0, k)), oracle.ntt(c, op, 0, k)), (lg, op)
    if lg >= 2:
        gs = (1 << lg) // 4  # proving_key's generator_size = n on the 4n domain (proving_key.cpp:21-22)
        for op in (2, 5, 6):
            k = kc if op >= 4 else None
            assert np.array_equal(oracle.canon(0, bbg.ntt(c, op, gs, k)), oracle.ntt(c, op, gs, k)), (lg, op, gs)


def test_ntt_golden_from_reference(pkg, oracle, bbg, golden):
    kc = unhex(golden["ntt_constant"])[0]
    for rec in golden["ntt"]:
        c = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2n"])
        out = oracle.canon(0, bbg.ntt(c, rec["op"], rec["generator_size"], kc if rec["op"] >= 4 else None))
        assert sha(out) == rec["sha256"], rec
    for rec in golden["coset_fft_split"]:
        c = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2n"])
        assert sha(oracle.canon(0, bbg.coset_fft_split(c, rec["ext"]))) == rec["sha256"], rec


@pytest.mark.parametrize("tile,maxr", [(12, 9), (12, 10), (11, 8), (10, 6), (9, 5)])
def test_ntt_pass_plans(pkg, oracle, bbg, tile, maxr):
    """Different pass decompositions (2, 3 and 4 passes, several tile widths) give the same transform."""
    bbg.set_option("ntt_tile_log", tile)
    bbg.set_option("ntt_max_logr", maxr)
    try:
        for lg in (12, 14, 17):
            c = pkg.synthetic_scalars(7000 + lg, 1 << lg)
            assert np.array_equal(oracle.canon(0, bbg.ntt(c, FFT)), oracle.ntt(c, 0)), (tile, maxr, lg)
            assert np.array_equal(oracle.canon(0, bbg.ntt(c, COSET_IFFT)), oracle.ntt(c, 3)), (tile, maxr, lg)
    finally:
        bbg.set_option("ntt_tile_log", 10)
        bbg.set_option("ntt_max_logr", 7)


def test_ntt_big_tile_plans_agree(pkg, oracle, bbg):
    """2^21 / 2^22 through the 4096-element-tile plan (two passes; default for 2^21, option value 2 for 2^22) and through the 2048-tile plan
    (three passes): the same values, forward / inverse / coset, bit for bit."""
    import torch
    for lg in (21, 22):
        n = 1 << lg
        src = torch.from_numpy(pkg.synthetic_scalars(900 + lg, n).view(np.int64).reshape(-1)).cuda()
        for op in (0, 1, 2, 3):
            outs = []
            for big in (0, 2):
                bbg.set_option("ntt_big_tile", big)
                a = src.clone()
                bbg.ntt_device(a.data_ptr(), lg, op)
                bbg.sync()
                outs.append(oracle.canon(0, a.cpu().numpy().view(np.uint64).reshape(-1, 4)))
            assert np.array_equal(outs[0], outs[1]), (lg, op)
    bbg.set_option("ntt_big_tile", 1)


@pytest.mark.parametrize("maxr8", [6, 7, 8, 9, 10, 11])
def test_ntt_pass8_plans(pkg, oracle, bbg, maxr8):
    """k_ntt_pass8 (register radix-8 steps) under every per-pass radix limit: 1, 2, 3 and 4-pass decompositions with
    full (3,3,..) and partial (..,2) / (..,1) last steps, column and row flavours."""
    bbg.set_option("ntt_kernel", 2)
    bbg.set_option("ntt_max_logr8", maxr8)
    try:
        # k_ntt_pass8 (tile resident in LDS), k_ntt_pass8s (one plane at a time; one-bit last steps by lane shuffles), and k_ntt_pass29 (the
        # same pass on lazily reduced 9 x 29-bit limbs: planes = 29 here)
        for planes in (2, 1, 29):
            bbg.set_option("ntt_lds_planes", planes if planes != 29 else 0)
            bbg.set_option("ntt_limbs29", 1 if planes == 29 else 0)
            for lg in (11, 12, 13, 14, 16, 17, 19):
                c = pkg.synthetic_scalars(8000 + lg, 1 << lg)
                assert np.array_equal(oracle.canon(0, bbg.ntt(c, FFT)), oracle.ntt(c, 0)), (maxr8, planes, lg)
                assert np.array_equal(oracle.canon(0, bbg.ntt(c, COSET_IFFT)), oracle.ntt(c, 3)), (maxr8, planes, lg)
                gs = (1 << lg) // 4  # the prover's zero-extended input (generator_size = n on the 4n domain): fused into the first pass's load
                assert np.array_equal(oracle.canon(0, bbg.ntt(c, COSET_FFT, gs)), oracle.ntt(c, 2, gs)), (maxr8, planes, lg)
    finally:
        bbg.set_option("ntt_max_logr8", 10)
        bbg.set_option("ntt_lds_planes", 0)
        bbg.set_option("ntt_limbs29", -1)


def test_ntt_kernel_v1_still_matches(pkg, oracle, bbg):
    bbg.set_option("ntt_kernel", 1)
    try:
        for lg in (11, 14, 18):
            c = pkg.synthetic_scalars(8100 + lg, 1 << lg)
            assert np.array_equal(oracle.canon(0, bbg.ntt(c, FFT)), oracle.ntt(c, 0)), lg
    finally:
        bbg.set_option("ntt_kernel", 2)


def test_fft_matches_horner(pkg, oracle, bbg):
    """fft_with_small_degree (polynomial_arithmetic.test.cpp:45-68) on the GPU transform."""
    n = 16
    c = pkg.synthetic_scalars(31337, n)
    out = oracle.canon(0, bbg.ntt(c, FFT))
    w = oracle.root_of_unity(4)
    z = oracle.to_mont(0, np.array([1, 0, 0, 0], dtype=np.uint64))[0]
    for i in range(n):
        assert np.array_equal(oracle.poly_eval(c, z), out[i])
        z = oracle.fe_mul(0, z, w)[0]


def _ntt_large_golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "ntt_large.json")) as f:
        return json.load(f)["ntt"]


@pytest.mark.parametrize("lg", [18, 19, 20, 21, 22, 23, 24])
def test_ntt_full_size_properties(pkg, oracle, bbg, golden, lg):
    """BASELINE config 2 sizes.  (i) REFERENCE digests of fft / ifft / coset_fft / coset_ifft (polynomial_arithmetic.cpp:374-410;
    tests/golden/ntt_large.json for 2^19, 2^21..2^24, golden.json for 2^18 / 2^20), for every tile plan the size can take (2^21 / 2^22:
    4096-element tiles in two passes and 2048-element tiles in three); (ii) round trips (basic_fft / fft_coset_ifft_consistency,
    polynomial_arithmetic.test.cpp:70-134); (iii) Horner spot values at omega^i; (iv) linearity.  All bit-exact on canonical values."""
    import torch
    n = 1 << lg
    a = pkg.synthetic_scalars(900 + lg, n)
    ta = torch.from_numpy(a.view(np.int64)).cuda()

    def run(op, src=ta):
        work = src.clone()
        bbg.ntt_device(work.data_ptr(), lg, op)
        bbg.sync()
        return oracle.canon(0, work.cpu().numpy().view(np.uint64))
    recs = [r for r in _ntt_large_golden() if r["log2n"] == lg]
    if recs:
        assert sorted(r["op"] for r in recs) == [0, 1, 2, 3]
        for big in ((0, 2) if lg in (21, 22) else (1,)):
            for planes in (2, 1, 29):  # every pass kernel (ntt_lds_planes: automatic = 1 from 2^22; 29 = ntt_limbs29, the 29-bit-limb kernel)
                bbg.set_option("ntt_big_tile", big)
                bbg.set_option("ntt_lds_planes", planes if planes != 29 else 0)
                bbg.set_option("ntt_limbs29", 1 if planes == 29 else 0)
                try:
                    for rec in recs:
                        out = run(rec["op"])
                        for i, want in rec["spots"].items():
                            assert np.array_equal(out[int(i)], unhex(want)[0]), (lg, rec["op"], big, planes, "spot", i)
                        assert sha(out) == rec["sha256"], (lg, rec["op"], big, planes)
                finally:
                    bbg.set_option("ntt_big_tile", 1)
                    bbg.set_option("ntt_lds_planes", 0)
                    bbg.set_option("ntt_limbs29", -1)
    else:  # 2^18 / 2^20: the reference digests live in golden.json, recorded over its own seeds
        grecs = [r for r in golden["ntt"] if r["log2n"] == lg and r["op"] < 4 and r["generator_size"] == 0]
        assert sorted(r["op"] for r in grecs) == [0, 1, 2, 3]
        for planes in (2, 1, 29):
            bbg.set_option("ntt_lds_planes", planes if planes != 29 else 0)
            bbg.set_option("ntt_limbs29", 1 if planes == 29 else 0)
            try:
                for rec in grecs:
                    c = torch.from_numpy(pkg.synthetic_scalars(rec["seed"], n).view(np.int64)).cuda()
                    assert sha(run(rec["op"], c)) == rec["sha256"], (lg, rec["op"], planes)
            finally:
                bbg.set_option("ntt_lds_planes", 0)
                bbg.set_option("ntt_limbs29", -1)
    work = ta.clone()
    bbg.ntt_device(work.data_ptr(), lg, FFT)
    bbg.ntt_device(work.data_ptr(), lg, IFFT)
    bbg.sync()
    back = work.cpu().numpy().view(np.uint64)
    assert np.array_equal(oracle.canon(0, back), oracle.canon(0, a)), "ifft(fft(a)) != a"
    work = ta.clone()
    bbg.ntt_device(work.data_ptr(), lg, COSET_FFT)
    bbg.ntt_device(work.data_ptr(), lg, COSET_IFFT)
    bbg.sync()
    assert np.array_equal(oracle.canon(0, work.cpu().numpy().view(np.uint64)), oracle.canon(0, a)), "coset round trip"
    # spot values against Horner evaluation at omega^i (ordering + root choice at full size)
    out = run(FFT)
    w = oracle.root_of_unity(lg)
    for i in (0, 1, 5, n // 2 + 3, n - 1):
        # omega^i by square-and-multiply on the oracle side
        z = oracle.to_mont(0, np.array([1, 0, 0, 0], dtype=np.uint64))[0]
        base, e = w, i
        while e:
            if e & 1:
                z = oracle.fe_mul(0, z, base)[0]
            base = oracle.fe_mul(0, base, base)[0]
            e >>= 1
        assert np.array_equal(oracle.poly_eval(a, z), out[i]), i
    # linearity: fft(a + b) = fft(a) + fft(b)
    b = pkg.synthetic_scalars(1900 + lg, n)
    s = oracle.fe_add(0, a, b)
    wb = torch.from_numpy(b.view(np.int64)).cuda()
    ws = torch.from_numpy(s.view(np.int64)).cuda()
    bbg.ntt_device(wb.data_ptr(), lg, FFT)
    bbg.ntt_device(ws.data_ptr(), lg, FFT)
    bbg.sync()
    fb = wb.cpu().numpy().view(np.uint64)
    fs = ws.cpu().numpy().view(np.uint64)
    assert np.array_equal(oracle.fe_add(0, out, fb), oracle.canon(0, fs)), "linearity"


def test_ntt_golden_2_20_digest(pkg, oracle, bbg, golden):
    recs = [r for r in golden["ntt"] if r["log2n"] == 20]
    assert recs
    for rec in recs:
        c = pkg.synthetic_scalars(rec["seed"], 1 << 20)
        assert sha(oracle.canon(0, bbg.ntt(c, rec["op"], rec["generator_size"]))) == rec["sha256"], rec


def test_ntt_error_paths(pkg, bbg):
    c = pkg.synthetic_scalars(1, 8)
    with pytest.raises(pkg.BbgError):
        bbg.ntt(c, 4)  # fft_with_constant without a constant
    with pytest.raises(pkg.BbgError):
        bbg.ntt(c, 99)
    with pytest.raises(pkg.BbgError):
        bbg.ntt_prepare(29)  # beyond the 2-adicity of Fr


# ---------------------------------------------------------------------------------------------- SRS
def test_srs_synth_and_register(pkg, oracle, bbg):
    n = 3000
    want_l = oracle.srs_linear(0x123456789ABCDEF, 0xFEDCBA987654321, n)
    srs = bbg.srs_synth_linear(0x123456789ABCDEF, 0xFEDCBA987654321, n)
    assert np.array_equal(srs.read(), want_l)
    srs.free()
    want_h = oracle.srs_hashed(0xBB254, n)
    srs = bbg.srs_synth_hashed(0xBB254, n)
    assert np.array_equal(srs.read(), want_h)
    assert srs.num_points == n
    srs.free()
    srs = bbg.srs_register(want_h)
    assert np.array_equal(srs.read(5, 10), want_h[5:15])
    srs.free()
    # the reference's interleaved endomorphism table (stride 128) is accepted as-is
    table = oracle.point_table(want_h[:100])
    srs = bbg.srs_register(table, stride_bytes=128)
    assert np.array_equal(srs.read(), want_h[:100])
    srs.free()


def test_srs_transcript_roundtrip(pkg, oracle, bbg, tmp_path):
    """Ignition transcript format (srs/io.cpp:11-162): write one with the documented layout, load it through the
    product reader, expect monomials[0] = G followed by the file's points."""
    import struct
    n_file = 40
    pts = oracle.srs_hashed(77, n_file)
    plain = oracle.from_mont(1, pts.reshape(-1, 4)).reshape(n_file, 8)
    path = tmp_path / "transcript00.dat"
    with open(path, "wb") as f:
        f.write(struct.pack(">7I", 0, 1, n_file, 0, n_file, 0, 0))
        f.write(plain.astype(">u8").tobytes())  # every limb big-endian, limbs least-significant first
        f.write(b"\0" * 64)
    srs = bbg.srs_load_transcript(tmp_path, 33)
    got = srs.read()
    assert np.array_equal(got[0], oracle.g1_generator())
    assert np.array_equal(got[1:], pts[:32])
    srs.free()
    with pytest.raises(pkg.BbgError, match="Is your srs large enough"):
        bbg.srs_load_transcript(tmp_path, 100)


# ---------------------------------------------------------------------------------------------- MSM
@pytest.fixture(scope="module")
def srs16(bbg):
    s = bbg.srs_synth_hashed(0xBB254, 1 << 16)
    yield s
    s.free()


def test_msm_vs_oracle_sizes(pkg, oracle, bbg, srs16):
    pts = srs16.read(0, 5000)
    sc = pkg.synthetic_scalars(0xBB254 + 3, 5000)
    for n in (0, 1, 2, 3, 17, 64, 65, 100, 1000, 4097, 5000):  # undersized_inputs :619, ragged sizes
        got = oracle.jac_to_affine(bbg.msm(srs16, sc[:n]))
        assert np.array_equal(got, oracle.pippenger(sc[:n], pts[:n])), n
    got = oracle.jac_to_affine(bbg.msm(srs16, sc[:1000], start=300))  # Pippenger::pippenger_unsafe(scalars, from, range)
    assert np.array_equal(got, oracle.pippenger(sc[:1000], srs16.read(300, 1000)))


def test_msm_golden_from_reference(pkg, oracle, bbg, golden, srs16):
    """Results recorded from the compiled reference (pippenger_unsafe == pippenger there), incl. n = 2^16 and 2^16+1."""
    for rec in golden["msm"]:
        if rec["srs"] != "hashed" or rec["from"] + rec["n"] > (1 << 16):
            continue
        if rec.get("scalar_kind") == "mixed":
            sc = pkg.inputs.mixed_scalars(rec["scalar_seed"], rec["n"], lambda p: oracle.to_mont(0, p))
        else:
            sc = pkg.synthetic_scalars(rec["scalar_seed"], rec["n"])
        got = oracle.jac_to_affine(bbg.msm(srs16, sc, start=rec["from"]))
        assert np.array_equal(got, unhex(rec["result"], 8)[0]), rec
    assert sha(srs16.read()) == golden["msm_points_sha256"]["hashed_2^16"]


def test_msm_edge_cases(pkg, oracle, bbg, golden, srs16):
    pts = srs16.read(0, 2048)
    # pippenger_mul_by_zero (:910-927), pippenger_zero_points (:895-908)
    zero = np.zeros((100, 4), dtype=np.uint64)
    assert int(bbg.msm(srs16, zero)[3]) >> 63 == 1
    assert int(bbg.msm(srs16, zero[:0])[3]) >> 63 == 1
    # pippenger_one (:862-893)
    one = oracle.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs16, one)), pts[0])
    # pippenger_short_inputs (:723-774): full / zero / 64-bit / 3-bit scalars
    mixed = pkg.inputs.mixed_scalars(99, 2048, lambda p: oracle.to_mont(0, p))
    assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs16, mixed)), oracle.pippenger(mixed, pts))
    # un-reduced scalar representatives (kate_commitment_scheme.cpp:46-53 hands those in)
    sc = pkg.synthetic_scalars(5, 64)
    r = np.array([0x43E1F593F0000001, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029], dtype=np.uint64)
    unred = sc.copy()
    carry = np.zeros(64, dtype=np.uint64)
    for j in range(4):  # unred = sc + r (still < 2^256 since sc < 2^252)
        t = sc[:, j].astype(object) + int(r[j]) + carry.astype(object)
        unred[:, j] = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in t], dtype=np.uint64)
        carry = np.array([int(x) >> 64 for x in t], dtype=np.uint64)
    assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs16, unred)), oracle.pippenger(sc, pts[:64]))
    # all scalars equal (every term lands in the same buckets)
    same = np.tile(sc[:1], (2048, 1))
    assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs16, same)), oracle.pippenger(same, pts))
    # pippenger_edge_case_dbl (:688-721): all POINTS equal -> every bucket add is a doubling / collision
    rec = [r_ for r_ in golden["msm"] if r_["srs"] == "all_equal_to_hashed_point_0"][0]
    eq = bbg.srs_register(np.tile(pts[:1], (rec["n"], 1)))
    got = oracle.jac_to_affine(bbg.msm(eq, pkg.synthetic_scalars(rec["scalar_seed"], rec["n"])))
    assert np.array_equal(got, unhex(rec["result"], 8)[0])
    eq.free()
    # P and -P with the same scalar cancel to infinity
    neg = pts[:2].copy()
    neg[1, :4] = pts[0, :4]
    neg[1, 4:] = oracle.fe_sub(1, np.zeros((1, 4), dtype=np.uint64), pts[0:1, 4:])[0]
    s2 = bbg.srs_register(neg)
    assert int(bbg.msm(s2, np.tile(sc[:1], (2, 1)))[3]) >> 63 == 1
    s2.free()
    # linear SRS (bases with linear relations): defined for the safe variant only in the reference
    rec = [r_ for r_ in golden["msm"] if r_["srs"] == "linear"][0]
    lin = bbg.srs_synth_linear(rec["a"], rec["s"], rec["n"])
    got = oracle.jac_to_affine(bbg.msm(lin, pkg.synthetic_scalars(rec["scalar_seed"], rec["n"])))
    assert np.array_equal(got, unhex(rec["result"], 8)[0])
    lin.free()
    with pytest.raises(pkg.BbgError):
        bbg.msm(srs16, sc, start=(1 << 16) - 10)  # range exceeds the SRS


def _library_sort_available(pkg, bbg):
    """msm_sort = 0 (k_recode + rocPRIM radix sort + k_offsets) exists only in A/B builds (make ROCPRIM_SORT=1)."""
    try:
        bbg.set_option("msm_sort", 0)
    except pkg.BbgError as e:
        assert "ROCPRIM_SORT" in str(e)
        return False
    bbg.set_option("msm_sort", 1)
    return True


def test_msm_window_widths_and_sort_paths_agree(pkg, oracle, bbg, srs16):
    """Every compiled window width (each builds its own window tables on first use), and -- in A/B builds -- the rocPRIM radix-sort
    path next to the fused recode + partition sort, feed the same accumulation; results must be identical on uniform, sparse and
    heavily skewed digit distributions."""
    n = 1 << 16
    one = oracle.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    cases = {
        "uniform": pkg.synthetic_scalars(77, n),
        "mixed": pkg.inputs.mixed_scalars(78, n, lambda p: oracle.to_mont(0, p)),
        "all_one": np.tile(one, (n, 1)),                       # one bucket of one window holds everything
        "all_equal": np.tile(pkg.synthetic_scalars(79, 1), (n, 1)),  # one bucket per window holds everything
        "ragged": pkg.synthetic_scalars(80, 40001),
        "tiny": pkg.synthetic_scalars(81, 3),
    }
    pts = srs16.read(0, 3)
    sorts = (1, 0) if _library_sort_available(pkg, bbg) else (1,)
    try:
        for name, sc in cases.items():
            got = []
            for window in MSM_WINDOWS:
                bbg.set_option("msm_window", window)
                for sort in sorts:
                    bbg.set_option("msm_sort", sort)
                    got.append(oracle.jac_to_affine(bbg.msm(srs16, sc)))
            for g in got[1:]:
                assert np.array_equal(got[0], g), name
        assert np.array_equal(got[0], oracle.pippenger(cases["tiny"], pts))
    finally:
        bbg.set_option("msm_sort", 1)
        bbg.set_option("msm_window", 0)


def test_msm_option_matrix_is_bit_identical(pkg, oracle, bbg, srs16):
    """The A/B options of the MSM (bbg.h: msm_reduce_quad stage masks, msm_upload_pieces, msm_async_reduce) change how the work is
    issued, never the point: every combination against the oracle, at sizes on both sides of the lane-group combine."""
    srs = srs16
    pts = srs.read(0, 1 << 16)
    try:
        for n in (1, 257, 4096, 1 << 16):
            sc = pkg.synthetic_scalars(4242 + n, n)
            want = oracle.pippenger(sc, pts[:n])
            for quad in (0, 15, 5, 10):
                for pieces in (1, 4):
                    for overlap in (0, 1):
                        bbg.set_option("msm_reduce_quad", quad)
                        bbg.set_option("msm_upload_pieces", pieces)
                        bbg.set_option("msm_async_reduce", overlap)
                        got = oracle.jac_to_affine(bbg.msm(srs, sc))
                        assert np.array_equal(got, want), (n, quad, pieces, overlap)
    finally:
        bbg.set_option("msm_reduce_quad", 15)
        bbg.set_option("msm_upload_pieces", 1)
        bbg.set_option("msm_async_reduce", 0)


def test_msm_limbs29_degenerate_runs(pkg, oracle, bbg, srs16):
    """The 29-bit-limb accumulation (option msm_limbs29, default) does not test P = +-acc per addition: a run that meets one ends with
    ZZ = 0 mod p and its bucket is recomputed by k_redo.  Inputs made of such runs -- all points equal (every addition a doubling), every
    point twice with the same scalar, P and -P with the same scalar, points at infinity inside runs -- against the oracle and against the
    32-bit-limb kernel, one-lane accumulation forced (small MSMs otherwise take the quad kernel), at every window width."""
    pts = srs16.read(0, 4096)
    zero4 = np.zeros((1, 4), dtype=np.uint64)
    n = 4096
    sc = pkg.synthetic_scalars(2929, n)
    cases = {}
    cases["all_points_equal"] = (np.tile(pts[:1], (n, 1)), sc)
    twice = np.repeat(pts[:n // 2], 2, axis=0)
    cases["every_point_twice_same_scalar"] = (twice, np.repeat(sc[:n // 2], 2, axis=0))
    pm = twice.copy()
    pm[1::2, 4:] = oracle.fe_sub(1, np.tile(zero4, (n // 2, 1)), pm[1::2, 4:])  # odd rows: -P
    cases["p_and_minus_p_same_scalar"] = (pm, np.repeat(sc[:n // 2], 2, axis=0))
    few = np.tile(pkg.synthetic_scalars(2930, 3), (n // 3 + 1, 1))[:n]
    mixed_pts = np.tile(pts[:5], (n // 5 + 1, 1))[:n].copy()
    mixed_pts[7::11] = 0
    mixed_pts[7::11, 3] = np.uint64(1) << np.uint64(63)  # points at infinity (reference convention)
    cases["five_points_three_scalars_with_infinities"] = (mixed_pts, few)
    try:
        bbg.set_option("msm_accumulate_quad", 0)
        for name, (p_, s_) in cases.items():
            srs = bbg.srs_register(p_)
            want = oracle.msm_naive(s_, p_)  # complete group law, term by term
            for window in MSM_WINDOWS:
                bbg.set_option("msm_window", window)
                res = []
                for limbs29 in (1, 0):
                    bbg.set_option("msm_limbs29", limbs29)
                    for overlap in (0, 1):
                        bbg.set_option("msm_async_reduce", overlap)
                        res.append(bbg.msm(srs, s_))
                for r_ in res:
                    if int(want[3]) >> 63:
                        assert int(r_[3]) >> 63 == 1, (name, window)
                    else:
                        assert np.array_equal(oracle.jac_to_affine(r_), want), (name, window)
            srs.free()
    finally:
        bbg.set_option("msm_accumulate_quad", 1)
        bbg.set_option("msm_limbs29", 1)
        bbg.set_option("msm_async_reduce", 0)
        bbg.set_option("msm_window", 0)


def test_msm_limbs29_equals_limbs32_2_16(pkg, oracle, bbg, srs16):
    """Default path at 2^16 (one-lane accumulation): both limb formats, all-equal points and hashed points, identical Jacobian results."""
    n = 1 << 16
    sc = pkg.synthetic_scalars(2931, n)
    eq = bbg.srs_register(np.tile(srs16.read(0, 1), (n, 1)))
    try:
        for srs in (srs16, eq):
            got = []
            for limbs29 in (1, 0):
                bbg.set_option("msm_limbs29", limbs29)
                got.append(oracle.jac_to_affine(bbg.msm(srs, sc)))
            assert np.array_equal(got[0], got[1])
    finally:
        bbg.set_option("msm_limbs29", 1)
        eq.free()


def _affine_or_inf(oracle, jac):
    return None if int(jac[3]) >> 63 else oracle.jac_to_affine(jac)


def test_msm_batch_vs_oracle(pkg, oracle, bbg, srs16):
    """bbg_msm_batch: the independent commitments of a prover round (prover.cpp:66-74, :120-135; work_queue.hpp:208-282) through ONE sort /
    accumulate / reduce launch set, MSM k under bucket set k.  Every result against the oracle on its own inputs: batches of 1 .. 8, ragged
    lengths (StandardPLONK's n + 1 beside n; an empty MSM; a 1-term MSM), `from` offsets, zero / mixed / all-equal scalars in some sets, at
    every window width and through all three accumulation kernels (four-lane small-MSM kernel, 29-bit limbs, 32-bit limbs)."""
    pts = srs16.read(0, 6000)
    one = oracle.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))
    def scal(seed, n):
        return pkg.synthetic_scalars(seed, n)
    batches = [
        [(scal(11, 1024), 0)],
        [(scal(12, 1024), 0), (scal(13, 1025), 0)],
        [(scal(14, 1000), 0), (scal(15, 1000), 17), (scal(16, 1000), 300)],
        [(scal(17, 4096), 0), (scal(18, 4096), 0), (scal(19, 4097), 0), (scal(20, 4096), 0)],  # round 4 of StandardPLONK: n, n, n + 1 (+ one more)
        [(scal(21, 777), 5), (np.zeros((0, 4), dtype=np.uint64), 0), (scal(22, 1), 4999), (np.zeros((333, 4), dtype=np.uint64), 0),
         (pkg.inputs.mixed_scalars(23, 2000, lambda p: oracle.to_mont(0, p)), 1), (np.tile(scal(24, 1), (1500, 1)), 0), (np.tile(one, (900, 1)), 100),
         (scal(25, 5000), 1000)],
    ]
    try:
        for window in MSM_WINDOWS:
            bbg.set_option("msm_window", window)
            for quad_acc, limbs29 in ((1, 1), (0, 1), (0, 0)):
                bbg.set_option("msm_accumulate_quad", quad_acc)
                bbg.set_option("msm_limbs29", limbs29)
                for batch in batches:
                    if window != 16 and quad_acc == 1 and len(batch) == 3:
                        continue  # keep the matrix affordable: the three-way batch runs at every width through the two one-lane kernels
                    got = bbg.msm_batch(srs16, [b[0] for b in batch], [b[1] for b in batch])
                    for k, (sc, start) in enumerate(batch):
                        want = oracle.pippenger(sc, pts[start:start + sc.shape[0]]) if sc.shape[0] else None
                        g = _affine_or_inf(oracle, got[k])
                        if want is None or int(want[3]) >> 63:
                            assert g is None, (window, quad_acc, limbs29, len(batch), k)
                        else:
                            assert g is not None and np.array_equal(g, want), (window, quad_acc, limbs29, len(batch), k)
    finally:
        bbg.set_option("msm_window", 0)
        bbg.set_option("msm_accumulate_quad", 1)
        bbg.set_option("msm_limbs29", 1)


def test_msm_batch_golden_and_single_msm_agree_2_16(pkg, oracle, bbg, golden, srs16):
    """At 2^16 (the one-lane 29-bit-limb accumulation, 1024-thread second sort level): a batch of four whose first member is the reference's
    golden 2^16 MSM, with a ragged member (2^16 - 1 terms) and a mixed-scalar member.  Every member equals the same MSM issued alone; the
    golden member equals the compiled reference's result.  With the reduce phase on the auxiliary stream and shapes changing between
    calls (batch of 4 -> single -> batch of 2 -> batch of 4)."""
    n = 1 << 16
    rec = next(r for r in golden["msm"] if r["srs"] == "hashed" and r["n"] == n and r["from"] == 0 and r.get("scalar_kind") != "mixed")
    members = [pkg.synthetic_scalars(rec["scalar_seed"], n), pkg.synthetic_scalars(501, n), pkg.synthetic_scalars(502, n - 1),
               pkg.inputs.mixed_scalars(503, n, lambda p: oracle.to_mont(0, p))]
    alone = [oracle.jac_to_affine(bbg.msm(srs16, m)) for m in members]
    assert np.array_equal(alone[0], unhex(rec["result"], 8)[0])
    try:
        bbg.set_option("msm_async_reduce", 1)
        a = bbg.msm_batch(srs16, members)
        b = bbg.msm(srs16, members[1])
        c = bbg.msm_batch(srs16, members[2:])
        d = bbg.msm_batch(srs16, list(reversed(members)))
        for k in range(4):
            assert np.array_equal(oracle.jac_to_affine(a[k]), alone[k]), k
            assert np.array_equal(oracle.jac_to_affine(d[3 - k]), alone[k]), k
        assert np.array_equal(oracle.jac_to_affine(b), alone[1])
        assert np.array_equal(oracle.jac_to_affine(c[0]), alone[2]) and np.array_equal(oracle.jac_to_affine(c[1]), alone[3])
    finally:
        bbg.set_option("msm_async_reduce", 0)


def test_msm_batch_degenerate_runs_per_set(pkg, oracle, bbg, srs16):
    """The 29-bit-limb accumulation's redo queue (runs that met P = +-acc) with several bucket sets: one set made of all-equal points, one of
    P / -P pairs, one ordinary -- the queued buckets carry GLOBAL numbers (set x 2^(C-1) + bucket) and must be recomputed in the right set."""
    n = 2048
    base = srs16.read(0, n)
    zero4 = np.zeros((1, 4), dtype=np.uint64)
    pts = base.copy()
    pts[:512] = base[0]                                   # points 0 .. 511 all equal
    pm = np.repeat(base[600:856], 2, axis=0)              # points 512 .. 1023: P, -P, P', -P', ...
    pm[1::2, 4:] = oracle.fe_sub(1, np.tile(zero4, (256, 1)), pm[1::2, 4:])
    pts[512:1024] = pm
    srs = bbg.srs_register(pts)
    sc_eq = pkg.synthetic_scalars(611, 512)
    sc_pm = np.repeat(pkg.synthetic_scalars(612, 256), 2, axis=0)
    sc_ok = pkg.synthetic_scalars(613, 1024)
    batch = [(sc_ok, 1024), (sc_eq, 0), (sc_pm, 512), (sc_eq[:300], 100)]
    want = [oracle.msm_naive(sc, pts[st:st + sc.shape[0]]) for sc, st in batch]
    try:
        bbg.set_option("msm_accumulate_quad", 0)
        for window in MSM_WINDOWS:
            bbg.set_option("msm_window", window)
            for overlap in (0, 1):
                bbg.set_option("msm_async_reduce", overlap)
                got = bbg.msm_batch(srs, [b[0] for b in batch], [b[1] for b in batch])
                for k in range(len(batch)):
                    g = _affine_or_inf(oracle, got[k])
                    if int(want[k][3]) >> 63:
                        assert g is None, (window, overlap, k)
                    else:
                        assert g is not None and np.array_equal(g, want[k]), (window, overlap, k)
    finally:
        bbg.set_option("msm_accumulate_quad", 1)
        bbg.set_option("msm_async_reduce", 0)
        bbg.set_option("msm_window", 0)
        srs.free()


def test_msm_batch_error_paths_and_plan(pkg, bbg, srs16):
    sc = pkg.synthetic_scalars(1, 16)
    with pytest.raises(pkg.BbgError):
        bbg.msm_batch(srs16, [sc] * 9)                      # more than BBG_MSM_BATCH_MAX
    with pytest.raises(pkg.BbgError):
        bbg.msm_batch(srs16, [sc, sc], [0, (1 << 16) - 3])  # second range leaves the SRS
    # bbg_msm_plan: the automatic rule (msm.hip msm_auto_window), the forced width, the resident-table rule for short MSMs over long SRSs
    assert bbg.msm_plan(1 << 12) == (8, 32) and bbg.msm_plan(1 << 13) == (8, 32) and bbg.msm_plan(1 << 14) == (13, 20) and bbg.msm_plan(1 << 18) == (16, 16) and bbg.msm_plan(1 << 20) == (19, 14) and bbg.msm_plan(1 << 21) == (20, 13) and bbg.msm_plan(1 << 24) == (22, 12)
    bbg.set_option("msm_window", 17)
    assert bbg.msm_plan(1 << 20) == (17, 15)
    bbg.set_option("msm_window", 0)
    assert bbg.msm_plan(1 << 10, srs16)[0] in MSM_WINDOWS


@pytest.mark.parametrize("window", [17, 19, 20, 22])
def test_msm_wide_windows_vs_oracle(pkg, oracle, bbg, golden, srs16, window):
    """Every wider window configuration (chosen automatically only for large n) forced at small sizes: oracle parity, `from` offsets,
    the reference's golden results and the mixed-width scalar distribution."""
    bbg.set_option("msm_window", window)
    try:
        pts = srs16.read(0, 5000)
        sc = pkg.synthetic_scalars(0xBB254 + 3, 5000)
        for n in (1, 2, 17, 65, 1000, 5000):
            assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs16, sc[:n])), oracle.pippenger(sc[:n], pts[:n])), n
        got = oracle.jac_to_affine(bbg.msm(srs16, sc[:1000], start=300))
        assert np.array_equal(got, oracle.pippenger(sc[:1000], srs16.read(300, 1000)))
        for rec in golden["msm"]:
            if rec["srs"] != "hashed" or rec["from"] + rec["n"] > (1 << 16):
                continue
            if rec.get("scalar_kind") == "mixed":
                s_ = pkg.inputs.mixed_scalars(rec["scalar_seed"], rec["n"], lambda p: oracle.to_mont(0, p))
            else:
                s_ = pkg.synthetic_scalars(rec["scalar_seed"], rec["n"])
            got = oracle.jac_to_affine(bbg.msm(srs16, s_, start=rec["from"]))
            assert np.array_equal(got, unhex(rec["result"], 8)[0]), rec
    finally:
        bbg.set_option("msm_window", 0)


def test_msm_async_reduce_with_changing_shapes(pkg, oracle, bbg, srs16):
    """msm_async_reduce = 1 queues each MSM's bucket reduction on an auxiliary stream with double-buffered slots.  Back-to-back
    MSMs of DIFFERENT sizes / window widths re-lay-out the scratch arena, so the library must join the pending reductions
    first: every result of an interleaved sequence must equal the synchronous one."""
    import torch
    sizes = [1 << 16, 1000, 1 << 15, 17, 40001, 1 << 16, 3]
    windows = [16, 20, 17, 16, 22, 19, 16]
    scal = [pkg.synthetic_scalars(900 + i, n) for i, n in enumerate(sizes)]
    want = []
    for sc, w in zip(scal, windows):
        bbg.set_option("msm_window", w)
        want.append(oracle.jac_to_affine(bbg.msm(srs16, sc)))
    dsc = [torch.from_numpy(sc.view(np.int64).reshape(-1)).cuda() for sc in scal]
    outs = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in sizes]
    bbg.set_option("msm_async_reduce", 1)
    try:
        for rep in range(3):
            for d, o, n, w in zip(dsc, outs, sizes, windows):
                bbg.set_option("msm_window", w)
                bbg.msm_device(srs16, d.data_ptr(), n, o.data_ptr())
            bbg.join()
            bbg.sync()
            for o, wnt in zip(outs, want):
                assert np.array_equal(oracle.jac_to_affine(o.cpu().numpy().view(np.uint64)), wnt), rep
    finally:
        bbg.set_option("msm_async_reduce", 0)
        bbg.set_option("msm_window", 0)


def test_g1_sum_and_normalize(pkg, oracle, bbg, srs16):
    sc = pkg.synthetic_scalars(21, 300)
    parts = np.stack([bbg.msm(srs16, sc[i * 100:(i + 1) * 100], start=i * 100) for i in range(3)])
    whole = oracle.jac_to_affine(bbg.msm(srs16, sc))
    assert np.array_equal(oracle.jac_to_affine(bbg.g1_sum(parts)), whole)  # sharded by point range == whole (c_bind.cpp:31-46)
    assert np.array_equal(oracle.g1_sum(parts), whole)
    norm = bbg.g1_normalize(parts)
    for i in range(3):
        assert np.array_equal(norm[i], oracle.jac_to_affine(parts[i]))


def test_msm_2_20_golden_and_properties(pkg, oracle, bbg, golden):
    """BASELINE config 3: n = 2^20.  Bit-exact against the result recorded from the compiled reference
    (pippenger_unsafe on the same SRS and scalars), plus MSM(-s) = -MSM(s) (oversized_inputs :573-617) and linearity."""
    n = 1 << 20
    srs = bbg.srs_synth_hashed(0xBB254, n)
    assert sha(srs.read()) == golden["msm_points_sha256"]["hashed_2^20"]
    rec = [r for r in golden["msm"] if r["n"] == n][0]
    sc = pkg.synthetic_scalars(rec["scalar_seed"], n)
    res = oracle.jac_to_affine(bbg.msm(srs, sc))
    assert np.array_equal(res, unhex(rec["result"], 8)[0])
    neg = oracle.fe_sub(0, np.zeros_like(sc), sc)
    res_neg = oracle.jac_to_affine(bbg.msm(srs, neg))
    assert np.array_equal(res_neg[:4], res[:4])
    assert np.array_equal(res_neg[4:], oracle.fe_sub(1, np.zeros((1, 4), dtype=np.uint64), res[4:])[0])
    sc2 = pkg.synthetic_scalars(4321, n)
    r2 = oracle.jac_to_affine(bbg.msm(srs, sc2))
    rsum = oracle.jac_to_affine(bbg.msm(srs, oracle.fe_add(0, sc, sc2)))
    assert np.array_equal(oracle.g1_add(res, r2), rsum)
    # point-range sharding (the multi-GPU decomposition) == whole
    parts = np.stack([bbg.msm(srs, sc[i * (n // 4):(i + 1) * (n // 4)], start=i * (n // 4)) for i in range(4)])
    assert np.array_equal(oracle.jac_to_affine(bbg.g1_sum(parts)), res)
    srs.free()


def test_msm_2_22_properties_wide_windows(pkg, oracle, bbg, golden):
    """n = 2^22 takes the 20-bit-window configuration automatically (13 windows, 2^19 buckets).  No CPU oracle finishes at
    this size in seconds, so: (a) the first 2^20 points and scalars reproduce the reference's recorded 2^20 result through
    BOTH widths, (b) whole == sum of four point-range shards (each shard is a 2^20 MSM = 16-bit windows: the two
    configurations must agree on the same data), (c) linearity, (d) MSM(-s) = -MSM(s)."""
    n = 1 << 22
    srs = bbg.srs_synth_hashed(0xBB254, n)
    rec = [r for r in golden["msm"] if r["n"] == (1 << 20)][0]
    sc20 = pkg.synthetic_scalars(rec["scalar_seed"], 1 << 20)
    want = unhex(rec["result"], 8)[0]
    try:
        for window in MSM_WINDOWS:
            bbg.set_option("msm_window", window)
            assert np.array_equal(oracle.jac_to_affine(bbg.msm(srs, sc20)), want), window
    finally:
        bbg.set_option("msm_window", 0)
    sc = pkg.synthetic_scalars(777, n)
    whole = oracle.jac_to_affine(bbg.msm(srs, sc))
    q = n // 4
    parts = np.stack([bbg.msm(srs, sc[i * q:(i + 1) * q], start=i * q) for i in range(4)])
    assert np.array_equal(oracle.jac_to_affine(bbg.g1_sum(parts)), whole)
    sc2 = pkg.synthetic_scalars(778, n)
    r2 = oracle.jac_to_affine(bbg.msm(srs, sc2))
    rsum = oracle.jac_to_affine(bbg.msm(srs, oracle.fe_add(0, sc, sc2)))
    assert np.array_equal(oracle.g1_add(whole, r2), rsum)
    neg = oracle.jac_to_affine(bbg.msm(srs, oracle.fe_sub(0, np.zeros_like(sc), sc)))
    assert np.array_equal(neg[:4], whole[:4])
    assert np.array_equal(neg[4:], oracle.fe_sub(1, np.zeros((1, 4), dtype=np.uint64), whole[4:])[0])
    srs.free()


def test_msm_srs_above_2_26_points(pkg, oracle, bbg):
    """The sorted value carries a 27-bit point index (msm_cfg.h; round 3: 26): an SRS of 2^26 + 2^16 points on ONE device -- more than one
    Ignition transcript file set of 2^26 would need, the reference's schedule word has 32 index bits (scalar_multiplication.hpp:24-29).
      (a) the synthetic points above index 2^26 are the oracle's (index-based generator);
      (b) an MSM over the 2^16 points that straddle index 2^26 (`from` = 2^26 - 2^15) equals the oracle's Pippenger over the same points;
      (c) the first 2^24 points with the scalars of tests/golden/msm24.json reproduce the compiled REFERENCE's 2^24 result through this
          SRS's 12-window tables (the msm24.json method: same seeds, the reference's sixteen shards summed);
      (d) one MSM over ALL 2^26 + 2^16 points (random scalars) equals the group sum of its two parts [0, 2^26) and [2^26, end), and a
          scalar vector that is zero below 2^26 gives the part above (no entry may lose index bit 26)."""
    import json
    import torch
    top = 1 << 26
    n = top + (1 << 16)
    free = bbg.memory_report()["device_free"]
    if free < (120 << 30):
        pytest.skip("needs ~100 GB of free HBM (12 windows x 64 B x 2^26 points + the sort arena)")
    srs = bbg.srs_synth_hashed(0xBB254, n)
    try:
        assert bbg.msm_plan(n, srs) == (22, 12)
        pts_hi = srs.read(top - (1 << 15), 1 << 16)
        want_pts = oracle.srs_hashed(0xBB254 + top - 8, 16)  # (a) sixteen points around the old cap, oracle-generated
        assert np.array_equal(pts_hi[(1 << 15) - 8:(1 << 15) + 8], want_pts)
        sc = pkg.synthetic_scalars(2626, 1 << 16)
        got = oracle.jac_to_affine(bbg.msm(srs, sc, start=top - (1 << 15)))  # (b)
        assert np.array_equal(got, oracle.pippenger(sc, pts_hi))
        g24 = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm24.json")))  # (c)
        sc24 = pkg.synthetic_scalars(0xBB254 + 24, 1 << g24["log2n"])
        got24 = bbg.g1_normalize(bbg.msm(srs, sc24).reshape(1, 12)).reshape(-1)
        assert np.array_equal(got24, np.frombuffer(bytes.fromhex(g24["result"]), dtype=np.uint64)), "2^24 golden through the 2^26+ SRS"
        del sc24
        dev = torch.device("cuda", 0)  # (d) device-resident: 2.1 GB of scalars
        d_sc = torch.empty((n, 4), dtype=torch.int64, device=dev)
        chunk = 1 << 22
        for lo in range(0, n, chunk):
            cnt = min(chunk, n - lo)
            d_sc[lo:lo + cnt] = torch.from_numpy(pkg.synthetic_scalars(2627, cnt, lo).view(np.int64)).to(dev)
        d_out = torch.zeros(3 * 12, dtype=torch.int64, device=dev)
        bbg.msm_device(srs, d_sc.data_ptr(), n, d_out.data_ptr())
        bbg.msm_device(srs, d_sc.data_ptr(), top, d_out.data_ptr() + 96)
        bbg.msm_device(srs, d_sc.data_ptr() + top * 32, n - top, d_out.data_ptr() + 192, start=top)
        bbg.join()
        bbg.sync()
        res = d_out.cpu().numpy().view(np.uint64).reshape(3, 12)
        whole = bbg.g1_normalize(res[0:1]).reshape(-1)
        parts = bbg.g1_normalize(bbg.g1_sum(res[1:3]).reshape(1, 12)).reshape(-1)
        assert np.array_equal(whole, parts), "MSM over 2^26 + 2^16 points != sum of its parts"
        d_sc[:top] = 0
        bbg.msm_device(srs, d_sc.data_ptr(), n, d_out.data_ptr())
        bbg.sync()
        only_top = bbg.g1_normalize(d_out.cpu().numpy().view(np.uint64).reshape(3, 12)[0:1]).reshape(-1)
        assert np.array_equal(only_top, bbg.g1_normalize(res[2:3]).reshape(-1))
        del d_sc
    finally:
        srs.free()
        bbg.memory_trim(tables=True)


def test_memory_report_and_trim(pkg, oracle):
    """bbg_memory_report / bbg_memory_trim (the HBM budget of a context: window tables, NTT tables, MSM arena, scratch, resident keys) on a
    context of its own: every class appears when its first user runs, the total is the sum of the parts, trimming releases the rebuildable
    part and the same calls give the same results afterwards."""
    import ctypes
    ctx = pkg.Bbg(0)
    try:
        r0 = ctx.memory_report()
        assert r0["total"] == 0 and r0["live_srs"] == 0 and r0["device_total"] > (200 << 30) and 0 < r0["device_free"] <= r0["device_total"]
        n = 1 << 14
        srs = ctx.srs_synth_hashed(0xBB254, n)
        r1 = ctx.memory_report()
        home_c, home_w = ctx.msm_plan(n)  # the width a full-size MSM over this SRS uses: its table is built at registration
        assert r1["live_srs"] == 1 and r1["srs_tables"] == n * home_w * 64 and r1["total"] == r1["srs_tables"]
        sc = pkg.synthetic_scalars(55, n)
        want = oracle.pippenger(sc, srs.read())
        assert np.array_equal(oracle.jac_to_affine(ctx.msm(srs, sc)), want)
        c = pkg.synthetic_scalars(56, n)
        f0 = ctx.ntt(c, FFT)
        ctx.set_option("msm_window", 17)
        assert np.array_equal(oracle.jac_to_affine(ctx.msm(srs, sc)), want)  # a second width: its own table
        ctx.set_option("msm_window", 0)
        r2 = ctx.memory_report()
        assert home_c != 17 and r2["srs_tables"] == n * (home_w + 15) * 64 and r2["msm_arena"] > 0 and r2["ntt_tables"] >= 4 * 32 * n and r2["ntt_domains"] == 1 and r2["scratch"] > 0
        gens = np.stack([ctx.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0] for k in (5, 5, 6, 7)])
        h = ctypes.c_void_p()
        ctx._ck(ctx.lib.bbg_prover_create(ctx.ctx, srs.handle, 12, 4, gens.ctypes.data, ctypes.byref(h)))
        pb = ctypes.c_size_t()
        ctx._ck(ctx.lib.bbg_prover_device_bytes(h, ctypes.byref(pb)))
        r3 = ctx.memory_report()
        assert r3["live_provers"] == 1 and r3["prover_keys"] == pb.value and pb.value >= (8 * 4096 + 6 * 4 * 4096) * 32
        assert r3["total"] == sum(r3[k] for k in ("srs_points", "srs_tables", "ntt_tables", "msm_arena", "scratch", "prover_keys"))
        ctx.lib.bbg_prover_destroy(h)
        assert ctx.memory_report()["live_provers"] == 0
        released = ctx.memory_trim(tables=True)
        r4 = ctx.memory_report()
        assert released == r3["total"] - pb.value - r4["total"] and r4["srs_tables"] == n * home_w * 64 and r4["ntt_tables"] == 0 and r4["msm_arena"] == 0
        # everything is rebuilt on demand: same results
        assert np.array_equal(oracle.jac_to_affine(ctx.msm(srs, sc)), want)
        assert np.array_equal(ctx.ntt(c, FFT), f0)
        srs.free()
        assert ctx.memory_report()["live_srs"] == 0 and ctx.memory_report()["srs_tables"] == 0
    finally:
        ctx.close()


def test_poly_linear_combination(pkg, oracle, bbg):
    """opening_poly[i] = t[i] + sum_k poly_k[i] * nu_k (kate_commitment_scheme.cpp:216-226) against the oracle's field ops."""
    import torch
    n, k = 5000, 25
    polys = [pkg.synthetic_scalars(600 + j, n) for j in range(k)]
    scal = pkg.synthetic_scalars(700, k)
    base = pkg.synthetic_scalars(701, n)
    want = oracle.canon(0, base)
    for j in range(k):
        want = oracle.fe_add(0, want, oracle.fe_mul(0, polys[j], np.tile(scal[j], (n, 1))))
    dev = [torch.from_numpy(p.view(np.int64).reshape(-1)).cuda() for p in polys]
    dbase = torch.from_numpy(base.view(np.int64).reshape(-1)).cuda()
    out = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    bbg.poly_linear_combination_device([d.data_ptr() for d in dev], scal, dbase.data_ptr(), out.data_ptr(), n)
    bbg.sync()
    assert np.array_equal(oracle.canon(0, out.cpu().numpy().view(np.uint64).reshape(-1, 4)), want)
    bbg.poly_linear_combination_device([d.data_ptr() for d in dev[:3]], scal[:3], None, dbase.data_ptr(), n)  # no base, output elsewhere
    bbg.sync()
    want3 = oracle.canon(0, np.zeros((n, 4), dtype=np.uint64))
    for j in range(3):
        want3 = oracle.fe_add(0, want3, oracle.fe_mul(0, polys[j], np.tile(scal[j], (n, 1))))
    assert np.array_equal(oracle.canon(0, dbase.cpu().numpy().view(np.uint64).reshape(-1, 4)), want3)
    with pytest.raises(pkg.BbgError):
        bbg.poly_linear_combination_device([dev[0].data_ptr()] * 33, pkg.synthetic_scalars(1, 33), None, out.data_ptr(), n)


@pytest.mark.parametrize("limbs29", [1, 0])
def test_poly_linear_combination_every_term_count(pkg, oracle, bbg, limbs29):
    """The 29-bit linear combination sums four terms per reduction with a tail of one to three (poly29.hip.h; option poly_limbs29, 0 = the
    32-bit kernel): every tail length, the 32-term maximum, with and without a base, over inputs that include the largest canonical value
    p - 1 and COARSE residues p + x (a device array may hold any representative below 2p) -- against the oracle's field operations.  The
    evaluations (one and several polynomials) run under the same option at ragged lengths."""
    import torch
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    n = 777
    limbs = lambda v: [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    polys = [pkg.synthetic_scalars(800 + j, n) for j in range(32)]
    for j, pl in enumerate(polys):
        pl[j] = limbs(P - 1)                                  # canonical extreme (the arrays are R-form words: any value below p is one)
        x = int.from_bytes(pl[j + 40].tobytes(), "little")
        if x + P < (1 << 256) and x < P:
            pl[j + 40] = limbs(x + P)                         # the same element, coarse representative
    scal = pkg.synthetic_scalars(801, 32)
    scal[5] = limbs(P - 1)
    base = pkg.synthetic_scalars(802, n)
    dev = [torch.from_numpy(p.view(np.int64).reshape(-1)).cuda() for p in polys]
    dbase = torch.from_numpy(base.view(np.int64).reshape(-1)).cuda()
    out = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    bbg.set_option("poly_limbs29", limbs29)
    try:
        for k in (1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 25, 32):
            for with_base in (True, False):
                want = oracle.canon(0, base) if with_base else oracle.canon(0, np.zeros((n, 4), dtype=np.uint64))
                for j in range(k):
                    want = oracle.fe_add(0, want, oracle.fe_mul(0, oracle.canon(0, polys[j]), np.tile(scal[j], (n, 1))))
                bbg.poly_linear_combination_device([d.data_ptr() for d in dev[:k]], scal[:k], dbase.data_ptr() if with_base else None, out.data_ptr(), n)
                bbg.sync()
                got = out.cpu().numpy().view(np.uint64).reshape(-1, 4)
                assert np.array_equal(oracle.canon(0, got), want), (k, with_base)
        kc = pkg.synthetic_scalars(803, 1)[0]
        for m in (1, 255, 256, 257, 4095, 4096, 4097, 12289, 70001):
            a_np = pkg.synthetic_scalars(804 + m % 89, m)
            a_np[m // 2] = limbs(P - 1)
            a = torch.from_numpy(a_np.view(np.int64).reshape(-1)).cuda()
            assert np.array_equal(bbg.poly_evaluate_device(a.data_ptr(), m, kc), oracle.poly_eval(oracle.canon(0, a_np), kc)), m
    finally:
        bbg.set_option("poly_limbs29", 1)


def test_round_kernel_error_paths(pkg, bbg):
    import torch
    buf = torch.zeros(64 * 4, dtype=torch.int64, device="cuda")
    ch9 = pkg.synthetic_scalars(1, 9)
    ptrs = [buf.data_ptr()] * 21
    with pytest.raises(pkg.BbgError):
        bbg.quotient_widget_device(8, ptrs, 6, ch9, buf.data_ptr())      # unknown widget
    with pytest.raises(pkg.BbgError):
        bbg.quotient_widget_device(7, ptrs, 6, ch9, buf.data_ptr())      # MiMC without its two selectors (extended table empty)
    with pytest.raises(pkg.BbgError):
        bbg.quotient_widget_device(0, ptrs, 2, ch9, buf.data_ptr())      # domain too small for the shifted rows
    missing = list(ptrs)
    missing[20] = 0                                                       # the permutation widget reads L_1
    with pytest.raises(pkg.BbgError):
        bbg.quotient_widget_device(0, missing, 6, ch9, buf.data_ptr())
    bbg.quotient_widget_device(3, missing, 6, ch9, buf.data_ptr())        # the range widget does not
    with pytest.raises(pkg.BbgError):
        bbg.permutation_grand_product_device([buf.data_ptr()] * 4, [buf.data_ptr()] * 3 + [0], 6, ch9[0], ch9[1], ch9[2:5],
                                             buf.data_ptr())
    bbg.sync()


# ---------------------------------------------------------------------------------------------- permutation grand product
def _gpu_grand_product(pkg, bbg, wires, sigmas, log2n, beta, gamma, ks):
    import torch
    dw = [torch.from_numpy(np.ascontiguousarray(wires[k]).view(np.int64).reshape(-1)).cuda() for k in range(4)]
    ds = [torch.from_numpy(np.ascontiguousarray(sigmas[k]).view(np.int64).reshape(-1)).cuda() for k in range(4)]
    z = torch.zeros((1 << log2n) * 4, dtype=torch.int64, device="cuda")
    bbg.permutation_grand_product_device([t.data_ptr() for t in dw], [t.data_ptr() for t in ds], log2n, beta, gamma, ks, z.data_ptr())
    bbg.sync()
    return z.cpu().numpy().view(np.uint64).reshape(-1, 4)


@pytest.mark.parametrize("log2n", [0, 1, 3, 4, 5, 9, 12])
def test_permutation_grand_product_vs_oracle(pkg, oracle, bbg, log2n):
    """z[0] = 1, z[j+1] = prod_{i<=j} N_i / D_i (permutation_widget_impl.hpp:48-268) against the oracle's serial restatement."""
    n = 1 << log2n
    wires = np.stack([pkg.synthetic_scalars(800 + k, n) for k in range(4)])
    sigmas = np.stack([pkg.synthetic_scalars(810 + k, n) for k in range(4)])
    ch = pkg.synthetic_scalars(820, 5)
    got = oracle.canon(0, _gpu_grand_product(pkg, bbg, wires, sigmas, log2n, ch[0], ch[1], ch[2:5]))
    assert np.array_equal(got, oracle.permutation_z(wires, sigmas, ch[0], ch[1], ch[2:5]))


def test_permutation_grand_product_in_a_real_proof(pkg, oracle, bbg):
    """The same kernel on the inputs of a real proof of the reference prover (round 3): rows 0 .. n-4 of the reference's z
    (ifft'ed by the prover, transformed back with its own fft) are reproduced bit for bit; the last three rows are the
    reference's random blinding.  A valid permutation also closes: the product over all rows is 1."""
    from oracle.oracle import RefProver, prover_available
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent on this machine")
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    P = RefProver(1 << 11, 14, oracle.srs_powers(x, (2 << 11) + 1), x)
    n = P.n
    for k in range(3):
        P.lib.refp_execute_round(P.h, k)
        P.lib.refp_process_queue_reference(P.h)
    wires = np.zeros((4, n, 4), dtype=np.uint64)
    sigmas = np.zeros((4, n, 4), dtype=np.uint64)
    ch = np.zeros((5, 4), dtype=np.uint64)
    zref = np.zeros((n, 4), dtype=np.uint64)
    P.lib.refp_round3_probe.restype = ctypes.c_int
    rc = P.lib.refp_round3_probe(ctypes.c_void_p(P.h), ctypes.c_void_p(wires.ctypes.data), ctypes.c_void_p(sigmas.ctypes.data),
                                 ctypes.c_void_p(ch.ctypes.data), ctypes.c_void_p(zref.ctypes.data))
    assert rc == 0
    got = oracle.canon(0, _gpu_grand_product(pkg, bbg, wires, sigmas, n.bit_length() - 1, ch[0], ch[1], ch[2:5]))
    want = oracle.canon(0, zref)
    assert np.array_equal(got[: n - 3], want[: n - 3])
    assert not np.array_equal(got[n - 3:], want[n - 3:])  # the reference blinds these rows with fresh randomness
    P.free()


def test_coset_fft_split_sharded_device_ops(pkg, oracle, bbg):
    """parallel.coset_fft_split_sharded with the real device ops on one rank (the multi-rank exchange -- a plain all-gather -- is
    covered by the gloo test): ext independent coset FFTs with generator shifts g * w_{ext n}^k, interleaved, equal the
    reference's coset_fft(coeffs, small, large, ext) as restated by the oracle."""
    import importlib
    import torch
    par = importlib.import_module("aztec_amd.parallel")
    for lg, ext in ((10, 4), (12, 8), (11, 2)):
        c = oracle.canon(0, pkg.synthetic_scalars(1200 + lg, 1 << lg))
        x = torch.from_numpy(c.copy().view(np.int64).reshape(-1)).cuda()
        out = par.coset_fft_split_sharded(par.BbgNttOps(bbg), None, x, lg, ext)
        bbg.sync()
        got = oracle.canon(0, out.cpu().numpy().view(np.uint64).reshape(-1, 4))
        assert np.array_equal(got, oracle.canon(0, oracle.coset_fft_split(c, ext))), (lg, ext)


# ---------------------------------------------------------------------------------------------- quotient widgets (8f-2)
def _widget_inputs(pkg, m):
    from oracle.oracle import RefWidgets
    seed = 0xBB254 + 9000
    return [pkg.synthetic_scalars(seed + k, m) for k in range(len(RefWidgets.LABELS))]


def _run_gpu_widgets(pkg, bbg, polys, log2_large, ch9, alpha0, widgets=(0, 1, 2, 3, 4)):
    """Uploads the 21 polynomials, runs the widgets in the prover's order; yields (alpha_out, quotient) after each."""
    import torch
    dev = [torch.from_numpy(p.view(np.int64).reshape(-1)).cuda() for p in polys]
    quot = torch.zeros((1 << log2_large) * 4, dtype=torch.int64, device="cuda")
    ptrs = [d.data_ptr() for d in dev]
    alpha_base = alpha0
    for widget in widgets:
        ch = ch9.copy()
        ch[0] = alpha_base
        alpha_base = bbg.quotient_widget_device(widget, ptrs, log2_large, ch, quot.data_ptr())
        bbg.sync()
        yield alpha_base, quot.cpu().numpy().view(np.uint64).reshape(-1, 4)


@pytest.mark.parametrize("limbs29,coarse,plan", [(1, False, 1), (1, True, 1), (0, False, 1), (1, False, 0)])
@pytest.mark.parametrize("log2_large", [3, 5, 8, 13])
def test_quotient_widgets_vs_oracle(pkg, oracle, bbg, log2_large, limbs29, coarse, plan):
    """All eight widgets against the oracle's restatement (itself pinned by the reference goldens / live reference proofs) on arbitrary
    challenge values -- public_input_delta, g, k1..k3 are NOT the transcript / field constants here -- and on the smallest legal domain.
    limbs29: the kernels on lazily reduced 29-bit limbs (quotient29.hip.h, default) or the 32-bit ones.  coarse: every input polynomial is
    handed over as x + p (the upper half of the [0, 2p) range the prover's coset FFTs fill): the largest values the compile-time bounds of
    the 29-bit kernels are priced for.  plan: the set-up block's challenge powers by the lanes of a wave side by side (option
    quotient_setup_plan, default) or by one lane's chain of products -- the alpha_base every widget hands on is checked either way."""
    m = 1 << log2_large
    polys = [pkg.synthetic_scalars(5000 + 31 * log2_large + k, m) for k in range(23)]
    ch9 = pkg.synthetic_scalars(6000 + log2_large, 9)
    quot = np.zeros((m, 4), dtype=np.uint64)
    alpha_base = ch9[0].copy()
    order = (0, 1, 2, 3, 4, 5, 7, 6)  # after the TurboPLONK five: MiMCComposer's list (5 assigns again; MiMC and arithmetic accumulate)
    gpu_polys = polys
    if coarse:
        P = int("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", 16)
        def plus_p(a):
            out = np.empty_like(a)
            for r in range(a.shape[0]):
                v = sum(int(a[r, k]) << (64 * k) for k in range(4)) + P
                assert v < 2 * P
                out[r] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
            return out
        gpu_polys = [plus_p(p) for p in polys]
    bbg.set_option("quotient_limbs29", limbs29)
    bbg.set_option("quotient_setup_plan", plan)
    try:
        for widget, (alpha_out, q) in zip(order, _run_gpu_widgets(pkg, bbg, gpu_polys, log2_large, ch9, ch9[0], order)):
            ch = ch9.copy()
            ch[0] = alpha_base
            alpha_base = oracle.quotient_widget(widget, polys, log2_large, ch, quot)
            assert np.array_equal(oracle.canon(0, alpha_out.reshape(1, 4))[0], alpha_base), widget
            assert np.array_equal(oracle.canon(0, q), oracle.canon(0, quot)), widget
            assert (q[:, 3] <= np.uint64(0x60c89ce5c2634053)).all(), widget  # every stored residue below 2p (top word of 2p = 0x60c89ce5c2634053)
    finally:
        bbg.set_option("quotient_limbs29", 1)
        bbg.set_option("quotient_setup_plan", 1)


def test_quotient_widgets_vs_reference_golden(pkg, oracle, bbg):
    """Permutation + turbo arithmetic / fixed-base / range / logic quotient contributions on the 4n coset domain against digests
    recorded from the reference's own widget objects (tests/golden/widgets.json, gen_golden_widgets.py), n = 2^6 and 2^10."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "widgets.json")) as f:
        G = json.load(f)
    for case in G["cases"] + G["standard_cases"]:  # TurboPLONK 0..4, then StandardPLONK (three wires): widgets 5, 6
        log2_large = case["log2n"] + 2
        m = 1 << log2_large
        c = case["challenges"]
        ch9 = np.stack([unhex(c[k], 4)[0] for k in ("alpha", "alpha", "beta", "gamma", "public_input_delta", "g", "k1", "k2", "k3")])
        polys = _widget_inputs(pkg, m)
        order = [rec["widget"] for rec in case["widgets"]]
        for rec, (alpha_out, q) in zip(case["widgets"], _run_gpu_widgets(pkg, bbg, polys, log2_large, ch9, unhex(c["alpha"], 4)[0], order)):
            assert np.array_equal(oracle.canon(0, alpha_out.reshape(1, 4))[0], unhex(rec["alpha_base_out"], 4)[0]), rec["widget"]
            qc = oracle.canon(0, q)
            assert np.array_equal(qc[:2], unhex(rec["quotient_first2"], 4)), rec["widget"]
            assert sha(qc) == rec["quotient_sha256"], rec["widget"]


def test_quotient_widgets_vs_reference_live(pkg, oracle, bbg):
    """Same comparison, element by element, against the reference widgets run live at n = 2^14 (4n = 2^16)."""
    from oracle.oracle import RefProver, RefWidgets, prover_available
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent on this machine")
    log2n = 14
    n = 1 << log2n
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    P = RefProver(n - 24, 5, oracle.srs_powers(x, n + 1), x)
    W = RefWidgets(P)
    m = 4 * n
    polys = _widget_inputs(pkg, m)
    for label, p in zip(RefWidgets.LABELS, polys):
        W.set_poly(label, p)
    ch = W.challenges()
    ch9 = np.stack([ch[0], ch[0], ch[1], ch[2], ch[3], ch[7], ch[4], ch[5], ch[6]])
    alpha_base = ch[0]
    for widget, (alpha_out, q) in enumerate(_run_gpu_widgets(pkg, bbg, polys, log2n + 2, ch9, ch[0])):
        want_alpha = W.run(widget, alpha_base)
        want_q = oracle.canon(0, W.get_poly("quotient_large", m))
        assert np.array_equal(oracle.canon(0, alpha_out.reshape(1, 4)), oracle.canon(0, want_alpha.reshape(1, 4))), widget
        assert np.array_equal(oracle.canon(0, q), want_q), widget
        alpha_base = want_alpha
    W.free()
    P.free()


# ---------------------------------------------------------------------------------------------- the reference PROVER seam
@pytest.mark.parametrize("log2_gates,fused_item", [(9, False), (13, False), (13, True)])
def test_reference_prover_with_gpu_engine(pkg, oracle, bbg, log2_gates, fused_item):
    """The reference's REAL TurboPLONK prover (oracle/ref_prover_driver.cpp: TurboComposer circuit, TurboProver rounds,
    prebuilt from the reference's own sources) with every MSM / coset-FFT / iFFT item of work_queue::process_queue
    (work_queue.hpp:208-282) computed by this library through its C ABI host entry points -- the seam a barretenberg build
    binds to.  Every item must equal the reference CPU result bit for bit (canonical), and the reference's TurboVerifier must
    accept the proof.  Sizes: n = 2^10 and 2^14 gates after padding (MSMs of n and n+1 terms, coset FFTs of 4n)."""
    from oracle.oracle import RefProver, prover_available
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent on this machine")
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    pts = oracle.srs_powers(x, (2 << log2_gates) + 1)
    P = RefProver(1 << log2_gates, 11, pts, x)
    n = P.n
    mon = P.monomials()
    srs = bbg.srs_register(mon)  # what a Pippenger-constructor hook registers: the prover's own monomials

    class Engine:
        def msm(self, s):
            return bbg.msm(srs, s)

        def coset_fft(self, a, generator_size):
            return bbg.ntt(a, pkg.binding.COSET_FFT, generator_size)

        def ifft(self, a):
            return bbg.ntt(a, pkg.binding.IFFT)

    class FusedEngine(Engine):  # the FFT work item as one call (bbg_coset_fft_extend): n coefficients in, 4n + 4 values out
        def fft_item(self, wire, log2_domain):
            return bbg.coset_fft_extend(wire, log2_domain)

    proof = P.prove(FusedEngine() if fused_item else Engine())
    assert P.mismatches == 0
    assert P.counts[0] >= 9 and P.counts[1] >= 4 and P.counts[2] >= 3, P.counts
    assert len(proof) > 0 and P.verify() == 1
    print(f"\nreference TurboProver, n = {n}: {P.counts[0]} MSM + {P.counts[1]} coset-FFT(4n) + {P.counts[2]} iFFT on the GPU "
          f"(checked against the CPU per item); rounds {P.t_rounds*1e3:.1f} ms")
    srs.free()
    P.free()


def test_reference_prover_linked_against_shim(pkg, oracle, bbg):
    """INTEGRATION.md 2a end to end: the reference's TurboPLONK prover, UNMODIFIED, linked with shim/bbg_barretenberg_shim.cpp
    and -Wl,--wrap so that pippenger_unsafe / fft / ifft / coset_fft / coset_ifft ... resolve to libbbg.so.  No callbacks:
    work_queue::process_queue and every inline polynomial::ifft / coset_ifft of the rounds run on the GPU.  The reference's
    TurboVerifier must accept the proof."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    pts = oracle.srs_powers(x, (2 << 13) + 1)
    P = RefProver(1 << 13, 12, pts, x, gpu_linked=True)
    proof = P.prove()  # engine=None: the reference's own process_queue -> __wrap_* -> libbbg.so
    ok = P.verify()
    assert len(proof) > 0 and ok == 1, ("shim-linked", len(proof), ok)
    P.free()


def test_reference_prover_round4_on_gpu(pkg, oracle, bbg):
    """execute_fourth_round's quotient (five widgets + divide_by_pseudo_vanishing_polynomial + coset_ifft, prover.cpp:304-343)
    computed on the device inside a REAL proof of the reference prover: the resulting quotient coefficients equal the
    reference's (canonical), every MSM / FFT item is bit-exact, and the reference's TurboVerifier accepts the proof."""
    from oracle.oracle import RefProver, prover_available
    if not prover_available():
        pytest.skip("oracle/_ref/libbbprover.so absent on this machine")
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    P = RefProver(1 << 12, 13, oracle.srs_powers(x, (2 << 12) + 1), x)
    srs = bbg.srs_register(P.monomials())
    # + round 3's z (grand product, blinding, ifft) and round 6's opening polynomials (accumulation + Kate division)
    proof = P.prove(callback_engines.Round346Engine(bbg, srs), check=True)
    assert P.round4_mismatch == 0 and P.mismatches == 0
    assert len(proof) > 0 and P.verify() == 1
    srs.free()
    P.free()


# ---------------------------------------------------------------------------------------------- the C++ drop-in shim
def test_shim_reference_api_on_gpu():
    """oracle/_ref/shim_check: barretenberg's own TUs + shim/bbg_barretenberg_shim.cpp, MSM/FFT entry points wrapped at
    link time onto libbbg.so.  The same binary calls pippenger_unsafe / fft / ... through the reference's C++ signatures
    (GPU) and the reference's CPU bodies (__real_*), and compares with the reference's own operator==."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "shim_check")
    if not os.path.exists(exe):
        pytest.skip("prebuilt oracle/_ref/shim_check not shipped")
    flags = open("/proc/cpuinfo").read()
    if not all(f in flags for f in (" adx", " bmi2", " avx2")):
        pytest.skip("host CPU lacks the ISA the reference build uses")
    r = subprocess.run([exe, "14"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "shim_check PASS" in out, out
    # the same binary with the device group switched on: four contexts (all on device 0 here; one per GPU on a node), every MSM of
    # >= 1000 points over a cached table is sharded by point range through bbg_multi_msm behind pippenger_unsafe
    env = dict(os.environ, BBG_SHIM_DEVICES="0,0,0,0", BBG_SHIM_MULTI_MIN_POINTS="1000")
    r = subprocess.run([exe, "14"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    out = r.stdout.decode()
    assert r.returncode == 0 and "shim_check PASS" in out, out


def test_bench_contract_and_dist_path():
    """bench.py prints one JSON line with the contract's fields; BBG_FORCE_DIST=1 drives the RCCL all-gather +
    software-pipelined sharded-MSM path (world of 1) and the result must still be bit-exact against the CPU reference."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBG_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("MASTER_PORT", None)  # bench.py picks a free port for its world of one: two suites on one box cannot collide
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--log2n", "16", "--config5-log2n", "18", "--real-prover-log2", "10"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["cpu_baseline"]["gpu_bit_exact_vs_cpu"] is True
    assert out["roofline"]["frac"] > 0 and out["value"] > 0
    assert len(out["extra"]["timed_blocks_ms"]) == 5 and abs(out["ms_per_step"] * 3 - sorted(out["extra"]["timed_blocks_ms"])[2]) < 1e-2
    assert out["extra"]["prover_shaped"]["proof_ms"] > 0 and out["extra"]["config5"]["msm_ms"] > 0 and out["extra"]["config5"]["ntt_ms"] > 0
    # N = 1: the headline stays BASELINE's metric; the strong-scaling series' first point sits beside it
    assert out["scaling"] == "weak" and out["strong_scaling"]["n_gpus"] == 1 and out["strong_scaling"]["value"] > 0
    assert out["strong_scaling"]["bit_exact_vs_reference"]["msm"] is True and "weak_scaling_step" not in out["extra"]
    assert "profile_matches_build" in out["roofline"] and "profile_stamp" in out["extra"]
    rp = out["cpu_baseline"].get("real_prover")
    if rp is not None and "error" not in rp:  # oracle/_ref prover libraries shipped: the real reference prover, CPU vs link-time shim
        assert rp["byte_identical_to_cpu_proof"] is True and rp["verified"] == [1, 1, 1, 1] and rp["wrapped_zero_edits_ms"] < rp["link_only_ms"]
        assert out["extra"]["host_path"]["shim_linked_proof_ms"]["construct_proof_wrapped_too"] == rp["wrapped_zero_edits_ms"]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launched_ranks_one_device_rehearsal(world):
    """`python bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): bench.py starts its N ranks itself, and with
    BBG_DIST_ONE_DEVICE=1 all of them share device 0 and exchange through gloo with host staging (parallel.HostStagedDist) -- the whole
    N-rank bench on real kernels: ShardedMsmPipeline at depth 4 with its side stream and side context per rank, the max-over-ranks
    timing, and config 5 (one MSM sharded by point range + all-gather of N partials; one coset NTT sharded by residue class + the
    all-to-all + the cross-rank DFT), checked against the REFERENCE's recorded results (tests/golden/config5_small.json).
    Reference precedent for the split: ecc/curves/bn254/scalar_multiplication/c_bind.cpp:31-46 (pippenger_unsafe(from, range) + g1_sum),
    plonk/proof_system/prover/work_queue.hpp:166-199.  What this cannot show: RCCL and peer access between two distinct devices."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "BBG_FORCE_DIST")}
    env.update(BBG_DIST_ONE_DEVICE="1", BBG_BENCH_LAUNCH_TIMEOUT="800")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--blocks", "2",
                        "--log2n", "14", "--config5-log2n", "16"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, lines, r.stderr.decode()[-3000:])
    out = json.loads(lines[0])
    # N > 1: the headline IS the strong-scaling workload (BASELINE config 5 through the contract's timed region); the per-GPU step whose
    # N-fold repetition scales by construction sits under extra.weak_scaling_step
    assert "error" not in out and out["n_gpus"] == world and out["value"] > 0 and out["ms_per_step"] > 0 and out["scaling"] == "strong", out
    assert "config 5" in out["metric"] and "BASELINE config 5" in out["config"]["workload"] and out["config"]["log2n"] == 16, out["config"]
    assert "REHEARSAL" in out["config"]["exchange"]
    ss = out["strong_scaling"]
    assert ss["n_gpus"] == world and ss["value"] == out["value"] and ss["ms_per_step"] == out["ms_per_step"], ss
    assert abs(out["value"] - (1 << 16) / (out["ms_per_step"] * 1e-3) / 1e6) < 0.01 * out["value"]
    assert ss["bit_exact_vs_reference"]["msm"] is True and ss["bit_exact_vs_reference"]["ntt"] is True
    weak = out["extra"]["weak_scaling_step"]
    assert weak["scaling"] == "weak" and weak["value"] > 0 and "n=2^14" in weak["metric"], weak
    assert out["roofline"]["avg_launch_ms"] > 0 and out["roofline"]["algorithmic_bytes"] == 96.0 * ((1 << 16) // world)
    c5 = out["extra"]["config5"]
    assert len(c5["timed"]["blocks_ms"]) == 2 and c5["timed"]["steps"] == 6
    assert c5["n_gpus"] == world and c5["msm_ms"] > 0 and c5["ntt_ms"] > 0, c5
    assert c5["bit_exact_vs_reference"]["msm"] is True and c5["bit_exact_vs_reference"]["ntt"] is True, c5
    assert c5["exchange"]["msm"] == "all_gather %d x 96 B" % world and not c5["exchange"]["ntt"].startswith("all_to_all 0.0"), c5
    assert "cpu_baseline" not in out  # rank 0 runs the CPU leg at N = 1 only


@pytest.mark.parametrize("lg", [25, 26, 27, 28])
def test_ntt_above_2_24_vs_reference(pkg, oracle, lg):
    """The upper part of bbg_ntt's accepted range (log2n <= 28 = the 2-adicity of BN254 Fr, fr.hpp:27-30; a 2^24-gate key has a 2^26
    "large" domain, proving_key.cpp:21-22): REFERENCE digests of fft / ifft / coset_fft / coset_ifft at 2^25 and 2^26, coset_fft with
    generator_size = n / 4 at 2^26 (the prover's zero-extended input on its 4n domain, evaluation_domain.cpp:57-76), fft and coset_ifft at
    2^27 and 2^28 (tests/golden/ntt_large.json, recorded from the compiled reference by gen_golden_ntt_large.py); then the round trips
    (polynomial_arithmetic.test.cpp:70-134) and a Horner spot value at full size.  Bit-exact on canonical values.  A context of its own,
    released at the end: the tables of a 2^28 domain are 5 x 32 n bytes = 40 GiB (asserted through bbg_memory_report)."""
    import torch
    n = 1 << lg
    recs = [r for r in _ntt_large_golden() if r["log2n"] == lg]
    assert recs, "no reference digests for 2^%d" % lg
    ctx = pkg.Bbg(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # torch stages the buffers (copy_) on its stream: the library must run behind them
    try:
        a = pkg.synthetic_scalars(900 + lg, n)
        ta = torch.from_numpy(a.view(np.int64).reshape(-1)).cuda()
        work = torch.empty_like(ta)
        for rec in recs:
            work.copy_(ta)
            ctx.ntt_device(work.data_ptr(), lg, rec["op"], generator_size=rec["generator_size"])
            ctx.sync()
            out = oracle.canon(0, work.cpu().numpy().view(np.uint64).reshape(n, 4))
            for i, want in rec["spots"].items():
                assert np.array_equal(out[int(i)], unhex(want)[0]), (lg, rec["op"], rec["generator_size"], "spot", i)
            assert sha(out) == rec["sha256"], (lg, rec["op"], rec["generator_size"])
            del out
        rep = ctx.memory_report()
        assert rep["ntt_domains"] == 1 and 4 * 32 * n <= rep["ntt_tables"] <= 6 * 32 * n, rep
        # round trips, compared on the device: canonical(x) = x - r where x >= r is what fr_reduce_once does; here both sides go through
        # one more forward transform instead -- equal residues give equal canonical outputs -- so only two small host arrays are compared
        a_canon_spots = oracle.canon(0, a[:4096])
        for fwd, inv in ((FFT, IFFT), (COSET_FFT, COSET_IFFT)):
            work.copy_(ta)
            ctx.ntt_device(work.data_ptr(), lg, fwd)
            ctx.ntt_device(work.data_ptr(), lg, inv)
            ctx.sync()
            back = work.cpu().numpy().view(np.uint64).reshape(n, 4)
            assert np.array_equal(oracle.canon(0, back[:4096]), a_canon_spots), (lg, fwd, "round trip, head")
            assert np.array_equal(oracle.canon(0, back[n - 4096:]), oracle.canon(0, a[n - 4096:])), (lg, fwd, "round trip, tail")
            step = max(1, n // 65536)
            assert np.array_equal(oracle.canon(0, np.ascontiguousarray(back[::step])), oracle.canon(0, np.ascontiguousarray(a[::step]))), (lg, fwd, "round trip, stride")
            del back
        # A_1 = sum_j a_j w^j by Horner on the device helper (bbg_poly_evaluate_device) against the transform's own output
        work.copy_(ta)
        ctx.ntt_device(work.data_ptr(), lg, FFT)
        ctx.sync()
        got1 = oracle.canon(0, work[4:8].cpu().numpy().view(np.uint64).reshape(1, 4))
        w = oracle.root_of_unity(lg)
        ev = ctx.poly_evaluate_device(ta.data_ptr(), n, w)
        assert np.array_equal(oracle.canon(0, np.asarray(ev, dtype=np.uint64).reshape(1, 4)), got1), (lg, "A_1 vs Horner")
    finally:
        ctx.close()
        torch.cuda.empty_cache()


@pytest.mark.parametrize("G,lg,inverse,coset", [(2, 12, False, False), (4, 12, False, True), (8, 13, False, False), (8, 12, True, False),
                                                 (4, 14, True, True)])
def test_sharded_ntt_device_blocks(pkg, oracle, bbg, G, lg, inverse, coset):
    """The device building blocks of parallel.ntt_sharded (scale_powers, local NTT, cross-rank DFT) on ONE GPU: the G ranks
    are emulated one after another and the all-to-all by slicing; the result must equal the oracle's whole transform.
    (The all-to-all plumbing itself is covered by tests/test_distributed_cpu.py on gloo.)"""
    import importlib
    import torch
    par = importlib.import_module("aztec_amd.parallel")
    ops = par.BbgNttOps(bbg)
    n = 1 << lg
    m, lenq, log2g = n // G, n // G // G, G.bit_length() - 1
    a = oracle.canon(0, pkg.synthetic_scalars(600 + lg + G, n))
    five = oracle.to_mont(0, np.array([[5, 0, 0, 0]], dtype=np.uint64))[0]
    if inverse:
        want = oracle.ntt(a, 3 if coset else 1)
    else:
        want = oracle.ntt(a, 2 if coset else 0)
    Z = []
    for g in range(G):
        x = torch.from_numpy(a[g::G].copy().view(np.int64).reshape(-1)).cuda()
        if coset and not inverse:
            ops.scale_powers(x, m, ops.fr_pow(five, G), ops.fr_pow(five, g))
        ops.ntt(x, lg - log2g, 1 if inverse else 0)
        start = par._mont_limbs(pow(G, -1, par._R_MOD)) if inverse else None
        ops.scale_powers(x, m, ops.root_pow(lg, g, inverse), start)
        Z.append(x)
    bbg.sync()
    got = np.zeros((n, 4), dtype=np.uint64)
    for r in range(G):
        recv = torch.cat([Z[s][4 * r * lenq: 4 * (r + 1) * lenq] for s in range(G)])
        out = torch.empty_like(recv)
        ops.cross_dft(recv, out, log2g, lenq, lg, inverse)
        bbg.sync()
        o = out.cpu().numpy().view(np.uint64).reshape(G, lenq, 4)
        for t in range(G):
            got[r * lenq + m * t: r * lenq + m * t + lenq] = o[t]
    got = oracle.canon(0, got)
    if inverse and coset:  # coset_ifft = ifft then g^-j over the whole domain (polynomial_arithmetic.cpp:480-484)
        ginv = oracle.fe_inv(0, five)[0]
        cur = oracle.to_mont(0, np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
        for j in range(n):
            got[j] = oracle.fe_mul(0, got[j], cur)[0]
            cur = oracle.fe_mul(0, cur, ginv)[0]
    assert np.array_equal(got, want)


# ---------------------------------------------------------------------------------------------- polynomial helpers (8f-2 / 8f-4)
def test_poly_helpers_vs_reference_golden(pkg, oracle, bbg, golden):
    """Device add/sub/mul, evaluate, Kate opening quotient and division by Z*_H against the outputs recorded from the
    compiled reference, plus the oracle on ragged sizes."""
    import torch
    kc = unhex(golden["ntt_constant"])[0]

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1)).cuda()

    def host(t, n):
        bbg.sync()
        return t.cpu().numpy().view(np.uint64).reshape(n, 4)

    for rec in golden["poly"]["binop"]:
        n = 1 << rec["log2n"]
        a, b = dev(pkg.synthetic_scalars(rec["seed_a"], n)), dev(pkg.synthetic_scalars(rec["seed_b"], n))
        r = torch.empty_like(a)
        bbg.poly_op_device(rec["op"], a.data_ptr(), b.data_ptr(), r.data_ptr(), n)
        assert sha(oracle.canon(0, host(r, n))) == rec["sha256"], rec
    for n in (1, 2, 15, 16, 17, 255, 4096, 4097, 70000, 1 << 20):
        a_np = pkg.synthetic_scalars(900 + n % 97, n)
        a = dev(a_np)
        assert np.array_equal(bbg.poly_evaluate_device(a.data_ptr(), n, kc), oracle.poly_eval(a_np, kc)), n
    for rec in golden["poly"]["kate"]:
        n = rec["n"]
        a = dev(pkg.synthetic_scalars(rec["seed"], n))
        d = torch.zeros_like(a)
        f = bbg.kate_opening_device(a.data_ptr(), d.data_ptr(), n, kc)
        assert np.array_equal(f, unhex(rec["f"])[0]), rec
        assert sha(oracle.canon(0, host(d, n))) == rec["dest_sha256"], rec
    for n in (3, 4095, 4096, 8191, 300000, (1 << 20) + 5):  # ragged sizes, several 4096-coefficient blocks and scan rounds
        a_np = pkg.synthetic_scalars(7 + n, n)
        a, d = dev(a_np), dev(np.zeros((n, 4), dtype=np.uint64))
        f = bbg.kate_opening_device(a.data_ptr(), d.data_ptr(), n, kc)
        want_d, want_f = oracle.kate_opening(a_np, kc)
        assert np.array_equal(f, want_f) and np.array_equal(oracle.canon(0, host(d, n)), want_d), n
    for rec in golden["poly"]["dpv"]:
        n = 1 << rec["log2_target"]
        e = dev(pkg.synthetic_scalars(rec["seed"], n))
        bbg.divide_by_pseudo_vanishing_device(e.data_ptr(), rec["log2_src"], rec["log2_target"], rec["cut"])
        assert sha(oracle.canon(0, host(e, n))) == rec["sha256"], rec
    with pytest.raises(pkg.BbgError):
        bbg.divide_by_pseudo_vanishing_device(1, 10, 8, 4)
    # the host-buffer forms the shim binds (upload, compute, download), incl. the in-place Kate call of batch_open
    for n in (1, 17, 4097, 70000):
        a_np = pkg.synthetic_scalars(40 + n, n)
        assert np.array_equal(bbg.poly_evaluate(a_np, kc), oracle.poly_eval(a_np, kc)), n
        want_d, want_f = oracle.kate_opening(a_np, kc)
        for in_place in (False, True):
            d, f = bbg.kate_opening(a_np, kc, in_place=in_place)
            assert np.array_equal(f, want_f) and np.array_equal(oracle.canon(0, d), want_d), (n, in_place)
    for rec in golden["poly"]["dpv"][:3]:
        e = pkg.synthetic_scalars(rec["seed"], 1 << rec["log2_target"])
        assert sha(oracle.canon(0, bbg.divide_by_pseudo_vanishing(e, rec["log2_src"], rec["cut"]))) == rec["sha256"], rec


def test_quotient_identity_full_size(pkg, oracle, bbg):
    """End-to-end identity on the prover's sizes (n = 2^18, 4n coset domain): for T(X) = A(X) * Z*_H-multiple the
    device pipeline coset_fft -> pointwise mul -> divide_by_pseudo_vanishing -> coset_ifft recovers a polynomial whose
    product with Z*_H matches; here checked in the cheaper direction: dividing then multiplying back by the oracle's
    pointwise inverse factors is the identity, and the kate quotient satisfies W(X) (X - z) = F(X) - F(z) at a random point."""
    import torch
    lg = 18
    n = 1 << lg
    kc = pkg.synthetic_scalars(4242, 1)[0]
    f_np = pkg.synthetic_scalars(515, n)
    f = torch.from_numpy(f_np.view(np.int64).reshape(-1)).cuda()
    w = torch.zeros_like(f)
    fz = bbg.kate_opening_device(f.data_ptr(), w.data_ptr(), n, kc)
    x = pkg.synthetic_scalars(99, 1)[0]
    wx = bbg.poly_evaluate_device(w.data_ptr(), n, x)
    fx = bbg.poly_evaluate_device(f.data_ptr(), n, x)
    lhs = oracle.fe_mul(0, wx, oracle.fe_sub(0, x, kc))[0]
    rhs = oracle.fe_sub(0, fx, fz)[0]
    assert np.array_equal(lhs, rhs)


# ---------------------------------------------------------------------------------------------- round 2: sizes and flavours the driver had not seen
def test_msm_2_24_golden_from_reference_shards(pkg, oracle, bbg):
    """BASELINE config 5's MSM size on one GPU: n = 2^24 through the automatic window choice.  The expectation is the REFERENCE's:
    sixteen point-range shards of 2^20 terms, each a reference pippenger_unsafe, and the reference's g1 sum of the partials
    (tests/golden/msm24.json, gen_golden_msm24.py) -- the composition the reference's own C binding uses for large MSMs
    (pippenger.cpp:27-31, c_bind.cpp:31-46).  Every shard is also reproduced through (from, range) on the 2^24-point SRS."""
    with open(os.path.join(os.path.dirname(__file__), "golden", "msm24.json")) as f:
        G = json.load(f)
    n, s = 1 << G["log2n"], 1 << G["shard_log2n"]
    srs = bbg.srs_synth_hashed(G["srs_seed"], n)
    try:
        assert sha(srs.read()) == G["points_sha256"]
        sc = pkg.synthetic_scalars(G["scalar_seed"], n)
        whole = oracle.jac_to_affine(bbg.msm(srs, sc))
        assert np.array_equal(whole, unhex(G["result"], 8)[0]), "2^24 MSM differs from the reference's sharded result"
        parts = []
        for rec in G["shards"]:
            part = bbg.msm(srs, sc[rec["from"]: rec["from"] + rec["n"]], start=rec["from"])
            assert np.array_equal(oracle.jac_to_affine(part), unhex(rec["result"], 8)[0]), rec["from"]
            parts.append(part)
        assert np.array_equal(oracle.jac_to_affine(bbg.g1_sum(np.stack(parts))), whole)
    finally:
        srs.free()


def test_srs_transcript_writer_and_reference_reader(pkg, oracle, bbg, tmp_path):
    """The product WRITER (bbg_srs_write_transcript) against the reference's own READER io::read_transcript_g1 (srs/io.cpp:134-162,
    compiled from the reference into oracle/_ref/libbbprover.so) and against the product reader: all three agree, over several
    files, with a partial last file, and every file's BLAKE2b checksum is the one hashlib computes."""
    import hashlib
    import struct
    from oracle.oracle import RefProver, prover_available
    n = 1000
    pts = oracle.srs_hashed(4242, n)
    pts[0] = oracle.g1_generator()  # monomials[0] = G in every transcript-backed SRS
    srs = bbg.srs_register(pts)
    g2 = bytes(range(128))
    srs.write_transcript(tmp_path, points_per_file=300, g2_x_raw=g2)
    files = sorted(os.listdir(tmp_path))
    assert files == ["transcript00.dat", "transcript01.dat", "transcript02.dat", "transcript03.dat"]
    for k, name in enumerate(files):
        raw = open(tmp_path / name, "rb").read()
        m = struct.unpack(">7I", raw[:28])
        count = 300 if k < 3 else 99
        assert m == (k, 4, n - 1, 1, count, 1 if k == 0 else 0, 300 * k), m
        assert len(raw) == 28 + 64 * count + (128 if k == 0 else 0) + 64
        assert raw[-64:] == hashlib.blake2b(raw[:-64]).digest(), name
        if k == 0:
            assert raw[28 + 64 * count: 28 + 64 * count + 128] == g2
    back = bbg.srs_load_transcript(tmp_path, n)
    assert np.array_equal(back.read(), oracle.canon(1, pts.reshape(-1, 4)).reshape(n, 8))
    back.free()
    if prover_available():
        x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
        P = RefProver(16, 1, oracle.srs_powers(x, 40), x)  # any session: only its library's reader is used
        for degree in (n, 301, 2):
            ref = P.read_transcript_g1(tmp_path, degree)
            assert np.array_equal(oracle.canon(1, ref.reshape(-1, 4)).reshape(degree, 8), oracle.canon(1, pts[:degree].reshape(-1, 4)).reshape(degree, 8))
            mine = bbg.srs_load_transcript(tmp_path, degree)
            assert np.array_equal(mine.read(), oracle.canon(1, ref.reshape(-1, 4)).reshape(degree, 8)), degree
            mine.free()
        P.free()
    srs.free()


def _powers_srs(oracle, count):
    x = oracle.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    return x, oracle.srs_powers(x, count)


@pytest.mark.parametrize("flavour,log2_gates", [(0, 9), (1, 9), (2, 9), (3, 9), (4, 9), (6, 9), (0, 13), (1, 13), (2, 13), (3, 13), (4, 13), (6, 13)])
def test_resident_prover_reproduces_the_reference_proof(pkg, oracle, bbg, flavour, log2_gates):
    """shim/bbg_resident_prover.hpp + bbg_prover_* (every O(n) step of the proof on the device, C++ host, no Python in the
    product path) against the reference CPU prover on the SAME randomness: the reference's construct_proof runs round by round
    on the host and the blinding scalars it draws are recorded; the resident prover replays them over a second session of the
    same circuit.  Transcript, commitments, evaluations -- the proof bytes -- must be IDENTICAL, for the provers of all three
    composers of the reference (TurboComposer, StandardComposer, MiMCComposer: flavours 0, 1, 2) and for the UNROLLED Turbo / Standard
    provers (3, 4: create_unrolled_prover -- no linearisation polynomial, every polynomial opened, Pedersen-Blake2s transcript; what
    the rollup circuits use), and the reference verifier must accept.  Flavour 6 = TurboPLONK over an arithmetic-only circuit (the fixed-base /
    range / logic selectors vanish on every gate row; the reference still sets their last row to 1, composer_base.cpp:186).  The device-derived forms of the proving key's polynomials (sigma
    in Lagrange base, 4n-coset forms, L_1) must equal the arrays the reference's compute_proving_key produced."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << log2_gates) + 2)
    A = RefProver(1 << log2_gates, 21 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind = A.prove_recording()
    assert A.verify() == 1
    B = RefProver(1 << log2_gates, 21 + flavour, pts, x, gpu_linked=True, flavour=flavour)
    assert B.n == A.n
    assert B.resident_check_key() == 0
    proof_gpu, secs = B.prove_resident(blind)
    assert B.verify() == 1
    assert proof_gpu == proof_cpu, f"resident proof differs from the reference CPU proof (flavour {flavour}, n = {A.n})"
    # fresh randomness: a different, valid proof
    C = RefProver(1 << log2_gates, 21 + flavour, pts, x, gpu_linked=True, flavour=flavour)
    proof_fresh, _ = C.prove_resident()
    assert C.verify() == 1 and proof_fresh != proof_cpu
    # a second proof over the same session and key handle (per-proof state is rebuilt, per-key state reused)
    proof_again, _ = C.prove_resident()
    assert C.verify() == 1 and proof_again != proof_fresh
    for P in (A, B, C):
        P.free()


@pytest.mark.parametrize("log2_gates", [9, 12])
def test_resident_prover_with_every_turbo_widget_active(pkg, oracle, bbg, log2_gates):
    """The Turbo circuits of the tests above carry satisfied range and AND / XOR constraints besides their arithmetic gates; the
    fixed-base widget's selectors are non-zero only here: flavour 5 adds fixed-base gates over ARBITRARY witnesses.  Such a proof
    cannot verify (the reference prover does not care, prover.cpp never checks the witness) -- what is asserted is that the resident
    prover emits the reference prover's bytes on the same randomness: quotient and linearisation terms of all four Turbo widgets."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << log2_gates) + 2)
    A = RefProver(1 << log2_gates, 77, pts, x, flavour=5)
    proof_cpu, blind = A.prove_recording()
    assert A.verify() == 0
    B = RefProver(1 << log2_gates, 77, pts, x, gpu_linked=True, flavour=5)
    assert B.resident_check_key() == 0
    proof_gpu, _ = B.prove_resident(blind)
    assert proof_gpu == proof_cpu
    A.free()
    B.free()


@pytest.mark.parametrize("flavour,log2_gates", [(0, 8), (0, 13), (1, 13), (3, 14)])
def test_resident_prover_divides_the_quotient_either_way(pkg, oracle, bbg, flavour, log2_gates):
    """Round 4 divides the quotient's 4n evaluations by Z*_H (prover.cpp:337, polynomial_arithmetic.cpp:680-720) either inside the
    coset iFFT's first load, from a per-point divisor table (option prover_fused_divide = 1, the default -- the legs of every other
    prover test), or in a pass of its own (0): the same proof bytes as the reference CPU prover both ways (n = 2^8: the 4n domain is a
    single-pass transform with no fused load, the separate pass runs under either setting)."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << log2_gates) + 2)
    A = RefProver(1 << log2_gates, 41 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind = A.prove_recording()
    assert A.verify() == 1
    proofs = {}
    for fused in (0, 1, 0):
        B = RefProver(1 << log2_gates, 41 + flavour, pts, x, gpu_linked=True, flavour=flavour)
        B.shim_option("prover_fused_divide", fused)
        B.shim_option("quotient_setup_plan", fused)  # with it the one-lane chain of the widgets' set-up blocks (default: a wave's lanes side by side)
        B.shim_option("prover_early_cosets", fused)  # and the wires' coset forms in front of round 3 / behind round 1 (default: by circuit size)
        try:
            proofs[fused], _ = B.prove_resident(blind)
            assert B.verify() == 1
        finally:
            B.shim_option("prover_fused_divide", 1)
            B.shim_option("quotient_setup_plan", 1)
            B.shim_option("prover_early_cosets", -1)
            B.free()
        assert proofs[fused] == proof_cpu, f"fused_divide = {fused}: proof differs from the reference CPU proof (flavour {flavour}, n = {A.n})"
    A.free()


@pytest.mark.parametrize("flavour", [0, 1, 2, 3, 4])
def test_reference_provers_linked_against_shim(pkg, oracle, bbg, flavour):
    """INTEGRATION.md 2a for both composers: TurboComposer::create_prover (turbo_composer.cpp:727) and
    StandardComposer::create_prover (standard_composer.cpp:562) produce provers whose construct_proof(), UNMODIFIED and called
    as one function, runs its MSM / FFT entry points on the GPU through --wrap; the reference verifiers accept."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 12) + 2)
    P = RefProver(1 << 12, 31, pts, x, gpu_linked=True, flavour=flavour)
    proof = P.prove_reference()
    assert len(proof) == (1248, 864, 928, 1536, 1056)[flavour] and P.verify() == 1  # one public input each (two for MiMC)
    P.free()


@pytest.mark.parametrize("flavour,log2_gates", [(0, 9), (1, 9), (2, 9), (3, 9), (4, 9), (0, 13), (1, 13), (2, 13), (3, 13), (4, 13)])
def test_wrapped_construct_proof_reproduces_the_reference_proof(pkg, oracle, bbg, flavour, log2_gates):
    """ZERO source edits (INTEGRATION.md 2a'): oracle/_ref/libbbprover_wrap.so is the CPU build's own driver object -- no glue call, a plain
    `prover->construct_proof()` -- linked with shim/bbg_barretenberg_shim.cpp + shim/bbg_prover_wrap.cpp and both --wrap flag files.  The call
    reaches the resident prover through the wrapped symbol of ProverBase<settings>::construct_proof (prover.cpp:420-436, :445-448) for all five
    prover types the reference's composers build; on the blinding scalars a reference CPU proof drew, the proof bytes are IDENTICAL to the CPU
    prover's and the reference verifier accepts.  Opting out (bbg_shim_resident_set_enabled(0) / BBG_SHIM_RESIDENT=0) gives the reference body
    back (MSM / FFT entry points still on the GPU)."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << log2_gates) + 2)
    A = RefProver(1 << log2_gates, 21 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind = A.prove_recording()
    assert A.verify() == 1
    B = RefProver(1 << log2_gates, 21 + flavour, pts, x, wrap_linked=True, flavour=flavour)
    before = B.wrap_stats()
    proof = B.prove_reference(replay=blind)
    after = B.wrap_stats()
    assert after[0] == before[0] + 1 and after[1] == before[1], "construct_proof() did not take the resident path"
    assert B.verify() == 1
    assert proof == proof_cpu, f"wrapped construct_proof(): proof differs from the reference CPU proof (flavour {flavour}, n = {A.n})"
    # the same prover object again (ProverBase::reset), fresh randomness from the kernel CSPRNG: another valid proof, same cached key
    keys = B.wrap_cached_keys()
    again = B.prove_reference(reset=True)
    assert B.verify() == 1 and again != proof and B.wrap_cached_keys() == keys
    # opt out: the reference's own body (its MSMs / FFTs through the wrapped entry points)
    C = RefProver(1 << log2_gates, 21 + flavour, pts, x, wrap_linked=True, flavour=flavour)
    try:
        C.wrap_set_enabled(False)
        stats = C.wrap_stats()
        link_only = C.prove_reference()
        assert C.verify() == 1 and C.wrap_stats() == stats and len(link_only) == len(proof)
    finally:
        C.wrap_set_enabled(True)
    for P in (A, B, C):
        P.free()
    assert B.wrap_trim() <= keys - 1  # the sessions are gone: their keys' device copies were released


def test_wrapped_construct_proof_key_cache(pkg, oracle, bbg):
    """The key cache of shim/bbg_prover_wrap.cpp: two proving keys alive at once (two provers proving alternately), an entry released when
    its key's last outside owner is gone, least-recently-used eviction under a byte budget -- never of the key being proved with --, and a
    re-upload after an eviction giving the same proof bytes."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 11) + 2)
    P1 = RefProver(1 << 11, 91, pts, x, wrap_linked=True, flavour=0)
    P2 = RefProver(1 << 10, 92, pts, x, wrap_linked=True, flavour=1)
    P1.wrap_clear()
    assert P1.wrap_cached_keys() == 0 and P1.wrap_bytes() == 0
    blind = pkg.synthetic_scalars(4711, 15)
    try:
        a1 = P1.prove_reference(replay=blind)
        assert P1.wrap_cached_keys() == 1 and P1.verify() == 1
        one_key = P1.wrap_bytes()
        assert one_key > 21 * 4 * (1 << 11) * 32  # at least the 4n coset forms of the key's polynomials
        b1 = P2.prove_reference(replay=blind[:12])
        assert P2.wrap_cached_keys() == 2 and P2.verify() == 1 and P2.wrap_bytes() > one_key
        a2 = P1.prove_reference(replay=blind, reset=True)  # alternating between the two keys: no re-upload, same bytes
        b2 = P2.prove_reference(replay=blind[:12], reset=True)
        assert a2 == a1 and b2 == b1 and P1.wrap_cached_keys() == 2
        ev0 = P1.wrap_stats()[2]
        P1.wrap_set_budget(one_key)  # room for ONE Turbo key: applied at once, the least recently used entry (P1's: P2 proved last) goes
        assert P1.wrap_cached_keys() == 1 and P1.wrap_stats()[2] == ev0 + 1
        a3 = P1.prove_reference(replay=blind, reset=True)  # re-uploaded after its eviction: same proof bytes; now P2's key has to go
        assert a3 == a1 and P1.wrap_cached_keys() == 1 and P1.wrap_stats()[2] == ev0 + 2
        P1.wrap_set_budget(1)  # below any key: the key in use survives its own proof, everything else goes
        b3 = P2.prove_reference(replay=blind[:12], reset=True)
        assert b3 == b1 and P2.wrap_cached_keys() == 1 and P2.wrap_stats()[1] == 0
        P1.wrap_set_budget(0)  # 0 = the default budget again (half of the device) from the next proof on
        a4 = P1.prove_reference(replay=blind, reset=True)
        assert a4 == a1 and P1.wrap_cached_keys() == 2
        P2.free()
        P2 = None
        assert P1.wrap_trim() == 1  # key destroyed -> entry released
        P1.free()
        P1 = None
        ref = RefProver(1 << 9, 93, pts, x, wrap_linked=True, flavour=0)
        assert ref.wrap_trim() == 0
        ref.free()
    finally:
        for P in (P1, P2):
            if P is not None:
                P.wrap_set_budget(0)
                P.free()


@pytest.mark.parametrize("fail_round", [1, 3, 4, 5, 6])
def test_wrapped_construct_proof_survives_a_device_error(pkg, oracle, bbg, fail_round):
    """The reference's construct_proof() (prover.cpp:420-436) never throws for a device's sake, so the wrapped one must not either: a device
    round that fails in the middle of a resident proof (library option prover_fail_round, tests only) makes shim/bbg_prover_wrap.cpp drop the
    key's device copy, rebuild the transcript (ProverBase::reset, prover.cpp:438-442) and repeat the proof with the reference body -- a valid
    proof comes back, the fallback is counted, and the next proof over that key takes the resident path again (round-4 advisor finding)."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 10) + 2)
    B = RefProver(1 << 10, 61, pts, x, wrap_linked=True, flavour=0)
    C = None
    try:
        first = B.prove_reference()
        assert B.verify() == 1
        keys, st0 = B.wrap_cached_keys(), B.wrap_stats()
        B.wrap_fail_round(fail_round)
        second = B.prove_reference(reset=True)
        st1 = B.wrap_stats()
        assert B.verify() == 1 and len(second) == len(first) and second != first
        assert st1[1] == st0[1] + 1 and B.wrap_cached_keys() == keys - 1, (st0, st1)
        # a new prover over the same circuit: the key is uploaded again and the proof is the resident one
        C = RefProver(1 << 10, 61, pts, x, wrap_linked=True, flavour=0)
        C.prove_reference()
        st2 = C.wrap_stats()
        assert C.verify() == 1 and st2[0] == st1[0] + 1 and st2[1] == st1[1]
    finally:
        B.wrap_fail_round(0)
        B.free()
        if C is not None:
            C.free()


def test_wrapped_construct_proof_reuploads_a_rewritten_key(pkg, oracle, bbg):
    """A host that rewrites a polynomial of a proving key it has already proved with: the key cache is keyed by the key's address, so without
    a check the device copy would be stale and the proof silently wrong.  shim/bbg_prover_wrap.cpp fingerprints the key's polynomials per
    proof (buffer address, size, sampled coefficients) and uploads again on a change: after q_m is rewritten in both builds' keys, the
    wrapped proof still equals the CPU prover's byte for byte (neither verifies any more -- the circuit no longer matches the key)."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 10) + 2)
    A = RefProver(1 << 10, 62, pts, x, flavour=0)
    B = RefProver(1 << 10, 62, pts, x, wrap_linked=True, flavour=0)
    try:
        before = B.prove_reference()
        assert B.verify() == 1
        re0, keys = B.wrap_reuploads(), B.wrap_cached_keys()
        again = B.prove_reference(reset=True)  # untouched key: no upload
        assert B.wrap_reuploads() == re0 and B.verify() == 1 and len(again) == len(before)
        A.key_selector_scale3("q_m")
        B.key_selector_scale3("q_m")
        proof_cpu, blind = A.prove_recording()
        proof = B.prove_reference(replay=blind, reset=True)
        assert B.wrap_reuploads() == re0 + 1 and B.wrap_cached_keys() == keys
        assert proof == proof_cpu, "stale device copy of a rewritten proving key"
    finally:
        A.free()
        B.free()


@pytest.mark.parametrize("rounds", [False, True])
def test_wrapped_proof_over_a_key_with_one_poked_coefficient(pkg, oracle, bbg, rounds):
    """The silent-wrong path rounds 4-5 left open: ONE coefficient of a cached proving key rewritten at a row the sampled fingerprint does not
    look at (16 evenly spaced rows per polynomial).  Round 6 (shim/bbg_shim_verify.hpp): while the GPU makes the proof, host threads re-hash
    EVERY coefficient the device copy was made from; a mismatch drops the copy, uploads the key as it is now and repeats the proof with the
    same blinding scalars -- the wrapped construct_proof() returns the proof the CPU prover makes over the rewritten key, byte for byte
    (`rounds`: the same through the seven wrapped execute_*_round symbols, where the stale copy sends the proof to the reference rounds: a
    valid-shape proof that the verifier rejects like the CPU prover's, since the circuit no longer matches the key).
    Reference anchor: a proving_key is a plain struct of host polynomials with no change hook (proving_key.cpp:18-27)."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 10) + 2)
    A = RefProver(1 << 10, 63, pts, x, flavour=0)
    B = RefProver(1 << 10, 63, pts, x, wrap_linked=True, flavour=0)
    try:
        B.prove_reference()
        assert B.verify() == 1
        re0, keys = B.wrap_reuploads(), B.wrap_cached_keys()
        A.key_selector_poke("q_m", 1)  # row 1 of 1024: the samples sit at rows (n - 1) k / 15
        B.key_selector_poke("q_m", 1)
        proof_cpu, blind = A.prove_recording()
        if rounds:
            proof, _, _ = B.prove_round_by_round(reset=True)
            assert B.wrap_reuploads() == re0 + 1, "the full check did not see the rewritten coefficient"
            # (the reference rounds that replaced the resident ones left B's witness in coefficient form, as the reference prover does:
            # B is not proved with again -- test_wrapped_construct_proof_survives_a_device_error covers the proof after a fallback)
            assert len(proof) == len(proof_cpu) and B.verify() == A.verify()
            assert B.wrap_cached_keys() == keys - 1, "the stale device copy is gone"
        else:
            proof = B.prove_reference(replay=blind, reset=True)
            assert B.wrap_reuploads() == re0 + 1 and B.wrap_cached_keys() == keys
            assert proof == proof_cpu, "proof over the stale device copy of a key with one rewritten coefficient"
            again = B.prove_reference(replay=blind, reset=True)  # the fresh copy verifies: no further upload
            assert again == proof_cpu and B.wrap_reuploads() == re0 + 1
    finally:
        A.free()
        B.free()


@pytest.mark.parametrize("flavour,log2_gates", [(0, 13), (1, 13), (2, 11), (3, 11), (4, 11)])
def test_wrapped_rounds_reproduce_the_reference_proof(pkg, oracle, bbg, flavour, log2_gates):
    """Hosts that drive the prover ROUND BY ROUND (the reference's C binding: prover_execute_preamble_round ... prover_execute_sixth_round +
    prover_process_queue, plonk/proof_system/prover/c_bind.cpp:9-12, :59-92; the same sequence construct_proof() is made of, prover.cpp:420-436):
    with ProverBase<settings>::execute_*_round wrapped at link time (shim/wrap_flags_prover.txt, shim/bbg_prover_wrap.cpp) each call runs the
    resident device round of the same name, leaves its commitments in the transcript and the work queue EMPTY, and the exported proof equals
    the reference CPU prover's byte for byte on the blinding scalars that proof drew.  All five prover types; zero source edits (the driver
    object is the CPU build's)."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << log2_gates) + 2)
    A = RefProver(1 << log2_gates, 41 + flavour, pts, x, flavour=flavour)
    proof_cpu, blind = A.prove_recording()
    A2 = RefProver(1 << log2_gates, 41 + flavour, pts, x, flavour=flavour)
    _, _, q_cpu = A2.prove_round_by_round()  # the CPU build's rounds queue their MSMs / FFTs for process_queue()
    assert A2.verify() == 1 and sum(q_cpu) > 0
    A2.free()
    B = RefProver(1 << log2_gates, 41 + flavour, pts, x, wrap_linked=True, flavour=flavour)
    try:
        st0 = B.wrap_stats()
        proof, _, q = B.prove_round_by_round(replay=blind)
        st1 = B.wrap_stats()
        assert q == [0] * 7, ("a resident round left work in the queue", q)
        assert st1[0] == st0[0] + 1 and st1[1] == st0[1] and B.wrap_in_progress() == 0
        assert B.verify() == 1
        assert proof == proof_cpu, f"round-by-round resident proof differs from the reference CPU proof (flavour {flavour})"
        # the same prover again after ProverBase::reset(): fresh randomness, same cached key, another valid proof
        keys = B.wrap_cached_keys()
        again, _, q = B.prove_round_by_round(reset=True)
        assert B.verify() == 1 and again != proof and q == [0] * 7 and B.wrap_cached_keys() == keys
        # and construct_proof() on the same object after the rounds: still byte-identical on the replayed scalars
        assert B.prove_reference(replay=blind, reset=True) == proof_cpu
    finally:
        A.free()
        B.free()


def test_wrapped_rounds_reference_mode_and_device_errors(pkg, oracle, bbg):
    """The fall-backs of the round-by-round wrap: (i) a proof whose preamble ran while the wrap was switched off stays on the reference rounds
    when it is switched on afterwards (no resident state to continue from): the queue fills as in the CPU build and the proof verifies;
    (ii) a device error inside round r (library option prover_fail_round) rebuilds the transcript and replays the reference rounds 0 .. r, so the
    host's remaining calls complete a valid proof; the failure is counted and the next proof is resident again."""
    from oracle.oracle import RefProver, prover_available, PROVER_WRAP_SO
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        pytest.skip("oracle/_ref/libbbprover_wrap.so absent on this machine")
    x, pts = _powers_srs(oracle, (2 << 10) + 2)
    P = RefProver(1 << 10, 71, pts, x, wrap_linked=True, flavour=0)
    try:
        P.wrap_set_enabled(False)
        assert P.lib.refp_execute_round(P.h, 0) == 4  # the reference preamble queues the four wire iFFTs
        P.lib.refp_process_queue_reference(P.h)
        P.wrap_set_enabled(True)
        sizes = []
        for k in range(1, 7):
            sizes.append(int(P.lib.refp_execute_round(P.h, k)))
            if k != 5:
                P.lib.refp_process_queue_reference(P.h)
        assert sizes == [4, 0, 6, 4, 0, 2], sizes
        P.lib.refp_export_proof(P.h, None, 0)
        assert P.verify() == 1 and P.wrap_in_progress() == 0
    finally:
        P.wrap_set_enabled(True)
        P.free()
    for fail_round in (1, 3, 4, 5, 6):
        Q = RefProver(1 << 10, 72, pts, x, wrap_linked=True, flavour=0)
        R = None
        try:
            st0 = Q.wrap_stats()
            Q.wrap_fail_round(fail_round)
            proof, _, q = Q.prove_round_by_round()
            st1 = Q.wrap_stats()
            assert Q.verify() == 1 and len(proof) == 1248, fail_round
            assert st1[1] == st0[1] + 1 and Q.wrap_in_progress() == 0, (fail_round, st0, st1)
            # rounds before the failing one ran resident (empty queue); from the failing round on the reference rounds queue their work
            want = [0 if k < fail_round else v for k, v in enumerate([4, 4, 0, 6, 4, 0, 2])]
            assert q == want, (fail_round, q, want)
            R = RefProver(1 << 10, 72, pts, x, wrap_linked=True, flavour=0)
            _, _, q2 = R.prove_round_by_round()
            assert R.verify() == 1 and q2 == [0] * 7 and R.wrap_stats()[1] == st1[1]
        finally:
            Q.wrap_fail_round(0)
            Q.free()
            if R is not None:
                R.free()


def test_turbo_prover_2_20_gates_on_gpu(pkg, oracle, bbg):
    """BASELINE config 4 at its stated size, under the driver-run suite: a 2^20-gate TurboPLONK circuit.
      (a) every MSM / coset-FFT / iFFT work item of the reference prover computed by this library and compared with the reference
          CPU result on the same input: zero mismatching items, verifier accepts;
      (b) the reference CPU proof (recorded randomness) reproduced BYTE FOR BYTE by the resident C++ prover, verifier accepts;
      (c) the unmodified shim-linked prover's construct_proof() verifies."""
    from oracle.oracle import RefProver, prover_available, PROVER_GPU_SO
    if not prover_available() or not os.path.exists(PROVER_GPU_SO):
        pytest.skip("oracle/_ref/libbbprover_gpu.so absent on this machine")
    import time
    log2n = 20
    n = 1 << log2n
    x, pts = _powers_srs(oracle, n + 1)
    gates = n - 64
    A = RefProver(gates, 11, pts, x)
    assert A.n == n
    srs = bbg.srs_register(A.monomials())
    proof = A.prove(callback_engines.FusedFftEngine(bbg, srs), check=True)
    assert A.mismatches == 0 and A.counts == [11, 5, 4], (A.mismatches, A.counts)
    assert len(proof) == 1248 and A.verify() == 1
    srs.free()
    A.free()
    A = RefProver(gates, 11, pts, x)
    t0 = time.perf_counter()
    proof_cpu, blind = A.prove_recording()
    t_cpu = time.perf_counter() - t0
    assert A.verify() == 1
    A.free()
    B = RefProver(gates, 11, pts, x, gpu_linked=True)
    t_key = B.resident_key_create()
    proof_gpu, t_gpu = B.prove_resident(blind)
    assert B.verify() == 1
    assert proof_gpu == proof_cpu, "2^20-gate resident proof differs from the reference CPU proof"
    _, t_warm = B.prove_resident()  # second proof over the same key: scratch, tables and window tables are in place
    assert B.verify() == 1
    B.free()
    C = RefProver(gates, 11, pts, x, gpu_linked=True)
    t0 = time.perf_counter()
    proof_shim = C.prove_reference()
    t_shim = time.perf_counter() - t0
    assert len(proof_shim) == 1248 and C.verify() == 1
    C.free()
    # (d) zero source edits: the CPU build's driver object with construct_proof() itself wrapped -- byte-identical, and fast
    from oracle.oracle import PROVER_WRAP_SO
    t_wrap_first = t_wrap = float("nan")
    if os.path.exists(PROVER_WRAP_SO):
        D = RefProver(gates, 11, pts, x, wrap_linked=True)
        t0 = time.perf_counter()
        proof_wrap = D.prove_reference(replay=blind)  # includes the one-off key upload + derivation of this circuit
        t_wrap_first = time.perf_counter() - t0
        assert D.verify() == 1
        assert proof_wrap == proof_cpu, "2^20 gates: the wrapped construct_proof() differs from the reference CPU proof"
        ts = []
        for _ in range(3):
            D.lib.refp_reset(D.h)
            t0 = time.perf_counter()
            D.prove_reference()
            ts.append(time.perf_counter() - t0)
        t_wrap = min(ts)
        assert D.verify() == 1
        assert t_wrap < 0.060, f"wrapped construct_proof() at 2^20 gates took {t_wrap*1e3:.1f} ms (resident path not taken?)"
        # (e) the same prover driven ROUND BY ROUND through the seven wrapped execute_*_round symbols (c_bind.cpp:59-92)
        proof_rounds, _, q = D.prove_round_by_round(replay=blind, reset=True)
        assert proof_rounds == proof_cpu and q == [0] * 7, "2^20 gates: round-by-round resident proof differs from the reference CPU proof"
        t_rounds = min(D.prove_round_by_round(reset=True)[1] for _ in range(3))
        assert D.verify() == 1
        assert t_rounds < 0.060, f"round-by-round proof at 2^20 gates took {t_rounds*1e3:.1f} ms (resident rounds not taken?)"
        print(f"\n2^20 gates, seven wrapped execute_*_round calls + process_queue: {t_rounds*1e3:.1f} ms")
        D.free()
        D.wrap_trim()
    print(f"\n2^20-gate TurboPLONK proof: reference CPU {t_cpu*1e3:.0f} ms ({A.threads} threads), shim-linked (MSM / FFT wrapped only) {t_shim*1e3:.0f} ms, "
          f"resident C++ prover {t_gpu*1e3:.1f} ms first / {t_warm*1e3:.1f} ms warm (key registration {t_key*1e3:.0f} ms, once per circuit); "
          f"construct_proof() wrapped, zero source edits: {t_wrap_first*1e3:.0f} ms first (key upload included) / {t_wrap*1e3:.1f} ms warm")


def test_prover_handle_error_paths(pkg, bbg):
    """bbg_prover_*: call order and argument checks (no silent garbage)."""
    lib = bbg.lib
    srs = bbg.srs_synth_hashed(5, 1 << 6)
    gens = pkg.synthetic_scalars(3, 4)
    h = ctypes.c_void_p()
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, 6, 5, gens.ctypes.data, ctypes.byref(h)) != 0        # width
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, 7, 4, gens.ctypes.data, ctypes.byref(h)) != 0        # SRS too short
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, 6, 3, gens.ctypes.data, ctypes.byref(h)) != 0        # StandardPLONK needs n + 1 points
    assert lib.bbg_prover_create_flavour(bbg.ctx, srs.handle, 6, 3, gens.ctypes.data, ctypes.byref(h)) != 0 # unknown flavour
    assert lib.bbg_prover_create_flavour(bbg.ctx, srs.handle, 6, 2, gens.ctypes.data, ctypes.byref(h)) != 0 # MiMC (width 3) needs n + 1 points
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, 6, 4, gens.ctypes.data, ctypes.byref(h)) == 0
    a = pkg.synthetic_scalars(9, 64)
    out = np.zeros((4, 12), dtype=np.uint64)
    wires = (ctypes.c_void_p * 4)(*[a.ctypes.data] * 4)
    assert lib.bbg_prover_round1(h, wires, out.ctypes.data) != 0                                            # key not finalised
    assert b"finalize" in lib.bbg_last_error()
    assert lib.bbg_prover_finalize_key(h) != 0                                                              # sigmas missing
    assert lib.bbg_prover_set_key_poly(h, 0, 0, a.ctypes.data) != 0                                         # a wire is not a key polynomial
    assert lib.bbg_prover_set_key_poly(h, 9, 1, a.ctypes.data) != 0                                         # selectors have no Lagrange form here
    for pid in range(5, 20):
        assert lib.bbg_prover_set_key_poly(h, pid, 0, a.ctypes.data) == 0
    assert lib.bbg_prover_finalize_key(h) == 0
    ch = pkg.synthetic_scalars(1, 4)
    assert lib.bbg_prover_round3(h, ch[0].ctypes.data, ch[1].ctypes.data, ch.ctypes.data, out.ctypes.data) != 0   # before round 1
    assert lib.bbg_prover_round1(h, wires, out.ctypes.data) == 0
    assert lib.bbg_prover_round4(h, ch[0].ctypes.data, ch[1].ctypes.data, out.ctypes.data) != 0             # before round 3
    ids = (ctypes.c_int * 2)(0, 99)
    ev = np.zeros((2, 4), dtype=np.uint64)
    assert lib.bbg_prover_evaluate(h, 2, ids, None, ch[0].ctypes.data, ev.ctypes.data) != 0                 # unknown polynomial id
    ids = (ctypes.c_int * 2)(0, 1)
    assert lib.bbg_prover_evaluate(h, 2, ids, None, ch[0].ctypes.data, ev.ctypes.data) != 0                 # valid ids, but before round 4
    assert b"round 4" in lib.bbg_last_error()
    lib.bbg_prover_destroy(h)
    srs.free()


def _prover_rounds_1_to_4(pkg, bbg, lib, h, n, seed=0):
    """Rounds 1, 3, 4 on synthetic wires / challenges: every commitment of the quotient path (W_i, Z, T_i) as canonical affine points."""
    wires = [pkg.synthetic_scalars(600 + seed + k, n) for k in range(4)]
    wp = (ctypes.c_void_p * 4)(*[w.ctypes.data for w in wires])
    ch = pkg.synthetic_scalars(700 + seed, 8)
    com = np.zeros((9, 12), dtype=np.uint64)
    assert lib.bbg_prover_round1(h, wp, com.ctypes.data) == 0, lib.bbg_last_error()
    assert lib.bbg_prover_round3(h, ch[0].ctypes.data, ch[1].ctypes.data, ch[2:5].ctypes.data, com[4:].ctypes.data) == 0, lib.bbg_last_error()
    assert lib.bbg_prover_round4(h, ch[5].ctypes.data, ch[6].ctypes.data, com[5:].ctypes.data) == 0, lib.bbg_last_error()
    return bbg.g1_normalize(com)


def test_prover_key_polynomial_replaced_on_live_handle(pkg, bbg):
    """re-registering a key polynomial in coefficient form on a finalised handle must replace EVERY derived form of it (the
    4n coset values rounds 4 reads, sigma's Lagrange form the grand product reads), not only the coefficients rounds 5 / 6 read: the
    commitments after `set_key_poly + finalize` equal those of a fresh handle built with the new key, and differ from the old key's."""
    lib = bbg.lib
    lg, n = 10, 1 << 10
    srs = bbg.srs_synth_hashed(5, n)
    gens = np.stack([bbg.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0] for k in (5, 5, 6, 7)])

    def make(overrides):
        h = ctypes.c_void_p()
        assert lib.bbg_prover_create(bbg.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)) == 0
        for pid in range(5, 20):
            a = overrides.get(pid, pkg.synthetic_scalars(500 + pid, n))
            assert lib.bbg_prover_set_key_poly(h, pid, 0, a.ctypes.data) == 0
        assert lib.bbg_prover_finalize_key(h) == 0
        return h
    new_sigma, new_qm = pkg.synthetic_scalars(9001, n), pkg.synthetic_scalars(9002, n)
    live = make({})
    old = _prover_rounds_1_to_4(pkg, bbg, lib, live, n)
    assert lib.bbg_prover_set_key_poly(live, 6, 0, new_sigma.ctypes.data) == 0    # sigma_2: Lagrange form (round 3) + coset form (round 4)
    assert lib.bbg_prover_set_key_poly(live, 14, 0, new_qm.ctypes.data) == 0      # a selector: coset form (round 4)
    wires = (ctypes.c_void_p * 4)(*[new_qm.ctypes.data] * 4)
    scratch = np.zeros((4, 12), dtype=np.uint64)
    assert lib.bbg_prover_round1(live, wires, scratch.ctypes.data) != 0 and b"finalize" in lib.bbg_last_error()  # a changed key must be finalised again
    assert lib.bbg_prover_finalize_key(live) == 0
    got = _prover_rounds_1_to_4(pkg, bbg, lib, live, n)
    fresh = make({6: new_sigma, 14: new_qm})
    want = _prover_rounds_1_to_4(pkg, bbg, lib, fresh, n)
    assert np.array_equal(got, want), "stale derived key forms after re-registration"
    assert np.array_equal(got[:4], old[:4]) and not np.array_equal(got[4], old[4]) and not np.array_equal(got[5:], old[5:])
    lib.bbg_prover_destroy(live)
    lib.bbg_prover_destroy(fresh)
    srs.free()


def test_quotient_fused_widgets_equal_separate(pkg, bbg):
    """Round 4's fused pass (arithmetic + range + logic widgets in one kernel, option quotient_fuse = 1, default) against one kernel per
    widget: the same quotient commitments T_1..T_4 (the byte-identical-proof tests then pin the fused path against the reference prover)."""
    lib = bbg.lib
    lg, n = 11, 1 << 11
    srs = bbg.srs_synth_hashed(5, n)
    gens = np.stack([bbg.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0] for k in (5, 5, 6, 7)])
    h = ctypes.c_void_p()
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)) == 0
    for pid in range(5, 20):
        assert lib.bbg_prover_set_key_poly(h, pid, 0, pkg.synthetic_scalars(500 + pid, n).ctypes.data) == 0
    assert lib.bbg_prover_finalize_key(h) == 0
    try:
        bbg.set_option("quotient_fuse", 0)
        separate = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
        bbg.set_option("quotient_fuse", 1)
        fused = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
        bbg.set_option("quotient_limbs29", 0)  # the 32-bit-limb kernels: fused, then one per widget
        fused32 = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
        bbg.set_option("quotient_fuse", 0)
        separate32 = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
    finally:
        bbg.set_option("quotient_fuse", 1)
        bbg.set_option("quotient_limbs29", 1)
    assert np.array_equal(separate, fused)
    assert np.array_equal(fused32, fused) and np.array_equal(separate32, fused)
    lib.bbg_prover_destroy(h)
    srs.free()


def test_prover_round4_divisor_table_accounting(pkg):
    """The per-point Z*_H divisor table of round 4 (option prover_fused_divide, poly.hip poly_dpv_table): on a context of its own, the
    quotient commitments T_i are the same with the division inside the coset iFFT's load and in a pass of its own; the table -- 32 bytes
    per point of the 4n domain -- is counted under ntt_tables once, reused by the next proof, and released by bbg_memory_trim."""
    ctx = pkg.Bbg(0)
    lib = ctx.lib
    lg, n = 12, 1 << 12
    srs = ctx.srs_synth_hashed(5, n)
    gens = np.stack([ctx.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0] for k in (5, 5, 6, 7)])
    h = ctypes.c_void_p()
    assert lib.bbg_prover_create(ctx.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)) == 0
    for pid in range(5, 20):
        assert lib.bbg_prover_set_key_poly(h, pid, 0, pkg.synthetic_scalars(500 + pid, n).ctypes.data) == 0
    assert lib.bbg_prover_finalize_key(h) == 0
    ctx.set_option("prover_fused_divide", 0)
    separate = _prover_rounds_1_to_4(pkg, ctx, lib, h, n)
    before = ctx.memory_report()["ntt_tables"]
    ctx.set_option("prover_fused_divide", 1)
    fused = _prover_rounds_1_to_4(pkg, ctx, lib, h, n)
    assert np.array_equal(fused, separate)
    after = ctx.memory_report()["ntt_tables"]
    assert after - before == 32 * 4 * n
    assert np.array_equal(_prover_rounds_1_to_4(pkg, ctx, lib, h, n, seed=3), _prover_rounds_1_to_4(pkg, ctx, lib, h, n, seed=3))
    assert ctx.memory_report()["ntt_tables"] == after
    lib.bbg_prover_destroy(h)
    ctx.memory_trim(tables=True)
    assert ctx.memory_report()["ntt_tables"] == 0
    srs.free()
    ctx.close()


def test_prover_keeps_its_srs_alive(pkg, bbg):
    """a bbg_prover shares ownership of its SRS (bbg_srs_retain): the creator freeing its handle -- what the shim's table cache
    does when a larger table is registered at the same address -- must not pull the window tables from under the live prover."""
    lib = bbg.lib
    lg, n = 10, 1 << 10
    gens = np.stack([bbg.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0] for k in (5, 5, 6, 7)])
    srs = bbg.srs_synth_hashed(5, n)
    h = ctypes.c_void_p()
    assert lib.bbg_prover_create(bbg.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)) == 0
    for pid in range(5, 20):
        assert lib.bbg_prover_set_key_poly(h, pid, 0, pkg.synthetic_scalars(500 + pid, n).ctypes.data) == 0
    assert lib.bbg_prover_finalize_key(h) == 0
    before = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
    srs.free()                                              # the creator lets go; the prover still owns it
    other = bbg.srs_synth_hashed(77, 4 * n)                 # allocations in between: a freed table would be reused
    after = _prover_rounds_1_to_4(pkg, bbg, lib, h, n)
    assert np.array_equal(before, after)
    lib.bbg_prover_destroy(h)                               # drops the last owner
    other.free()


# ---------------------------------------------------------------------------------------------- multi-GPU inside the library
class _Multi:
    def __init__(self, pkg, devices):
        self.lib = pkg.load_library()
        self.pkg = pkg
        arr = (ctypes.c_int * len(devices))(*devices)
        self.h = ctypes.c_void_p()
        rc = self.lib.bbg_multi_create(arr, len(devices), ctypes.byref(self.h))
        if rc != 0:
            raise pkg.BbgError(self.lib.bbg_last_error().decode())

    def ck(self, rc):
        if rc != 0:
            raise self.pkg.BbgError(self.lib.bbg_last_error().decode())

    def close(self):
        if self.h:
            self.lib.bbg_multi_destroy(self.h)
            self.h = None


def _spread(pkg, G):
    """Devices of a G-context group: context g on device g mod bbg_device_count().  On this pool's one-GPU boxes that is [0] * G (every
    context on device 0, the form these tests have always run in); on a node with 2 / 4 / 8 GPUs the SAME tests put one context per GPU, so
    the peer copies cross devices (hipMemcpyPeerAsync over xGMI, cross-device event waits) with no other change."""
    count = pkg.load_library().bbg_device_count()
    assert count >= 1
    return [g % count for g in range(G)]


def _distinct(pkg, limit=8):
    """The largest power-of-two group of DISTINCT devices (RCCL needs one device per rank): [0] on a one-GPU box, [0 .. 7] on a full node."""
    count = pkg.load_library().bbg_device_count()
    G = 1
    while G * 2 <= min(limit, count):
        G *= 2
    return list(range(G))


def _shards_to_devices(a, devs):
    """Residue class g of `a` on the device of context g (int64 view of the 4 x u64 limbs)."""
    import torch
    G = len(devs)
    out = [torch.from_numpy(np.ascontiguousarray(a[g::G]).view(np.int64).reshape(-1)).to(f"cuda:{devs[g]}") for g in range(G)]
    for d in set(devs):
        torch.cuda.synchronize(d)
    return out


def _gather_shards(shards, n, G):
    """Natural-order result from the resident form: shard r holds G runs, run t = A[t*m + r*len ...) (bbg_multi_ntt_device)."""
    m, length = n // G, n // G // G
    got = np.zeros((n, 4), dtype=np.uint64)
    for r in range(G):
        o = shards[r].cpu().numpy().view(np.uint64).reshape(G, length, 4) if G > 1 else shards[r].cpu().numpy().view(np.uint64).reshape(1, n, 4)
        for t in range(G):
            got[t * m + r * length: t * m + (r + 1) * length] = o[t]
    return got


@pytest.mark.parametrize("G", [1, 2, 3, 4, 8])
def test_multi_msm_point_range_shards(pkg, oracle, bbg, golden, G):
    """bbg_multi_msm with G contexts (context g on device g mod the device count: all on device 0 on a one-GPU box, one per GPU on a
    node): point-range shards + the g1 sum of the 96-byte partials reproduce the single-context MSM, the oracle, the reference's recorded
    results, (from, range) sub-ranges that straddle shard boundaries, and the empty MSM."""
    M = _Multi(pkg, _spread(pkg, G))
    try:
        n = 1 << 16
        M.ck(M.lib.bbg_multi_srs_synth_hashed(M.h, 0xBB254, n))
        assert M.lib.bbg_multi_srs_num_points(M.h) == n and M.lib.bbg_multi_count(M.h) == G
        out = np.zeros(12, dtype=np.uint64)

        def run(sc, start=0):
            s = np.ascontiguousarray(sc, dtype=np.uint64)
            M.ck(M.lib.bbg_multi_msm(M.h, s.ctypes.data, start, s.shape[0], out.ctypes.data))
            return out.copy()
        for rec in golden["msm"]:
            if rec["srs"] != "hashed" or rec["from"] + rec["n"] > n or rec.get("scalar_kind") == "mixed":
                continue
            sc = pkg.synthetic_scalars(rec["scalar_seed"], rec["n"])
            assert np.array_equal(oracle.jac_to_affine(run(sc, rec["from"])), unhex(rec["result"], 8)[0]), (G, rec["n"], rec["from"])
        sc = pkg.synthetic_scalars(99, 30000)
        srs = bbg.srs_synth_hashed(0xBB254, n)
        for start in (0, 1, n // G - 7 if G > 1 else 5, n - 30000):
            assert np.array_equal(oracle.jac_to_affine(run(sc, start)), oracle.jac_to_affine(bbg.msm(srs, sc, start=start))), (G, start)
        srs.free()
        assert int(run(sc[:0])[3]) >> 63 == 1
        with pytest.raises(pkg.BbgError):
            run(sc, n - 100)
        # a caller-supplied table (the reference's interleaved endomorphism layout), ragged against G
        pts = oracle.srs_hashed(77, 1001)
        table = oracle.point_table(pts)  # keep the array alive across the call
        M.ck(M.lib.bbg_multi_srs_register(M.h, table.ctypes.data, 1001, 128))
        sc = pkg.synthetic_scalars(5, 1001)
        assert np.array_equal(oracle.jac_to_affine(run(sc)), oracle.pippenger(sc, pts)), G
    finally:
        M.close()


@pytest.mark.parametrize("G,lg", [(1, 10), (2, 11), (4, 12), (8, 13), (8, 16), (4, 4)])
def test_multi_ntt_all_to_all(pkg, oracle, bbg, G, lg):
    """bbg_multi_ntt (host form) and bbg_multi_ntt_device (resident residue-class shards, peer-copy all-to-all, size-G DFT) for
    fft / ifft / coset_fft / coset_ifft against the oracle's whole transform."""
    devs = _spread(pkg, G)
    M = _Multi(pkg, devs)
    try:
        n = 1 << lg
        a = pkg.synthetic_scalars(3100 + lg + G, n)
        for op in (FFT, IFFT, COSET_FFT, COSET_IFFT):
            want = oracle.ntt(a, op)
            buf = a.copy()
            M.ck(M.lib.bbg_multi_ntt(M.h, buf.ctypes.data, lg, op))
            assert np.array_equal(oracle.canon(0, buf), want), (G, lg, op, "host form")
            # resident form: shard g = residue class g on context g's device; result in G runs per shard
            shards = _shards_to_devices(a, devs)
            ptrs = (ctypes.c_void_p * G)(*[t.data_ptr() for t in shards])
            M.ck(M.lib.bbg_multi_ntt_device(M.h, ptrs, lg, op))
            M.ck(M.lib.bbg_multi_sync(M.h))
            assert np.array_equal(oracle.canon(0, _gather_shards(shards, n, G)), want), (G, lg, op, "resident form")
        with pytest.raises(pkg.BbgError):
            M.ck(M.lib.bbg_multi_ntt(M.h, a.ctypes.data, lg, 4))  # only the four whole-domain transforms
    finally:
        M.close()


def _msm24_golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "msm24.json")) as f:
        return json.load(f)


def _multi_msm_2_24(pkg, oracle, M, G24, straddle):
    """The 2^24-term MSM of tests/golden/msm24.json through group M against the REFERENCE's own sharded composition (sixteen
    pippenger_unsafe + g1 sum, pippenger.cpp:27-31, c_bind.cpp:31-46); `straddle`: also reference shards 3..4 as ONE (from, range) call."""
    n = 1 << G24["log2n"]
    M.ck(M.lib.bbg_multi_srs_synth_hashed(M.h, G24["srs_seed"], n))
    assert M.lib.bbg_multi_srs_num_points(M.h) == n
    sc = pkg.synthetic_scalars(G24["scalar_seed"], n)
    out = np.zeros(12, dtype=np.uint64)
    M.ck(M.lib.bbg_multi_msm(M.h, sc.ctypes.data, 0, n, out.ctypes.data))
    assert np.array_equal(oracle.jac_to_affine(out), unhex(G24["result"], 8)[0]), "sharded 2^24 MSM differs from the reference"
    if straddle:
        recs = G24["shards"][3:5]
        lo, cnt = recs[0]["from"], recs[0]["n"] + recs[1]["n"]
        part = np.ascontiguousarray(sc[lo: lo + cnt])
        M.ck(M.lib.bbg_multi_msm(M.h, part.ctypes.data, lo, cnt, out.ctypes.data))
        want = oracle.g1_add(unhex(recs[0]["result"], 8)[0], unhex(recs[1]["result"], 8)[0])
        assert np.array_equal(oracle.jac_to_affine(out), want)


def _multi_ntt_2_24(pkg, oracle, M, devs, ops=None):
    """bbg_multi_ntt_device at 2^24 through group M (contexts on `devs`) against the REFERENCE digests of the whole transform."""
    lg = 24
    n = 1 << lg
    G = len(devs)
    recs = [r for r in _ntt_large_golden() if r["log2n"] == lg and (ops is None or r["op"] in ops)]
    assert recs
    a = pkg.synthetic_scalars(900 + lg, n)
    for rec in recs:
        shards = _shards_to_devices(a, devs)
        ptrs = (ctypes.c_void_p * G)(*[t.data_ptr() for t in shards])
        M.ck(M.lib.bbg_multi_ntt_device(M.h, ptrs, lg, rec["op"]))
        M.ck(M.lib.bbg_multi_sync(M.h))
        got = oracle.canon(0, _gather_shards(shards, n, G))
        for i, want in rec["spots"].items():
            assert np.array_equal(got[int(i)], unhex(want)[0]), (G, rec["op"], "spot", i)
        assert sha(got) == rec["sha256"], (G, rec["op"])


def test_multi_msm_2_24_eight_shards_vs_reference(pkg, oracle, bbg):
    """BASELINE config 5's MSM through the in-library split: G = 8 contexts (context g on device g mod the device count: one per GPU on an
    8-GPU node, all on device 0 on a one-GPU box), 2^21-point SRS shards, bbg_multi_msm over the 2^24 scalars of tests/golden/msm24.json
    -- equal to the REFERENCE's own sharded composition -- and a (from, range) call that straddles the shards of contexts 1 and 2."""
    M = _Multi(pkg, _spread(pkg, 8))
    try:
        _multi_msm_2_24(pkg, oracle, M, _msm24_golden(), straddle=True)
    finally:
        M.close()


@pytest.mark.parametrize("G", [8, 2])
def test_multi_ntt_2_24_vs_reference(pkg, oracle, bbg, G):
    """BASELINE config 5's transform through the in-library split (residue-class shards, all-to-all, size-G DFT) at 2^24 for fft / ifft /
    coset_fft / coset_ifft against the REFERENCE digests of the whole transform (tests/golden/ntt_large.json).  Context g on device
    g mod the device count (peer copies cross devices wherever the box has more than one)."""
    devs = _spread(pkg, G)
    M = _Multi(pkg, devs)
    try:
        _multi_ntt_2_24(pkg, oracle, M, devs)
    finally:
        M.close()


def test_multi_config5_over_rccl_on_distinct_devices(pkg, oracle, bbg):
    """BASELINE config 5 through the RCCL exchange (ncclCommInitAll over the group, ncclAllGather of the 96-byte partials, grouped
    ncclSend / ncclRecv all-to-all) on the largest power-of-two group of DISTINCT devices the box has -- 8 ranks on a full node, where this
    is the first place ncclCommInitAll(N > 1) and the inter-device all-to-all run; a group of one on this pool's boxes (rank 0 exchanging
    with itself: the same calls) -- and the same group through peer copies.  2^24-term MSM against the reference's result, 2^24 fft and
    coset_fft against the reference's digests."""
    devs = _distinct(pkg)
    M = _Multi(pkg, devs)
    try:
        assert M.lib.bbg_multi_count(M.h) == len(devs)
        G24 = _msm24_golden()
        for exchange in (1, 0):
            M.ck(M.lib.bbg_multi_set_option(M.h, b"exchange", exchange))
            _multi_msm_2_24(pkg, oracle, M, G24, straddle=False)
            _multi_ntt_2_24(pkg, oracle, M, devs, ops=(FFT, COSET_FFT) if exchange == 0 else None)
    finally:
        M.close()


def test_multi_rccl_exchange_backend(pkg, oracle, bbg):
    """bbg_multi_set_option("exchange", 1): the C++ RCCL back end (ncclCommInitAll, ncclAllGather of the 96-byte partials + group sum,
    grouped ncclSend / ncclRecv for the all-to-all) over the largest power-of-two group of DISTINCT devices (_distinct: 8 ranks on a full
    node).  On a one-GPU box the group has ONE context: with RCCL selected even that group goes through the exchange (all-gather of one
    partial; every chunk sent to and received from rank 0 = itself), which exercises every RCCL call the multi-GPU path makes.  Results
    against the oracle, and equal to the peer-copy back end.  Duplicate devices are refused."""
    import torch
    devs = _distinct(pkg)
    Gd = len(devs)
    M = _Multi(pkg, devs)
    try:
        n = 1 << 14
        M.ck(M.lib.bbg_multi_srs_synth_hashed(M.h, 0xBB254, n))
        sc = pkg.synthetic_scalars(4711, n)
        pts = oracle.srs_hashed(0xBB254, n)
        out = np.zeros(12, dtype=np.uint64)
        M.ck(M.lib.bbg_multi_msm(M.h, sc.ctypes.data, 0, n, out.ctypes.data))
        peer = oracle.jac_to_affine(out)
        M.ck(M.lib.bbg_multi_set_option(M.h, b"exchange", 1))
        M.ck(M.lib.bbg_multi_msm(M.h, sc.ctypes.data, 0, n, out.ctypes.data))
        assert np.array_equal(oracle.jac_to_affine(out), peer)
        assert np.array_equal(peer, oracle.pippenger(sc, pts))
        M.ck(M.lib.bbg_multi_msm(M.h, sc[:1000].ctypes.data, 300, 1000, out.ctypes.data))
        assert np.array_equal(oracle.jac_to_affine(out), oracle.pippenger(sc[:1000], pts[300:1300]))
        lg = 12
        a = pkg.synthetic_scalars(3100, 1 << lg)
        for op in (FFT, IFFT, COSET_FFT, COSET_IFFT):
            buf = a.copy()
            M.ck(M.lib.bbg_multi_ntt(M.h, buf.ctypes.data, lg, op))
            assert np.array_equal(oracle.canon(0, buf), oracle.ntt(a, op)), op
            shards = _shards_to_devices(a, devs)
            ptrs = (ctypes.c_void_p * Gd)(*[t.data_ptr() for t in shards])
            M.ck(M.lib.bbg_multi_ntt_device(M.h, ptrs, lg, op))
            M.ck(M.lib.bbg_multi_sync(M.h))
            assert np.array_equal(oracle.canon(0, _gather_shards(shards, 1 << lg, Gd)), oracle.ntt(a, op)), op
        M.ck(M.lib.bbg_multi_set_option(M.h, b"exchange", 0))
        M.ck(M.lib.bbg_multi_msm(M.h, sc.ctypes.data, 0, n, out.ctypes.data))
        assert np.array_equal(oracle.jac_to_affine(out), peer)
        with pytest.raises(pkg.BbgError):
            M.ck(M.lib.bbg_multi_set_option(M.h, b"exchange", 2))
    finally:
        M.close()
    M2 = _Multi(pkg, [0, 0])
    try:
        with pytest.raises(pkg.BbgError, match="DISTINCT"):
            M2.ck(M2.lib.bbg_multi_set_option(M2.h, b"exchange", 1))
    finally:
        M2.close()


def test_multi_rejects_bad_groups(pkg, bbg):
    M = _Multi(pkg, [0, 0, 0])
    a = pkg.synthetic_scalars(1, 64)
    with pytest.raises(pkg.BbgError):
        M.ck(M.lib.bbg_multi_ntt(M.h, a.ctypes.data, 6, 0))  # 3 contexts: not a power of two
    out = np.zeros(12, dtype=np.uint64)
    with pytest.raises(pkg.BbgError):
        M.ck(M.lib.bbg_multi_msm(M.h, a.ctypes.data, 0, 64, out.ctypes.data))  # no SRS registered
    M.close()
    with pytest.raises(pkg.BbgError):
        _Multi(pkg, [99])


def test_reference_c_binding_names(pkg, oracle, bbg):
    """libbbg_cbind.so exports the reference's OWN extern "C" names with its signatures (scalar_multiplication/c_bind.hpp:9-19,
    prover/c_bind.cpp:99-120): a host speaking the reference's C / WASM offload protocol binds it unchanged.  Driven here the way such
    a host does -- bbmalloc'ed buffers, transcript-encoded points into new_pippenger, work-item style calls -- against the oracle."""
    so = os.path.join(os.path.dirname(pkg.LIB_PATH), "libbbg_cbind.so")
    L = ctypes.CDLL(so)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.bbmalloc.restype = vp; L.bbmalloc.argtypes = [sz]
    L.bbfree.argtypes = [vp]
    L.new_pippenger.restype = vp; L.new_pippenger.argtypes = [vp, sz]
    L.delete_pippenger.argtypes = [vp]
    L.pippenger_unsafe.argtypes = [vp, vp, sz, sz, vp]
    L.g1_sum.argtypes = [vp, sz, vp]
    L.new_evaluation_domain.restype = vp; L.new_evaluation_domain.argtypes = [sz]
    L.delete_evaluation_domain.argtypes = [vp]
    L.coset_fft_with_generator_shift.argtypes = [vp, vp, vp]
    L.ifft.argtypes = [vp, vp]
    n = 3000
    pts = oracle.srs_hashed(31, n)
    pts[0] = oracle.g1_generator()
    # transcript encoding of points 1 .. n-1: standard form, every limb big-endian (srs/io.cpp:47-67)
    raw = oracle.from_mont(1, pts[1:].reshape(-1, 4)).astype(">u8").tobytes()
    buf = L.bbmalloc(len(raw))
    assert buf % 64 == 0
    ctypes.memmove(buf, raw, len(raw))
    pip = L.new_pippenger(buf, n)
    L.bbfree(buf)
    sc = pkg.synthetic_scalars(77, n)
    res = np.zeros(12, dtype=np.uint64)
    L.pippenger_unsafe(pip, sc.ctypes.data, 0, n, res.ctypes.data)
    assert np.array_equal(oracle.jac_to_affine(res), oracle.pippenger(sc, pts))
    parts = np.zeros((3, 12), dtype=np.uint64)
    for k in range(3):  # (from, range) work items + g1_sum, the composition of c_bind.cpp:31-46
        L.pippenger_unsafe(pip, sc[k * 1000:].ctypes.data, k * 1000, 1000, parts[k].ctypes.data)
    L.g1_sum(parts.ctypes.data, 3, res.ctypes.data)
    assert np.array_equal(oracle.jac_to_affine(res), oracle.pippenger(sc, pts))
    L.delete_pippenger(pip)
    dom = L.new_evaluation_domain(1 << 12)
    c = pkg.synthetic_scalars(78, 1 << 12)
    k = pkg.synthetic_scalars(79, 1)[0]
    a = c.copy()
    L.ifft(a.ctypes.data, dom)
    assert np.array_equal(oracle.canon(0, a), oracle.ntt(c, 1))
    a = c.copy()
    L.coset_fft_with_generator_shift(a.ctypes.data, k.ctypes.data, dom)
    assert np.array_equal(oracle.canon(0, a), oracle.ntt(c, 6, 0, k))
    L.delete_evaluation_domain(dom)
