"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/bbg.h declares, and the
product fails loudly (no fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bbg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build_library()
    lib = ctypes.CDLL(pkg.LIB_PATH) if False else pkg.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} is declared in include/bbg.h but not exported by libbbg.so"
    # the binding's prototype table covers the whole header
    assert sorted(pkg.binding.EXPORTED_SYMBOLS) == declared


def test_fails_loudly_without_gpu(pkg):
    lib = pkg.load_library()
    if lib.bbg_device_count() > 0:
        pytest.skip("a GPU is visible here")
    h = ctypes.c_void_p()
    rc = lib.bbg_init(0, ctypes.byref(h))
    assert rc == -3  # BBG_E_NODEVICE
    assert b"no CPU fallback" in lib.bbg_last_error()
    with pytest.raises(pkg.BbgError):
        pkg.Bbg(0)


def test_product_does_not_touch_the_oracle():
    """The product path must never import / link / call anything under oracle/."""
    pkg_dir = os.path.join(ROOT, "aztec-2.0_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                code = "\n".join(l for l in text.splitlines() if not l.strip().startswith(("#", "//", "*", '"""')))
                assert "import oracle" not in code and "from oracle" not in code and "libbn254_oracle" not in code and "libbbref" not in code, f
    out = os.popen(f"readelf -d {os.path.join(pkg_dir, 'csrc', 'libbbg.so')} 2>/dev/null").read()
    assert "oracle" not in out and "bbref" not in out


def test_input_generator_is_prefix_stable(pkg):
    import numpy as np
    a = pkg.synthetic_scalars(0xBB254, 1000)
    b = pkg.synthetic_scalars(0xBB254, 10)
    assert np.array_equal(a[:10], b)
    assert int(a[:, 3].max()) < (1 << 60)
    # splitmix64 known answer: seed 0 -> first output 0xE220A8397B1DCDAF
    assert int(pkg.splitmix64_limbs(0, 1)[0]) == 0xE220A8397B1DCDAF


def test_reference_c_binding_names_exported(pkg):
    """libbbg_cbind.so: the reference's own extern "C" names for this path (ecc/.../scalar_multiplication/c_bind.hpp:9-19,
    plonk/proof_system/prover/c_bind.cpp:99-120), and nothing else."""
    so = os.path.join(os.path.dirname(pkg.LIB_PATH), "libbbg_cbind.so")
    if not os.path.exists(so):
        pkg.build_library()
    names = {l.split()[-1] for l in os.popen(f"nm -D --defined-only {so}").read().splitlines() if " T " in l}
    assert names == {"bbmalloc", "bbfree", "new_pippenger", "delete_pippenger", "pippenger_unsafe", "g1_sum", "coset_fft_with_generator_shift",
                     "ifft", "new_evaluation_domain", "delete_evaluation_domain"}, names


def test_limb29_constants_match_the_field():
    """The constants the 29-bit-limb arithmetic is built on (csrc/curve29.hip.h, field29.hip.h), recomputed with Python integers: 2^261 mod p
    (the R'-form of one), -p^-1 mod 32 (the digit of the division by 32) and the column bound that lets 18 limb products share a 64-bit
    accumulator."""
    import re
    p = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # BN254 Fq (fq.hpp:11-41)
    src = open(os.path.join(ROOT, "aztec-2.0_amd", "csrc", "curve29.hip.h")).read()
    words = re.search(r"FQ_R261\[8\] = \{([^}]*)\}", src).group(1)
    value = sum(int(w.strip().rstrip("u"), 16) << (32 * i) for i, w in enumerate(words.split(",")))
    assert value == pow(2, 261, p)
    assert (-pow(p, -1, 32)) % 32 == (32 - pow(p, 7, 32)) % 32 == 9   # f29_div32_to_fe's digit multiplier
    assert 18 * ((1 << 29) - 1) ** 2 < 1 << 63                          # 9 a*b + 9 m*p products per column never overflow
    assert 9 * ((1 << 31) + (1 << 29)) * ((1 << 29) + 8) + 2 * 9 * ((1 << 29) + 8) ** 2 < 1 << 64  # f29_mul_sub2's three chains
    for m_mult, e in ((34, 30), (24, 30), (12, 31), (64, 30)):          # the spread constants of curve29.hip.h: top limb stays positive
        assert ((m_mult * p) >> 232) > (1 << (e - 29))
    assert 128 * p < 1 << 261                                            # every lazily reduced value of the mixed addition fits nine limbs


def test_generated_mad_chains_are_in_sync():
    """csrc/mad_chains29.hip.h is generated (scripts/gen_mad29.py): the committed header must be what the generator renders."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_mad29", os.path.join(ROOT, "scripts", "gen_mad29.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.render() == open(mod.PATH).read()


def test_bench_refuses_impossible_shapes_before_any_process_group():
    """bench.py validates the shapes the sharded paths cannot take (G^2 | n for the residue-class NTT split, a power-of-two world <= 8,
    WORLD_SIZE == --gpus) BEFORE any rendezvous: every rank leaves with status 2, rank 0 prints a JSON line with an `error` field -- no
    GPU and no process group is touched, so the check runs here."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    cases = [
        (["--gpus", "8", "--config5-log2n", "5"], {"WORLD_SIZE": "8", "RANK": "0"}, "G^2"),
        (["--gpus", "3"], {"WORLD_SIZE": "3", "RANK": "0"}, "power-of-two"),
        (["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0"}, "WORLD_SIZE"),
        (["--gpus", "1", "--log2n", "28"], {"WORLD_SIZE": "1", "RANK": "0"}, "2^27"),
    ]
    for argv, envx, needle in cases:
        r = subprocess.run([sys.executable, bench] + argv, env=dict(os.environ, **envx), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 2, (argv, r.stderr.decode()[-500:])
        line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
        assert line["value"] is None and needle in line["error"], line
    # a rank other than 0 leaves silently with the same status
    r = subprocess.run([sys.executable, bench, "--gpus", "3"], env=dict(os.environ, WORLD_SIZE="3", RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and not r.stdout.decode().strip()


def test_bench_launches_its_own_ranks_and_always_leaves_a_line():
    """`python bench.py --gpus N` with NO launcher around it (WORLD_SIZE unset) starts its N ranks itself (bench.py self_launch; reference
    precedent for the N-way split it then runs: ecc/curves/bn254/scalar_multiplication/c_bind.cpp:31-46, plonk/proof_system/prover/
    work_queue.hpp:166-199).  Whatever happens it prints exactly ONE JSON line, exits non-zero on failure and never hangs: too few devices
    (this box has none), an impossible shape, a rank that dies, a rank that never answers (grace period), the hard limit."""
    import json
    import subprocess
    import sys
    import time
    bench = os.path.join(ROOT, "bench.py")
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}

    def run(argv, **envx):
        t0 = time.time()
        r = subprocess.run([sys.executable, bench] + argv, env=dict(base, **envx), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1, (argv, lines, r.stderr.decode()[-500:])
        return r.returncode, json.loads(lines[0]), time.time() - t0

    import torch
    if not torch.cuda.is_available():
        rc, line, _ = run(["--gpus", "8"])
        assert rc == 3 and line["value"] is None and line["n_gpus"] == 8 and "GPU(s) visible" in line["error"], line
        rc, line, _ = run(["--gpus", "2"], BBG_DIST_ONE_DEVICE="1")
        assert rc == 3 and "0 GPU(s) visible, 1 needed" in line["error"], line
    rc, line, _ = run(["--gpus", "3"])
    assert rc == 2 and "power-of-two" in line["error"], line
    rc, line, _ = run(["--gpus", "8", "--config5-log2n", "5"])
    assert rc == 2 and "G^2" in line["error"], line
    if not torch.cuda.is_available():
        # both ranks start and die (no device to open): the parent says which rank left with which code
        rc, line, _ = run(["--gpus", "2"], BBG_BENCH_SKIP_DEVICE_CHECK="1", BBG_BENCH_GRACE_S="2")
        assert rc != 0 and line["value"] is None and "exited with code" in line["error"] and line["exit_codes"] == [1, 1], line
        # rank 1 never answers, rank 0 dies: after the grace period the parent stops rank 1 (SIGKILL to the group it started) and reports
        rc, line, el = run(["--gpus", "2"], BBG_BENCH_SKIP_DEVICE_CHECK="1", BBG_BENCH_TEST_HANG_RANK="1", BBG_TEST_HOOKS="1", BBG_BENCH_GRACE_S="2")
        assert rc != 0 and "rank 0 exited with code 1" in line["error"] and line["exit_codes"] == [1, -9] and el < 60, (line, el)
    if not torch.cuda.is_available():
        # started by a launcher (WORLD_SIZE set) on a node with fewer devices than ranks: every rank leaves before any rendezvous, rank 0 with a line
        for rank in (0, 1):
            r = subprocess.run([sys.executable, bench, "--gpus", "2", "--no-config5"], env=dict(base, WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", RANK=str(rank),
                               LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
            assert r.returncode == 3 and len(lines) == (1 if rank == 0 else 0), (rank, r.returncode, lines)
            if rank == 0:
                assert "ranks on this node" in json.loads(lines[0])["error"]
    # nobody answers at all: the hard limit
    rc, line, el = run(["--gpus", "2", "--no-config5"], BBG_BENCH_SKIP_DEVICE_CHECK="1", BBG_BENCH_TEST_HANG_RANK="0", BBG_TEST_HOOKS="1", BBG_BENCH_LAUNCH_TIMEOUT="4",
                       BBG_BENCH_GRACE_S="1")
    assert rc != 0 and line["value"] is None and line["error"] and el < 60, (line, el)


def test_bench_promotes_the_strong_scaling_workload_at_n_gt_1():
    """bench.py --gpus N (N > 1): the line's headline is BASELINE config 5 -- ONE 2^24 MSM + ONE 2^24 coset NTT per step sharded over the N GPUs,
    `scaling: "strong"` -- and the per-GPU 2^20 step (near-linear by construction: every rank owns its own points) moves under
    extra.weak_scaling_step; at N = 1 the headline stays BASELINE.json's metric and the same series' first point is `strong_scaling`.
    Checked on bench.promote_config5 with a recorded-shape line (no GPU); the one-device rehearsal under -m gpu checks the real line.
    Reference precedent for the split: ecc/curves/bn254/scalar_multiplication/c_bind.cpp:31-46."""
    import argparse
    import copy
    import sys
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(config5_log2n=24, log2n=20, steps=20, warmup=3)
    base = {"metric": "BN254 G1 MSM Mscalar-mults/s (+ Fr NTT Gfield-ops/s in extra) at n=2^20", "value": 2900.0, "unit": "Mscalar-mults/s", "n_gpus": 4,
            "steps": 20, "warmup": 3, "ms_per_step": 1.45, "scaling": "weak",
            "config": {"workload": "per GPU and step: ...", "log2n": 20, "sharding": "point-range", "exchange": "RCCL (torch.distributed nccl)"},
            "roofline": {"kernel": "k_accumulate29", "frac": 0.012}, "extra": {}}
    c5 = {"n_gpus": 4, "msm_ms": 5.2, "ntt_ms": 0.6, "bit_exact_vs_reference": {"msm": True, "ntt": True},
          "exchange": {"msm": "all_gather 4 x 96 B", "ntt": "all_to_all 96.0 MiB per rank"},
          "timed": {"steps": 20, "warmup": 3, "blocks_ms": [120.0, 118.0, 119.0], "ms_per_step": 5.95, "value_mscalar_per_s": 2819.7,
                    "ntt_gfield_ops_per_s": 101.5, "per_rank": {"points": 1 << 22, "msm_window_bits": 20, "msm_windows": 13,
                                                                "accumulate_avg_launch_ms": 3.9, "phase_ms_per_step": {}}}}

    class FakeDist:
        def get_world_size(self): return 4
        def get_backend(self): return "nccl"
    out = copy.deepcopy(base)
    bench.promote_config5(out, c5, args, 4, FakeDist(), False)
    assert out["scaling"] == "strong" and out["value"] == 2819.7 and out["ms_per_step"] == 5.95
    assert "config 5" in out["metric"] and "BASELINE config 5" in out["config"]["workload"] and out["config"]["log2n"] == 24
    assert "nccl" in out["config"]["exchange"] and out["config"]["exchange"].endswith(": 4")
    assert out["strong_scaling"]["value"] == out["value"] and out["strong_scaling"]["n_gpus"] == 4
    weak = out["extra"]["weak_scaling_step"]
    assert weak["value"] == 2900.0 and weak["scaling"] == "weak" and weak["roofline"]["frac"] == 0.012
    r = out["roofline"]
    assert r["algorithmic_bytes"] == 96.0 * (1 << 22) and abs(r["achieved"] - 96.0 * (1 << 22) / 3.9e-3 / 1e9) < 0.01 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # N = 1: nothing is promoted, the series' first point is recorded
    one = copy.deepcopy(base)
    one["n_gpus"] = 1
    bench.promote_config5(one, dict(c5, n_gpus=1), args, 1, None, False)
    assert one["scaling"] == "weak" and one["value"] == 2900.0 and one["strong_scaling"]["n_gpus"] == 1 and "weak_scaling_step" not in one["extra"]
    # an extra that failed (guarded() returned an error) leaves the line as it was
    bad = copy.deepcopy(base)
    bench.promote_config5(bad, {"error": "timeout after 600 s"}, args, 4, FakeDist(), False)
    assert bad == base


@pytest.mark.parametrize("threads", ["1", "8"])
def test_shim_content_hash_host_logic(tmp_path, threads):
    """shim/bbg_shim_verify.hpp -- the hash behind the shim's full-content verification of cached point tables and proving keys (round 6) -- is
    plain host C++: the digest is a function of the contents alone (not of which thread hashed which piece, nor of the thread count), every
    single-bit change changes it (piece borders included), polynomial order and length are part of it, a point's hash depends on its index.
    Built with g++ and run here; the GPU suite exercises it through the shim (shim_check, test_wrapped_proof_over_a_key_with_one_poked_coefficient).
    What it guards has no hook in the reference: pippenger.cpp:33-36 frees the table, proving_key.cpp:18-27 is a plain struct."""
    import subprocess
    exe = str(tmp_path / "verify_hash_check")
    src = os.path.join(ROOT, "tests", "tools", "verify_hash_check.cpp")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", exe, src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()
    outs = []
    for t in (threads, "3"):
        r = subprocess.run([exe], env=dict(os.environ, BBG_SHIM_VERIFY_THREADS=t), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        assert r.returncode == 0 and "verify_hash_check PASS" in r.stdout.decode(), r.stdout.decode()
        outs.append([ln for ln in r.stdout.decode().splitlines() if ln.startswith("digest ")])
    assert outs[0] and outs[0] == outs[1], "the digest depends on the thread count"


def test_small_msm_final_stage_dataflow_model():
    """csrc/msm_tiny.hip k_tiny_final on INTEGERS (any abelian group will do): 64 quads, two buckets per quad, slices summed, pair sums,
    Hillis-Steele suffix scan, contribution 2 (tail + y) + x, tree -- must equal sum_b b * B_b, the bucket-weighted sum the reference takes with
    running sums (scalar_multiplication.cpp:773-783).  Also the digit recoding of MsmCfg<8>: 31 windows of 8 bits + one of 7, signed digits with
    carry, the narrow window filed under bucket 2 d against a table point of half the weight (msm_cfg.h) -- sum_w sign_w * bucket_w * 2^table_offset(w) = k."""
    import random
    rng = random.Random(6)
    NB, S = 128, 4
    for _ in range(20):
        parts = [[rng.randrange(-10**9, 10**9) for _ in range(S)] for _ in range(NB)]
        B = [sum(p) for p in parts]
        want = sum((b + 1) * B[b] for b in range(NB))
        NQ = NB // 2
        x = [B[2 * t] for t in range(NQ)]
        y = [B[2 * t + 1] for t in range(NQ)]
        v = [x[t] + y[t] for t in range(NQ)]
        d = 1
        while d < NQ:
            v = [v[t] + (v[t + d] if t + d < NQ else 0) for t in range(NQ)]
            d <<= 1
        c = [2 * ((v[t + 1] if t + 1 < NQ else 0) + y[t]) + x[t] for t in range(NQ)]
        assert sum(c) == want
    # MsmCfg<8>
    C = 8
    windows = (254 + C) // C
    nwide = 255 - windows * (C - 1)
    assert (windows, nwide) == (32, 31)
    width = lambda w: C if w < nwide else C - 1
    offset = lambda w: w * (C - 1) + min(w, nwide)
    scale = lambda w: 0 if w < nwide else 1
    assert offset(windows) == 255
    r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    for k in [0, 1, r - 1, (1 << 253) + 12345, (1 << 254) - 1 if (1 << 254) - 1 < r else r - 2] + [rng.randrange(r) for _ in range(200)]:
        carry, total = 0, 0
        for w in range(windows):
            full = 1 << width(w)
            dgt = ((k >> offset(w)) & (full - 1)) + carry
            neg = dgt > full // 2
            mag = (full - dgt) if neg else dgt
            carry = 1 if neg else 0
            bucket = mag << scale(w)
            assert 0 <= bucket <= 128
            total += (-1 if neg else 1) * bucket * (1 << (offset(w) - scale(w)))
        assert carry == 0 and total == k
