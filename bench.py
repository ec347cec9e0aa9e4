#!/usr/bin/env python3
"""Headline benchmark: BN254 G1 MSM + Fr NTT at n = 2^20 on N MI355X (BASELINE.json `metric`, config 3 + config 2).

One STEP = one pass of the prover hot path over one batch of synthetic input, per GPU:
    1 x Pippenger MSM of n = 2^20 scalars over the device-resident SRS  (+ for N > 1: RCCL all-gather of the N 96-byte
    partials and the group sum -- the point-range sharding of one N*2^20-point MSM), then
    1 x forward NTT of n = 2^20 coefficients (in place, device resident).
Inputs are resident in HBM before the timed region (SRS registered once, like the Pippenger constructor; twiddles
built once, like compute_lookup_table).  `value` = scalars processed by all ranks per second over the WHOLE step
(MSM + NTT), so it under-states the MSM-only rate; the separately event-timed MSM / NTT rates are in `extra`.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...          (no launcher: bench.py starts its N ranks itself, one per GPU -- self_launch() below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    BBG_DIST_ONE_DEVICE=1 python bench.py --gpus N ...   (REHEARSAL: all N ranks on device 0, exchanges staged through the host over gloo --
                                                          the whole N-rank code path on one GPU; its timings say nothing about xGMI)

Rank 0 prints ONE JSON line.  At N = 1, rank 0 also runs the CPU baseline on the host cores (the real reference
binary oracle/_ref/libbbref.so when it runs on this CPU, else the oracle port) on a bounded sample, and checks the GPU
results bit-exactly against it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0xBB254
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MAD_PEAK_TOPS = 33.8           # v_mad_u64_u32 alone: 4.66 clocks per wave at 2.4 GHz on 1024 SIMDs (bench_micro/issue_rates.hip, profiles/r03_issue_rates.txt)


def _latest_pmc_profile():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_v*.txt")))
    return files[-1] if files else os.path.join(ROOT, "profiles", "r01_pmc_v2.txt")


PMC_PROFILE = _latest_pmc_profile()


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    try:
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()
    except OSError:
        return None


def profile_stamp():
    """The committed profile's build stamp (first line: `# build: libbbg.so sha256 <hex> ...`, scripts/refresh_profiles.sh) against the library
    this run loads: the numbers bench.py reads from the profile (traffic, SQ_INSTS_VALU -> valu_issue / step_issue_floor) describe the build
    the profile was taken on."""
    import re
    lib = os.path.join(ROOT, "aztec-2.0_amd", "csrc", "libbbg.so")
    have = _sha256(lib)
    want = None
    try:
        m = re.search(r"libbbg\.so sha256 ([0-9a-f]{64})", open(PMC_PROFILE).readline())
        want = m.group(1) if m else None
    except OSError:
        pass
    return {"profile": "profiles/" + os.path.basename(PMC_PROFILE), "profile_libbbg_sha256": want, "loaded_libbbg_sha256": have,
            "profile_matches_build": bool(want and have and want == have)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary (profiles/r01_pmc_v2.txt: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same bench command; KiB per dispatch).  MI355X_MICROARCH.md's gfx950
    correction (FETCH_SIZE x2) applies to wide coalesced streams; k_accumulate's traffic is 64-byte gathers, for which the
    counter is uncalibrated, so the raw sum is reported.  None when the profile is absent."""
    path = PMC_PROFILE
    try:
        tot = 0.0
        for line in open(path):
            f = line.split()
            if len(f) >= 4 and any(kernel in tok for tok in f[:-3]) and f[-3] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot += float(f[-1]) * 1024.0
        return round(tot) if tot else None
    except OSError:
        return None


def pmc_counter(kernel, counter):
    """Sum over the kernel's variants of one counter's average per dispatch (KiB for FETCH_SIZE / WRITE_SIZE), or None."""
    try:
        tot, seen = 0.0, False
        for line in open(PMC_PROFILE):
            f = line.split()
            if len(f) >= 4 and any(kernel in tok for tok in f[:-3]) and f[-3] == counter:
                tot += float(f[-1])
                seen = True
        return tot if seen else None
    except OSError:
        return None


VALU_PEAK_GWAVE = 1024 * 2.4 / 4  # 256 CUs x 4 SIMDs, one wave-instruction per 4 clocks at 2.4 GHz = 614 G wave-instructions/s


def pmc_clock_ghz(kernel):
    """Effective clock of `kernel` under the profiler, GHz of ONE XCD: the committed profile's "effective clock per kernel" table
    (GRBM_GUI_ACTIVE / duration of the same dispatches, summed over the 8 XCDs -> / 8); dispatch-weighted mean over the kernel's variants."""
    try:
        num = den = 0.0
        for line in open(PMC_PROFILE):
            f = line.split()
            if len(f) >= 5 and any(kernel in tok for tok in f[:-4]) and "." in f[-1] and "." in f[-2] and f[-3].isdigit() and f[-4].isdigit():
                num += float(f[-1]) * int(f[-4])
                den += int(f[-4])
        return round(num / den / 8.0, 3) if den else None
    except (OSError, ValueError):
        return None


def valu_issue(acc_ms, ntt_ms):
    """What actually bounds these kernels: VALU instruction issue.  SQ_INSTS_VALU (wave-instructions per launch, committed PMC pass)
    over the launch duration measured in THIS run, against the chip's issue peak."""
    acc, ntt = pmc_counter("k_accumulate", "SQ_INSTS_VALU"), pmc_counter("k_ntt_pass", "SQ_INSTS_VALU")  # whichever pass kernels the profiled build ran (k_ntt_pass8 / 8s / 29)
    out = {"unit": "G wave-instructions/s", "peak": round(VALU_PEAK_GWAVE, 1), "source": "profiles/" + os.path.basename(PMC_PROFILE) + " SQ_INSTS_VALU",
           "profile_matches_build": profile_stamp()["profile_matches_build"]}
    # `peak` prices a SIMD at 2.4 GHz; the kernels do not run there (the mad-bound accumulation is power-limited to ~2.03 GHz): the same
    # fraction against the clock the profile measured for that kernel (GRBM_GUI_ACTIVE / duration) is given beside it
    def entry(key, insts, ms, ghz):
        ach = insts / (ms * 1e-3) / 1e9
        e = {key: insts, "achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GWAVE, 3), "measured_clock_ghz": ghz}
        if ghz:
            e["peak_at_measured_clock"] = round(1024 * ghz / 4, 1)
            e["frac_at_measured_clock"] = round(ach / (1024 * ghz / 4), 3)
        return e
    if acc:
        out["msm_accumulate"] = entry("insts_per_launch", acc, acc_ms, pmc_clock_ghz("k_accumulate"))
    if ntt:
        out["ntt_whole"] = entry("insts_per_ntt", ntt, ntt_ms, pmc_clock_ghz("k_ntt_pass"))
    return out


def pmc_traffic_ntt():
    """HBM bytes of ONE whole NTT (both pass kernels, one launch each) from the committed PMC passes.  The passes stream wide
    coalesced rows, the case for which MI355X_MICROARCH.md prescribes FETCH_SIZE x 2 on gfx950; WRITE_SIZE is taken as reported."""
    fetch, write = pmc_counter("k_ntt_pass", "FETCH_SIZE"), pmc_counter("k_ntt_pass", "WRITE_SIZE")
    if fetch is None or write is None:
        return None
    return round((2.0 * fetch + write) * 1024.0)


# The contract is ONE JSON line on stdout.  Native libraries (RCCL prints "Librccl path : ..." when a process group comes up)
# write to file descriptor 1 behind Python's back, so fd 1 is pointed at stderr for the whole run and the JSON line goes to
# a saved duplicate of the original stdout.
REAL_STDOUT = os.dup(1)
os.dup2(2, 1)


METRIC_FMT = "BN254 G1 MSM Mscalar-mults/s (+ Fr NTT Gfield-ops/s in extra) at n=2^%d"


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def shape_problem(world, args):
    """Why the sharded paths cannot take this world / these sizes (None = fine).  Decided from the arguments alone, so every rank -- and the
    self-launching parent before it starts anybody -- comes to the same answer."""
    if world != args.gpus:
        return "WORLD_SIZE = %d but --gpus %d" % (world, args.gpus)
    if world < 1 or world & (world - 1) or world > 8:
        return "the residue-class NTT split needs a power-of-two world <= 8 (got %d)" % world
    if not args.no_config5 and ((1 << args.config5_log2n) // world) % world:
        return "config 5: G^2 must divide n (G = %d, n = 2^%d)" % (world, args.config5_log2n)
    if args.log2n < 1 or args.log2n > 27:
        return "--log2n must be 1 .. 27 (2^27 points per device is the entry format's cap)"
    return None


def emit_error(args, world, problem, **more):
    """The contract's ONE JSON line, for a run that cannot produce a number: `value` null and an `error` that says why."""
    line = {"metric": METRIC_FMT % args.log2n, "value": None, "unit": "Mscalar-mults/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "error": problem}
    line.update(more)
    os.write(REAL_STDOUT, (json.dumps(line) + "\n").encode())


def visible_devices():
    """HIP devices this process could open (0 without a GPU or without the runtime) -- asked in a child process so that the launching parent
    never initialises the runtime its ranks are about to use."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, timeout=300)
        return int(r.stdout.decode().strip().splitlines()[-1]) if r.returncode == 0 else 0
    except (subprocess.TimeoutExpired, ValueError, IndexError, OSError):
        return 0


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks (this same file, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    the environment -- what torch.distributed.run would have set), hand rank 0's ONE JSON line through, and return the worst exit code.
    Never hangs and never leaves without a line: an impossible shape, too few devices, a rank that dies (the others are given
    BBG_BENCH_GRACE_S to leave by themselves, then stopped) and the hard limit BBG_BENCH_LAUNCH_TIMEOUT all end in a line with `error`.
    Only the exact processes started here are ever signalled."""
    import signal
    import subprocess
    import tempfile
    world = args.gpus
    os.environ["WORLD_SIZE"] = str(world)  # what shape_problem() compares --gpus with
    problem = shape_problem(world, args)
    if problem:
        emit_error(args, world, problem)
        return 2
    one_device = os.environ.get("BBG_DIST_ONE_DEVICE") == "1"
    skip_check = os.environ.get("BBG_BENCH_SKIP_DEVICE_CHECK") == "1"  # tests only: lets the CPU suite see a rank die / hang under the launcher
    have = world if skip_check else visible_devices()
    need = 1 if one_device else world
    if have < need:
        emit_error(args, world, "%d GPU(s) visible, %d needed%s" % (have, need, "" if one_device else
                   " (BBG_DIST_ONE_DEVICE=1 rehearses the N-rank path on one device)"), visible_devices=have)
        return 3
    limit = float(os.environ.get("BBG_BENCH_LAUNCH_TIMEOUT", "1500"))
    grace = float(os.environ.get("BBG_BENCH_GRACE_S", "30"))
    port = free_port()
    line_file = tempfile.TemporaryFile()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdin=subprocess.DEVNULL,
                                      stdout=line_file if r == 0 else subprocess.DEVNULL, stderr=2, start_new_session=True))
    t0 = time.monotonic()
    failed_at, why = None, None
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() is not None and p.returncode != 0]
        if bad and failed_at is None:
            failed_at, why = now, "rank %d exited with code %d" % bad[0]
        if now - t0 > limit:
            why = why or "no answer within BBG_BENCH_LAUNCH_TIMEOUT = %d s" % limit
            break
        if failed_at is not None and now - failed_at > grace:
            break
        time.sleep(0.1)
    for p in procs:  # whoever is still here after a failure or the limit: these exact process groups, nothing else
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()
    rcs = [p.returncode for p in procs]
    if why is None:
        why = next(("rank %d exited with code %d" % (r, rc) for r, rc in enumerate(rcs) if rc), None)
    line_file.seek(0)
    lines = [ln for ln in line_file.read().decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if lines:
        os.write(REAL_STDOUT, (lines[-1] + "\n").encode())
    else:
        emit_error(args, world, why or "rank 0 left without a line", exit_codes=rcs)
    worst = max((abs(rc) for rc in rcs), default=0)
    return (worst if worst < 256 else 1) if (worst or lines) else 1


def promote_config5(out, c5, args, world, dist, one_device):
    """The strong-scaling series.  Every line gets a top-level "strong_scaling" object -- BASELINE config 5 (ONE 2^24 MSM + ONE 2^24 coset NTT
    per step, total work fixed as N grows) through the contract's own timed region -- so that the N = 1, 2, 4, 8 lines hold ONE comparable
    series.  At N > 1 that workload IS the line's headline (`value`, `ms_per_step`, `scaling: "strong"`, `config.workload`, `roofline`); the
    per-GPU 2^20 step whose N-fold repetition scales by construction (every rank owns its own points; the only exchange is 96 bytes) moves
    to extra.weak_scaling_step.  At N = 1 the headline stays BASELINE.json's metric (n = 2^20, one GPU)."""
    t = c5.get("timed") if isinstance(c5, dict) else None
    if not t:
        return
    lg = args.config5_log2n
    n = 1 << lg
    pr = t["per_rank"]
    out["strong_scaling"] = {"workload": "BASELINE config 5: per step ONE 2^%d-point MSM + ONE 2^%d coset NTT, sharded over the N GPUs" % (lg, lg),
                             "n_gpus": world, "value": t["value_mscalar_per_s"], "unit": "Mscalar-mults/s", "ms_per_step": t["ms_per_step"],
                             "ntt_gfield_ops_per_s": t["ntt_gfield_ops_per_s"], "steps": t["steps"], "warmup": t["warmup"],
                             "bit_exact_vs_reference": c5["bit_exact_vs_reference"]}
    if world == 1:
        return
    out["extra"]["weak_scaling_step"] = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"],
                                         "scaling": "weak", "workload": out["config"]["workload"], "roofline": out["roofline"],
                                         "note": "every rank owns its own 2^%d points: near-linear by construction, not the scaling claim" % args.log2n}
    out["metric"] = ("BN254 G1 MSM Mscalar-mults/s (+ Fr coset-NTT Gfield-ops/s in strong_scaling) -- BASELINE config 5: n=2^%d over %d GPUs, strong scaling"
                     % (lg, world))
    out["value"], out["ms_per_step"], out["scaling"] = t["value_mscalar_per_s"], t["ms_per_step"], "strong"
    out["steps"], out["warmup"] = t["steps"], t["warmup"]
    out["config"]["workload"] = ("BASELINE config 5, per step: ONE 2^%d-point Pippenger MSM sharded by point range over %d GPUs (resident SRS shards; "
                                 "all-gather of the %d 96-B partials + group sum) + ONE 2^%d coset NTT sharded by residue class (one all-to-all of "
                                 "%s + a size-%d DFT across ranks)" % (lg, world, world, lg, c5["exchange"]["ntt"].replace("all_to_all ", ""), world))
    out["config"]["log2n"] = lg
    out["config"]["sharding"] = "MSM: point range; NTT: residue class"
    acc_ms = pr["accumulate_avg_launch_ms"]
    alg = 96.0 * pr["points"]  # 96 B per term; one launch of this rank's accumulation covers its 2^lg / N terms
    ach = alg / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
    out["roofline"] = {"kernel": "k_accumulate29 (MSM bucket accumulation, 9 x 29-bit limbs; %d-bit windows on this rank's %d points)" % (pr["msm_window_bits"], pr["points"]),
                       "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None,
                       "algorithmic_bytes": alg, "avg_launch_ms": round(acc_ms, 4),
                       "note": "per rank and launch; 256-bit modular integer work: the binding resource is v_mad_u64_u32 issue (extra.alu of the N = 1 line)"}
    if dist is not None and not one_device:
        try:
            backend = dist.get_backend()
        except Exception:  # noqa: BLE001
            backend = "?"
        out["config"]["exchange"] = "RCCL (torch.distributed backend '%s'); ranks in the process group as it reports them: %d" % (backend, dist.get_world_size())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=5, help="the timed block of --steps steps is repeated this many times; the MEDIAN block is reported")
    ap.add_argument("--no-config5", action="store_true", help="skip extra.config5 (one 2^24 MSM + one 2^24 coset NTT, strong scaling)")
    ap.add_argument("--no-prover-shaped", action="store_true", help="skip extra.prover_shaped (BASELINE config 4 on the resident prover rounds)")
    ap.add_argument("--config5-log2n", type=int, default=24)
    ap.add_argument("--real-prover-log2", type=int, default=16, help="cpu_baseline.real_prover: gates of the reference TurboPLONK circuit proved on the host "
                    "cores and through the link-time shim (0 = skip; 2^20 takes ~1 min of circuit construction)")
    ap.add_argument("--no-sweeps", action="store_true", help="skip extra.host_path / extra.ntt_sweep / extra.msm_sweep (SURVEY 8d tables)")
    ap.add_argument("--reduce-priority", type=int, default=-1, help="A/B: 1 = low-priority auxiliary stream for the MSM reduce phase (library default), 0 = normal")
    ap.add_argument("--msm-window", type=int, default=0, help="bucket window width: 0 = library default, or a compiled width (A/B runs)")
    ap.add_argument("--limbs29", type=int, default=-1, help="A/B: bucket accumulation on 9 x 29-bit limbs (1, default) or 8 x 32 (0)")
    ap.add_argument("--acc-waves", type=int, default=-1, help="A/B: lane segments per SIMD lane of the bucket accumulation (0 = automatic)")
    ap.add_argument("--reduce-quad", type=int, default=-1, help="A/B: msm_reduce_quad stage mask (library default 14)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))  # no launcher around us: start the N ranks here (each re-enters main() with RANK / WORLD_SIZE set)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_device = os.environ.get("BBG_DIST_ONE_DEVICE") == "1"  # rehearsal: every rank on device 0, gloo + host staging (parallel.HostStagedDist)
    # tests only, and only where the environment asked for test hooks (BBG_TEST_HOOKS=1): a rank that never answers (the launcher's limits)
    if os.environ.get("BBG_TEST_HOOKS") == "1" and os.environ.get("BBG_BENCH_TEST_HANG_RANK") == str(rank) and world > 1:
        time.sleep(3600)
    if one_device:
        local_rank = 0
    # Shapes the sharded paths cannot take are refused HERE, before any process group exists: every rank sees the same arguments and
    # leaves, rank 0 with a JSON line that says why -- never a rank waiting in a collective the others did not enter.
    problem = shape_problem(world, args)
    if problem:
        if rank == 0:
            emit_error(args, world, problem)
        raise SystemExit(2)

    import torch
    # Fewer devices than ranks on this node (a launcher started N ranks on a box with fewer GPUs): every rank sees the same counts and leaves
    # before any rendezvous, rank 0 with a line that says why -- not a rank waiting in init_process_group for one that died in set_device.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    ndev = torch.cuda.device_count()
    if not one_device and world > 1 and ndev < local_world and os.environ.get("BBG_BENCH_SKIP_DEVICE_CHECK") != "1":  # (tests: let a rank die instead)
        if rank == 0:
            emit_error(args, world, "%d GPU(s) visible, %d ranks on this node (BBG_DIST_ONE_DEVICE=1 rehearses the N-rank path on one device)" % (ndev, local_world),
                       visible_devices=ndev)
        raise SystemExit(3)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import importlib
    par = importlib.import_module("aztec_amd.parallel")

    dist = None
    force_dist = os.environ.get("BBG_FORCE_DIST") == "1"  # exercise the RCCL + pipeline code path with a world of 1
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # only without a launcher, i.e. a world of one (BBG_FORCE_DIST): any free port will do
            os.environ["MASTER_PORT"] = str(free_port())
        torch.cuda.set_device(local_rank)
        if one_device:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
            dist = par.HostStagedDist(dist_mod)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist = dist_mod
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    lg = args.log2n
    n = 1 << lg
    bbg = pkg.Bbg(local_rank)
    bbg.set_stream(torch.cuda.current_stream().cuda_stream)
    # bucket reduction of MSM i overlaps sort/accumulate of step i+1; BBG_BENCH_INLINE_REDUCE=1 (profiling only: scripts/refresh_profiles.sh)
    # keeps it on the main stream so that a kernel trace shows every kernel alone on the device
    bbg.set_option("msm_async_reduce", 0 if os.environ.get("BBG_BENCH_INLINE_REDUCE") == "1" else 1)
    if one_device and args.reduce_priority < 0:
        # rehearsal: N processes' queues share one device; a LOW-priority reduce stream is then starved for seconds by the other ranks' work
        bbg.set_option("msm_reduce_priority", 0)
    if args.msm_window:
        bbg.set_option("msm_window", args.msm_window)
    if args.reduce_priority >= 0:
        bbg.set_option("msm_reduce_priority", args.reduce_priority)
    if args.limbs29 >= 0:
        bbg.set_option("msm_limbs29", args.limbs29)
    if args.acc_waves >= 0:
        bbg.set_option("msm_acc_waves", args.acc_waves)
    if args.reduce_quad >= 0:
        bbg.set_option("msm_reduce_quad", args.reduce_quad)

    # ---- setup (untimed): SRS shard resident in HBM, scalars / coefficients resident, twiddles built
    start = rank * n  # weak scaling: every rank owns n points of a world*n-point SRS
    srs = bbg.srs_synth_hashed(SEED + start, n)  # P_{start+i}: the hashed generator is index-based
    scalars = pkg.synthetic_scalars(SEED + 3, n, start)
    coeffs = pkg.synthetic_scalars(SEED + 100 + lg, n)
    d_scalars = torch.from_numpy(scalars.view(np.int64)).to(dev)
    d_coeffs = torch.from_numpy(coeffs.view(np.int64)).to(dev)
    d_coeffs_work = d_coeffs.clone()
    d_result = torch.zeros(12, dtype=torch.int64, device=dev)
    bbg.ntt_prepare(lg)

    # N > 1: one world*n-point MSM per step, sharded by point range; the all-gather + group sum of step i-1 is issued
    # while step i's local MSM is still reducing (parallel.ShardedMsmPipeline); flush() inside the timed region
    # completes the last one, so exactly K global MSMs are finished when the clock stops.
    pipe = None
    if dist is not None:
        # the all-gather + group sums of earlier steps run on a side stream (a second context bound to it), four steps' partials at a time
        side = torch.cuda.Stream()
        bbg_side = pkg.Bbg(local_rank)
        bbg_side.set_stream(side.cuda_stream)
        pipe = par.ShardedMsmPipeline(par.BbgOps(bbg, srs, bbg_side), dist, lambda k: torch.zeros(k, dtype=torch.int64, device=dev), side_stream=side, depth=4)

    def step():
        if pipe is not None:
            pipe.submit(d_scalars, n)
        else:
            bbg.msm_device(srs, d_scalars.data_ptr(), n, d_result.data_ptr())
        bbg.ntt_device(d_coeffs_work.data_ptr(), lg, 0)

    def fence():
        if pipe is not None:
            pipe.flush()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # The timed block: EXACTLY --steps steps between two fences (barrier + synchronize), max over ranks.  It is repeated --blocks
    # times and the MEDIAN block is the one reported (value, ms_per_step and the per-kernel event times all come from it): a single
    # 37 ms block moves by +-4 % from run to run.
    blocks = []
    for _ in range(max(1, args.blocks)):
        if pipe is not None:
            pipe.reset()
        bbg.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        pr = {k: bbg.profile_get(k) for k in ("msm_recode", "msm_sort", "msm_offsets", "msm_accumulate", "msm_reduce", "ntt_pass")}
        bbg.profile_enable(False)
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        blocks.append((el, pr))
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    elapsed, prof = blocks[order[len(order) // 2]]
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * n / (elapsed / args.steps) / 1e6

    # ---- per-kernel numbers from the HIP events recorded inside the timed region
    def avg(name):
        ms, cnt = prof[name]
        return (ms / cnt) if cnt else float("nan")

    msm_ms = sum(prof[k][0] for k in ("msm_recode", "msm_sort", "msm_offsets", "msm_accumulate", "msm_reduce")) / args.steps
    ntt_ms = prof["ntt_pass"][0] / args.steps
    acc_ms = avg("msm_accumulate")
    alg_bytes_msm = 96.0 * n                      # 32-B scalar + 64-B base per term, one launch covers all n terms
    achieved = alg_bytes_msm / (acc_ms * 1e-3) / 1e9
    stamp = profile_stamp()
    roofline = {"kernel": "k_accumulate29 (MSM bucket accumulation, 9 x 29-bit limbs)", "bound": "hbm", "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic("k_accumulate"),
                "profile_matches_build": stamp["profile_matches_build"],
                "traffic_source": "profiles/" + os.path.basename(PMC_PROFILE) + " (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; bytes per launch at n=2^20)",
                "algorithmic_bytes": alg_bytes_msm, "avg_launch_ms": round(acc_ms, 4),
                "note": "256-bit modular integer work: the binding resource is v_mad_u64_u32 issue, see extra.alu"}
    width, msm_windows = bbg.msm_plan(n, srs)  # what the library ran with (bbg_msm_plan), not a copy of its rule
    msm_windows = float(msm_windows)
    pass_ms = avg("ntt_pass")
    ntt_alg = 64.0 * n
    ntt_passes = prof["ntt_pass"][1] / max(1, args.steps)
    nplan = bbg.ntt_plan(lg)  # what the library ran the transform as (bbg_ntt_plan), not a copy of its rule
    extra = {
        "msm_ms": round(msm_ms, 4), "msm_mscalar_per_s_per_gpu": round(n / msm_ms / 1e3, 2),
        "ntt_ms": round(ntt_ms, 4), "ntt_gfield_ops_per_s_per_gpu": round(1.5 * n * lg / ntt_ms / 1e6, 2),
        "msm_phase_ms": {k: round(prof[k][0] / args.steps, 4) for k in prof if k.startswith("msm_")},
        # 64 n algorithmic bytes for the WHOLE transform -- `frac` is that over the transform's time (all its launches);
        # the per-launch figure (64 n charged to each pass) is kept under per_launch_frac
        "roofline_ntt": {"kernel": "%s (%s; %d launches per 2^%d transform, radix %s)" % (
                             nplan["kernel"], {"k_ntt_pass29": "NTT pass on lazily reduced 9 x 29-bit limbs", "k_ntt_pass8": "radix-8 register NTT pass, 32-bit limbs",
                                               "k_ntt_pass8s": "radix-8 register NTT pass, one-plane exchange", "k_ntt_pass": "radix-2 NTT pass in LDS"}.get(nplan["kernel"], ""),
                             nplan["passes"], lg, " x ".join("2^%d" % r for r in nplan["log_radix"])),
                         "bound": "hbm", "achieved": round(ntt_alg / (ntt_ms * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ntt_alg / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes": ntt_alg, "whole_ntt_ms": round(ntt_ms, 4),
                         "traffic": pmc_traffic_ntt(),
                         "traffic_source": "profiles/" + os.path.basename(PMC_PROFILE) + " (2 x FETCH_SIZE + WRITE_SIZE summed over the k_ntt_pass* launches of one transform: the "
                                           "guide's gfx950 correction for wide coalesced streams)",
                         "avg_launch_ms": round(pass_ms, 4), "launches_per_ntt": ntt_passes,
                         "per_launch_frac": round(ntt_alg / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "whole_ntt_frac": round(ntt_alg / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        "timed_blocks_ms": [round(b[0] * 1e3, 3) for b in blocks], "reported_block": "median",
        "profile_stamp": stamp,
        "valu_issue": valu_issue(acc_ms, ntt_ms),
        "alu": {"unit": "T v_mad_u64_u32/s", "peak_measured": MAD_PEAK_TOPS,
                # windows x n mixed additions x 1467 mads (k_accumulate29: 7 products of 162 + 2 squares of 126 + 1 double product of 243 on
                # 29-bit limbs; 14 windows of 19 bits at n = 2^20); NTT: n/2*(lg - passes) + n*(passes-1) Fr products x 136 (32-bit limbs)
                "msm_windows": msm_windows,
                "msm_accumulate": round(msm_windows * n * 1467 / (acc_ms * 1e-3) / 1e12, 2),
                "ntt": round((n / 2 * (lg - ntt_passes) + n * (ntt_passes - 1)) * 136 / (ntt_ms * 1e-3) / 1e12, 2)},
    }

    out = {
        "metric": "BN254 G1 MSM Mscalar-mults/s (+ Fr NTT Gfield-ops/s in extra) at n=2^%d" % lg,
        "value": round(value, 3), "unit": "Mscalar-mults/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 limbs (256-bit Montgomery integers)", "data": "synthetic",
        "config": {"workload": "per GPU and step: 1 Pippenger MSM (n=2^%d scalars, hashed synthetic SRS resident in HBM) + 1 forward "
                               "NTT (n=2^%d); N>1 = one N*2^%d-point MSM sharded by point range, RCCL all-gather of 96-B partials" % (lg, lg, lg),
                   "log2n": lg, "sharding": "point-range" if world > 1 else "none",
                   "exchange": ("REHEARSAL: all %d ranks on ONE device, gloo with host staging (BBG_DIST_ONE_DEVICE=1) -- not a scaling number" % world) if one_device
                               else ("RCCL (torch.distributed nccl)" if dist is not None else "none")},
        "roofline": roofline, "extra": extra,
    }

    if pipe is not None:
        d_result = pipe.last_result()
    # ---- CPU baseline + bit-exact check against it (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, bbg, srs, scalars, coeffs, d_result, d_coeffs, lg, value)
        if args.real_prover_log2 and out["cpu_baseline"].get("kind") == "reference":
            try:  # main thread only: the reference's OpenMP state (see reference_prover_sequence below)
                out["cpu_baseline"]["real_prover"] = real_prover_baseline(args.real_prover_log2)
            except Exception as e:  # noqa: BLE001 -- reported in the line, never raised
                out["cpu_baseline"]["real_prover"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # The extras below must never cost the run its line: each runs under a guard (exception -> {"error": ...}; no answer within
    # the limit, e.g. a rank stuck in a collective -> {"error": "timeout"}, the line is printed and the process leaves without
    # waiting for the stuck call).
    stuck = []

    def guarded(name, fn, limit_s):
        import threading
        box = {}

        def run():
            try:
                torch.cuda.set_device(dev)
                box["value"] = fn()
            except BaseException as e:  # noqa: BLE001 -- reported in the line, never raised
                box["error"] = "%s: %s" % (type(e).__name__, e)

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(limit_s)
        if th.is_alive():
            stuck.append(name)
            return {"error": "timeout after %d s" % limit_s}
        return box["value"] if "value" in box else {"error": box.get("error", "no result")}

    # ---- BASELINE config 4 (TurboPLONK prover sequence) on the resident prover rounds, one GPU
    if rank == 0 and world == 1 and not args.no_prover_shaped and lg >= 10:
        extra["prover_shaped"] = guarded("prover_shaped", lambda: prover_shaped(pkg, bbg, srs, lg), 300)
        if "error" not in extra["prover_shaped"] and "cpu_baseline" in out and out["cpu_baseline"].get("kind") == "reference":
            # the reference binary stays on the MAIN thread: its OpenMP pippenger keeps per-thread state sized by the team it first saw,
            # and a call from another host thread (another OpenMP root) corrupts memory -- measured: SIGSEGV
            try:
                ref_ms = reference_prover_sequence(srs, scalars, coeffs, lg)
            except Exception as e:  # noqa: BLE001
                ref_ms = {"error": "%s: %s" % (type(e).__name__, e)}
            if "error" not in ref_ms:
                out["cpu_baseline"]["prover_shaped_ms"] = ref_ms
                extra["prover_shaped"]["reference_same_sequence_ms"] = ref_ms["total_ms"]
                extra["prover_shaped"]["speedup_vs_reference_sequence"] = round(ref_ms["total_ms"] / extra["prover_shaped"]["proof_ms"], 1)
    extra["step_issue_floor"] = step_issue_floor()
    if extra["step_issue_floor"]:
        extra["step_issue_floor_ms"] = extra["step_issue_floor"]["ms"]
    # ---- the PCIe-inclusive drop-in entry points (host buffers in / out), one GPU
    if rank == 0 and world == 1 and not args.no_sweeps and not stuck:
        extra["host_path"] = guarded("host_path", lambda: host_path(pkg, bbg, srs, lg), 120)
        rp = out.get("cpu_baseline", {}).get("real_prover", {})
        if "error" not in extra["host_path"] and "link_only_ms" in rp:  # the unmodified prover through the link-time shim (cpu_baseline.real_prover)
            extra["host_path"]["shim_linked_proof_ms"] = {"log2_gates": rp["log2_gates"], "msm_fft_wrapped_only": rp["link_only_ms"],
                                                          "construct_proof_wrapped_too": rp["wrapped_zero_edits_ms"], "reference_cpu": rp["cpu_ms"]}
        # "scalar H2D copy and result D2H INCLUDED for the drop-in path, and also reported device-resident".  `value` is the
        # device-resident figure; this object is the other one, in one place: what a caller of the reference signatures gets.
        hp = extra["host_path"]
        if "error" not in hp:
            extra["dropin"] = {
                "what": "the drop-in boundary's own numbers: host buffers in (pageable memory, PCIe) and out, one blocking call each -- what "
                        "pippenger_unsafe() / ifft() / coset_fft() cost a host that links the shim; `value` above is the same MSM with scalars resident in HBM",
                "msm_ms": hp["bbg_msm_ms"], "msm_mscalar_per_s": hp["msm_mscalar_per_s"], "log2n": lg,
                "msm_vs_device_resident_step": round(hp["msm_mscalar_per_s"] / value, 3) if value else None,
                "ifft_ms": hp["bbg_ntt_ifft_ms"], "coset_fft_n_to_4n_ms": hp["bbg_coset_fft_extend_4n_ms"],
                # a whole proof at the two boundaries a host can link (INTEGRATION.md 2a / 2a'), measured in this run at the size cpu_baseline.real_prover ran ...
                "proof_ms_this_run": hp.get("shim_linked_proof_ms"),
                # ... and at BASELINE config 4's size as recorded on this hardware (2^20 gates takes a minute of circuit construction per run)
                "proof_ms_2^20_gates_recorded": {"msm_fft_entry_points_wrapped_only": 1011.4, "construct_proof_wrapped_too": 27.23, "reference_cpu": 4802.6, "host_threads": 64,
                                                  "source": "profiles/r04_real_prover.txt (same hardware, round 4; python bench.py --real-prover-log2 20 re-measures)"},
                "prover_shaped_resident_ms": (extra.get("prover_shaped") or {}).get("proof_ms"),
            }
    # ---- BASELINE config 5: ONE 2^24 MSM + ONE 2^24 coset NTT over the N GPUs (strong scaling: total work fixed as N grows)
    if not args.no_config5 and not stuck:
        srs.free()
        srs = None
        # timed_steps: the contract's K-step region on THIS workload too -- at N > 1 it becomes the line's headline (promote_config5 below),
        # at N = 1 it is the first point of the same series (top-level "strong_scaling")
        c5_side = (side, bbg_side) if dist is not None else None
        c5 = guarded("config5", lambda: config5(pkg, par, bbg, dist, dev, rank, world, args.config5_log2n, timed_steps=args.steps,
                                                warmup=max(1, min(args.warmup, 3)), blocks=min(3, max(1, args.blocks)), side=c5_side), 600)
        if rank == 0:
            extra["config5"] = c5
            promote_config5(out, c5, args, world, dist, one_device)
    # ---- BASELINE config 2 / 3 across sizes (isolated, one GPU)
    if rank == 0 and world == 1 and not args.no_sweeps and not stuck:
        if srs is not None:
            srs.free()
            srs = None
        extra["ntt_sweep"] = guarded("ntt_sweep", lambda: ntt_sweep(pkg, bbg, dev), 120)
        if not stuck:
            extra["msm_sweep"] = guarded("msm_sweep", lambda: msm_sweep(pkg, bbg, dev), 180)
    if rank == 0:
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(out) + "\n").encode())
    if stuck:
        os._exit(0)  # a guarded extra never returned: do not wait for it (or for the process group) on the way out
    if srs is not None:
        srs.free()
    bbg.close()
    if dist is not None:
        dist.destroy_process_group()


def step_issue_floor():
    """Sum of SQ_INSTS_VALU over every kernel of ONE step (committed PMC pass of this same bench command) / the chip's VALU issue peak:
    the time the step's instruction stream needs if every SIMD issued a VALU instruction every 4 clocks."""
    import re
    try:
        lines = open(PMC_PROFILE).read().splitlines()
    except OSError:
        return None
    m = next((mm for mm in (re.search(r"--steps (\d+) --warmup (\d+)", ln) for ln in lines[:4]) if mm), None)  # the header names the profiled command
    launches = (int(m.group(1)) + int(m.group(2))) if m else 4
    tot, per_kernel = 0.0, {}
    for line in lines:
        f = line.split()
        if len(f) >= 4 and f[-3] == "SQ_INSTS_VALU" and int(f[-2]) >= launches:
            k = " ".join(f[:-3]).replace("void bbg::", "").replace("bbg::", "")
            v = float(f[-1]) * int(f[-2]) / launches
            per_kernel[k] = round(v / 1e6, 2)
            tot += v
    if not tot:
        return None
    return {"ms": round(tot / (VALU_PEAK_GWAVE * 1e9) * 1e3, 4), "valu_wave_insts_per_step_M": round(tot / 1e6, 1),
            "per_kernel_M": per_kernel, "peak_G_per_s": round(VALU_PEAK_GWAVE, 1), "source": "profiles/" + os.path.basename(PMC_PROFILE),
            "profile_matches_build": profile_stamp()["profile_matches_build"]}


def host_path(pkg, bbg, srs, lg, reps=11):
    """SURVEY 8(d) drop-in timing: the host-buffer entry points the link-time shim calls (scalars / coefficients in pageable host memory
    in, result in host memory out: PCIe-inclusive), median of `reps` wall-clock calls.  Never `value`."""
    import ctypes
    n = 1 << lg
    lib, vp = bbg.lib, ctypes.c_void_p
    hs = pkg.synthetic_scalars(SEED + 3, n)
    hc = pkg.synthetic_scalars(SEED + 100 + lg, n)
    out = np.zeros(12, dtype=np.uint64)
    big = np.zeros((4 * n + 4, 4), dtype=np.uint64)

    def med(fn, k=reps):
        fn()
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(sorted(ts)[len(ts) // 2] * 1e3, 3)
    bbg.set_option("msm_async_reduce", 0)
    try:
        r = {"n": "2^%d" % lg, "reps": reps, "timing": "wall clock around the blocking C-ABI call, median",
             "bbg_msm_ms": med(lambda: bbg._ck(lib.bbg_msm(bbg.ctx, srs.handle, vp(hs.ctypes.data), 0, n, vp(out.ctypes.data)))),
             "bbg_ntt_ifft_ms": med(lambda: bbg._ck(lib.bbg_ntt(bbg.ctx, vp(hc.ctypes.data), lg, 1, 0, None))),
             "bbg_coset_fft_extend_4n_ms": med(lambda: bbg._ck(lib.bbg_coset_fft_extend(bbg.ctx, vp(hc.ctypes.data), lg, lg + 2, vp(big.ctypes.data))), 7)}
    finally:
        bbg.set_option("msm_async_reduce", 1)
    r["msm_mscalar_per_s"] = round(n / r["bbg_msm_ms"] / 1e3, 1)
    return r


def ntt_sweep(pkg, bbg, dev, sizes=(18, 19, 20, 21, 22, 23, 24)):
    """BASELINE config 2: fft / ifft / coset_fft / coset_ifft, device resident, in place, ISOLATED (nothing else on the GPU).  Method (the one
    every isolated NTT figure in README / DESIGN uses): HIP events on the launch stream around a burst of back-to-back transforms, best of 3
    bursts, ms per transform."""
    import torch
    rows = {}
    for lg in sizes:
        n = 1 << lg
        a = torch.from_numpy(pkg.synthetic_scalars(11, n).view(np.int64).reshape(-1)).to(dev)
        bbg.ntt_prepare(lg)
        burst = 50 if lg <= 21 else 12
        row = []
        for op in (0, 1, 2, 3):
            for _ in range(3):
                bbg.ntt_device(a.data_ptr(), lg, op)
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(burst):
                    bbg.ntt_device(a.data_ptr(), lg, op)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / burst)
            row.append(round(best, 4))
        rows["2^%d" % lg] = {"fft_ms": row[0], "ifft_ms": row[1], "coset_fft_ms": row[2], "coset_ifft_ms": row[3],
                             "fft_gfield_ops_per_s": round(1.5 * n * lg / row[0] / 1e6, 1), "fft_hbm_frac_64n": round(64.0 * n / (row[0] * 1e-3) / 8e12, 4)}
        del a
    return {"method": "HIP events around a burst of back-to-back in-place transforms (50; 12 from 2^22), best of 3 bursts; isolated, device resident", "sizes": rows}


def msm_sweep(pkg, bbg, dev, sizes=(16, 20, 22, 24)):
    """BASELINE config 3 across sizes: n scalars over an n-point hashed SRS (window tables resident), device-resident scalars.
    standalone = one call + synchronise (wall clock, median of 7); pipelined = a burst of calls with the reduce phase of call i overlapping
    call i+1 (msm_async_reduce), per call."""
    import torch
    rows = {}
    out = torch.zeros(12, dtype=torch.int64, device=dev)
    for lg in sizes:
        n = 1 << lg
        srs = bbg.srs_synth_hashed(SEED, n)
        sc = torch.from_numpy(pkg.synthetic_scalars(SEED + 3, n).view(np.int64).reshape(-1)).to(dev)

        def med(fn, reps):
            fn()
            bbg.join(); bbg.sync()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                bbg.join(); bbg.sync()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3
        bbg.set_option("msm_async_reduce", 0)
        sa = med(lambda: bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr()), 7)
        bbg.set_option("msm_async_reduce", 1)
        burst = 16 if lg <= 20 else 4

        def run_burst():
            for _ in range(burst):
                bbg.msm_device(srs, sc.data_ptr(), n, out.data_ptr())
        pl = med(run_burst, 5) / burst
        rows["2^%d" % lg] = {"standalone_ms": round(sa, 3), "pipelined_ms": round(pl, 3), "mscalar_per_s_pipelined": round(n / pl / 1e3, 1)}
        srs.free()
        del sc
    return {"method": "wall clock, device-resident scalars; standalone = call + sync (median of 7), pipelined = burst of 16 (4 from 2^22) calls, per call (median of 5)",
            "sizes": rows}


def config5(pkg, par, bbg, dist, dev, rank, world, lg, steps=3, timed_steps=0, warmup=1, blocks=1, side=None):
    """BASELINE config 5 -- one 2^lg MSM (point-range shards, all-gather of the 96-byte partials + group sum) and one 2^lg coset NTT
    (residue-class shards, one all-to-all, size-N DFT) over the N ranks, inputs resident, STRONG scaling (total work fixed as N grows).

    Two measurements: (i) the legs one at a time (max over ranks, best of `steps`): msm_ms / ntt_ms; (ii) with timed_steps = K > 0 the
    contract's timed region on THIS workload: `warmup` untimed steps, then `blocks` blocks of exactly K steps (1 sharded MSM + 1 sharded
    coset NTT each) between two fences (pipeline flush + barrier + synchronize), max over ranks, median block -> `timed` (the N > 1
    headline of the line: value = 2^lg / seconds per step).  `side` = (stream, context): the exchange of earlier MSMs' partials runs there.

    Self-checking: the inputs are the seeded ones of tests/golden/msm24.json (hashed SRS 0xBB254, synthetic_scalars(0xBB254 + 24))
    and tests/golden/ntt_large.json (synthetic_scalars(900 + lg), op coset_fft), both recorded from the compiled REFERENCE, so at
    lg = 24 rank 0 compares the group sum with the reference's 2^24 result and the SHA-256 of the gathered, canonicalised transform
    with the reference's digest: `bit_exact_vs_reference`.  Only committed fixture DATA is read; nothing under oracle/ is used."""
    import hashlib
    import torch
    n = 1 << lg
    golden_dir = os.path.join(ROOT, "tests", "golden")
    want_msm = want_ntt = None
    try:
        g24 = json.load(open(os.path.join(golden_dir, "msm24.json")))
        if g24["log2n"] == lg:
            want_msm = np.frombuffer(bytes.fromhex(g24["result"]), dtype=np.uint64)
        for rec in json.load(open(os.path.join(golden_dir, "ntt_large.json")))["ntt"]:
            if rec["log2n"] == lg and rec["op"] == 2 and rec["generator_size"] == 0:
                want_ntt = rec["sha256"]
        # the same two legs recorded from the reference at sizes a one-device rehearsal runs in seconds (gen_golden_config5_small.py)
        for rec in json.load(open(os.path.join(golden_dir, "config5_small.json")))["sizes"]:
            if rec["log2n"] == lg:
                want_msm = np.frombuffer(bytes.fromhex(rec["msm_result"]), dtype=np.uint64)
                want_ntt = rec["ntt_sha256"]
    except (OSError, KeyError, ValueError):
        pass
    start, count = par.shard_range(n, rank, world)
    srs = bbg.srs_synth_hashed(SEED + start, count)
    d_scalars = torch.from_numpy(pkg.synthetic_scalars(SEED + 24, count, start).view(np.int64).reshape(-1)).to(dev)
    m = n // world
    d_x = torch.from_numpy(pkg.synthetic_scalars_strided(900 + lg, m, rank, world).view(np.int64).reshape(-1)).to(dev)
    new_tensor = lambda k: torch.zeros(k, dtype=torch.int64, device=dev)  # noqa: E731
    pipe = par.ShardedMsmPipeline(par.BbgOps(bbg, srs), dist, new_tensor)
    ops = par.BbgNttOps(bbg)
    five = np.array([[5, 0, 0, 0]], dtype=np.uint64)
    shift = bbg.field_op(0, 5, five)[0]  # the coset generator 5 in Montgomery form (fr.hpp:44-59)
    bbg.ntt_prepare(lg - (world.bit_length() - 1))
    work = d_x.clone()
    spare = [torch.empty_like(d_x), torch.empty_like(d_x)]  # the all-to-all's receive buffer and the cross-rank DFT's output, reused every step
    turn = [0]

    def new_like(_):
        turn[0] ^= 1
        return spare[turn[0]]
    last = {}

    def fence(p=None):
        if p is not None:
            p.flush()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def msm_step():
        pipe.reset()
        pipe.submit(d_scalars, count)
        last["msm"] = pipe.flush()

    def ntt_step(restore=True):
        if restore:
            work.copy_(d_x)
        if world > 1:
            last["ntt"] = par.ntt_sharded(ops, dist, work, lg, coset_shift=shift, new_like=new_like)
        else:
            bbg.ntt_device(work.data_ptr(), lg, 2)
            last["ntt"] = work

    def max_over_ranks(t):
        if dist is not None:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    def best(fn):
        fn()
        fence()
        ts = []
        for _ in range(steps):
            fence()
            t0 = time.perf_counter()
            fn()
            fence()
            ts.append(max_over_ranks(time.perf_counter() - t0))
        return min(ts)

    t_msm, t_ntt = best(msm_step), best(ntt_step)
    # ---- the contract's timed region on this workload (N > 1: the line's headline)
    timed = None
    if timed_steps > 0:
        if side is not None and dist is not None:
            tpipe = par.ShardedMsmPipeline(par.BbgOps(bbg, srs, side[1]), dist, new_tensor, side_stream=side[0], depth=4)
        else:
            tpipe = par.ShardedMsmPipeline(par.BbgOps(bbg, srs), dist, new_tensor)

        def step():  # the transform runs on its own previous output from the second step on: any input is the same work (no restore copy)
            tpipe.submit(d_scalars, count)
            ntt_step(restore=False)
        work.copy_(d_x)
        for _ in range(warmup):
            step()
        fence(tpipe)
        tblocks = []
        for _ in range(max(1, blocks)):
            tpipe.reset()
            bbg.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(timed_steps):
                step()
            fence(tpipe)
            el = time.perf_counter() - t0
            pr = {k: bbg.profile_get(k) for k in ("msm_recode", "msm_sort", "msm_accumulate", "msm_reduce", "ntt_pass")}
            bbg.profile_enable(False)
            tblocks.append((max_over_ranks(el), pr))
        order = sorted(range(len(tblocks)), key=lambda i: tblocks[i][0])
        el, pr = tblocks[order[len(order) // 2]]
        acc_ms = pr["msm_accumulate"][0] / max(1, pr["msm_accumulate"][1])
        width, windows = bbg.msm_plan(count, srs)
        timed = {"steps": timed_steps, "warmup": warmup, "blocks_ms": [round(b[0] * 1e3, 3) for b in tblocks], "reported_block": "median",
                 "ms_per_step": round(el * 1e3 / timed_steps, 4), "value_mscalar_per_s": round(n / (el / timed_steps) / 1e6, 3),
                 "ntt_gfield_ops_per_s": round(1.5 * n * lg / (el / timed_steps) / 1e9, 2),
                 "per_rank": {"points": count, "msm_window_bits": width, "msm_windows": windows, "accumulate_avg_launch_ms": round(acc_ms, 4),
                              "phase_ms_per_step": {k: round(v[0] / timed_steps, 4) for k, v in pr.items()}}}
    # ---- self-check against the reference's recorded results (after the timed regions; the legs are re-run on the recorded inputs)
    check = {"msm": None, "ntt": None, "source": "tests/golden/msm24.json, ntt_large.json, config5_small.json (compiled reference)"}
    if timed is not None:
        msm_step()
        ntt_step()
        fence()
    if want_msm is not None:
        jac = last["msm"].cpu().numpy().view(np.uint64).reshape(1, 12)
        if rank == 0:
            check["msm"] = bool(np.array_equal(bbg.g1_normalize(jac).reshape(-1), want_msm))
    if want_ntt is not None:
        nat = par.gather_natural_order(dist, last["ntt"], lg)  # rank g holds out[t*(m/G) + q] = A[(g*m/G + q) + m*t]  (parallel.ntt_sharded)
        if rank == 0:
            check["ntt"] = hashlib.sha256(pkg.fr_reduce_once(nat).tobytes()).hexdigest() == want_ntt
    srs.free()
    res = {"workload": "ONE 2^%d-point MSM + ONE 2^%d coset NTT over %d GPU(s), strong scaling (north_star's split: point-range MSM shards + "
                       "all-gather of 96-B partials; residue-class NTT shards + one all-to-all)" % (lg, lg, world),
           "n_gpus": world, "msm_ms": round(t_msm * 1e3, 3), "msm_mscalar_per_s": round(n / t_msm / 1e6, 2),
           "ntt_ms": round(t_ntt * 1e3, 3), "ntt_gfield_ops_per_s": round(1.5 * n * lg / t_ntt / 1e9, 2),
           "ntt_includes_input_restore_copy": True, "bit_exact_vs_reference": check,
           "exchange": {"msm": "all_gather %d x 96 B" % world, "ntt": "all_to_all %.1f MiB per rank" % (32.0 * m * (world - 1) / world / 2**20)}}
    if timed is not None:
        res["timed"] = timed
    return res


def prover_shaped(pkg, bbg, srs, lg, reps=5):
    """BASELINE config 4's sequence -- what one TurboPLONK proof asks of the hot path: 11 MSM(n) + 5 iFFT(n) + 5 coset FFT(4n) +
    1 coset iFFT(4n) and every O(n) step between them (grand product, five quotient widgets, division by Z*_H, 16 evaluations,
    linearisation, two opening accumulations, two Kate divisions) -- through the product's resident prover rounds (bbg_prover_*,
    the entry points shim/bbg_resident_prover.hpp drives from the reference's TurboProver).  Synthetic key / witness polynomials and
    challenges (timing does not depend on the values); wires go up over PCIe per proof, 11 commitments come down.
    tests/test_gpu_parity.py runs the same rounds under the reference's real prover and checks the proof byte for byte."""
    import ctypes
    lib, n = bbg.lib, 1 << lg
    to_mont = lambda k: bbg.field_op(0, 5, np.array([[k, 0, 0, 0]], dtype=np.uint64))[0]
    gens = np.stack([to_mont(5), to_mont(5), to_mont(6), to_mont(7)])
    h = ctypes.c_void_p()
    bbg._ck(lib.bbg_prover_create(bbg.ctx, srs.handle, lg, 4, gens.ctypes.data, ctypes.byref(h)))
    try:
        for pid in range(5, 20):  # sigma_1..4, q_1..q_5, q_m, q_c, q_arith, q_fixed_base, q_range, q_logic (coefficient forms)
            a = pkg.synthetic_scalars(SEED + 500 + pid, n)
            bbg._ck(lib.bbg_prover_set_key_poly(h, pid, 0, a.ctypes.data))
        t0 = time.perf_counter()
        bbg._ck(lib.bbg_prover_finalize_key(h))
        t_key = time.perf_counter() - t0
        wires = [pkg.synthetic_scalars(SEED + 600 + k, n) for k in range(4)]
        wp = (ctypes.c_void_p * 4)(*[w.ctypes.data for w in wires])
        ch = pkg.synthetic_scalars(SEED + 700, 40)
        com = np.zeros((4, 12), dtype=np.uint64)
        ev = np.zeros((32, 4), dtype=np.uint64)
        ids16 = (ctypes.c_int * 16)(0, 0, 1, 1, 2, 2, 3, 3, 4, 15, 16, 17, 5, 6, 7, 21)  # manifest order: w_i, w_i_omega, z_omega, q_c, q_arith, q_ecc_1, sigma_1..3, t
        sh16 = (ctypes.c_int * 16)(0, 1, 0, 1, 0, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0)
        lin = (ctypes.c_int * 12)(4, 8, 9, 10, 11, 12, 13, 14, 15, 18, 19, 16)
        at_zeta = (ctypes.c_int * 14)(0, 1, 2, 3, 15, 16, 17, 5, 6, 7, 23, 24, 25, 26)
        at_omega = (ctypes.c_int * 5)(0, 1, 2, 3, 4)
        rounds = []
        for _ in range(reps + 1):
            t = [time.perf_counter()]
            bbg._ck(lib.bbg_prover_round1(h, wp, com.ctypes.data)); t.append(time.perf_counter())
            bbg._ck(lib.bbg_prover_round3(h, ch[0].ctypes.data, ch[1].ctypes.data, ch[2:5].ctypes.data, com.ctypes.data)); t.append(time.perf_counter())
            bbg._ck(lib.bbg_prover_round4(h, ch[5].ctypes.data, ch[6].ctypes.data, com.ctypes.data)); t.append(time.perf_counter())
            bbg._ck(lib.bbg_prover_evaluate(h, 16, ids16, sh16, ch[7].ctypes.data, ev.ctypes.data))
            bbg._ck(lib.bbg_prover_linearise(h, 12, lin, ch[8:20].ctypes.data, ch[7].ctypes.data, ev.ctypes.data)); t.append(time.perf_counter())
            bbg._ck(lib.bbg_prover_round6(h, 14, at_zeta, ch[20:34].ctypes.data, 5, at_omega, ch[34:39].ctypes.data, ch[7].ctypes.data,
                                          ch[39].ctypes.data, None, com.ctypes.data, com[1:].ctypes.data)); t.append(time.perf_counter())
            rounds.append([t[i + 1] - t[i] for i in range(5)])
        rounds = rounds[1:]  # the first pass allocates scratch and builds tables
        tot = sorted(sum(r) for r in rounds)
        med = [sorted(r[i] for r in rounds)[len(rounds) // 2] for i in range(5)]
        return {"workload": "TurboPLONK prover sequence at n = 2^%d on the resident prover rounds (11 MSM + 5 iFFT + 5 coset FFT(4n) + coset iFFT(4n) + "
                            "all O(n) round arithmetic); wires uploaded per proof (PCIe-inclusive), commitments downloaded" % lg,
                "proof_ms": round(tot[len(tot) // 2] * 1e3, 3), "proof_ms_min": round(tot[0] * 1e3, 3), "reps": len(rounds),
                "round_ms": {"round1_wires_ifft_commit": round(med[0] * 1e3, 3), "round3_z": round(med[1] * 1e3, 3), "round4_quotient": round(med[2] * 1e3, 3),
                             "round5_evaluate_linearise": round(med[3] * 1e3, 3), "round6_openings": round(med[4] * 1e3, 3)},
                "key_finalize_ms_once_per_circuit": round(t_key * 1e3, 1)}
    finally:
        lib.bbg_prover_destroy(h)


def reference_prover_sequence(srs, scalars, coeffs, lg):
    """The reference binary (oracle/_ref/libbbref.so) on the same MSM / FFT sequence on the host cores, once: 11 pippenger_unsafe(n),
    5 ifft(n), 5 coset_fft(4n, generator_size n), 1 coset_ifft(4n).  (The O(n) round arithmetic of the reference prover is NOT included:
    this is the MSM + FFT wall-clock north_star's 10x target refers to.)  CPU baseline leg only."""
    from oracle.oracle import Ref
    ref = Ref()
    n = 1 << lg
    ctx = ref.msm(srs.read())
    t_msm = sum(ctx.run(scalars, 0, True)[1] for _ in range(11))
    ctx.free()
    dom = ref.domain(lg, 0)
    t_ifft = sum(dom.run(coeffs, 1)[1] for _ in range(5))
    dom.free()
    big = np.zeros((4 * n, 4), dtype=np.uint64)
    big[:n] = coeffs
    dom = ref.domain(lg + 2, n)
    t_cfft = sum(dom.run(big, 2)[1] for _ in range(5))
    t_cifft = dom.run(big, 3)[1]
    dom.free()
    return {"total_ms": round((t_msm + t_ifft + t_cfft + t_cifft) * 1e3, 1), "msm_x11_ms": round(t_msm * 1e3, 1), "ifft_x5_ms": round(t_ifft * 1e3, 1),
            "coset_fft_4n_x5_ms": round(t_cfft * 1e3, 1), "coset_ifft_4n_ms": round(t_cifft * 1e3, 1), "threads": ref.num_threads()}


def real_prover_baseline(log2_gates):
    """CPU baseline leg: the reference's REAL TurboPLONK prover (its own composer, prover and verifier compiled from /root/reference into
    oracle/_ref) on a 2^log2_gates-gate circuit -- on the host cores as shipped, and the SAME driver object linked with the shim
    (oracle/_ref/libbbprover_wrap.so: MSM / FFT entry points and construct_proof() wrapped at link time, zero source edits), first with the
    prover wrap switched off (link-only: the round arithmetic stays on the host), then on.  The wrapped proof replays the blinding scalars
    the CPU proof drew and must equal it byte for byte; every proof is verified by the reference verifier."""
    from oracle.oracle import Oracle, RefProver, PROVER_WRAP_SO, prover_available
    if not prover_available() or not os.path.exists(PROVER_WRAP_SO):
        return {"error": "oracle/_ref/libbbprover[_wrap].so not shipped"}
    O = Oracle()
    x = O.to_mont(0, np.array([[0x1234567890ABCDEF, 0xFEDCBA, 0, 0]], dtype=np.uint64))[0]
    gates = (1 << log2_gates) - 64
    pts = O.srs_powers(x, (1 << log2_gates) + 2)
    # the CPU prover is timed like the wrapped one: a first session pays page faults, OpenMP team start-up and table builds, the SECOND
    # (a fresh session: the reference body cannot prove twice on one prover object) is the one compared with the warm wrapped proofs
    A = RefProver(gates, 11, pts, x)
    t0 = time.perf_counter()
    A.prove_recording()
    t_cpu_first = time.perf_counter() - t0
    A.free()
    A = RefProver(gates, 11, pts, x)
    t0 = time.perf_counter()
    cpu, blind = A.prove_recording()
    t_cpu = time.perf_counter() - t0
    ok = [A.verify()]
    threads = A.threads
    A.free()
    # link-only: a warm-up session first (window tables, twiddles, scratch); the reference body cannot prove twice on one prover object
    # (it leaves the witness in coefficient form), so the timed proof gets a session of its own
    t_link = None
    for timed in (False, True):
        W = RefProver(gates, 11, pts, x, wrap_linked=True)
        try:
            W.wrap_set_enabled(False)
            t0 = time.perf_counter()
            W.prove_reference()
            t_link = time.perf_counter() - t0
            if timed:
                ok.append(W.verify())
        finally:
            W.wrap_set_enabled(True)
            W.free()
    W = RefProver(gates, 11, pts, x, wrap_linked=True)
    t0 = time.perf_counter()
    got = W.prove_reference(replay=blind)
    t_first = time.perf_counter() - t0
    ok.append(W.verify())
    warm = []
    for _ in range(5):
        W.lib.refp_reset(W.h)
        t0 = time.perf_counter()
        W.prove_reference()
        warm.append(time.perf_counter() - t0)
    ok.append(W.verify())  # after the loop: the proofs run back to back
    W.free()
    W.wrap_trim()
    t_wrap = sorted(warm)[2]
    return {"prover": "TurboPLONK (TurboComposer::create_prover), reference build, arithmetic circuit", "log2_gates": log2_gates, "host_threads": threads,
            "cpu_ms": round(t_cpu * 1e3, 1), "cpu_first_ms": round(t_cpu_first * 1e3, 1), "link_only_ms": round(t_link * 1e3, 1),
            "wrapped_zero_edits_first_ms": round(t_first * 1e3, 1), "timing": "cpu_ms / link_only_ms: second session (warm); wrapped_zero_edits_ms: median of 5 warm proofs; *_first_ms: cold",
            "speedup_wrapped_vs_cpu_cold": round(t_cpu_first / t_first, 1),
            "wrapped_zero_edits_ms": round(t_wrap * 1e3, 2), "byte_identical_to_cpu_proof": got == cpu, "verified": ok,
            "speedup_wrapped_vs_cpu": round(t_cpu / t_wrap, 1)}


def config1_baseline(pkg, bbg, oracle, ref, restore_threads):
    """BASELINE config 1 -- "barretenberg CPU pippenger + fft bench, n = 2^16, bit-exact reference, no GPU" -- in THIS run: the reference
    binary's pippenger_unsafe (scalar_multiplication.cpp:923) and fft (polynomial_arithmetic.cpp:374) at n = 2^16 on the host cores (team capped
    at 16 threads: its compute_wnaf_states is unsafe with a large team on a small input), with the GPU's results on the same inputs
    checked against them and timed beside them (device-resident inputs)."""
    import torch
    lg = 16
    n = 1 << lg
    threads = min(os.cpu_count() or 1, 16)
    srs = bbg.srs_synth_hashed(SEED + 16, n)
    try:
        points = srs.read()
        scalars = pkg.synthetic_scalars(SEED + 16 + 3, n)
        coeffs = pkg.synthetic_scalars(SEED + 100 + lg, n)
        ref.set_threads(threads)
        try:
            ctx = ref.msm(points)
            runs = [ctx.run(scalars, 0, True) for _ in range(4)]
            ctx.free()
            dom = ref.domain(lg, 0)
            ntt_runs = [dom.run(coeffs, 0) for _ in range(6)]
            dom.free()
        finally:
            ref.set_threads(restore_threads)
        cpu_msm, cpu_ntt = min(t for _, t in runs[1:]), min(t for _, t in ntt_runs[1:])
        dev = torch.device("cuda", torch.cuda.current_device())
        d_sc = torch.from_numpy(scalars.view(np.int64)).to(dev)
        d_out = torch.zeros(12, dtype=torch.int64, device=dev)
        d_co = torch.from_numpy(coeffs.view(np.int64)).to(dev)

        def med(fn, reps=9):
            fn()
            bbg.join(); bbg.sync()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                bbg.join(); bbg.sync()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]
        gpu_msm_s = med(lambda: bbg.msm_device(srs, d_sc.data_ptr(), n, d_out.data_ptr()))
        gpu_msm = oracle.jac_to_affine(d_out.cpu().numpy().view(np.uint64))
        work = d_co.clone()
        bbg.ntt_device(work.data_ptr(), lg, 0)
        bbg.sync()
        gpu_ntt = oracle.canon(0, work.cpu().numpy().view(np.uint64))
        gpu_ntt_s = med(lambda: bbg.ntt_device(work.data_ptr(), lg, 0))
        return {"n": "2^16", "cores": threads, "kind": "reference", "msm_ms": round(cpu_msm * 1e3, 2), "msm_mscalar_per_s": round(n / cpu_msm / 1e6, 3),
                "fft_ms": round(cpu_ntt * 1e3, 3), "fft_gfield_ops_per_s": round(1.5 * n * lg / cpu_ntt / 1e9, 3),
                "sample": "reference binary (oracle/_ref): pippenger_unsafe best of 3 warm + fft best of 5 warm, n = 2^16, hashed SRS / splitmix scalars",
                "gpu_msm_ms": round(gpu_msm_s * 1e3, 3), "gpu_fft_ms": round(gpu_ntt_s * 1e3, 4),
                "gpu_bit_exact_vs_cpu": bool(np.array_equal(runs[-1][0], gpu_msm) and np.array_equal(ntt_runs[-1][0], gpu_ntt))}
    finally:
        srs.free()


def cpu_baseline(pkg, bbg, srs, scalars, coeffs, d_result, d_coeffs, lg, gpu_value):
    """Times the CPU path on the host cores on a bounded sample of the same workload and checks parity."""
    from oracle.oracle import Oracle, Ref, ref_available
    n = 1 << lg
    oracle = Oracle()
    gpu_msm = oracle.jac_to_affine(d_result.cpu().numpy().view(np.uint64))
    work = d_coeffs.clone()
    bbg.ntt_device(work.data_ptr(), lg, 0)
    bbg.sync()
    gpu_ntt = oracle.canon(0, work.cpu().numpy().view(np.uint64))
    points = srs.read()
    if ref_available():
        ref = Ref()
        if lg < 18:  # the reference's CPU pippenger is unsafe with a large OpenMP team on small inputs
            ref.set_threads(min(os.cpu_count() or 1, 16))

        ctx = ref.msm(points)
        best = 1e9
        for _ in range(2):
            res, t = ctx.run(scalars, 0, True)
            best = min(best, t)
        ctx.free()
        dom = ref.domain(lg, 0)
        best_ntt = 1e9
        for _ in range(3):
            ntt_out, t = dom.run(coeffs, 0)
            best_ntt = min(best_ntt, t)
        dom.free()
        parity = bool(np.array_equal(res, gpu_msm) and np.array_equal(ntt_out, gpu_ntt))
        val = n / best / 1e6
        cores = ref.num_threads()
        try:
            config1 = config1_baseline(pkg, bbg, oracle, ref, cores)
        except Exception as e:  # noqa: BLE001 -- reported in the line, never raised
            config1 = {"error": "%s: %s" % (type(e).__name__, e)}
        return {"value": round(val, 3), "unit": "Mscalar-mults/s", "cores": cores, "kind": "reference", "config1": config1,
                "sample": "reference binary (oracle/_ref): pippenger_unsafe n=2^%d best of 2 (%.1f ms) + fft n=2^%d best of 3 "
                          "(%.1f ms), same SRS/scalars/coefficients as the GPU run" % (lg, best * 1e3, lg, best_ntt * 1e3),
                "ntt_gfield_ops_per_s": round(1.5 * n * lg / best_ntt / 1e9, 3), "msm_ms": round(best * 1e3, 2),
                "ntt_ms": round(best_ntt * 1e3, 2), "gpu_bit_exact_vs_cpu": parity, "gpu_over_cpu": round(gpu_value / val, 1)}
    # port: the oracle is a scalar restatement; sample 2^16 terms of the same inputs
    m = min(n, 1 << 16)
    t0 = time.perf_counter()
    res = oracle.pippenger(scalars[:m], points[:m])
    t_msm = time.perf_counter() - t0
    part = oracle.jac_to_affine(bbg.msm(srs, scalars[:m]))
    t0 = time.perf_counter()
    ntt_out = oracle.ntt(coeffs, 0)
    t_ntt = time.perf_counter() - t0
    parity = bool(np.array_equal(res, part) and np.array_equal(ntt_out, gpu_ntt))
    val = m / t_msm / 1e6
    return {"value": round(val, 3), "unit": "Mscalar-mults/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "oracle port: bucket MSM on the first 2^16 terms (%.1f ms) + full fft n=2^%d (%.1f ms)" % (t_msm * 1e3, lg, t_ntt * 1e3),
            "ntt_gfield_ops_per_s": round(1.5 * n * lg / t_ntt / 1e9, 3), "gpu_bit_exact_vs_cpu": parity,
            "gpu_over_cpu": round(gpu_value / val, 1)}


if __name__ == "__main__":
    main()
