"""bench.py's launcher, per-rank supervisors (the first-contact LADDER, round 5) and N-rank plumbing, without a GPU: `python bench.py --gpus 2`
with no WORLD_SIZE must start its ranks itself -- from a parent that never touches the device -- and relay ONE well-formed JSON line
with n_gpus = 2; the torchrun form the driver uses for N > 1 must keep working; a rung that fails, hangs or fails its self-check on ANY
rank must be left by ALL ranks for the next one, and the line must say which rung produced it and why the earlier ones were left.
tests/bench_dryrun.py is bench.py with the device operations swapped for tests/dp_oracle_ops.py over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "tests", "bench_dryrun.py")   # the main() of bench.py with CPU stand-ins for rank_main


def _env():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return env


def _one_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]   # gloo itself chats on stdout ("[Gloo] Rank 0 is connected ...")
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "rccl"]


def test_self_launch_two_ranks_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["rccl"]["world"] == 2 and d["rccl"]["launched_by"] == "1" and "self" in d["rccl"]["launcher"]
    assert d["config"]["per_gpu_batch"] * 2 == d["config"]["global_batch"] and d["config"]["parallelism"] == "dp2"
    # the ladder's first rung produced the line, and the step-1 self-check ran and passed on it
    assert d["rccl"]["mode"] == "default" and d["rccl"]["rung"] == "1 of 2" and d["rccl"]["fallback_reason"] is None
    sc = d["rccl"]["selfcheck"]
    assert sc["world_from_communicator"] == 2 == sc["world_measured_by_allreduce"] and sc["loss_rel_diff"] <= 1e-6
    assert sc["params_identical_before_step_1"] and sc["params_identical_after_last_step"] and sc["violations"] == []


def test_torchrun_form_still_works():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl"]["world"] == 2 and d["rccl"]["launched_by"] == "0" and "launcher" not in d["rccl"]
    assert d["rccl"]["mode"] == "default"   # the driver's own torchrun job walks the same ladder: every rank it starts is a supervisor


def test_single_rank_and_emulated_world():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["per_gpu_batch"] == d["config"]["global_batch"]
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--emulate-world", "4"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["per_gpu_batch"] * 4 == d["config"]["global_batch"] and "EMULATED" in d["metric"]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def _run(args, env, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_failed_abi_rung_falls_to_the_next_one():
    """--dp-backend auto puts the C-ABI communicator on the first rung; when it fails on every rank the job reruns as fresh children on the
    torch.distributed rung.  --dp-backend abi: its only other rung is "plain" (torch.distributed, one all-reduce)."""
    env = _env()
    env["LRCN_BENCH_DRYRUN_FAIL_ABI"] = "1"
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dp-backend", "auto"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl"]["mode"] == "default" and d["rccl"]["rung"] == "2 of 3" and "abi" in d["rccl"]["fallback_reason"]
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--dp-backend", "abi"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["rccl"]["mode"] == "plain" and d["rccl"]["env_of_rung"]["LRCN_DP_BACKEND"] == "torch"


@pytest.mark.parametrize("how", ["crash", "hang", "selfcheck"])
def test_one_rank_failing_on_the_default_rung_moves_every_rank_to_plain(how):
    """Rank 1 alone crashes / stops making progress (a hung collective) / breaks the replicas before the self-check on the "default" rung:
    rank 0's supervisor stops its own child as soon as rank 1's verdict is in (or its heartbeat is stale), and both rerun as fresh children
    on "plain" -- one all-reduce of the flat buffer, no sparse exchange -- whose line names the rung and the reason."""
    env = _env()
    env["LRCN_BENCH_STALL_S"] = "6"
    key = {"crash": "LRCN_BENCH_DRYRUN_FAIL_RUNG", "hang": "LRCN_BENCH_DRYRUN_HANG_RUNG", "selfcheck": "LRCN_BENCH_DRYRUN_BAD_SELFCHECK"}[how]
    env[key] = "default" if how == "selfcheck" else "default:1"
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    rc = d["rccl"]
    assert rc["mode"] == "plain" and rc["rung"] == "2 of 2" and rc["world"] == 2
    assert rc["env_of_rung"] == {"LRCN_DP_BUCKETS": "0", "LRCN_DP_SPARSE_EMBED": "0", "LRCN_DP_BACKEND": "torch"}
    assert "default: rank" in rc["fallback_reason"], rc["fallback_reason"]
    assert {"crash": "exit code", "hang": "no progress", "selfcheck": "exit code"}[how] in rc["fallback_reason"], rc["fallback_reason"]
    assert rc["selfcheck"]["violations"] == [] and rc["selfcheck"]["params_identical_after_last_step"]
    assert "rung 'default' failed" in r.stderr


def test_a_teardown_that_hangs_after_the_line_does_not_cost_the_rung_its_number():
    """Rank 1 never returns from its teardown on the first rung, after the timed region, the reduction and rank 0's print: its supervisor stops it
    after LRCN_BENCH_TEARDOWN_S and the rung still counts -- the line comes from "default", not from a rerun on "plain"."""
    env = _env()
    env["LRCN_BENCH_DRYRUN_HANG_TEARDOWN"] = "default:1"
    env["LRCN_BENCH_TEARDOWN_S"] = "2"
    env["LRCN_BENCH_STALL_S"] = "60"
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["rccl"]["mode"] == "default" and d["rccl"]["rung"] == "1 of 2" and d["rccl"]["fallback_reason"] is None


def test_a_rung_that_exits_zero_without_its_line_is_left_by_every_rank():
    """ADVICE r5 (launch.py): all ranks exit 0 on "default" but rank 0 printed nothing.  The verdict must be the same on every supervisor --
    rank 0 publishes the missing line as its own failure -- so that both ranks rerun on "plain" (before: rank 1 left with rc 0, rank 0
    raised IndexError / would have gone on alone into a rendezvous nobody joins)."""
    env = _env()
    env["LRCN_BENCH_DRYRUN_NO_LINE"] = "default"
    env["LRCN_BENCH_STALL_S"] = "20"
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["rccl"]["mode"] == "plain" and d["rccl"]["rung"] == "2 of 2" and d["rccl"]["world"] == 2
    assert "printed no result line" in d["rccl"]["fallback_reason"], d["rccl"]["fallback_reason"]
    assert "Traceback" not in r.stderr
    # and on the LAST rung: an error for every rank, no line, no traceback
    env["LRCN_BENCH_RUNGS_ONLY"] = "plain"
    env["LRCN_BENCH_DRYRUN_NO_LINE"] = "plain"
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "giving up" in r.stderr and "IndexError" not in r.stderr and "printed no result line" in r.stderr   # (torchrun prints its own ChildFailedError)


def test_eight_ranks_walk_the_ladder_with_a_stalled_rank_five():
    """the shape the driver launches (8 ranks, torchrun form) on CPU stand-ins: rank 5 stops making progress on the
    default rung; all eight supervisors stop their children and rerun on "plain"; the line carries n_gpus = 8 and a passed self-check."""
    env = _env()
    env["LRCN_BENCH_STALL_S"] = "10"
    env["LRCN_BENCH_DRYRUN_HANG_RUNG"] = "default:5"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29631",
           BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    rc = d["rccl"]
    assert d["n_gpus"] == 8 and rc["world"] == 8 and rc["mode"] == "plain" and rc["rung"] == "2 of 2"
    # (a hung collective stalls EVERY rank at the same stage: the reason names the lowest stalled rank, not necessarily the culprit)
    assert "default: rank" in rc["fallback_reason"] and "no progress" in rc["fallback_reason"], rc["fallback_reason"]
    assert rc["selfcheck"]["violations"] == [] and rc["selfcheck"]["params_identical_after_last_step"]
    assert d["config"]["per_gpu_batch"] == 1


def test_eight_ranks_default_rung_self_launched():
    env = _env()
    env["OMP_NUM_THREADS"] = "1"
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 8 and d["rccl"]["mode"] == "default" and d["rccl"]["rung"] == "1 of 2" and d["rccl"]["fallback_reason"] is None
    assert d["rccl"]["selfcheck"]["violations"] == []


def test_every_rung_failing_is_an_error_without_a_line():
    env = _env()
    env["LRCN_BENCH_DRYRUN_FAIL_ABI"] = "1"
    env["LRCN_BENCH_DRYRUN_FAIL_RUNG"] = "plain:0"
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dp-backend", "abi"], env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "giving up" in r.stderr


def test_selfcheck_violation_on_the_last_rung_is_reported_not_fatal():
    env = _env()
    env["LRCN_BENCH_RUNGS_ONLY"] = "plain"
    env["LRCN_BENCH_DRYRUN_BAD_SELFCHECK"] = "plain"
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["rccl"]["mode"] == "plain" and "params" in d["rccl"]["selfcheck"]["violations"]
    assert d["rccl"]["selfcheck"]["params_identical_after_last_step"] is False


def test_parent_never_imports_torch():
    """The self-launching parent must not initialise the GPU: it may not even import torch (a HIP context in the parent of the ranks is
    exactly what the contract forbids).  Checked by running the parent with an import hook that fails on torch."""
    code = ("import sys, runpy\n"
            "class Block:\n"
            "    def find_spec(self, name, path=None, target=None):\n"
            "        if name == 'torch' or name.startswith('torch.'):\n"
            "            raise ImportError('parent imported torch')\n"
            "sys.meta_path.insert(0, Block())\n"
            "sys.argv = [%r, '--gpus', '2', '--steps', '1', '--warmup', '0']\n"
            "runpy.run_path(%r, run_name='__main__')\n" % (BENCH, BENCH))
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["n_gpus"] == 2


def test_pmc_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed --pmc pass (a counter pass is its own run): bench.py quotes it only while csrc/'s digest
    equals the one stamped into the profile file, and says so otherwise (the number went stale silently)."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    val, note = bench.pmc_traffic("f32", 256)
    assert val is None and "configuration" in note
    val, note = bench.pmc_traffic("bf16", 32)
    assert val is None
    val, note = bench.pmc_traffic("bf16", 256)
    assert (val is None and "stale" in note) or (val > 1e8 and bench.csrc_digest() in note)
    monkeypatch.setattr(bench, "csrc_digest", lambda: "0" * 16)    # as if a kernel source had changed since the measurement
    val, note = bench.pmc_traffic("bf16", 256)
    assert val is None and "stale" in note


@pytest.mark.parametrize("how", ["sigterm", "watchdog"])
def test_ranks_never_outlive_the_launcher(tmp_path, how):
    """The ranks run in their own session: a SIGTERM to the parent (a harness timeout, Ctrl-C) or the watchdog must take them down with it,
    or a hung collective keeps N processes on the GPUs.  The dry-run ranks hang on purpose and leave their pids behind."""
    import signal
    import time
    env = _env()
    env["LRCN_BENCH_DRYRUN_HANG"] = str(tmp_path / "pid")
    env["LRCN_BENCH_WATCHDOG_EXACT"] = "1"   # take --watchdog-s as given (by default it is raised to what the ladder's rungs may need)
    args = [sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"] + (["--watchdog-s", "8"] if how == "watchdog" else [])
    p = subprocess.Popen(args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids = []
    for _ in range(600):
        pids = [int(open(tmp_path / f).read()) for f in os.listdir(tmp_path) if open(tmp_path / f).read().strip()]
        if len(pids) == 2:
            break
        time.sleep(0.1)
    assert len(pids) == 2, "the ranks never started"
    if how == "sigterm":
        p.send_signal(signal.SIGTERM)
    p.communicate(timeout=60)
    assert p.returncode != 0
    for _ in range(100):
        alive = [q for q in pids if os.path.exists("/proc/%d" % q) and "zombie" not in open("/proc/%d/status" % q).read()]
        if not alive:
            break
        time.sleep(0.1)
    assert not alive, alive


def test_outer_watchdog_outlasts_the_ladder(monkeypatch):
    """the launcher's single outer watchdog must not fire while a fallback rung still has budget: n_rungs x (rung_s + teardown)
    + one stall window.  Checked on the arithmetic (run_ranks is replaced; nothing is launched)."""
    sys.path.insert(0, ROOT)
    import bench
    from lrcn_amd import launch as lch
    seen = {}
    monkeypatch.setattr(lch, "run_ranks", lambda script, argv, n, env, wd: (seen.setdefault("wd", wd), (0, ""))[1])
    monkeypatch.delenv("LRCN_BENCH_WATCHDOG_EXACT", raising=False)
    monkeypatch.setenv("LRCN_BENCH_RUNG_S", "900")
    monkeypatch.setenv("LRCN_BENCH_STALL_S", "300")
    for backend, n_rungs in (("torch", 2), ("auto", 3)):
        seen.clear()
        a = bench.parse_args(["--gpus", "2", "--dp-backend", backend])
        bench.launch(a, ["--gpus", "2", "--dp-backend", backend])
        assert len(lch.default_rungs(backend)) == n_rungs
        assert seen["wd"] >= n_rungs * 960 + 300, (backend, seen["wd"])
    seen.clear()
    a = bench.parse_args(["--gpus", "2", "--watchdog-s", "100000"])
    bench.launch(a, ["--gpus", "2"])
    assert seen["wd"] == 100000


def test_sub_reports_and_rungs_are_plain_functions():
    """roofline.sub is arithmetic on lrcn_profile_segment's accumulators (bench.sub_reports), and the ladder's rung list is data
    (launch.default_rungs): both checked without a GPU."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from lrcn_amd import launch
    segs = {"update": (2.0, 8, 8 * 1.116e9), "rec_fwd": (8.0, 16, 16 * 88e6), "rec_bwd": (0.0, 0, 0.0), "embed_gather": (0.16, 8, 8 * 12.3e6),
            "embed_grad": (0.72, 8, 8 * 54.8e6), "preprocess": (0.2, 8, 8 * 115.6e6), "upload": (5.6, 8, 8 * 38.5e6)}
    sub = bench.sub_reports(segs, 8)
    assert "recurrence_weight_stream_bwd" not in sub                      # a segment that never ran is not reported
    assert abs(sub["adam"]["GB/s"] - 4464.0) < 1.0 and sub["adam"]["peak_GB/s"] == 8000.0 and abs(sub["adam"]["frac_of_peak"] - 0.558) < 1e-3
    assert sub["upload"]["peak_GB/s"] == 64.0 and abs(sub["upload"]["GB/s"] - 55.0) < 0.1   # the upload is priced against PCIe, not HBM
    assert abs(sub["adam"]["ms_per_step"] - 0.25) < 1e-9 and abs(sub["adam"]["algorithmic_MB_per_step"] - 1116.0) < 1e-6
    assert "TFLOP/s" not in sub["recurrence_weight_stream_fwd"]
    # round 6: at 256 rows beside the capped convolution grids the recurrence is also priced as what it is -- contractions on the free CUs
    sub = bench.sub_reports(segs, 8, {"gflop_per_step": 45.056, "free_cus": 32})
    r = sub["recurrence_weight_stream_fwd"]
    assert abs(r["TFLOP/s"] - 45.1) < 0.1 and abs(r["frac_of_free_cu_mfma_peak"] - 45.056 / 314.5) < 1e-3 and "mfma on 32 free CUs" in r["bound_in_fact"]
    names = [r[0] for r in launch.default_rungs("torch")]
    assert names == ["default", "plain"] and [r[0] for r in launch.default_rungs("auto")] == ["abi", "default", "plain"]
    plain = dict(launch.default_rungs("abi"))["plain"]
    assert plain["LRCN_DP_BACKEND"] == "torch" and plain["LRCN_DP_BUCKETS"] == "0" and plain["LRCN_DP_SPARSE_EMBED"] == "0" and plain["LRCN_FUSED_UPDATE"] == "0"
    a = bench.parse_args(["--emulate-world", "8"])
    assert a.emulate_world == 8 and not a.replicated_update and bench.metric_name(a).startswith("EMULATED")
    assert bench.metric_name(bench.parse_args([])) == bench.HEADLINE_METRIC


def test_hw_sampler_without_the_sysfs_nodes_is_not_an_error(tmp_path):
    """bench.HwSampler (round 6: shader clock and socket power held in the timed region) reads amdgpu hwmon nodes; where they do not exist -- this
    container, another driver -- the line carries {"available": false} and nothing raises.  With a directory that looks like a hwmon node it
    samples on its thread and reports medians in MHz / W."""
    import time as _time
    sys.path.insert(0, ROOT)
    import bench
    h = bench.HwSampler(0)
    h.start()
    assert h.stop() == {"available": False}
    (tmp_path / "freq1_input").write_text("2150000000\n")
    (tmp_path / "power1_input").write_text("1300000000\n")
    (tmp_path / "power1_cap").write_text("1400000000\n")
    h = bench.HwSampler(0, period_s=0.001, hwmon_dir=str(tmp_path))
    _time.sleep(0.02)
    assert h.samples == []   # the thread exists but records nothing before start()
    h.start()
    _time.sleep(0.05)
    r = h.stop()
    assert r["available"] and r["sclk_mhz"]["median"] == 2150.0 and r["socket_power_w"]["median"] == 1300.0 and r["power_cap_w"] == 1400.0
    assert r["sclk_mhz"]["n"] >= 5

