"""bench.py's launcher and N-rank plumbing, without a GPU: `python bench.py --gpus 2` with no WORLD_SIZE must start its
ranks itself -- from a parent that never touches the device -- and relay ONE well-formed JSON line with n_gpus = 2; the torchrun form the
driver uses for N > 1 must keep working.  LRCN_BENCH_DRYRUN=1 swaps the device operations for tests/dp_oracle_ops.py over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    env["LRCN_BENCH_DRYRUN"] = "1"
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


def test_failed_job_is_reported_and_auto_backend_retries(tmp_path):
    """The launcher returns the job's failure (no line), and --dp-backend auto reruns with torch after a failed abi attempt: the dry run
    fails on purpose when LRCN_DP_BACKEND=abi and LRCN_BENCH_DRYRUN_FAIL_ABI=1."""
    env = _env()
    env["LRCN_BENCH_DRYRUN_FAIL_ABI"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--dp-backend", "abi"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--dp-backend", "auto"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and "attempt 2 of 2" in d["rccl"]["launcher"]


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
