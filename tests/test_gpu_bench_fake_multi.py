"""GPU: `bench.py --gpus 2` end to end on ONE GPU (LRCN_BENCH_FAKE_MULTI=1: both ranks share device 0 and the process group is gloo -- RCCL
refuses two ranks per device), i.e. the launcher, the per-rank supervisors of the first-contact ladder, the N-rank control flow of dp.py
(row shards, global normaliser, per-group [event -> all-reduce -> fused Adam] pipeline on a probed update stream, sparse exchange of the
embedding gradient) and the step-1 self-check, all on the REAL kernels with only the transport swapped.  A validation, never a measurement:
the line says so in its metric.  (tests/test_bench_launcher.py drives every rung of the ladder with injected failures on the CPU.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_walk_the_ladder_and_pass_the_self_check():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["LRCN_BENCH_FAKE_MULTI"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("VALIDATION") and d["n_gpus"] == 2 and d["value"] > 0
    rc = d["rccl"]
    assert rc["world"] == 2 and rc["mode"] == "default" and rc["rung"] == "1 of 2" and rc["fallback_reason"] is None
    assert rc["pipeline"]["per_group_pipeline"] and rc["pipeline"]["sparse_embedding_exchange"] and rc["pipeline"]["update_stream_on_its_own_queue"] is not None
    sc = rc["selfcheck"]
    assert sc["violations"] == []
    assert sc["world_from_communicator"] == 2 == sc["world_measured_by_allreduce"]
    assert sc["loss_rel_diff"] <= 1e-5 and sc["sparse_vs_dense_embed_grad_rel"] <= 1e-4
    assert sc["params_identical_before_step_1"] and sc["params_identical_after_warmup"] and sc["params_identical_after_last_step"]
    assert 3.0 < d["config"]["last_loss"] < 9.3   # below ln V = 9.27 after six steps on the two alternating batches, and finite
