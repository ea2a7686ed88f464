#!/usr/bin/env python3
"""bench.py with CPU stand-ins for the device (TEST INFRASTRUCTURE; tests/test_bench_launcher.py runs this file instead of bench.py).

The SAME launcher, per-rank supervisors and ladder (lrcn_amd/launch.py), rank set-up, sharding, trainer (dp.DataParallelTrainer), step-1
self-check, barrier + max-over-ranks timing and JSON assembly as `python bench.py --gpus N`, over gloo on the CPU, with
tests/dp_oracle_ops.py's stand-ins for the device operations and tiny dimensions: bench.rank_main -- everything that touches the GPU -- is
replaced by dryrun_rank_main below, and bench.SCRIPT by this file so that the ranks the launcher starts are dry runs too.  It proves
that `--gpus N` ends with one well-formed line with n_gpus = N, and that every rung of the ladder is reachable, before an N-GPU node
exists; it measures nothing.  Failure injection (environment):
    LRCN_BENCH_DRYRUN_FAIL_ABI=1             a rank whose rung selects the C-ABI communicator exits with an error
    LRCN_BENCH_DRYRUN_FAIL_RUNG=name[:rank]  that rank (default: every rank) exits with an error on the named rung
    LRCN_BENCH_DRYRUN_HANG_RUNG=name[:rank]  that rank stops making progress on the named rung (a hung collective)
    LRCN_BENCH_DRYRUN_BAD_SELFCHECK=name     on the named rung rank 1 perturbs a parameter before the self-check (replicas differ)
    LRCN_BENCH_DRYRUN_HANG_TEARDOWN=name[:rank]  that rank never returns from its teardown AFTER the line has been printed
    LRCN_BENCH_DRYRUN_NO_LINE=name           on the named rung rank 0 finishes with exit code 0 but never prints its line
    LRCN_BENCH_DRYRUN_HANG=path              every rank writes its pid to path.<rank> and sleeps (launcher kill tests)"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import bench  # noqa: E402


def _hit(spec, rung, rank):
    if not spec:
        return False
    name, _, r = spec.partition(":")
    return name == rung and (r == "" or int(r) == rank)


def dryrun_rank_main(a, world, rank, local_rank):
    import numpy as np
    import torch
    import torch.distributed as dist
    from lrcn_amd import dp
    from lrcn_amd.launch import beat
    import dp_oracle_ops as doo
    rung = os.environ.get("LRCN_BENCH_RUNG", "-")
    beat("started")
    if os.environ.get("LRCN_BENCH_DRYRUN_FAIL_ABI") and os.environ.get("LRCN_DP_BACKEND") == "abi":
        raise SystemExit("dry run: simulated failure of the C-ABI communicator")
    if _hit(os.environ.get("LRCN_BENCH_DRYRUN_FAIL_RUNG"), rung, rank):
        raise SystemExit("dry run: simulated failure of rank %d on rung %r" % (rank, rung))
    if os.environ.get("LRCN_BENCH_DRYRUN_HANG"):   # a rank stuck in a collective: the file named here receives its pid, then it sleeps
        with open(os.environ["LRCN_BENCH_DRYRUN_HANG"] + ".%d" % rank, "w") as f:
            f.write(str(os.getpid()))
        time.sleep(3600)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    beat("process group up")
    if _hit(os.environ.get("LRCN_BENCH_DRYRUN_HANG_RUNG"), rung, rank):
        time.sleep(3600)
    E = H = 16
    V, T, Bg = 37, 3, 8
    rows = dp.shard_rows(Bg, world * a.emulate_world, rank)
    param, optim, ops = doo.make(E, H, V, seed=42)
    trainer = dp.DataParallelTrainer(None, param, optim, Bg, world, rank, pdrop=0.0, seed=7, ops=ops, backend="torch")
    rng = np.random.default_rng(7)
    feats_g = (rng.standard_normal((Bg, 4096)) * 0.01).astype(np.float32)
    toks_g = rng.integers(3, V, size=(T, Bg)).astype(np.int32)
    feats, toks = torch.as_tensor(feats_g[rows]), toks_g[:, rows]
    strict = int(os.environ.get("LRCN_BENCH_RUNG_INDEX", "0")) + 1 < int(os.environ.get("LRCN_BENCH_RUNGS", "1"))
    selfcheck = None
    if world > 1:
        if os.environ.get("LRCN_BENCH_DRYRUN_BAD_SELFCHECK") == rung and rank == 1:
            param[8].add_(1e-3)
        selfcheck = trainer.self_check(feats, toks_g)
        bad = [k for k, ok in (("world", selfcheck["world_from_communicator"] == world == selfcheck["world_measured_by_allreduce"]),
                               ("loss", selfcheck["loss_rel_diff"] <= 1e-5), ("params", selfcheck["params_identical_before_step_1"])) if not ok]
        selfcheck["violations"] = bad
        if bad and strict:
            raise SystemExit("dry run: self-check failed on rung %r: %s" % (rung, bad))
        beat("self-check passed")

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        trainer.step(None, toks, feats=feats)
    barrier()
    beat("warm-up done")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step(None, toks, feats=feats)
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt_s = float(tt.item())
    beat("timed region done")
    if selfcheck is not None:
        selfcheck["params_identical_after_last_step"] = bool(trainer.check_replicas()[0])
    loss = trainer.loss_value()
    if rank == 0 and os.environ.get("LRCN_BENCH_DRYRUN_NO_LINE") != rung:
        print(json.dumps({"metric": "DRYRUN (CPU stand-ins, gloo) -- " + bench.metric_name(a), "value": Bg * a.steps / dt_s, "unit": "images/sec",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_s / a.steps, "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "dryrun",
                          "config": {"workload": "dry run of the launcher and the N-rank plumbing", "global_batch": Bg,
                                     "per_gpu_batch": rows.stop - rows.start, "parallelism": "dp%d" % world, "last_loss": loss},
                          "rccl": {"world": dist.get_world_size() if world > 1 else 1, "world_argv": a.gpus, "backend": "gloo-dryrun",
                                   "launched_by": os.environ.get("LRCN_BENCH_LAUNCHED", "0"), "selfcheck": selfcheck,
                                   "pipeline": trainer.describe(), "env_of_rung": {k: os.environ.get(k) for k in ("LRCN_DP_BUCKETS", "LRCN_DP_SPARSE_EMBED", "LRCN_DP_BACKEND")}}}),
              flush=True)
    if world > 1:
        dist.barrier()
    beat("finished")
    if _hit(os.environ.get("LRCN_BENCH_DRYRUN_HANG_TEARDOWN"), rung, rank):   # e.g. a destroy_process_group that never returns
        time.sleep(3600)
    if world > 1:
        dist.destroy_process_group()
    return 0


bench.rank_main = dryrun_rank_main
bench.SCRIPT = os.path.abspath(__file__)

if __name__ == "__main__":
    sys.exit(bench.main())
