"""Start the N ranks of a one-node job from a parent that never touches the GPU.

A process that has initialised the device cannot safely be replaced or forked, so `--gpus N` programs (bench.py, tools/lrcn.py) start
their ranks as a CHILD job -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port
<free> script.py <same flags>`, one process per GPU over RCCL -- wait for it under a watchdog and pass its output and exit code on.
This module imports neither torch nor the HIP library.  The ranks run in their own session; every way out of the wait (watchdog,
SIGTERM / SIGINT to the parent, any exception) ends exactly that process group, so no rank is left holding a GPU.

Round 5 -- the LADDER (`supervise_rank`).  A multi-GPU job that fails or hangs on its first contact with real links must still end
with a number, and a process that has touched the GPU can neither be re-executed nor trusted to recover.  So every rank process that
torchrun (or the driver) starts is a GPU-free SUPERVISOR: it runs the real rank as a fresh CHILD process, rung after rung --
"default" (the full pipeline) -> "plain" (one all-reduce + one replicated Adam, no sparse exchange, no queue probes, one VGG forward
per step) -- each rung under a stall watchdog fed by the child's heartbeat; the supervisors of one job agree on a rung's outcome
through a rendezvous directory on the node's /tmp (one node only, as the contract says; no torch, no sockets), stop their children as
soon as ANY rank of the rung has failed, and move on together.  The rung that produced the number is named in the line
(`rccl.mode`, `rccl.fallback_reason`)."""
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time


def find_free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_ranks(script, argv, nranks, env_extra=None, watchdog_s=900.0, capture=True):
    """-> (return code, captured stdout or None).  124 = stopped by the watchdog."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(find_free_port()), os.path.abspath(script)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if capture else None, stderr=None, text=True, start_new_session=True)

    def stop_job():
        if proc.poll() is not None:
            return
        try:
            os.killpg(proc.pid, signal.SIGTERM)   # the exact process group started above (its own session), nothing else
            for _ in range(30):
                if proc.poll() is not None:
                    break
                time.sleep(0.1)
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, _frame):
        stop_job()
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        try:
            out, _ = proc.communicate(timeout=watchdog_s)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            stop_job()
            out, _ = proc.communicate()
            rc = 124
    finally:
        stop_job()   # no-op when the job has ended
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc, out


# ----------------------------------------------------------------------------------------------------------------- the ladder
PLAIN_ENV = {  # the most conservative N-rank step this repo has: what a first contact falls back to
    "LRCN_DP_BACKEND": "torch", "LRCN_DP_SHARD_ADAM": "0", "LRCN_DP_SPARSE_EMBED": "0", "LRCN_DP_BUCKETS": "0", "LRCN_DP_GROUP_ADAM": "0",
    "LRCN_DP_QUEUE_PROBE": "0", "LRCN_DP_WG_STREAM_PROBE": "0", "LRCN_FUSED_UPDATE": "0", "LRCN_VGG_CHUNK_IMAGES": "0",
}


def default_rungs(dp_backend="torch"):
    """[(name, env)] in the order they are tried.  'auto' puts the C-ABI communicator first."""
    rungs = []
    if dp_backend == "auto":
        rungs.append(("abi", {"LRCN_DP_BACKEND": "abi"}))
    rungs.append(("default", {"LRCN_DP_BACKEND": "abi" if dp_backend == "abi" else "torch"}))
    rungs.append(("plain", dict(PLAIN_ENV)))
    return rungs


def _set_pdeathsig():
    """preexec of a supervised child: die (SIGKILL) with the supervisor, however the supervisor ends -- a supervisor that torchrun
    SIGKILLs cannot run a handler, and an orphaned rank would keep its GPU."""
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)   # PR_SET_PDEATHSIG
    except Exception:
        pass


def _rendezvous_dir(world):
    """One directory per job on this node: keyed by the process that started the supervisors (torchrun's agent is the parent of every
    rank) and the job's master port."""
    key = "lrcn_ladder_%s_%s_%d_w%d" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MASTER_PORT", "0"), os.getppid(), world)
    d = os.path.join(os.environ.get("LRCN_LADDER_DIR", tempfile.gettempdir()), key)
    os.makedirs(d, exist_ok=True)
    return d


def _write_atomic(path, text):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def beat(stage):
    """Called by the supervised child at every stage it completes: the supervisor's stall watchdog measures from the last beat."""
    path = os.environ.get("LRCN_BENCH_HEARTBEAT")
    if path:
        try:
            with open(path, "a") as f:
                f.write("%.3f %s\n" % (time.time(), stage))
        except OSError:
            pass


def supervise_rank(script, argv, rungs, stall_s=300.0, rung_s=900.0, is_line=None, annotate=None):
    """This process = one rank as torchrun started it (RANK / WORLD_SIZE / MASTER_* in the environment); it never touches the GPU.
    Runs `script argv` as a child per rung (env: the rung's knobs, LRCN_BENCH_CHILD=1, LRCN_BENCH_RUNG=name, a MASTER_PORT of the rung's
    own, LRCN_BENCH_HEARTBEAT) until one rung ends with rc 0 on EVERY rank; rank 0 then prints the child's JSON line, annotated with the
    rung's name and the reasons the earlier rungs were left.  -> exit code (0, or the last rung's)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    rdv = _rendezvous_dir(world)
    child = [None]

    def stop_child():
        p = child[0]
        if p is None or p.poll() is not None:
            return
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)   # the child's own session (started below), nothing else
            except (ProcessLookupError, PermissionError):
                return
            for _ in range(20):
                if p.poll() is not None:
                    return
                time.sleep(0.1)

    def on_signal(signum, _frame):
        stop_child()
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    reasons = []
    rc_final = 1
    try:
        for k, (name, renv) in enumerate(rungs):
            # a fresh rendezvous port per rung: rank 0 picks it, the others read it (the agent's store keeps the dead rung's keys)
            port_file = os.path.join(rdv, "rung%d.port" % k)
            if rank == 0:
                _write_atomic(port_file, str(find_free_port()))
            t0 = time.time()
            while _read(port_file) is None:
                if time.time() - t0 > 120:
                    raise SystemExit("lrcn ladder: rank %d never saw rung %d's port file in %s" % (rank, k, rdv))
                time.sleep(0.05)
            env = dict(os.environ)
            env.update(renv)
            hb = os.path.join(rdv, "rung%d.rank%d.beat" % (k, rank))
            env.update({"LRCN_BENCH_CHILD": "1", "LRCN_BENCH_RUNG": name, "LRCN_BENCH_RUNG_INDEX": str(k), "LRCN_BENCH_RUNGS": str(len(rungs)),
                        "MASTER_PORT": _read(port_file).strip(), "LRCN_BENCH_HEARTBEAT": hb, "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)   # rank 0 of the rung hosts the rung's own store
            _write_atomic(hb, "%.3f spawned\n" % time.time())
            p = subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env, stdout=subprocess.PIPE if rank == 0 else None,
                                 stderr=None, text=True, start_new_session=True, preexec_fn=_set_pdeathsig)
            child[0] = p
            out_chunks = []
            if rank == 0:   # drain the pipe without blocking the watchdog loop
                import threading
                th = threading.Thread(target=lambda: out_chunks.append(p.stdout.read()), daemon=True)
                th.start()
            why = None
            finished_but_stuck = False
            t_start = time.time()
            while p.poll() is None:
                time.sleep(0.05)
                now = time.time()
                last_stage = ((_read(hb) or "").strip().splitlines() or [""])[-1]
                if last_stage.endswith(" finished") and now - os.path.getmtime(hb) > float(os.environ.get("LRCN_BENCH_TEARDOWN_S", "20")):
                    # the rank has measured, reduced and (rank 0) printed; it hangs in its teardown (destroy_process_group, a library's atexit):
                    # the number stands, the process does not
                    finished_but_stuck = True
                    stop_child()
                    break
                peers = [r for r in range(world) if r != rank and (_read(os.path.join(rdv, "rung%d.rank%d.rc" % (k, r))) or "0").split()[0] != "0"]
                if peers:
                    why = "rank %s failed in this rung" % ",".join(map(str, peers))
                elif now - os.path.getmtime(hb) > stall_s:
                    last = (_read(hb) or "").strip().splitlines()[-1:]
                    why = "no progress for %.0f s after stage %r" % (stall_s, " ".join(last[0].split()[1:]) if last else "?")
                elif now - t_start > rung_s:
                    why = "not finished after %.0f s" % rung_s
                if why:
                    stop_child()
                    break
            rc = p.wait()
            if rank == 0:
                th.join(timeout=5)
            if finished_but_stuck:
                rc = 0
            mine = 0 if (rc == 0 and why is None) else (rc if rc not in (0, None) else 124)
            out = "".join(out_chunks)
            lines = [ln for ln in out.splitlines() if is_line(ln)] if (rank == 0 and is_line) else []
            if rank == 0 and is_line is not None and mine == 0 and not lines:
                # a rung that exits 0 everywhere WITHOUT its line has failed, and every supervisor must learn so from the same place: the
                # verdict below is read from the rc files alone, so rank 0 publishes the missing line as its own non-zero code before any
                # rank reads (ranks 1..N-1 used to leave with rc 0 while rank 0 went on to the next rung alone)
                mine, why = 65, "exited 0 but printed no result line"
            _write_atomic(os.path.join(rdv, "rung%d.rank%d.rc" % (k, rank)), "%d %s" % (mine, why or ("exit code %d" % rc if rc else "ok")))
            # the rung's verdict: every rank's code (a rank that stopped its child because a peer failed reports so)
            t0 = time.time()
            codes = {}
            while len(codes) < world:
                for r in range(world):
                    txt = _read(os.path.join(rdv, "rung%d.rank%d.rc" % (k, r)))
                    if txt is not None:
                        codes[r] = txt
                if time.time() - t0 > stall_s + 60:
                    break
                time.sleep(0.05)
            bad = {r: c for r, c in codes.items() if c.split()[0] != "0"}
            missing = [r for r in range(world) if r not in codes]
            if not bad and not missing:
                if rank == 0:
                    line = lines[-1] if lines else ""
                    if annotate and line:
                        line = annotate(line, name, k, len(rungs), reasons)
                    sys.stdout.write((line + "\n") if line else out)
                    sys.stdout.flush()
                rc_final = 0
                break
            first = sorted(bad.items())[0] if bad else (missing[0], "? no verdict")
            cause = [c for r, c in sorted(bad.items()) if "failed in this rung" not in c] or [first[1]]
            reasons.append("%s: rank %d %s" % (name, first[0], " ".join(cause[0].split()[1:]) or cause[0]))
            if rank == 0:
                print("lrcn ladder: rung %r failed (%s)%s" % (name, reasons[-1], "; trying %r" % rungs[k + 1][0] if k + 1 < len(rungs) else "; giving up"),
                      file=sys.stderr, flush=True)
                if out.strip() and not lines:
                    sys.stderr.write(out[-2000:])
            rc_final = mine or 1
        if rc_final != 0:
            # torchrun stops every worker as soon as one exits non-zero: rank 0 (which has just reported why) leaves first
            done = os.path.join(rdv, "gave_up")
            if rank == 0:
                _write_atomic(done, "1")
            else:
                t0 = time.time()
                while _read(done) is None and time.time() - t0 < 10:
                    time.sleep(0.05)
                time.sleep(0.2)
    finally:
        stop_child()
        for sg, h in old.items():
            signal.signal(sg, h)
        # the rendezvous directory goes when every supervisor has left it (rank 0 waits briefly for the others' exit marks)
        try:
            _write_atomic(os.path.join(rdv, "exit.rank%d" % rank), "1")
            if rank == 0:
                t0 = time.time()
                while time.time() - t0 < 3 and any(_read(os.path.join(rdv, "exit.rank%d" % r)) is None for r in range(world)):
                    time.sleep(0.05)
                import shutil
                shutil.rmtree(rdv, ignore_errors=True)
        except OSError:
            pass
    return rc_final
