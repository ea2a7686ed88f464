"""Start the N ranks of a one-node job from a parent that never touches the GPU.

A process that has initialised the device cannot safely be replaced or forked, so `--gpus N` programs (bench.py, tools/lrcn.py) start
their ranks as a CHILD job -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port
<free> script.py <same flags>`, one process per GPU over RCCL -- wait for it under a watchdog and pass its output and exit code on.
This module imports neither torch nor the HIP library.  The ranks run in their own session; every way out of the wait (watchdog,
SIGTERM / SIGINT to the parent, any exception) ends exactly that process group, so no rank is left holding a GPU."""
import os
import signal
import socket
import subprocess
import sys
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
