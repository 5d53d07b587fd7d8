"""Start the N ranks of a multi-GPU run from a plain `python bench.py --gpus N` (one process per GPU, torch.distributed.run).

The reference has no multi-GPU path and no launcher (kmc_main.cpp:35-37 selects one device).  The parent never touches the GPU:
a process that has initialised HIP must not exec or fork GPU children on this pool, so the ranks are a child `python -m
torch.distributed.run` started before anything imports torch.cuda.  Rank 0's JSON line is relayed as the LAST line of stdout;
everything else the ranks print goes to stderr; the parent exits with the child's code.
"""
import json
import os
import socket
import subprocess
import sys


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_command(script, argv, nranks, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def run_ranks(script, argv, nranks, timeout=None) -> int:
    """Returns the exit code of the rank group (non-zero if any rank failed, 6 if no JSON line came back)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / cross-process device memory need it on this host driver
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    cmd = rank_command(script, argv, nranks, free_port())
    # the rank group runs in its own session: on a time-out the WHOLE group is ended (SIGTERM, then SIGKILL), not just the launcher --
    # orphaned ranks would keep holding the GPU while the caller reports a failure
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
        r = subprocess.CompletedProcess(cmd, proc.returncode, out, None)
    except subprocess.TimeoutExpired:
        sys.stderr.write("launch: the rank group did not finish within %s s -- ending process group %d\n" % (timeout, proc.pid))
        out = ""
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(proc.pid, sig)          # start_new_session: the child's pid is its process-group id
            except ProcessLookupError:
                break
            try:
                out, _ = proc.communicate(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        r = subprocess.CompletedProcess(cmd, 3, out or "", None)
    line = None
    for ln in (r.stdout or "").splitlines():
        s = ln.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s); line = s
                continue
            except ValueError:
                pass
        sys.stderr.write(ln + "\n")
    if line is None:
        print(json.dumps({"metric": "KMC steps/sec", "value": None, "n_gpus": nranks, "error": "no JSON line from rank 0 (exit code %d)" % r.returncode}), flush=True)
        return r.returncode or 6
    print(line, flush=True)
    return r.returncode
