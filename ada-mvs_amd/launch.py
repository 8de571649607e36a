"""One process per GPU, started as fresh children of a process that has not touched the GPU.

`bench.py --gpus N` without a launcher around it becomes the launcher through `run_ranks`: N children with torchrun's
environment contract (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT).  What the
reference has in this place is `nn.DataParallel` inside one process (reference predict_whu.py:82-83); a rank per GPU has
failure modes that has not, and this module is where they are handled:

  * a rank that EXITS non-zero leaves the others inside a collective: they are ended by PID and the launcher returns that rank's
    code;
  * a rank that HANGS (a collective that never completes, a GPU that stopped answering) would leave the run sitting until an
    outer timeout with no diagnostic: after `timeout` seconds every rank still alive is ended by PID -- SIGTERM, then SIGKILL after
    a grace period -- and the launcher returns 124 (the code of coreutils' `timeout`);
  * the LAUNCHER is told to stop (SIGTERM / SIGHUP / SIGINT from an outer `timeout`, a scheduler or `gpurun`'s own limit): the
    ranks are no process group of their own and would be orphaned inside a collective, holding the GPUs: the signal is caught,
    every rank is ended by PID and the launcher returns 128 + signal;
  * ranks other than 0 have no stdout of interest, but their stderr is all there is when they fail: every rank's stderr goes to
    a file of its own and its last lines are relayed on any failure; the files themselves are KEPT after a failed run (their
    directory is named in the report) and removed after a good one.

Never an exec: the children are fresh processes (`subprocess.Popen`), the launcher itself imports nothing that initialises the
GPU (this module imports no torch).
"""
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

DEFAULT_TIMEOUT_S = 900.0
TIMEOUT_EXIT_CODE = 124
GRACE_S = 5.0


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None, extra=None):
    """torchrun's environment for one rank of a single-node world."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver of this pool only supports dmabuf IPC
    if extra:
        env.update(extra)
    return env


def _tail(path, lines):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - 16384))
            txt = f.read().decode("utf-8", "replace")
    except OSError:
        return ""
    return "\n".join(txt.splitlines()[-lines:])


def _end(procs, grace=GRACE_S):
    """End the given ranks by PID: SIGTERM, and SIGKILL for whatever is still alive after `grace` seconds."""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.monotonic() + grace
    while time.monotonic() < t_end and any(p.poll() is None for p in procs):
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
    for p in procs:
        try:
            p.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            pass


def run_ranks(cmd, world, timeout=DEFAULT_TIMEOUT_S, extra_env=None, tail_lines=25, out=None, err=None, poll_s=0.05):
    """Start `world` fresh processes of `cmd` (a list), one rank each, and wait for them.

    Returns (code, stdout of rank 0 as bytes).  code = 0 when every rank exits 0; the first failing rank's code when one
    exits non-zero (the others are ended by PID); TIMEOUT_EXIT_CODE when the deadline passes with ranks still alive (all
    ended by PID).  On any failure the last `tail_lines` lines of EVERY rank's stderr are written to `err` (default
    sys.stderr), each under a header that names the rank, its pid and how it ended.  Rank 0's stdout is relayed to `out`
    (default sys.stdout) after the run; the other ranks' stdout is discarded."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    tmp = tempfile.mkdtemp(prefix="adamvs_ranks_")
    procs, logs, files = [], [], []
    code = 1                                                   # what `finally` sees if anything below raises
    told = []                                                  # signals the launcher itself received
    old_handlers = {}
    if threading.current_thread() is threading.main_thread():  # signal.signal works on the main thread only
        for sig in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
            try:
                old_handlers[sig] = signal.signal(sig, lambda n, _f: told.append(n))
            except (OSError, ValueError):
                pass
    try:
        for r in range(world):
            path = os.path.join(tmp, "rank%d.stderr" % r)
            f = open(path, "wb")
            files.append(f)
            logs.append(path)
            procs.append(subprocess.Popen(list(cmd), env=rank_env(r, world, port, extra=extra_env),
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=f))
        out0 = []
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = time.monotonic() + timeout if timeout and timeout > 0 else None
        code, how = 0, {}
        pending = list(range(world))
        while pending:
            if told:
                code = 128 + told[0]
                for q in pending:
                    how[q] = "ended by the launcher, which received signal %d" % told[0]
                _end([procs[q] for q in pending])
                pending = []
                break
            for r in list(pending):
                c = procs[r].poll()
                if c is None:
                    continue
                pending.remove(r)
                how[r] = "exit code %d" % c
                if c != 0 and code == 0:
                    code = c if c > 0 else 128 - c             # a signal's negative code, the shell's way
                    for q in pending:
                        how[q] = "ended by the launcher after rank %d failed" % r
                    _end([procs[q] for q in pending])
                    pending = []
                    break
            if pending and deadline is not None and time.monotonic() > deadline:
                code = TIMEOUT_EXIT_CODE
                for q in pending:
                    how[q] = "still running after %.0f s: ended by the launcher" % timeout
                _end([procs[q] for q in pending])
                pending = []
            if pending:
                time.sleep(poll_s)
        reader.join(10)
        for f in files:
            f.close()
        if code != 0:
            err.write("launcher: %d rank(s), exit code %d%s; every rank's full stderr is kept in %s\n" % (
                world, code, " (deadline of %.0f s passed: --launch-timeout)" % timeout if code == TIMEOUT_EXIT_CODE else "", tmp))
            for r in range(world):
                err.write("---- rank %d (pid %d): %s; last stderr lines:\n%s\n" % (
                    r, procs[r].pid, how.get(r, "exit code %s" % procs[r].returncode), _tail(logs[r], tail_lines) or "(none)"))
            err.flush()
        else:                                                  # warnings of a good run are still worth seeing (rank 0's only)
            t = _tail(logs[0], tail_lines)
            if t:
                err.write(t + "\n")
        data = b"".join(out0)
        if out is not None:
            try:
                out.write(data.decode("utf-8", "replace"))
                out.flush()
            except (OSError, ValueError):
                pass
        return code, data
    finally:
        _end([p for p in procs if p.poll() is None], grace=1.0)
        for sig, h in old_handlers.items():
            try:
                signal.signal(sig, h)
            except (OSError, ValueError):
                pass
        for f in files:
            if not f.closed:
                f.close()
        if code == 0:                                          # a failed run keeps its logs (the report names the directory)
            for path in logs:
                try:
                    os.remove(path)
                except OSError:
                    pass
            try:
                os.rmdir(tmp)
            except OSError:
                pass
