"""The self-launcher of `bench.py --gpus N` (ada-mvs_amd/launch.py) on CPU: fresh child ranks over gloo.

What the reference has in this place is nn.DataParallel inside one process (reference predict_whu.py:82-83); one process per
GPU can hang in a collective, and the first real multi-GPU run must not sit until an outer timeout without a diagnostic.
"""
import io
import os
import subprocess
import sys
import time

from conftest import ROOT
import ada_mvs_amd  # noqa: F401
from ada_mvs_amd import launch

WORKER = [sys.executable, os.path.join(ROOT, "tests", "launch_worker.py")]


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:                                   # a zombie still answers kill(0)
        with open("/proc/%d/stat" % pid) as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _kept_logs(txt):
    """The directory a failed run keeps its ranks' stderr in (named in the report); removed here after a look inside."""
    import shutil
    d = txt.split("full stderr is kept in ")[1].split("\n")[0].strip()
    names = sorted(os.listdir(d))
    shutil.rmtree(d, ignore_errors=True)
    return names


def test_all_ranks_succeed_and_rank0_stdout_is_relayed():
    out, err = io.StringIO(), io.StringIO()
    code, data = launch.run_ranks(WORKER + ["ok"], 2, timeout=120, out=out, err=err)
    assert code == 0
    # (gloo itself writes a "[Gloo] Rank 0 is connected ..." line to stdout; RCCL does not)
    assert out.getvalue().strip().splitlines()[-1] == '{"ok": true, "world": 2}' and data.decode() == out.getvalue()


def test_a_rank_sleeping_outside_the_barrier_trips_the_deadline():
    """Rank 1 never joins: rank 0 sits in barrier().  The deadline ends BOTH by PID, the code is 124 and every rank's last
    stderr lines are relayed with how it ended."""
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    code, _ = launch.run_ranks(WORKER + ["hang"], 2, timeout=20, out=out, err=err)
    dt = time.monotonic() - t0
    assert code == launch.TIMEOUT_EXIT_CODE == 124
    assert 20 <= dt < 20 + 2 * launch.GRACE_S + 10
    txt = err.getvalue()
    assert "deadline of 20 s passed" in txt
    pids = []
    for r in (0, 1):
        assert "---- rank %d (pid " % r in txt and "rank %d of 2 is up" % r in txt
        pids.append(int(txt.split("---- rank %d (pid " % r)[1].split(")")[0]))
    assert txt.count("still running after 20 s: ended by the launcher") == 2
    assert not any(_pid_alive(p) for p in pids)
    assert '"ok"' not in out.getvalue()
    assert _kept_logs(txt) == ["rank0.stderr", "rank1.stderr"]


def test_a_launcher_told_to_stop_ends_its_ranks():
    """SIGTERM to the launcher itself (an outer `timeout`, a scheduler): the ranks are ended by PID, the code is 128 + 15."""
    prog = ("import sys, os; sys.path.insert(0, %r); import ada_mvs_amd; from ada_mvs_amd import launch;"
            "sys.exit(launch.run_ranks([sys.executable, %r, 'hang'], 2, timeout=300)[0])" % (ROOT, WORKER[1]))
    p = subprocess.Popen([sys.executable, "-c", prog], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    # wait until both ranks exist (children of the launcher), then tell the launcher to stop
    kids, t_end = [], time.monotonic() + 120
    while len(kids) < 2 and time.monotonic() < t_end:
        try:
            with open("/proc/%d/task/%d/children" % (p.pid, p.pid)) as f:
                kids = [int(x) for x in f.read().split()]
        except OSError:
            kids = []
        time.sleep(0.2)
    assert len(kids) == 2
    time.sleep(1.0)
    p.send_signal(15)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + 15
    assert not any(_pid_alive(k) for k in kids)
    assert "received signal 15" in err
    _kept_logs(err)


def test_a_failing_rank_ends_the_others_and_its_stderr_is_relayed():
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    code, _ = launch.run_ranks(WORKER + ["fail"], 3, timeout=300, out=out, err=err)
    assert code == 7 and time.monotonic() - t0 < 120
    txt = err.getvalue()
    assert "rank 1: simulated failure" in txt and "ended by the launcher after rank 1 failed" in txt
    _kept_logs(txt)


def test_bench_parses_launch_timeout_before_torch_and_relays_rank_failures():
    """`python bench.py --gpus 2 --launch-timeout 60` here (no GPU): both ranks fail on their own; the launcher exits
    non-zero -- not 124 -- within the deadline and says which rank said what."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ADAMVS_DIST_BACKEND"] = "gloo"
    # hide every device from the ranks: the failure must not depend on the host (a box with two GPUs would run real ranks)
    env["HIP_VISIBLE_DEVICES"] = env["ROCR_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-timeout=60", "--workload", "tiny"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 124)
    assert "---- rank 0 (pid " in r.stderr and "---- rank 1 (pid " in r.stderr
    _kept_logs(r.stderr)
