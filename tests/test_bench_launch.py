"""`python bench.py --gpus N` must start its own N rank processes (the driver's command shape) BEFORE
the parent touches the GPU, relay rank 0's single JSON line and propagate a failing rank's return
code.  --dry-launch: the ranks report what they were started with and do no GPU work (runs on CPU)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench(*extra, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra], capture_output=True, text=True,
                          timeout=280, env=e)


@pytest.mark.timeout(300)
def test_gpus_2_launches_two_fresh_ranks_and_relays_one_json_line():
    p = _bench("--gpus", "2", "--dry-launch")
    assert p.returncode == 0, p.stderr[-800:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout          # exactly ONE line on stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2
    ranks = sorted(out["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert all(r["world"] == 2 and r["master"] == "127.0.0.1" for r in ranks)
    assert len({r["pid"] for r in ranks}) == 2 and len({r["ppid"] for r in ranks}) == 1
    # the parent never mapped the HIP runtime or the library: its children are fresh processes, nothing that has
    # initialised the GPU is forked or replaced
    assert "launcher_loaded_hip_library=False" in p.stderr, p.stderr[-400:]


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    p = _bench("--gpus", "2", "--dry-launch", "--dry-fail-rank", "1")
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(300)
def test_under_an_external_launcher_the_script_does_not_launch_again():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
                        "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=280, env=e)
    assert p.returncode == 0, p.stderr[-800:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(out["ranks"]) == 2 and "launcher pid" not in p.stderr


@pytest.mark.timeout(300)
def test_a_rank_dying_after_rendezvous_ends_the_launcher_within_its_limit():
    """a rank that dies BEHIND the rendezvous leaves its peers waiting; the launcher's
    wall-clock limit terminates the child process group and the script returns non-zero -- it does not hang."""
    import time

    t0 = time.time()
    p = _bench("--gpus", "2", "--dry-launch", "--dry-die-after-rendezvous", "1", env={"KC_BENCH_LAUNCH_LIMIT_S": "25"})
    assert p.returncode != 0
    assert time.time() - t0 < 120
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
