"""CPU: the process-wide host worker pool (csrc/kc_pool.h) hammered from several caller
threads while a third resizes it -- what two controller contexts on two threads and
kc_set_host_threads do in the host-trig fallback.  Plain build, and under ThreadSanitizer when the
toolchain has it."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "pool_stress.cpp"
INC = ROOT / "kompass-core_amd" / "csrc"


def _build(tmp_path, name, flags):
    exe = tmp_path / name
    cmd = ["g++", "-std=c++17", f"-I{INC}", str(SRC), "-o", str(exe), "-lpthread"] + flags
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return exe if p.returncode == 0 else None, p.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_pool_two_callers_and_a_resizer(tmp_path):
    exe, err = _build(tmp_path, "pool_stress", ["-O2"])
    assert exe is not None, err
    p = subprocess.run([str(exe), "20000", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert " 0 bad" in p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_pool_under_thread_sanitizer(tmp_path):
    exe, err = _build(tmp_path, "pool_stress_tsan", ["-O1", "-g", "-fsanitize=thread"])
    if exe is None:
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    p = subprocess.run([str(exe), "3000", "3"], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0, p.stdout + p.stderr
