"""CPU: the vector forms of kc_dwa_set_scan's per-beam host loops (csrc/kc_scan_tables.h) against their scalar
definitions, bit for bit (tests/native/scan_tables.cpp)."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "native" / "scan_tables.cpp"
INC = ROOT / "kompass-core_amd" / "csrc"


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_vector_forms_equal_scalar_forms(tmp_path):
    exe = tmp_path / "scan_tables"
    p = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", f"-I{INC}", str(SRC), "-o", str(exe)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 bad" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_reused_window_lattice_equals_a_fresh_one(tmp_path):
    """hm::build_window_lattice keeps and refills its index tables in place (tests/native/window_lattice.cpp)."""
    exe = tmp_path / "window_lattice"
    p = subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", f"-I{INC}", f"-I{ROOT / 'include'}",
                        str(ROOT / "tests" / "native" / "window_lattice.cpp"), "-o", str(exe)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 bad" in r.stdout
