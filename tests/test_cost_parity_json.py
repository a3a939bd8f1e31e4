"""Row J1: the reference's CPU/GPU cost-parity harness (COST_PARITY_JSON dumps of
cost_evaluator_test + tests/test_cost_parity.py:132-188) applied to this build:
"cpu" = the oracle, "hip" = kc_cost_evaluate on the MI355X.  The reference
accepts 1e-4 relative drift between its two backends; here the twelve cases must
agree bit for bit, and all twelve known answers hold on the device."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cost_parity as cp

GOLD = json.loads((Path(__file__).parent / "golden" / "cost_kat.json").read_text())


def _check_schema(doc, backend):
    assert doc["schema_version"] == 1 and doc["backend"] == backend
    assert set(doc["tests"]) == set(cp.CASES)
    for name, rec in doc["tests"].items():
        assert list(rec) == ["costs"] and len(rec["costs"]) == len(cp.CASES[name])


def test_cpu_dump_has_the_reference_schema_and_the_known_answers(tmp_path):
    out = tmp_path / "cpu.json"
    env = dict(os.environ, COST_PARITY_JSON=str(out))
    subprocess.run([sys.executable, str(Path(cp.__file__)), "--backend", "cpu"], check=True, env=env, timeout=120)
    doc = json.loads(out.read_text())
    _check_schema(doc, "cpu")
    _known_answers(doc)
    # no variable, no file (cost_evaluator_test.cpp:186-189)
    env.pop("COST_PARITY_JSON")
    subprocess.run([sys.executable, str(Path(cp.__file__)), "--backend", "cpu"], check=True, env=env, timeout=120,
                   cwd=tmp_path)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["cpu.json"]


def _known_answers(doc):
    for name, g in GOLD["cost"].items():
        got = doc["tests"][name]["costs"]
        if "expected" in g:
            for v, e in zip(got, g["expected"]):
                assert abs(v) <= 1e-12 if e == 0.0 else abs(v - e) <= g["tol"] * min(abs(v), abs(e)), (name, v, e)
    # goal_cost_arc_remaining_on_curved_path: closed forms of cost_evaluator_test.cpp:244-262
    R, total = 2.0, cp._reference_path(cp._circle34(2.0, 60, 0.05, 20.0))["total"]
    follow, chord = doc["tests"]["goal_cost_arc_remaining_on_curved_path"]["costs"]
    tol = GOLD["cost"]["goal_cost_arc_remaining_on_curved_path"]["rel_tol"]
    assert abs(follow - (total - R * 0.5) / total) <= tol * follow
    assert abs(chord - (1.0 + np.sqrt(0.5) / total)) <= tol * chord
    assert follow < chord


@pytest.mark.gpu
def test_hip_dump_equals_cpu_dump(tmp_path):
    cpu = cp.dump("cpu", tmp_path / "cpu.json")
    hip = cp.dump("hip", tmp_path / "hip.json")
    _check_schema(json.loads((tmp_path / "hip.json").read_text()), "hip")
    _known_answers(hip)                      # all twelve known answers on the device
    rows, failures = cp.compare(cpu, hip)    # the reference's comparison, relative 1e-4
    assert not failures, failures
    for name, i, c, g, _, _ in rows:         # and this build's bar: the same float
        assert np.float32(c) == np.float32(g), (name, i, c, g)
    # the label the reference's own script asks of a device build
    assert cp.dump("hip", tmp_path / "gpu.json", label="gpu")["backend"] == "gpu"
    assert subprocess.run([sys.executable, str(Path(cp.__file__)), "--compare", str(tmp_path / "cpu.json"),
                           str(tmp_path / "gpu.json")], timeout=120).returncode == 0
