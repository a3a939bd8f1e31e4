"""Staged hand-off of the host's trig table (round 3): a large table (cfg3: 257 rows x 100 steps) is produced in
stages of consecutive rows, every stage dealt over all host workers; the worker that completes a stage publishes
16 seq + (stages done) in the sequence word and a workgroup of the three-kernel roll-out -- which takes its samples
in row order -- waits only for the stage of its highest row.  Same table, same poses: cycles with stages on, off
and with other stage sizes are bit-equal to the oracle, cycle after cycle on one context (the sequence word of one
table must never pass for the next one's), and a late host still fails the cycle instead of hanging it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path[:0] = [%(root)r, %(root)r + "/kompass-core_amd", %(root)r + "/tests"]
import kompass_hip as kh, synthetic as syn
from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle_mt
inp = syn.make_controller_inputs("cfg3", seed=1, scale=0.7, scene="mid")   # 90 x 179 samples: 504 workgroups, 2 rounds
assert len(inp["vx"]) > 8192 and len(set(np.round(inp["omega"], 12))) * inp["P"] >= 16384
ctx = hip_context(kh, inp)
for k in range(4):                                   # new pose = new table, every cycle
    st = (0.05 * k, -0.02 * k, 0.3 * k - 0.4, 0.0)
    cur = dict(inp, state=st, seg_xyz=inp["seg_xyz"])
    o = oracle_cycle_mt(cur)
    h = hip_cycle(kh, cur, ctx=ctx)
    assert h["res"]["n_admissible"] == len(o["raw"]), (k, h["res"]["n_admissible"], len(o["raw"]))
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["px"].view(np.uint32), o["px"].view(np.uint32))
    np.testing.assert_array_equal(h["py"].view(np.uint32), o["py"].view(np.uint32))
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    assert h["res"]["index"] == o["index"]
    assert ctx.get_option("last_cycle_single_launch") == 0.0
ctx.close()
print("OK")
"""


@pytest.mark.parametrize("env", [dict(), dict(KC_TRIG_STAGES="0"), dict(KC_TRIG_STAGE_MIN="1"), dict(KC_TRIG_STAGE_MIN="3", KC_HOST_THREADS="5"),
                                 dict(KC_HOST_THREADS="1")])
def test_staged_table_cycles_match_the_oracle(env):
    e = dict(os.environ, KC_DEVICE_TRIG="0", **env)   # (the host table is what is staged)
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-3000:]


def test_a_late_host_fails_the_cycle_of_a_staged_table():
    code = r"""
import sys
sys.path[:0] = [%(root)r, %(root)r + "/kompass-core_amd", %(root)r + "/tests"]
import kompass_hip as kh, synthetic as syn
from helpers import hip_context
inp = syn.make_controller_inputs("cfg3", seed=1, scale=0.7, scene="mid")
ctx = hip_context(kh, inp)
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
try:
    ctx.cycle(inp["state"], inp["P"])
    print("NO ERROR")
except kh.KompassHipError as e:
    print("ERR", e)
r = ctx.cycle(inp["state"], inp["P"])          # the context is usable at once
print("THEN", r.n_admissible > 0)
ctx.close()
""" % dict(root=ROOT)
    e = dict(os.environ, KC_TEST_LATE_FLAG_MS="120", KC_DEVICE_TRIG="0")
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "ERR" in p.stdout and "gave up waiting" in p.stdout and "THEN True" in p.stdout, p.stdout
