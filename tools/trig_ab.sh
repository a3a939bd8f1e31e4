#!/bin/bash
# usage (on the GPU box, from the repo root): tools/trig_ab.sh > out.txt
# Same-box A/B of device trig against the host libm table: resident cycles of cfg2 / cfg3 / cfg5
# (tools/opt_ab.py: one process, alternating contexts) and, alternating processes, the fresh-input step and the
# class-level cycle with a point cloud and with LaserScan input.
for c in cfg2 cfg3 cfg5; do python tools/opt_ab.py device_trig=0 $c 2>&1 | sed 's/^/[resident, base = device trig] /'; done
for r in 1 2; do for v in 1 0; do
  echo "== round $r KC_DEVICE_TRIG=$v"
  KC_DEVICE_TRIG=$v python tools/fresh_breakdown.py 2>&1 | tail -5
  KC_DEVICE_TRIG=$v python tools/class_cycle.py 2>&1 | tail -3
  KC_DEVICE_TRIG=$v python tools/class_cycle_scan.py 2>&1 | tail -3
done; done
