"""Plain scan time by beam direction: x-major lines write along the columns of the column-major grid, y-major
lines one cache line per cell.  python tools/mapper_dir_time.py [beams] [side]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
side = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m = kh.MapperContext(side, side, 0.05, (0, 0, 0), 0.0, n)
m.timing_enable(True) if hasattr(m, "timing_enable") else None
for name, lo, hi in (("x-major (+-40 deg)", -0.7, 0.7), ("y-major (50..130 deg)", 0.87, 2.27), ("all directions", -np.pi, np.pi)):
    ang = np.linspace(lo, hi, n, endpoint=False)
    rng = np.full(n, 0.05 * side * 0.45)
    for _ in range(20): m.scan_to_grid_device(ang, rng)
    t0 = time.perf_counter()
    for _ in range(200): m.scan_to_grid_device(ang, rng)
    print(f"{name:24s} {1e6 * (time.perf_counter() - t0) / 200:6.1f} us per scan")
