"""Where the tiled scan spends its time: kernel times (HIP events) over beam
count and beam length.  python tools/tiles_probe.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kompass-core_amd")]
import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

for n, scale, bayes in [(4096, 4.0, True), (4096, 1.0, True), (1024, 4.0, True), (16384, 4.0, True), (4096, 4.0, False)]:
    ang, rng = syn.dense_scan(n, scale)
    m = kh.MapperContext(1000, 1000, 0.05, (0, 0, 0), 0.0, n)
    if bayes:
        m.enable_bayes()
    m.timing_enable(True)
    acc = {}
    for it in range(30):
        (m.scan_to_grid_baysian_device if bayes else m.scan_to_grid_device)(ang, rng)
        m.sync()
        if it >= 5:
            for k, v in m.timings():
                acc.setdefault(k, []).append(v)
    print(n, scale, bayes, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in acc.items()}, flush=True)
    m.close()
