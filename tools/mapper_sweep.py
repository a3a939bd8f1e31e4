"""LocalMapper scan -> grid over grid sizes, beam counts and ranges (plain and Bayesian): a search for cliffs of the
mapper path.  python tools/mapper_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

for side, res in ((100, 0.1), (200, 0.1), (400, 0.05), (1000, 0.05), (2000, 0.05)):
    for beams in (360, 1440, 4096, 16384):
        for rscale in (0.3, 1.0, 4.0):
            ang, rng = syn.dense_scan(beams, rscale)
            row = []
            for bayes in (False, True):
                m = kh.MapperContext(side, side, res, (0, 0, 0), 0.0, beams)
                if bayes:
                    m.enable_bayes()
                f = (lambda: m.scan_to_grid_baysian_device(ang, rng)) if bayes else (lambda: m.scan_to_grid_device(ang, rng))
                for i in range(20):
                    f()
                ts = []
                for i in range(100):
                    t = time.perf_counter()
                    f()
                    ts.append(time.perf_counter() - t)
                row.append(np.percentile(ts, 50) * 1e6)
                m.close()
            print("grid %4d^2 @ %.2f m, %5d beams, ranges x %.1f: plain %.1f us, Bayesian %.1f us (grid resident on the device)" % (side, res, beams, rscale, row[0], row[1]), flush=True)
