"""Point cloud -> laserscan (kc_cloud_to_laserscan, host buffer in, ranges out) and the CriticalZoneChecker over input
sizes: a search for cliffs.  python tools/cloud_zone_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

rng = np.random.default_rng(0)
for n in (1000, 10000, 100000, 1000000, 4000000):
    for bins in (360, 2048, 16384):
        xyz = np.zeros((n, 4), np.float32)
        xyz[:, :2] = rng.uniform(-20, 20, (n, 2))
        xyz[:, 2] = rng.uniform(-0.5, 1.5, n)
        host = xyz.reshape(-1).view(np.int8)
        ctx = kh.CloudContext(max_bytes=host.size, max_bins=bins)
        f = lambda: ctx.to_laserscan(host, 16, n * 16, 1, n, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=bins)
        for _ in range(5):
            f()
        ts = []
        for _ in range(30):
            t = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t)
        print("cloud %8d points -> %5d bins: %.1f us per call, host buffer in (%.2f GB/s)" %
              (n, bins, np.percentile(ts, 50) * 1e6, host.size / np.percentile(ts, 50) / 1e9), flush=True)
        ctx.close()
for beams in (90, 360, 1440, 4096, 16384):
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    z = kh.ZoneContext(syn.CYLINDER, [0.1, 0.4], (0, 0, 0.1), (0, 0, 0, 1), 160.0, 0.3, 0.6, ang, 0.1, 2.0, 20.0)
    r = rng.uniform(0.2, 8.0, beams)
    for _ in range(20):
        z.check(r, True)
    ts = []
    for _ in range(200):
        t = time.perf_counter()
        z.check(r, True)
        ts.append(time.perf_counter() - t)
    print("zone check, %5d beams: %.1f us per call" % (beams, np.percentile(ts, 50) * 1e6), flush=True)
    z.close()
