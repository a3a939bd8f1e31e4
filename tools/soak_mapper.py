import sys, time, zlib
import os; ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np, kompass_hip as kh, synthetic as syn
m = kh.MapperContext(400, 300, 0.05, (0.1, -0.05, 0), 0.3, 2048)
m.enable_bayes(0.6, 0.9, 0.1, 0.1, 20.0, 0.2)
ang, rng = syn.dense_scan(2048, 1.5)
scans = [rng * (1 + 0.03 * k) for k in range(5)]
first = {}; n = bad = 0
t_end = time.perf_counter() + 25
r = np.random.default_rng(1)
prevs = [r.uniform(0.05, 0.95, (400, 300)).astype(np.float32) for _ in range(2)]
while time.perf_counter() < t_end:
    k = n % 5; pv = (n // 5) % 2
    if n % 5 == 0:
        m.set_previous_prob(prevs[pv])
    g, p = m.scan_to_grid_baysian(ang, scans[k])
    h = (zlib.crc32(np.ascontiguousarray(g).tobytes()), zlib.crc32(np.ascontiguousarray(p).tobytes()))
    key = (k, pv)
    if key not in first: first[key] = h
    elif first[key] != h:
        bad += 1; print("MISMATCH", n, key)
    n += 1
print(n, "Bayesian scans,", bad, "mismatches")
c = kh.CloudContext(max_bytes=16 * 200000, max_bins=1024)
xyz = np.zeros((200000, 4), np.float32); xyz[:, :3] = r.uniform(-20, 20, (200000, 3)); xyz[:, 2] = r.uniform(0, 1, 200000)
buf = xyz.reshape(-1).view(np.int8)
ref = None; n2 = bad2 = 0
t_end = time.perf_counter() + 15
while time.perf_counter() < t_end:
    out = c.to_laserscan(buf, 16, 200000 * 16, 1, 200000, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=1024)
    h = zlib.crc32(np.ascontiguousarray(out).tobytes())
    if ref is None: ref = h
    elif ref != h: bad2 += 1
    n2 += 1
print(n2, "cloud conversions,", bad2, "mismatches")
