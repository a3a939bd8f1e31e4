"""M5: raw point cloud -> laserscan.

CPU part: the oracle restatement of pointCloudToLaserScanFromRaw
(utils/pointcloud.h:116-259) against the properties the reference's own tests
hold (tests/test_pointcloud_data.py:154-268: ring, origin filter, z filter).
GPU part: the HIP path through the C ABI against the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import ko

STRIDE = 16  # x, y, z float32 + 4 bytes of padding (test_pointcloud_data.py:140-146)


def cloud_bytes(xyz, stride=STRIDE, lead=0):
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    rec = np.zeros((len(xyz), stride), np.uint8)
    rec[:, lead:lead + 12] = xyz.view(np.uint8).reshape(-1, 12)
    return rec.reshape(-1).view(np.int8)


def ring(n, radius=1.0, z=0.5):
    th = np.linspace(0.0, 2.0 * np.pi, n, endpoint=False)
    return np.column_stack([radius * np.cos(th), radius * np.sin(th), np.full(n, z)])


def test_oracle_ring_populates_bins():
    # test_pointcloud_data.py:154-200
    n, max_range, step = 100, 10.0, 0.05
    r, a = ko.pointcloud_to_laserscan(cloud_bytes(ring(n)), STRIDE, n * STRIDE, 1, n, 0, 4, 8,
                                      max_range, 0.0, 1.0, angle_step=step)
    bins = int(np.ceil(2.0 * np.pi / step))
    assert r.shape == (bins,) and a.shape == (bins,)
    np.testing.assert_array_equal(a, np.arange(bins) * step)
    hit = r[r < max_range]
    assert len(hit) > 0.4 * bins
    assert np.all(np.abs(hit - 1.0) < 1e-3)


def test_oracle_origin_points_are_filtered():
    # test_pointcloud_data.py:203-228
    n = 50
    r, _ = ko.pointcloud_to_laserscan(cloud_bytes(np.zeros((n, 3))), STRIDE, n * STRIDE, 1, n, 0, 4, 8,
                                      5.0, -1.0, 1.0, angle_step=0.1)
    assert np.all(r == 5.0)


def test_oracle_z_filter_rejects_above_ceiling():
    # test_pointcloud_data.py:231-259
    n = 40
    r, _ = ko.pointcloud_to_laserscan(cloud_bytes(ring(n, z=3.0)), STRIDE, n * STRIDE, 1, n, 0, 4, 8,
                                      10.0, 0.0, 1.0, angle_step=0.1)
    assert np.all(r == 10.0)


def test_oracle_num_bins_overload_and_minimum():
    # pointcloud.h:205-259: bin = int(angle / 2pi * num_bins); closest point wins
    pts = np.array([[2.0, 0.1, 0.0], [1.0, 0.1, 0.0], [-0.1, 3.0, 0.0], [-1.5, -0.1, 0.0], [0.1, -2.5, 0.0]])
    r = ko.pointcloud_to_laserscan(cloud_bytes(pts), STRIDE, len(pts) * STRIDE, 1, len(pts), 0, 4, 8,
                                   10.0, -1.0, -1.0, num_bins=4)
    f = lambda x, y: float(np.sqrt(np.float32(np.float32(x) * np.float32(x) + np.float32(y) * np.float32(y))))
    np.testing.assert_array_equal(r, [f(1.0, 0.1), f(-0.1, 3.0), f(-1.5, -0.1), f(0.1, -2.5)])


def _random_cloud(rng, n, edge_points=200, bins=360):
    xyz = np.column_stack([rng.uniform(-8, 8, n), rng.uniform(-8, 8, n), rng.uniform(-0.5, 2.5, n)])
    # points exactly on / next to bin edges, on the axes, at the origin, out of range
    k = rng.integers(0, bins, edge_points)
    th = k * (2.0 * np.pi / bins) + rng.choice([0.0, 1e-7, -1e-7, 3e-7], edge_points)
    rad = rng.uniform(0.5, 6.0, edge_points)
    edge = np.column_stack([rad * np.cos(th), rad * np.sin(th), np.full(edge_points, 0.3)])
    special = np.array([[1.0, 0.0, 0.1], [1.0, -0.0, 0.1], [2.0, -1e-30, 0.1], [-3.0, 0.0, 0.1], [-3.0, -0.0, 0.1],
                        [0.0, 2.0, 0.1], [0.0, -2.0, 0.1], [0.0, 0.0, 0.1], [5e-4, 5e-4, 0.1], [1e-3, 0.0, 0.1],
                        [30.0, 1.0, 0.1], [np.inf, 1.0, 0.1], [1.0, np.nan, 0.1], [1.0, 1.0, np.nan]])
    out = np.vstack([xyz, edge, special]).astype(np.float32)
    rng.shuffle(out)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["packed16", "stride32_lead4", "rows_with_padding", "unaligned"])
def test_gpu_matches_oracle_bit_for_bit(layout):
    import kompass_hip as kh

    rng = np.random.default_rng(11)
    xyz = _random_cloud(rng, 20000)
    n = len(xyz)
    if layout == "packed16":
        data, step, row, h, w, xo, yo, zo = cloud_bytes(xyz), 16, n * 16, 1, n, 0, 4, 8
    elif layout == "stride32_lead4":
        data, step, row, h, w, xo, yo, zo = cloud_bytes(xyz, 32, 4), 32, n * 32, 1, n, 4, 8, 12
    elif layout == "rows_with_padding":
        w, h = 100, n // 100
        body = cloud_bytes(xyz[:w * h]).view(np.uint8).reshape(h, w * 16)
        padded = np.concatenate([body, np.zeros((h, 8), np.uint8)], axis=1)  # row_step not a multiple of point_step
        data, step, row, xo, yo, zo = padded.reshape(-1).view(np.int8), 16, w * 16 + 8, 0, 4, 8
    else:  # records of 13 bytes: every float is misaligned
        data, step, row, h, w, xo, yo, zo = cloud_bytes(xyz, 13, 1), 13, n * 13, 1, n, 1, 5, 9
    ctx = kh.CloudContext(max_bytes=len(data), max_bins=720)
    for kw in (dict(angle_step=0.0175), dict(num_bins=360), dict(num_bins=720), dict(angle_step=1.0)):
        for max_z in (2.0, -1.0):
            want = ko.pointcloud_to_laserscan(data, step, row, h, w, xo, yo, zo, 12.0, 0.0, max_z, **kw)
            got = ctx.to_laserscan(data, step, row, h, w, xo, yo, zo, 12.0, 0.0, max_z, **kw)
            if "angle_step" in kw:
                np.testing.assert_array_equal(got[1].view(np.uint64), want[1].view(np.uint64))
                got, want = got[0], want[0]
            np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
            assert 0 < ctx.last_rebinned() < n // 20  # edge points went to the host, the bulk did not
    ctx.close()


@pytest.mark.gpu
def test_gpu_edge_cases():
    import kompass_hip as kh

    ctx = kh.CloudContext()
    # empty cloud, nothing in range, negative max_range, truncated buffer
    r = ctx.to_laserscan(np.zeros(0, np.int8), 16, 0, 0, 0, 0, 4, 8, 7.0, 0.0, 1.0, num_bins=8)
    assert np.all(r == 7.0) and len(r) == 8
    pts = cloud_bytes(ring(64, radius=20.0))
    r = ctx.to_laserscan(pts, 16, 64 * 16, 1, 64, 0, 4, 8, 7.0, 0.0, 1.0, num_bins=8)
    assert np.all(r == 7.0)
    r = ctx.to_laserscan(pts, 16, 64 * 16, 1, 64, 0, 4, 8, -1.0, 0.0, 1.0, num_bins=8)
    assert np.all(r == -1.0)
    cut = cloud_bytes(ring(64))[:-6]  # the last record's z is out of bounds: skipped by both
    want = ko.pointcloud_to_laserscan(cut, 16, 64 * 16, 1, 64, 0, 4, 8, 7.0, 0.0, 1.0, num_bins=32)
    got = ctx.to_laserscan(cut, 16, 64 * 16, 1, 64, 0, 4, 8, 7.0, 0.0, 1.0, num_bins=32)
    np.testing.assert_array_equal(got, want)
    with pytest.raises((ValueError, kh.KompassHipError)):
        ctx.to_laserscan(pts, 0, 64 * 16, 1, 64, 0, 4, 8, 7.0, 0.0, 1.0, num_bins=8)
    ctx.close()


@pytest.mark.gpu
def test_gpu_large_cloud_properties():
    """1M points (BASELINE-sized cloud for this row): size-independent
    properties -- every range is a distance that occurs in its bin, none is
    beaten by a point of the same bin, idempotent."""
    import kompass_hip as kh

    rng = np.random.default_rng(5)
    n = 1_000_000
    xyz = np.column_stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(0.0, 1.0, n)]).astype(np.float32)
    data = cloud_bytes(xyz)
    ctx = kh.CloudContext(max_bytes=len(data), max_bins=2048)
    r1 = ctx.to_laserscan(data, 16, n * 16, 1, n, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=2048)
    r2 = ctx.to_laserscan(data, 16, n * 16, 1, n, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=2048)
    np.testing.assert_array_equal(r1, r2)
    d = np.sqrt((xyz[:, 0] * xyz[:, 0] + xyz[:, 1] * xyz[:, 1]).astype(np.float32)).astype(np.float64)
    ang = np.arctan2(xyz[:, 1].astype(np.float64), xyz[:, 0].astype(np.float64))
    ang[ang < 0] += 2 * np.pi
    t = ang / (2 * np.pi) * 2048
    b = np.minimum(t.astype(np.int64), 2047)
    inner = np.abs(t - np.round(t)) > 1e-3  # points well inside their bin: numpy's bin is the reference's
    best = np.full(2048, 25.0)
    np.minimum.at(best, b[inner], d[inner])
    assert np.all(r1 <= best)                       # nothing of the bin beats the result
    assert np.all(np.isin(r1[r1 < 25.0], d))        # every result is some point's distance
    want = ko.pointcloud_to_laserscan(data, 16, n * 16, 1, n, 0, 4, 8, 25.0, 0.0, 1.0, num_bins=2048)
    np.testing.assert_array_equal(r1.view(np.uint64), want.view(np.uint64))
    ctx.close()


@pytest.mark.gpu
def test_module_pointcloud_to_laserscan_reference_tests():
    """tests/test_pointcloud_data.py:154-259 of the reference, against this
    build's kompass_cpp.utils (same call, same keyword names)."""
    from kompass_cpp.utils import pointcloud_to_laserscan_from_raw

    n = 100
    cloud = cloud_bytes(ring(n))
    ranges, angles = pointcloud_to_laserscan_from_raw(
        data=cloud, point_step=STRIDE, row_step=n * STRIDE, height=1, width=n, x_offset=0, y_offset=4, z_offset=8,
        max_range=10.0, min_z=0.0, max_z=1.0, angle_step=0.05)
    ranges, angles = np.asarray(ranges), np.asarray(angles)
    bins = int(np.ceil(2.0 * np.pi / 0.05))
    assert ranges.shape == (bins,) and angles.shape == (bins,)
    hit = ranges[ranges < 10.0]
    assert len(hit) > 0.4 * bins and np.all(np.abs(hit - 1.0) < 1e-3)
    want, _ = ko.pointcloud_to_laserscan(cloud, STRIDE, n * STRIDE, 1, n, 0, 4, 8, 10.0, 0.0, 1.0, angle_step=0.05)
    np.testing.assert_array_equal(ranges, want)

    zeros = cloud_bytes(np.zeros((50, 3)))
    r, _ = pointcloud_to_laserscan_from_raw(data=zeros, point_step=STRIDE, row_step=50 * STRIDE, height=1, width=50,
                                            x_offset=0, y_offset=4, z_offset=8, max_range=5.0, min_z=-1.0,
                                            max_z=1.0, angle_step=0.1)
    assert np.all(np.asarray(r) == 5.0)
    above = cloud_bytes(ring(40, z=3.0))
    r, _ = pointcloud_to_laserscan_from_raw(data=above, point_step=STRIDE, row_step=40 * STRIDE, height=1, width=40,
                                            x_offset=0, y_offset=4, z_offset=8, max_range=10.0, min_z=0.0,
                                            max_z=1.0, angle_step=0.1)
    assert np.all(np.asarray(r) == 10.0)
    r = pointcloud_to_laserscan_from_raw(data=cloud, point_step=STRIDE, row_step=n * STRIDE, height=1, width=n,
                                         x_offset=0, y_offset=4, z_offset=8, max_range=10.0, min_z=0.0, max_z=1.0,
                                         num_bins=64)
    np.testing.assert_array_equal(np.asarray(r), ko.pointcloud_to_laserscan(cloud, STRIDE, n * STRIDE, 1, n, 0, 4, 8,
                                                                           10.0, 0.0, 1.0, num_bins=64))


@pytest.mark.gpu
def test_local_mapper_pointcloud_overload():
    """LocalMapper(is_pointcloud=True).scan_to_grid(data, ...) =
    pointcloud -> ranges over scan_size bins -> scanToGrid with angles
    i * 2 pi / scan_size (local_mapper.cpp:243-251, local_mapper.h:38-56)."""
    import kompass_cpp

    rng = np.random.default_rng(3)
    n, scan = 5000, 360
    xyz = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-4, 4, n), rng.uniform(0.0, 1.5, n)]).astype(np.float32)
    data = cloud_bytes(xyz)
    H = W = 200
    for cls in (kompass_cpp.mapping.LocalMapper, kompass_cpp.mapping.LocalMapperGPU):
        m = cls(grid_height=H, grid_width=W, resolution=0.05, laserscan_position=np.array([0.0, 0.0, 0.0], np.float32),
                laserscan_orientation=0.0, is_pointcloud=True, scan_size=scan, angle_step=0.01, max_height=1.0,
                min_height=0.1, range_max=6.0, max_points_per_line=32)
        g = np.array(m.scan_to_grid(data=data, point_step=16, row_step=n * 16, height=1, width=n, x_offset=0,
                                    y_offset=4, z_offset=8))
        ranges = ko.pointcloud_to_laserscan(data, 16, n * 16, 1, n, 0, 4, 8, np.float32(6.0), np.float32(0.1),
                                            np.float32(1.0), num_bins=scan)
        want = ko.scan_to_grid(H, W, 0.05, (0, 0, 0), 0.0, np.arange(scan) * (2.0 * np.pi / scan), ranges)
        np.testing.assert_array_equal(g, want)
        assert (g == 100).sum() > 50


# ---- PointCloud2 fields of any datatype (PointFieldType 1-8, utils/pointcloud.h:37-87) ------------------------
FIELD_DTYPES = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32, 7: np.float32, 8: np.float64}


def typed_cloud(xyz, ftype, pad=3, lead=1):
    """records of [lead bytes][x][y][z][pad bytes] with fields of datatype `ftype` (values cast, not scaled)."""
    dt = np.dtype(FIELD_DTYPES[ftype])
    vals = np.asarray(xyz, np.float64)
    if dt.kind in "iu":
        info = np.iinfo(dt)
        vals = np.clip(np.rint(vals), info.min, info.max)
    vals = vals.astype(dt)
    step = lead + 3 * dt.itemsize + pad
    rec = np.zeros((len(vals), step), np.uint8)
    rec[:, lead:lead + 3 * dt.itemsize] = vals.view(np.uint8).reshape(len(vals), -1)
    return rec.reshape(-1).view(np.int8), step, (lead, lead + dt.itemsize, lead + 2 * dt.itemsize), vals


@pytest.mark.parametrize("ftype", sorted(FIELD_DTYPES))
def test_oracle_typed_fields_decode_like_load_and_cast_val(ftype):
    rng = np.random.default_rng(ftype)
    xyz = np.column_stack([rng.uniform(-20, 20, 300), rng.uniform(-20, 20, 300), rng.uniform(0, 2, 300)])
    data, step, (xo, yo, zo), vals = typed_cloud(xyz, ftype)
    r = ko.pointcloud_to_laserscan(data, step, len(vals) * step, 1, len(vals), xo, yo, zo, 50.0, -1.0, -1.0,
                                   num_bins=90, field_type=ftype)
    # the same points as float32 records: the decode is `static_cast<float>(value)`
    f = vals.astype(np.float32)
    want = ko.pointcloud_to_laserscan(cloud_bytes(f), STRIDE, len(f) * STRIDE, 1, len(f), 0, 4, 8, 50.0, -1.0, -1.0,
                                      num_bins=90)
    np.testing.assert_array_equal(r, want)
    assert (r < 50.0).sum() > 20


def _libm_atan2f():
    import ctypes
    import ctypes.util
    f = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6").atan2f
    f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
    return f


_atan2f = _libm_atan2f()


def _load_and_cast_bytes(buf, off, ftype):
    """load_and_cast_val (utils/pointcloud.h:49-87) restated from the bytes up: the field's bytes are gathered one by
    one from `off` (any alignment), assembled little-endian into the field's type, `static_cast<float>` of that."""
    size = np.dtype(FIELD_DTYPES[ftype]).itemsize
    raw = bytes(int(buf[off + i]) & 0xFF for i in range(size))
    if ftype in (1, 3, 5):
        return np.float32(int.from_bytes(raw, "little", signed=True))
    if ftype in (2, 4, 6):
        return np.float32(int.from_bytes(raw, "little", signed=False))
    import struct
    with np.errstate(over="ignore"):
        return np.float32(struct.unpack("<f" if ftype == 7 else "<d", raw)[0])


def _scan_from_bytes(buf, nbytes, step, n, offs, ftype, max_range, min_z, max_z, num_bins):
    """pointCloudToLaserScanFromRaw (pointcloud.h:205-259) in plain Python over `_load_and_cast_bytes`; a record whose
    furthest field would read past `nbytes` is skipped (:139-146, with the field's own size)."""
    size = np.dtype(FIELD_DTYPES[ftype]).itemsize
    out = np.full(num_bins, float(max_range))
    for k in range(n):
        start = k * step
        if start + max(offs) + size > nbytes:
            continue
        x, y, z = (_load_and_cast_bytes(buf, start + o, ftype) for o in offs)
        with np.errstate(over="ignore", invalid="ignore"):
            r2 = np.float32(np.float32(x * x) + np.float32(y * y))
        if float(r2) < 1e-6 or float(z) < min_z or (max_z >= 0.0 and float(z) > max_z):
            continue
        if not (np.isfinite(x) and np.isfinite(y)):
            continue
        ang = float(_atan2f(float(y), float(x)))   # std::atan2(float, float) = libm's atan2f (numpy's own float32
        #                                            arctan2 is 1 ulp off at x == y, which moves points across bins)
        if ang < 0.0:
            ang += 2.0 * np.pi
        b = min(int((ang / (2.0 * np.pi)) * num_bins), num_bins - 1)
        d = float(np.sqrt(r2))
        if d < out[b]:
            out[b] = d
    return out


@pytest.mark.parametrize("ftype", sorted(FIELD_DTYPES))
@pytest.mark.parametrize("lead", [0, 1, 3])
def test_oracle_typed_decode_byte_by_byte(ftype, lead):
    """the C oracle's typed decode against a restatement from the bytes up, per datatype: unaligned field
    offsets (lead 1 and 3), extreme values (integers beyond 2^24 that round in the cast, UINT32 above 2^31, doubles
    that overflow float, NaN / inf), and a buffer that ends INSIDE the last record's z field.
    Parity of the non-FLOAT32 decode itself stays unpinned: the reference holds no fixture for it and uses
    load_and_cast_val only inside its device kernels."""
    dt = np.dtype(FIELD_DTYPES[ftype])
    rng = np.random.default_rng(100 * ftype + lead)
    n = 120
    xyz = np.column_stack([rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(0, 2, n)])
    data, step, offs, vals = typed_cloud(xyz, ftype, pad=int(rng.integers(0, 4)), lead=lead)
    data = data.copy()
    rec = data.view(np.uint8).reshape(n, step)
    if dt.kind in "iu":
        info = np.iinfo(dt)
        extremes = np.array([[info.max, info.min, 1], [info.min, info.max, 0], [info.max - 1, 3, 1],
                             [16777217 % (int(info.max) + 1), 5, 1]], dtype=np.int64).astype(dt)
    else:
        big = 1e300 if ftype == 8 else 3e38
        extremes = np.array([[big, 1.0, 1.0], [np.nan, 1.0, 1.0], [2.0, np.inf, 1.0], [-0.0, 1e-4, 1.0]], dtype=dt)
    rec[:4, lead:lead + 3 * dt.itemsize] = extremes.view(np.uint8).reshape(4, -1)
    nbytes = data.size - step + offs[2] + dt.itemsize - 1      # the last record's z field lacks its last byte
    want = _scan_from_bytes(data, nbytes, step, n, offs, ftype, 60.0, -1.0, -1.0, 72)
    got = ko.pointcloud_to_laserscan(data[:nbytes], step, n * step, 1, n, *offs, 60.0, -1.0, -1.0, num_bins=72,
                                     field_type=ftype)
    np.testing.assert_array_equal(got, want)
    full = ko.pointcloud_to_laserscan(data, step, n * step, 1, n, *offs, 60.0, -1.0, -1.0, num_bins=72, field_type=ftype)
    np.testing.assert_array_equal(full, _scan_from_bytes(data, data.size, step, n, offs, ftype, 60.0, -1.0, -1.0, 72))
    assert (want < 60.0).sum() > 10


@pytest.mark.gpu
@pytest.mark.parametrize("ftype", sorted(FIELD_DTYPES))
def test_hip_typed_fields_match_oracle(ftype):
    import kompass_hip as kh

    rng = np.random.default_rng(100 + ftype)
    n = 20000
    scale = 100.0 if ftype in (1, 2) else 30.0
    xyz = np.column_stack([rng.uniform(-scale, scale, n), rng.uniform(-scale, scale, n), rng.uniform(0, 3, n)])
    data, step, (xo, yo, zo), vals = typed_cloud(xyz, ftype, pad=int(rng.integers(0, 5)), lead=int(rng.integers(0, 4)))
    ctx = kh.CloudContext(max_bytes=data.size, max_bins=720)
    for kw in (dict(angle_step=0.01), dict(num_bins=360)):
        for nbytes in (data.size, data.size - step // 2 - 1):   # the last record cut short: bounds check with the field's size
            d = data[:nbytes]
            got = ctx.to_laserscan(d, step, len(vals) * step, 1, len(vals), xo, yo, zo, 120.0, 0.5, 2.5, field_type=ftype, **kw)
            want = ko.pointcloud_to_laserscan(d, step, len(vals) * step, 1, len(vals), xo, yo, zo, 120.0, 0.5, 2.5,
                                              field_type=ftype, **kw)
            if "angle_step" in kw:
                np.testing.assert_array_equal(got[0].view(np.uint64), want[0].view(np.uint64))
                np.testing.assert_array_equal(got[1], want[1])
            else:
                np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))
    with pytest.raises(ValueError):
        ctx.to_laserscan(data, step, len(vals) * step, 1, len(vals), xo, yo, zo, 120.0, 0.5, 2.5, num_bins=8, field_type=9)


@pytest.mark.gpu
def test_zone_checker_takes_typed_clouds():
    import kompass_hip as kh

    ang = np.linspace(0.0, 2.0 * np.pi, 360, endpoint=False)
    args = (kh.CYLINDER, [0.3, 0.5], (0.2, 0.0, 0.1), (0, 0, 0, 1), 160.0, 1.0, 4.0, ang, 0.0, 3.0, 40.0)
    rng = np.random.default_rng(7)
    for ftype in (3, 5, 8):
        xyz = np.column_stack([rng.uniform(-9, 9, 4000), rng.uniform(-9, 9, 4000), rng.uniform(0, 2, 4000)])
        data, step, (xo, yo, zo), vals = typed_cloud(xyz, ftype)
        z = kh.ZoneContext(*args)
        o = ko.CriticalZone(*args, field_type=ftype)
        for fwd in (True, False):
            got = z.check_cloud(data, step, len(vals) * step, 1, len(vals), xo, yo, zo, fwd, field_type=ftype)
            want = o.check_cloud(data, step, len(vals) * step, 1, len(vals), xo, yo, zo, fwd)
            assert np.float32(got) == np.float32(want)
