"""Mapper grid -> controller hand-off on the device.

The OCCUPIED cells of a device-resident LocalMapper grid become the controller's
sensor data without a host round trip.  Expected result: the oracle's point-list
cycle on the list a host would have extracted from the same grid.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

from helpers import assert_cycle_equal, hip_context, oracle_cycle  # noqa: E402


def _central(H, W):
    return int(round(H // 2)) - 1, int(round(W // 2)) - 1   # local_mapper.h:26-27


class _DeviceArray:
    """A device copy of a host array, made with the HIP runtime directly (a grid
    this library did not produce)."""

    def __init__(self, host):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]
        host = np.ascontiguousarray(host)
        self.p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.p), host.nbytes) == 0
        assert self.hip.hipMemcpy(self.p, host.ctypes.data_as(C.c_void_p), host.nbytes, 1) == 0

    @property
    def ptr(self):
        return self.p.value

    def free(self):
        if self.p:
            self.hip.hipFree(self.p)
            self.p = C.c_void_p()


def _points_from_grid(grid, res):
    """What a host would do with the grid: column-major walk, OCCUPIED cells ->
    ((i - c0) res, (j - c1) res, 0) in float."""
    H, W = grid.shape
    c0, c1 = _central(H, W)
    jj, ii = np.nonzero(grid.T == 100)      # column-major order
    pts = np.zeros((len(ii), 3), np.float32)
    pts[:, 0] = (ii - c0).astype(np.float32) * np.float32(res)
    pts[:, 1] = (jj - c1).astype(np.float32) * np.float32(res)
    return pts


def _cycle_from_ctx(ctx, inp):
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    res = ctx.cycle(st, inp["P"])
    px, py, raw, costs = ctx.get_samples(with_costs=True)
    out = dict(px=px.copy(), py=py.copy(), raw=raw.copy(), costs=costs.copy(), res=res.as_dict())
    if res.found:
        out["best"] = ctx.get_best()
    return out


@pytest.mark.parametrize("name,scale,side,res,beams,rscale", [
    ("cfg1", 1.0, 200, 0.05, 360, 0.3),
    ("cfg2", 0.25, 500, 0.05, 2048, 0.6),
    ("cfg2", 0.25, 1000, 0.05, 4096, 0.8),
])
def test_mapper_to_controller_on_device(name, scale, side, res, beams, rscale):
    inp = syn.make_controller_inputs(name, seed=3, scale=scale)
    ang, rng = syn.dense_scan(beams, rscale)
    m = kh.MapperContext(side, side, res, (0, 0, 0), 0.0, beams)
    grid = m.scan_to_grid(ang, rng).copy()          # host copy: only for the expectation
    inp["points"] = _points_from_grid(grid, res)
    assert len(inp["points"]) > 50
    o = oracle_cycle(inp)
    assert 0 < len(o["raw"]) < len(inp["vx"]), "scene must drop some samples and keep some"
    ctx = hip_context(kh, inp)
    for _ in range(2):                               # twice: counters re-armed, buffers reused
        m.scan_to_grid_device(ang, rng)              # scan in flight: ordering is the library's job
        ctx.set_grid_from_mapper(inp["state"], m, inp["max_range"])
        assert_cycle_equal(o, _cycle_from_ctx(ctx, inp))
    # same state as the point-list entry with the extracted list
    ctx2 = hip_context(kh, inp)
    ctx2.set_points(inp["state"], inp["points"], inp["max_range"])
    h2 = _cycle_from_ctx(ctx2, inp)
    assert_cycle_equal(o, h2)


@pytest.mark.parametrize("shape,dims,density", [
    (kh.CYLINDER, [0.1, 0.4], 0.004),     # device build
    (kh.BOX, [0.3, 0.2, 0.4], 0.004),
    (kh.CYLINDER, [0.1, 0.4], 0.12),      # > 16 k occupied cells: host lists from the device list
    (kh.SPHERE, [0.15], 0.004),           # spheres: the layer codes of the device build (one layer here)
])
def test_foreign_grid_on_device(shape, dims, density):
    """kc_dwa_set_grid_device on a grid this library did not produce (a torch tensor)."""
    inp = syn.make_controller_inputs("cfg2", seed=4, scale=0.25)
    inp["robot"] = dict(shape=shape, dims=dims)
    H, W, res = 420, 380, 0.05
    r = np.random.default_rng(8)
    grid = r.choice(np.array([-1, 0, 100], np.int32), size=(H, W), p=[0.3, 0.7 - density, density])
    c0, c1 = _central(H, W)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d2 = (ii - c0) ** 2 + (jj - c1) ** 2
    sparse = r.random((H, W)) < 0.004
    near = d2 < (3.0 / res) ** 2                                         # clutter only thins out near the robot
    grid[near & (grid == 100) & ~sparse] = 0
    grid[d2 < (0.6 / res) ** 2] = 0                                      # free disc around the robot
    inp["points"] = _points_from_grid(grid, res)
    o = oracle_cycle(inp)
    assert len(o["raw"]) > 0
    assert (len(inp["points"]) > 16384) == (density > 0.1)
    ctx = hip_context(kh, inp)
    dev = _DeviceArray(grid.T.copy().reshape(-1))                                  # i + j*H
    ctx.set_grid_device(inp["state"], dev.ptr, H, W, res, max_sensor_range=inp["max_range"])
    assert_cycle_equal(o, _cycle_from_ctx(ctx, inp))
    # pose batch on the same sensor state walks the host lists (fetched lazily from the device list)
    x, y, yaw = r.random(300) * 8 - 4, r.random(300) * 8 - 4, r.random(300) * 6.28 - 3.14
    got = ctx.check_poses(x, y, yaw)
    want = np.array([o["coll"].check_at(a, b, c) for a, b, c in zip(x, y, yaw)], bool)
    np.testing.assert_array_equal(np.asarray(got, bool), want)
    dev.free()


def test_empty_grid_and_errors():
    inp = syn.make_controller_inputs("cfg1", seed=1, scale=1.0)
    H = W = 64
    ctx = hip_context(kh, inp)
    dev = _DeviceArray(np.full(H * W, -1, np.int32))
    ctx.set_grid_device(inp["state"], dev.ptr, H, W, 0.1, max_sensor_range=inp["max_range"])
    inp["points"] = np.zeros((0, 3), np.float32)
    o = oracle_cycle(inp)
    assert len(o["raw"]) == len(inp["vx"])          # nothing to collide with
    assert_cycle_equal(o, _cycle_from_ctx(ctx, inp))
    with pytest.raises(ValueError):
        ctx.set_grid_device(inp["state"], dev.ptr, 0, W, 0.1)
    with pytest.raises(ValueError):
        ctx.set_grid_device(inp["state"], 0, H, W, 0.1)
    dev.free()
