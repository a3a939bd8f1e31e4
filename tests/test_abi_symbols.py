"""CPU-side checks of the drop-in boundary: libkompass_hip.so loads, exports
every symbol include/kompass_hip.h declares, and refuses to compute without a
HIP device (no silent CPU fallback).  No GPU needed."""
import re
from pathlib import Path

import pytest

import kompass_hip as kh

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "kompass_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 35
    L = kh.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in kompass_hip.h but not exported"
    assert set(names) == set(kh.SIGNATURES), "python binding table out of sync with the header"
    assert L.kc_abi_version() == 1


def test_key_pack_roundtrip_and_order():
    L = kh.lib()
    ks = []
    for cost, idx in [(0.0, 7), (0.0, 3), (1.5, 0), (-2.0, 9), (1e-30, 1), (3.0e38, 2)]:
        k = L.kc_key_pack(cost, idx)
        assert L.kc_key_index(k) == idx
        assert abs(L.kc_key_cost(k) - cost) <= abs(cost) * 1e-7
        ks.append((k, cost, idx))
    # int64 order == (cost, index) lexicographic order (LowestCost::combine)
    assert sorted(ks) == sorted(ks, key=lambda t: (t[1], t[2]))
    assert L.kc_key_pack(float("inf"), 1) == (1 << 63) - 1
    assert L.kc_key_pack(float("nan"), 1) == (1 << 63) - 1
    assert L.kc_key_pack(-0.0, 4) == L.kc_key_pack(0.0, 4)


def test_no_cpu_fallback_without_device():
    if kh.device_count() > 0:
        pytest.skip("a HIP device is visible here")
    with pytest.raises(kh.KompassHipError):
        kh.DwaContext(kh.CYLINDER, [0.1, 0.4])
    with pytest.raises(kh.KompassHipError):
        kh.MapperContext(10, 10, 0.1)


def test_argument_validation_precedes_device_use():
    with pytest.raises(ValueError):
        kh.DwaContext(7, [0.1, 0.4])  # Invalid robot geometry type
    with pytest.raises(ValueError):
        kh.MapperContext(0, 10, 0.1)
