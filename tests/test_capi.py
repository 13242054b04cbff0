"""CPU-side checks of the boundary: the HIP library loads, exports every symbol include/ftgp.h declares,
and refuses loudly to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ftgp.h")).read()
    return sorted(set(re.findall(r"\b(ftgp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted("ftgp_" + s for s in capi.API_SYMBOLS)


def test_library_exports_every_declared_symbol(product):
    for sym in header_symbols():
        assert hasattr(product.dll, sym), sym


def test_shipped_library_is_the_trees_sources_without_diagnostics(product):
    """ftgp_build_info: the library under ft_grandprix_amd/lib was built by __graft_entry__.build() from the sources in the tree (their
    hash, tools/evidence.py sha) and with no diagnostic switch (csrc/diag/ftgp_diag.inc compiles every hook out)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from evidence import sha
    finally:
        sys.path.pop(0)
    info = product.build_info()
    assert info["abi"] == str(capi.ABI_VERSION)
    assert info["diag"] == "none", info
    assert info["sources"] == sha(), "ft_grandprix_amd/lib/libftgp.so was not built from these sources: run __graft_entry__.build()"
    text = open(os.path.join(ROOT, "ft_grandprix_amd", "csrc", "ftgp_kernels.hip")).read()
    assert "#ifdef FTGP_" not in text and "#ifndef FTGP_NO" not in text      # hooks live in csrc/diag/ftgp_diag.inc behind -DFTGP_DIAG


def test_struct_layouts_match_the_header(product, oracle):
    # both libraries fill FtgpVehicle through the same C struct: identical bytes => identical layout and constants
    a, b = product.default_vehicle(), oracle.default_vehicle()
    assert bytes(a) == bytes(b)
    assert C.sizeof(capi.FtgpVehicle) == 8 * 41 and a.softener_radius == pytest.approx(0.03172) and a.kind == 0
    assert bytes(product.tricycle_vehicle()) == bytes(oracle.tricycle_vehicle()) and product.tricycle_vehicle().kind == 1
    assert a.mass == pytest.approx(5.632768) and a.wheel_x[0] == 0.06925 and a.lidar_x == -0.0525


def test_no_gpu_means_loud_failure_not_fallback(product):
    if product.fn("device_count")() >= 1:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.FtgpError) as ei:
        capi.Env(product, load_track("small-circle"), n_envs=1, n_rays=8)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_bad_arguments_are_rejected(oracle):
    t = load_track("small-circle")
    with pytest.raises(capi.FtgpError):
        capi.Env(oracle, t, n_envs=0)
    with pytest.raises(capi.FtgpError):
        capi.Env(oracle, t, n_envs=1, cars_per_env=9)


def test_box_field_must_stay_addressable_with_32_bits(product):
    """The march addresses the sector box field (64 planes x 2 bytes per pixel, ring included) with a 32-bit byte offset:
    ftgp_create refuses an image whose field would reach 4 GiB instead of wrapping silently.  Error path only -- the check
    sits in front of the device probe and allocates nothing (the bitmap pointer is never read)."""
    class Huge:
        width, height = 8192, 8192
        px_size_x = px_size_y = 40.0 / 8192
        origin_x, origin_y = 0.0, 0.0
        bits = np.zeros((1, 256), dtype=np.uint32)          # one row stands in for the 256-MB bitmap; words_per_row is what is checked
        path = np.zeros((100, 2))
    with pytest.raises(capi.FtgpError) as ei:
        capi.Env(product, Huge, n_envs=1, n_rays=8)
    assert ei.value.code == -1 and "4 GiB" in str(ei.value)
    assert (2 * 8194 * 4096 + 255) // 256 * 256 * 64 > 2 ** 32 >= (2 * 8194 * 4095 + 255) // 256 * 256 * 64    # 8192 x 4093 is the last height that fits


def test_product_package_never_references_the_oracle():
    pkg = os.path.join(ROOT, "ft_grandprix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "libftgp_oracle" not in text and "oracle/" not in text.replace("the oracle/", ""), f
