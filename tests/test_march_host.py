"""The SHIPPED march (ftgp_march.h: ftgp_ray_init/step/fix/commit), box search (ftgp_box_entry) and host tables compiled for the host and checked against the plain-DDA
specification on millions of rays (random + rays from pixel corners + axis-aligned / diagonal directions)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from ft_grandprix_amd.track import load_track

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("march") / "march_check")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-fopenmp", "-x", "hip",
                           os.path.join(ROOT, "tools", "march_check.cpp"), "-o", out, "-ldl", "-w"])
    return out


@pytest.mark.parametrize("name", ["track", "inkscape"])
def test_shipped_march_equals_specification_on_host(harness, tmp_path, name):
    t = load_track(name)
    raw = tmp_path / f"{name}.raw"
    with open(raw, "wb") as f:
        np.array([t.width, t.height, t.words_per_row], dtype=np.int32).tofile(f)
        t.bits.tofile(f)
    scale = 1.0 / t.px_size_x
    r = subprocess.run([harness, str(raw), "1000000", "11", str(scale)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "grid_wall: 0 mismatching pixels" in r.stdout and ", 0 mismatches" in r.stdout


@pytest.mark.parametrize("coarse", [8, 16])
def test_coarse_planes_through_the_sector_table_equal_specification_on_host(harness, tmp_path, coarse):
    """The field as ftgp_create lays it out for large batches: `coarse` planes, a ray's sector -- always found among 64 -- mapped to its
    plane by the sector table (ftgp_sector_table) -- same bits as the plain DDA."""
    t = load_track("small-circle")
    raw = tmp_path / "small-circle.raw"
    with open(raw, "wb") as f:
        np.array([t.width, t.height, t.words_per_row], dtype=np.int32).tofile(f)
        t.bits.tofile(f)
    r = subprocess.run([harness, str(raw), "600000", str(coarse), str(1.0 / t.px_size_x), str(coarse)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"coarse planes: {coarse}" in r.stdout and ", 0 mismatches" in r.stdout
