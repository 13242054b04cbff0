"""K2 (the rangefinder sweep that replaces MuJoCo's sensors) against the reference's own 2-D ray cast, raycast.py:5-21, through
its golden scans (fixture G2: 4 tracks x 32 origins x {36, 1080} rays, produced by running the reference).

K2 returns the exact distance to the first wall-pixel boundary; fakelidar sphere-traces a Euclidean distance transform with
truncated pixel lookups and stops within eps = 2 px of a wall.  So, ray by ray, fakelidar may exceed K2 only by its own overshoot
(< 1.5 px) and otherwise stops short of it -- by 0 .. 2.5 px typically, by more where a ray grazes a wall (there sphere tracing
ends early).  Agreement of that kind on every ray pins K2's ray ORDER (index 0 = rear, counter-clockwise), its orientation, its
frame (pixel -> world mapping, y up) and its scale to the reference's code; a mirrored, rotated or shifted fan fails it at once
(checked below by deliberately breaking the mapping).

CPU: the oracle (bit-identical to the GPU on every ray, tests/test_gpu_parity.py); `-m gpu`: libftgp.so itself.
"""
import numpy as np
import pytest

from tests.helpers import k2_minus_fakelidar_square_pixels, k2_minus_fakelidar_along_k2_rays

TRACKS = ["track", "circle", "small-circle", "inkscape"]


def check(d):
    assert d.min() >= -1.5, d.min()                              # K2 * px - 0.03 * px >= fakelidar - 1.5 px, every ray
    assert 0.0 <= np.median(d) <= 2.5, np.median(d)              # ... and the median difference in [0, 2.5] px
    inside = ((d >= -1.5) & (d <= 4.0)).mean()
    assert inside >= 0.85, inside                                # the tail is grazing rays, where sphere tracing stops early


@pytest.mark.parametrize("name", TRACKS)
@pytest.mark.parametrize("R", [36, 1080])
def test_oracle_k2_matches_reference_fakelidar_goldens(oracle, name, R):
    check(k2_minus_fakelidar_square_pixels(oracle, name, R))


@pytest.mark.parametrize("name", ["circle", "small-circle", "inkscape"])
def test_oracle_k2_in_the_stretched_wall_frame(oracle, name):
    """2133 x 1600 images are squashed onto the 40 x 40 map (mushr.em.xml:17-20): pixels are not square there."""
    check(k2_minus_fakelidar_along_k2_rays(oracle, name, 1080))


def test_the_comparison_is_sensitive_to_ray_order_and_orientation(oracle):
    """The same K2 sweep against the goldens taken the other way round, a quarter turn off, or just one degree (3 rays) off:
    the every-ray bound fails by tens to hundreds of pixels, i.e. the test above does pin index 0, the sense of rotation and
    the frame."""
    from tests.helpers import golden
    scan = np.load(golden("g2_fakelidar.npz"))["track_1080_scan"]
    k2_px = k2_minus_fakelidar_square_pixels(oracle, "track", 1080) + scan
    for wrong, gross in ((scan[:, ::-1], True), (np.roll(scan, 270, axis=1), True), (np.roll(scan, 3, axis=1), False), (np.roll(scan, -3, axis=1), False)):
        e = k2_px - wrong
        assert e.min() < -100.0
        assert not gross or ((e >= -1.5) & (e <= 4.0)).mean() < 0.2


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRACKS)
def test_gpu_k2_matches_reference_fakelidar_goldens(product, oracle, name):
    for R in (36, 1080):
        d = k2_minus_fakelidar_square_pixels(product, name, R)
        check(d)
        np.testing.assert_array_equal(d, k2_minus_fakelidar_square_pixels(oracle, name, R))
    if name != "track":
        d = k2_minus_fakelidar_along_k2_rays(product, name, 1080)
        check(d)
        np.testing.assert_allclose(d, k2_minus_fakelidar_along_k2_rays(oracle, name, 1080), rtol=0, atol=1e-9)
