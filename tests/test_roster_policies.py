"""The roster on the device: ftgp_set_car_policies + FTGP_POLICY_PER_CAR -- every car slot of an env its own bundled driver, as the
reference builds one Driver() per roster entry and calls them one by one (custom.py:1097-1104,1398-1411; template/cars/cars.json is
nidc, fast, nidc).  Pins: the drivers themselves are pinned by G1 (reference-run controls); here the DISPATCH is checked -- per-car
launches against single-driver launches, host Python drivers against the device, GPU against oracle."""
import os
import sys

import numpy as np
import pytest

from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track

ROSTER = ["nidc", "fast", "nidc"]            # template/cars/cars.json:1-5


def scans(seed, n_cars, n_rays):
    rng = np.random.default_rng(seed)
    r = rng.uniform(0.2, 8.0, size=(n_cars, n_rays)).astype(np.float32)
    r[:, ::7] *= 0.3                          # disparities
    return r


def check_dispatch(lib, n_rays=90):
    """policy_eval("per_car") = every car evaluated by its own driver's single-policy call."""
    t = load_track("track")
    with capi.Env(lib, t, n_envs=5, cars_per_env=3, n_rays=n_rays) as e:
        r = scans(3, e.n_cars, n_rays)
        e.set_car_policies(ROSTER)
        mixed = e.policy_eval("per_car", r)
        single = {p: e.policy_eval(p, r) for p in set(ROSTER)}
        for ci in range(e.n_cars):
            np.testing.assert_array_equal(mixed[ci], single[ROSTER[ci % 3]][ci])
        assert not np.array_equal(single["nidc"], single["fast"])          # the two drivers do differ on these scans


def test_oracle_per_car_dispatch(oracle):
    check_dispatch(oracle)


def test_per_car_needs_a_roster_and_bundled_drivers(oracle):
    t = load_track("small-circle")
    with capi.Env(oracle, t, n_envs=2, cars_per_env=2, n_rays=36) as e:
        with pytest.raises(capi.FtgpError) as ei:
            e.rollout("per_car", 1)
        assert ei.value.code == -4
        with pytest.raises(capi.FtgpError):
            e.set_car_policies(["nidc", "host"])
        with pytest.raises(ValueError):
            e.set_car_policies(["nidc"])
        e.set_car_policies(["lobotomy", "random"])
        e.rollout("per_car", 3)
        ctrl = e.ctrl()
        assert (ctrl[0::2] == 0).all() and (ctrl[1::2, 0] > 0).all()


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_roster_with_its_own_drivers_against_one_device_rollout(oracle):
    """template/cars/cars.json -- nidc, fast, nidc -- driven by the reference's OWN driver objects, one per car, step by step
    (sim.Simulator, the host path) against ONE rollout("roster"): the closed loops stay together (the Python drivers see binary64
    copies of the binary32 scans; the restated K5 agrees to rounding, as in test_unmodified_reference_drivers_close_the_loop)."""
    from ft_grandprix_amd.sim import Simulator
    if REF not in sys.path:
        sys.path.insert(0, REF)
    roster = Simulator.load_roster(os.path.join(REF, "template", "cars", "cars.json"))
    t = load_track("track")
    host = Simulator(t, roster, n_envs=2, n_rays=90, lib=oracle)
    dev = Simulator(t, roster, n_envs=2, n_rays=90, lib=oracle)
    try:
        assert host.roster_policies() == ROSTER
        assert [type(vs.driver).__module__ for vs in host.vehicle_states[:3]] == ["ft_grandprix.nidc", "ft_grandprix.fast", "ft_grandprix.nidc"]
        with np.errstate(all="ignore"):
            host.drive(250)
        dev.rollout("roster", 250)
        np.testing.assert_array_equal(host.env.progress(), dev.env.progress())
        np.testing.assert_allclose(host.env.pose(), dev.env.pose(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(host.env.ctrl(), dev.env.ctrl(), rtol=0, atol=1e-6)
        assert [vs.laps for vs in host.vehicle_states] == [vs.laps for vs in dev.vehicle_states]
    finally:
        host.close(); dev.close()


def test_roster_of_python_only_drivers_is_not_taken_to_the_device(oracle):
    from ft_grandprix_amd.sim import Simulator
    sim = Simulator(load_track("small-circle"), [{"driver": "ft_grandprix_amd.drivers.follow_gap", "name": "own"}], n_rays=36, lib=oracle)
    try:
        assert sim.roster_policies() is None
        with pytest.raises(ValueError):
            sim.rollout("roster", 5)
        sim.rollout("lobotomy", 5)
    finally:
        sim.close()


@pytest.mark.gpu
def test_gpu_per_car_dispatch(product):
    check_dispatch(product)
    check_dispatch(product, n_rays=1080)


@pytest.mark.gpu
@pytest.mark.parametrize("name,roster,rays", [("track", ROSTER, 1080), ("circle", ["fast", "random", "lobotomy", "nidc"], 90),
                                              ("inkscape", ["fast"], 360)])
def test_gpu_roster_rollout_matches_the_oracle(product, oracle, name, roster, rays):
    """Closed loop with the roster's drivers on the device, GPU against oracle: scans, controls, race state bit for bit; in one
    launch and step by step."""
    t = load_track(name)
    kw = dict(n_envs=24, cars_per_env=len(roster), n_rays=rays, spawn_mode=1, seed=5)
    with capi.Env(product, t, **kw) as g, capi.Env(oracle, t, **kw) as o:
        oracle.dll.oracle_set_threads(o.h, 8)
        g.set_car_policies(roster); o.set_car_policies(roster)
        for n in (1, 1, 60, 240):
            g.rollout("per_car", n); o.rollout("per_car", n)
            np.testing.assert_array_equal(g.lidar(), o.lidar())
            np.testing.assert_array_equal(g.ctrl(), o.ctrl())
            np.testing.assert_array_equal(g.progress(), o.progress())
            np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=1e-12)
        g.reset(); o.reset()                               # the roster survives a reset
        g.rollout("per_car", 30); o.rollout("per_car", 30)
        np.testing.assert_array_equal(g.lidar(), o.lidar())


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(n_rays=90, lidar_mode="fakelidar"), dict(n_rays=1081), dict(n_rays=36, lap_target=1)])
def test_gpu_roster_in_the_other_kernels_and_shapes(product, oracle, kw):
    """The roster through the FAKELIDAR kernel, with an odd ray count (no opposite-ray pairs), and to the end of a one-lap race
    (finished cars fall back to the null driver, custom.py:1446): GPU against oracle."""
    t = load_track("small-circle" if "lap_target" in kw else "track")
    kw = dict(n_envs=12, cars_per_env=3, spawn_mode=0, seed=2, **kw)
    with capi.Env(product, t, **kw) as g, capi.Env(oracle, t, **kw) as o:
        oracle.dll.oracle_set_threads(o.h, 8)
        g.set_car_policies(ROSTER); o.set_car_policies(ROSTER)
        for n in (1, 99, 1400 if "lap_target" in kw else 200):
            g.rollout("per_car", n); o.rollout("per_car", n)
            np.testing.assert_array_equal(g.lidar(), o.lidar())
            np.testing.assert_array_equal(g.ctrl(), o.ctrl())
            np.testing.assert_array_equal(g.progress(), o.progress())
            np.testing.assert_array_equal(g.winners(), o.winners())
        assert g.kernel_name().endswith(", true>")


@pytest.mark.gpu
def test_gpu_single_driver_roster_equals_the_single_policy_launch(product):
    """A roster of one driver is the single-policy launch: same kernel, same bits (and the cover table of the right driver)."""
    t = load_track("track")
    kw = dict(n_envs=16, n_rays=1080, spawn_mode=1, seed=9)
    for p in ("fast", "nidc"):
        with capi.Env(product, t, **kw) as a, capi.Env(product, t, **kw) as b:
            b.set_car_policies([p])
            a.rollout(p, 200); b.rollout("per_car", 200)
            np.testing.assert_array_equal(a.lidar(), b.lidar())
            np.testing.assert_array_equal(a.pose(), b.pose())
            np.testing.assert_array_equal(a.ctrl(), b.ctrl())


@pytest.mark.gpu
def test_gpu_runner_takes_the_reference_roster_to_the_device(product, tmp_path, capsys):
    """python -m ft_grandprix_amd.sim --cars <cars.json layout> --device-policy roster: the file's driver strings (module path and
    file:// form, custom.py:1097-1104) pick the device drivers; the race state equals a direct per-car rollout."""
    import json
    from ft_grandprix_amd import sim as simmod
    roster = [{"driver": "ft_grandprix.nidc", "name": "red car"}, {"driver": "file://ft_grandprix/fast.py", "name": "orange car"},
              {"driver": "ft_grandprix.nidc", "name": "green car"}]
    f = tmp_path / "cars.json"
    f.write_text(json.dumps(roster))
    assert simmod.main(["--cars", str(f), "--track", "track", "--steps", "600", "--envs", "2", "--rays", "90", "--device-policy", "roster"]) == 0
    out = capsys.readouterr().out
    assert "after 600 steps" in out and "red car" in out and "orange car" in out
    with capi.Env(product, load_track("track"), n_envs=2, cars_per_env=3, n_rays=90) as e:
        e.set_car_policies(ROSTER)
        e.rollout("per_car", 600)
        prog = e.progress()
    for label, row in zip(("red car", "orange car", "green car"), prog[:3]):
        line = next(x for x in out.splitlines() if label in x)
        assert f"laps {int(row[0]):3d}" in line


@pytest.mark.gpu
def test_gpu_full_size_roster_and_fakelidar_properties(product, oracle):
    """At BASELINE's batch size: the reference's roster (4096 envs x 3 cars, every car its own driver on the device) and the FAKELIDAR mode
    (1024 envs) -- launch-split invariance of everything, range sanity, the oracle on a prefix of the envs, the metrics record."""
    for name, kw, roster, steps, prefix in (
            ("track", dict(n_envs=4096, cars_per_env=3, n_rays=1080, spawn_mode=0, seed=1234, lap_target=3), ROSTER, 60, 4),
            ("circle", dict(n_envs=1024, cars_per_env=3, n_rays=360, spawn_mode=0, seed=7, lidar_mode="fakelidar"), ROSTER, 90, 8)):
        t = load_track(name)
        with capi.Env(product, t, **kw) as g, capi.Env(product, t, **kw) as g2, capi.Env(oracle, t, **dict(kw, n_envs=prefix)) as o:
            oracle.dll.oracle_set_threads(o.h, 8)
            for e in (g, g2, o):
                e.set_car_policies(roster)
            g.rollout("per_car", steps); g2.rollout("per_car", steps // 3); g2.rollout("per_car", steps - steps // 3); o.rollout("per_car", steps)
            r = g.lidar()
            np.testing.assert_array_equal(r, g2.lidar())
            np.testing.assert_array_equal(g.pose(), g2.pose())
            np.testing.assert_array_equal(g.ctrl(), g2.ctrl())
            np.testing.assert_array_equal(g.progress(), g2.progress())
            assert ((r == -1) | ((r >= 0) & (r < 60))).all()
            n = prefix * 3
            np.testing.assert_array_equal(r[:n], o.lidar())
            np.testing.assert_array_equal(g.ctrl()[:n], o.ctrl())
            np.testing.assert_array_equal(g.progress()[:n], o.progress())
            m = g.metrics_local()
            assert m[0] == kw["n_envs"] * steps and m[1] == g.n_cars
            # the two nidc cars and the fast car do not drive alike
            c = g.ctrl().reshape(kw["n_envs"], 3, 2)
            assert not np.array_equal(c[:, 0], c[:, 1])
