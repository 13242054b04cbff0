"""Host-side driver surface (ft_grandprix_amd.sim) on CPU, with the oracle standing in for the device library.

When the reference checkout is present (/root/reference, never on the GPU box) its OWN nidc / fast / template
drivers are loaded unmodified through roster strings and driven for a few hundred steps; the closed-loop result must
agree with the oracle's restated drivers (K5) run on the same worlds."""
import json
import os
import sys

import numpy as np
import pytest

from ft_grandprix_amd import capi
from ft_grandprix_amd.sim import Simulator, LobotomyDriver, resolve_driver_path, VehicleStateSnapshot
from ft_grandprix_amd.track import load_track

REF = "/root/reference"
have_ref = os.path.isdir(os.path.join(REF, "ft_grandprix"))


def test_roster_path_resolution_matches_reference_rules():
    assert resolve_driver_path("file://ft_grandprix/nidc.py") == "ft_grandprix.nidc"      # custom.py:1097-1098
    assert resolve_driver_path("ft_grandprix.fast") == "ft_grandprix.fast"                # custom.py:1099-1102
    assert resolve_driver_path("http://x/y.py") is None                                   # custom.py:1103-1104


def test_import_failure_falls_back_to_null_driver_and_arity_is_sniffed(oracle):
    import types

    class V2:
        def process_lidar(self, ranges, state):
            assert isinstance(state, VehicleStateSnapshot) and len(state.velocity) == 3
            return 1.0, 0.1
    m = types.ModuleType("ftgp_v2_driver"); m.Driver = V2; sys.modules["ftgp_v2_driver"] = m
    sim = Simulator(load_track("small-circle"), [{"driver": "ftgp_v2_driver", "name": "a"}, {"driver": "does.not.exist", "name": "b"}],
                    n_envs=2, n_rays=36, lib=oracle)
    assert [vs.v2 for vs in sim.vehicle_states] == [True, False, True, False]
    assert isinstance(sim.vehicle_states[1].driver, LobotomyDriver)
    assert [vs.offset for vs in sim.vehicle_states[:2]] == [10, 12]                       # (i + 5) * 2, custom.py:1112
    sim.drive(20)
    c = sim.env.ctrl()
    np.testing.assert_array_equal(c[0], [1.0, 0.1]); np.testing.assert_array_equal(c[1], [0.0, 0.0])
    snap = sim.snapshots()[0]
    assert snap.time == 20 / 0.004                                                       # steps / timestep (sic), custom.py:1397
    assert sim.ranking()[0] in (0, 2)
    sim.close()


@pytest.mark.skipif(not have_ref, reason="reference checkout not present")
@pytest.mark.parametrize("name,policy", [("ft_grandprix.nidc", "nidc"), ("file://ft_grandprix/fast.py", "fast")])
def test_unmodified_reference_drivers_close_the_loop(oracle, name, policy):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    t = load_track("track")
    cars = [{"driver": name, "name": "ref"}]
    sim = Simulator(t, cars, n_envs=3, n_rays=90, lib=oracle, spawn_mode=1, seed=9)
    assert type(sim.vehicle_states[0].driver).__module__.startswith("ft_grandprix.")
    with capi.Env(oracle, t, n_envs=3, n_rays=90, spawn_mode=1, seed=9) as dev:
        with np.errstate(all="ignore"):
            sim.drive(300)
        dev.rollout(policy, 300)
        # the Python drivers see float64 copies of the f32 scan; the restated K5 agrees to rounding, so the closed loops stay together
        np.testing.assert_array_equal(sim.env.progress(), dev.progress())
        np.testing.assert_allclose(sim.env.pose(), dev.pose(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(sim.env.lidar(), dev.lidar(), rtol=0, atol=1e-4)
    sim.close()


@pytest.mark.skipif(not have_ref, reason="reference checkout not present")
def test_reference_template_driver_runs_unmodified(oracle):
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sim = Simulator(load_track("small-circle"), [{"driver": "drivers.template", "name": "t"}], n_envs=1, n_rays=36, lib=oracle)
    assert sim.vehicle_states[0].v2
    sim.drive(50)
    np.testing.assert_array_equal(sim.env.ctrl(), 0.0)
    np.testing.assert_array_equal(sim.env.steps(), [50])
    sim.close()


def test_headless_runner_in_the_shape_of_drive_py(oracle, monkeypatch, capsys, tmp_path):
    """`python -m ft_grandprix_amd.sim --cars roster.json --track ... --steps N` (drive.py:69-115: roster + track in, loop out): a
    roster in the reference's layout (template/cars/cars.json), one driver of this package, one module that does not exist (null
    driver, custom.py:1106-1109); the CPU oracle stands in for the GPU library."""
    from ft_grandprix_amd import sim as simmod
    roster = tmp_path / "cars.json"
    roster.write_text(json.dumps([{"driver": "ft_grandprix_amd.drivers.follow_gap", "name": "gap"},
                                  {"driver": "file://no/such/driver.py", "name": "ghost"}]))
    monkeypatch.setattr(capi, "load", lambda: oracle)
    assert simmod.main(["--cars", str(roster), "--track", "circle", "--steps", "400", "--rays", "90", "--report", "200"]) == 0
    out = capsys.readouterr().out
    assert "-- step 200" in out and "-- after 400 steps (1.600 s of simulated time)" in out
    assert out.count("gap") >= 2 and out.count("ghost") >= 2
    # the car with a driver has moved along the track, the null driver's car has not
    lines = [l for l in out.splitlines()[-2:]]
    assert "gap" in lines[0] and "ghost" in lines[1]                  # ranking: best first
    # the same roster on the device policy: one launch, same surface
    assert simmod.main(["--cars", str(roster), "--track", "circle", "--steps", "300", "--device-policy", "nidc"]) == 0
    assert "-- after 300 steps" in capsys.readouterr().out


def test_follow_gap_driver_keeps_a_car_on_the_track(oracle):
    """The package's own example driver through the plugin surface: 1500 steps on `circle` without leaving the corridor."""
    t = load_track("circle")
    sim = Simulator(t, [{"driver": "ft_grandprix_amd.drivers.follow_gap", "name": "gap"}], n_envs=2, n_rays=90, lib=oracle, spawn_mode=1, seed=4)
    sim.drive(1500)
    assert not any(vs.off_track for vs in sim.vehicle_states)
    assert all(vs.absolute_completion() >= 5 for vs in sim.vehicle_states), [vs.absolute_completion() for vs in sim.vehicle_states]
    sim.close()


def test_lap_times_are_a_ring_of_the_newest_32(oracle):
    """VehicleState.times is unbounded in the reference (custom.py:124,1351-1363); here the newest FTGP_MAX_LAP_TIMES are kept in a
    ring (lap time k in slot k % 32) beside the true count.  40 forward laps driven through the lap logic alone."""
    import ctypes as C
    n_laps, per_lap = 40, 5
    closest = np.array([(5 + (0, 25, 50, 75, 99)[k % per_lap]) % 100 for k in range(n_laps * per_lap + 1)], dtype=np.int32)   # completion 0, 25, 50, 75, 99, 0, ...: 5 steps per lap
    off = np.zeros(len(closest), dtype=np.uint8)
    out = np.zeros((len(closest), 6), dtype=np.int32)
    ring = np.zeros(capi.MAX_LAP_TIMES)
    assert oracle.dll.oracle_progress_trace(5, 1000, 0.004, len(closest), closest.ctypes.data, off.ctypes.data, out.ctypes.data, ring.ctypes.data) == 0
    count = int(out[-1, 3])
    assert count == n_laps and int(out[-1, 0]) == n_laps                  # a car starts with good_start set (custom.py:113): every crossing appends
    times = capi.lap_time_list(count, ring)
    assert len(times) == capi.MAX_LAP_TIMES and all(abs(t - per_lap * 0.004) < 1e-12 for t in times)
    assert capi.lap_time_list(3, ring) == [ring[0], ring[1], ring[2]]


def _counting_driver(name, out):
    import types

    class D:
        calls = 0

        def process_lidar(self, ranges):
            D.calls += 1
            return out
    m = types.ModuleType(name); m.Driver = D; sys.modules[name] = m
    return D


def test_detach_control_runs_the_drivers_and_writes_no_ctrl(oracle):
    """Option "detach_control" (custom.py:952,1421-1423): process_lidar is still called, vehicle_state.speed / .steering_angle are
    set, data.ctrl stays what it was."""
    D = _counting_driver("ftgp_detached_driver", (2.0, 0.3))
    sim = Simulator(load_track("small-circle"), [{"driver": "ftgp_detached_driver", "name": "a"}], n_rays=36, lib=oracle)
    sim.drive(3)
    np.testing.assert_array_equal(sim.env.ctrl()[0], [2.0, 0.3])
    sim.env.set_ctrl(np.array([[0.5, -0.1]]))
    sim.detach_control = True
    before = D.calls
    sim.drive(5)
    assert D.calls == before + 5                                                         # drivers called ...
    np.testing.assert_array_equal(sim.env.ctrl()[0], [0.5, -0.1])                        # ... ctrl unchanged
    assert (sim.vehicle_states[0].speed, sim.vehicle_states[0].steering_angle) == (2.0, 0.3)
    sim.close()


def test_manual_control_overrides_the_watched_car_only(oracle):
    """Options "manual_control" / "always_invoke_driver" (custom.py:954-957,1403,1413-1416): the watched car takes the keyboard's
    controls -- coasting at 0.99 of its throttle when no key is held -- the others keep their drivers'; with always_invoke_driver
    off the drivers are not called at all and the unwatched cars get (0, 0)."""
    D = _counting_driver("ftgp_manual_driver", (1.5, 0.2))
    roster = [{"driver": "ftgp_manual_driver", "name": "a"}, {"driver": "ftgp_manual_driver", "name": "b"}]
    sim = Simulator(load_track("small-circle"), roster, n_rays=36, lib=oracle, manual_control=True)
    sim.watching, sim.manual_speed, sim.manual_steering_angle = 1, 3.0, -0.4                # custom.py:958: manual_control_speed = 3
    sim.step()
    c = sim.env.ctrl()
    np.testing.assert_array_equal(c[0], [1.5, 0.2]); np.testing.assert_array_equal(c[1], [3.0, -0.4])
    sim.manual_speed = 0.0                                                               # key released: 0.99 of the current throttle
    sim.step()
    np.testing.assert_allclose(sim.env.ctrl()[1], [3.0 * 0.99, -0.4], rtol=0, atol=1e-15)
    sim.always_invoke_driver = False
    before = D.calls
    sim.step()
    assert D.calls == before
    np.testing.assert_array_equal(sim.env.ctrl()[0], [0.0, 0.0])
    sim.close()


def test_rangefinder_tilt_turns_the_fakelidar_fan(oracle):
    """Option "rangefinder_tilt" (custom.py:986,1387): linspace(tilt + yaw + pi, yaw - pi, r): scan i is turned by tilt * (1 - i / r).
    tilt = 0 is the default fan bit for bit; a tilt of one ray spacing turns the rear ray onto its neighbour's direction."""
    from ft_grandprix_amd.sim import tilted_fan
    import math
    R = 36
    f0 = tilted_fan(R, 0.0)
    want = np.array([[math.sin(math.radians(360.0 / R * j - 90.0)), -math.cos(math.radians(360.0 / R * j - 90.0))] for j in range(R)])
    np.testing.assert_array_equal(f0, want)
    step = 2 * math.pi / R
    f1 = tilted_fan(R, step)
    np.testing.assert_allclose(f1[0], want[-1], rtol=0, atol=1e-15)                      # the rear ray: turned by the whole tilt, clockwise in the world
    np.testing.assert_allclose(f1[R // 2], [math.sin(math.radians(90.0) - step / 2), -math.cos(math.radians(90.0) - step / 2)], rtol=0, atol=1e-15)
    t = load_track("small-circle")
    a = Simulator(t, [{"driver": "ft_grandprix_amd.drivers.follow_gap", "name": "a"}], n_rays=R, lib=oracle, lidar_mode="fakelidar")
    b = Simulator(t, [{"driver": "ft_grandprix_amd.drivers.follow_gap", "name": "a"}], n_rays=R, lib=oracle, lidar_mode="fakelidar", rangefinder_tilt=step)
    a.step(); b.step(); a.step(); b.step()
    ra, rb = a.env.lidar()[0], b.env.lidar()[0]
    assert not np.array_equal(ra, rb) and abs(rb[0] - ra[-1]) < 0.05 * max(ra[-1], 1e-9) + 0.05   # b's rear ray looks where a's last ray looks
    a.close(); b.close()
