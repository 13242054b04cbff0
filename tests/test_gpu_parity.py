"""GPU parity: the HIP path through the C-ABI vs the CPU oracle and the reference-generated fixtures.

Bars (BASELINE.json north_star): bit-exact lap counters and race flags; LiDAR ranges and pose within 1e-4
over 1000 steps.  The specification fixes every rounding, so in practice ranges and poses are expected to
be bit-identical too; the asserts below state the contractual tolerance and additionally report exactness.
"""
import numpy as np
import pytest

from ft_grandprix_amd import capi
from ft_grandprix_amd.track import load_track
from tests.helpers import golden

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star tolerance for floating-point outputs


def both(product, oracle, track, **kw):
    g, o = capi.Env(product, track, **kw), capi.Env(oracle, track, **kw)
    oracle.dll.oracle_set_threads(o.h, 8)      # envs are independent: OpenMP over envs does not change results
    return g, o


def assert_same_state(g, o, exact=True):
    np.testing.assert_array_equal(g.progress(), o.progress())          # counters: bit-exact
    np.testing.assert_array_equal(g.steps(), o.steps())
    cg, tg = g.lap_times(); co, to = o.lap_times()
    np.testing.assert_array_equal(cg, co)
    np.testing.assert_allclose(tg, to, rtol=0, atol=1e-12)
    rg, ro = g.lidar(), o.lidar()
    np.testing.assert_allclose(rg, ro, rtol=0, atol=TOL)
    np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=TOL)
    np.testing.assert_allclose(g.ctrl(), o.ctrl(), rtol=0, atol=TOL)
    np.testing.assert_allclose(g.snapshot(), o.snapshot(), rtol=0, atol=TOL)
    if exact:
        np.testing.assert_array_equal(rg, ro)
        np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=1e-12)


def test_library_is_hip_and_device_present(product):
    assert product.fn("device_count")() >= 1


def test_device_selftest_fast_reciprocal_is_ieee_division(product):
    """The ray set-up computes |1/d| with v_rcp_f32 + one fused Newton step; the specification says IEEE division.
    ftgp_selftest compares the two over all 2^32 binary32 bit patterns on the device."""
    assert capi.selftest(product) == 0


@pytest.mark.parametrize("name", ["track", "circle", "small-circle", "inkscape"])
@pytest.mark.parametrize("R", [36, 1080])
def test_lidar_single_sweep_bit_exact(product, oracle, name, R):
    """K2 + K4: one sweep from seeded spawn poses on every track."""
    g, o = both(product, oracle, load_track(name), n_envs=96, n_rays=R, spawn_mode=1, seed=3)
    with g, o:
        np.testing.assert_array_equal(g.pose(), o.pose())            # K4 spawn incl. yaw jitter
        np.testing.assert_array_equal(g.lidar(), 0.0)                # all-zero scan right after reset
        g.step(1); o.step(1)
        rg, ro = g.lidar(), o.lidar()
        assert rg.shape == (96, R)
        np.testing.assert_array_equal(rg, ro)
        assert (rg >= 0).all() or (rg[rg < 0] == -1).all()


def test_lidar_f32_march_vs_f64_truth(product, oracle):
    """The f32 skip-march against the oracle's plain binary64 DDA: within 1e-4 except grazing rays."""
    t = load_track("track")
    g, o = both(product, oracle, t, n_envs=128, n_rays=1080, spawn_mode=1, seed=5)
    with g, o:
        oracle.dll.oracle_set_lidar_mode(o.h, 1)
        g.step(1); o.step(1)
        d = np.abs(g.lidar().astype(np.float64) - o.lidar())
        frac_bad = (d > TOL).mean()
        assert frac_bad < 2e-3, frac_bad                              # corner-grazing rays only
        assert np.median(d) < 1e-5


@pytest.mark.parametrize("policy", ["nidc", "fast", "random", "lobotomy"])
def test_closed_loop_1000_steps(product, oracle, policy):
    """K5 -> K2 -> K1 -> K3 for 1000 steps, device policy in the loop (configs 2/3 at oracle-sized batch)."""
    t = load_track("track" if policy != "nidc" else "circle")
    g, o = both(product, oracle, t, n_envs=48, n_rays=1080, spawn_mode=1, seed=1234, lap_target=2)
    with g, o:
        for chunk in (1, 7, 92, 400, 500):                           # uneven launch sizes: state carries across launches
            g.rollout(policy, chunk); o.rollout(policy, chunk)
            assert_same_state(g, o)
        assert g.steps()[0] == 1000


@pytest.mark.parametrize("R", [8, 37, 2000, 4100])
def test_odd_ray_counts_closed_loop(product, oracle, R):
    """Ray counts away from 1080: a window that is not a multiple of anything (37), the smallest the drivers accept (8),
    and scans long enough for more than 32 samples per lane in the driver passes (4100: 49 per lane) -- the workgroup
    shape, the LDS window layout and its float4 flush all depend on R."""
    t = load_track("circle")
    g, o = both(product, oracle, t, n_envs=20, n_rays=R, spawn_mode=1, seed=9, lap_target=2)
    with g, o:
        for policy, steps in (("fast", 150), ("nidc", 150)):
            g.rollout(policy, steps); o.rollout(policy, steps)
            assert_same_state(g, o)


def test_small_config_36_rays_template_driver(product, oracle):
    """Config 1: 1 env, small-circle, 36 rays, drivers.template (returns (0, 0)) through the host driver path."""
    from ft_grandprix_amd.sim import Simulator

    class Driver:                               # the v2 signature of drivers/template.py:2
        def process_lidar(self, ranges, state):
            assert ranges.shape == (36,) and ranges.dtype == np.float64
            assert state.time == state.time and hasattr(state, "absolute_completion")
            return 0, 0

    import sys, types
    mod = types.ModuleType("ftgp_test_template_driver"); mod.Driver = Driver
    sys.modules["ftgp_test_template_driver"] = mod
    t = load_track("small-circle")
    cars = [{"driver": "ftgp_test_template_driver", "name": "template"}]
    sg = Simulator(t, cars, n_envs=1, n_rays=36, lib=product)
    so = Simulator(t, cars, n_envs=1, n_rays=36, lib=oracle)
    sg.drive(100); so.drive(100)
    assert_same_state(sg.env, so.env)
    assert sg.vehicle_states[0].v2
    sg.close(); so.close()


def test_host_ctrl_path_with_python_driver(product, oracle):
    """ftgp_set_ctrl + ftgp_step(1) per iteration with a stateful Python driver; one driver raises."""
    from ft_grandprix_amd.sim import Simulator
    import sys, types

    class Follow:                               # v1 signature
        def __init__(self): self.k = 0
        def process_lidar(self, ranges):
            self.k += 1
            if self.k == 5: raise RuntimeError("boom")     # ctrl of this car must stay at its previous value
            n = len(ranges)
            front = ranges[3 * n // 8: 5 * n // 8]
            return 1.5, float(np.clip((np.argmax(front) - len(front) / 2) * 2 * np.pi / n, -0.5, 0.5))

    mod = types.ModuleType("ftgp_test_follow_driver"); mod.Driver = Follow
    sys.modules["ftgp_test_follow_driver"] = mod
    t = load_track("circle")
    cars = [{"driver": "ftgp_test_follow_driver", "name": "a"}, {"driver": "no.such.module", "name": "b"}]
    sg = Simulator(t, cars, n_envs=3, n_rays=90, lib=product)
    so = Simulator(t, cars, n_envs=3, n_rays=90, lib=oracle)
    assert type(sg.vehicle_states[1].driver).__name__ == "LobotomyDriver"
    sg.drive(150); so.drive(150)
    assert_same_state(sg.env, so.env)
    assert sg.env.pose()[0, 7:9].any()
    sg.close(); so.close()


@pytest.mark.parametrize("R", [36, 90, 1080])
def test_g1_device_policies_match_reference_drivers(product, R):
    """K5 against the outputs of the reference's own nidc / fast drivers (fixture g1_drivers.npz)."""
    gld = np.load(golden("g1_drivers.npz"))
    scans = gld[f"scans_{R}"]
    t = load_track("small-circle")
    with capi.Env(product, t, n_envs=len(scans), n_rays=R) as g:
        for name in ("nidc", "fast"):
            out = g.policy_eval(name, scans)
            ref = gld[f"{name}_{R}"]
            np.testing.assert_allclose(out, ref, rtol=0, atol=1e-6)
            # argmax index is exact when the steering angle is: (idx - m/2) * 2pi/R
            np.testing.assert_allclose(out[:, 1], ref[:, 1], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(g.policy_eval("lobotomy", scans), 0.0)


def test_disparity_threshold_edge_cases(product, oracle):
    """K5 decides |cur - prev| > 0.6 in binary32 and falls back to the reference's binary64 comparison only when the binary32
    difference rounds to the float next to 0.6: scans full of such pairs (both outcomes) against the oracle."""
    T = np.float32(0.60000002384185791015625)
    rng = np.random.default_rng(5)
    pairs = {True: [], False: []}
    while min(len(v) for v in pairs.values()) < 40:
        prev = np.float32(rng.uniform(0.01, 3.0))
        for k in range(-4, 5):
            cur = np.float32(prev + np.float32(0.6))
            for _ in range(abs(k)):
                cur = np.nextafter(cur, np.float32(np.inf if k > 0 else -np.inf))
            if np.float32(abs(np.float32(cur - prev))) == T:
                pairs[bool(abs(float(cur) - float(prev)) > 0.6)].append((prev, cur))
    assert all(len(v) >= 40 for v in pairs.values())
    R = 90
    eighth = R // 8
    scans = np.full((80, R), 2.5, dtype=np.float32)
    for row in range(80):
        prev, cur = pairs[row % 2 == 0][row // 2]
        at = eighth + 5 + (row * 7) % (R - 2 * eighth - 12)
        lo, hi = (prev, cur) if row % 4 < 2 else (cur, prev)
        scans[row, :at] = lo
        scans[row, at:] = hi
        scans[row, (at + 20) % R] = np.float32(9.0)            # somewhere to steer to
    t = load_track("small-circle")
    with capi.Env(product, t, n_envs=len(scans), n_rays=R) as g, capi.Env(oracle, t, n_envs=len(scans), n_rays=R) as o:
        for name in ("nidc", "fast"):
            np.testing.assert_array_equal(g.policy_eval(name, scans), o.policy_eval(name, scans))
        assert len(np.unique(g.policy_eval("nidc", scans)[:, 1])) > 4


def test_k5_wide_window_and_many_disparities(product, oracle):
    """K5 walks the front window 64 samples at a time and lists disparities 64 at a time: windows of more than 32 x 64 samples
    (64-bit flag words), scans with thousands of disparities (many list chunks, rebuilt from the kept flags), and windows
    that end in the middle of a 64-sample group -- against the oracle, exactly."""
    rng = np.random.default_rng(11)
    t = load_track("small-circle")
    for R in (4000, 1081, 200):
        scans = np.empty((12, R), dtype=np.float32)
        for row in range(12):
            base = rng.uniform(0.5, 6.0, R).astype(np.float32) if row % 3 == 0 else np.full(R, 3.0, dtype=np.float32)
            if row % 3 == 1:                                    # a few hundred steps of random height
                for at in rng.choice(R, size=min(R // 4, 300), replace=False):
                    base[at:] += np.float32(rng.choice([-0.7, 0.7, 1.3]))
                base = np.abs(base) + np.float32(0.2)
            if row % 3 == 2:                                    # smooth, one disparity at each end of the window
                e = R // 8
                base[e + 1:] += np.float32(1.0); base[R - e - 1:] += np.float32(1.0)
            if row % 4 == 3:
                base[rng.integers(0, R, 5)] = np.float32(-1.0)      # rays that hit nothing
            scans[row] = base
        with capi.Env(product, t, n_envs=len(scans), n_rays=R) as g, capi.Env(oracle, t, n_envs=len(scans), n_rays=R) as o:
            for name in ("nidc", "fast"):
                np.testing.assert_array_equal(g.policy_eval(name, scans), o.policy_eval(name, scans), err_msg=f"{name} R={R}")


@pytest.mark.parametrize("trace", ["forward", "reverse_start", "back_and_forth", "fast_jumps"])
def test_g5_progress_block_on_gpu(product, trace):
    """K3 against the reference's progress block executed verbatim (fixture g5_progress.npz)."""
    from tests.test_oracle_golden import replay_progress_trace
    replay_progress_trace(product, trace)


def test_lap_time_ring_on_gpu(product, oracle):
    """More laps than FTGP_MAX_LAP_TIMES: the true count and the ring of the newest 32 (lap time k in slot k % 32), GPU = oracle."""
    from tests.test_oracle_golden import laps_by_teleport
    cg, rg, pg = laps_by_teleport(product)
    co, ro, po = laps_by_teleport(oracle)
    assert cg == co == 37
    np.testing.assert_array_equal(rg, ro)
    np.testing.assert_array_equal(pg, po)


def test_g4_accessor_table_on_gpu(product):
    """Rows a5 / a6: lap_completion / absolute_completion of the reference's truth table (fixture G4) read back through
    ftgp_get_progress columns 2-3 and ftgp_get_snapshot columns 7-8."""
    from tests.test_oracle_golden import drive_g4_table
    drive_g4_table(product)


def test_forward_lap_by_driving(product, oracle):
    """A whole lap driven by the on-device nidc driver (circle, 90 rays, lap_target 1): the forward crossing appends the lap
    time inside the persistent kernel, `finished` is reached by driving, finished cars get the null driver and their
    rangefinders are switched off -- GPU vs oracle, uneven launch sizes."""
    t = load_track("circle")
    g, o = both(product, oracle, t, n_envs=16, n_rays=90, spawn_mode=1, seed=7, lap_target=1)
    with g, o:
        for chunk in (1, 4999, 5000, 3333, 6667):
            g.rollout("nidc", chunk); o.rollout("nidc", chunk)
            assert_same_state(g, o)
        p = g.progress()
        cnt, times = g.lap_times()
        fin = p[:, 4] == 1
        assert fin.sum() >= 8                                          # most cars drive the lap forwards ...
        assert (p[fin, 0] == 1).all() and (cnt[fin] == 1).all()        # ... laps + 1, one lap time, custom.py:1357-1366
        assert ((times[fin, 0] > 20) & (times[fin, 0] < 80)).all()     # 83 track units at <= 1.5 units/s
        assert (p[~fin, 0] < 0).any()                                  # the others lap backwards (negative laps, custom.py:1352-1356)
        np.testing.assert_array_equal(g.lidar()[fin], 0.0)             # shadow_rangefinders, custom.py:1436-1439
        np.testing.assert_array_equal(g.ctrl()[fin], 0.0)              # LobotomyDriver, custom.py:1446
        assert g.lidar()[~fin].any()


def test_config4_shard_shape_random_policy(product, oracle):
    """BASELINE.json configs[3] as one rank sees it: a 4096-env shard at env_base = 7 * 4096 (rank 7 of 8) under the
    counter-based random policy, against an oracle prefix of the same shard."""
    t = load_track("track")
    kw = dict(n_rays=1080, spawn_mode=1, seed=1234, env_base=7 * 4096)
    with capi.Env(product, t, n_envs=4096, **kw) as g, capi.Env(oracle, t, n_envs=24, **kw) as o:
        oracle.dll.oracle_set_threads(o.h, 8)
        g.rollout("random", 120); o.rollout("random", 120)
        np.testing.assert_array_equal(g.lidar()[:24], o.lidar())
        np.testing.assert_array_equal(g.progress()[:24], o.progress())
        np.testing.assert_array_equal(g.ctrl()[:24], o.ctrl())           # same draws: keyed (seed, GLOBAL car index, step)
        np.testing.assert_allclose(g.pose()[:24], o.pose(), rtol=0, atol=1e-12)
        with capi.Env(product, t, n_envs=24, n_rays=1080, spawn_mode=1, seed=1234) as g0:
            g0.rollout("random", 120)
            assert (g0.ctrl() != g.ctrl()[:24]).any()                   # a different shard draws different controls


def test_bubble_wrap_softeners(product, oracle):
    """f-3: option bubble_wrap (custom.py:1041-1055) adds the four wheel softeners to the wall-contact set; naive_flatten
    (custom.py:1338-1339) is accepted and changes nothing on a planar model."""
    t = load_track("track")
    kw = dict(n_envs=64, n_rays=90, spawn_mode=1, seed=11)
    g, o = both(product, oracle, t, bubble_wrap=True, **kw)
    with g, o, capi.Env(product, t, **kw) as plain, capi.Env(product, t, naive_flatten=True, **kw) as flat:
        for e in (g, o, plain, flat):
            e.rollout("random", 1500)                                  # random controls: plenty of wall contacts
        assert_same_state(g, o)
        assert (np.abs(g.pose() - plain.pose()) > 1e-6).any()          # the softeners did touch something
        np.testing.assert_array_equal(flat.pose(), plain.pose())
        np.testing.assert_array_equal(flat.lidar(), plain.lidar())


def test_enlarged_vehicle_box_is_seen(product, oracle):
    """The conservative cull before the inter-vehicle ray tests derives its radius from the vehicle's own extents: a car
    with a chassis box three times the default must still be seen exactly as the oracle sees it."""
    t = load_track("track")
    v = product.default_vehicle()
    v.box_xmin, v.box_xmax, v.box_ymin, v.box_ymax = -0.31, 0.31, -0.14, 0.14
    kw = dict(n_envs=12, cars_per_env=4, n_rays=1080, spawn_mode=0, lap_target=3, vehicle=v)
    g, o = both(product, oracle, t, **kw)
    with g, o, capi.Env(oracle, t, **dict(kw, vehicle=None)) as small:
        g.step(1); o.step(1); small.step(1)
        np.testing.assert_array_equal(g.lidar(), o.lidar())
        assert (g.lidar() != small.lidar()).any()
        g.rollout("fast", 200); o.rollout("fast", 200)
        assert_same_state(g, o)


@pytest.mark.parametrize("n_rays,envs", [(130, 20), (200, 37), (333, 9), (1000, 24), (1084, 70)])
def test_sweep_task_table_for_other_fans(product, oracle, n_rays, envs):
    """The sweep's task descriptors (DeviceParams::task_tab) for ray counts other than 1080: a last pair of two rays in one group (130), a partial
    last pair and window ends inside groups of either half (200), an odd count -- single groups, no pairs -- (333), a half that is no multiple of
    anything (1000, 1084); batches that leave a ragged last workgroup.  Closed loops of both device drivers -- they read the scan windows the
    deliveries fill -- against the oracle, bit for bit."""
    t = load_track("track")
    for policy in ("nidc", "fast"):
        g, o = both(product, oracle, t, n_envs=envs, n_rays=n_rays, spawn_mode=1, seed=n_rays)
        with g, o:
            g.step(1); o.step(1)
            np.testing.assert_array_equal(g.lidar(), o.lidar())
            g.rollout(policy, 400); o.rollout(policy, 400)
            assert_same_state(g, o)


@pytest.mark.parametrize("how", ["puck_outside_the_box", "switch"])
def test_inter_vehicle_test_with_the_pucks_circle(product, oracle, how, monkeypatch):
    """The inter-vehicle ray test leaves the LiDAR puck's circle out when it lies inside the chassis box (both bundled vehicles: the box's time
    is then the minimum to the bit).  The circle's code path stays what the oracle computes: a vehicle whose puck sticks out of its box behind
    (so that rays from behind meet the puck first), and the bundled vehicle with the test forced on (FTGP_PUCK_TEST=1)."""
    t = load_track("track")
    v = product.default_vehicle()
    if how == "switch":
        monkeypatch.setenv("FTGP_PUCK_TEST", "1")
    else:
        v.box_xmin = -0.06                       # the puck spans x in [-0.0825, -0.0225]
    kw = dict(n_envs=12, cars_per_env=4, n_rays=1080, spawn_mode=0, lap_target=3, vehicle=v)
    g, o = both(product, oracle, t, **kw)
    with g, o, capi.Env(oracle, t, **dict(kw, vehicle=None)) as plain:
        g.step(1); o.step(1); plain.step(1)
        np.testing.assert_array_equal(g.lidar(), o.lidar())
        if how != "switch":
            assert (g.lidar() != plain.lidar()).any()                  # the shortened box is seen as such (rays from behind now end on the puck)
        g.rollout("fast", 200); o.rollout("fast", 200)
        assert_same_state(g, o)


def test_user_track_from_png_and_svg_on_gpu(product, oracle, tmp_path):
    """f-2 end to end: a generated PNG + SVG centre-line (H V L A Q C S commands) in the reference's template layout ->
    load_track_from_template -> ftgp_create -> sweep and closed loop, GPU vs oracle (a 640 x 480 image: 32 x 24 chunks)."""
    from ft_grandprix_amd.track import load_track_from_template
    from tests.helpers import write_template_track
    write_template_track(str(tmp_path), "generated")
    t = load_track_from_template(str(tmp_path), "generated")
    g, o = both(product, oracle, t, n_envs=40, n_rays=1080, spawn_mode=1, seed=3, lap_target=2)
    with g, o:
        g.step(1); o.step(1)
        np.testing.assert_array_equal(g.lidar(), o.lidar())
        assert (g.lidar() > 0).mean() > 0.99
        g.rollout("fast", 500); o.rollout("fast", 500)
        assert_same_state(g, o)


@pytest.mark.parametrize("cars", [1, 3])
def test_tricycle_vehicle_closed_loop(product, oracle, cars):
    """f-4: the legacy tricycle of template/car.em.xml (torque motors on two driven wheels, dt 0.0075) as a second vehicle kind:
    host controls, the random policy and the on-device drivers, GPU vs oracle."""
    t = load_track("track")
    v = product.tricycle_vehicle()
    g, o = both(product, oracle, t, n_envs=24 // cars, cars_per_env=cars, n_rays=90, spawn_mode=1 if cars == 1 else 0, seed=4, dt=0.0075, vehicle=v)
    with g, o:
        rng = np.random.default_rng(1)
        ctrl = np.stack([rng.uniform(-1, 5, g.n_cars), rng.uniform(-1.5, 1.5, g.n_cars)], axis=1)
        g.set_ctrl(ctrl); o.set_ctrl(ctrl)
        g.step(300); o.step(300)
        assert_same_state(g, o)
        assert np.abs(g.pose()[:, 7:9]).max() > 0.2
        for policy, steps in (("random", 300), ("nidc", 300)):
            g.rollout(policy, steps); o.rollout(policy, steps)
            assert_same_state(g, o)


def test_multi_car_env_config5(product, oracle):
    """Config 5: 4 cars per env share a world -- inter-vehicle rays and car-car contact."""
    t = load_track("track")
    g, o = both(product, oracle, t, n_envs=24, cars_per_env=4, n_rays=1080, spawn_mode=0, lap_target=3)
    with g, o:
        g.step(1); o.step(1)
        rg = g.lidar()
        np.testing.assert_array_equal(rg, o.lidar())
        # reference spawn (custom.py:1112): cars are 2 path points apart, so somebody sees a neighbour
        single = capi.Env(oracle, t, n_envs=1, cars_per_env=1, n_rays=1080)
        single.step(1)
        assert (np.abs(rg[0] - single.lidar()[0]) > 1e-3).any()
        single.close()
        for chunk in (99, 400):
            g.rollout("fast", chunk); o.rollout("fast", chunk)
            assert_same_state(g, o)


def test_masked_reset(product, oracle):
    t = load_track("circle")
    g, o = both(product, oracle, t, n_envs=16, n_rays=90, spawn_mode=1)
    with g, o:
        g.rollout("nidc", 60); o.rollout("nidc", 60)
        mask = np.zeros(16, dtype=np.uint8); mask[[1, 5, 15]] = 1
        g.reset(mask); o.reset(mask)
        assert_same_state(g, o)
        st = g.steps()
        assert (st[mask == 1] == 0).all() and (st[mask == 0] == 60).all()
        assert (g.lidar()[5] == 0).all() and g.lidar()[4].any()
        g.rollout("nidc", 40); o.rollout("nidc", 40)
        assert_same_state(g, o)


def test_metrics_record(product, oracle):
    t = load_track("track")
    g, o = both(product, oracle, t, n_envs=40, n_rays=90, spawn_mode=1, lap_target=1)
    with g, o:
        g.rollout("fast", 700); o.rollout("fast", 700)
        mg, mo = g.metrics_local(), o.metrics_local()
        np.testing.assert_array_equal(mg[:6], mo[:6])
        np.testing.assert_allclose(mg[6:], mo[6:], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(g.metrics_allgather()[0][:6], mo[:6])     # world_size 1: no communicator needed


def test_headline_size_properties_and_oracle_prefix(product, oracle):
    """Config 3 at full size (4096 envs x 1080 rays, fast driver): size-independent properties + oracle on a prefix.

    Envs are independent and env e's spawn depends only on (e, seed), so an oracle batch of the first 32 envs
    must equal the first 32 envs of the 4096-env GPU batch.
    """
    t = load_track("track")
    kw = dict(n_rays=1080, spawn_mode=1, seed=1234)
    with capi.Env(product, t, n_envs=4096, **kw) as g, capi.Env(product, t, n_envs=4096, **kw) as g2, \
            capi.Env(oracle, t, n_envs=32, **kw) as o:
        g.rollout("fast", 200); g2.rollout("fast", 120); g2.rollout("fast", 80); o.rollout("fast", 200)
        r = g.lidar()
        # determinism / launch-split invariance: checksum of checksums
        np.testing.assert_array_equal(r, g2.lidar())
        np.testing.assert_array_equal(g.pose(), g2.pose())
        np.testing.assert_array_equal(g.progress(), g2.progress())
        # range sanity: a hit is non-negative and shorter than the map diagonal, a miss is exactly -1
        assert ((r == -1) | ((r >= 0) & (r < 60))).all()
        # oracle prefix
        np.testing.assert_array_equal(r[:32], o.lidar())
        np.testing.assert_array_equal(g.progress()[:32], o.progress())
        np.testing.assert_allclose(g.pose()[:32], o.pose(), rtol=0, atol=TOL)
        assert g.last_kernel_ms() > 0


@pytest.mark.parametrize("R,cars", [(1080, 1), (90, 3), (200, 1), (130, 2)])
def test_custom_fan_takes_the_unpaired_sweep(product, oracle, R, cars):
    """FtgpConfig.fan_dirs: a caller's fan is not point-symmetric to the bit, so the sweep cannot derive a ray from its opposite and
    marches every group on its own (the rangefinders' own table is built symmetric and takes the paired path: every other test).
    Both against the oracle, closed loop; with the default fan the two paths must agree with each other as well (FTGP_NO_PAIRS)."""
    import os
    t = load_track("track")
    ang = np.deg2rad(360.0 / R * np.arange(R) - 90.0 + 0.37)               # the MJCF's sites turned by 0.37 degrees ...
    ang[R // 2:] += 3e-5                                                   # ... and the second half a little further: no ray has an exact opposite
    fan = np.stack([np.sin(ang), -np.cos(ang)], axis=1)
    assert not np.array_equal(fan[R // 2:].astype(np.float32), -fan[: R - R // 2].astype(np.float32))
    kw = dict(n_envs=12, cars_per_env=cars, n_rays=R, spawn_mode=1, seed=3)
    with capi.Env(product, t, fan_dirs=fan, **kw) as g, capi.Env(oracle, t, fan_dirs=fan, **kw) as o:
        for n in (1, 40, 300):
            g.rollout("nidc", n); o.rollout("nidc", n)
            np.testing.assert_array_equal(g.lidar(), o.lidar())
            np.testing.assert_array_equal(g.progress(), o.progress())
            np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=1e-12)
    os.environ["FTGP_NO_PAIRS"] = "1"
    try:
        single = capi.Env(product, t, **kw)
    finally:
        del os.environ["FTGP_NO_PAIRS"]
    with single, capi.Env(product, t, **kw) as paired:
        single.rollout("fast", 250); paired.rollout("fast", 250)
        np.testing.assert_array_equal(single.lidar(), paired.lidar())
        np.testing.assert_array_equal(single.pose(), paired.pose())


@pytest.mark.parametrize("sectors", [8, 16, 32, 64])
def test_results_do_not_depend_on_the_sector_count(product, oracle, sectors):
    """ftgp_create picks the number of direction sectors of the box field by batch size (8 / 16 for large batches, 64 for small ones);
    FTGP_SECTORS_RT forces it.  Any choice must give the specification's ranges: closed loop against the oracle, two tracks (square and
    stretched pixels), single- and multi-car."""
    import os
    os.environ["FTGP_SECTORS_RT"] = str(sectors)
    try:
        for name, cars, policy in (("track", 1, "fast"), ("inkscape", 3, "nidc")):
            t = load_track(name)
            g, o = both(product, oracle, t, n_envs=24, cars_per_env=cars, n_rays=1080, spawn_mode=1, seed=sectors)
            with g, o:
                for n in (1, 120, 400):
                    g.rollout(policy, n); o.rollout(policy, n)
                    np.testing.assert_array_equal(g.lidar(), o.lidar())
                    np.testing.assert_array_equal(g.progress(), o.progress())
                    np.testing.assert_allclose(g.pose(), o.pose(), rtol=0, atol=1e-12)
    finally:
        del os.environ["FTGP_SECTORS_RT"]


def test_lap_time_ring_pop_beyond_the_ring_on_gpu(product):
    """The lap-time list beyond the ring's size with a backward crossing (times.pop()) -- the slot the popped entry had overwritten reads
    NaN, not the popped time -- against a Python list, and the 64-bit race steps (ftgp_get_race_steps) beside the int32 row."""
    from tests.test_oracle_golden import check_laps_with_a_pop
    check_laps_with_a_pop(product)
    t = load_track("track")
    with capi.Env(product, t, n_envs=1, n_rays=8, lap_target=2) as g:        # three laps by teleport, five steps each (laps_by_teleport)
        pose = g.pose(); pose[:, 7:] = 0.0
        for k in range(15):
            pose[0, 0:2] = t.path[(10 + (25, 50, 75, 99, 0)[k % 5]) % 100]
            g.set_pose(pose); g.step(1)
        p, rs = g.progress(), g.race_steps()
        assert p[0, 0] == 3 and p[0, 4] == 1
        assert rs[0, 1] == 10 == p[0, 9]                 # `finished` was set when the second lap was counted: step 10
        assert rs[0, 0] == 15 == p[0, 6]                 # vehicle_state.start: the last counted crossing
    with capi.Env(product, t, n_envs=2, n_rays=8) as g:
        np.testing.assert_array_equal(g.race_steps(), [[0, -1], [0, -1]])


def test_launch_with_recorded_events_gives_the_same_results(product):
    """FTGP_LAUNCH_PLAIN=1 (hipEventRecord around the launch instead of events on the dispatch packet): same results, and a kernel
    time of the same order."""
    import os
    t = load_track("track")
    kw = dict(n_envs=64, n_rays=1080, spawn_mode=1, seed=4)
    with capi.Env(product, t, **kw) as a:
        a.rollout("fast", 150); ms_a = a.last_kernel_ms()
        os.environ["FTGP_LAUNCH_PLAIN"] = "1"
        try:
            with capi.Env(product, t, **kw) as b:
                b.rollout("fast", 150); ms_b = b.last_kernel_ms()
                np.testing.assert_array_equal(a.lidar(), b.lidar())
                np.testing.assert_array_equal(a.pose(), b.pose())
                np.testing.assert_array_equal(a.metrics_local(), b.metrics_local())
        finally:
            del os.environ["FTGP_LAUNCH_PLAIN"]
    assert 0.0 < ms_a < 10 * ms_b and ms_b < 10 * ms_a


def test_gpu_shards_reproduce_the_monolithic_batch(product):
    """a shard [env_base, env_base + n) must equal the same slice of the whole batch (two handles, one GPU)."""
    from ft_grandprix_amd import dist as ftdist
    t = load_track("track")
    kw = dict(n_rays=1080, spawn_mode=1, seed=1234)
    with capi.Env(product, t, n_envs=96, **kw) as mono:
        mono.rollout("random", 150)
        recs = []
        for rank in range(2):
            with ftdist.make_shard(product, t, 96, rank, 2, **kw) as sh:
                start, count = ftdist.shard_range(96, rank, 2)
                sh.rollout("random", 150)
                np.testing.assert_array_equal(sh.lidar(), mono.lidar()[start:start + count])
                np.testing.assert_array_equal(sh.pose(), mono.pose()[start:start + count])
                recs.append(sh.metrics_local())
        tot, ref = ftdist.reduce_metrics(np.stack(recs)), ftdist.reduce_metrics(mono.metrics_local()[None])
        assert tot == {**ref, "ranks": 2}


def test_bench_two_ranks_rccl_refusal_is_loud_and_labelled(product):
    """`bench.py --gpus 2` on the box's one GPU WITHOUT the rehearsal switch: RCCL refuses a two-rank communicator on one device
    (ncclCommInitRank returns "invalid usage"), every rank learns of it, and the metrics record travels over the host rendezvous
    instead -- said on stderr and in the JSON line; with FTGP_BENCH_RCCL_REQUIRED=1 the same refusal is fatal (rc 3)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FTGP_BENCH_COLLECTIVE")}
    env["FTGP_RCCL_INIT_TIMEOUT"] = "90"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--envs-per-gpu", "64", "--rays", "90"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["metrics_allgather"]["ranks"] == 2
    if product.fn("device_count")() >= 2:            # two ranks, two devices: RCCL forms the communicator and there is nothing to refuse
        assert d["metrics_allgather"]["collective"].startswith("rccl")
        return
    assert "could not be set up" in p.stderr and "falling back to the host TCP gather" in p.stderr
    assert d["metrics_allgather"]["collective"].startswith("host TCP gather -- FALLBACK: ncclCommInitRank failed")
    env["FTGP_BENCH_RCCL_REQUIRED"] = "1"
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 3 and "could not be set up" in p.stderr


def test_rccl_communicator_single_rank(product):
    """C1 through the real RCCL entry points (ncclGetUniqueId / ncclCommInitRank / ncclAllGather) with world_size 1."""
    t = load_track("circle")
    with capi.Env(product, t, n_envs=8, n_rays=36, spawn_mode=1) as g:
        uid = capi.comm_unique_id(product)
        assert len(uid) == 128 and any(uid)
        g.comm_init(uid, 0, 1)
        g.rollout("nidc", 40)
        rec = g.metrics_allgather()
        assert rec.shape == (1, capi.METRIC_DOUBLES)
        np.testing.assert_array_equal(rec[0], g.metrics_local())
        assert rec[0][0] == 8 * 40 and rec[0][1] == 8


@pytest.mark.parametrize("with_comm", [False, True])
def test_metrics_exchange_in_two_halves_beside_the_next_launch(product, with_comm):
    """ftgp_metrics_allgather_begin / _end (side stream, overlapped with the next step kernel), with a one-rank RCCL
    communicator (the real ncclAllGather on the side stream) and without one: the exchange begun behind launch k is collected
    after launch k + 1 -- even k + 2 -- has been enqueued and still delivers launch k's record; this rank's own record
    (ftgp_metrics_local) is never overwritten by gathered ones (rank > 0 read rank 0's record from the shared buffer)."""
    t = load_track("circle")
    with capi.Env(product, t, n_envs=8, n_rays=36, spawn_mode=1) as g, capi.Env(product, t, n_envs=8, n_rays=36, spawn_mode=1) as twin:
        if with_comm:
            g.comm_init(capi.comm_unique_id(product), 0, 1)
        g.rollout("nidc", 40); twin.rollout("nidc", 40)
        g.metrics_allgather_begin()
        with pytest.raises(capi.FtgpError) as ei:
            g.metrics_allgather_begin()                  # one exchange at a time
        assert ei.value.code == -4
        g.rollout("nidc", 30)                            # launch k + 1: the other record slot
        g.rollout("nidc", 30)                            # launch k + 2: launch k's slot again -- waits on the device for the open exchange
        rec = g.metrics_allgather_end()
        np.testing.assert_array_equal(rec[0], twin.metrics_local())          # launch k's record (40 steps), not a later one
        assert rec[0][0] == 8 * 40
        with pytest.raises(capi.FtgpError):
            g.metrics_allgather_end()                    # nothing open
        twin.rollout("nidc", 60)
        np.testing.assert_array_equal(g.metrics_local(), twin.metrics_local())             # the newest state: 100 steps
        np.testing.assert_array_equal(g.metrics_allgather()[0], twin.metrics_local())      # begin + end in one call
        np.testing.assert_array_equal(g.metrics_local(), twin.metrics_local())             # ... which leaves this rank's record alone
        g.reset(); twin.reset()                          # no launch has produced this state's record: the metrics kernel does
        g.metrics_allgather_begin()
        np.testing.assert_array_equal(g.metrics_allgather_end()[0], twin.metrics_local())
        np.testing.assert_array_equal(g.lidar(), twin.lidar())
        # an exchange promises the record of the state at its begin, whatever refreshes the slot before its end: begin ->
        # reset -> ftgp_metrics_local (the metrics kernel writes the post-reset record) -> end still delivers the pre-reset one
        g.rollout("nidc", 25); twin.rollout("nidc", 25)
        before = twin.metrics_local()
        g.metrics_allgather_begin()
        g.reset(); twin.reset()
        np.testing.assert_array_equal(g.metrics_local(), twin.metrics_local())             # the post-reset state: steps 0
        assert g.metrics_local()[0] == 0 and before[0] == 8 * 25
        np.testing.assert_array_equal(g.metrics_allgather_end()[0], before)


def test_bench_repeats_median_and_overlap_fields(product):
    """The bench line of a short launch: several launches timed one by one, the median reported, `repeats` in the line, the
    roofline's `bound` derived from the committed counters with both fractions side by side."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["repeats"] >= 5 and d["repeats"] % 2 == 1 and d["steps"] == 20
    assert d["ms_per_step_best"] <= d["ms_per_step"] <= d["ms_per_step_worst"]
    assert d["value"] == pytest.approx(4096 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    r = d["roofline"]
    assert r["frac"] == r["frac_hbm"] == pytest.approx(r["achieved"] / 8000.0)
    assert r["bound"] in ("hbm", "valu-issue") and (r["bound"] == "hbm" or r["frac_valu_pipe_lower"] >= 0.5)
    # nothing that calls itself a fraction exceeds 1 (round 4 shipped a "busy" figure of 1.03: a counter that ticks per instruction, times 4)
    fracs = {k: v for k, v in r.items() if k.startswith("frac") and isinstance(v, float)}
    fracs.update({"valu." + k: v for k, v in r.get("valu", {}).items() if ("frac" in k or "occupancy" in k) and isinstance(v, float)})
    assert fracs and all(0.0 <= v <= 1.0 for v in fracs.values()), fracs
    pr = d["per_rank"]
    assert len(pr["kernel_ms_median"]) == 1 and pr["kernel_ms_min"][0] <= pr["kernel_ms_median"][0] <= pr["kernel_ms_max"][0]
    assert d["metrics_allgather"]["end_wait_ms_max"] >= d["metrics_allgather"]["end_wait_ms_median"] >= 0.0
    assert r["kernel_ms_per_launch"] * d["repeats"] < 1e3
    assert d["metrics_allgather"]["sum_steps"] == 4096 * (5 + 20 * d["repeats"])       # the record collected last is the last launch's


@pytest.mark.parametrize("name", ["track", "circle", "small-circle", "inkscape"])
def test_g2_fakelidar_on_gpu_bit_exact(product, name):
    """Row a3: the fakelidar-compat kernel against the outputs of the reference's own raycast.fakelidar (fixture G2)."""
    from scipy.ndimage import distance_transform_edt
    gld = np.load(golden("g2_fakelidar.npz"))
    t = load_track(name)
    dt = distance_transform_edt(~t.wall_mask())
    origins = gld[f"{name}_origins"]
    for R in (36, 1080):
        ang = gld[f"{name}_{R}_angles"]
        scan, pts = capi.fakelidar(product, dt, origins, np.cos(ang), np.sin(ang), eps=2.0)
        np.testing.assert_array_equal(scan, gld[f"{name}_{R}_scan"])
        np.testing.assert_array_equal(pts, gld[f"{name}_{R}_points"])
    # a ray that leaves through the right edge is the reference's IndexError
    with pytest.raises(capi.FtgpError):
        capi.fakelidar(product, np.full((64, 64), 10.0), [[60.0, 32.0]], [[1.0]], [[0.0]])


def test_adversarial_rays_through_pixel_corners(product, oracle):
    """LiDAR centres on exact pixel corners / centres and headings at multiples of 45 degrees, in the pixel-aligned
    frame: many rays run along pixel boundaries or through corners (ties of the DDA).  GPU vs the plain-DDA spec."""
    import dataclasses
    t0 = load_track("track")
    t = dataclasses.replace(t0, px_size_x=0.025, px_size_y=0.025, origin_x=0.0, origin_y=0.0)   # the 'pixel' frame
    n = 64
    g, o = both(product, oracle, t, n_envs=n, n_rays=1080, spawn_mode=1, seed=2)
    with g, o:
        oracle.dll.oracle_set_lidar_mode(o.h, 2)            # the specification itself (plain DDA)
        pose = o.pose()
        rng = np.random.default_rng(0)
        yaw = (np.pi / 4) * rng.integers(0, 8, n)
        px = np.floor(pose[:, 0] / 0.025) + rng.choice([0.0, 0.5], n)
        py = np.floor(-pose[:, 1] / 0.025) + rng.choice([0.0, 0.5], n)
        # place the LiDAR centre (body x = -0.0525) on the chosen pixel coordinate
        pose[:, 0] = px * 0.025 + 0.0525 * np.cos(yaw)
        pose[:, 1] = -py * 0.025 + 0.0525 * np.sin(yaw)
        pose[:, 3], pose[:, 6] = np.cos(yaw / 2), np.sin(yaw / 2)
        pose[:, 7:] = 0
        g.set_pose(pose); o.set_pose(pose)
        g.step(1); o.step(1)
        np.testing.assert_array_equal(g.lidar(), o.lidar())


@pytest.mark.parametrize("name", ["track", "inkscape"])
def test_lidar_from_anywhere_on_the_image(product, oracle, name):
    """LiDAR centres anywhere on the image -- on walls, off the track, at the border, a few off the image -- and headings
    that make direction components exactly zero: the GPU march against the plain-DDA specification, bit for bit."""
    t = load_track(name)
    n = 256
    g, o = both(product, oracle, t, n_envs=n, n_rays=360, spawn_mode=1, seed=11)
    with g, o:
        oracle.dll.oracle_set_lidar_mode(o.h, 2)            # the specification itself (plain DDA)
        pose = o.pose()
        rng = np.random.default_rng(17)
        w, h = t.width * t.px_size_x, t.height * t.px_size_y
        u = rng.uniform(-0.02, 1.02, n); v = rng.uniform(-0.02, 1.02, n)
        u[:16] = rng.choice([0.0, 1.0], 16) + rng.uniform(-1e-3, 1e-3, 16)         # hugging the left / right border
        yaw = rng.uniform(-np.pi, np.pi, n)
        yaw[16:48] = (np.pi / 2) * rng.integers(-2, 3, 32)                          # axis-aligned cars
        pose[:, 0] = t.origin_x + u * w
        pose[:, 1] = t.origin_y - v * h
        pose[:, 3], pose[:, 6] = np.cos(yaw / 2), np.sin(yaw / 2)
        pose[:, 7:] = 0
        g.set_pose(pose); o.set_pose(pose)
        g.step(1); o.step(1)
        rg, ro = g.lidar(), o.lidar()
        np.testing.assert_array_equal(rg, ro)
        assert (rg == -1).any() and (rg == 0).any() and (rg > 1.0).any()           # off-image starts, starts on walls, long rays


def test_large_image_4000x3000(product, oracle):
    """A 4000 x 3000 synthetic oval (1 cm pixels: the LiDAR ring spans 6 pixels, eps = 2^-9): one sweep against the
    plain-DDA specification and a closed loop against the oracle, bit for bit."""
    from ft_grandprix_amd.track import synthetic_oval
    t = synthetic_oval(width=4000, height=3000, half_width_px=70.0, wall_px=2.0)
    g, o = both(product, oracle, t, n_envs=48, n_rays=360, spawn_mode=1, seed=4)
    with g, o:
        oracle.dll.oracle_set_lidar_mode(o.h, 2)            # the specification itself (plain DDA)
        g.step(1); o.step(1)
        rg = g.lidar()
        np.testing.assert_array_equal(rg, o.lidar())
        assert (rg > 0).mean() > 0.9 and rg.max() > 5.0
        oracle.dll.oracle_set_lidar_mode(o.h, 0)
        g.rollout("nidc", 150); o.rollout("nidc", 150)
        assert_same_state(g, o)


@pytest.mark.parametrize("n_envs,cars,rays,policy", [(4101, 1, 36, "fast"), (1100, 3, 90, "nidc"), (517, 1, 360, "random")])
def test_ragged_batches(product, oracle, n_envs, cars, rays, policy):
    """Batch sizes that leave the last workgroup partly filled (8 cars per workgroup at 4101 envs, 12 at 1100 x 3, one at 517):
    every env against the oracle after a closed loop."""
    t = load_track("circle")
    g, o = both(product, oracle, t, n_envs=n_envs, cars_per_env=cars, n_rays=rays, spawn_mode=1 if cars == 1 else 0, seed=8)
    with g, o:
        g.rollout(policy, 60); o.rollout(policy, 60)
        assert_same_state(g, o)
        g.reset(); o.reset()
        g.rollout(policy, 7); o.rollout(policy, 7)
        assert_same_state(g, o)


def test_finished_cars_become_ghosts(product, oracle):
    """custom.py:1367-1371,1441-1466: a car that reached lap_target gets the null driver, stops colliding and is invisible."""
    t = load_track("circle")
    g, o = both(product, oracle, t, n_envs=6, cars_per_env=3, n_rays=90, lap_target=0)   # lap_target 0: finished at once
    with g, o:
        assert g.progress()[:, 4].all()
        g.rollout("fast", 50); o.rollout("fast", 50)
        assert_same_state(g, o)
        np.testing.assert_array_equal(g.ctrl(), 0.0)
        np.testing.assert_array_equal(g.lidar(), 0.0)               # their own rangefinders are switched off (custom.py:1436-1439)


def test_workgroup_shape_does_not_change_a_bit(product, oracle, monkeypatch):
    """Which wave / lane marches which ray, how many cars share a workgroup and its ray pool, and how many waves sweep it
    are scheduling only: every shape must return the oracle's bits (single- and multi-car envs)."""
    t = load_track("inkscape")
    for cars, shapes in ((1, ((16, 16), (1, 16), (5, 3), (16, 1))), (3, ((15, 16), (3, 7), (6, 2)))):
        kw = dict(n_envs=40 // cars, cars_per_env=cars, n_rays=1080, spawn_mode=1 if cars == 1 else 0, seed=21, lap_target=2)
        envs = []
        for cpb, wpb in shapes:
            monkeypatch.setenv("FTGP_CARS_PER_BLOCK", str(cpb)); monkeypatch.setenv("FTGP_WAVES_PER_BLOCK", str(wpb))
            envs.append(capi.Env(product, t, **kw))
        monkeypatch.delenv("FTGP_CARS_PER_BLOCK"); monkeypatch.delenv("FTGP_WAVES_PER_BLOCK")
        o = capi.Env(oracle, t, **kw); oracle.dll.oracle_set_threads(o.h, 8)
        for e in envs + [o]:
            e.rollout("fast", 250)
        for e in envs:
            np.testing.assert_array_equal(e.lidar(), envs[0].lidar())
            np.testing.assert_array_equal(e.pose(), envs[0].pose())
            np.testing.assert_array_equal(e.progress(), envs[0].progress())
            assert_same_state(e, o)
        for e in envs + [o]:
            e.close()


def test_config2_and_config5_full_size_properties(product, oracle):
    """BASELINE.json configs[1] (1024 envs, circle, nidc) and configs[4] (4096 envs x 4 cars) at full size:
    launch-split invariance (a checksum of everything), range sanity, and the oracle on a prefix of the envs."""
    for name, kw, policy, steps, prefix in (
            ("circle", dict(n_envs=1024, n_rays=1080, spawn_mode=1, seed=1234), "nidc", 150, 16),
            ("track", dict(n_envs=4096, cars_per_env=4, n_rays=1080, spawn_mode=0, seed=1234, lap_target=3), "fast", 60, 4)):
        t = load_track(name)
        okw = dict(kw, n_envs=prefix)
        with capi.Env(product, t, **kw) as g, capi.Env(product, t, **kw) as g2, capi.Env(oracle, t, **okw) as o:
            oracle.dll.oracle_set_threads(o.h, 8)
            g.rollout(policy, steps); g2.rollout(policy, steps // 3); g2.rollout(policy, steps - steps // 3); o.rollout(policy, steps)
            r = g.lidar()
            np.testing.assert_array_equal(r, g2.lidar())
            np.testing.assert_array_equal(g.pose(), g2.pose())
            np.testing.assert_array_equal(g.progress(), g2.progress())
            assert ((r == -1) | ((r >= 0) & (r < 60))).all()
            n = prefix * kw.get("cars_per_env", 1)
            np.testing.assert_array_equal(r[:n], o.lidar())
            np.testing.assert_array_equal(g.progress()[:n], o.progress())
            np.testing.assert_allclose(g.pose()[:n], o.pose(), rtol=0, atol=TOL)
            m = g.metrics_local()
            assert m[0] == kw["n_envs"] * steps and m[1] == g.n_cars


def test_bench_two_ranks_host_gather():
    """`bench.py --gpus 2` with no launcher: it starts its own two ranks (both on this box's one GPU: a rehearsal of the N > 1
    plumbing -- rendezvous, env shards, barrier, max-over-ranks timing -- with the metrics gathered over the host instead of
    RCCL, which cannot form a two-rank communicator on one device), and rank 0 prints ONE JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FTGP_BENCH_COLLECTIVE"] = "host"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--envs-per-gpu", "256", "--repeats", "3"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["repeats"] == 3 and d["metrics_allgather"]["ranks"] == 2 and d["metrics_allgather"]["sum_steps"] == 2 * 256 * (2 + 3 * 5)
    assert d["value"] == pytest.approx(2 * 256 * 5 / (d["ms_per_step"] * 5e-3), rel=1e-6)
    # what every rank saw, before the max over ranks: the numbers that explain a scaling curve that bends
    pr = d["per_rank"]
    assert len(pr["kernel_ms_median"]) == len(pr["kernel_ms_min"]) == len(pr["kernel_ms_max"]) == len(pr["wall_ms_median"]) == 2
    assert all(lo <= md <= hi for lo, md, hi in zip(pr["kernel_ms_min"], pr["kernel_ms_median"], pr["kernel_ms_max"])) and pr["kernel_ms_spread_over_ranks"] >= 1.0
    assert d["metrics_allgather"]["end_wait_ms_max"] >= d["metrics_allgather"]["end_wait_ms_median"] >= 0.0


def test_launch_metrics_record_equals_the_metrics_kernel(product, oracle):
    """The record the step kernel's last workgroup leaves behind (read without a second kernel) against ftgp_metrics_kernel on the
    same state (a handle created with the diagnostic switch FTGP_NO_FUSED_METRICS) and against the oracle, through lap events,
    ragged workgroups and calls that invalidate the cached record."""
    import os
    t = load_track("circle")
    kw = dict(n_envs=37, n_rays=90, lap_target=1, spawn_mode=1, seed=7)
    g = capi.Env(product, t, **kw)                       # one rank, no communicator: partial records to pinned memory, added up by the host
    os.environ["FTGP_NO_FUSED_METRICS"] = "1"
    try:
        k = capi.Env(product, t, **kw)
    finally:
        del os.environ["FTGP_NO_FUSED_METRICS"]
    os.environ["FTGP_NO_HOST_SUM"] = "1"                 # the device-side hand-off (what a rank with a communicator runs)
    try:
        d = capi.Env(product, t, **kw)
    finally:
        del os.environ["FTGP_NO_HOST_SUM"]
    o = capi.Env(oracle, t, **kw)
    oracle.dll.oracle_set_threads(o.h, 8)
    with g, k, d, o:
        for n in (1, 250, 9000, 10000):
            for e in (g, k, d, o):
                e.rollout("nidc", n)
            np.testing.assert_array_equal(g.metrics_local(), k.metrics_local())
            np.testing.assert_array_equal(d.metrics_local(), k.metrics_local())
            np.testing.assert_array_equal(g.metrics_local(), o.metrics_local())
            np.testing.assert_array_equal(g.metrics_allgather()[0], k.metrics_local())
        assert g.metrics_local()[4] > 0 and np.isfinite(g.metrics_local()[6])          # some cars finished: lap times in the record
        mask = np.zeros(37, dtype=np.uint8); mask[::2] = 1
        for e in (g, k, o):
            e.reset(mask)                                                                # the cached record no longer describes the state
        np.testing.assert_array_equal(g.metrics_local(), k.metrics_local())
        np.testing.assert_array_equal(g.metrics_local(), o.metrics_local())
        for e in (g, k, o):
            e.step(3)
        np.testing.assert_array_equal(g.metrics_local(), o.metrics_local())


def test_config4_all_eight_shards_equal_the_monolithic_batch(product):
    """BASELINE.json configs[3] at its full size on one GPU: the 32768-env random-policy batch run as ONE handle, and as the eight
    4096-env shards the 8-GPU job consists of (env_base = rank * 4096, one after the other on this GPU).  Every shard must
    reproduce its slice of the monolithic batch bit for bit -- spawn poses, random controls, scans, progress -- and the metrics
    records of the shards must add up to the monolithic one: what the 8-GPU job computes is then what one big GPU would."""
    from ft_grandprix_amd import dist as ftdist
    t = load_track("track")
    kw = dict(n_rays=1080, spawn_mode=1, seed=1234)
    steps = 40
    with capi.Env(product, t, n_envs=32768, **kw) as mono:
        mono.rollout("random", steps)
        lid, prog, ctrl, pose, rec = mono.lidar(), mono.progress(), mono.ctrl(), mono.pose(), mono.metrics_local()
    recs = []
    for rank in range(8):
        with ftdist.make_shard(product, t, 32768, rank, 8, **kw) as sh:
            assert (sh.n_envs, sh.env_base) == (4096, rank * 4096)
            sh.rollout("random", steps)
            sl = slice(rank * 4096, (rank + 1) * 4096)
            np.testing.assert_array_equal(sh.lidar(), lid[sl])
            np.testing.assert_array_equal(sh.progress(), prog[sl])
            np.testing.assert_array_equal(sh.ctrl(), ctrl[sl])
            np.testing.assert_array_equal(sh.pose(), pose[sl])
            recs.append(sh.metrics_local())
    tot, ref = ftdist.reduce_metrics(np.stack(recs)), ftdist.reduce_metrics(rec[None])
    for k in capi.METRIC_FIELDS:
        assert tot[k] == ref[k], k
    assert tot["steps"] == 32768 * steps and tot["ranks"] == 8


@pytest.mark.parametrize("case", range(8))
def test_randomised_worlds(product, oracle, case):
    """Seeded differential test over the whole configuration space at once: procedural tracks of odd sizes (square and stretched
    pixels, thin and thick corridors), ray counts, cars per env, batch sizes that leave workgroups ragged, drivers, friction,
    bubble_wrap -- every combination closed-loop against the oracle, state compared after uneven launches."""
    from ft_grandprix_amd.track import synthetic_oval
    rng = np.random.default_rng(1000 + case)
    w = int(rng.integers(300, 900)); h = int(rng.integers(260, 700))
    t = synthetic_oval(width=w, height=h, half_width_px=float(rng.uniform(14, 30)), wall_px=float(rng.uniform(0.8, 2.5)),
                       name=f"rand{case}", frame=("mjcf", "pixel")[case % 2] if case % 3 else "mjcf")
    cars = int(rng.choice([1, 1, 2, 3, 5]))
    rays = int(rng.choice([8, 24, 90, 333, 720, 1080, 1500]))
    envs = int(rng.integers(3, 70))
    policy = str(rng.choice(["nidc", "fast", "nidc", "fast", "random", "lobotomy"]))
    v = product.default_vehicle()
    v.friction = float(rng.uniform(0.3, 1.5))
    kw = dict(n_envs=envs, cars_per_env=cars, n_rays=rays, spawn_mode=int(rng.integers(0, 2)), seed=int(rng.integers(1, 10 ** 6)),
              lap_target=0 if case == 0 else int(rng.integers(1, 4)), bubble_wrap=bool(rng.integers(0, 2)), vehicle=v)
    g, o = capi.Env(product, t, **kw), capi.Env(oracle, t, **kw)
    oracle.dll.oracle_set_threads(o.h, 8)
    with g, o:
        for n in (1, int(rng.integers(2, 40)), int(rng.integers(40, 260))):
            g.rollout(policy, n); o.rollout(policy, n)
            assert_same_state(g, o)
        np.testing.assert_array_equal(g.winners(), o.winners())
        np.testing.assert_array_equal(g.metrics_local(), o.metrics_local())
