"""N > 1 layout on CPU: shards reproduce the monolithic batch; the metrics all-gather over gloo (world_size 2)."""
import os
import socket
import sys
import time

import numpy as np
import pytest

from ft_grandprix_amd import capi, dist as ftdist
from ft_grandprix_amd.track import load_track

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_partition():
    for total, world in ((4096, 8), (32768, 8), (10, 3), (7, 8)):
        spans = [ftdist.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
    with pytest.raises(ValueError):
        ftdist.shard_range(8, 2, 2)


@pytest.mark.parametrize("policy", ["fast", "random"])
def test_shards_reproduce_the_monolithic_batch(oracle, policy):
    t = load_track("circle")
    kw = dict(n_rays=90, spawn_mode=1, seed=77, lap_target=2)
    with capi.Env(oracle, t, n_envs=24, **kw) as mono:
        mono.rollout(policy, 120)
        recs = []
        for rank in range(3):
            with ftdist.make_shard(oracle, t, 24, rank, 3, **kw) as sh:
                start, count = ftdist.shard_range(24, rank, 3)
                assert sh.env_base == start and sh.n_envs == count
                sh.rollout(policy, 120)
                np.testing.assert_array_equal(sh.lidar(), mono.lidar()[start:start + count])
                np.testing.assert_array_equal(sh.pose(), mono.pose()[start:start + count])
                np.testing.assert_array_equal(sh.progress(), mono.progress()[start:start + count])
                recs.append(sh.metrics_local())
        tot, ref = ftdist.reduce_metrics(np.stack(recs)), ftdist.reduce_metrics(mono.metrics_local()[None])
        for k in capi.METRIC_FIELDS:
            assert tot[k] == ref[k], k


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from tests.helpers import load_oracle
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        ora = load_oracle()
        t = load_track("circle")
        with ftdist.make_shard(ora, t, 12, rank, world, n_rays=36, spawn_mode=1, seed=5, lap_target=1) as sh:
            sh.rollout("fast", 150)
            recs = ftdist.gather_metrics(sh, ftdist.GlooGather())       # the N > 1 exchange, over gloo
            dist.barrier()
            q.put((rank, recs, sh.metrics_local()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_metrics_allgather(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank holds every rank's record, in rank order
    np.testing.assert_array_equal(out[0][1], out[1][1])
    for r in range(2):
        np.testing.assert_array_equal(out[0][1][r], out[r][2])
    t = load_track("circle")
    with capi.Env(oracle, t, n_envs=12, n_rays=36, spawn_mode=1, seed=5, lap_target=1) as mono:
        mono.rollout("fast", 150)
        ref = ftdist.reduce_metrics(mono.metrics_local()[None])
    tot = ftdist.reduce_metrics(out[0][1])
    assert tot["ranks"] == 2
    for k in capi.METRIC_FIELDS:
        assert tot[k] == ref[k], k


def _rdzv_main(rank, world, port, q):
    """One rank of a torch-free job: shard + metrics gather over ft_grandprix_amd.dist.Rendezvous, and the id exchange
    with a failing rank 0 (every rank must raise together, none may stay blocked)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), FTGP_JOB_TOKEN="t-rdzv")
    sys.path.insert(0, ROOT)
    from tests.helpers import load_oracle
    rdzv = ftdist.Rendezvous.from_env(timeout=60)
    try:
        uid = ftdist.exchange_unique_id(rdzv, lambda: bytes(range(128)))
        try:
            ftdist.exchange_unique_id(rdzv, lambda: (_ for _ in ()).throw(RuntimeError("librccl.so not loadable")))
            err = ""
        except RuntimeError as exc:
            err = str(exc)
        ora = load_oracle()
        t = load_track("circle")
        with ftdist.make_shard(ora, t, 12, rank, world, n_rays=36, spawn_mode=1, seed=5, lap_target=1) as sh:
            sh.rollout("fast", 150)
            recs = ftdist.gather_metrics(sh, rdzv)
            rdzv.barrier()
            tmax = rdzv.max([float(rank), 10.0 - rank])
            q.put((rank, recs, sh.metrics_local(), uid, err, tmax))
    finally:
        rdzv.close()


def test_world_size_3_tcp_rendezvous_without_torch(oracle):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_rdzv_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        np.testing.assert_array_equal(out[0][1], out[r][1])          # every rank holds every rank's record, in rank order
        np.testing.assert_array_equal(out[0][1][r], out[r][2])
        assert out[r][3] == bytes(range(128))                       # the 128-byte id arrived on every rank
        assert "librccl.so not loadable" in out[r][4]                # ... and so did rank 0's failure
        np.testing.assert_array_equal(out[r][5], [world - 1.0, 10.0])
    t = load_track("circle")
    with capi.Env(oracle, t, n_envs=12, n_rays=36, spawn_mode=1, seed=5, lap_target=1) as mono:
        mono.rollout("fast", 150)
        ref = ftdist.reduce_metrics(mono.metrics_local()[None])
    tot = ftdist.reduce_metrics(out[0][1])
    assert tot["ranks"] == world
    for k in capi.METRIC_FIELDS:
        assert tot[k] == ref[k], k


def test_rendezvous_survives_stray_and_short_clients():
    """a client that connects to the rendezvous port and sends garbage, too little, or nothing must be dropped,
    not take rank 0 down; the real rank still gets in afterwards."""
    import socket
    import threading
    port = _free_port()
    box = {}

    def rank0():
        try:
            box["r0"] = ftdist.Rendezvous(0, 2, "127.0.0.1", port, token="t-stray", timeout=30)
        except Exception as exc:             # noqa: BLE001 - asserted below
            box["err"] = exc
    th = threading.Thread(target=rank0); th.start()
    strays = []
    deadline = time.time() + 10
    while time.time() < deadline and len(strays) < 3:
        try:
            c = socket.create_connection(("127.0.0.1", port + 1), timeout=1.0)
        except OSError:
            time.sleep(0.05); continue
        strays.append(c)
        if len(strays) == 1:
            c.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"x" * 80)      # a full-length but foreign hello
        elif len(strays) == 2:
            c.sendall(b"abc"); c.close()                            # short, then gone
        # the third one says nothing at all and stays open: only its own 5-s handshake timeout is spent on it
    assert len(strays) == 3
    r1 = ftdist.Rendezvous(1, 2, "127.0.0.1", port, token="t-stray", timeout=30)
    th.join(timeout=30)
    assert "err" not in box and not th.is_alive(), box.get("err")
    r0 = box["r0"]
    got = {}
    t2 = threading.Thread(target=lambda: got.setdefault("a", r0.allgather_bytes(b"zero")))
    t2.start()
    assert r1.allgather_bytes(b"one") == [b"zero", b"one"]
    t2.join(timeout=10)
    assert got["a"] == [b"zero", b"one"]
    for c in strays:
        c.close()
    r0.close(); r1.close()


# ---------------------------------------------------------------- bench.py --gpus N: the self-launcher
def test_bench_launcher_stops_the_job_when_one_rank_fails(tmp_path):
    """One rank dies, the other would wait for it (rendezvous, RCCL bootstrap): the launcher must stop the survivor and exit
    non-zero with the failing rank's code -- promptly, not after the survivor's timeout."""
    import bench
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "r = int(os.environ['RANK'])\n"
                     "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['LOCAL_RANK'] == str(r)\n"
                     "open(sys.argv[1] + f'/started{r}', 'w').close()\n"
                     "if r == 1:\n    sys.exit(5)\n"
                     "time.sleep(120)\nopen(sys.argv[1] + '/survived', 'w').close()\n")
    t0 = time.time()
    rc = bench.launch_ranks(2, [str(tmp_path)], script=str(child))
    assert rc == 5 and time.time() - t0 < 30
    assert (tmp_path / "started0").exists() and (tmp_path / "started1").exists() and not (tmp_path / "survived").exists()
    ok = tmp_path / "ok.py"
    ok.write_text("import sys\nsys.exit(0)\n")
    assert bench.launch_ranks(3, [], script=str(ok)) == 0


def test_bench_gpus_2_without_a_gpu_fails_loudly():
    """`python bench.py --gpus 2` with no launcher in the environment starts its own two ranks; with no HIP device each of them
    refuses to run (no CPU fallback) and the parent's exit code says so."""
    import subprocess
    from ft_grandprix_amd import capi
    if capi.load().fn("device_count")() >= 1:
        pytest.skip("a GPU is present: the happy path is tests/test_gpu_parity.py::test_bench_two_ranks_host_gather")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FTGP_BENCH_COLLECTIVE"] = "host"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "16"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert p.stderr.count("needs a HIP device") >= 1 and p.stdout.strip() == ""


# ---------------------------------------------------------------- the exchange overlaps the next launch
class _SlowComm:
    """A host communicator whose gather takes `delay` seconds more (the stand-in for a slow collective)."""

    def __init__(self, comm, delay):
        self.comm, self.delay, self.calls = comm, delay, 0

    def all_gather(self, rec):
        self.calls += 1
        time.sleep(self.delay)
        return self.comm.all_gather(rec)


def _overlap_main(rank, world, port, q):
    """One rank: the oracle stands in for the GPU (a `launch` is a blocking CPU rollout), the metrics travel over a Rendezvous of
    their own inside a HostExchange worker thread.  Timed twice: plain gather, and a gather slowed by most of a launch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), FTGP_JOB_TOKEN="t-overlap")
    sys.path.insert(0, ROOT)
    from tests.helpers import load_oracle
    rdzv = ftdist.Rendezvous.from_env(timeout=60)
    rdzv_x = ftdist.Rendezvous.from_env(timeout=60, channel="x")
    try:
        ora = load_oracle()
        t = load_track("circle")
        with ftdist.make_shard(ora, t, 8, rank, world, n_rays=90, spawn_mode=1, seed=5) as sh:
            t0 = time.perf_counter(); sh.rollout("nidc", 40); per_step = (time.perf_counter() - t0) / 40
            steps = max(20, int(0.25 / per_step))                       # a launch of about a quarter of a second
            steps = int(rdzv.max([steps])[0])
            out = {}
            for label, delay in (("plain", 0.0), ("slow", 0.2)):
                comm = _SlowComm(rdzv_x, delay)
                ex = ftdist.HostExchange(sh, comm)
                sh.rollout("nidc", steps); ex.begin()                   # the warm-up launch and its exchange
                timed = ftdist.run_timed(sh, "nidc", steps, 4, ex, rdzv.barrier, kernel_ms=lambda: None)
                ex.close()
                walls = rdzv.max(timed["wall_s"])
                out[label] = (walls.tolist(), timed["records"], sh.metrics_local(), comm.calls)
            q.put((rank, steps, out))
    finally:
        rdzv_x.close(); rdzv.close()


def test_metrics_exchange_overlaps_the_next_launch_two_ranks(oracle):
    """Two ranks, host gather standing in for the collective: a gather slowed by 0.2 s -- most of a launch -- must not lengthen
    the launches it runs beside (serial, every launch would take 0.2 s longer), and the records collected at the end are the
    last launch's, from every rank, in rank order."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    out = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for label in ("plain", "slow"):
        recs0, recs1 = out[0][2][label][1], out[1][2][label][1]
        np.testing.assert_array_equal(recs0, recs1)                                   # every rank holds every rank's record
        np.testing.assert_array_equal(recs0[0], out[0][2][label][2])                  # ... of the LAST launch (nothing ran since)
        np.testing.assert_array_equal(recs0[1], out[1][2][label][2])
        assert out[0][2][label][3] == 5                                               # warm-up + 4 timed launches: one exchange each
    plain, slow = np.median(out[0][2]["plain"][0]), np.median(out[0][2]["slow"][0])
    assert slow < plain + 0.1, (plain, slow)            # serial would be plain + 0.2


def test_run_timed_device_exchange_order(oracle):
    """The call order ftdist.run_timed imposes on a DeviceExchange-like object: launch k is enqueued, THEN exchange k - 1 is
    collected, THEN exchange k is begun, THEN the launch is synchronised -- never two exchanges open, none left open."""
    log = []

    class Env:
        def rollout(self, policy, steps): log.append("launch")
        def last_kernel_ms(self): log.append("sync"); return 1.0

    class Ex:
        after_sync, open = False, False
        def begin(self): assert not self.open; self.open = True; log.append("begin")
        def end(self): assert self.open; self.open = False; log.append("end"); return "rec"

    ex = Ex(); ex.begin()                                # the warm-up's exchange
    r = ftdist.run_timed(Env(), "fast", 20, 3, ex, barrier=lambda: log.append("barrier"))
    assert log == ["begin"] + ["barrier", "launch", "end", "begin", "sync"] * 3 + ["end"]
    assert r["records"] == "rec" and len(r["wall_s"]) == 3 and r["kernel_ms"] == [1.0] * 3
