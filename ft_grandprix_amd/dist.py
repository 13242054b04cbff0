"""Multi-GPU layout of the hot path: envs are independent worlds, so a batch is cut into
contiguous shards, one handle (one process, one GPU) per shard, with no data-path collective.  The only exchange
is the end-of-step metrics all-gather: RCCL over xGMI from the C shim (``Env.metrics_allgather``) on GPUs, or any
object exposing ``all_gather(np.ndarray) -> np.ndarray`` (``Rendezvous`` below; the CPU tests also use gloo).

``Rendezvous`` is the host-side plumbing one process per GPU needs around that -- shipping the 128-byte RCCL id from
rank 0, a barrier, the max-over-ranks of a timing -- over plain TCP on the launcher's MASTER_ADDR, so the host path
needs neither PyTorch nor MPI."""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import capi


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous env range [start, start + count) owned by ``rank``: GPU g owns envs [g*N/G, (g+1)*N/G)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    start = rank * total_envs // world
    end = (rank + 1) * total_envs // world
    return start, end - start


def make_shard(lib: capi.CLib, track, total_envs: int, rank: int, world: int, **env_kwargs) -> capi.Env:
    """The handle for this rank's slice of a ``total_envs`` batch (same seed on every rank: identity comes from env_base)."""
    start, count = shard_range(total_envs, rank, world)
    return capi.Env(lib, track, n_envs=count, env_base=start, **env_kwargs)


def reduce_metrics(records: np.ndarray) -> Dict[str, float]:
    """[world][FTGP_METRIC_DOUBLES] per-rank records -> whole-job totals."""
    r = np.asarray(records, dtype=np.float64).reshape(-1, capi.METRIC_DOUBLES)
    out = {k: float(r[:, i].sum()) for i, k in enumerate(capi.METRIC_FIELDS[:6])}
    out["min_lap_time"] = float(r[:, 6].min())
    out["max_lap_time"] = float(r[:, 7].max())
    out["ranks"] = int(r.shape[0])
    return out


class GlooGather:
    """all_gather of small float64 records over an initialised torch.distributed (gloo) group -- CPU tests only."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist

    def all_gather(self, rec: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float64))
        outs = [self.torch.empty_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])


def gather_metrics(env: capi.Env, comm: Optional[object] = None) -> np.ndarray:
    """Per-rank metrics records of the whole job.  comm=None: the handle's own RCCL communicator (or world 1)."""
    if comm is None:
        return env.metrics_allgather()
    return comm.all_gather(env.metrics_local())


class DeviceExchange:
    """The end-of-launch metrics all-gather on the handle's own communicator (RCCL over xGMI; or a single rank), in two halves:
    ``begin()`` right after a launch has been enqueued puts the exchange on the side stream behind that launch and returns at
    once, ``end()`` waits for that exchange alone -- so the exchange of launch k runs beside launch k + 1."""
    after_sync = False               # begin() wants the launch enqueued, not finished

    def __init__(self, env: capi.Env):
        self.env, self.open = env, False

    def begin(self):
        self.env.metrics_allgather_begin()
        self.open = True

    def end(self) -> np.ndarray:
        self.open = False
        return self.env.metrics_allgather_end()

    def close(self):
        pass


class HostExchange:
    """The same exchange over a host communicator (``Rendezvous`` or ``GlooGather``: anything with ``all_gather``): ``begin()``
    -- after the launch has finished -- takes this rank's record (already in pinned memory) and hands it to a worker thread,
    which gathers while the caller enqueues and runs the next launch; ``end()`` collects.  The worker never touches the
    handle.  ``comm`` must not be used by the caller while an exchange is open (give the exchange a communicator of its own)."""
    after_sync = True                # begin() needs the launch finished

    def __init__(self, env: capi.Env, comm):
        from concurrent.futures import ThreadPoolExecutor
        self.env, self.comm, self.open = env, comm, False
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.fut = None

    def begin(self):
        rec = self.env.metrics_local()
        self.fut = self.pool.submit(self.comm.all_gather, rec)
        self.open = True

    def end(self) -> np.ndarray:
        self.open = False
        return self.fut.result()

    def close(self):
        self.pool.shutdown(wait=True)


def run_timed(env: capi.Env, policy, steps: int, repeats: int, exchange, barrier=None, kernel_ms=None) -> Dict[str, object]:
    """``repeats`` launches of exactly ``steps`` steps, each timed on its own (barrier, then wall clock from before the launch to
    after its synchronisation).  The metrics exchange of launch k is begun inside launch k's timed region and collected inside
    launch k + 1's, after that launch has been enqueued: it overlaps the next launch instead of sitting serially between two
    (an exchange begun before the call -- the warm-up's -- is collected during the first launch).  The last launch's exchange is
    collected after the loop.  ``kernel_ms()`` (default ``env.last_kernel_ms``) is what synchronises with a launch.
    Returns {"wall_s": [...], "kernel_ms": [...], "exchange_end_s": [how long each collected exchange's end() took], "records": the
    newest collected records}."""
    sync = kernel_ms if kernel_ms is not None else env.last_kernel_ms
    walls, kms, ends, records = [], [], [], None
    for _ in range(repeats):
        if barrier is not None:
            barrier()
        t0 = time.perf_counter()
        env.rollout(policy, steps)                       # enqueued: returns at once on the GPU
        if exchange.open:
            te = time.perf_counter()
            records = exchange.end()                     # the previous launch's exchange (it ran beside this enqueue / launch)
            ends.append(time.perf_counter() - te)        # what end() made this launch's host side wait (an exchange that overlapped: next to nothing)
        if not exchange.after_sync:
            exchange.begin()                             # this launch's: side stream, behind the launch
        k = sync()                                       # HIP events on the launch stream; also synchronises
        if exchange.after_sync:
            exchange.begin()                             # this launch's record to the worker thread
        walls.append(time.perf_counter() - t0)
        kms.append(float(k) if k is not None else float("nan"))
    if exchange.open:
        records = exchange.end()
    return {"wall_s": walls, "kernel_ms": kms, "exchange_end_s": ends, "records": records}


class Rendezvous:
    """All ranks of one job on one node: rank 0 listens, the others connect; every operation is a gather at rank 0
    followed by a broadcast (a few hundred bytes -- latency is irrelevant, ordering is what matters).

    The launcher's MASTER_PORT itself belongs to the launcher's own store, so the exchange uses MASTER_PORT + 1 + k for
    the first k in 0..15 that rank 0 can bind; clients find it by a handshake carrying the job token."""

    MAGIC = b"FTGPRDZV"

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29500, token: str = "", timeout: float = 120.0):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        self.rank, self.world, self.timeout = rank, world, timeout
        self.peers: List[Optional[socket.socket]] = [None] * world
        self.sock: Optional[socket.socket] = None
        hello = self.MAGIC + token.encode()[:56].ljust(56, b"\0")
        if world == 1:
            return
        if rank == 0:
            srv, err = None, None
            for k in range(16):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port + 1 + k)); srv.listen(world)
                    break
                except OSError as e:
                    err, srv = e, None
            if srv is None:
                raise OSError(f"rendezvous: no free port near {port}: {err}")
            deadline = time.time() + timeout
            got = 0
            try:
                while got < world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError(f"rendezvous: {world - 1 - got} of {world - 1} ranks never connected")
                    srv.settimeout(left)
                    try:
                        c, _ = srv.accept()
                    except socket.timeout:
                        continue
                    # a stray, short or silent client (a port scanner, another job) must not take rank 0 down: its handshake
                    # gets a short timeout of its own and any failure just drops that one connection
                    try:
                        c.settimeout(min(5.0, timeout)); c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        h = self._recv_exact(c, len(hello) + 4)
                        r = struct.unpack("<i", h[len(hello):])[0]
                        if h[:len(hello)] != hello or not (0 < r < world) or self.peers[r] is not None:
                            c.close(); continue                 # somebody else's client
                        c.sendall(b"OK"); c.settimeout(timeout)
                    except (OSError, ConnectionError):
                        c.close(); continue
                    self.peers[r] = c; got += 1
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while self.sock is None:
                for k in range(16):
                    try:
                        c = socket.create_connection((addr, port + 1 + k), timeout=2.0)
                        c.settimeout(timeout); c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        c.sendall(hello + struct.pack("<i", rank))
                        if self._recv_exact(c, 2) == b"OK":
                            self.sock = c
                            break
                        c.close()
                    except OSError:
                        continue
                if self.sock is None:
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank 0 did not answer")
                    time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout: float = 120.0, channel: str = "") -> "Rendezvous":
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run (or bench.py's own launcher) export them.
        ``channel`` names a second, independent connection set of the same job (e.g. for a ``HostExchange`` worker thread)."""
        token = os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("FTGP_JOB_TOKEN", ""))
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                   os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")),
                   (channel + ":" + token) if channel else token, timeout)

    @staticmethod
    def _recv_exact(c: socket.socket, n: int) -> bytes:
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous: peer closed the connection")
            buf += part
        return buf

    def _send(self, c: socket.socket, data: bytes):
        c.sendall(struct.pack("<I", len(data)) + data)

    def _recv(self, c: socket.socket) -> bytes:
        return self._recv_exact(c, struct.unpack("<I", self._recv_exact(c, 4))[0])

    def allgather_bytes(self, data: bytes) -> List[bytes]:
        """Every rank's payload, in rank order, on every rank."""
        if self.world == 1:
            return [data]
        if self.rank == 0:
            parts = [data] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
            for r in range(1, self.world):
                self._send(self.peers[r], blob)
            return parts
        self._send(self.sock, data)
        blob, parts, o = self._recv(self.sock), [], 0
        for _ in range(self.world):
            n = struct.unpack("<I", blob[o:o + 4])[0]
            parts.append(blob[o + 4:o + 4 + n]); o += 4 + n
        return parts

    def broadcast_bytes(self, data: Optional[bytes]) -> bytes:
        """Rank 0's payload on every rank (the other ranks pass None)."""
        return self.allgather_bytes(data if self.rank == 0 else b"")[0]

    def barrier(self):
        self.allgather_bytes(b"")

    def all_gather(self, rec: np.ndarray) -> np.ndarray:
        """[world, ...] stack of every rank's float64 record (same interface as GlooGather)."""
        r = np.ascontiguousarray(rec, dtype=np.float64)
        return np.stack([np.frombuffer(p, dtype=np.float64).reshape(r.shape) for p in self.allgather_bytes(r.tobytes())])

    def max(self, values) -> np.ndarray:
        """Element-wise maximum over ranks."""
        return self.all_gather(np.asarray(values, dtype=np.float64)).max(axis=0)

    def close(self):
        for c in self.peers + [self.sock]:
            if c is not None:
                try:
                    c.close()
                except OSError:
                    pass
        self.peers, self.sock = [None] * self.world, None


def exchange_unique_id(rdzv: Rendezvous, make_id) -> bytes:
    """Rank 0 creates the RCCL unique id with ``make_id()`` and ships it; a failure on rank 0 travels as a marker, so
    that every rank raises together instead of leaving the others blocked in the exchange."""
    payload = None
    if rdzv.rank == 0:
        try:
            payload = b"\x01" + bytes(make_id())
        except Exception as exc:            # noqa: BLE001 - forwarded to every rank
            payload = b"\x00" + str(exc).encode()[:400]
    got = rdzv.broadcast_bytes(payload)
    if got[:1] != b"\x01":
        raise RuntimeError("rank 0 could not create the RCCL id: " + got[1:].decode(errors="replace"))
    return got[1:]
