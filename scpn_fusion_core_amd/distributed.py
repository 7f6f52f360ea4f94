"""Ensemble sharding across the GPUs of one node: one process per GPU, no data-path collective, no PyTorch.

The equilibria of an ensemble (coil / shape scan, shot replay) are independent units, so the path shards across members:
rank r owns a contiguous block of ``ceil(total / world)`` members, solves it with its resident
:class:`~scpn_fusion_core_amd.ensemble.EnsembleSolver`, and the only exchange step is the final all-gather of the dense psi
blocks -- RCCL over xGMI, bound through the C ABI (``scpn_comm_*`` in include/scpn_hip.h, librccl dlopen'ed by the library).

Rendezvous (before any rank touches its GPU): rank 0 creates the 128-byte RCCL unique id and serves it on a TCP socket at
``MASTER_ADDR:port``; the other ranks connect and read it.  ``port`` is ``MASTER_PORT``, or ``MASTER_PORT + 1`` under
``torchrun`` / ``python -m torch.distributed.run`` (which is only the process launcher here: its agent keeps its own store on
``MASTER_PORT``); ``SCPN_RDZV_PORT`` overrides.  The small per-member result table (converged, iterations, residual, axis ...)
travels over the same kind of socket (:func:`gather_rows`): host data, a few hundred bytes per rank.
"""

from __future__ import annotations

import ctypes as C
import os
import socket
import struct
import sys
import time

import numpy as np

ID_BYTES = 128


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [start, stop) of ``total`` members owned by ``rank`` (blocks of ceil(total/world))."""
    if total < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("need total >= 0, world >= 1 and 0 <= rank < world")
    per = -(-total // world)
    start = min(rank * per, total)
    return start, min(start + per, total)


def shard_sizes(total: int, world: int) -> list[int]:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def plan_ensemble(rank: int, world: int, *, members_per_rank: int = 0, total_members: int = 0) -> dict:
    """How a job of ``world`` ranks splits an ensemble, for this ``rank``.

    ``total_members > 0``: STRONG scaling -- a fixed global ensemble cut into contiguous blocks of ceil(total / world) (the last
    rank may hold fewer, never none).  Otherwise WEAK scaling: ``members_per_rank`` on every rank.  Returns ``{"scaling", "total",
    "start", "stop", "members" (this rank's), "block" (what every rank sends into the gather: short blocks are padded)}``."""
    strong = total_members > 0
    total = int(total_members) if strong else int(members_per_rank) * world
    if total < 1:
        raise ValueError("the ensemble needs at least one member")
    if min(shard_sizes(total, world)) < 1:
        raise ValueError(f"{total} members cannot be cut into {world} non-empty blocks of {-(-total // world)}")
    start, stop = shard_range(total, rank, world)
    return {"scaling": "strong" if strong else "weak", "total": total, "start": start, "stop": stop, "members": stop - start,
            "block": -(-total // world)}


# ------------------------------------------------------------------------------------------------- TCP plumbing (host data)
def rendezvous_port(env=None) -> int:
    env = os.environ if env is None else env
    if env.get("SCPN_RDZV_PORT"):
        return int(env["SCPN_RDZV_PORT"])
    port = int(env.get("MASTER_PORT", "29400"))
    return port + 1 if env.get("TORCHELASTIC_RUN_ID") else port


def _recv_exact(conn: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket early")
        buf += chunk
    return bytes(buf)


def _connect(addr: str, port: int, timeout: float) -> socket.socket:
    deadline = time.monotonic() + timeout
    while True:
        try:
            return socket.create_connection((addr, port), timeout=5.0)
        except OSError:
            if time.monotonic() > deadline:
                raise TimeoutError(f"no rendezvous server at {addr}:{port} after {timeout:.0f} s")
            time.sleep(0.05)


def exchange(rank: int, world: int, payload: bytes, *, addr: str, port: int, timeout: float = 120.0) -> list[bytes]:
    """All ranks contribute ``payload`` (any length) and all receive the list of the ``world`` payloads in rank order.
    Rank 0 is the server: it accepts world-1 connections, reads (rank, length, bytes) from each, then sends everybody the
    whole table.  One round trip, used for the unique id (only rank 0's entry is non-empty) and for result tables."""
    if world == 1:
        return [payload]
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        table: list[bytes | None] = [None] * world
        table[0] = payload
        conns = []
        try:
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                conn.settimeout(timeout)
                r, n = struct.unpack("!ii", _recv_exact(conn, 8))
                if not (0 < r < world) or table[r] is not None:
                    raise ConnectionError(f"unexpected rank {r} at the rendezvous")
                table[r] = _recv_exact(conn, n)
                conns.append(conn)
            blob = b"".join(struct.pack("!i", len(p)) + p for p in table)  # type: ignore[arg-type]
            for conn in conns:
                conn.sendall(blob)
        finally:
            for conn in conns:
                conn.close()
            srv.close()
        return list(table)  # type: ignore[arg-type]
    deadline = time.monotonic() + timeout
    while True:  # a connection that lands in the backlog of the PREVIOUS round's closing server is reset: try again
        conn = _connect(addr, port, timeout)
        try:
            conn.settimeout(timeout)
            conn.sendall(struct.pack("!ii", rank, len(payload)) + payload)
            out = []
            for _ in range(world):
                (n,) = struct.unpack("!i", _recv_exact(conn, 4))
                out.append(_recv_exact(conn, n))
            return out
        except (ConnectionError, socket.timeout):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)
        finally:
            conn.close()


def gather_rows(rank: int, world: int, rows: np.ndarray, total: int, *, addr: str, port: int) -> np.ndarray:
    """Assemble the (total, k) float64 table of per-member result rows from the ranks' (n_local, k) blocks (on every rank)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    lo, hi = shard_range(total, rank, world)
    if rows.ndim != 2 or rows.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {rows.shape[0] if rows.ndim == 2 else '?'} rows, expected {hi - lo}")
    parts = exchange(rank, world, rows.tobytes(), addr=addr, port=port)
    k = rows.shape[1]
    blocks = [np.frombuffer(p, dtype=np.float64).reshape(-1, k) for p in parts]
    for r, blk in enumerate(blocks):
        if blk.shape[0] != shard_sizes(total, world)[r]:
            raise ValueError(f"rank {r} sent {blk.shape[0]} rows")
    return np.concatenate(blocks, axis=0)


def gather_rank_stats(rank: int, world: int, stats: dict, *, addr: str, port: int) -> list[dict]:
    """One small JSON record per rank, assembled on every rank in rank order (the ``per_rank`` table of the bench line: what makes
    a sub-linear scaling curve attributable to imbalance, to the gather or to the launch path).  ``world == 1``: no socket is opened."""
    import json

    rec = json.dumps({"rank": int(rank), **stats}).encode()
    parts = [rec] if world == 1 else exchange(rank, world, rec, addr=addr, port=port)
    out = [json.loads(p.decode()) for p in parts]
    if [r.get("rank") for r in out] != list(range(world)):
        raise ValueError(f"per-rank records out of order: {[r.get('rank') for r in out]}")
    return out


def free_port(addr: str = "127.0.0.1") -> int:
    """A TCP port that is free now AND whose successor is free (the rendezvous may use ``port + 1``)."""
    for _ in range(64):
        with socket.socket() as a:
            a.bind((addr, 0))
            port = a.getsockname()[1]
            try:
                with socket.socket() as b:
                    b.bind((addr, port + 1))
            except OSError:
                continue
        return port
    raise OSError("no pair of free rendezvous ports")


def spawn_ranks(argv: list[str], n: int, *, env=None, addr: str = "127.0.0.1", timeout: float | None = None) -> tuple[int, str]:
    """Start ``n`` fresh child processes of ``argv`` -- one rank per GPU -- and wait for them.

    The launcher a ``--gpus N`` command line uses when nobody has set ``WORLD_SIZE``: each child gets
    ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT``; rank 0's stdout is captured and returned (the one JSON line),
    every other stream is passed through to this process's stderr.  The parent must not have touched a GPU: it only forks
    children (no exec of a process that has initialised HIP).  Returns ``(exit code, rank-0 stdout)``; the exit code is the
    first non-zero child code (a rank killed by a signal counts as 128 + signal), and when one rank fails the others are
    terminated instead of being left to wait at a rendezvous that will never complete."""
    import subprocess

    if n < 1:
        raise ValueError("need at least one rank")
    base = dict(os.environ if env is None else env)
    port = free_port(addr)
    procs = []
    for r in range(n):
        e = {**base, "RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
             "MASTER_ADDR": addr, "MASTER_PORT": str(port)}
        e.pop("TORCHELASTIC_RUN_ID", None)
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
    deadline = None if timeout is None else time.monotonic() + timeout
    out0, rc = "", 0
    pending = set(range(n))
    try:
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0:
                    # rank 0's pipe is drained through communicate() ONLY (it keeps what earlier, timed-out calls already read;
                    # a direct read of p.stdout afterwards would lose that part)
                    try:
                        o, _ = p.communicate(timeout=0.2)
                    except subprocess.TimeoutExpired:
                        continue
                    out0 = o or ""
                    code = p.returncode
                else:
                    code = p.poll()
                    if code is None:
                        continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
            if rc != 0 and 0 in pending:
                # a rank failed while rank 0 is still running: give rank 0 a moment to print its line (the documented
                # fall-back exits non-zero on every rank), then stop waiting
                try:
                    o, _ = procs[0].communicate(timeout=120.0)
                    out0 = o or ""
                    pending.discard(0)
                except subprocess.TimeoutExpired:
                    pass
                break
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            if pending and 0 not in pending:
                time.sleep(0.05)
    finally:
        for r in pending:  # exact PIDs this function started
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    return rc, out0


class CommUnavailable(RuntimeError):
    """The RCCL communicator could not be created on at least one rank; EVERY rank raises this (same verdict everywhere)."""


def common_verdict(rank: int, world: int, error: bytes, *, addr: str, port: int) -> None:
    """One verdict for the whole job: a rank whose ``ncclCommInitRank`` failed must not fall back alone while its peers wait
    inside a collective for it.  Every rank ships ``b""`` (ok) or its error text over the rendezvous socket; if any rank
    failed, every rank raises the same :class:`CommUnavailable` (and the caller tears its own communicator down)."""
    verdicts = exchange(rank, world, error, addr=addr, port=port)
    bad = [(r, v.decode(errors="replace")) for r, v in enumerate(verdicts) if v]
    if bad:
        raise CommUnavailable("; ".join(f"rank {r}: {m}" for r, m in bad))


class HostSync:
    """Barrier / max-reduction over the rendezvous sockets alone (no device, no RCCL): what a launcher script or a benchmark falls
    back to for its TIMING when the RCCL communicator cannot be created -- never a substitute for the psi gather."""

    def __init__(self, rank: int, world: int, *, addr: str, port: int):
        self.rank, self.world, self.addr, self.port = rank, world, addr, port

    @classmethod
    def from_env(cls, env=None) -> "HostSync":
        env = os.environ if env is None else env
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), addr=env.get("MASTER_ADDR", "127.0.0.1"),
                   port=rendezvous_port(env))

    def allreduce_max(self, value: float) -> float:
        parts = exchange(self.rank, self.world, struct.pack("!d", float(value)), addr=self.addr, port=self.port)
        return max(struct.unpack("!d", p)[0] for p in parts)

    def barrier(self) -> None:
        self.allreduce_max(0.0)


def negotiate_communicator(rank: int, world: int, *, preflight, init, finalize, addr: str, port: int) -> None:
    """Bring the communicator up in two steps with one verdict after each, so that no rank is ever alone inside a collective.

    ``preflight()`` is the LOCAL part (library loaded, this rank's device selectable, ABI version) and ``init()`` the collective
    part (``ncclCommInitRank``); both return ``b""`` or an error text.  The pre-flight verdicts are exchanged first: a rank that
    fails there (no librccl, ``LOCAL_RANK`` beyond the visible devices, wrong ABI) makes EVERY rank raise :class:`CommUnavailable`
    before anybody calls ``init`` -- its peers would otherwise sit inside ``ncclCommInitRank`` waiting for a rank that never
    comes, and the failed rank at a rendezvous nobody serves.  The verdicts of ``init`` are exchanged the same way; a rank whose
    own ``init`` succeeded tears its communicator down (``finalize``) when a peer's did not."""
    common_verdict(rank, world, preflight(), addr=addr, port=port)
    err = init()
    try:
        common_verdict(rank, world, err, addr=addr, port=port)
    except CommUnavailable:
        if not err:
            finalize()
        raise


# ------------------------------------------------------------------------------------------------- RCCL communicator
class EnsembleComm:
    """The RCCL communicator of one rank (``scpn_comm_*``).  Create with :meth:`from_env` BEFORE the first device call of the
    process: the unique id travels over TCP first, ``scpn_comm_init`` then selects this rank's GPU."""

    def __init__(self, rank: int, world: int, device: int, *, addr: str, port: int):
        from . import _lib

        self._lib = _lib
        self.rank, self.world, self.device, self.addr, self.port = rank, world, device, addr, port
        L = _lib.lib()
        ident = C.create_string_buffer(ID_BYTES)
        payload = b""
        if rank == 0:
            # a failure to create the id (no librccl on the node) is shipped to the other ranks in place of the id, so that every
            # rank leaves the rendezvous with the same verdict instead of waiting for a server that never comes
            try:
                _lib.check(L.scpn_comm_unique_id(ident), "scpn_comm_unique_id")
                payload = ident.raw
            except Exception as exc:
                payload = b"ERR " + str(exc).encode()[:200]
        got = exchange(rank, world, payload, addr=addr, port=port)[0]
        if len(got) != ID_BYTES:
            raise ConnectionError("no RCCL unique id from rank 0: " + (got[4:].decode(errors="replace") if got[:4] == b"ERR " else "malformed"))

        def text(rc: int, what: str) -> bytes:
            return b"" if rc == _lib.SCPN_OK else (_lib.last_error() or f"{what} failed").encode()[:300]

        def preflight() -> bytes:
            return text(L.scpn_comm_prepare(device, None), "scpn_comm_prepare")

        def init() -> bytes:
            # RCCL prints a version banner on the C stdout of rank 0 while it initialises; callers of this module own stdout
            # (bench.py prints exactly one JSON line there), so fd 1 points at stderr for the duration of the call and C stdio is
            # flushed before it is restored
            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                rc = L.scpn_comm_init(rank, world, got, device)
                try:
                    C.CDLL(None).fflush(None)
                except Exception:
                    pass
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            return text(rc, "scpn_comm_init")

        negotiate_communicator(rank, world, preflight=preflight, init=init, finalize=L.scpn_comm_finalize, addr=addr, port=port)
        self._recv = None
        self._recv_bytes = 0
        self._per_checked: set[int] = set()

    @classmethod
    def from_env(cls, env=None) -> "EnsembleComm":
        env = os.environ if env is None else env
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0")),
                   addr=env.get("MASTER_ADDR", "127.0.0.1"), port=rendezvous_port(env))

    def info(self) -> dict:
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        self._lib.check(self._lib.lib().scpn_comm_info(C.byref(r), C.byref(w), C.byref(v)), "scpn_comm_info")
        return {"rank": r.value, "world": w.value, "rccl_version": v.value}

    def allgather_psi(self, es, total: int | None = None) -> int:
        """Launch the all-gather of ``es``'s psi blocks (asynchronous); returns the device address of the gathered
        ``[world][per][nz][nr]`` buffer, valid after :meth:`wait`.  ``per = ceil(total / world)`` is the block EVERY rank sends:
        a rank that owns fewer members (the last block of an uneven ``total``, see :func:`shard_range`) pads its block with
        zeros, so all ranks issue the collective with the same count and the receive buffer holds ``world`` whole blocks.
        Without ``total`` the block is this rank's batch, which then has to be the same on every rank -- checked once per
        block size over the rendezvous socket before the first collective, a mismatch raises on every rank."""
        per = es.batch if total is None else -(-int(total) // self.world)
        if per < es.batch:
            raise ValueError(f"rank {self.rank} holds {es.batch} members but the common block is {per}")
        if per not in self._per_checked:
            seen = {struct.unpack("!i", p)[0] for p in exchange(self.rank, self.world, struct.pack("!i", per), addr=self.addr, port=self.port)}
            if len(seen) != 1:
                raise ValueError(f"ranks disagree on the all-gather block size: {sorted(seen)} members")
            self._per_checked.add(per)
        need = self.world * per * es.NZ * es.NR * 8
        L = self._lib.lib()
        if self._recv is None or self._recv_bytes != need:
            self.wait()
            if self._recv is not None:
                self._lib.check(L.scpn_device_free(self._recv), "scpn_device_free")
            p = C.c_void_p()
            self._lib.check(L.scpn_device_alloc(need, C.byref(p)), "scpn_device_alloc")
            self._recv, self._recv_bytes = p, need
        self._lib.check(L.scpn_comm_allgather_psi(es._h, per, self._recv), "scpn_comm_allgather_psi")
        return int(self._recv.value)

    @staticmethod
    def unpad(gathered: np.ndarray, total: int) -> np.ndarray:
        """``[world][per][nz][nr]`` (as gathered) -> ``[total][nz][nr]`` in member order, padding dropped."""
        world, per = gathered.shape[:2]
        sizes = shard_sizes(total, world)
        return np.concatenate([gathered[r, :sizes[r]] for r in range(world)], axis=0)

    def wait(self) -> None:
        self._lib.check(self._lib.lib().scpn_comm_wait(), "scpn_comm_wait")

    def download(self, shape: tuple[int, ...]) -> np.ndarray:
        """The gathered buffer as a host array (after :meth:`wait`)."""
        out = np.empty(shape, dtype=np.float64)
        if out.nbytes != self._recv_bytes:
            raise ValueError("shape does not match the gathered buffer")
        self.wait()
        self._lib.check(self._lib.lib().scpn_device_download(out.ctypes.data, self._recv, out.nbytes), "scpn_device_download")
        return out

    def allreduce_max(self, value: float) -> float:
        """Max over the ranks (blocking): the per-run time of the slowest rank, and the barrier of the timed region."""
        v = C.c_double(float(value))
        self._lib.check(self._lib.lib().scpn_comm_allreduce_max(C.byref(v)), "scpn_comm_allreduce_max")
        return float(v.value)

    def barrier(self) -> None:
        self.allreduce_max(0.0)

    def close(self) -> None:
        L = self._lib.lib()
        self.wait()
        if self._recv is not None:
            L.scpn_device_free(self._recv)
            self._recv = None
        L.scpn_comm_finalize()
