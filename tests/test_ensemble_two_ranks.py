"""CPU, world_size 2: the N>1 path -- member sharding, the TCP rendezvous that carries the RCCL unique id, and the assembly of
the gathered per-member blocks / result table in rank order.

No GPU and no PyTorch: the per-rank device solve is replaced by the CPU oracle (test infrastructure) and the RCCL all-gather by
the host-side socket exchange of the same module (``distributed.exchange`` / ``gather_rows``, which is also what carries the
128-byte id and the result table in production), so the sharding / padding / ordering logic that ``bench.py --gpus N`` and a
multi-GPU user run is exercised exactly."""
import multiprocessing as mp
import os
import socket

import numpy as np

from tests._problems import iter_like_cfg

TOTAL = 3  # odd on purpose: the last rank's block is shorter


def member_cfg(i):
    cfg = iter_like_cfg(17, tol=1e-5, max_iter=60)
    cfg["physics"]["plasma_current_target"] = 15.0 * (1.0 + 0.05 * i)
    return cfg


def solve_block(start, stop):
    from oracle import oracle as orc

    psi, table = [], []
    for i in range(start, stop):
        r = orc.picard_solve(member_cfg(i))
        psi.append(r["psi"])
        table.append([float(r["converged"]), r["iterations"], r["residual"], r["psi_axis"]])
    return np.array(psi).reshape(-1, 17, 17), np.array(table).reshape(-1, 4)


def _worker(rank, world, port, out_dir):
    from scpn_fusion_core_amd.distributed import ID_BYTES, exchange, gather_rows, rendezvous_port, shard_range

    env = {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world)}
    assert rendezvous_port(env) == port and rendezvous_port({**env, "TORCHELASTIC_RUN_ID": "x"}) == port + 1
    # (1) the rendezvous of the unique id: only rank 0 contributes, everybody ends up with rank 0's 128 bytes
    ident = bytes((7 * i + 1) % 256 for i in range(ID_BYTES)) if rank == 0 else b""
    got = exchange(rank, world, ident, addr="127.0.0.1", port=port)[0]
    # (2) shard, "solve", gather -- twice in a row on the same port (a new server per round)
    a, b = shard_range(TOTAL, rank, world)
    psi, table = solve_block(a, b)
    all_tab = gather_rows(rank, world, table, TOTAL, addr="127.0.0.1", port=port)
    all_psi = gather_rows(rank, world, psi.reshape(b - a, -1), TOTAL, addr="127.0.0.1", port=port).reshape(TOTAL, 17, 17)
    # (3) the timing barrier / max-reduction a launcher falls back to when no RCCL communicator can be created
    from scpn_fusion_core_amd.distributed import HostSync

    hs = HostSync(rank, world, addr="127.0.0.1", port=port)
    hs.barrier()
    tmax = hs.allreduce_max(1.5 + rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), psi=all_psi, table=all_tab, ident=np.frombuffer(got, dtype=np.uint8), tmax=tmax)


def test_two_rank_sharded_ensemble_matches_serial(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    psi_ref, tab_ref = solve_block(0, TOTAL)
    want_id = np.array([(7 * i + 1) % 256 for i in range(128)], dtype=np.uint8)
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert got["psi"].shape == (TOTAL, 17, 17)
        assert np.array_equal(got["psi"], psi_ref) and np.array_equal(got["table"], tab_ref)
        assert np.array_equal(got["ident"], want_id)
        assert float(got["tmax"]) == 2.5


def test_gather_rows_rejects_wrong_block_size():
    import pytest

    from scpn_fusion_core_amd.distributed import gather_rows

    with pytest.raises(ValueError):
        gather_rows(0, 1, np.zeros((2, 4)), 3, addr="127.0.0.1", port=1)
    assert gather_rows(0, 1, np.ones((3, 2)), 3, addr="127.0.0.1", port=1).shape == (3, 2)
