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
    # (4) the psi all-gather with an uneven total: every rank sends a block padded to ceil(TOTAL / world) members (what
    #     EnsembleComm.allgather_psi does on the device), the ranks agree on the block size first, the padding is dropped after
    from scpn_fusion_core_amd.distributed import CommUnavailable, EnsembleComm, common_verdict
    import struct

    per = -(-TOTAL // world)
    seen = {struct.unpack("!i", p)[0] for p in exchange(rank, world, struct.pack("!i", per), addr="127.0.0.1", port=port)}
    assert seen == {per}
    block = np.zeros((per, 17, 17))
    block[: b - a] = psi
    parts = exchange(rank, world, block.tobytes(), addr="127.0.0.1", port=port)
    assert len({len(p) for p in parts}) == 1  # equal counts on every rank: the condition ncclAllGather needs
    gathered = np.stack([np.frombuffer(p, dtype=np.float64).reshape(per, 17, 17) for p in parts])
    padded_psi = EnsembleComm.unpad(gathered, TOTAL)
    # (5) one verdict for the whole job: rank 1 "fails" its communicator, BOTH ranks must raise
    raised = ""
    try:
        common_verdict(rank, world, b"ncclCommInitRank: invalid usage" if rank == 1 else b"", addr="127.0.0.1", port=port)
    except CommUnavailable as exc:
        raised = str(exc)
    common_verdict(rank, world, b"", addr="127.0.0.1", port=port)  # and all-ok passes on both
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), psi=all_psi, table=all_tab, ident=np.frombuffer(got, dtype=np.uint8), tmax=tmax,
             padded_psi=padded_psi, raised=np.array(raised))


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
        assert np.array_equal(got["padded_psi"], psi_ref)
        assert "rank 1: ncclCommInitRank" in str(got["raised"])


def _preflight_worker(rank, world, port, out_dir, fail_rank):
    """A rank whose LOCAL pre-flight fails (no such device, no librccl) must take every rank out BEFORE the collective init."""
    import time

    from scpn_fusion_core_amd.distributed import CommUnavailable, negotiate_communicator

    calls = []

    def preflight():
        calls.append("preflight")
        return b"device 1 is not visible to this rank (1 device(s))" if rank == fail_rank else b""

    def init():  # stands for ncclCommInitRank: a collective that never returns when a peer is missing
        calls.append("init")
        if fail_rank >= 0:
            time.sleep(600)
        return b""

    t0 = time.monotonic()
    raised = ""
    try:
        negotiate_communicator(rank, world, preflight=preflight, init=init, finalize=lambda: calls.append("finalize"), addr="127.0.0.1",
                               port=port)
    except CommUnavailable as exc:
        raised = str(exc)
    np.savez(os.path.join(out_dir, f"pre{rank}.npz"), raised=np.array(raised), calls=np.array(",".join(calls)), dt=time.monotonic() - t0)


def _run_preflight(tmp_path, fail_rank):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_preflight_worker, args=(r, 2, port, str(tmp_path), fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return [np.load(tmp_path / f"pre{r}.npz") for r in range(2)]


def test_a_rank_failing_before_the_collective_takes_every_rank_out(tmp_path):
    for got in _run_preflight(tmp_path, fail_rank=1):
        assert "rank 1: device 1 is not visible" in str(got["raised"])
        assert str(got["calls"]) == "preflight"  # nobody entered the collective, nothing to tear down
        assert float(got["dt"]) < 30.0


def test_two_step_negotiation_passes_when_every_rank_is_ok(tmp_path):
    for got in _run_preflight(tmp_path, fail_rank=-1):
        assert str(got["raised"]) == "" and str(got["calls"]) == "preflight,init"


def test_gather_rows_rejects_wrong_block_size():
    import pytest

    from scpn_fusion_core_amd.distributed import gather_rows

    with pytest.raises(ValueError):
        gather_rows(0, 1, np.zeros((2, 4)), 3, addr="127.0.0.1", port=1)
    assert gather_rows(0, 1, np.ones((3, 2)), 3, addr="127.0.0.1", port=1).shape == (3, 2)


# ------------------------------------------------------------------------------------------------- `bench.py --gpus N` launcher
_STUB = """
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert "TORCHELASTIC_RUN_ID" not in os.environ
sys.path.insert(0, {root!r})
from scpn_fusion_core_amd.distributed import HostSync
t = HostSync.from_env().allreduce_max(float(r))          # the ranks really meet on the port they were given
print("chatter from rank", r)                            # non-JSON stdout of rank 0 must not become the result line
if r == {fail_rank}:
    sys.exit(7)
if r == 0:
    import time
    time.sleep({rank0_delay})                            # rank 0 has more to do than its peers (result line, parity gate)
    print(json.dumps({{"n_gpus": w, "tmax": t, "argv": sys.argv[1:]}}))
    sys.exit({rank0_rc})
"""


def _run_stub(tmp_path, n, fail_rank=-1, rank0_delay=0.0, rank0_rc=0):
    import json
    import sys

    from scpn_fusion_core_amd.distributed import spawn_ranks
    from tests.conftest import ROOT

    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB.format(root=str(ROOT), fail_rank=fail_rank, rank0_delay=rank0_delay, rank0_rc=rank0_rc))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    rc, out = spawn_ranks([sys.executable, str(stub), "--gpus", str(n)], n, env=env, timeout=120)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    return rc, [json.loads(ln) for ln in lines]


def test_spawner_starts_one_rank_per_gpu_and_forwards_rank0(tmp_path):
    rc, lines = _run_stub(tmp_path, 3)
    assert rc == 0 and len(lines) == 1
    assert lines[0] == {"n_gpus": 3, "tmax": 2.0, "argv": ["--gpus", "3"]}


def test_spawner_reports_a_failed_rank(tmp_path):
    rc, _lines = _run_stub(tmp_path, 2, fail_rank=1)
    assert rc == 7


def test_spawner_keeps_rank0_line_when_every_rank_exits_nonzero(tmp_path):
    """The documented no-communicator fall-back: every rank exits 3, the peers before rank 0 has printed its (invalid-marked) line."""
    rc, lines = _run_stub(tmp_path, 2, fail_rank=1, rank0_delay=1.0, rank0_rc=7)
    assert rc == 7 and len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_bench_refuses_a_world_size_that_contradicts_gpus(tmp_path):
    import subprocess
    import sys

    from tests.conftest import ROOT

    env = {**os.environ, "WORLD_SIZE": "1", "RANK": "0"}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_bench_ends_at_once_when_a_rank_has_no_gpu_and_no_communicator():
    """No GPU here: both ranks fail the communicator's pre-flight, tell each other that they have no device either, and the job ends
    with the documented exit code 3 within seconds -- not at a barrier the missing rank never reaches."""
    import subprocess
    import sys
    import time

    from tests.conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-roofline", "--no-newton", "--no-configs"], env=env, capture_output=True, text=True, timeout=170)
    if "not every rank has a GPU" not in r.stderr:  # (a GPU box: the rehearsal there is tools' business, not this CPU test's)
        import pytest

        pytest.skip("GPUs are visible to this process")
    assert r.returncode == 3 and not r.stdout.strip() and time.monotonic() - t0 < 60.0


# ---- strong scaling (bench.py --total-members): a FIXED shape-scan ensemble cut over the ranks
STRONG_TOTAL = 5  # odd: rank 1's block is one member short and is padded in the gather


def _scan_solve(start, stop):
    """Members [start, stop) of the config-5 coil/shape scan (currents, I_p AND kappa through the coil heights) on a 17 x 17 grid,
    solved by the CPU oracle's free-boundary loop (what each rank's device solve is checked against on the GPU box)."""
    from oracle import oracle as orc
    from scpn_fusion_core_amd.cases import member_cfg as scan_cfg, shape_scan_member_table

    cfg = iter_like_cfg(17, tol=1e-5, max_iter=80)
    t = shape_scan_member_table(cfg, STRONG_TOTAL)
    psi = []
    for m in range(start, stop):
        cm = scan_cfg(cfg, t["currents"][m], t["targets"][m], t["coil_r"][m], t["coil_z"][m])
        co = cm["coils"]
        r = orc.free_boundary_solve(cm, [c["r"] for c in co], [c["z"] for c in co], [c["current"] for c in co], [1.0] * len(co),
                                    max_outer=3, tol=1e-4)
        psi.append(r["psi"])
    return np.array(psi).reshape(-1, 17 * 17)


def _strong_worker(rank, world, port, out_dir):
    from scpn_fusion_core_amd.distributed import EnsembleComm, exchange, plan_ensemble

    plan = plan_ensemble(rank, world, members_per_rank=32, total_members=STRONG_TOTAL)
    psi = _scan_solve(plan["start"], plan["stop"])
    block = np.zeros((plan["block"], 17 * 17))
    block[: plan["members"]] = psi
    parts = exchange(rank, world, block.tobytes(), addr="127.0.0.1", port=port)
    gathered = np.stack([np.frombuffer(p, dtype=np.float64).reshape(plan["block"], 17 * 17) for p in parts])
    np.savez(os.path.join(out_dir, f"strong{rank}.npz"), psi=EnsembleComm.unpad(gathered, STRONG_TOTAL), plan=np.array(
        [plan["total"], plan["start"], plan["stop"], plan["members"], plan["block"]]), scaling=np.array(plan["scaling"]))


def test_strong_scaling_plan_and_two_rank_shape_scan(tmp_path):
    import pytest

    from scpn_fusion_core_amd.cases import shape_scan_member_table
    from scpn_fusion_core_amd.distributed import plan_ensemble

    # the plan itself: 256 members over 1 / 2 / 4 / 8 ranks are even blocks; weak mode multiplies; empty blocks are refused
    for world in (1, 2, 4, 8):
        plans = [plan_ensemble(r, world, members_per_rank=32, total_members=256) for r in range(world)]
        assert [p["members"] for p in plans] == [256 // world] * world and plans[-1]["stop"] == 256 and plans[0]["scaling"] == "strong"
        weak = plan_ensemble(world - 1, world, members_per_rank=32)
        assert weak["scaling"] == "weak" and weak["total"] == 32 * world and weak["members"] == weak["block"] == 32
    with pytest.raises(ValueError, match="non-empty"):
        plan_ensemble(0, 4, total_members=5)  # blocks of 2: the fourth rank would hold nothing
    # the scan table: member m does not depend on the ensemble's size in its coil geometry, kappa spans [1.5, 1.9]
    cfg = iter_like_cfg(17)
    t5, t256 = shape_scan_member_table(cfg, STRONG_TOTAL), shape_scan_member_table(cfg, 256)
    assert np.array_equal(t5["coil_z"], t256["coil_z"][:STRONG_TOTAL]) and np.array_equal(t5["currents"], t256["currents"][:STRONG_TOTAL])
    assert np.array_equal(t5["targets"], t256["targets"][:STRONG_TOTAL]) and np.ptp(t256["targets"]) > 2.0  # I_p too (own generator stream), +-10 % of 15
    assert 1.5 <= t256["kappa"].min() < 1.52 and 1.88 < t256["kappa"].max() <= 1.9
    assert np.allclose(t256["coil_z"][:, 0], 1.2 * t256["kappa"] * 2.0) and np.allclose(t256["coil_z"][:, 3], -1.1 * 1.2 * t256["kappa"] * 2.0)
    assert np.array_equal(t256["coil_z"][:, 4], np.zeros(256)) and np.array_equal(t256["coil_r"], np.tile(t256["coil_r"][0], (256, 1)))
    # two ranks solve their blocks of the 5-member scan and gather them (padded block of rank 1): equal to the serial solve
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    serial = _scan_solve(0, STRONG_TOTAL)
    assert len({serial[m].tobytes() for m in range(STRONG_TOTAL)}) == STRONG_TOTAL  # (the members really differ)
    for rank, (a, b) in enumerate(((0, 3), (3, 5))):
        got = np.load(tmp_path / f"strong{rank}.npz")
        assert list(got["plan"]) == [STRONG_TOTAL, a, b, b - a, 3] and str(got["scaling"]) == "strong"
        assert np.array_equal(got["psi"], serial)


def _stats_worker(rank, world, port, out_dir):
    import json

    from scpn_fusion_core_amd.distributed import gather_rank_stats

    rows = gather_rank_stats(rank, world, {"members": 32, "solve_ms": 100.0 + rank, "gather_wait_ms": 0.5 * rank, "final_barrier_ms": 1.0,
                                           "picard_iterations_max": 419 - rank, "picard_iterations_min": 413, "converged_members": 32},
                             addr="127.0.0.1", port=port)
    with open(os.path.join(out_dir, f"stats{rank}.json"), "w") as fh:
        json.dump(rows, fh)


def test_per_rank_table_of_the_bench_line(tmp_path):
    """``bench.py``'s ``per_rank`` table: one record per rank, in rank order, the same on every rank -- the keys a first 8-GPU run
    needs to tell imbalance (solve_ms, picard_iterations_max) from gather cost (gather_wait_ms, final_barrier_ms)."""
    import json

    from scpn_fusion_core_amd.distributed import gather_rank_stats

    one = gather_rank_stats(0, 1, {"members": 32}, addr="127.0.0.1", port=1)  # world 1: no socket
    assert one == [{"rank": 0, "members": 32}]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tables = [json.load(open(tmp_path / f"stats{r}.json")) for r in range(2)]
    assert tables[0] == tables[1] and [row["rank"] for row in tables[0]] == [0, 1]
    for row in tables[0]:
        assert {"rank", "members", "solve_ms", "gather_wait_ms", "final_barrier_ms", "picard_iterations_max", "picard_iterations_min",
                "converged_members"} <= set(row)
    assert tables[0][1]["solve_ms"] == 101.0 and tables[0][1]["picard_iterations_max"] == 418
    # the bench line wires exactly this table (static check of the source: bench.py needs a GPU to run)
    src = open(os.path.join(os.path.dirname(__file__), "..", "bench.py")).read()
    assert 'line["per_rank"] = rank_table' in src and "gather_rank_stats(rank, world" in src and '"gather_ms"' in src
