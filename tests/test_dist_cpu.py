"""world_size-2 gloo tests (CPU) of the N>1 path: row-range sharding + the benchmark's control plane.

The data path has no collective; what must hold is that the ranks' row ranges tile
the global table exactly, that each rank's shard of the synthetic table equals the corresponding
slice of the global table, and that results placed by row offset reproduce the single-process
result.  The CPU oracle stands in for the GPU here (this is a test of the sharding logic).
"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_path, rows_per_rank, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from infera_amd import shard, synth
    from oracle import oracle

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    row0, row1 = shard.row_range(rank, world, rows_per_rank)
    x = synth.table(42, row0, row1 - row0, 128)
    y = oracle.Model(model_path).predict(x)
    shard.barrier()
    # control plane: max over ranks of a per-rank scalar
    t = shard.max_over_ranks(float(rank + 1))
    assert t == float(world)
    # result "reassembly" = placement by row offset (here via a file per rank)
    np.save(os.path.join(outdir, f"y_{rank}.npy"), y)
    np.save(os.path.join(outdir, f"r_{rank}.npy"), np.array([row0, row1]))
    # gather the row counts to check the whole-job aggregate the bench reports
    cnt = torch.tensor([row1 - row0], dtype=torch.int64)
    dist.all_reduce(cnt)
    assert int(cnt.item()) == world * rows_per_rank
    dist.destroy_process_group()


def test_row_range_sharding_world2(tmp_path, models):
    from infera_amd import synth
    from oracle import oracle

    world, rows_per_rank = 2, 3000
    mp.spawn(_worker, args=(world, _free_port(), models["mlp"], rows_per_rank, str(tmp_path)), nprocs=world, join=True)
    full = oracle.Model(models["mlp"]).predict(synth.table(42, 0, world * rows_per_rank, 128))
    got = np.empty_like(full)
    covered = np.zeros(len(full), bool)
    for r in range(world):
        row0, row1 = np.load(tmp_path / f"r_{r}.npy")
        assert not covered[row0:row1].any()
        covered[row0:row1] = True
        got[row0:row1] = np.load(tmp_path / f"y_{r}.npy")
    assert covered.all() and np.array_equal(got, full)


def test_shard_arithmetic():
    from infera_amd import shard

    assert [shard.row_range(r, 4, 10) for r in range(4)] == [(0, 10), (10, 20), (20, 30), (30, 40)]
    for total, world in [(10, 3), (7, 8), (100, 8), (0, 2)]:
        spans = [shard.split_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [shard.chunk_owner(i, 8) for i in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert shard.max_over_ranks(3.5) == 3.5  # no process group: identity


def test_numa_aware_slot_dealing(built):
    """caller threads are dealt to the GPUs of their own socket first.  An 8-GPU node with GPUs 0-3 on NUMA
    node 0 and 4-7 on node 1: threads on node 0 round-robin over slots 0-3, threads on node 1 over 4-7; unknown topology or
    a node without GPUs falls back to round-robin over all slots; one slot is always slot 0."""
    from infera_amd import capi

    topo = [0, 0, 0, 0, 1, 1, 1, 1]
    assert [capi.choose_slot(topo, 0, k, 100 + k) for k in range(8)] == [0, 1, 2, 3, 0, 1, 2, 3]
    assert [capi.choose_slot(topo, 1, k, 100 + k) for k in range(8)] == [4, 5, 6, 7, 4, 5, 6, 7]
    assert [capi.choose_slot(topo, 2, 0, k) for k in range(10)] == [k % 8 for k in range(10)]     # a socket without GPUs
    assert [capi.choose_slot(topo, -1, 0, k) for k in range(10)] == [k % 8 for k in range(10)]    # thread's node unknown
    assert [capi.choose_slot([-1] * 4, 0, k, k) for k in range(6)] == [0, 1, 2, 3, 0, 1]          # GPU topology unknown
    assert [capi.choose_slot([1, 0, 1, 0], 0, k, 7) for k in range(4)] == [1, 3, 1, 3]             # interleaved topology
    assert capi.choose_slot([0], 1, 5, 9) == 0 and capi.choose_slot([], 0, 0, 0) == 0
    # every slot of a node gets the same share: 64 threads per node on the 8-GPU topology
    counts = [0] * 8
    for node in (0, 1):
        for k in range(64):
            counts[capi.choose_slot(topo, node, k, 0)] += 1
    assert counts == [16] * 8


def test_balanced_slot_dealing_never_starves_the_other_socket():
    """the load-aware policy home_slot() applies.  Workers that all START on one socket fill its GPUs first and spill
    to the other socket's one round later -- no slot may carry more than one thread above the least-loaded slot, and no slot
    stays idle; threads spread over both sockets get their own node's GPUs."""
    from infera_amd import capi

    topo = [0, 0, 0, 0, 1, 1, 1, 1]
    load = [0] * 8
    order = []
    for _ in range(24):  # every thread on node 0
        s = capi.choose_slot_balanced(topo, load, 0)
        load[s] += 1
        order.append(s)
    assert order[:8] == [0, 1, 2, 3, 0, 1, 2, 3] and sorted(order[8:12]) == [4, 5, 6, 7]
    assert max(load) - min(load) <= 2 and min(load) >= 2, load
    load = [0] * 8
    for k in range(32):  # alternating nodes: everyone stays local
        node = k % 2
        s = capi.choose_slot_balanced(topo, load, node)
        assert topo[s] == node
        load[s] += 1
    assert load == [4] * 8
    assert capi.choose_slot_balanced(topo, [3, 3, 3, 3, 0, 0, 0, 0], -1) == 4      # node unknown: least loaded
    assert capi.choose_slot_balanced([0], [5], 1) == 0 and capi.choose_slot_balanced([-1, -1], [1, 0], 0) == 1


def test_bench_relaunches_itself_when_gpus_is_asked_without_a_launcher():
    """`python bench.py --gpus 8` with no torch.distributed.run around it used to scan ONE shard and print "n_gpus": 1.
    The decision is a pure function: N > 1 and no WORLD_SIZE / RANK in the environment -> the launcher command for N ranks on 127.0.0.1;
    already a rank, N = 1 or --host-path (one process over N slots by definition) -> None."""
    sys.path.insert(0, ROOT)
    import bench

    argv = ["--gpus", "8", "--steps", "3", "--share-device", "0"]
    cmd = bench.relaunch_command(8, False, argv, {})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and cmd[-len(argv):] == argv
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536 and os.path.basename(cmd[-len(argv) - 1]) == "bench.py"
    assert bench.relaunch_command(8, False, argv, {"WORLD_SIZE": "8", "RANK": "3"}) is None   # already a rank of a launch
    assert bench.relaunch_command(1, False, ["--gpus", "1"], {}) is None
    assert bench.relaunch_command(8, True, argv + ["--host-path"], {}) is None
    # ... and a rank whose WORLD_SIZE disagrees with --gpus refuses to run (it used to pass silently when WORLD_SIZE was unset)
    import subprocess

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in p.stderr
