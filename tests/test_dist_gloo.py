"""N > 1 path on CPU: contig sharding + the final gather over gloo (world_size 2)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_lpt_partition_balances_and_keeps_order():
    from jaeger_amd.dist import lpt_partition
    rng = np.random.default_rng(0)
    w = rng.integers(1, 200, 1000)
    parts = lpt_partition(w, 8)
    loads = [int(w[p].sum()) for p in parts]
    assert sorted(np.concatenate(parts).tolist()) == list(range(1000))
    assert max(loads) - min(loads) <= int(w.max())
    assert all((np.diff(p) > 0).all() for p in parts)


def test_lpt_partition_eight_parts_with_empty_and_unequal_shards():
    """8 ranks (one node): fewer contigs than ranks leaves shards empty, a dominant contig leaves them unequal; the
    partition still covers every contig once, keeps FASTA order inside a shard, and restore_order undoes it."""
    from jaeger_amd.dist import lpt_partition, restore_order
    for rows in (np.array([5, 1, 9, 2, 7], np.int64),                  # 5 contigs on 8 ranks: 3 empty shards
                 np.array([400, 1, 1, 2, 1, 3, 1, 1, 2, 1, 1, 5], np.int64)):   # one contig holds 95 % of the windows
        groups = lpt_partition(rows, 8)
        assert len(groups) == 8 and sorted(np.concatenate(groups).tolist()) == list(range(len(rows)))
        assert all((np.diff(g) > 0).all() for g in groups if len(g) > 1)
        loads = np.array([int(rows[g].sum()) for g in groups])
        assert loads.max() == max(int(rows.max()), loads.max()) and (loads == 0).sum() == max(0, 8 - len(rows))
        full = np.arange(int(rows.sum()) * 2, dtype=np.float32).reshape(-1, 2)
        first = np.cumsum(rows) - rows
        parts = [np.concatenate([full[first[i]:first[i] + rows[i]] for i in g]) if len(g) else np.zeros((0, 2), np.float32)
                 for g in groups]
        np.testing.assert_array_equal(restore_order(parts, groups, rows), full)


def test_restore_order_roundtrip():
    from jaeger_amd.dist import lpt_partition, restore_order
    rng = np.random.default_rng(1)
    rows = rng.integers(1, 9, 50)
    full = rng.normal(size=(int(rows.sum()), 3)).astype(np.float32)
    first = np.cumsum(rows) - rows
    groups = lpt_partition(rows, 4)
    parts = [np.concatenate([full[first[i]:first[i] + rows[i]] for i in g]) for g in groups]
    np.testing.assert_array_equal(restore_order(parts, groups, rows), full)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from jaeger_amd.dist import gather_rows, lpt_partition, restore_order
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # windows per contig; with 4 ranks only three contigs exist: LPT leaves one rank an EMPTY shard, the others unequal ones
    rows = np.array([3, 1, 4, 1, 5, 9, 2, 6], np.int64) if world == 2 else np.array([17, 3, 11], np.int64)
    first = np.cumsum(rows) - rows
    groups = lpt_partition(rows, world)
    mine = groups[rank]
    # each rank "classifies" its contigs: logits = window index (so order is checkable)
    local = np.concatenate([np.arange(first[i], first[i] + rows[i]) for i in mine] + [np.zeros(0)]).astype(np.float32)
    local = np.stack([local, local * 2], axis=1)
    got = gather_rows(torch.from_numpy(local), dst=0)
    if rank == 0:
        full = restore_order([g.numpy() for g in got], groups, rows)
        np.save(Path(out_dir) / "full.npy", full)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gather_rows_world2(tmp_path, world):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    n = 31
    np.testing.assert_array_equal(full[:, 0], np.arange(n, dtype=np.float32))
    np.testing.assert_array_equal(full[:, 1], 2 * np.arange(n, dtype=np.float32))
