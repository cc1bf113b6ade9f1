"""world_size=2 gloo rehearsal of the multi-GPU path: pairs shard with no data-path
collective, one all-gather of fixed-size result records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cvo_slam_amd import shard


def test_shard_range_partitions_exactly():
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                rg = shard.shard_range(n, r, world)
                seen += list(rg)
            assert seen == list(range(n))
            sizes = [len(shard.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(shard.shard_range(512, 3, 8)) == list(range(192, 256))     # BASELINE config 4: 64 pairs per GPU


def test_c_abi_shard_range_is_the_python_one(hiplib):
    """cvo_shard_range (include/cvo_hip.h) and shard.shard_range deal the same contiguous blocks (no GPU needed)."""
    from cvo_slam_amd import api
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert list(api.shard_range(n, r, world)) == list(shard.shard_range(n, r, world))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_range(n_pairs, rank, world)
    # a rank "aligns" its own pairs: the record of global pair p is a deterministic function of p
    local = torch.stack([torch.arange(shard.RESULT_FLOATS, dtype=torch.float32) + 100.0 * p for p in mine]) if len(mine) else \
        torch.zeros((0, shard.RESULT_FLOATS))
    table = shard.gather_results(local, n_pairs, world)
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [8, 13])
def test_gather_results_world2(tmp_path, n_pairs):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    want = np.stack([np.arange(shard.RESULT_FLOATS, dtype=np.float32) + 100.0 * p for p in range(n_pairs)])
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"table_{r}.npy"), want)   # every rank holds the full table in pair order


def _worker_blocks(rank, world, port, n_pairs, fail_rank, out_dir):
    """The padded form the C ABI gathers (cvo_batch_gather_results_padded): every rank sends cvo_shard_block records -- its own,
    then padding with status CVO_ERR_PADDING; a rank whose launch failed sends its error code in every record."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from cvo_slam_amd import api
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_range(n_pairs, rank, world)
    blk = api.shard_block(n_pairs, world)
    block = torch.zeros((blk, shard.RESULT_FLOATS)); block[:, 15] = float(api.CVO_ERR_PADDING)
    for k, p in enumerate(mine):
        block[k] = torch.arange(shard.RESULT_FLOATS, dtype=torch.float32) + 100.0 * p; block[k, 15] = 0.0
    if rank == fail_rank:
        block[:] = 0.0; block[:, 15] = float(api.CVO_ERR_INVALID)
    table = shard.gather_blocks(block, world)
    out, err = api.compact_records(table.numpy(), n_pairs, world)
    np.save(os.path.join(out_dir, f"blocks_{rank}.npy"), out); np.save(os.path.join(out_dir, f"err_{rank}.npy"), np.array([err]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs,world,fail_rank", [(7, 2, -1), (10, 2, -1), (10, 3, -1), (2, 3, -1), (10, 2, 1)])
def test_padded_blocks_gather_and_compact(hiplib, tmp_path, n_pairs, world, fail_rank):
    """Uneven shards (the reference's <= 10 loop-closure candidates, keyframe_graph.cpp:622-731) and a failed rank: every rank
    enters the collective with a block of the same size, cvo_compact_records restores global pair order and reports the failure."""
    from cvo_slam_amd import api
    mp.spawn(_worker_blocks, args=(world, _free_port(), n_pairs, fail_rank, str(tmp_path)), nprocs=world, join=True)
    want = np.stack([np.arange(shard.RESULT_FLOATS, dtype=np.float32) + 100.0 * p for p in range(n_pairs)]); want[:, 15] = 0
    for r in range(world):
        got = np.load(tmp_path / f"blocks_{r}.npy"); err = int(np.load(tmp_path / f"err_{r}.npy")[0])
        if fail_rank < 0:
            assert err == 0
            np.testing.assert_array_equal(got, want)
        else:
            assert err == api.CVO_ERR_INVALID                               # every rank learns of it
            ok = [p for q in range(world) if q != fail_rank for p in shard.shard_range(n_pairs, q, world)]
            np.testing.assert_array_equal(got[ok], want[ok])
            bad = list(shard.shard_range(n_pairs, fail_rank, world))
            assert np.all(got[bad, 15] == api.CVO_ERR_INVALID)
