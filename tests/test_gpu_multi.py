"""Multi-GPU path of the C ABI: contiguous block sharding + ONE RCCL all-gather of the 64-byte result records
enqueued behind the align launch (cvo_comm_*, cvo_batch_gather_results, cvo_gather_results, cvo_multi_*).  A one-GPU box
exercises it with one-rank communicators (ncclCommInitAll on [0] and ncclCommInitRank with n_ranks = 1); the same calls
run unchanged on N devices.  The world-2 variant runs two ranks on the one GPU with the real HIP path per rank and gloo
for the exchange (RCCL refuses two ranks on one device)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(n, seed0=700):
    from cvo_slam_amd import synth
    return [synth.make_small_pair(seed0 + i, n=300 + 37 * (i % 5)) for i in range(n)]


def _records(res):
    return np.array([list(r["transform"].ravel()) + [r["iter"], r["A_nonzero"], r["iterations_run"], r["status"]] for r in res], np.float32)


def test_single_process_multi_object_gathers_what_the_batches_computed(hiplib):
    ca = hiplib
    pairs = _pairs(6)
    ref = ca.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        ref.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    want = _records(ref.align(len(pairs)))
    ref.close()
    M = ca.CvoMulti([0], max_pairs_per_device=len(pairs))            # ncclCommInitAll over the devices given
    b = M.batch(0)
    for i, p in enumerate(pairs):
        b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    for _ in range(2):                                               # twice: the gather follows every launch, no host sync in between
        b.reset_states()
        M.align_async(len(pairs))
        got = M.wait(0)
        np.testing.assert_array_equal(got, want)
    M.close()


def test_rank_communicator_and_in_stream_gather(hiplib):
    """The one-process-per-GPU form: unique id -> ncclCommInitRank -> pack + all-gather on the launch's stream."""
    import torch
    ca = hiplib
    from cvo_slam_amd import api
    pairs = _pairs(5, 720)
    comm = ca.CvoComm(api.comm_unique_id(), 1, 0, device=0)
    b = ca.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    recv = torch.full((len(pairs), api.RESULT_FLOATS), -1.0, dtype=torch.float32, device="cuda")
    b.align_async(len(pairs))
    b.gather_results(comm, len(pairs), recv.data_ptr())              # queued behind the kernel: nothing has been waited for yet
    res = b.wait(len(pairs))
    np.testing.assert_array_equal(recv.cpu().numpy(), _records(res))
    b.close(); comm.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, n_pairs, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import cvo_slam_amd as ca
    from cvo_slam_amd import api, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)                                         # both ranks share the one GPU of the box
    mine = api.shard_range(n_pairs, rank, world)
    pairs = _pairs(n_pairs, 740)
    b = ca.CvoBatch(max(1, len(mine)))
    for k, g in enumerate(mine):
        p = pairs[g]
        b.set_pair(k, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    send = torch.zeros((len(mine), api.RESULT_FLOATS), dtype=torch.float32, device="cuda")
    if len(mine):
        b.align_async(len(mine))
        b.results_to_device(send.data_ptr(), len(mine))              # the pack kernel of the gather path, same stream
        b.wait()
    table = shard.gather_results(send, n_pairs, world)               # gloo here; RCCL (cvo_batch_gather_results) on a multi-GPU node
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table.cpu().numpy())
    dist.barrier(); dist.destroy_process_group()
    b.close()


def test_world2_ranks_run_the_hip_path_and_gather(hiplib, tmp_path):
    import torch.multiprocessing as mp
    n_pairs, world = 7, 2
    mp.spawn(_rank_main, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    pairs = _pairs(n_pairs, 740)
    ref = hiplib.CvoBatch(n_pairs)
    for i, p in enumerate(pairs):
        ref.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    want = _records(ref.align(n_pairs)); ref.close()
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"table_{r}.npy"), want)    # every rank: all pairs, global order, same bits
