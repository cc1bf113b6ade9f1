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


def test_gathers_on_the_communicators_own_stream(hiplib):
    """The fallback mode of include/cvo_hip.h ("ORDER INVARIANT"): every gather of a communicator on one stream of its own, behind an event of the align launch,
    the launch's stream continuing behind the gather -- two batch objects in flight on their own streams, the records in place when each wait returns."""
    import torch
    ca = hiplib
    from cvo_slam_amd import api
    assert "librccl" in api.comm_library_path()
    pairs = _pairs(4, 720)
    comm = ca.CvoComm(api.comm_unique_id(), 1, 0, device=0)
    comm.set_gather_stream(True)
    bs = [ca.CvoBatch(len(pairs)) for _ in range(2)]
    for b in bs:
        for i, p in enumerate(pairs):
            b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    recvs = [torch.full((len(pairs), api.RESULT_FLOATS), -1.0, dtype=torch.float32, device="cuda") for _ in bs]
    for rnd in range(3):
        for b, r in zip(bs, recvs):
            r.fill_(-1.0); torch.cuda.synchronize()
            b.reset_states(); b.align_async(len(pairs)); b.gather_results(comm, len(pairs), r.data_ptr())
        for b, r in zip(bs, recvs):
            res = b.wait(len(pairs))
            np.testing.assert_array_equal(r.cpu().numpy(), _records(res))
    for b in bs:
        b.close()
    comm.close()


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


# ----------------------------------------------------------------------------- uneven shards, ranks without pairs, failed launches
def test_padded_block_of_a_short_rank_and_status_records(hiplib):
    """A rank whose block is shorter than the longest one (cvo_shard_range deals 2,2,1,1,... for the reference's <= 10 loop-closure
    candidates) sends n_block records: its own, then padding with status CVO_ERR_PADDING.  A rank whose launch failed sends its
    error code in every record -- and still enters the collective (here: one-rank communicator, the call must simply return)."""
    import torch
    ca = hiplib
    from cvo_slam_amd import api
    pairs = _pairs(3, 760)
    comm = ca.CvoComm(api.comm_unique_id(), 1, 0, device=0)
    b = ca.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    n_block = 5                                                      # two padding records: beyond the batch's max_pairs + 1, the table has to grow with its contents
    recv = torch.full((n_block, api.RESULT_FLOATS), -1.0, dtype=torch.float32, device="cuda")
    for rnd in range(2):                                             # second round: the padding is already in place, nothing is refilled
        b.reset_states(); b.align_async(len(pairs))
        b.gather_results_padded(comm, len(pairs), n_block, recv.data_ptr())
        res = b.wait(len(pairs))
        got = recv.cpu().numpy()
        np.testing.assert_array_equal(got[: len(pairs)], _records(res))
        assert np.all(got[len(pairs):, :15] == 0) and np.all(got[len(pairs):, 15] == api.CVO_ERR_PADDING)
    # a launch that cannot be made (more pairs than the batch holds): the error code travels in the records
    with pytest.raises(ca.CvoError) as e:
        b.align_async(len(pairs) + 1)
    b.gather_results_padded(comm, 0, n_block, recv.data_ptr(), launch_status=e.value.code)
    torch.cuda.synchronize()
    got = recv.cpu().numpy()
    assert np.all(got[:, 15] == e.value.code) and np.all(got[:, :15] == 0)
    # and the next good launch gathers good records again
    b.reset_states(); b.align_async(len(pairs)); b.gather_results_padded(comm, len(pairs), len(pairs) + 1, recv.data_ptr())
    res = b.wait(len(pairs)); got = recv.cpu().numpy()
    np.testing.assert_array_equal(got[: len(pairs)], _records(res)); assert got[len(pairs), 15] == api.CVO_ERR_PADDING
    # a rank with no pairs at all (fewer pairs than ranks) only sends padding; it has never launched anything
    b2 = ca.CvoBatch(1)
    b2.gather_results_padded(comm, 0, 2, recv.data_ptr())
    torch.cuda.synchronize()
    assert np.all(recv.cpu().numpy()[:2, 15] == api.CVO_ERR_PADDING)
    # a rank whose block cannot be prepared (it asks for a record no launch has written) returns the error AND has entered the collective:
    # its peers find CVO_ERR_RANK_FAILED records in the gathered table instead of waiting for the rank forever
    recv.fill_(-1.0)
    with pytest.raises(ca.CvoError) as e2:
        b2.gather_results_padded(comm, 1, 2, recv.data_ptr())
    assert "collective entered" in str(e2.value)
    torch.cuda.synchronize()
    got = recv.cpu().numpy()
    assert np.all(got[:2, 15] == api.CVO_ERR_RANK_FAILED) and np.all(got[:2, :15] == 0) and np.all(got[2:] == -1.0)
    _, first_err = api.compact_records(got[:2], 2, 1)
    assert first_err == api.CVO_ERR_RANK_FAILED
    # the same for a block larger than the communicator's standing failure block (1024 records): it grows
    big = torch.full((1500, api.RESULT_FLOATS), -1.0, dtype=torch.float32, device="cuda")
    with pytest.raises(ca.CvoError):
        b.gather_results_padded(comm, len(pairs) + 1, 1500, big.data_ptr())   # more records than the last launch aligned
    torch.cuda.synchronize()
    assert np.all(big.cpu().numpy()[:, 15] == api.CVO_ERR_RANK_FAILED)
    # and the rank is fine afterwards
    b.reset_states(); b.align_async(len(pairs)); b.gather_results_padded(comm, len(pairs), len(pairs) + 1, recv.data_ptr())
    res = b.wait(len(pairs)); got = recv.cpu().numpy()
    np.testing.assert_array_equal(got[: len(pairs)], _records(res)); assert got[len(pairs), 15] == api.CVO_ERR_PADDING
    b.close(); b2.close(); comm.close()


def test_multi_object_with_uneven_counts_and_a_failed_device(hiplib):
    ca = hiplib
    from cvo_slam_amd import api
    pairs = _pairs(4, 780)
    ref = ca.CvoBatch(len(pairs))
    for i, p in enumerate(pairs):
        ref.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    want = _records(ref.align(3)); ref.close()
    M = ca.CvoMulti([0], max_pairs_per_device=len(pairs))
    b = M.batch(0)
    for i, p in enumerate(pairs):
        b.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    M.align_async_v([3])
    np.testing.assert_array_equal(M.wait(0), want)
    with pytest.raises(ca.CvoError):                                  # a pair count the device's batch cannot hold: refused before anything is enqueued
        M.align_async_v([len(pairs) + 1])
    M.close()


def _rank_main_padded(rank, world, port, n_pairs, out_dir):
    """Two ranks sharing the one GPU, each running the real HIP path on its block; the blocks they exchange are the C ABI's padded
    ones (cvo_batch_padded_records), carried by gloo here and by ncclAllGather (cvo_batch_gather_results_padded) on a multi-GPU node."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import cvo_slam_amd as ca
    from cvo_slam_amd import api, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    mine = api.shard_range(n_pairs, rank, world); blk = api.shard_block(n_pairs, world)
    pairs = _pairs(n_pairs, 740)
    b = ca.CvoBatch(max(1, len(mine)))
    if len(mine):
        b.set_pairs([(pairs[g].fixed.xyz, pairs[g].fixed.feat, pairs[g].moving.xyz, pairs[g].moving.feat) for g in mine])
        b.align_async(len(mine))
    ptr = b.padded_records(len(mine), blk)
    if len(mine):
        b.wait()
    torch.cuda.synchronize()
    table = shard.gather_blocks(shard.device_view(ptr, blk), world)
    out, err = api.compact_records(table.cpu().numpy(), n_pairs, world)
    assert err == 0
    np.save(os.path.join(out_dir, f"ptable_{rank}.npy"), out)
    dist.barrier(); dist.destroy_process_group()
    b.close()


@pytest.mark.parametrize("n_pairs", [7, 10])
def test_world2_uneven_blocks_through_the_padded_records(hiplib, tmp_path, n_pairs):
    import torch.multiprocessing as mp
    world = 2 if n_pairs == 7 else 3                                  # 4+3 and 4+3+3
    mp.spawn(_rank_main_padded, args=(world, _free_port(), n_pairs, str(tmp_path)), nprocs=world, join=True)
    pairs = _pairs(n_pairs, 740)
    ref = hiplib.CvoBatch(n_pairs)
    for i, p in enumerate(pairs):
        ref.set_pair(i, p.fixed.xyz, p.fixed.feat, p.moving.xyz, p.moving.feat)
    want = _records(ref.align(n_pairs)); ref.close()
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"ptable_{r}.npy"), want)
