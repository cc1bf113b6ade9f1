"""Multi-GPU sharding of independent frame pairs.

Pairs are independent alignments, so they shard with no data-path collective:
rank r owns a contiguous block of pair indices and uploads only those clouds to its
GPU.  The only exchange is one all-gather of fixed-size result records
(CVO_RESULT_FLOATS floats = 64 bytes per pair: 3x4 transform, iter, A_nonzero,
iterations_run, status) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in the
CPU tests.  The reference is single-process (no counterpart to cite); the batch
source it corresponds to is the loop-closure candidate loop, keyframe_graph.cpp:622-731.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

RESULT_FLOATS = 16


def shard_range(n_pairs: int, rank: int, world: int) -> range:
    """Contiguous block of global pair indices owned by `rank` (sizes differ by at most 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_pairs, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def gather_results(local: torch.Tensor, n_pairs: int, world: int | None = None) -> torch.Tensor:
    """All-gather the per-rank result records into the global (n_pairs, RESULT_FLOATS) table,
    in global pair order, on every rank.  `local` is (n_local, RESULT_FLOATS)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        assert local.shape[0] == n_pairs
        return local.clone()
    if local.is_cuda and dist.get_backend() != "nccl":        # rehearsal on a backend without device collectives: stage through the host
        return gather_results(local.cpu(), n_pairs, world).to(local.device)
    rank = dist.get_rank()
    mine = shard_range(n_pairs, rank, world)
    assert local.shape == (len(mine), RESULT_FLOATS), (local.shape, len(mine))
    max_local = len(shard_range(n_pairs, 0, world))          # rank 0 always holds a largest block
    send = torch.zeros((max_local, RESULT_FLOATS), dtype=local.dtype, device=local.device)
    send[: len(mine)] = local
    recv = torch.empty((world * max_local, RESULT_FLOATS), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send)
    recv = recv.view(world, max_local, RESULT_FLOATS)
    parts = [recv[r, : len(shard_range(n_pairs, r, world))] for r in range(world)]
    return torch.cat(parts, dim=0)


class _DeviceView:
    """A block of result records at a raw device address (cvo_batch_padded_records / cvo_batch_result_records) as something torch can
    wrap without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr: int, n_records: int):
        self.__cuda_array_interface__ = {"shape": (n_records, RESULT_FLOATS), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_view(ptr: int, n_records: int) -> torch.Tensor:
    return torch.as_tensor(_DeviceView(ptr, n_records), device="cuda")


def gather_blocks(block: torch.Tensor, world: int | None = None) -> torch.Tensor:
    """All-gather equal-sized record blocks (every rank: its own records, then padding -- cvo_batch_padded_records) into the
    rank-major (world * n_block, RESULT_FLOATS) table; cvo_compact_records / api.compact_records puts it in global pair order."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return block.clone()
    if block.is_cuda and dist.get_backend() != "nccl":        # rehearsal on a backend without device collectives: stage through the host
        return gather_blocks(block.cpu(), world).to(block.device)
    recv = torch.empty((world * block.shape[0], RESULT_FLOATS), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(recv, block.contiguous())
    return recv
