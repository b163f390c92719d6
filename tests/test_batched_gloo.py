"""CPU-only, world_size 2 over gloo: the batched many-arrays front end (N > 1 path).

The per-rank executor is injected: on the GPU box it is the HIP path (HipShardExecutor); here the
oracle stands in for the device so that sharding, ownership and the end-of-batch all-gather are
exercised without a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_partition_the_batch():
    from vulkan_radix_sort_amd.batched import shard_indices
    for world in (1, 2, 3, 8):
        for num in (0, 1, 7, 8, 9, 64):
            seen = []
            for r in range(world):
                idx = shard_indices(num, r, world)
                assert all(i % world == r for i in idx)
                seen += idx
            assert sorted(seen) == list(range(num))
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def test_hip_executor_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vulkan_radix_sort_amd.batched import HipShardExecutor
    with pytest.raises(RuntimeError):
        HipShardExecutor()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, num_arrays, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import load_oracle
    from vulkan_radix_sort_amd.batched import BatchedSorter
    orc = load_oracle()

    def executor(arrays):  # stand-in for the device: sorts in place, like vrdxCmdSortKeyValue
        for k, v in arrays:
            sk, sv, _ = orc.sort(k.numpy().view(np.uint32), None if v is None else v.numpy().view(np.uint32))
            k.copy_(torch.from_numpy(sk.view(np.int32)))
            if v is not None:
                v.copy_(torch.from_numpy(sv.view(np.int32)))
        return 0

    bs = BatchedSorter(executor=executor)
    mine = bs.my_indices(num_arrays)
    arrays = []
    for i in mine:  # array i is generated from seed i on its owner, like BASELINE.md's config 5
        k, v = orc.generate(i + 1, 3000 + 17 * i, 32)
        arrays.append((torch.from_numpy(k.view(np.int32).copy()), torch.from_numpy(v.view(np.int32).copy())))
    records = bs.sort_shard(arrays)
    assert len(records) == world and [r.rank for r in records] == list(range(world))
    assert all(r.status == 0 for r in records)
    assert records[rank].items == sum(3000 + 17 * i for i in mine)
    assert sum(r.items for r in records) == sum(3000 + 17 * i for i in range(num_arrays))
    assert bs.aggregate_gitems_per_s(records) > 0
    for i, (k, v) in zip(mine, arrays):
        np.save(os.path.join(out_dir, f"k{i}.npy"), k.numpy().view(np.uint32))
        np.save(os.path.join(out_dir, f"v{i}.npy"), v.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batched_sort_over_gloo(tmp_path, oracle):
    world, num_arrays = 2, 5
    mp.spawn(_worker, args=(world, _free_port(), num_arrays, str(tmp_path)), nprocs=world, join=True)
    for i in range(num_arrays):
        k, v = oracle.generate(i + 1, 3000 + 17 * i, 32)
        ek, ev, _ = oracle.sort(k, v)
        assert np.array_equal(np.load(tmp_path / f"k{i}.npy"), ek)
        assert np.array_equal(np.load(tmp_path / f"v{i}.npy"), ev)
