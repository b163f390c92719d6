"""Batched many-arrays front end: independent arrays sharded across the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` in
the CPU tests).  A single sort does not shard (a distributed radix sort would need an all-to-all of
every key per digit exchange -- SURVEY.md section 8e), so the unit of distribution is a whole
array: array ``i`` belongs to rank ``i % world_size`` and is sorted there by that rank's own
``VrdxSorter`` with its own stream and storage buffer.  There is NO collective on the data path;
the only communication is the end-of-batch all-gather of one small ``(status, elapsed_ns, items)``
record per rank, which doubles as the cross-GPU completion barrier for wall-time measurement.

The reference has no multi-GPU code at all (single device, single queue:
bench/vulkan_benchmark.cc:103,128-135); this module is new functionality required by
BASELINE.json ("batched variant shards independent arrays across the 8 GPUs of one node").
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence


def shard_indices(num_arrays: int, rank: int, world_size: int) -> List[int]:
    """Indices of the arrays rank `rank` owns: round-robin, array i -> rank i % world_size."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    return list(range(rank, num_arrays, world_size))


@dataclass
class BatchRecord:
    rank: int
    status: int        # 0 = ok, otherwise the device failure word (bounded look-back expired)
    elapsed_ns: int    # this rank's wall time for its shard, enqueue -> completion
    items: int         # elements this rank sorted


class HipShardExecutor:
    """Sorts this rank's arrays on its GPU through the C-ABI (libvrdx_hip.so).  No fallback.

    One ``VrdxSorter``, one stream (torch's current stream of that device) and one storage buffer per
    executor, i.e. per GPU.  ``enqueue`` never blocks the host -- like the
    ``vrdxCmdSort*`` calls it wraps -- and ``finish`` synchronises once and returns the sorter's sticky
    failure word (``vrdxHipReadSorterStatus``): the OR over EVERY sort enqueued since the last
    ``finish``, although they all share one storage buffer whose own failure word each of them clears.

    All sorts of an executor share ONE storage buffer (histogram, status rows, tickets, scratch arrays), which
    the reference's contract allows only for sorts that are not in flight at the same time.  A caller that
    switches torch's current stream between two ``enqueue`` calls would break that, so the executor orders
    every stream it is handed behind the previous one's work (``wait_stream``: a device-side dependency, the
    host never blocks) and tells the caching allocator that the storage is in use there (``record_stream``);
    a storage buffer that had to grow is kept alive until ``finish``."""

    def __init__(self, device: Optional[int] = None):
        import torch
        from .api import Sorter
        if not torch.cuda.is_available():
            raise RuntimeError("HipShardExecutor needs a GPU (there is no CPU fallback)")
        self.torch = torch
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.sorter = Sorter(self.device)
        self._storage = None
        self._retired = []  # outgrown storage buffers that sorts in flight may still use (dropped by finish())
        self._streams = []  # every stream sorts were enqueued on since the last finish()
        self._last = None   # the stream of the most recent enqueue

    def _storage_for(self, nbytes: int):
        if self._storage is None or self._storage.numel() < nbytes:
            if self._storage is not None:
                self._retired.append(self._storage)  # sorts in flight (on whatever stream) still use it
            self._storage = self.torch.empty(nbytes, dtype=self.torch.uint8, device=f"cuda:{self.device}")
        return self._storage

    def _check(self, t, what):
        if t.dtype.itemsize != 4 or t.dim() != 1 or not t.is_contiguous():
            raise TypeError(f"{what}: expected a contiguous 1-d tensor of 4-byte elements, got {t.dtype} {tuple(t.shape)}")
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"{what}: lives on {t.device}, this executor sorts on cuda:{self.device}")

    def enqueue(self, arrays: Sequence[tuple]) -> int:
        """arrays: (keys_tensor, values_tensor_or_None) pairs resident on this GPU (int32/uint32
        storage, sorted in place as uint32).  Returns the number of elements enqueued."""
        torch = self.torch
        current = torch.cuda.current_stream(self.device)
        stream = current.cuda_stream
        if all(st.cuda_stream != stream for st in self._streams):
            self._streams.append(current)
        if self._last is not None and self._last.cuda_stream != stream:
            current.wait_stream(self._last)  # the shared storage: never two sorts in flight at once
        self._last = current
        need, items = 16, 0
        for keys, values in arrays:
            self._check(keys, "keys")
            if values is not None:
                self._check(values, "values")
                if values.numel() != keys.numel():
                    raise ValueError("keys and values differ in length")
            n = keys.numel()
            req = (self.sorter.key_value_storage_requirements(n) if values is not None
                   else self.sorter.storage_requirements(n))
            need = max(need, req.size)
        storage = self._storage_for(need)
        storage.record_stream(current)  # (allocated on another stream, possibly: the allocator must know)
        for keys, values in arrays:
            n = keys.numel()
            if values is None:
                self.sorter.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0)
            else:
                self.sorter.cmd_sort_key_value(stream, n, keys.data_ptr(), 0, values.data_ptr(), 0,
                                               storage.data_ptr(), 0)
            items += n
        return items

    def finish(self) -> int:
        """Waits for everything enqueued -- on the stream(s) it was enqueued on, whatever the current stream is by
        now -- and returns the OR of the failure bits of all of it (0 = ok; ``describe_status`` names the bits)."""
        streams = self._streams or [self.torch.cuda.current_stream(self.device)]
        self._streams = []
        self._last = None
        for st in streams[:-1]:
            st.synchronize()
        status = self.sorter.read_sorter_status(streams[-1].cuda_stream)
        self._retired = []
        return status

    def __call__(self, arrays: Sequence[tuple]) -> int:
        self.enqueue(arrays)
        return self.finish()

    def last_plan_taken(self) -> bool:
        """Whether the two-trip plan recorded for the LAST sort enqueued (``Sorter.describe_plan``: hybrid-8 or msd) ran on
        the device, or the four passes behind it did (``vrdxHipReadPlanVerdict``; waits for that sort)."""
        if self._storage is None:
            return False
        stream = (self._last or self.torch.cuda.current_stream(self.device)).cuda_stream
        return self.sorter.plan_taken(stream, self._storage.data_ptr(), 0)

    def last_plan_verdict(self) -> int:
        """``vrdxHipReadPlanVerdict`` for the last sort enqueued (``VERDICT_*``; waits for that sort)."""
        if self._storage is None:
            return 0
        stream = (self._last or self.torch.cuda.current_stream(self.device)).cuda_stream
        return self.sorter.read_plan_verdict(stream, self._storage.data_ptr(), 0)

    def plan_counters(self):
        """(sorts recorded with the MSD plan in front, how many of them the device turned down) since the executor was
        made (``vrdxHipReadPlanCounters``; waits for the sorts in flight)."""
        stream = (self._last or self.torch.cuda.current_stream(self.device)).cuda_stream
        return self.sorter.read_plan_counters(stream)

    def close(self):
        self.sorter.destroy()


def describe_status(word: int) -> str:
    """The independent diagnoses of vrdxHipReadSorterStatus (include/vk_radix_sort.h, VRDX_HIP_STATUS_*): bit 0 = a
    look-back gave up on the DEVICE (bounded spin expired, the result of that sort is unspecified); bit 1 = the periodic
    repeat of the LDS lane-order check failed (the one-atomic ranking rests on it: call ``Sorter.recheck``); bit 30 = an element count beyond
    2^30 - 4 (where the reference's uint32 size math wraps) was clamped; bit 31 = the
    RUNTIME refused an enqueue of a sort (fill, copy or kernel launch) on the host side, i.e. that sort never ran as
    recorded."""
    if word == 0:
        return "ok"
    if word == 0xFFFFFFFF:
        return "status unreadable (the read-back itself failed)"
    parts = []
    if word & 0x80000000:
        parts.append("an enqueue was refused by the HIP runtime (bit 31)")
    if word & 0x1:
        parts.append("a device-side look-back spin expired (bit 0)")
    if word & 0x2:
        parts.append("LDS atomics were seen out of lane order by the periodic re-check (bit 1)")
    if word & 0x40000000:
        parts.append("an element count beyond 2^30 - 4 was clamped: that sort's tail is unsorted (bit 30)")
    if word & 0x3FFFFFFC:
        parts.append("unknown bits 0x%x" % (word & 0x3FFFFFFC))
    return "; ".join(parts)


class BatchedSorter:
    """Shards a batch of independent arrays over the ranks of a process group."""

    def __init__(self, executor: Optional[Callable[[Sequence[tuple]], int]] = None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world_size = dist.get_world_size(group) if self.distributed else 1
        self.executor = executor if executor is not None else HipShardExecutor()

    def my_indices(self, num_arrays: int) -> List[int]:
        return shard_indices(num_arrays, self.rank, self.world_size)

    def sort_shard(self, arrays: Sequence[tuple]) -> List[BatchRecord]:
        """`arrays` are THIS rank's arrays (already resident on its device).  Returns one record
        per rank, identical on every rank."""
        import torch
        t0 = time.perf_counter_ns()
        status = int(self.executor(arrays))
        elapsed = time.perf_counter_ns() - t0
        items = int(sum(int(k.numel()) if hasattr(k, "numel") else len(k) for k, _ in arrays))
        return self.gather(status, elapsed, items)

    def gather(self, status: int, elapsed_ns: int, items: int) -> List[BatchRecord]:
        """The only collective of the batched variant: one 24-byte record per rank, all-gathered (RCCL
        over xGMI with the nccl backend); it is also the cross-GPU completion barrier."""
        import torch
        mine = torch.tensor([status, elapsed_ns, items], dtype=torch.int64)
        if not self.distributed:
            return [BatchRecord(0, status, elapsed_ns, items)]
        if self.dist.get_backend(self.group) == "nccl":
            # RCCL wants the tensor on THIS rank's GPU: the executor's device, not whatever is current
            device = getattr(self.executor, "device", None)
            mine = mine.to(f"cuda:{device}" if device is not None else "cuda")
        gathered = [torch.empty_like(mine) for _ in range(self.world_size)]
        self.dist.all_gather(gathered, mine, group=self.group)
        return [BatchRecord(r, int(g[0]), int(g[1]), int(g[2])) for r, g in enumerate(gathered)]

    @staticmethod
    def aggregate_gitems_per_s(records: Sequence[BatchRecord]) -> float:
        worst = max(r.elapsed_ns for r in records)
        return sum(r.items for r in records) / worst if worst > 0 else 0.0
