"""GPU tests of the batched many-arrays variant (BASELINE.json configs[4]) and of the bench driver's
hip / rocprim backends, on however many GPUs the box has (one on the
driver's test box: the front ends size themselves from the device count).

  * HipShardExecutor (one VrdxSorter + stream + storage per GPU) against the oracle: 8 arrays of
    mixed sizes, keys-only and key+value, all sharing the executor's one storage buffer;
  * BatchedSorter over a real process group with the nccl (= RCCL) backend, world_size = the number of
    GPUs, launched through torch.distributed.run BEFORE anything in this process touches a GPU context
    of its own (a child process);
  * `bench/bench hip --devices G`: the same variant from one C++ host process, records exchanged with
    ncclAllGather;
  * `bench/bench hip` and `bench/bench rocprim` sweeps: the reference's protocol, its correctness check
    and its CSV (bench/bench.cc:41-112,116-207; bench/cuda_benchmark.cu:37-126 analogue).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench", "bench")


def _bench_exe():
    if not os.path.exists(BENCH):
        subprocess.run(["make", "-C", os.path.dirname(BENCH)], check=True)
    return BENCH


def test_hip_shard_executor_sorts_mixed_arrays_through_one_storage(oracle):
    import torch
    from vulkan_radix_sort_amd.batched import BatchedSorter, HipShardExecutor
    device = torch.cuda.current_device()
    ex = HipShardExecutor(device)
    sizes = [1 << 20, 5000, (1 << 22) + 12345, 70001, 1, 16385, 3_000_001, 262144]
    arrays, expected = [], []
    for i, n in enumerate(sizes):
        k, v = oracle.generate(100 + i, n, 32 if i % 3 else 12)
        dk = torch.from_numpy(k.view(np.int32).copy()).cuda(device)
        dv = torch.from_numpy(v.view(np.int32).copy()).cuda(device) if i % 2 == 0 else None
        arrays.append((dk, dv))
        expected.append(oracle.sort(k, v if dv is not None else None))
    bs = BatchedSorter(executor=ex)  # no process group: one rank owning every array
    assert bs.my_indices(len(sizes)) == list(range(len(sizes)))
    records = bs.sort_shard(arrays)
    assert len(records) == 1 and records[0].status == 0 and records[0].items == sum(sizes)
    for (dk, dv), (ek, ev, _) in zip(arrays, expected):
        assert np.array_equal(dk.cpu().numpy().view(np.uint32), ek)
        if dv is not None:
            assert np.array_equal(dv.cpu().numpy().view(np.uint32), ev)
    # wrong residency / element size are refused, not mis-sorted
    with pytest.raises(TypeError):
        ex.enqueue([(torch.zeros(8, dtype=torch.int64, device=f"cuda:{device}"), None)])
    with pytest.raises(ValueError):
        ex.enqueue([(torch.zeros(8, dtype=torch.int32), None)])
    ex.close()


def test_hip_shard_executor_orders_sorts_enqueued_on_different_streams(oracle):
    """All sorts of an executor share ONE storage buffer (histogram, status rows, tickets, scratch arrays), so two
    of them must never be in flight at once.  A caller that switches torch's current stream between two enqueue()
    calls gets a device-side dependency between the streams (wait_stream), the storage is recorded on every stream
    that uses it, and a buffer that had to grow stays alive until finish(): several large sorts alternating between
    two streams, the later ones needing a bigger storage, with no host synchronisation in between -- every result
    bit-exact, sorter status 0.  (Without the ordering these sorts overlap on the GPU and clobber each other's
    status rows and scratch arrays.)"""
    import torch
    from vulkan_radix_sort_amd.batched import HipShardExecutor
    device = torch.cuda.current_device()
    ex = HipShardExecutor(device)
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    sizes = [(1 << 22) + 3, (1 << 22) + 3, 3_000_001, (1 << 23) + 77, 1 << 22, (1 << 23) + 77]   # the fourth grows the storage
    arrays, expected = [], []
    for i, n in enumerate(sizes):
        k, v = oracle.generate(300 + i, n, 32)
        dk = torch.from_numpy(k.view(np.int32).copy()).cuda(device)
        dv = torch.from_numpy(v.view(np.int32).copy()).cuda(device) if i % 2 == 0 else None
        arrays.append((dk, dv))
        expected.append(oracle.sort(k, v if dv is not None else None))
    torch.cuda.synchronize()
    for i, pair in enumerate(arrays):
        with torch.cuda.stream(streams[i % 2]):
            ex.enqueue([pair])
    assert ex.finish() == 0
    torch.cuda.synchronize()
    for (dk, dv), (ek, ev, _) in zip(arrays, expected):
        assert np.array_equal(dk.cpu().numpy().view(np.uint32), ek)
        if dv is not None:
            assert np.array_equal(dv.cpu().numpy().view(np.uint32), ev)
    ex.close()


def test_one_gpus_share_of_the_batched_config_at_full_size(oracle, golden):
    """BASELINE.json configs[4] at G = 1: the eight independent N = 2^25 key+value arrays (seeds 1..8, the reference's
    generator) that the 8-GPU run shards one per GPU, sorted here by ONE GPU's executor through its one storage
    buffer.  Seeds 1 and 2 against the committed checksums of the reference's CPU sort, the rest by
    size-independent properties on the GPU (sorted, same multiset of keys and of values); sticky status 0."""
    import torch
    from vulkan_radix_sort_amd.batched import BatchedSorter, HipShardExecutor
    device = torch.cuda.current_device()
    n = 1 << 25
    ex = HipShardExecutor(device)
    bs = BatchedSorter(executor=ex)
    arrays, sums = [], []
    for seed in range(1, 9):
        k, v = oracle.generate(seed, n, 32)
        dk = torch.from_numpy(k.view(np.int32)).cuda(device)
        dv = torch.from_numpy(v.view(np.int32)).cuda(device)
        arrays.append((dk, dv))
        sums.append((int(k.astype(np.uint64).sum()), int(v.astype(np.uint64).sum())))
        del k, v
    records = bs.sort_shard(arrays)
    assert len(records) == 1 and records[0].status == 0 and records[0].items == 8 * n
    for seed, (dk, dv) in enumerate(arrays, start=1):
        uk = dk.to(torch.int64) & 0xFFFFFFFF
        assert bool((uk[1:] >= uk[:-1]).all()), seed
        assert int(uk.sum().item()) == sums[seed - 1][0], seed
        assert int((dv.to(torch.int64) & 0xFFFFFFFF).sum().item()) == sums[seed - 1][1], seed
        del uk
        rows = [h for h in golden["hashes"] if h["n"] == n and h["seed"] == seed and h["bits"] == 32]
        if rows:
            assert f"{oracle.hash(dk.cpu().numpy().view(np.uint32)):016x}" == rows[0]["sorted_keys_hash"], seed
            assert f"{oracle.hash(dv.cpu().numpy().view(np.uint32)):016x}" == rows[0]["sorted_values_hash"], seed
    assert sum(1 for h in golden["hashes"] if h["n"] == n and h["bits"] == 32 and h["seed"] in (1, 2)) == 2
    ex.close()


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["VRDX_ROOT"])
from oracle import load_oracle
from vulkan_radix_sort_amd.batched import BatchedSorter, HipShardExecutor
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl")
orc = load_oracle()
bs = BatchedSorter(executor=HipShardExecutor(local))
assert bs.world_size == world and bs.rank == rank
num = 8
mine = bs.my_indices(num)
arrays, want = [], []
for i in mine:
    k, v = orc.generate(i + 1, 200000 + 1237 * i, 32)
    arrays.append((torch.from_numpy(k.view(np.int32).copy()).cuda(local), torch.from_numpy(v.view(np.int32).copy()).cuda(local)))
    want.append(orc.sort(k, v))
records = bs.sort_shard(arrays)
assert [r.rank for r in records] == list(range(world)) and all(r.status == 0 for r in records)
assert sum(r.items for r in records) == sum(200000 + 1237 * i for i in range(num))
for (dk, dv), (ek, ev, _) in zip(arrays, want):
    assert np.array_equal(dk.cpu().numpy().view(np.uint32), ek) and np.array_equal(dv.cpu().numpy().view(np.uint32), ev)
if rank == 0:
    print("BATCHED_OK world=%d" % world)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_batched_sorter_over_rccl_on_every_gpu_of_the_box(tmp_path):
    import torch
    gpus = torch.cuda.device_count()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, VRDX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and f"BATCHED_OK world={gpus}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_py_reports_the_ranks_it_ran_and_refuses_a_mismatch():
    """`python bench.py --gpus G` (no launcher) starts its own G ranks; with more GPUs asked for than the
    box has it exits non-zero instead of silently measuring fewer."""
    import torch
    gpus = torch.cuda.device_count()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus + 1), "--steps", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "exposes" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                        "--log2n", "22", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["unit"] == "GItems/s" and line["value"] > 0
    assert len(line["config"]["ranks"]) == gpus and all(x["status"] == 0 for x in line["config"]["ranks"])
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


def test_bench_driver_batched_mode_in_cpp():
    """bench hip --devices G: one host process, one sorter + stream + storage per GPU, ncclAllGather of the
    per-GPU records; checked against the cpu backend (array 0 of every GPU) and by sortedness + checksum."""
    import torch
    gpus = torch.cuda.device_count()
    r = subprocess.run([_bench_exe(), "hip", "--devices", str(gpus), "--arrays", "4", "--log2n", "21"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Correctness check passed (4 arrays)" in r.stdout and f"over {gpus} GPU(s)" in r.stdout
    assert r.stdout.count("status 0") == gpus
    r = subprocess.run([_bench_exe(), "hip", "--devices", str(gpus + 1)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "exposes" in r.stderr


@pytest.mark.parametrize("backend,points,lo,hi", [("hip", 3, 18, 22), ("rocprim", 2, 18, 20)])
def test_bench_driver_backends(tmp_path, backend, points, lo, hi):
    """The reference's protocol end to end on the GPU: one-shot correctness check against the cpu backend,
    then the sweep; CSV = the reference's seven columns + achieved_GBps + hbm_fraction."""
    out = tmp_path / f"{backend}.csv"
    r = subprocess.run([_bench_exe(), backend, "--points", str(points), "--min-log2n", str(lo), "--max-log2n", str(hi),
                        "-o", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Correctness check passed" in r.stdout
    lines = [l for l in out.read_text().splitlines() if not l.startswith("#")]
    assert lines[0] == "backend,n,sort,gpu_ms,cpu_ms,gpu_gitems_s,cpu_gitems_s,achieved_GBps,hbm_fraction"
    rows = [l.split(",") for l in lines[1:]]
    assert len(rows) == 2 * points and [r_[2] for r_ in rows] == ["keys", "kv"] * points
    assert all(r_[0] == backend and float(r_[3]) > 0 and 0 < float(r_[8]) < 1 for r_ in rows)
    assert int(rows[0][1]) == 1 << lo and int(rows[-1][1]) == 1 << hi
    if backend == "hip":
        assert out.read_text().startswith("# version: vrdx-hip")


def test_bench_driver_graph_replay_mode(tmp_path):
    """bench hip --graph: every sort is captured once per (N, mode) into a hipGraph and the timed runs replay it (the
    reference's record-once / submit-many model, bench/vulkan_benchmark.cc:292-302).  The one-shot correctness check
    runs THROUGH the replayed graphs (keys-only, then key+value indirect: the device-side count is read on replay);
    same CSV; the version line says which mode produced it."""
    out = tmp_path / "graph.csv"
    r = subprocess.run([_bench_exe(), "hip", "--graph", "--points", "3", "--min-log2n", "18", "--max-log2n", "22", "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "Correctness check passed" in r.stdout
    text = out.read_text()
    assert text.startswith("# version: vrdx-hip") and "[hipGraph replay]" in text.splitlines()[0]
    rows = [l.split(",") for l in text.splitlines()[2:]]
    assert len(rows) == 6 and all(float(r_[3]) > 0 and float(r_[4]) > 0 for r_ in rows)
