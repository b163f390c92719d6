#!/usr/bin/env python3
"""bench.py -- headline benchmark of the vrdxCmdSort* hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 25]

A "step" is one complete sort (whatever plan the library records for that size: at 2^25 the MSD plan --
histogram, spine, scatter by a ten-bit window (for uniform keys the top ten bits), one in-LDS sort per bucket --
with the four onesweep passes behind it as the device-side fallback; which of the two RAN in the timed region is
read back from the device and stated, `roofline.plan_proof`) of one batch of synthetic input:
N = 2^25 uniform-random u32 keys (BASELINE.json configs[1]); the key+value figure (configs[2]) is
measured the same way and reported in the same JSON line under "key_value".  Inputs are resident
in HBM before the timed region starts: every step sorts its own pre-generated array in place, so
the timed region holds exactly K sorts and nothing else (the reference likewise excludes upload
and read-back: bench/vulkan_benchmark.cc:267-290,306-316, and uses fresh data per run:
bench/bench.cc:83-84).

Behind the headline region (rank 0 of a 1-GPU run only) the line also carries the reference's size curve
(bench/bench.cc:17-20,161-203: 1 warm-up + 10 timed runs on fresh data, median GPU time) at N = 2^18 ... 2^25
for keys-only and key+value under "sweep", one HBM-resident point at N = 2^27 (far past the 256 MiB
Infinity Cache) under "hbm_resident", and BASELINE.json configs[3] -- the adversarial inputs at the headline size:
all-equal, all-0xFFFFFFFF, descending, ascending, few-distinct(4), 24-bit keys; GItems/s, slowdown against uniform
keys, the device's verdict, parity with values = iota -- under "adversarial".  The key+value half of the metric is
`value_key_value` at the top level, next to `value`.

Host memory is O(1) in --steps: the input streams are generated a few at a time on host threads, uploaded
and freed; the pristine copies live on the device.

With --gpus N > 1 this is the batched many-arrays variant (BASELINE.json configs[4]): one rank per
GPU, every rank sorts its own independent arrays through vulkan_radix_sort_amd.batched
(HipShardExecutor: one VrdxSorter, one stream, one storage buffer per GPU), there is no collective
on the data path -- only the 24-byte end-of-batch record all-gather over RCCL -- and value = items
sorted by all ranks / max-over-ranks time ("scaling": "weak").  Either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment) or plainly as `python bench.py --gpus N`, which then starts that launcher itself as
a child process BEFORE anything touches a GPU.  The JSON's n_gpus is the number of ranks RCCL
actually connected; a mismatch with --gpus is an error (exit code 2), never a silent N = 1.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290
KEYS_BYTES_PER_ITEM = 36.0   # 4 (fused histogram read) + 4 passes x (4 read + 4 write)
KV_BYTES_PER_ITEM = 68.0     # 4 + 4 x (8 + 8)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)    # bench/bench.cc:16 kTimedRuns
    p.add_argument("--warmup", type=int, default=1)    # bench/bench.cc:15 kWarmupRuns
    p.add_argument("--log2n", type=int, default=25)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-sweep", action="store_true")   # only the headline region and the roofline
    p.add_argument("--no-adversarial", action="store_true")   # (with --no-sweep: every launch a profiler sees is a headline sort)
    return p.parse_args()


def reference_stream(seed, n):
    """keys = the first n raw outputs of std::mt19937(seed), values = the next n
    (bench/data_generator.cc:20-25 under libstdc++).  numpy's legacy RandomState seeds MT19937 with init_genrand(seed)
    like std::mt19937 does and hands out the raw 32-bit words for the full range (checked against the reference's
    generator by tests/test_oracle.py::test_bench_input_stream_is_the_reference_generator)."""
    import numpy as np
    # (dtype uint32 with the full range hands out the raw words directly: the same stream as the uint64 detour of
    # earlier rounds, twenty times faster -- 0.36 s instead of 7 s per 2^26 words)
    raw = np.random.RandomState(seed).randint(0, 1 << 32, size=2 * n, dtype=np.uint32)
    return raw[:n], raw[n:]


def to_device(torch, a, device):
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(device)


def random_u32(torch, n, seed, device):
    """Keys only: the reference stream of that seed, resident on the device."""
    return to_device(torch, reference_stream(seed, n)[0], device)


GENERATOR_THREADS = 2   # streams in flight on the host at a time: 2 x 256 MiB at N = 2^25 (peak RSS 2.0 GiB, of which the runtime is most)


def upload_streams(torch, n, seeds, device):
    """Device-resident pristine copies of the reference streams of `seeds`: (keys tensors, values tensors).
    Generated GENERATOR_THREADS at a time on host threads (numpy's generator releases the GIL), uploaded and freed
    at once, so the host holds at most that many streams whatever --steps is (the round-4 line kept every step's
    input on the host: 11 GiB and half a minute per rank at the driver's --steps 20 --warmup 5)."""
    from concurrent.futures import ThreadPoolExecutor
    keys, values = [], []
    with ThreadPoolExecutor(GENERATOR_THREADS) as pool:
        for lo in range(0, len(seeds), GENERATOR_THREADS):
            for k, v in pool.map(lambda seed: reference_stream(seed, n), seeds[lo:lo + GENERATOR_THREADS]):
                keys.append(to_device(torch, k, device))
                values.append(to_device(torch, v, device))
                del k, v
    return keys, values


def timed_sorts(torch, dist, executor, pristine, n, steps, warmup, key_value, device, distributed):
    """Returns (wall_seconds_for_K_steps_max_over_ranks, per-step gpu ms list of this rank).
    Every sort goes through the batched front end's per-GPU executor (one array per step and rank).
    pristine = (keys, values) device tensors, one pair per step: the keys-only measurement sorts device-side copies,
    the key+value measurement (which runs last) the pristine arrays themselves."""
    total = warmup + 2 * steps
    if key_value:
        keys, values = pristine[0][:total], pristine[1][:total]
    else:
        keys, values = [k.clone() for k in pristine[0][:total]], None

    def one(i):
        executor.enqueue([(keys[i], values[i] if key_value else None)])  # vrdxCmdSort[KeyValue]: never blocks

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    counters_before = executor.plan_counters()   # (outside the timed region)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):      # the timed region: exactly K sorts, nothing else on the stream
        one(warmup + s)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # What the DEVICE made of the plan the host recorded, for every sort of the timed region: the sorter counts the MSD plans
    # it records and the device counts those it turns down (vrdxHipReadPlanCounters); the last sort's verdict word as well.
    counters_after = executor.plan_counters()
    proof = {"plan_taken_by_last_sort": executor.last_plan_taken(),
             "msd_plans_recorded_in_timed_region": counters_after[0] - counters_before[0],
             "msd_plans_declined_in_timed_region": counters_after[1] - counters_before[1]}
    # per-sort GPU time (median reported next to the headline): K more sorts on fresh data, outside
    # the timed region, each between two events on the sort's stream
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    for s in range(steps):
        starts[s].record()
        one(warmup + steps + s)
        ends[s].record()
    torch.cuda.synchronize()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_step_ms = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    status = executor.finish()  # the sorter's sticky word: every sort above, not only the last one
    if status != 0:
        from vulkan_radix_sort_amd.batched import describe_status
        raise RuntimeError(f"sorter status 0x{status:08x}: {describe_status(status)}")
    # the last step's output must be sorted (cheap sanity check outside the timed region)
    k = keys[-1].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    if not bool((k[1:] >= k[:-1]).all()):
        raise RuntimeError("output not sorted")
    del keys, values
    torch.cuda.empty_cache()
    return elapsed, per_step_ms, proof


def stage_profile(torch, sorter, pristine, n, key_value, device, repeats=5):
    """Per-kernel event intervals from the 15-slot timestamp contract (HIP events on the sort's own stream) AS
    STAMPED, i.e. each including one event record (main() subtracts the calibrated cost of that,
    vrdxHipEventOverheadNs, to get kernel time).  Returns a dict of mean intervals in ms:
      the MSD plan      {"histogram", "spine", "scatter", "bucket", "fallback"}  (slots 1-2, 2-3, 3-4, 4-5, 5-14: the
                        launches of the fallback that are not also a launch of the plan; they return on its verdict)
      the four passes   {"histogram", "sweep"}  (slots 1-2 and the mean of the four downsweep intervals)
    The stamped sort runs right BEHIND another sort of a different array, like every sort of the timed region does: a
    kernel is charged for the write-back of what the kernel before it has just written, and with
    the stream busy the host's enqueue latency stays out of the stamps.  pristine: device-resident (keys, values)
    lists of at least 2 * (repeats + 1) arrays, copied on the device for every run."""
    import vulkan_radix_sort_amd as vrdx
    stream = torch.cuda.current_stream().cuda_stream
    req = sorter.key_value_storage_requirements(n) if key_value else sorter.storage_requirements(n)
    storage = torch.empty(req.size, dtype=torch.uint8, device=device)
    pool = vrdx.QueryPool(15)
    msd = sorter.describe_plan(n, key_value).name == "msd"
    acc = {}

    def record(keys, values, query_pool):
        if key_value:
            sorter.cmd_sort_key_value(stream, n, keys.data_ptr(), 0, values.data_ptr(), 0, storage.data_ptr(), 0,
                                      query_pool, 0)
        else:
            sorter.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0, query_pool, 0)

    for r in range(repeats + 1):
        arrays = []
        for i in (2 * r, 2 * r + 1):
            i %= len(pristine[0])
            arrays.append((pristine[0][i][:n].clone(), pristine[1][i][:n].clone() if key_value else None))
        torch.cuda.synchronize()
        record(arrays[0][0], arrays[0][1], None)   # the sort in front
        record(arrays[1][0], arrays[1][1], pool)   # the stamped one
        torch.cuda.synchronize()
        ts = pool.results_ns()
        if r == 0:
            continue
        if msd:
            parts = {"histogram": ts[2] - ts[1], "spine": ts[3] - ts[2], "scatter": ts[4] - ts[3],
                     "bucket": ts[5] - ts[4], "fallback": ts[14] - ts[5]}
        else:
            parts = {"histogram": ts[2] - ts[1],
                     "sweep": sum(ts[4 + 3 * p] - ts[3 + 3 * p] for p in range(4)) / 4.0}
        for name, ns in parts.items():
            acc.setdefault(name, []).append(ns / 1e6)
    pool.destroy()
    return {name: sum(v) / len(v) for name, v in acc.items()}


def size_curve(torch, executor, sorter, pristine, device, log2_sizes, runs=11):
    """The reference's size curve (bench/bench.cc:15-20,66-112: 1 warm-up + 10 timed runs on fresh data per size, median)
    for keys-only and key+value: GPU time between two events on the sort's stream around the sort, one sort at a
    time, upload excluded (the data are device-side copies of prefixes of the pristine mt19937 streams: the first n
    outputs of std::mt19937(seed) are exactly DataGenerator(seed)'s keys).  hbm_fraction = the HBM bytes of the plan
    the library records for that size (vrdxHipDescribePlan) over the median time, against the 8 TB/s peak."""
    points = []
    for lg in log2_sizes:
        n = 1 << lg
        point = {"log2n": lg, "n": n}
        for kv in (False, True):
            ms = []
            for r in range(runs):
                src = r % len(pristine[0])
                k = pristine[0][src][:n].clone()
                v = pristine[1][src][:n].clone() if kv else None
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                start.record()
                executor.enqueue([(k, v)])
                end.record()
                torch.cuda.synchronize()
                if r > 0:
                    ms.append(start.elapsed_time(end))
            med = sorted(ms)[len(ms) // 2]
            plan = sorter.describe_plan(n, kv)
            point["key_value" if kv else "keys"] = priced_point(plan, executor.last_plan_taken(), n, med)
        points.append(point)
    status = executor.finish()
    if status != 0:
        raise RuntimeError(f"sorter status 0x{status:08x} in the size curve")
    return points


def hbm_resident_point(torch, executor, sorter, pristine, device, lg=27, runs=6):
    """One point far past the 256 MiB Infinity Cache: N = 2^27 (keys-only 36 B x 2^27 = 4.8 GB of traffic with the four
    passes), keys = the concatenation of four pristine mt19937 streams.  Same protocol as the curve (1 + 5 runs)."""
    n = 1 << lg
    parts = n // pristine[0][0].numel()
    if parts < 1 or parts > len(pristine[0]):
        return None
    base_k = torch.cat(pristine[0][:parts])
    base_v = torch.cat(pristine[1][:parts])
    out = {"log2n": lg, "n": n, "data": f"concatenation of {parts} std::mt19937 streams"}
    for kv in (False, True):
        ms = []
        for r in range(runs):
            k = base_k.clone()
            v = base_v.clone() if kv else None
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            start.record()
            executor.enqueue([(k, v)])
            end.record()
            torch.cuda.synchronize()
            if r > 0:
                ms.append(start.elapsed_time(end))
            del k, v
        med = sorted(ms)[len(ms) // 2]
        plan = sorter.describe_plan(n, kv)
        out["key_value" if kv else "keys"] = priced_point(plan, executor.last_plan_taken(), n, med)
    status = executor.finish()
    if status != 0:
        raise RuntimeError(f"sorter status 0x{status:08x} at N = 2^{lg}")
    return out


def bytes_moved_per_item(plan, taken):
    """HBM bytes per element of the sort that RAN: the recorded plan's when the device took it (or when the plan is not
    one the device decides about), the four passes' otherwise."""
    decided_on_device = plan.name in ("msd", "hybrid-8")
    return int(plan.bytesPerElement if (taken or not decided_on_device) else plan.fallbackBytesPerElement)


def priced_point(plan, taken, n, ms):
    per_item = bytes_moved_per_item(plan, taken)
    return {"gpu_ms": ms, "gitems_per_s": n / (ms * 1e-3) / 1e9, "plan": plan.name,
            "plan_taken": bool(taken) if plan.name in ("msd", "hybrid-8") else None,
            "bytes_per_item": per_item, "hbm_fraction": per_item * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}


ADVERSARIAL_PATTERNS = ("uniform", "all-equal", "all-0xFFFFFFFF", "descending", "ascending", "few-distinct(4)", "24-bit",
                        "bell-shaped top byte")


def adversarial_block(torch, executor, sorter, pristine, n, device, runs=6):
    """BASELINE.json configs[3] / BASELINE.md section 3: N = 2^25 adversarial keys (all-equal, all-0xFFFFFFFF -- the
    padding sentinel --, descending N-1-i, ascending, few-distinct: four values; 24-bit keys like the reference's
    DataGenerator::Generate(n, 24), bench/data_generator.cc:15; and mildly skewed keys, a bell-shaped top byte), keys-only
    and key+value with values = iota.  Protocol of
    the size curve (bench/bench.cc:66-112: 1 warm-up + 5 timed runs, median GPU time between two events around the sort,
    input restored on the device before every run).  Per pattern and mode: GItems/s, slowdown against uniform keys in
    the same protocol, which sort ran (the device's verdict), and a proof of the permutation computed on the device:
    keys ascending; key+value: keys_out == keys_in[values_out] and, inside every run of equal keys, values_out strictly
    increasing -- together: THE stable sort (the oracle's predicate, bench/bench.cc:41-64, without the host)."""
    import vulkan_radix_sort_amd as vrdx
    iota = torch.arange(n, dtype=torch.int32, device=device)
    u = pristine[0][0][:n]

    def u32(x):
        return x if x < (1 << 31) else x - (1 << 32)

    def make(pattern):
        if pattern == "uniform":
            return u.clone()
        if pattern == "all-equal":
            return torch.full((n,), u32(0x12345678), dtype=torch.int32, device=device)
        if pattern == "all-0xFFFFFFFF":
            return torch.full((n,), -1, dtype=torch.int32, device=device)
        if pattern == "descending":
            return torch.arange(n - 1, -1, -1, dtype=torch.int32, device=device)
        if pattern == "ascending":
            return iota.clone()
        if pattern == "few-distinct(4)":
            four = torch.tensor([3, -1, 0x00010000, 0x7F000000], dtype=torch.int32, device=device)
            return four[(u & 3).to(torch.int64)]
        if pattern == "24-bit":
            return ((u.to(torch.int64) & 0xFFFFFFFF) >> 8).to(torch.int32)
        if pattern == "bell-shaped top byte":
            # mildly skewed keys (the mean of four uniform bytes in the top byte: the fullest ten-bit bucket holds 1.5 x the
            # mean, more than the plan's buckets leave room for): what the device turns down by the COUNT, not by the sample
            r = u.to(torch.int64) & 0xFFFFFFFF
            top = ((r & 255) + ((r >> 8) & 255) + ((r >> 16) & 255) + (r >> 24)) >> 2
            low = pristine[0][1][:n].to(torch.int64) & 0x00FFFFFF
            k = (top << 24) | low
            return torch.where(k >= (1 << 31), k - (1 << 32), k).to(torch.int32)
        raise ValueError(pattern)

    def unsigned(t):
        return t.to(torch.int64) & 0xFFFFFFFF

    out, base = [], {}
    for pattern in ADVERSARIAL_PATTERNS:
        src = make(pattern)
        entry = {"keys_pattern": pattern}
        for kv in (False, True):
            ms = []
            for r in range(runs):
                k = src.clone()
                v = iota.clone() if kv else None
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                start.record()
                executor.enqueue([(k, v)])
                end.record()
                torch.cuda.synchronize()
                if r > 0:
                    ms.append(start.elapsed_time(end))
            med = sorted(ms)[len(ms) // 2]
            plan = sorter.describe_plan(n, kv)
            taken = executor.last_plan_taken()
            identical = executor.last_plan_verdict() == vrdx.VERDICT_MSD_SORTED   # all keys identical: only the histogram read them
            ku = unsigned(k)
            ok = bool((ku[1:] >= ku[:-1]).all())
            if kv:
                vi = v.to(torch.int64)
                ok = ok and bool((unsigned(src)[vi] == ku).all())
                ok = ok and bool(((ku[1:] > ku[:-1]) | (vi[1:] > vi[:-1])).all())
                del vi
            del ku
            base.setdefault(kv, med)
            entry["key_value" if kv else "keys"] = {
                "gpu_ms": med, "gitems_per_s": n / (med * 1e-3) / 1e9, "slowdown_vs_uniform": med / base[kv],
                "plan_taken": bool(taken),
                "ran": (plan.name + ": all keys identical, nothing moved") if identical else plan.name if taken else "four-passes",
                "bytes_per_item": 4 if identical else bytes_moved_per_item(plan, taken), "parity": "ok" if ok else "MISMATCH"}
            if not ok:
                raise RuntimeError(f"adversarial input {pattern} ({'key+value' if kv else 'keys'}): wrong result")
            del k, v
        out.append(entry)
        del src
    status = executor.finish()
    if status != 0:
        raise RuntimeError(f"sorter status 0x{status:08x} in the adversarial block")
    return out


def cpu_baseline(n):
    """The reference's CPU path (bench/cpu_benchmark.cc: std::sort / std::stable_sort, 1 thread)
    timed on this box's host cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import load_oracle, load_reference
    orc = load_oracle()
    ref = load_reference()
    keys, values = orc.generate(1, n, 32)
    if ref is not None:
        kind = "reference"
        _, ns_keys = ref.sort_keys(keys)
        _, _, ns_kv = ref.sort_key_value(keys, values)
    else:
        kind = "port"
        _, ns_keys = orc.port_sort_keys(keys)
        _, _, ns_kv = orc.port_sort_key_value(keys, values)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": n / ns_keys, "unit": "GItems/s", "cores": 1, "kind": kind,
        "sample": f"1 run of N=2^{n.bit_length() - 1} mt19937 u32: keys-only std::sort {ns_keys / 1e6:.0f} ms, "
                  f"key+value std::stable_sort(index) {ns_kv / 1e6:.0f} ms (timed like bench/cpu_benchmark.cc:22-25,38-41)",
        "key_value_value": n / ns_kv, "host_cpu": model, "host_threads_available": os.cpu_count(),
    }


def kernel_name(version, which):
    """Dominant kernel of the sort at the bench size, from the library's own description of its tile
    choice ("... keys=1024x32x2 key-value=1024x32 ..."): AxB -> onesweep_kernel<A, B, ...>,
    AxBx2 -> onesweep_pair_kernel<A, B, ...> (two sub-tiles per workgroup)."""
    import re
    m = re.search(which + r"=(\d+)x(\d+)(x2)?", version)
    if not m:
        return "onesweep_kernel"
    if m.group(3):
        # keys-only, one-atomic ranking; last argument: even-split tiles (sorts of one round only, not the bench size)
        return "onesweep_pair_kernel<%s, %s, false>" % (m.group(1), m.group(2))
    kv = "true" if which == "key-value" else "false"
    return "onesweep_kernel<%s, %s, %s, true, false>" % (m.group(1), m.group(2), kv)


def latest_pmc_traffic(version):
    """HBM bytes per onesweep launch from the rocprofv3 PMC passes committed under profiles/ (collected
    offline by tools/profile_round.sh: counters cannot be read from inside this process).  Only a file
    stamped with THIS library (same kernel source digest and tile choice) is believed; otherwise the
    bench line says traffic = null rather than quote bytes measured on other kernels."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None
    if pmc.get("kernel_source_sha256") != kernel_source_digest() or pmc.get("library") != version:
        return None
    return pmc


def committed_rocprof_averages(version):
    """Per-kernel average durations (us) of the rocprofv3 --kernel-trace --stats run of THIS command committed under
    profiles/ (profiles/kernel_stats.json, written by tools/kernel_stats.py --json from the newest
    rNN_rocprofv3_kernel_stats.csv), so that the event-timed figures of this line can be compared with the profiler's
    from the JSON alone.  "matches_this_build" says whether that file was taken with these kernel sources and this
    tile choice; a stale file is still quoted, flagged.  None when no such file is there."""
    path = os.path.join(ROOT, "profiles", "kernel_stats.json")
    try:
        with open(path) as f:
            stats = json.load(f)
    except (OSError, ValueError):
        return None
    stats["matches_this_build"] = (stats.get("kernel_source_sha256") == kernel_source_digest()
                                   and stats.get("library") == version)
    return stats


def committed_ceiling():
    """What fraction of the HBM peak the pass-shaped kernels can reach in this formulation, from the ablation run
    committed under profiles/ (tools/r05/ceiling.sh -> profiles/r05_ceiling.json), with the file it comes from."""
    path = os.path.join(ROOT, "profiles", "r05_ceiling.json")
    try:
        with open(path) as f:
            c = json.load(f)
    except (OSError, ValueError):
        return None
    c["source"] = "profiles/r05_ceiling.json"
    return c


def kernel_source_digest():
    import hashlib
    h = hashlib.sha256()
    for name in ("vrdx_kernels.hip", "vrdx_kernels.h", "vrdx_layout.h", "vrdx_api.cpp"):
        with open(os.path.join(ROOT, "vulkan_radix_sort_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching a GPU runtime (torch.cuda.device_count() may initialise HIP in
    the parent of the ranks): the KFD topology in sysfs (nodes with SIMDs are GPUs), narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  None when sysfs says nothing; the ranks then check
    themselves (a world size that differs from --gpus is exit code 2, see main())."""
    import glob
    count = 0
    if not os.path.isdir("/sys/class/kfd"):
        return 0  # no amdgpu compute driver on this machine at all
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count") and int(line.split()[1]) > 0:
                        count += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        listed = os.environ.get(var)
        if listed is not None:
            count = min(count, len([x for x in listed.split(",") if x.strip() != ""]))
    return count


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (torch.distributed.run) as a CHILD
    process -- this process imports neither torch nor the HIP library (the GPUs are counted in sysfs), so it
    never touches a GPU -- and exit with the child's code.  (Never exec from a process that holds a GPU context.)"""
    import socket
    import subprocess
    have = visible_gpu_count()
    if have is not None and have < args.gpus:
        print(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    t_process = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    import vulkan_radix_sort_amd as vrdx
    from vulkan_radix_sort_amd.batched import BatchedSorter, HipShardExecutor

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():   # (a rank process: querying the GPU runtime is fine here)
        print(f"[bench] --gpus {args.gpus} but this node exposes {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    n_gpus = dist.get_world_size() if distributed else 1   # the ranks RCCL actually connected
    if args.gpus != n_gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but {n_gpus} rank(s) are running: refusing to report", file=sys.stderr)
        sys.exit(2)

    import resource
    torch.zeros(1, device=device)   # the runtime is up: what the process weighs before it holds a single input
    rss_runtime_mib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
    n = 1 << args.log2n
    executor = HipShardExecutor(local_rank)   # one VrdxSorter + stream + storage for this GPU
    sorter = executor.sorter
    batch = BatchedSorter(executor=executor)

    # Inputs: seeds 1, 2, ... like the reference's runs (bench/bench.cc:83-84), other ranks continue the sequence;
    # warm-up | the K timed steps | K more, each bracketed by events -- and, on the rank that reports the curve,
    # CURVE_STREAMS more that are only ever copied from (size curve, stage profile, the 2^27 point).
    total = args.warmup + 2 * args.steps
    extras = rank == 0 and n_gpus == 1
    CURVE_STREAMS = 11
    seed0 = 1 + (total + CURVE_STREAMS) * rank
    t_setup = time.perf_counter()
    pristine = upload_streams(torch, n, list(range(seed0, seed0 + total + (CURVE_STREAMS if extras else 2))), device)
    setup_s = time.perf_counter() - t_setup
    to_first_region_s = time.perf_counter() - t_process   # start of main() -> inputs resident, first barrier next
    rss_inputs_mib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
    fresh = (pristine[0][total:], pristine[1][total:])   # never sorted in place
    wall_keys, steps_keys, proof_keys = timed_sorts(torch, dist, executor, pristine, n, args.steps, args.warmup, False, device, distributed)
    wall_kv, steps_kv, proof_kv = timed_sorts(torch, dist, executor, pristine, n, args.steps, args.warmup, True, device, distributed)
    stamped_keys = stage_profile(torch, sorter, fresh, n, False, device, repeats=5 if extras else 1)
    stamped_kv = stage_profile(torch, sorter, fresh, n, True, device, repeats=5 if extras else 1)
    # ONE definition of a launch duration on this line: the event interval around the kernel minus what a pair of event
    # records adds to a kernel of known duration on this stream (vrdxHipEventOverheadNs: a kernel that times itself
    # with the device's wall clock, bracketed the same way, median of eight) = kernel time, which is what rocprofv3
    # reports for the same launches (roofline.rocprof).  The raw intervals stay beside it as stamped_launch_ms.
    try:
        overhead_ms = vrdx.event_overhead_ns(torch.cuda.current_stream().cuda_stream) / 1e6
        source = ("HIP events on the sort's stream (15-slot timestamp contract) minus the calibrated event overhead "
                  "(vrdxHipEventOverheadNs)")
    except vrdx.VrdxError:   # the calibration kernel could not run: report the raw intervals and say so
        overhead_ms = 0.0
        source = "HIP events on the sort's stream (15-slot timestamp contract), UNCALIBRATED: includes one event record"
    # end-of-batch record of every rank (the batched variant's only collective; 24 bytes per rank)
    records = batch.gather(0, int(wall_kv * 1e9), total * n)

    def median(xs):
        s = sorted(xs)
        return s[len(s) // 2]

    value_keys = n_gpus * args.steps * n / wall_keys / 1e9
    value_kv = n_gpus * args.steps * n / wall_kv / 1e9
    med_keys_ms, med_kv_ms = median(steps_keys), median(steps_kv)
    version = vrdx.version_string()
    pmc = latest_pmc_traffic(version)
    box = os.uname().nodename

    def kernels_of(stamped, key_value):
        """Per-kernel entries of one mode: duration (stamped and calibrated), algorithmic bytes, achieved rate."""
        plan = sorter.describe_plan(n, key_value)
        item = 8.0 if key_value else 4.0   # bytes of one element (key [+ value]) in one direction
        if plan.name == "msd":
            names = {"histogram": "histogram_msd_kernel<32u, %du>" % plan.bits,
                     "spine": "spine_msd_kernel<%du>" % plan.bits,
                     # the plan's scatter / bucket launches are also pass 0 / pass 1 of the fallback (one kernel, two roles
                     # chosen on the device); the other passes remain as launches that return on the verdict.  How many
                     # launches have a second role (0 | 1 | 2) is in the plan's launch count: histogram + spine + scatter +
                     # buckets + the passes that are launches of their own
                     "fallback": None}
            fused = 8 - int(plan.launches)
            kv = "true" if key_value else "false"
            names["scatter"] = ("msd_scatter_or_pass0_kernel<%du, %s, false>" % (plan.bits, kv)) if fused >= 1 \
                else ("scatter_msd_kernel<%du, %s>" % (plan.bits, kv))
            names["bucket"] = ("msd_buckets_or_pass1_kernel<%du, %s, false>" % (plan.bits, kv)) if fused >= 2 \
                else ("bucket_sort2_kernel<%du, 36, %s>" % (plan.bits, kv))
            names["fallback"] = ("%d x " % (4 - fused)) + kernel_name(version, "key-value" if key_value else "keys") \
                + " (returning on the verdict)"
            bytes_of = {"histogram": 4.0 * n, "spine": 0.0, "scatter": 2 * item * n, "bucket": 2 * item * n, "fallback": 0.0}
            what = {"histogram": "HBM read", "spine": "launch latency (4 MiB of 16-bit counts)",
                    "scatter": "HBM read + write in runs of 128 bytes", "bucket": "LDS (two in-LDS passes per key between one read and one write)",
                    "fallback": "launch latency"}
        else:
            names = {"histogram": "histogram_kernel<32u, false>", "sweep": kernel_name(version, "key-value" if key_value else "keys")}
            bytes_of = {"histogram": 4.0 * n, "sweep": 2 * item * n}
            what = {"histogram": "HBM read", "sweep": "HBM read + write (the tile's latency chain)"}
        out = {}
        for part, ms_stamped in stamped.items():
            ms = max(ms_stamped - overhead_ms, 1e-6)
            entry = {"kernel": names[part], "avg_launch_ms": ms, "stamped_launch_ms": ms_stamped,
                     "algorithmic_bytes_per_launch": bytes_of[part], "limited_by": what[part]}
            if bytes_of[part] > 0:
                entry["achieved"] = bytes_of[part] / (ms * 1e-3) / 1e9
                entry["frac"] = entry["achieved"] / HBM_PEAK_GBPS
            traffic = ((pmc or {}).get("plan_kernels") or {}).get(("key_value:" if key_value else "keys:") + part)
            if traffic:
                entry["traffic"] = traffic
                entry["traffic_GBps"] = traffic / (ms * 1e-3) / 1e9
                entry["traffic_frac"] = entry["traffic_GBps"] / HBM_PEAK_GBPS
            out[part] = entry
        return plan, out

    plan_keys, kernels_keys = kernels_of(stamped_keys, False)
    plan_kv, kernels_kv = kernels_of(stamped_kv, True)
    # The whole sort is priced with the bytes of the plan that RAN in the timed region -- proven, not assumed: uniform keys
    # must have taken the plan the host recorded in every one of its sorts (the device counts the plans it turns down).
    # (Should the device ever turn the plan down on these keys -- it cannot at 2^25: the capacity is 22 sigma above the mean
    # bucket -- the line is still printed, priced with the four passes' bytes, and says so: `plan_proof`.)
    def ran_as_recorded(plan, proof):
        return plan.name not in ("msd", "hybrid-8") or (proof["plan_taken_by_last_sort"] and
                                                        (plan.name != "msd" or proof["msd_plans_declined_in_timed_region"] == 0))
    for mode, plan, proof in (("keys-only", plan_keys, proof_keys), ("key+value", plan_kv, proof_kv)):
        proof["ran_as_recorded"] = ran_as_recorded(plan, proof)
        if not proof["ran_as_recorded"]:
            print(f"[bench] {mode}: the device turned the recorded {plan.name} plan down in the timed region: {proof}", file=sys.stderr)
    bytes_keys = bytes_moved_per_item(plan_keys, proof_keys["ran_as_recorded"])
    bytes_kv = bytes_moved_per_item(plan_kv, proof_kv["ran_as_recorded"])

    def dominant(kernels):
        movers = {k: v for k, v in kernels.items() if v["algorithmic_bytes_per_launch"] > 0 and k != "histogram"}
        return max(movers.items(), key=lambda kv: kv[1]["avg_launch_ms"])

    dom_part, dom = dominant(kernels_keys)
    dom_part_kv, dom_kv = dominant(kernels_kv)
    roofline = {
        "bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBPS,
        "unit": "GB/s", "frac": dom["frac"], "traffic": dom.get("traffic"),
        "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
        "stamped_launch_ms": dom["stamped_launch_ms"], "event_overhead_ms": overhead_ms,
        "launch_time_source": source, "measured_on": box,
        "note": ("the dominant kernel is the one with the longest launch among those that move the data; 'bound' "
                 "prices it against HBM as the contract asks, 'limited_by' says what it actually waits for"),
        "limited_by": dom["limited_by"], "plan": plan_keys.name, "plan_bits": int(plan_keys.bits),
        "plan_proof": proof_keys,
        "kernels": kernels_keys,
        "whole_sort": {"algorithmic_bytes": float(bytes_keys) * n,
                       "bytes_per_item": bytes_keys,
                       "achieved": bytes_keys * n / (med_keys_ms * 1e-3) / 1e9,
                       "frac": bytes_keys * n / (med_keys_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       "four_pass_equivalent_frac": KEYS_BYTES_PER_ITEM * n / (med_keys_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        "key_value": {"kernel": dom_kv["kernel"], "avg_launch_ms": dom_kv["avg_launch_ms"],
                      "stamped_launch_ms": dom_kv["stamped_launch_ms"], "achieved": dom_kv["achieved"],
                      "frac": dom_kv["frac"], "traffic": dom_kv.get("traffic"), "limited_by": dom_kv["limited_by"],
                      "plan": plan_kv.name, "plan_bits": int(plan_kv.bits), "plan_proof": proof_kv, "kernels": kernels_kv,
                      "whole_sort_bytes_per_item": bytes_kv,
                      "whole_sort_achieved": bytes_kv * n / (med_kv_ms * 1e-3) / 1e9,
                      "whole_sort_frac": bytes_kv * n / (med_kv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "four_pass_equivalent_frac": KV_BYTES_PER_ITEM * n / (med_kv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
    }
    if "traffic" in dom:
        roofline["traffic_GBps"], roofline["traffic_frac"] = dom["traffic_GBps"], dom["traffic_frac"]
    ceiling = committed_ceiling()
    if ceiling:
        roofline["ceiling_frac"] = ceiling
    rocprof = committed_rocprof_averages(version)
    if rocprof:
        roofline["rocprof"] = rocprof   # (carries the box it was taken on: "measured_on")

    result = {
        "metric": "GItems/s at N=2^25 (keys & key+value); achieved HBM GB/s vs peak",
        "value": value_keys, "unit": "GItems/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_keys / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32",
        "data": "synthetic: raw std::mt19937(seed) outputs, seeds 1.. (keys = first N, values = next N; SURVEY 8d)",
        "config": {"workload": f"N=2^{args.log2n} uniform-random u32 keys-only, 1xMI355X per rank "
                               f"(BASELINE.json configs[1]); key+value (configs[2]) under key_value",
                   "n": n, "arrays_per_step_per_gpu": 1,
                   "parallelism": "independent arrays, one per GPU, no data-path collective" if distributed else "single GPU",
                   "tile": version, "plan": plan_keys.name, "ranks": [{"rank": r.rank, "status": r.status} for r in records]},
        "median_gpu_ms_per_sort": med_keys_ms, "median_gitems_per_s": n / (med_keys_ms * 1e-3) / 1e9,
        # the key+value half of the metric (configs[2]), top level so that it is read with the keys-only `value`
        "value_key_value": value_kv, "ms_per_step_key_value": wall_kv / args.steps * 1e3,
        "key_value": {"value": value_kv, "unit": "GItems/s", "ms_per_step": wall_kv / args.steps * 1e3,
                      "median_gpu_ms_per_sort": med_kv_ms, "median_gitems_per_s": n / (med_kv_ms * 1e-3) / 1e9},
        "targets": {"cub_onesweep_rtx5080_keys": 22.36, "cub_onesweep_rtx5080_key_value": 11.67,
                    "note": "north-star floor from the reference README (other hardware), not a vs_baseline"},
        "roofline": roofline,
        "setup": {"input_generation_and_upload_s": setup_s, "seconds_to_first_timed_region": to_first_region_s,
                  "streams": len(pristine[0]), "host_streams_in_flight": GENERATOR_THREADS,
                  "rss_mib_runtime_up": rss_runtime_mib, "rss_mib_inputs_resident": rss_inputs_mib},
    }
    if extras and args.log2n >= 23 and not args.no_adversarial:
        # BASELINE.json configs[3]: the adversarial inputs at the headline size, in every run of this command
        t_adv = time.perf_counter()
        try:
            result["adversarial"] = adversarial_block(torch, executor, sorter, fresh, n, device)
        except RuntimeError as e:   # (a wrong result or a device failure here must not cost the headline its line: it is reported)
            result["adversarial_error"] = str(e)
            print(f"[bench] adversarial block failed: {e}", file=sys.stderr)
        result["adversarial_protocol"] = ("bench/bench.cc:66-112 as in `sweep`: 1 warm-up + 5 timed sorts per pattern and mode, "
                                          "median GPU time; values = iota; slowdown against the 'uniform' row; plan_taken = the "
                                          "device's verdict (vrdxHipReadPlanVerdict); parity proven on the device; measured on "
                                          + box + " in %.1f s" % (time.perf_counter() - t_adv))
    if extras and not args.no_sweep:
        t_curve = time.perf_counter()
        result["sweep"] = size_curve(torch, executor, sorter, fresh, device, list(range(18, args.log2n + 1)))
        if args.log2n == 25:
            result["hbm_resident"] = hbm_resident_point(torch, executor, sorter, fresh, device)
        result["sweep_protocol"] = ("bench/bench.cc:15-20,66-112: 1 warm-up + 10 timed sorts of fresh std::mt19937 data per "
                                    "size and mode, median, GPU time between two events around the sort; measured on " + box +
                                    " in %.1f s" % (time.perf_counter() - t_curve))
    del pristine, fresh
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(n)
    result["setup"]["peak_rss_mib"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0   # (Linux: KiB; incl. the cpu_baseline leg)
    if rank == 0:
        print(json.dumps(result))
    executor.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
