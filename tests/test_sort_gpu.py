"""GPU parity tests: the HIP path, called through the C-ABI (ctypes -> libvrdx_hip.so), must be
bit-identical to the oracle on the same seeded inputs, to the golden fixtures produced by the
reference's CPU backend, and -- at BASELINE.json's full size (N = 2^25) -- must pass the
size-independent properties (sortedness, stability via iota values, multiset checksum,
idempotence) plus the committed full-size checksums.

The predicate is the reference's own (bench/bench.cc:41-64): keys == std::sort, (keys, values) ==
std::stable_sort by key; the edge cases are the ones the reference never tests.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


@pytest.fixture(scope="module")
def sorter(torch_mod):
    import vulkan_radix_sort_amd as vrdx
    s = vrdx.Sorter()  # raises VrdxError if the HIP library cannot drive this GPU: no fallback
    yield s
    s.destroy()


def _u32_to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint32).view(np.int32)).cuda()


def _to_u32(t):
    return t.cpu().numpy().view(np.uint32)


def gpu_sort(torch, sorter, keys, values=None, count=None, indirect=False, max_count=None, poison=True,
             query_pool=None, storage_out=None):
    """Runs one vrdxCmdSort* on torch's current stream; returns (keys, values) numpy arrays of the
    FULL buffers (so callers can check the untouched tail).  storage_out: a list that receives the storage tensor."""
    n_buf = len(keys)
    n = n_buf if count is None else count
    dk = _u32_to_dev(torch, keys)
    dv = _u32_to_dev(torch, values) if values is not None else None
    max_count = n if max_count is None else max_count
    req = (sorter.key_value_storage_requirements(max_count) if values is not None
           else sorter.storage_requirements(max_count))
    assert req.usage == 0x22
    storage = torch.full((req.size + 256,), 0xA5 if poison else 0, dtype=torch.uint8, device="cuda")
    storage[req.size:] = 0x5A  # guard band behind the storage
    stream = torch.cuda.current_stream().cuda_stream
    if indirect:
        dcount = _u32_to_dev(torch, np.array([n, 0, 0, 0], dtype=np.uint32))
        if values is None:
            sorter.cmd_sort_indirect(stream, max_count, dcount.data_ptr(), 0, dk.data_ptr(), 0,
                                     storage.data_ptr(), 0, query_pool, 0)
        else:
            sorter.cmd_sort_key_value_indirect(stream, max_count, dcount.data_ptr(), 0, dk.data_ptr(), 0,
                                               dv.data_ptr(), 0, storage.data_ptr(), 0, query_pool, 0)
    else:
        if values is None:
            sorter.cmd_sort(stream, n, dk.data_ptr(), 0, storage.data_ptr(), 0, query_pool, 0)
        else:
            sorter.cmd_sort_key_value(stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0,
                                      query_pool, 0)
    torch.cuda.synchronize()
    if n > 0:
        assert sorter.read_status(stream, storage.data_ptr(), 0) == 0, "look-back spin expired"
    assert bool((storage[req.size:] == 0x5A).all()), "wrote past the storage requirement"
    if storage_out is not None:
        storage_out.append(storage)
    return _to_u32(dk), (_to_u32(dv) if dv is not None else None)


def check_against_oracle(torch, sorter, oracle, keys, values=None, **kw):
    count = kw.get("count")
    gk, gv = gpu_sort(torch, sorter, keys, values, **kw)
    ek, ev, _ = oracle.sort(keys, values, count=count)
    assert np.array_equal(gk, ek)
    if values is not None:
        assert np.array_equal(gv, ev)


MSD_FROM = 8_150_000        # vrdx_api.cpp MsdBits: sorts past the end of the eight-bit plan (8.1 M) record the MSD plan in front of the passes
MSD_FROM_KEYS = MSD_FROM    # (keys-only and key+value alike since the half-size bucket kernel)
MSD_HALF_UP_TO = 18_149_376  # ... with buckets of at most 18432 (512-thread bucket kernel) while ceil(n / 1024) * 104 // 100 <= 18432


def msd_capacity(n, bits):
    """the bucket capacity the recorder checks on the device (MsdBits: 3 % headroom over the mean bucket)"""
    mean = -(-n // (1 << bits))
    return 18432 if bits == 10 and mean * 104 // 100 <= 18432 else 36864


def decline_msd(keys):
    """Tests that are ABOUT the four passes (tile plans, block sums, the look-back) at sizes where uniform keys would take
    the MSD plan recorded in front of them: 40000 keys spread over the array get the same top eleven bits -- a bucket
    beyond the plan's capacity of 36864 whether it scatters by ten or eleven bits -- and the device turns the plan down.
    The keys stay as good as uniform for everything else.  Returns keys (modified in place)."""
    n = len(keys)
    if n >= MSD_FROM_KEYS:
        step = max(1, n // 40000)
        keys[::step][:40000] = (keys[::step][:40000] & np.uint32(0x001FFFFF)) | np.uint32(0x2AC << 21)
    return keys


def plan_word(storage):
    """low byte of word 1 of the storage (what vrdxHipReadPlanVerdict returns): 3 = the MSD plan recorded in front of the
    passes took the sort, 4 = it found all keys identical and left them alone"""
    return int(storage[4:8].cpu().numpy().view(np.uint32)[0]) & 0xFF


SIZES = [0, 1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 8191, 8192, 8193, 12411, 16383, 16384,
         16385, 24577, 65539, 262144, 262145, (1 << 20) + 7]


@pytest.mark.parametrize("n", SIZES)
def test_keys_and_key_value_match_oracle(torch_mod, sorter, oracle, n):
    k, v = oracle.generate(1, n, 32)
    check_against_oracle(torch_mod, sorter, oracle, k)
    check_against_oracle(torch_mod, sorter, oracle, k, v)


def test_golden_vectors(torch_mod, sorter, golden):
    arrays = golden["arrays"]
    for case in golden["vectors"]:
        tag = case["tag"]
        k, v = arrays[tag + "_keys"], arrays[tag + "_values"]
        gk, gv = gpu_sort(torch_mod, sorter, k, v)
        assert np.array_equal(gk, arrays[tag + "_sorted_keys"]), tag
        assert np.array_equal(gv, arrays[tag + "_sorted_values"]), tag
        gk, _ = gpu_sort(torch_mod, sorter, k)
        assert np.array_equal(gk, arrays[tag + "_sorted_keys"]), tag


def test_golden_hashes(torch_mod, sorter, oracle, golden):
    for h in golden["hashes"]:
        if h["n"] > (1 << 20) + 7:
            continue
        k, v = oracle.generate(h["seed"], h["n"], h["bits"])
        gk, gv = gpu_sort(torch_mod, sorter, k, v)
        assert f"{oracle.hash(gk):016x}" == h["sorted_keys_hash"], h
        assert f"{oracle.hash(gv):016x}" == h["sorted_values_hash"], h


@pytest.mark.parametrize("bits", [0, 1, 2, 4, 8, 12, 16, 24])
def test_duplicate_heavy_keys_are_stable(torch_mod, sorter, oracle, bits):
    for n in (5000, 70001, 1 << 20):
        k, _ = oracle.generate(7, n, bits)
        iota = np.arange(n, dtype=np.uint32)
        check_against_oracle(torch_mod, sorter, oracle, k, iota)
        hi = ((k.astype(np.uint64) << (32 - max(bits, 1))) & 0xFFFFFFFF).astype(np.uint32) | (k & 0xFF)
        check_against_oracle(torch_mod, sorter, oracle, hi, iota)


def test_adversarial_inputs(torch_mod, sorter, oracle):
    # BASELINE.json configs[3] at a size the oracle checks in seconds; full size is below
    n = (1 << 21) + 12345
    iota = np.arange(n, dtype=np.uint32)
    rng = np.random.default_rng(3)
    cases = {
        "all-equal": np.full(n, 0x12345678, np.uint32),
        "all-sentinel": np.full(n, 0xFFFFFFFF, np.uint32),
        "all-zero": np.zeros(n, np.uint32),
        "descending": (n - 1 - iota).astype(np.uint32),
        "ascending": iota.copy(),
        "few-distinct": rng.choice(np.array([0xFFFFFFFF, 0, 0x80000001, 0x7FFFFF00], np.uint32), size=n),
        "eighth-sentinel": np.where(rng.integers(0, 8, n) == 0, np.uint32(0xFFFFFFFF),
                                    rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)),
    }
    for p in range(4):
        r = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        cases[f"digit{p}-constant"] = r & np.uint32(~(0xFF << (8 * p)) & 0xFFFFFFFF)
    for name, k in cases.items():
        gk, gv = gpu_sort(torch_mod, sorter, k, iota)
        ek, ev, _ = oracle.sort(k, iota)
        assert np.array_equal(gk, ek), name
        assert np.array_equal(gv, ev), name


@pytest.mark.parametrize("kv", [False, True])
def test_indirect_count_and_untouched_tail(torch_mod, sorter, oracle, kv):
    # device-side count < host bound; elements at index >= count must not be touched
    # (downsweep.slang:199,220); the grid is sized from the bound (src/vk_radix_sort.h.in:353)
    for n_buf, count in [(9000, 5001), (9000, 0), (9000, 9000), (300000, 123457), (70000, 1)]:
        k, v = oracle.generate(11, n_buf, 32)
        check_against_oracle(torch_mod, sorter, oracle, k, v if kv else None, count=count, indirect=True,
                             max_count=n_buf)
        check_against_oracle(torch_mod, sorter, oracle, k, v if kv else None, count=count)
    # count above the bound is clamped to the bound
    k, v = oracle.generate(12, 20000, 32)
    gk, gv = gpu_sort(torch_mod, sorter, k, v if kv else None, count=30000, indirect=True, max_count=15000)
    ek, ev, _ = oracle.sort(k, v if kv else None, count=15000)
    assert np.array_equal(gk, ek) and (not kv or np.array_equal(gv, ev))


@pytest.mark.parametrize("kv", [False, True])
def test_indirect_count_with_large_bound(torch_mod, sorter, oracle, kv):
    """Indirect sorts whose host-side bound selects the big-tile kernels (12 M: the two-sub-tile kernel
    for keys-only, 32768-key tiles for key+value) while the device-side count is anything from 0 to the
    bound: workgroups beyond the real tile count leave at once, the tail is untouched."""
    bound = 12_000_017
    k, v = oracle.generate(13, bound, 32)
    for count in (0, 1, 65_537, 9_000_001, bound):
        check_against_oracle(torch_mod, sorter, oracle, k, v if kv else None, count=count, indirect=True,
                             max_count=bound)


def test_buffer_offsets_and_shared_buffer(torch_mod, sorter, oracle):
    # keys, values and the count in ONE buffer at different offsets, exactly like the reference's
    # only KV call site (bench/vulkan_benchmark.cc:346-358,386-388)
    torch = torch_mod
    n = 100003
    k, v = oracle.generate(5, n, 32)
    inout = (n * 4 + 15) // 16 * 16
    buf = torch.zeros(2 * inout + 16, dtype=torch.uint8, device="cuda")
    host = np.zeros(2 * inout + 16, dtype=np.uint8)
    host[:n * 4] = k.view(np.uint8)
    host[inout:inout + n * 4] = v.view(np.uint8)
    host[2 * inout:2 * inout + 4] = np.array([n], np.uint32).view(np.uint8)
    buf.copy_(torch.from_numpy(host))
    req = sorter.key_value_storage_requirements(n)
    pad = 4096  # non-zero storageOffset
    storage = torch.zeros(req.size + pad, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sorter.cmd_sort_key_value_indirect(stream, n, buf.data_ptr(), 2 * inout, buf.data_ptr(), 0, buf.data_ptr(),
                                       inout, storage.data_ptr(), pad)
    torch.cuda.synchronize()
    out = buf.cpu().numpy()
    ek, ev, _ = oracle.sort(k, v)
    assert np.array_equal(out[:n * 4].view(np.uint32), ek)
    assert np.array_equal(out[inout:inout + n * 4].view(np.uint32), ev)
    assert not storage[:pad].any()  # nothing before storageOffset


def test_storage_requirements_through_the_c_abi_match_the_golden_table(sorter, golden):
    """the sizes the exported calculators return (vrdxGetSorter[KeyValue]StorageRequirements through
    ctypes, on the GPU box's sorter) equal the table computed from the reference's formulas
    (src/vk_radix_sort.h.in:279-308), row by row; usage = STORAGE_BUFFER | TRANSFER_DST."""
    assert golden["usage"] == 0x22
    for row in golden["storage_align16"]:
        k = sorter.storage_requirements(row["n"])
        kv = sorter.key_value_storage_requirements(row["n"])
        assert (k.size, k.usage) == (row["keys"], 0x22), row
        assert (kv.size, kv.usage) == (row["key_value"], 0x22), row


@pytest.mark.parametrize("n", [70_001, 3_000_001])
def test_direct_modes_with_buffer_offsets_and_a_query_offset(torch_mod, sorter, oracle, n):
    """vrdxCmdSort / vrdxCmdSortKeyValue with non-zero keysOffset, valuesOffset and storageOffset (keys and values
    in ONE buffer, bench/vulkan_benchmark.cc:346-358) and the timestamps at [query, query + 15) of a larger pool
    (src/vk_radix_sort.h.in:39-50): query = 7 of 22 slots; the slots before stay unwritten, and
    vrdxHipGetQueryPoolResults(firstQuery = 7) reports the sort's 15 slots relative to its first."""
    import vulkan_radix_sort_amd as vrdx
    torch = torch_mod
    k, v = oracle.generate(21, n, 32)
    ek, ev, _ = oracle.sort(k, v)
    inout = (n * 4 + 15) // 16 * 16
    lead = 4096  # keysOffset
    host = np.full(lead + 2 * inout + 64, 0xEE, dtype=np.uint8)
    host[lead:lead + n * 4] = k.view(np.uint8)
    host[lead + inout:lead + inout + n * 4] = v.view(np.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    pad = 8192  # storageOffset
    for kv in (False, True):
        buf = torch.from_numpy(host.copy()).cuda()
        req = sorter.key_value_storage_requirements(n) if kv else sorter.storage_requirements(n)
        storage = torch.zeros(req.size + pad, dtype=torch.uint8, device="cuda")
        pool = vrdx.QueryPool(22)
        if kv:
            sorter.cmd_sort_key_value(stream, n, buf.data_ptr(), lead, buf.data_ptr(), lead + inout,
                                      storage.data_ptr(), pad, pool, 7)
        else:
            sorter.cmd_sort(stream, n, buf.data_ptr(), lead, storage.data_ptr(), pad, pool, 7)
        torch.cuda.synchronize()
        out = buf.cpu().numpy()
        assert np.array_equal(out[lead:lead + n * 4].view(np.uint32), ek)
        if kv:
            assert np.array_equal(out[lead + inout:lead + inout + n * 4].view(np.uint32), ev)
        else:
            assert np.array_equal(out[lead + inout:lead + inout + n * 4].view(np.uint32), v)   # values untouched
        assert bool((out[:lead] == 0xEE).all()) and bool((out[lead + n * 4:lead + inout] == 0xEE).all())
        assert bool((out[lead + inout + n * 4:] == 0xEE).all())
        assert not storage[:pad].any()                                   # nothing before storageOffset
        assert sorter.read_status(stream, storage.data_ptr(), pad) == 0
        ts = pool.results_ns(7, 15)
        assert len(ts) == 15 and ts[0] == 0 and all(b >= a for a, b in zip(ts, ts[1:])) and ts[14] > 0
        assert ts[4] - ts[3] > 0                                         # pass 0's "downsweep"
        with pytest.raises(vrdx.VrdxError):                              # slots 0..6 were never written: VK_NOT_READY
            pool.results_ns(0, 7)
        with pytest.raises(vrdx.VrdxError):                              # beyond the pool
            pool.results_ns(8, 15)
        pool.destroy()


def test_clean_storage_reuse_and_repeat(torch_mod, sorter, oracle):
    torch = torch_mod
    n = 300000
    k, v = oracle.generate(9, n, 32)
    ek, ev, _ = oracle.sort(k, v)
    gk, gv = gpu_sort(torch, sorter, k, v, poison=False)
    assert np.array_equal(gk, ek) and np.array_equal(gv, ev)
    # same storage, several sorts back to back on one stream, no host sync in between
    req = sorter.key_value_storage_requirements(n)
    storage = torch.empty(req.size, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for seed in (1, 2, 3):
        kk, vv = oracle.generate(seed, n, 32)
        dk, dv = _u32_to_dev(torch, kk), _u32_to_dev(torch, vv)
        sorter.cmd_sort_key_value(stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0)
        outs.append((kk, vv, dk, dv))
    torch.cuda.synchronize()
    for kk, vv, dk, dv in outs:
        ek, ev, _ = oracle.sort(kk, vv)
        assert np.array_equal(_to_u32(dk), ek) and np.array_equal(_to_u32(dv), ev)
    # idempotence: sorting sorted data changes nothing
    gk2, gv2 = gpu_sort(torch, sorter, ek, ev)
    assert np.array_equal(gk2, ek) and np.array_equal(gv2, ev)


def test_timestamp_contract(torch_mod, sorter, oracle):
    # 15 slots, monotone, slot 14 - slot 0 = whole sort (src/vk_radix_sort.h.in:39-50;
    # consumer formula bench/vulkan_benchmark.cc:330-337)
    import vulkan_radix_sort_amd as vrdx
    pool = vrdx.QueryPool(15)
    k, v = oracle.generate(1, 1 << 20, 32)
    kept = []
    gk, gv = gpu_sort(torch_mod, sorter, k, v, query_pool=pool, storage_out=kept)
    ts = pool.results_ns()
    assert len(ts) == 15 and ts[0] == 0
    assert all(b >= a for a, b in zip(ts, ts[1:]))
    up = sum(ts[2 + 3 * p] - ts[1 + 3 * p] for p in range(4))
    sp = sum(ts[3 + 3 * p] - ts[2 + 3 * p] for p in range(4))
    dn = sum(ts[4 + 3 * p] - ts[3 + 3 * p] for p in range(4))
    assert up > 0 and dn > 0 and up + sp + dn <= ts[14]
    # the look-back is fused into the pass ("spine" = 0) and slots written back to back share one event
    # (2^20 elements record the hybrid plan: its bucket sort sits in pass 1's "upsweep" slot)
    assert sp == 0 and ts[14] == ts[13] and ts[5] >= ts[4] and all(ts[2 + 3 * p] == ts[1 + 3 * p] for p in range(2, 4))
    # ... and on these uniform keys the plan applies (launches 1..3 have nothing to do): launch 0 says so in word 1 of
    # the storage (VRDX_OFF_PLAN: 1 = the hybrid plan, 2 = four passes) -- a fact, not a ratio of two durations
    assert int(kept[0][4:8].cpu().numpy().view(np.uint32)[0]) == 1
    # n == 0 still records all 15 slots
    gpu_sort(torch_mod, sorter, k[:0], v[:0], query_pool=pool)
    assert len(pool.results_ns()) == 15
    pool.destroy()


def test_timestamp_contract_of_the_msd_plan(torch_mod, sorter, oracle):
    """The same 15 slots when the MSD plan takes the sort (include/vk_radix_sort.h): its three stages carry the names they
    have in the reference -- slot 2 "upsweep" (the histogram with per-tile counts), 3 "spine" (a real stage here), 4
    "downsweep" (the scatter) -- the bucket sorts are pass 1's "upsweep" (slot 5), and the four returning passes share the
    slots behind.  Monotone, all recorded, the whole sort = slot 14 - slot 0."""
    import vulkan_radix_sort_amd as vrdx
    pool = vrdx.QueryPool(15)
    n = MSD_FROM + 11
    k, v = oracle.generate(1, n, 32)
    kept = []
    gk, gv = gpu_sort(torch_mod, sorter, k, v, query_pool=pool, storage_out=kept)
    ek, ev, _ = oracle.sort(k, v)
    assert np.array_equal(gk, ek) and np.array_equal(gv, ev) and plan_word(kept[0]) == 3
    ts = pool.results_ns()
    assert len(ts) == 15 and ts[0] == 0 and all(b >= a for a, b in zip(ts, ts[1:]))
    assert ts[2] > ts[1] and ts[3] > ts[2] and ts[4] > ts[3] and ts[5] > ts[4]      # four real stages
    assert ts[4] - ts[3] > 3 * (ts[3] - ts[2])                                        # ... of which the spine is the short one
    assert ts[6] == ts[5] and ts[9] == ts[8] == ts[7] and ts[14] == ts[13] and ts[14] - ts[5] < ts[5] - ts[4]
    pool.destroy()


def test_sorts_on_a_side_stream_and_two_sorters_concurrently(torch_mod, oracle):
    # sorter is immutable: different streams + different storage may run concurrently
    # (SURVEY.md section 8b "Threading")
    import vulkan_radix_sort_amd as vrdx
    torch = torch_mod
    n = 500000
    sorters = [vrdx.Sorter(0), vrdx.Sorter(None)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    data = []
    torch.cuda.synchronize()
    for i, (s, st) in enumerate(zip(sorters, streams)):
        k, v = oracle.generate(20 + i, n, 32)
        with torch.cuda.stream(st):
            dk, dv = _u32_to_dev(torch, k), _u32_to_dev(torch, v)
            storage = torch.empty(s.key_value_storage_requirements(n).size, dtype=torch.uint8, device="cuda")
            for _ in range(3):  # re-sorting sorted data is a no-op, keeps both streams busy
                s.cmd_sort_key_value(st.cuda_stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0)
        data.append((k, v, dk, dv, storage))
    torch.cuda.synchronize()
    for k, v, dk, dv, _ in data:
        ek, ev, _h = oracle.sort(k, v)
        assert np.array_equal(_to_u32(dk), ek) and np.array_equal(_to_u32(dv), ev)
    for s in sorters:
        s.destroy()


def test_capturable_into_a_hip_graph(torch_mod, sorter, oracle):
    # VkCommandBuffer == stream-ordered enqueue: recording inside a stream capture must work.
    # tests/native/vrdx_selftest checks this with the bare HIP graph API (hipStreamBeginCapture ->
    # vrdxCmdSort* -> hipGraphLaunch) on a cold process.  Through torch.cuda.CUDAGraph the replay was
    # once observed to be a silent no-op when the FIRST launch of the sort kernels in the process
    # happened inside the capture, so one eager sort runs first here; the cold case (capture as the very
    # first use of a new sorter in a new process) is test_graph_capture_in_a_fresh_process below.
    torch = torch_mod
    n = 200000
    k, v = oracle.generate(31, n, 32)
    dk, dv = _u32_to_dev(torch, k), _u32_to_dev(torch, v)
    storage = torch.empty(sorter.key_value_storage_requirements(n).size, dtype=torch.uint8, device="cuda")
    warm_k, warm_v = _u32_to_dev(torch, k), _u32_to_dev(torch, v)
    sorter.cmd_sort_key_value(torch.cuda.current_stream().cuda_stream, n, warm_k.data_ptr(), 0, warm_v.data_ptr(), 0,
                              storage.data_ptr(), 0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sorter.cmd_sort_key_value(torch.cuda.current_stream().cuda_stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0,
                                  storage.data_ptr(), 0)
    ek, ev, _ = oracle.sort(k, v)
    for _ in range(2):  # replay twice: fresh unsorted input each time
        dk.copy_(_u32_to_dev(torch, k))
        dv.copy_(_u32_to_dev(torch, v))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(_to_u32(dk), ek) and np.array_equal(_to_u32(dv), ev)
        assert sorter.read_status(torch.cuda.current_stream().cuda_stream, storage.data_ptr(), 0) == 0


def test_one_captured_graph_sorts_whatever_keys_it_is_replayed_on(torch_mod, sorter, oracle):
    """Record once, submit many (the reference's model, bench/vulkan_benchmark.cc:292-302) at a size that records the MSD plan:
    what the plan does is decided on the DEVICE at every replay -- where the scatter's window lies, whether the plan runs at
    all -- so ONE captured sort must be right for uniform keys (window at the top), 24-bit keys (window below a prefix),
    descending ids, all-identical keys (verdict 4: nothing moved), four distinct values and a broken prefix (both: the four
    passes), replayed in that order on the same buffers, keys-only and key+value."""
    import vulkan_radix_sort_amd as vrdx
    torch = torch_mod
    n = 9_000_017
    rng = np.random.default_rng(5)
    r = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    iota = np.arange(n, dtype=np.uint32)
    broken = (r >> np.uint32(8)).copy()
    broken[n // 2 + 3] |= np.uint32(0x80000000)
    inputs = [("uniform", r, vrdx.VERDICT_MSD_RUNS), ("24-bit", r >> np.uint32(8), vrdx.VERDICT_MSD_RUNS),
              ("descending", (n - 1 - iota).astype(np.uint32), vrdx.VERDICT_MSD_RUNS),
              ("all equal", np.full(n, 0xCAFEF00D, np.uint32), vrdx.VERDICT_MSD_SORTED),
              ("four values", np.array([7, 0xFFFFFFFF, 0x00020000, 0x7E000000], np.uint32)[rng.integers(0, 4, n)], vrdx.VERDICT_NONE),
              ("broken prefix", broken, vrdx.VERDICT_NONE), ("uniform again", r[::-1].copy(), vrdx.VERDICT_MSD_RUNS)]
    stream = torch.cuda.current_stream().cuda_stream
    for key_value in (False, True):
        assert sorter.describe_plan(n, key_value).name == "msd"
        dk, dv = _u32_to_dev(torch, r), _u32_to_dev(torch, iota)
        size = (sorter.key_value_storage_requirements(n) if key_value else sorter.storage_requirements(n)).size
        storage = torch.empty(size, dtype=torch.uint8, device="cuda")
        record = ((lambda st: sorter.cmd_sort_key_value(st, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0))
                  if key_value else (lambda st: sorter.cmd_sort(st, n, dk.data_ptr(), 0, storage.data_ptr(), 0)))
        record(stream)   # (one eager sort first, see test_capturable_into_a_hip_graph)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            record(torch.cuda.current_stream().cuda_stream)
        for name, k, verdict in inputs:
            dk.copy_(_u32_to_dev(torch, k))
            dv.copy_(_u32_to_dev(torch, iota))
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            ek, ep, _ = oracle.sort(k, iota)
            assert np.array_equal(_to_u32(dk), ek), (name, key_value)
            assert not key_value or np.array_equal(_to_u32(dv), ep), (name, key_value)
            assert sorter.read_plan_verdict(stream, storage.data_ptr(), 0) == verdict, (name, key_value)
            assert sorter.read_status(stream, storage.data_ptr(), 0) == 0


@pytest.mark.parametrize("key_value", [False, True])
def test_two_tile_scatter_at_its_tile_boundaries(torch_mod, sorter, oracle, key_value):
    """The keys-only scatter of the ten-bit MSD plan takes two consecutive tiles per workgroup and stages them in two halves
    (vrdx_kernels.hip, ScatterMsdPairBody): device-side counts that end exactly at, one before and one behind a tile, a
    pair of tiles, and the middle of either half -- with the bound at 2^25 (tiles of 32768 keys) and at a size whose tiles
    are smaller and odd in number (18149377: 739 tiles of 24576).  Key+value (one tile per workgroup) for comparison."""
    for bound in (1 << 25, 18_149_377):
        k, _ = oracle.generate(17, bound, 32)
        v = np.arange(bound, dtype=np.uint32) if key_value else None
        tile = 32768 if bound == 1 << 25 else 24576
        base = 200 * 2 * tile
        for count in (base, base - 1, base + 1, base + tile, base + tile - 1, base + tile + 1, base + tile // 2,
                      base + tile + tile // 2 + 5, bound - 1, bound):
            check_against_oracle(torch_mod, sorter, oracle, k, v, count=count, indirect=True, max_count=bound)


def _random_cases(count, seed):
    """(n, bits, key+value, indirect, count) drawn from a fixed stream: sizes log-uniform over 1 .. 3 M (every
    path: single workgroup, 1024x8 / x16 / x32 tiles), keys of 0 .. 32 significant bits (0, 8, 16, 24: passes
    with a constant digit are skipped or copied), shifted into a random byte position."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        n = int(2 ** rng.uniform(0, np.log2(3_000_000)))
        bits = int(rng.choice([0, 1, 3, 8, 12, 16, 24, 32]))
        shift = int(rng.integers(0, 33 - bits)) if bits < 32 else 0
        kv = bool(rng.integers(0, 2))
        indirect = bool(rng.integers(0, 2))
        cnt = int(rng.integers(0, n + 1)) if indirect else None
        cases.append((n, bits, shift, kv, indirect, cnt))
    return cases


@pytest.mark.parametrize("batch", range(4))
def test_randomized_sizes_entropies_and_modes(torch_mod, sorter, oracle, batch):
    """48 cases per run (4 batches of 12) from a fixed random stream: any size, any key width at any bit
    position, keys-only or key+value (values = iota), direct or indirect with any device-side count --
    each bit-exact against the oracle, the tail beyond the count untouched."""
    for n, bits, shift, kv, indirect, cnt in _random_cases(12, 1000 + batch):
        k, _ = oracle.generate(7 * batch + n % 13, n, bits)
        k = (k << np.uint32(shift)).astype(np.uint32) if shift else k
        v = np.arange(n, dtype=np.uint32) if kv else None
        gk, gv = gpu_sort(torch_mod, sorter, k, v, count=cnt, indirect=indirect, max_count=n if indirect else None)
        ek, ev, _ = oracle.sort(k, v, count=cnt)
        what = "n=%d bits=%d shift=%d kv=%s indirect=%s count=%s" % (n, bits, shift, kv, indirect, cnt)
        assert np.array_equal(gk, ek), what
        assert not kv or np.array_equal(gv, ev), what


def test_graph_capture_in_a_fresh_process(torch_mod):
    """vrdxCreateSorter loads every kernel it may launch (it sets their LDS limits), so a sort may be
    captured into a graph as the first thing a process does with the sorter: general path and
    single-workgroup path, checked against the oracle by tests/cold_capture_check.py."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "cold_capture_check.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": OK") == 2 and "NOT SORTED" not in out.stdout, out.stdout + out.stderr


def test_refused_enqueue_is_latched_in_the_sorter_status(torch_mod):
    """vrdxCmdSort* return void; an enqueue the runtime refuses shows up as bit 31 of
    vrdxHipReadSorterStatus, once (tests/enqueue_error_check.py, in a process of its own because the
    test hook is read once per process)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "enqueue_error_check.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_device_failure_is_sticky_in_the_sorter_status(torch_mod):
    """A look-back that gives up (forced in the -DVRDX_TESTING build by a spin limit of 0) sets the storage's
    failure word and the sorter's sticky word; the next sort on that storage clears only the former
    (tests/sticky_status_check.py, in a process of its own)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sticky_status_check.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.parametrize("n", [20_000, 300_000, 1_500_000, (1 << 21), 3_500_001, 3_700_001, 6_000_001, 7_600_000, 8_100_000])
def test_hybrid_plan_and_its_fallback_at_the_bucket_capacity(torch_mod, sorter, oracle, n):
    """Mid-size sorts record the hybrid plan (scatter by the top byte, then one workgroup per bucket); the DEVICE
    keeps the four passes when a bucket exceeds the capacity (4096 / 8192 / 16384, twice the mean; the largest one,
    32768 -- key+value stages keys and values through one buffer there -- is recorded up to a mean bucket of
    capacity / 1.03: 8.1 M elements).
    Uniform keys with ONE top byte brought to exactly the capacity (plan applies) and to capacity + 1 (four passes),
    keys-only and key+value (values = iota: the permutation itself), direct and indirect with a smaller count."""
    need = 2 * ((n + 255) // 256)
    cap = 4096 if need <= 4096 else 8192 if need <= 8192 else 16384 if need <= 16384 else 32768
    rng = np.random.default_rng(n)
    iota = np.arange(n, dtype=np.uint32)
    for heavy in (cap, cap + 1):
        k = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        k[(k >> 24) == 0x5A] ^= np.uint32(0x01000000)          # nobody in bucket 0x5A ...
        where = rng.choice(n, size=heavy, replace=False)
        k[where] = (k[where] & np.uint32(0x00FFFFFF)) | np.uint32(0x5A000000)   # ... except exactly `heavy` keys
        assert int(((k >> 24) == 0x5A).sum()) == heavy
        ek, ep, _ = oracle.sort(k, iota)
        gk, _ = gpu_sort(torch_mod, sorter, k)
        assert np.array_equal(gk, ek), heavy
        gk, gp = gpu_sort(torch_mod, sorter, k, iota)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), heavy
        count = n - n // 3
        ek, ep, _ = oracle.sort(k, iota, count=count)
        gk, gp = gpu_sort(torch_mod, sorter, k, iota, count=count, indirect=True, max_count=n)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), heavy


def test_four_pass_plan_at_mid_sizes_with_the_hybrid_plan_switched_off():
    """VRDX_HYBRID=0: the native parity battery (255 cases up to 3 M elements) on the four-pass plan alone."""
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, VRDX_HYBRID="0"))
    assert out.returncode == 0 and ", 0 failures" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_four_pass_plan_at_large_sizes_with_the_msd_plan_switched_off():
    """VRDX_MSD=0: sorts beyond the eight-bit plan's reach record the four passes alone (what N > 67 M runs, and what the MSD
    plan's tables are measured against).  The native battery of uniform keys in every mode, duplicates (stability), 24-bit
    keys (a trivial pass), one heavy bucket and descending keys at 12 M (one and a half rounds of tiles: tail split) and
    16.2 M elements."""
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    out = subprocess.run([exe, "msd", "12000003", "16200000"], capture_output=True, text=True, timeout=1200,
                         env=dict(os.environ, VRDX_MSD="0"))
    assert out.returncode == 0 and ", 0 failures" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("n", [256 * 1024 * 20, 256 * 1024 * 20 + 1, 256 * 1024 * 28 - 5, 256 * 2048 * 20, 256 * 2048 * 24 + 1,
                               256 * 2048 * 32 - 4095])
def test_even_split_tiles_at_their_slot_boundaries(torch_mod, sorter, oracle, n):
    """Keys-only sorts of one round of 1024x32 / 1024x32x2 tiles are cut into 256 EQUAL tiles of s slots per wave
    (PlanTiles in vrdx_layout.h, s a multiple of 4): sizes that fill s slots exactly, that need one key more
    (s + 4), and ragged ones, direct and indirect with a smaller count (whole tiles past the count, a ragged tile in
    the middle of the grid)."""
    k, _ = oracle.generate(11, n, 32)
    ek, _, _ = oracle.sort(k)
    gk, _ = gpu_sort(torch_mod, sorter, k)
    assert np.array_equal(gk, ek)
    count = n - n // 3 - 1
    ek, _, _ = oracle.sort(k, count=count)
    gk, _ = gpu_sort(torch_mod, sorter, k, count=count, indirect=True, max_count=n)
    assert np.array_equal(gk, ek)


ROUND = 256 * 32768   # one round of 32768-element tiles on the 256 CUs of an MI355X


@pytest.mark.parametrize("n,key_value", [
    (2 * ROUND + 1, False),                    # one key past one round of 65536-key tiles: a tail of ONE small tile holding one key
    (2 * ROUND + 256 * 8192, False),           # 256 tail tiles of four slots per wave and sub-tile, all full
    (2 * ROUND + 256 * 8192 + 1, False),       # one key more: eight slots
    (4 * ROUND + 3 * 65536 + 12345, False),    # two whole rounds of the two-sub-tile kernel + a ragged tail
    (2 * ROUND + 1, True),                     # key+value: two rounds of 32768 + one pair
    (2 * ROUND + 256 * 4096 + 1, True),        # tail of eight slots
    (3 * ROUND + ROUND // 2, True),            # the rest is exactly half a round: the last size the tail split takes
    (3 * ROUND + ROUND // 2 + 4097, True),     # ... and past it: full tiles again
])
def test_tail_split_tiles_at_their_boundaries(torch_mod, sorter, oracle, n, key_value):
    """Sorts of more than one round keep their whole rounds of full tiles and cut the rest into up to 256 small equal
    tiles (PlanTiles in vrdx_layout.h; keys-only: the two-sub-tile kernel at every size, key+value: 1024x32 while the
    rest is at most half a round).  Sizes that put one element, exactly full small tiles, one element more and a ragged
    rest behind the whole rounds; direct, and indirect with a device-side count that ends inside the FULL tiles (every
    tail tile and some full tiles start past the count) and one that ends inside the TAIL."""
    k, v = oracle.generate(13, n, 32)
    decline_msd(k)   # (these sizes record the MSD plan in front of the passes: the tile plans under test are the fallback's)
    values = v if key_value else None
    kept = []
    gpu_sort(torch_mod, sorter, k, values, storage_out=kept)
    assert plan_word(kept[0]) != 3
    check_against_oracle(torch_mod, sorter, oracle, k, values)
    whole = (n // ROUND) * ROUND if key_value else (n // (2 * ROUND)) * 2 * ROUND
    for count in (whole - 70001, whole + (n - whole) // 2 + 1):
        check_against_oracle(torch_mod, sorter, oracle, k, values, count=count, indirect=True, max_count=n)


@pytest.mark.parametrize("n,key_value", [(MSD_FROM, False), (MSD_FROM, True), (ROUND, False), (ROUND + 12345, True),
                                         (12_000_001, False), (12_000_001, True), (MSD_HALF_UP_TO, True),
                                         (MSD_HALF_UP_TO + 1, False), (MSD_HALF_UP_TO + 1, True), (20_000_003, True),
                                         (1 << 25, False), (1 << 25, True), (36_500_000, False), (37_000_001, False),
                                         (45_000_000, True)])
def test_msd_plan_and_its_fallback_at_the_bucket_capacity(torch_mod, sorter, oracle, n, key_value):
    """Sorts of 8.15 M elements and more record the MSD plan in front of their four passes (vrdx_kernels.hip, "MSD plan"):
    per-tile counts of the top ten or eleven bits, a spine, ONE stable scatter by those bits and one workgroup per bucket
    that sorts it by the remaining bits in two in-LDS passes; the DEVICE keeps the four passes when a bucket exceeds the
    capacity -- 18432 up to 18.1 M elements (the half-size bucket kernel, two workgroups per CU), 36864 beyond.  Uniform
    keys with ONE bucket brought to exactly the capacity (the plan applies: word 1 of the storage says 3) and to one more
    (it does not), keys-only or key+value (values = iota: the permutation itself), direct and indirect with a smaller
    device-side count; the first sizes of the plan, one round of tiles, both sides of the switch between the two bucket
    kernels, the headline size, the last ten-bit and first eleven-bit sizes."""
    info = sorter.describe_plan(n, key_value)
    assert info.name == "msd" and info.bits == (10 if n <= 36_600_000 else 11), (info.name, info.bits)
    bits = int(info.bits)
    cap = msd_capacity(n, bits)
    assert cap == (18432 if n <= MSD_HALF_UP_TO else 36864)
    # launches: histogram, spine, scatter, buckets and the passes of the fallback that are launches of their own -- the scatter
    # launch is also pass 0, and the full-size bucket launch pass 1 (keys-only since round 5, key+value since round 6)
    if "VRDX_MSD_FUSED" not in os.environ and "VRDX_TILE_CONFIG" not in os.environ:
        assert info.launches == (7 if n <= MSD_HALF_UP_TO else 6), info.launches
    shift, low = np.uint32(32 - bits), np.uint32((1 << (32 - bits)) - 1)
    bucket = 0x155
    rng = np.random.default_rng(n)
    iota = np.arange(n, dtype=np.uint32)
    for heavy in (cap, cap + 1):
        k = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        mine = (k >> shift) == bucket                               # nobody is in that bucket ...
        elsewhere = rng.integers(0, bucket, size=int(mine.sum()), dtype=np.uint64).astype(np.uint32)
        k[mine] = (k[mine] & low) | (elsewhere << shift)
        where = rng.choice(n, size=heavy, replace=False)
        k[where] = (k[where] & low) | np.uint32(bucket << int(shift))   # ... except exactly `heavy` keys
        assert int(((k >> shift) == bucket).sum()) == heavy
        values = iota if key_value else None
        ek, ep, _ = oracle.sort(k, values)
        kept = []
        gk, gp = gpu_sort(torch_mod, sorter, k, values, storage_out=kept)
        assert np.array_equal(gk, ek) and (not key_value or np.array_equal(gp, ep)), heavy
        assert (plan_word(kept[0]) == 3) == (heavy == cap), (heavy, plan_word(kept[0]))
        count = n - n // 3
        ek, ep, _ = oracle.sort(k, values, count=count)
        gk, gp = gpu_sort(torch_mod, sorter, k, values, count=count, indirect=True, max_count=n)
        assert np.array_equal(gk, ek) and (not key_value or np.array_equal(gp, ep)), heavy


@pytest.mark.parametrize("n", [MSD_FROM + 7, (1 << 25) - 12345])
def test_msd_plan_stability_window_choice_and_inputs_it_declines(torch_mod, sorter, oracle, n):
    """The MSD plan on duplicate-heavy keys that fit its buckets (keys = eleven top bits | one middle bit | three low bits:
    the stability of the scatter and of both bucket passes is what keeps equal keys' values in input order); on inputs
    whose top bits are constant, which since round 6 it TAKES -- the device puts the scatter's window below the common
    prefix of a sample of the keys (24-bit keys like DataGenerator::Generate(n, 24), /root/reference/bench/data_generator.cc:15;
    the same duplicates under a nine-bit prefix; twelve-bit keys: one short local pass); and on the inputs it must turn
    down: the sample is only the guess, so ONE key outside the prefix, planted where no sample looks (index n - 1 is
    sampled: n - 2 and the middle are not), must send the sort to the four passes and still come out bit-exact; four distinct
    values; eight-bit keys.  All keys identical: verdict 4, nothing moved."""
    import vulkan_radix_sort_amd as vrdx
    rng = np.random.default_rng(n)
    iota = np.arange(n, dtype=np.uint32)
    r = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    dup = (r & np.uint32(0xFFE00000)) | (r & np.uint32(7)) | (((r >> np.uint32(3)) & np.uint32(1)) << np.uint32(12))
    stream = torch_mod.cuda.current_stream().cuda_stream

    def run(name, k, expect):
        kept = []
        gk, gp = gpu_sort(torch_mod, sorter, k, iota, storage_out=kept)
        ek, ep, _ = oracle.sort(k, iota)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), name
        verdict = sorter.read_plan_verdict(stream, kept[0].data_ptr(), 0)
        assert verdict == plan_word(kept[0]) and verdict == expect, (name, verdict, expect)
        kept = []
        gk, _ = gpu_sort(torch_mod, sorter, k, storage_out=kept)
        assert np.array_equal(gk, ek) and plan_word(kept[0]) == expect, name

    runs, sorted_, passes = vrdx.VERDICT_MSD_RUNS, vrdx.VERDICT_MSD_SORTED, vrdx.VERDICT_NONE
    run("duplicates", dup, runs)
    k24 = r >> np.uint32(8)
    run("24-bit", k24, runs)
    run("duplicates under a prefix", np.uint32(0x5A800000) | (dup >> np.uint32(9)), runs)
    run("12-bit", r >> np.uint32(20), runs)
    for where in (n - 2, n // 2 + 1):
        broken = k24.copy()
        broken[where] |= np.uint32(0x40000000)
        run(f"24-bit with one key outside the prefix at {where}", broken, passes)
    four = np.array([0xFFFFFFFF, 0, 0x80000001, 0x7FFFFF00], np.uint32)[rng.integers(0, 4, n)]
    run("four values", four, passes)
    run("8-bit", r >> np.uint32(24), passes)
    same = np.full(n, 0x12345678, np.uint32)
    run("all equal", same, sorted_)
    same[n - 2] = 0x12345679
    run("all equal but one", same, passes)
    recorded, declined = sorter.read_plan_counters(stream)
    assert recorded >= 22 and declined >= 10 and declined < recorded, (recorded, declined)


@pytest.mark.parametrize("n", [1 << 25, 20_000_003, 12_000_001])
def test_msd_plan_takes_dense_sorted_ids(torch_mod, sorter, oracle, n):
    """BASELINE config 4's descending keys (and ascending ones): N - 1 - i fills the power of two above N well enough at
    these sizes that every bucket of the window the device chooses fits, so the two-trip plan runs (round 5: the four
    passes + 10 %).  Stable (values = iota), keys-only as well."""
    iota = np.arange(n, dtype=np.uint32)
    for name, k in (("descending", (n - 1 - iota).astype(np.uint32)), ("ascending", iota.copy())):
        kept = []
        gk, gp = gpu_sort(torch_mod, sorter, k, iota, storage_out=kept)
        assert np.array_equal(gk, iota) and np.array_equal(gp, iota if name == "ascending" else (n - 1 - iota)), name
        assert plan_word(kept[0]) == 3, (name, plan_word(kept[0]))
        kept = []
        gk, _ = gpu_sort(torch_mod, sorter, k, storage_out=kept)
        assert np.array_equal(gk, iota) and plan_word(kept[0]) == 3, name


@pytest.mark.parametrize("n", [ROUND, ROUND + 4097, ROUND + ROUND // 4 + 3, 2 * ROUND - 5, 2 * ROUND])
def test_block_sums_in_sorts_of_one_round(torch_mod, sorter, oracle, n):
    """Sorts of ONE round of 64 ... 256 tiles of 32768 keys and more on the four-pass plan take their prefixes from block
    sums instead of the look-back chain (BlockPrefix in vrdx_kernels.hip): 256 full 1024x32 tiles (keys-only and
    key+value), the two-sub-tile kernel's even-split tiles from 129 tiles up to 256 full ones; direct and indirect with
    a smaller count (whole blocks of 32 tiles past the count).  Values = iota: the permutation itself."""
    k, _ = oracle.generate(17, n, 32)
    # (uniform keys of these sizes take the MSD plan recorded in front of the passes: one value of the top ELEVEN bits
    # occurring 40000 times -- more than either bucket capacity -- sends the sort down its four passes, which is where the
    # block sums are)
    k[:: max(1, n // 40000)][:40000] = (k[:: max(1, n // 40000)][:40000] & np.uint32(0x001FFFFF)) | np.uint32(0x0AB << 23)
    iota = np.arange(n, dtype=np.uint32)
    kept = []
    ek, _, _ = oracle.sort(k)
    gk, _ = gpu_sort(torch_mod, sorter, k, storage_out=kept)
    assert np.array_equal(gk, ek)
    assert int(kept[0][4:8].cpu().numpy().view(np.uint32)[0]) != 3   # the four passes ran, not the plan in front of them
    check_against_oracle(torch_mod, sorter, oracle, k, iota)
    count = n - n // 3 - 7
    check_against_oracle(torch_mod, sorter, oracle, k, count=count, indirect=True, max_count=n)
    check_against_oracle(torch_mod, sorter, oracle, k, iota, count=count, indirect=True, max_count=n)


@pytest.mark.parametrize("n", [20_000, 70_001, 3_000_001, 9_000_001, (1 << 24) + 5])
def test_two_valued_bytes_take_the_ballot_ranking(torch_mod, sorter, oracle, n):
    """Slots whose 64 digits are ONE or TWO values are ranked with ballots instead of a 32- or 64-way same-address
    atomic (RankAtomic, vrdx_kernels.hip): small signed integers (bytes 1..3 are 0x00 | 0xFF, mixed lane by lane),
    dense sorted keys (the top pass sees k and k + 2^24 side by side) and a byte that alternates between two values
    with a run of a third one in between (slots of three digits fall back to the atomic in the middle of a chunk).
    Keys-only and key+value (values = iota: the permutation itself), in every size regime incl. the hybrid plan."""
    rng = np.random.default_rng(n)
    iota = np.arange(n, dtype=np.uint32)
    small_signed = rng.integers(-1000, 1000, size=n, dtype=np.int64).astype(np.int32).view(np.uint32)
    dense_sorted = (np.arange(n, dtype=np.uint64) // 3).astype(np.uint32)          # every key three times, ascending
    i = np.arange(n, dtype=np.uint64)
    mixed = (((i & 1) * 0xFF) << 24) | ((i % 4099 == 0) * np.uint64(0x7F0000)) | (rng.integers(0, 1 << 16, size=n, dtype=np.uint64))
    # slots of three and four digit values (round 5 tried a ballot form for up to four groups and did not adopt it: they
    # take the returning atomic, 16- to 32-way on as many counters -- correct, only slower): four distinct keys, and a byte of
    # three values under random low bits
    four = np.array([3, 0xFFFFFFFF, 0x00010000, 0x7F000000], np.uint32)[rng.integers(0, 4, n)]
    three = (np.array([0x11, 0x80, 0xFE], np.uint32)[rng.integers(0, 3, n)] << np.uint32(16)) | rng.integers(0, 1 << 8, size=n, dtype=np.uint64).astype(np.uint32)
    for k in (small_signed, dense_sorted, mixed.astype(np.uint32), four, three):
        ek, ep, _ = oracle.sort(k, iota)
        gk, _ = gpu_sort(torch_mod, sorter, k)
        assert np.array_equal(gk, ek)
        gk, gp = gpu_sort(torch_mod, sorter, k, iota)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep)


# one size inside every regime of the size-adaptive tile selection (ConfigIndex in vrdx_api.cpp; f =
# N / (256 CUs * 32768)): 1024x8 | 1024x16 | 1024x32 | two-sub-tile 1024x32x2 | 1024x16 between
# rounds | ... -- all ragged (odd) sizes
BREAK_POINT_SIZES = [int(f * (1 << 23)) + 12345 for f in (0.10, 0.20, 0.40, 0.58, 0.80, 1.2, 1.6, 1.99, 2.3, 2.8, 3.3)] + \
                    [(1 << 24), (1 << 24) + 1, (1 << 23) + 1]


@pytest.mark.parametrize("n", BREAK_POINT_SIZES)
def test_every_size_regime_matches_oracle(torch_mod, sorter, oracle, n):
    """Keys-only and key+value (values = iota: the permutation itself) are bit-exact in every regime
    of the tile-geometry selection, including the two-sub-tile kernel (8.4 M < N <= 16.8 M keys-only)."""
    k, _ = oracle.generate(3, n, 32)
    iota = np.arange(n, dtype=np.uint32)
    ek, ep, _ = oracle.sort(k, iota)
    gk, _ = gpu_sort(torch_mod, sorter, k)
    assert np.array_equal(gk, ek)
    gk, gp = gpu_sort(torch_mod, sorter, k, iota)
    assert np.array_equal(gk, ek) and np.array_equal(gp, ep)


# Key+value sorts of 2^24 < N <= 3 * 2^24 elements read their tiles with non-temporal loads (a second
# copy of the load sequence in the key+value kernels, vrdx_kernels.hip StreamingLoads): both edges of the
# window from both sides, ragged sizes inside it, and an indirect sort whose host-side bound is above the
# window while the device-side count is inside it (the kernel decides from the count it sorts).
@pytest.mark.parametrize("n", [(1 << 24) + 1, (1 << 24) + 32769 + 77, 3 << 24, (3 << 24) + 1])
def test_key_value_streaming_load_window(torch_mod, sorter, oracle, n):
    k, _ = oracle.generate(21, n, 32)
    iota = np.arange(n, dtype=np.uint32)
    ek, ep, _ = oracle.sort(k, iota)
    gk, gp = gpu_sort(torch_mod, sorter, k, iota)
    assert np.array_equal(gk, ek) and np.array_equal(gp, ep)
    if n == (3 << 24) + 1:
        count = 20_000_003
        ek, ep, _ = oracle.sort(k, iota, count=count)
        gk, gp = gpu_sort(torch_mod, sorter, k, iota, count=count, indirect=True, max_count=n)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep)


@pytest.mark.parametrize("n", [70_001, 3_000_001, 9_000_001, (1 << 24) + 5])
def test_constant_digit_passes_are_copied_correctly(torch_mod, sorter, oracle, n):
    """A pass whose digit is the same for every key is the identity permutation; the kernels detect it
    from the global histogram and copy the tile instead of ranking it.  16-bit keys (two such
    passes), keys with a constant low byte, a constant middle byte, and all-equal keys, in every
    tile-geometry regime (values = iota: the permutation itself)."""
    rng = np.random.default_rng(n)
    r = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    iota = np.arange(n, dtype=np.uint32)
    for name, k in (("16-bit", r & np.uint32(0xFFFF)),
                    ("low byte constant", (r & np.uint32(0xFFFFFF00)) | np.uint32(0x5A)),
                    ("middle byte constant", (r & np.uint32(0xFF00FFFF)) | np.uint32(0x00C30000)),
                    ("all equal", np.full(n, 0xDEADBEEF, np.uint32))):
        ek, ep, _ = oracle.sort(k, iota)
        gk, _ = gpu_sort(torch_mod, sorter, k)
        assert np.array_equal(gk, ek), name
        gk, gp = gpu_sort(torch_mod, sorter, k, iota)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), name


@pytest.mark.parametrize("n", [70_001, 9_000_001])
def test_every_combination_of_constant_bytes(torch_mod, sorter, oracle, n):
    """All 16 combinations of constant key bytes: whichever passes are skipped (and whichever single
    pass copies, when the number of ranking passes is odd), the result must end in the caller's
    buffers, sorted and stable.  n = 9 000 001 runs the two-sub-tile kernel for keys-only."""
    rng = np.random.default_rng(1000 + n)
    r = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    iota = np.arange(n, dtype=np.uint32)
    for mask in range(16):
        keep = np.uint32(sum(0xFF << (8 * b) for b in range(4) if not (mask >> b) & 1))
        const = np.uint32(sum((0x11 * (b + 3)) << (8 * b) for b in range(4) if (mask >> b) & 1))
        k = (r & keep) | const
        ek, ep, _ = oracle.sort(k, iota)
        gk, _ = gpu_sort(torch_mod, sorter, k)
        assert np.array_equal(gk, ek), mask
        gk, gp = gpu_sort(torch_mod, sorter, k, iota)
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), mask


@pytest.mark.parametrize("seed", [1, 2])
def test_full_size_2pow25_properties_and_golden_checksum(torch_mod, sorter, oracle, golden, seed):
    """BASELINE.json configs[1] and [2]: N = 2^25 uniform-random u32, keys-only and key+value."""
    torch = torch_mod
    n = 1 << 25
    k, v = oracle.generate(seed, n, 32)
    row = [h for h in golden["hashes"] if h["n"] == n and h["seed"] == seed][0]
    assert f"{oracle.hash(k):016x}" == row["input_keys_hash"]
    # keys-only
    gk, _ = gpu_sort(torch, sorter, k)
    assert f"{oracle.hash(gk):016x}" == row["sorted_keys_hash"]
    assert bool(np.all(gk[1:] >= gk[:-1]))
    # key+value with the reference's value stream: committed checksum from the reference CPU sort
    gk, gv = gpu_sort(torch, sorter, k, v)
    assert f"{oracle.hash(gk):016x}" == row["sorted_keys_hash"]
    assert f"{oracle.hash(gv):016x}" == row["sorted_values_hash"]
    # key+value with iota values: the permutation itself.  sorted + stable + is-a-permutation
    iota = np.arange(n, dtype=np.uint32)
    gk, gp = gpu_sort(torch, sorter, k, iota)
    assert bool(np.all(gk[1:] >= gk[:-1]))
    assert np.array_equal(k[gp], gk)                      # values followed their keys
    ties = gk[1:] == gk[:-1]
    assert bool(np.all(gp[1:][ties] > gp[:-1][ties]))     # stability
    assert int(gp.astype(np.uint64).sum()) == n * (n - 1) // 2 and len(np.unique(gp[:: 4096])) == len(gp[:: 4096])
    if seed == 1:
        ek, ep, _ = oracle.sort(k, iota)                  # ~6 s of CPU: the full bit-exact check
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep)


def test_full_size_adversarial(torch_mod, sorter, oracle):
    """BASELINE.json configs[3]: N = 2^25 all-equal / descending / ascending / few-distinct, values = iota.  Every
    pattern is compared ELEMENT BY ELEMENT with what the reference's stable sort yields: for all-equal and ascending keys
    that is the input itself with the identity permutation (no CPU sort needed to know it), for the other two the
    oracle's output."""
    torch = torch_mod
    n = 1 << 25
    iota = np.arange(n, dtype=np.uint32)
    rng = np.random.default_rng(4)
    for name, k in (("all-equal", np.full(n, 0x12345678, np.uint32)),
                    ("all-sentinel", np.full(n, 0xFFFFFFFF, np.uint32)),
                    ("ascending", iota.copy()),
                    ("descending", (n - 1 - iota).astype(np.uint32)),
                    ("few-distinct", rng.choice(np.array([3, 0xFFFFFFFF, 0x00010000, 0x7F000000], np.uint32), size=n))):
        gk, gp = gpu_sort(torch, sorter, k, iota)
        assert bool(np.all(gk[1:] >= gk[:-1])), name
        assert np.array_equal(k[gp], gk), name
        ties = gk[1:] == gk[:-1]
        assert bool(np.all(gp[1:][ties] > gp[:-1][ties])), name
        assert int(gp.astype(np.uint64).sum()) == n * (n - 1) // 2, name
        if name in ("descending", "few-distinct"):        # ~6 s of CPU each: the full bit-exact check
            ek, ep, _ = oracle.sort(k, iota)
        else:                                             # a stable sort of sorted keys moves nothing
            ek, ep = k, iota
        assert np.array_equal(gk, ek) and np.array_equal(gp, ep), name
        gk, _ = gpu_sort(torch, sorter, k)
        assert np.array_equal(gk, ek), name


def test_large_ragged_2pow28_properties(torch_mod, sorter):
    """N = 2^28 + 12345 keys-only (1 GiB of keys, ragged last tile): index arithmetic far above the
    benchmark size.  Size-independent properties only: sortedness and a multiset checksum."""
    torch = torch_mod
    n = (1 << 28) + 12345
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    keys = torch.randint(-(1 << 31), 1 << 31, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    total_before = int((keys.to(torch.int64) & 0xFFFFFFFF).sum().item())
    xor_before = int(torch.bitwise_xor(keys[: n // 2 * 2].view(-1, 2)[:, 0], keys[: n // 2 * 2].view(-1, 2)[:, 1]).sum().item())
    req = sorter.storage_requirements(n)
    storage = torch.empty(req.size, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sorter.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0)
    torch.cuda.synchronize()
    assert sorter.read_status(stream, storage.data_ptr(), 0) == 0
    u = keys.to(torch.int64) & 0xFFFFFFFF
    assert bool((u[1:] >= u[:-1]).all())
    assert int(u.sum().item()) == total_before
    del u, xor_before


def test_maximum_size_2pow30_minus_4_properties(torch_mod, sorter):
    """The largest count the interface admits: N = 2^30 - 4 keys.  The reference sizes its buffers in
    uint32 arithmetic (src/vk_radix_sort.h.in:105-115): Align(4 * N, 16) = (4N + 15) / 16 * 16 wraps
    from N = 2^30 - 3 on, and the storage requirement (which we reproduce bit for bit) collapses.
    4 GiB of keys + 4 GiB of storage; size-independent properties only, computed on the GPU in
    chunks: sortedness, multiset checksums (sum and sum of squares mod 2^64), failure word."""
    torch = torch_mod
    n = (1 << 30) - 4
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * (1 << 30):
        pytest.skip("needs 24 GiB of free HBM")
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    keys = torch.empty(n, dtype=torch.int32, device="cuda")
    chunk = 1 << 27
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        keys[lo:hi] = torch.randint(-(1 << 31), 1 << 31, (hi - lo,), generator=g, device="cuda",
                                    dtype=torch.int64).to(torch.int32)

    def checksums(t):
        s1 = s2 = 0
        for lo in range(0, n, chunk):
            u = t[lo:min(n, lo + chunk)].to(torch.int64) & 0xFFFFFFFF
            s1 = (s1 + int(u.sum().item())) & ((1 << 64) - 1)
            s2 = (s2 + int((u * u).sum().item())) & ((1 << 64) - 1)   # wraps mod 2^64 like int64 does
        return s1, s2

    before = checksums(keys)
    req = sorter.storage_requirements(n)
    assert req.size >= 4 * n
    storage = torch.empty(req.size, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    sorter.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0)
    torch.cuda.synchronize()
    assert sorter.read_status(stream, storage.data_ptr(), 0) == 0
    assert checksums(keys) == before
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk + 1)          # overlap by one element: boundaries are checked too
        u = keys[lo:hi].to(torch.int64) & 0xFFFFFFFF
        assert bool((u[1:] >= u[:-1]).all()), lo
    # A count BEYOND 2^30 - 4 is clamped to it -- and says so (VRDX_HIP_STATUS_COUNT_CLAMPED, bit 30 of the sorter's word;
    # the reference's uint32 size math wraps there, src/vk_radix_sort.h.in:105-115): indirect with a small device-side
    # count, so that only the bound is out of range and the (already sorted) prefix is what gets sorted.
    import vulkan_radix_sort_amd as vrdx
    assert sorter.read_sorter_status(stream) == 0
    count = torch.tensor([100003, 0, 0, 0], dtype=torch.int32, device="cuda")
    head_before = keys[:100003].clone()
    sorter.cmd_sort_indirect(stream, (1 << 30) + 12345, count.data_ptr(), 0, keys.data_ptr(), 0, storage.data_ptr(), 0)
    torch.cuda.synchronize()
    assert sorter.read_sorter_status(stream) == vrdx.STATUS_COUNT_CLAMPED
    assert sorter.read_sorter_status(stream) == 0                      # reading clears it
    assert bool((keys[:100003] == head_before).all())                  # sorted input stays as it is
    del storage


def test_native_selftest_binary(torch_mod):
    """The same battery from plain C++ (no torch in the process): tests/native/vrdx_selftest."""
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 failures" in r.stdout


def test_native_soak_overlapping_sorts(torch_mod):
    """20 s of randomized sorts (sizes up to 3M, random entropy, keys / key+value) alternating on two
    streams, each checked bit for bit against the oracle: hunts rare cross-workgroup races."""
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    r = subprocess.run([exe, "soak", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_general_path_at_small_sizes(torch_mod):
    """Sorts of <= 16384 elements normally take the single-workgroup kernel; VRDX_SMALL_SORT=0 sends
    them down the general path (clear + histogram + four onesweep passes), which must stay parity-clean
    at those sizes too (1-element sorts, ragged single tiles, indirect counts of 0)."""
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    env = dict(os.environ, VRDX_SMALL_SORT="0")
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n", [1, 255, 4096, 4097, 16383, 16384, 16385])
def test_small_sort_boundary_sizes(torch_mod, sorter, oracle, n):
    """Either side of the single-workgroup kernel's limits (4096: 256-thread form, 16384: 1024-thread
    form, 16385: general path), duplicate-heavy keys (8 significant bits) so that stability is visible,
    direct and indirect."""
    k, _ = oracle.generate(5, n, 8)
    iota = np.arange(n, dtype=np.uint32)
    check_against_oracle(torch_mod, sorter, oracle, k)
    check_against_oracle(torch_mod, sorter, oracle, k, iota)
    check_against_oracle(torch_mod, sorter, oracle, k, iota, count=max(n - 3, 0), indirect=True, max_count=n)


ALL_TILE_CONFIGS = ["1024x8", "1024x16", "1024x32", "1024x32x2"]  # == kTileConfigs in vrdx_kernels.hip


def _selftest(args, **env):
    exe = os.path.join(ROOT, "tests", "native", "vrdx_selftest")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True)
    return subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))


@pytest.mark.parametrize("config", ALL_TILE_CONFIGS)
def test_other_tile_configs(torch_mod, oracle, config):
    """Every compiled tile geometry is parity-clean at every size, not only where it is selected."""
    r = _selftest(["quick"], VRDX_TILE_CONFIG=config)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "keys=" + config in r.stdout


def test_version_string_lists_only_compiled_configs():
    """An unknown (e.g. pruned) geometry is refused loudly and the size-adaptive selection is used."""
    r = _selftest(["quick"], VRDX_TILE_CONFIG="512x16")
    assert r.returncode == 0 and "unknown VRDX_TILE_CONFIG" in r.stderr and "(size-adaptive)" in r.stdout


# ---- the ballot ranking (VRDX_RANK=ballot): the form that uses only architecturally defined behaviour --------
# vrdxCreateSorter selects it when the device check of the one-atomic ranking fails; it never does on an
# MI355X, so these tests force it.  Every <..., ATOMIC_RANK = false> instantiation is reached: the single-
# workgroup kernels (small sizes), 1024x8 / 1024x16 / 1024x32 keys-only and key+value, and the two-sub-tile
# kernel through its forced geometry.

@pytest.mark.parametrize("config", [None] + ALL_TILE_CONFIGS)
def test_ballot_ranking_native_battery(config):
    env = {"VRDX_RANK": "ballot"}
    if config is not None:
        env["VRDX_TILE_CONFIG"] = config
    r = _selftest(["quick"], **env)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.fixture(scope="module")
def ballot_sorter(torch_mod):
    import vulkan_radix_sort_amd as vrdx
    old = os.environ.get("VRDX_RANK")
    os.environ["VRDX_RANK"] = "ballot"  # read by vrdxCreateSorter
    try:
        s = vrdx.Sorter()
    finally:
        if old is None:
            del os.environ["VRDX_RANK"]
        else:
            os.environ["VRDX_RANK"] = old
    yield s
    s.destroy()


@pytest.mark.parametrize("n", [1, 255, 4096, 4097, 16383, 16384, 16385])
def test_ballot_ranking_small_sort_boundary_sizes(torch_mod, ballot_sorter, oracle, n):
    k, _ = oracle.generate(5, n, 8)
    iota = np.arange(n, dtype=np.uint32)
    check_against_oracle(torch_mod, ballot_sorter, oracle, k)
    check_against_oracle(torch_mod, ballot_sorter, oracle, k, iota)
    check_against_oracle(torch_mod, ballot_sorter, oracle, k, iota, count=max(n - 3, 0), indirect=True, max_count=n)


@pytest.mark.parametrize("n", [int(0.10 * (1 << 23)) + 12345, int(0.40 * (1 << 23)) + 12345, (1 << 23) + 1, (1 << 24) + 1])
def test_ballot_ranking_every_size_regime(torch_mod, ballot_sorter, oracle, n):
    """1024x8, 1024x16, 1024x32 and (where the atomic form would take the two-sub-tile kernel) 1024x32 again."""
    k, _ = oracle.generate(3, n, 32)
    iota = np.arange(n, dtype=np.uint32)
    ek, ep, _ = oracle.sort(k, iota)
    gk, _ = gpu_sort(torch_mod, ballot_sorter, k)
    assert np.array_equal(gk, ek)
    gk, gp = gpu_sort(torch_mod, ballot_sorter, k, iota)
    assert np.array_equal(gk, ek) and np.array_equal(gp, ep)


# ---- a7: the fused histogram table itself (not only sorts that come out right) ---------------------------------

def _histogram_table_after_sort(torch, sorter, keys, count=None, indirect=False, values=None):
    n_buf = len(keys)
    n = n_buf if count is None else count
    dk = _u32_to_dev(torch, keys)
    dv = _u32_to_dev(torch, values) if values is not None else None
    req = sorter.key_value_storage_requirements(n_buf) if values is not None else sorter.storage_requirements(n_buf)
    storage = torch.full((req.size,), 0xA5, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if indirect:
        dcount = _u32_to_dev(torch, np.array([n, 0, 0, 0], dtype=np.uint32))
        sorter.cmd_sort_indirect(stream, n_buf, dcount.data_ptr(), 0, dk.data_ptr(), 0, storage.data_ptr(), 0)
    elif values is not None:
        sorter.cmd_sort_key_value(stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0)
    else:
        sorter.cmd_sort(stream, n, dk.data_ptr(), 0, storage.data_ptr(), 0)
    torch.cuda.synchronize()
    # uint32[4][256] at byte 16 of the storage: where the reference keeps globalHistogram
    # (src/vk_radix_sort.h.in:405-406); raw counts of the valid keys here
    return storage[16:16 + 4096].cpu().numpy().view(np.uint32).reshape(4, 256), _to_u32(dk)


@pytest.mark.parametrize("n", [16385, 100_003, (1 << 20) + 7, 5_000_011, (1 << 24) + 3])
def test_histogram_table_matches_oracle_digit_counts(torch_mod, sorter, oracle, n):
    """upsweep on its own: the 4 x 256 table the fused histogram kernel leaves in the storage
    equals the oracle's digit counts of the n valid keys -- uniform, all-equal (every pass trivial: the table
    is the only thing those sorts compute), few-distinct, and with a device-side count below the bound."""
    k, v = oracle.generate(9, n, 32)
    table, _ = _histogram_table_after_sort(torch_mod, sorter, k)
    assert np.array_equal(table, oracle.digit_counts(k))
    table, _ = _histogram_table_after_sort(torch_mod, sorter, k, values=v)
    assert np.array_equal(table, oracle.digit_counts(k))
    equal = np.full(n, 0x12345678, np.uint32)
    table, gk = _histogram_table_after_sort(torch_mod, sorter, equal)
    assert np.array_equal(table, oracle.digit_counts(equal)) and np.array_equal(gk, equal)
    few = np.array([3, 0xFFFFFFFF, 0x00010000, 0x7F000000], np.uint32)[np.random.default_rng(n).integers(0, 4, n)]
    table, _ = _histogram_table_after_sort(torch_mod, sorter, few)
    assert np.array_equal(table, oracle.digit_counts(few))
    # keys under a common prefix (the MSD plan's histogram kernel then counts byte 3 in a table of its own), also with
    # a key that breaks the prefix and sends the sort to the four passes, which need all four tables
    narrow = (k >> np.uint32(7)) | np.uint32(0x04000000)
    table, _ = _histogram_table_after_sort(torch_mod, sorter, narrow)
    assert np.array_equal(table, oracle.digit_counts(narrow))
    narrow[n // 2 + 1] = 0xF0000001
    table, gk = _histogram_table_after_sort(torch_mod, sorter, narrow)
    assert np.array_equal(table, oracle.digit_counts(narrow)) and np.array_equal(gk, np.sort(narrow))
    count = n - n // 3
    table, gk = _histogram_table_after_sort(torch_mod, sorter, k, count=count, indirect=True)
    assert np.array_equal(table, oracle.digit_counts(k, count))
    assert np.array_equal(gk[:count], oracle.sort(k, count=count)[0][:count]) and np.array_equal(gk[count:], k[count:])


def test_sorter_status_is_sticky_across_sorts_sharing_one_storage(torch_mod, sorter, oracle):
    """vrdxHipReadSorterStatus: one word for every sort recorded with the sorter, read (and cleared) once at the
    end -- what a batch through ONE storage buffer checks, since each recorded sort clears the storage's own word."""
    torch = torch_mod
    stream = torch.cuda.current_stream().cuda_stream
    assert sorter.read_sorter_status(stream) == 0
    n = 300_000
    req = sorter.key_value_storage_requirements(n)
    storage = torch.full((req.size,), 0xA5, dtype=torch.uint8, device="cuda")
    expected, buffers = [], []
    for seed in range(4):
        k, v = oracle.generate(20 + seed, n - 1000 * seed, 32)
        dk, dv = _u32_to_dev(torch, k), _u32_to_dev(torch, v)
        sorter.cmd_sort_key_value(stream, len(k), dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0)
        buffers.append((dk, dv))
        expected.append(oracle.sort(k, v))
    assert sorter.read_sorter_status(stream) == 0
    for (dk, dv), (ek, ev, _) in zip(buffers, expected):
        assert np.array_equal(_to_u32(dk), ek) and np.array_equal(_to_u32(dv), ev)


def test_recheck_and_event_overhead_helpers(torch_mod, oracle):
    """vrdxHipRecheck repeats the device check behind the one-atomic ranking (and leaves a healthy sorter as it was);
    vrdxHipEventOverheadNs calibrates what a pair of event records adds to the kernel between them, against a kernel that
    times itself with the device's wall clock: a few microseconds, never zero, never the 40 us the kernel itself runs."""
    import vulkan_radix_sort_amd as vrdx
    s = vrdx.Sorter()
    s.recheck()
    k, v = oracle.generate(3, 300_001, 32)
    ek, ev, _ = oracle.sort(k, v)
    gk, gv = gpu_sort(torch_mod, s, k, v)
    assert np.array_equal(gk, ek) and np.array_equal(gv, ev)
    stream = torch_mod.cuda.current_stream().cuda_stream
    ns = vrdx.event_overhead_ns(stream)
    assert 200 < ns < 20_000, ns
    assert s.read_sorter_status(stream) == 0
    s.destroy()


def test_destroying_a_sorter_with_unread_failures_says_so():
    """The entry points return void like the reference's: a caller that never asks vrdxHipReadSorterStatus still learns
    that a sort gave up a look-back -- vrdxDestroySorter prints one line on stderr (test build: a spin limit of 0 and a
    delayed tile 0 make tile 1 give up, deterministically), and nothing when the status was read or clean."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np, torch\n"
        "import vulkan_radix_sort_amd as vrdx\n"
        "s = vrdx.Sorter(0); st = torch.cuda.current_stream().cuda_stream; n = 1 << 24\n"
        "keys = torch.from_numpy(np.random.default_rng(1).integers(0, 2**32, n, dtype=np.uint32).view(np.int32)).cuda()\n"
        "storage = torch.empty(s.storage_requirements(n).size, dtype=torch.uint8, device='cuda')\n"
        "s.cmd_sort(st, n, keys.data_ptr(), 0, storage.data_ptr(), 0); torch.cuda.synchronize()\n"
        "if sys.argv[1] == 'read': print('status 0x%x' % s.read_sorter_status(st))\n"
        "s.destroy()\n")
    # (VRDX_MSD=0, VRDX_BLOCK_SUMS=0: the classic look-back is the path with the spin; see tests/sticky_status_check.py)
    env = dict(os.environ, VRDX_TEST_SPIN_LIMIT="0", VRDX_MSD="0", VRDX_BLOCK_SUMS="0",
               VRDX_LIBRARY=os.path.join(ROOT, "build", "testing", "libvrdx_hip.so"))
    unread = subprocess.run([sys.executable, "-c", code, "unread"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert unread.returncode == 0, unread.stderr[-2000:]
    assert "sorter destroyed with unreported failures" in unread.stderr and "look-back spin expired" in unread.stderr
    read = subprocess.run([sys.executable, "-c", code, "read"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert read.returncode == 0 and "status 0x1" in read.stdout, read.stdout + read.stderr[-2000:]
    assert "sorter destroyed with unreported failures" not in read.stderr
