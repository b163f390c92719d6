"""CPU-only: pins the oracle (oracle/vrdx_oracle.c + oracle/cpu_sort.cc).

1. against the committed golden vectors, which were produced by the REFERENCE's own CPU backend
   (tests/golden/make_golden.py -> bench/cpu_benchmark.cc, bench/data_generator.cc);
2. against oracle/_ref itself wherever that library exists (authoring container, or shipped to
   the GPU box), on randomized inputs;
3. against the storage-size table of SURVEY.md section 8(a3).
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_storage_sizes_match_golden_table(oracle, golden):
    assert oracle.storage_usage() == golden["usage"] == 0x22
    for row in golden["storage_align16"]:
        assert oracle.storage_size(row["n"], False) == row["keys"], row
        assert oracle.storage_size(row["n"], True) == row["key_value"], row


def test_storage_known_answers(oracle):
    # SURVEY.md section 8(a3), derived by hand from src/vk_radix_sort.h.in:279-308 with A = 16
    table = {0: (4128, 4128), 1: (5168, 5184), 4096: (21536, 37920), 4097: (22576, 38976),
             1 << 18: (1118240, 2166816), (1 << 18) + 1: (1119280, 2167872),
             1 << 25: (142610464, 276828192)}
    for n, (keys, kv) in table.items():
        assert oracle.storage_size(n, False) == keys
        assert oracle.storage_size(n, True) == kv


def test_storage_offsets(oracle):
    # N = 2^25 -> hist @16, partition hist @4112, keys scratch @8392736,
    # values scratch @142610464
    o = oracle.storage_offsets(1 << 25)
    assert o == {"count": 0, "histogram": 16, "partition_histogram": 4112, "inout": 8392736,
                 "values_inout": 142610464, "partitions": 8192}


def test_generator_matches_golden(oracle, golden):
    for h in golden["hashes"]:
        if h["n"] > (1 << 20) + 7:
            continue
        k, v = oracle.generate(h["seed"], h["n"], h["bits"])
        assert [int(x) for x in k[:4]] == h["first_keys"]
        assert [int(x) for x in v[:4]] == h["first_values"]
        assert f"{oracle.hash(k):016x}" == h["input_keys_hash"]
        assert f"{oracle.hash(v):016x}" == h["input_values_hash"]
    # raw mt19937(42) stream, SURVEY.md section 8(c)
    k, _ = oracle.generate(42, 3, 32)
    assert [int(x) for x in k] == [1608637542, 3421126067, 4083286876]


def test_oracle_sort_matches_golden_vectors(oracle, golden):
    arrays = golden["arrays"]
    for case in golden["vectors"]:
        tag = case["tag"]
        k, v = arrays[tag + "_keys"], arrays[tag + "_values"]
        gk, gv = oracle.generate(case["seed"], case["n"], case["bits"])
        assert np.array_equal(gk, k) and np.array_equal(gv, v), tag
        sk, sv, _ = oracle.sort(k, v)
        assert np.array_equal(sk, arrays[tag + "_sorted_keys"]), tag
        assert np.array_equal(sv, arrays[tag + "_sorted_values"]), tag
        only_keys, _, _ = oracle.sort(k)
        assert np.array_equal(only_keys, arrays[tag + "_sorted_keys"]), tag
        pk, _ = oracle.port_sort_keys(k)
        pkk, pkv, _ = oracle.port_sort_key_value(k, v)
        assert np.array_equal(pk, sk) and np.array_equal(pkk, sk) and np.array_equal(pkv, sv), tag


def test_oracle_sort_matches_golden_hashes(oracle, golden):
    for h in golden["hashes"]:
        if h["n"] > (1 << 20) + 7:
            continue  # the 2^25 rows are checked on the GPU box (tests/test_sort_gpu.py)
        k, v = oracle.generate(h["seed"], h["n"], h["bits"])
        sk, sv, _ = oracle.sort(k, v)
        assert f"{oracle.hash(sk):016x}" == h["sorted_keys_hash"], h
        assert f"{oracle.hash(sv):016x}" == h["sorted_values_hash"], h


def test_oracle_is_a_stable_sort(oracle):
    rng = np.random.default_rng(5)
    for n in (0, 1, 5, 4096, 4097, 10000, 70001):
        for bits in (1, 3, 8, 17, 32):
            k = rng.integers(0, 1 << bits, size=n, dtype=np.uint64).astype(np.uint32)
            v = np.arange(n, dtype=np.uint32)
            sk, sv, _ = oracle.sort(k, v)
            order = np.argsort(k, kind="stable")
            assert np.array_equal(sk, k[order]) and np.array_equal(sv, v[order].astype(np.uint32))


def test_oracle_count_prefix_and_untouched_tail(oracle):
    # indirect semantics: only the first `count` elements take part; the rest is not touched
    rng = np.random.default_rng(11)
    k = rng.integers(0, 1 << 32, size=9000, dtype=np.uint64).astype(np.uint32)
    v = rng.integers(0, 1 << 32, size=9000, dtype=np.uint64).astype(np.uint32)
    sk, sv, _ = oracle.sort(k, v, count=5001)
    order = np.argsort(k[:5001], kind="stable")
    assert np.array_equal(sk[:5001], k[:5001][order]) and np.array_equal(sv[:5001], v[:5001][order])
    assert np.array_equal(sk[5001:], k[5001:]) and np.array_equal(sv[5001:], v[5001:])


def test_oracle_sentinel_keys_and_histogram(oracle):
    # 0xFFFFFFFF is the reference's padding value (upsweep.slang:32): real keys equal to it must
    # still be placed, and the global histogram the reference leaves behind is the exclusive scan
    # of the digit counts INCLUDING the padding of the ragged last partition.
    n = 4096 + 100
    k = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    k[::3] = 7
    v = np.arange(n, dtype=np.uint32)
    sk, sv, hist = oracle.sort(k, v)
    order = np.argsort(k, kind="stable")
    assert np.array_equal(sk, k[order]) and np.array_equal(sv, v[order].astype(np.uint32))
    counts = oracle.digit_counts(k)
    assert counts.sum(axis=1).tolist() == [n] * 4
    pad = 2 * 4096 - n
    for p in range(4):
        padded = counts[p].astype(np.int64)
        padded[255] += pad
        excl = np.concatenate([[0], np.cumsum(padded)[:-1]])
        assert np.array_equal(hist[p].astype(np.int64), excl)


def test_oracle_matches_reference_build(oracle, reference):
    if reference is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(2)
    for seed, n, bits in [(3, 0, 32), (3, 1, 32), (9, 4097, 32), (9, 100003, 32), (4, 65539, 6), (4, 300000, 32)]:
        k, v = oracle.generate(seed, n, bits)
        rk, rv = reference.generate(seed, n, bits)
        assert np.array_equal(k, rk) and np.array_equal(v, rv)
        sk, sv, _ = oracle.sort(k, v)
        rsk, rsv, _ = reference.sort_key_value(k, v)
        assert np.array_equal(sk, rsk) and np.array_equal(sv, rsv)
        assert np.array_equal(oracle.sort(k)[0], reference.sort_keys(k)[0])
    # adversarial inputs
    n = 50000
    for k in (np.full(n, 0x12345678, np.uint32), np.full(n, 0xFFFFFFFF, np.uint32),
              np.arange(n, dtype=np.uint32)[::-1].copy(),
              rng.choice(np.array([0, 1, 0x80000000, 0xFFFFFFFF], np.uint32), size=n)):
        v = np.arange(n, dtype=np.uint32)
        sk, sv, _ = oracle.sort(k, v)
        rsk, rsv, _ = reference.sort_key_value(k, v)
        assert np.array_equal(sk, rsk) and np.array_equal(sv, rsv)


def test_bench_input_stream_is_the_reference_generator(oracle, reference):
    """bench.py generates its inputs itself (it may touch the oracle only in its cpu_baseline leg): its stream must be
    the reference generator's -- keys = the first N raw mt19937(seed) outputs, values = the next N
    (bench/data_generator.cc:20-25)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for seed, n in ((1, 1000), (42, 4097), (7, 65539)):
        k, v = bench.reference_stream(seed, n)
        ok, ov = oracle.generate(seed, n, 32)
        assert k.dtype == np.uint32 and np.array_equal(k, ok) and np.array_equal(v, ov)
        if reference is not None:
            rk, rv = reference.generate(seed, n, 32)
            assert np.array_equal(k, rk) and np.array_equal(v, rv)
