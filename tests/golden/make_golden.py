#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json by running the REFERENCE's own CPU backend.

Run in the authoring container only (needs /root/reference, via oracle/_ref/libvrdx_ref.so built by
`make -C oracle`).  Inputs come from the reference's DataGenerator(seed).Generate(n, bits)
(bench/data_generator.cc:8,12-26); expected outputs from CpuBenchmark::Sort / SortKeyValue
(bench/cpu_benchmark.cc:19-53) -- the pair the reference's one correctness check uses
(bench/bench.cc:41-64).  Fixtures are data only: inputs, expected outputs, FNV-1a-64 checksums.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import load_oracle, load_reference  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

FULL_VECTOR_CASES = (
    [(42, n, 32) for n in (0, 1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097)]
    + [(seed, n, bits) for seed in (1, 7) for n in (513, 4097) for bits in (8, 4, 0)]
)
HASH_CASES = (
    [(seed, n, 32) for seed in (1, 7, 42) for n in (8191, 8193, 12411, 16385, 65539, 1 << 18, (1 << 20) + 7)]
    + [(seed, n, bits) for seed in (1, 7) for n in (65539, 1 << 18) for bits in (8, 4, 0)]
    + [(seed, 1 << 25, 32) for seed in (1, 2)]
)


def main():
    ref = load_reference()
    if ref is None:
        raise SystemExit("oracle/_ref/libvrdx_ref.so missing: run `make -C oracle` where /root/reference exists")
    orc = load_oracle()

    arrays = {}
    index = []
    for seed, n, bits in FULL_VECTOR_CASES:
        k, v = ref.generate(seed, n, bits)
        sk, _ = ref.sort_keys(k)
        kk, kv, _ = ref.sort_key_value(k, v)
        assert (sk == kk).all()
        tag = f"s{seed}_n{n}_b{bits}"
        arrays[tag + "_keys"] = k
        arrays[tag + "_values"] = v
        arrays[tag + "_sorted_keys"] = kk
        arrays[tag + "_sorted_values"] = kv
        index.append({"seed": seed, "n": n, "bits": bits, "tag": tag})
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **arrays)

    hashes = []
    for seed, n, bits in HASH_CASES:
        k, v = ref.generate(seed, n, bits)
        kk, kv, _ = ref.sort_key_value(k, v)
        sk, _ = ref.sort_keys(k)
        assert (sk == kk).all()
        hashes.append({
            "seed": seed, "n": n, "bits": bits,
            "first_keys": [int(x) for x in k[:4]], "first_values": [int(x) for x in v[:4]],
            "input_keys_hash": f"{orc.hash(k):016x}", "input_values_hash": f"{orc.hash(v):016x}",
            "sorted_keys_hash": f"{orc.hash(kk):016x}", "sorted_values_hash": f"{orc.hash(kv):016x}",
        })
        print("hashed", seed, n, bits, flush=True)

    # storage-size table from the reference formulas (src/vk_radix_sort.h.in:279-308); the values
    # are also listed in SURVEY.md section 8(a3).  Computed here by the oracle restatement and
    # cross-checked against the survey's hand-derived numbers below.
    survey = {0: (4128, 4128), 1: (5168, 5184), 4096: (21536, 37920), 4097: (22576, 38976),
              1 << 18: (1118240, 2166816), (1 << 18) + 1: (1119280, 2167872), 1 << 25: (142610464, 276828192)}
    storage = []
    for n in sorted(set(list(survey) + [2, 63, 4095, 8192, 8193, 12411, 16384, 16385, 65539, 1000000,
                                        (1 << 20) + 7, 1 << 24, (1 << 30) - 1])):
        ko, kvs = orc.storage_size(n, False), orc.storage_size(n, True)
        if n in survey:
            assert (ko, kvs) == survey[n], (n, ko, kvs)
        storage.append({"n": n, "keys": ko, "key_value": kvs})

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "source": "reference bench/cpu_benchmark.cc + bench/data_generator.cc via oracle/_ref",
                   "hash": "FNV-1a 64 over little-endian uint32 words",
                   "vectors": index, "hashes": hashes, "storage_align16": storage, "usage": 0x22}, f, indent=1)
    print("wrote", len(index), "vector cases,", len(hashes), "hash cases")


if __name__ == "__main__":
    main()
