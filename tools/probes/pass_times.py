"""Probe (GPU box): per-pass times of a 2^25 keys-only sort for synthetic key patterns that separate the DIGIT
structure of a pass from the MEMORY pattern of its scatter.   python3 tools/probes/pass_times.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vulkan_radix_sort_amd as vrdx

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 25)
half = n.bit_length() - 2   # i >> half: 0 | 1 by halves
i = torch.arange(n, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
rnd = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device="cuda", generator=g)
lo_sorted = i & 0xFFFFFF
lo_random = rnd & 0xFFFFFF
patterns = {
    "uniform random": rnd,
    "ascending, top byte 0 | 1 by halves": ((i >> half) << 24) | (i & 0xFFFFFF),
    "top byte 0 | 1 by halves, low random": ((i >> half) << 24) | lo_random,
    "top byte = (i >> 16) & 1 (alternating per 65536), low random": (((i >> 16) & 1) << 24) | lo_random,
    "top byte = (i >> 16) & 15 (16 values per 65536), low random": (((i >> 16) & 15) << 24) | lo_random,
    "top byte random from {0, 1}, low random": ((rnd >> 31) << 24) | lo_random,
    "top byte random from 4 values, low random": ((rnd >> 30) << 24) | lo_random,
}
sorter = vrdx.Sorter()
req = sorter.storage_requirements(n)
storage = torch.zeros(req.size, dtype=torch.uint8, device="cuda")
pool = vrdx.QueryPool(15)
stream = torch.cuda.current_stream().cuda_stream
print(vrdx.version_string())
for name, k64 in patterns.items():
    best = None
    for rep in range(3):
        keys = (k64 & 0xFFFFFFFF).to(torch.int64).to(torch.uint32) if hasattr(torch, "uint32") else None
        keys = (k64 & 0xFFFFFFFF).cpu().numpy().astype(np.uint32)
        dk = torch.from_numpy(keys.view(np.int32)).cuda()
        torch.cuda.synchronize()
        sorter.cmd_sort(stream, n, dk.data_ptr(), 0, storage.data_ptr(), 0, pool, 0)
        torch.cuda.synchronize()
        ts = pool.results_ns()
        passes = [(ts[4 + 3 * p] - ts[2 + 3 * p]) / 1e3 for p in range(4)]
        if best is None or sum(passes) < sum(best):
            best = passes
    out = dk.cpu().numpy().view(np.uint32)
    ok = bool(np.all(out[:-1] <= out[1:]))
    print("%-56s passes us: %s  %s" % (name, " ".join("%6.1f" % p for p in best), "ok" if ok else "NOT SORTED"))
