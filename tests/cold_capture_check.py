"""Graph capture as the FIRST use of a new sorter in a new process (general path and single-workgroup path),
checked against the oracle.  Run on the GPU box: python tests/cold_capture_check.py (tests/test_sort_gpu.py does)."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkan_radix_sort_amd as vrdx
from oracle import load_oracle
orc = load_oracle()
torch.cuda.set_device(0)
s = vrdx.Sorter(0)
failed = 0
for n in (200000, 5000):
    k, v = orc.generate(31, n, 32)
    dk = torch.from_numpy(k.view(np.int32).copy()).cuda(); dv = torch.from_numpy(v.view(np.int32).copy()).cuda()
    storage = torch.empty(s.key_value_storage_requirements(n).size, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s.cmd_sort_key_value(torch.cuda.current_stream().cuda_stream, n, dk.data_ptr(), 0, dv.data_ptr(), 0, storage.data_ptr(), 0)
    g.replay(); torch.cuda.synchronize()
    ek, ev, _ = orc.sort(k, v)
    ok = np.array_equal(dk.cpu().numpy().view(np.uint32), ek) and np.array_equal(dv.cpu().numpy().view(np.uint32), ev)
    print("cold capture n=%d:" % n, "OK" if ok else "NOT SORTED")
    failed += 0 if ok else 1
sys.exit(failed)
