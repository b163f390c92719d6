"""The host half of vrdxHipReadSorterStatus: an enqueue the runtime refuses is latched in the sorter and
reported once as bit 31.  A real refusal cannot be provoked safely (a destroyed stream crashes the runtime,
a bad pointer would fault the GPU), so the TEST BUILD of the library (make -C vulkan_radix_sort_amd/csrc testing,
-DVRDX_TESTING; the product has no such hook) has VRDX_TEST_INJECT_ENQUEUE_ERROR, which makes every
check report one while the work is enqueued as usual.  Run in a process of its own by tests/test_sort_gpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VRDX_TEST_INJECT_ENQUEUE_ERROR"] = "1"
os.environ["VRDX_LIBRARY"] = os.path.join(ROOT, "build", "testing", "libvrdx_hip.so")  # the -DVRDX_TESTING build
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vulkan_radix_sort_amd as vrdx

torch.cuda.set_device(0)
s = vrdx.Sorter(0)
stream = torch.cuda.current_stream().cuda_stream
n = 100000
host = np.random.default_rng(5).integers(0, 2**32, n, dtype=np.uint32)
keys = torch.from_numpy(host.view(np.int32).copy()).cuda()
storage = torch.empty(s.storage_requirements(n).size, dtype=torch.uint8, device="cuda")
assert s.read_sorter_status(stream) == 0
s.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0)
first = s.read_sorter_status(stream)
second = s.read_sorter_status(stream)
sorted_ok = bool(np.array_equal(keys.cpu().numpy().view(np.uint32), np.sort(host)))
print("status after a sort with refused enqueues injected: 0x%08x, then 0x%08x; sorted: %s" % (first, second, sorted_ok))
sys.exit(0 if first == 0x80000000 and second == 0 and sorted_ok else 1)
