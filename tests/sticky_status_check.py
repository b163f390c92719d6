"""The device half of vrdxHipReadSorterStatus: a look-back that gives up sets the failure word of the storage it
ran on AND the sorter's own sticky word; the next sort recorded on that storage clears the former, only
vrdxHipReadSorterStatus clears the latter.  A real give-up needs 2^18 fruitless trips, so the TEST BUILD of the
library (make -C vulkan_radix_sort_amd/csrc testing, -DVRDX_TESTING) takes the limit from VRDX_TEST_SPIN_LIMIT: with 0,
the first trip that has to wait gives up, and tile 0 of every pass holds its inclusive prefix back for ~0.3 ms
(TestDelayFirstTile in vrdx_kernels.hip), so that tile 1 is certain to wait: ONE sort suffices, deterministically.
Run in a process of its own by tests/test_sort_gpu.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VRDX_TEST_SPIN_LIMIT"] = "0"
# the CLASSIC look-back, the path with the spin: a sort of 2^24 keys would otherwise take the MSD plan (no look-back at all)
# or, with that off, block sums (one round of tiles: nothing is published behind the test's delay of tile 0)
os.environ["VRDX_MSD"] = "0"
os.environ["VRDX_BLOCK_SUMS"] = "0"
os.environ["VRDX_LIBRARY"] = os.path.join(ROOT, "build", "testing", "libvrdx_hip.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vulkan_radix_sort_amd as vrdx

torch.cuda.set_device(0)
s = vrdx.Sorter(0)
stream = torch.cuda.current_stream().cuda_stream
n = 1 << 24
rng = np.random.default_rng(11)
storage = torch.empty(s.storage_requirements(n).size, dtype=torch.uint8, device="cuda")
assert s.read_sorter_status(stream) == 0
keys = torch.from_numpy(rng.integers(0, 2**32, n, dtype=np.uint32).view(np.int32)).cuda()
s.cmd_sort(stream, n, keys.data_ptr(), 0, storage.data_ptr(), 0)   # tile 1 waits for the delayed tile 0 and gives up
torch.cuda.synchronize()
word = s.read_status(stream, storage.data_ptr(), 0)
# a small sort on the SAME storage: single-workgroup path, no look-back, clears the storage's failure word
small = torch.from_numpy(rng.integers(0, 2**32, 1000, dtype=np.uint32).view(np.int32)).cuda()
s.cmd_sort(stream, 1000, small.data_ptr(), 0, storage.data_ptr(), 0)
torch.cuda.synchronize()
after = s.read_status(stream, storage.data_ptr(), 0)
sticky = s.read_sorter_status(stream)
again = s.read_sorter_status(stream)
print("failure word 0x%x after the big sort, 0x%x after the next sort on that storage; sorter status 0x%x, then 0x%x"
      % (word, after, sticky, again))
sys.exit(0 if word == 1 and after == 0 and sticky == 1 and again == 0 else 1)
