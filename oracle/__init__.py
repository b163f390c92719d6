"""CPU oracle for the vrdxCmdSort* hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``vulkan_radix_sort_amd``) never does.

Pieces (all reached through ctypes, see :mod:`oracle.loader`):

* ``liboracle.so``  -- ``vrdx_oracle.c`` (step-by-step restatement of the reference's
  upsweep / spine / downsweep passes and of its storage-size math) and ``cpu_sort.cc`` (our port of
  the reference's ``CpuBenchmark`` + ``DataGenerator``).
* ``_ref/libvrdx_ref.so`` -- the reference's own ``bench/cpu_benchmark.cc`` +
  ``bench/data_generator.cc`` compiled from ``/root/reference`` (``make -C oracle``).
"""
from .loader import Oracle, Reference, load_oracle, load_reference, build  # noqa: F401
