"""vulkan_radix_sort_amd -- MI355X (gfx950) HIP backend behind the VrdxSorter / vrdxCmdSort* API.

The product is the C-ABI shared library ``libvrdx_hip.so`` (sources in ``csrc/``, public header
``include/vk_radix_sort.h``).  This package is the thin Python host-side mirror of that interface
used by the tests and by ``bench.py``: same names, same argument meaning, same error behaviour as
the reference's ``include/vk_radix_sort.h`` (see :mod:`vulkan_radix_sort_amd.api`).

There is no CPU fallback: importing :mod:`vulkan_radix_sort_amd.api` raises if the HIP library has
not been built (``make -C vulkan_radix_sort_amd/csrc`` or ``__graft_entry__.build()``).
"""
from .api import (  # noqa: F401
    VK_SUCCESS,
    VrdxError,
    VrdxSorterStorageRequirements,
    Sorter,
    QueryPool,
    library_path,
    load_library,
    EXPORTED_SYMBOLS,
    version_string,
    event_overhead_ns,
    VrdxHipPlanInfo,
    PLAN_NAMES,
    STATUS_LOOKBACK_GAVE_UP,
    STATUS_RANK_ORDER,
    STATUS_COUNT_CLAMPED,
    STATUS_ENQUEUE_REFUSED,
    VERDICT_NONE,
    VERDICT_HYBRID8_RUNS,
    VERDICT_HYBRID8_DECLINED,
    VERDICT_MSD_RUNS,
    VERDICT_MSD_SORTED,
)

__all__ = [
    "VK_SUCCESS",
    "VrdxError",
    "VrdxSorterStorageRequirements",
    "Sorter",
    "QueryPool",
    "library_path",
    "load_library",
    "EXPORTED_SYMBOLS",
    "version_string",
    "event_overhead_ns",
    "VrdxHipPlanInfo",
    "PLAN_NAMES",
    "STATUS_LOOKBACK_GAVE_UP",
    "STATUS_RANK_ORDER",
    "STATUS_COUNT_CLAMPED",
    "STATUS_ENQUEUE_REFUSED",
    "VERDICT_NONE",
    "VERDICT_HYBRID8_RUNS",
    "VERDICT_HYBRID8_DECLINED",
    "VERDICT_MSD_RUNS",
    "VERDICT_MSD_SORTED",
]
