"""ctypes binding of libvrdx_hip.so -- the eight vrdx* entry points of include/vk_radix_sort.h.

Mirrors the reference interface (src/vk_radix_sort.h.in:11-81 under /root/reference):

=============================================  ==============================================
reference (C++ / Vulkan)                        here
=============================================  ==============================================
``vrdxCreateSorter(&info, &sorter)``            ``Sorter(device=None)`` (raises ``VrdxError``)
``vrdxDestroySorter(sorter)``                   ``Sorter.destroy()`` / context manager
``vrdxGetSorterStorageRequirements``            ``Sorter.storage_requirements(n)``
``vrdxGetSorterKeyValueStorageRequirements``    ``Sorter.key_value_storage_requirements(n)``
``vrdxCmdSort``                                 ``Sorter.cmd_sort(stream, n, keys, keys_off, storage, storage_off, pool, query)``
``vrdxCmdSortIndirect``                         ``Sorter.cmd_sort_indirect(...)``
``vrdxCmdSortKeyValue``                         ``Sorter.cmd_sort_key_value(...)``
``vrdxCmdSortKeyValueIndirect``                 ``Sorter.cmd_sort_key_value_indirect(...)``
=============================================  ==============================================

``VkCommandBuffer`` is a ``hipStream_t`` (an ``int`` handle, e.g. ``torch.cuda.current_stream().cuda_stream``),
``VkBuffer`` is a device address (``tensor.data_ptr()``), offsets are bytes.  Like the reference,
the ``cmd_*`` calls validate nothing and never block the host.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

VK_SUCCESS = 0
VK_NOT_READY = 1
VK_ERROR_OUT_OF_HOST_MEMORY = -1
VK_ERROR_INITIALIZATION_FAILED = -3
VK_ERROR_FEATURE_NOT_PRESENT = -8

# VK_BUFFER_USAGE_STORAGE_BUFFER_BIT | VK_BUFFER_USAGE_TRANSFER_DST_BIT
STORAGE_USAGE = 0x20 | 0x02

EXPORTED_SYMBOLS = (
    # the reference's eight entry points (src/vk_radix_sort.h.in:24-81)
    "vrdxCreateSorter",
    "vrdxDestroySorter",
    "vrdxGetSorterStorageRequirements",
    "vrdxGetSorterKeyValueStorageRequirements",
    "vrdxCmdSort",
    "vrdxCmdSortIndirect",
    "vrdxCmdSortKeyValue",
    "vrdxCmdSortKeyValueIndirect",
    # HIP companions of the Vulkan objects callers create themselves
    "vrdxHipCreateQueryPool",
    "vrdxHipDestroyQueryPool",
    "vrdxHipGetQueryPoolResults",
    "vrdxHipReadStatus",
    "vrdxHipReadSorterStatus",
    "vrdxHipRecheck",
    "vrdxHipEventOverheadNs",
    "vrdxHipDescribePlan",
    "vrdxHipReadPlanVerdict",
    "vrdxHipReadPlanCounters",
    "vrdxHipVersionString",
)

# bits of vrdxHipReadSorterStatus (include/vk_radix_sort.h)
STATUS_LOOKBACK_GAVE_UP = 0x00000001
STATUS_RANK_ORDER = 0x00000002
STATUS_COUNT_CLAMPED = 0x40000000
STATUS_ENQUEUE_REFUSED = 0x80000000

# VRDX_HIP_PLAN_* (include/vk_radix_sort.h)
PLAN_NAMES = {0: "none", 1: "one-workgroup", 2: "four-passes", 3: "hybrid-8", 4: "hybrid-9", 5: "msd"}

# VRDX_HIP_VERDICT_* (include/vk_radix_sort.h): what the DEVICE made of the plan of the last sort on a storage
VERDICT_NONE = 0
VERDICT_HYBRID8_RUNS = 1
VERDICT_HYBRID8_DECLINED = 2
VERDICT_MSD_RUNS = 3
VERDICT_MSD_SORTED = 4


class VrdxError(RuntimeError):
    """A vrdx* call returned a VkResult other than VK_SUCCESS."""

    def __init__(self, what: str, result: int):
        super().__init__(f"{what} failed with VkResult {result}")
        self.result = result


class VrdxSorterCreateInfo(ctypes.Structure):
    # src/vk_radix_sort.h.in:18-22
    _fields_ = [("physicalDevice", ctypes.c_void_p), ("device", ctypes.c_void_p),
                ("pipelineCache", ctypes.c_void_p)]


class VrdxSorterStorageRequirements(ctypes.Structure):
    # src/vk_radix_sort.h.in:28-31
    _fields_ = [("size", ctypes.c_uint64), ("usage", ctypes.c_uint32)]


class VrdxHipPlanInfo(ctypes.Structure):
    # include/vk_radix_sort.h
    _fields_ = [("plan", ctypes.c_uint32), ("bits", ctypes.c_uint32), ("bytesPerElement", ctypes.c_uint32),
                ("fallbackBytesPerElement", ctypes.c_uint32), ("launches", ctypes.c_uint32)]

    @property
    def name(self) -> str:
        return PLAN_NAMES.get(int(self.plan), "?")


def in_tree_library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvrdx_hip.so")


def library_path() -> str:
    """The in-tree libvrdx_hip.so.  VRDX_LIBRARY names another BUILD of the same library instead (the -DVRDX_TESTING
    build of two tests, a tuning variant of tools/); it is never a fallback: a missing file is an error either way,
    and ``load_library`` says on stderr that the override is active and refuses a library of another version."""
    override = os.environ.get("VRDX_LIBRARY")
    if override:
        return os.path.abspath(override)
    return in_tree_library_path()


_LIB: Optional[ctypes.CDLL] = None


def load_library() -> ctypes.CDLL:
    """Loads libvrdx_hip.so.  There is deliberately no fallback of any kind."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C vulkan_radix_sort_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    if path != in_tree_library_path():
        # A stale VRDX_LIBRARY (left behind by a tuning script, or the test build with its injection hooks) silently
        # changes what production code runs: say so, and insist on the same version and ABI.
        import sys
        missing = [name for name in EXPORTED_SYMBOLS if not hasattr(lib, name)]
        if missing:
            raise ImportError(f"VRDX_LIBRARY={path} does not export {', '.join(missing)}: not a build of this library")
        lib.vrdxHipVersionString.restype = ctypes.c_char_p
        theirs = lib.vrdxHipVersionString().decode()
        if os.path.exists(in_tree_library_path()):
            own = ctypes.CDLL(in_tree_library_path())
            own.vrdxHipVersionString.restype = ctypes.c_char_p
            ours = own.vrdxHipVersionString().decode()
            if theirs.split()[:2] != ours.split()[:2]:
                raise ImportError(f"VRDX_LIBRARY={path} is '{theirs}', the in-tree library is '{ours}': version mismatch")
        print(f"vulkan_radix_sort_amd: VRDX_LIBRARY override active, loaded {path} ({theirs})", file=sys.stderr)
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    lib.vrdxCreateSorter.restype = ctypes.c_int32
    lib.vrdxCreateSorter.argtypes = [ctypes.POINTER(VrdxSorterCreateInfo), ctypes.POINTER(vp)]
    lib.vrdxDestroySorter.restype = None
    lib.vrdxDestroySorter.argtypes = [vp]
    for name in ("vrdxGetSorterStorageRequirements", "vrdxGetSorterKeyValueStorageRequirements"):
        fn = getattr(lib, name)
        fn.restype = None
        fn.argtypes = [vp, u32, ctypes.POINTER(VrdxSorterStorageRequirements)]
    lib.vrdxCmdSort.restype = None
    lib.vrdxCmdSort.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u32]
    lib.vrdxCmdSortIndirect.restype = None
    lib.vrdxCmdSortIndirect.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u64, vp, u32]
    lib.vrdxCmdSortKeyValue.restype = None
    lib.vrdxCmdSortKeyValue.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u64, vp, u32]
    lib.vrdxCmdSortKeyValueIndirect.restype = None
    lib.vrdxCmdSortKeyValueIndirect.argtypes = [vp, vp, u32, vp, u64, vp, u64, vp, u64, vp, u64, vp, u32]
    lib.vrdxHipCreateQueryPool.restype = ctypes.c_int32
    lib.vrdxHipCreateQueryPool.argtypes = [u32, ctypes.POINTER(vp)]
    lib.vrdxHipDestroyQueryPool.restype = None
    lib.vrdxHipDestroyQueryPool.argtypes = [vp]
    lib.vrdxHipGetQueryPoolResults.restype = ctypes.c_int32
    lib.vrdxHipGetQueryPoolResults.argtypes = [vp, u32, u32, ctypes.POINTER(u64)]
    lib.vrdxHipReadStatus.restype = u32
    lib.vrdxHipReadStatus.argtypes = [vp, vp, u64]
    lib.vrdxHipReadSorterStatus.restype = u32
    lib.vrdxHipReadSorterStatus.argtypes = [vp, vp]
    lib.vrdxHipRecheck.restype = ctypes.c_int32
    lib.vrdxHipRecheck.argtypes = [vp]
    lib.vrdxHipEventOverheadNs.restype = u64
    lib.vrdxHipEventOverheadNs.argtypes = [vp]
    lib.vrdxHipDescribePlan.restype = None
    lib.vrdxHipDescribePlan.argtypes = [vp, u32, ctypes.c_int, ctypes.POINTER(VrdxHipPlanInfo)]
    lib.vrdxHipReadPlanVerdict.restype = u32
    lib.vrdxHipReadPlanVerdict.argtypes = [vp, vp, u64]
    lib.vrdxHipReadPlanCounters.restype = ctypes.c_int32
    lib.vrdxHipReadPlanCounters.argtypes = [vp, vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    lib.vrdxHipVersionString.restype = ctypes.c_char_p
    lib.vrdxHipVersionString.argtypes = []
    _LIB = lib
    return lib


def version_string() -> str:
    return load_library().vrdxHipVersionString().decode()


def event_overhead_ns(command_buffer) -> int:
    """``vrdxHipEventOverheadNs``: what a pair of event records adds to the kernel between them on this stream (ns),
    measured against a kernel that times itself with the device's wall clock.  Raises if the calibration failed."""
    ns = int(load_library().vrdxHipEventOverheadNs(_handle(command_buffer)))
    if ns == 0xFFFFFFFFFFFFFFFF:
        raise VrdxError("vrdxHipEventOverheadNs", -4)
    return ns


def _handle(x) -> Optional[int]:
    if x is None:
        return None
    return int(x) or None


class QueryPool:
    """HIP stand-in for a VkQueryPool of timestamp queries (hipEvent_t per slot)."""

    def __init__(self, count: int = 15):
        self._lib = load_library()
        h = ctypes.c_void_p()
        r = self._lib.vrdxHipCreateQueryPool(count, ctypes.byref(h))
        if r != VK_SUCCESS:
            raise VrdxError("vrdxHipCreateQueryPool", r)
        self.handle = h.value
        self.count = count

    def results_ns(self, first: int = 0, count: Optional[int] = None):
        """vkGetQueryPoolResults analogue: ns since slot `first` for each slot (stream must be done)."""
        count = self.count - first if count is None else count
        data = (ctypes.c_uint64 * count)()
        r = self._lib.vrdxHipGetQueryPoolResults(self.handle, first, count, data)
        if r != VK_SUCCESS:
            raise VrdxError("vrdxHipGetQueryPoolResults", r)
        return [int(x) for x in data]

    def destroy(self):
        if self.handle:
            self._lib.vrdxHipDestroyQueryPool(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Sorter:
    """VrdxSorter.  ``device`` is a HIP ordinal, or None for the current device."""

    def __init__(self, device: Optional[int] = None):
        self._lib = load_library()
        info = VrdxSorterCreateInfo()
        encoded = None if device is None else device + 1  # VRDX_HIP_DEVICE(ordinal)
        info.physicalDevice = encoded
        info.device = encoded
        info.pipelineCache = None
        h = ctypes.c_void_p()
        r = self._lib.vrdxCreateSorter(ctypes.byref(info), ctypes.byref(h))
        if r != VK_SUCCESS:
            raise VrdxError("vrdxCreateSorter", r)
        self.handle = h.value

    # -- lifetime ---------------------------------------------------------------------------
    def destroy(self):
        if getattr(self, "handle", None):
            self._lib.vrdxDestroySorter(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- storage ----------------------------------------------------------------------------
    def storage_requirements(self, max_element_count: int) -> VrdxSorterStorageRequirements:
        req = VrdxSorterStorageRequirements()
        self._lib.vrdxGetSorterStorageRequirements(self.handle, max_element_count, ctypes.byref(req))
        return req

    def key_value_storage_requirements(self, max_element_count: int) -> VrdxSorterStorageRequirements:
        req = VrdxSorterStorageRequirements()
        self._lib.vrdxGetSorterKeyValueStorageRequirements(self.handle, max_element_count, ctypes.byref(req))
        return req

    # -- recording --------------------------------------------------------------------------
    def cmd_sort(self, command_buffer, element_count, keys, keys_offset, storage, storage_offset,
                 query_pool=None, query=0):
        self._lib.vrdxCmdSort(_handle(command_buffer), self.handle, element_count, _handle(keys), keys_offset,
                              _handle(storage), storage_offset, _pool(query_pool), query)

    def cmd_sort_indirect(self, command_buffer, max_element_count, indirect, indirect_offset, keys,
                          keys_offset, storage, storage_offset, query_pool=None, query=0):
        self._lib.vrdxCmdSortIndirect(_handle(command_buffer), self.handle, max_element_count,
                                      _handle(indirect), indirect_offset, _handle(keys), keys_offset,
                                      _handle(storage), storage_offset, _pool(query_pool), query)

    def cmd_sort_key_value(self, command_buffer, element_count, keys, keys_offset, values, values_offset,
                           storage, storage_offset, query_pool=None, query=0):
        self._lib.vrdxCmdSortKeyValue(_handle(command_buffer), self.handle, element_count, _handle(keys),
                                      keys_offset, _handle(values), values_offset, _handle(storage),
                                      storage_offset, _pool(query_pool), query)

    def cmd_sort_key_value_indirect(self, command_buffer, max_element_count, indirect, indirect_offset,
                                    keys, keys_offset, values, values_offset, storage, storage_offset,
                                    query_pool=None, query=0):
        self._lib.vrdxCmdSortKeyValueIndirect(_handle(command_buffer), self.handle, max_element_count,
                                              _handle(indirect), indirect_offset, _handle(keys), keys_offset,
                                              _handle(values), values_offset, _handle(storage),
                                              storage_offset, _pool(query_pool), query)

    # -- diagnostics ------------------------------------------------------------------------
    def read_status(self, command_buffer, storage, storage_offset=0) -> int:
        return int(self._lib.vrdxHipReadStatus(_handle(command_buffer), _handle(storage), storage_offset))

    def read_sorter_status(self, command_buffer) -> int:
        """OR of the failure bits of every sort recorded with this sorter since the previous call,
        whatever storage they used (``vrdxHipReadSorterStatus``); synchronises the stream."""
        return int(self._lib.vrdxHipReadSorterStatus(self.handle, _handle(command_buffer)))


    def describe_plan(self, element_count: int, key_value: bool) -> VrdxHipPlanInfo:
        """``vrdxHipDescribePlan``: the plan this sorter records for that many elements and the HBM bytes per element it
        moves (what a whole-sort roofline figure is priced with)."""
        info = VrdxHipPlanInfo()
        self._lib.vrdxHipDescribePlan(self.handle, element_count, 1 if key_value else 0, ctypes.byref(info))
        return info

    def read_plan_verdict(self, command_buffer, storage, storage_offset=0) -> int:
        """``vrdxHipReadPlanVerdict``: what the device made of the plan of the last sort on this storage (``VERDICT_*``):
        ``describe_plan`` is the host's intention, this is what ran.  Synchronises the stream."""
        return int(self._lib.vrdxHipReadPlanVerdict(_handle(command_buffer), _handle(storage), storage_offset))

    def plan_taken(self, command_buffer, storage, storage_offset=0) -> bool:
        """True when the two-trip plan recorded for the last sort on this storage (hybrid-8 or msd) actually ran."""
        return self.read_plan_verdict(command_buffer, storage, storage_offset) in (
            VERDICT_HYBRID8_RUNS, VERDICT_MSD_RUNS, VERDICT_MSD_SORTED)

    def read_plan_counters(self, command_buffer):
        """``vrdxHipReadPlanCounters``: (sorts recorded with the MSD plan in front, how many of them the device turned
        down) since the sorter was created.  Synchronises the stream."""
        recorded, declined = ctypes.c_uint32(0), ctypes.c_uint32(0)
        r = self._lib.vrdxHipReadPlanCounters(self.handle, _handle(command_buffer), ctypes.byref(recorded),
                                              ctypes.byref(declined))
        if r != VK_SUCCESS:
            raise VrdxError("vrdxHipReadPlanCounters", r)
        return int(recorded.value), int(declined.value)

    def recheck(self) -> None:
        """``vrdxHipRecheck``: repeats the device check behind the one-atomic ranking and falls back to the ballot
        ranking for later sorts if it fails (synchronous, ~1 ms)."""
        r = self._lib.vrdxHipRecheck(self.handle)
        if r != VK_SUCCESS:
            raise VrdxError("vrdxHipRecheck", r)


def _pool(p) -> Optional[int]:
    if p is None:
        return None
    if isinstance(p, QueryPool):
        return p.handle
    return _handle(p)
