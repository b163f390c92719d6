"""ctypes loaders for the oracle libraries.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_U32P = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(quiet: bool = True) -> None:
    """Compile liboracle.so and, when /root/reference is present, _ref/libvrdx_ref.so."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _as_u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


class Oracle:
    """Our CPU restatement (vrdx_oracle.c) and our port of the CPU backend (cpu_sort.cc)."""

    def __init__(self, path: str):
        lib = ctypes.CDLL(path)
        lib.vrdx_oracle_storage_size.restype = ctypes.c_uint64
        lib.vrdx_oracle_storage_size.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
        lib.vrdx_oracle_storage_usage.restype = ctypes.c_uint32
        lib.vrdx_oracle_storage_offsets.restype = None
        lib.vrdx_oracle_storage_offsets.argtypes = [ctypes.c_uint32, ctypes.c_uint32,
                                                    ctypes.POINTER(ctypes.c_uint64)]
        lib.vrdx_oracle_sort.restype = ctypes.c_int
        lib.vrdx_oracle_sort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.vrdx_oracle_digit_counts.restype = None
        lib.vrdx_oracle_digit_counts.argtypes = [_U32P, ctypes.c_uint32, _U32P]
        lib.vrdx_oracle_hash.restype = ctypes.c_uint64
        lib.vrdx_oracle_hash.argtypes = [_U32P, ctypes.c_uint64]
        lib.vrdx_port_sort_keys.restype = ctypes.c_int64
        lib.vrdx_port_sort_keys.argtypes = [_U32P, ctypes.c_uint64]
        lib.vrdx_port_sort_key_value.restype = ctypes.c_int64
        lib.vrdx_port_sort_key_value.argtypes = [_U32P, _U32P, ctypes.c_uint64]
        lib.vrdx_port_generate.restype = None
        lib.vrdx_port_generate.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_void_p, ctypes.c_void_p]
        self._lib = lib

    # -- storage math (src/vk_radix_sort.h.in:279-308) --------------------------------------
    def storage_size(self, n: int, key_value: bool, align: int = 16) -> int:
        return int(self._lib.vrdx_oracle_storage_size(n, align, 1 if key_value else 0))

    def storage_usage(self) -> int:
        return int(self._lib.vrdx_oracle_storage_usage())

    def storage_offsets(self, n: int, align: int = 16) -> dict:
        out = (ctypes.c_uint64 * 6)()
        self._lib.vrdx_oracle_storage_offsets(n, align, out)
        names = ["count", "histogram", "partition_histogram", "inout", "values_inout", "partitions"]
        return dict(zip(names, [int(x) for x in out]))

    # -- the sort itself (gpuSort restated) -------------------------------------------------
    def sort(self, keys, values=None, count: Optional[int] = None
             ) -> Tuple[np.ndarray, Optional[np.ndarray], np.ndarray]:
        """Returns (keys, values, global_histogram[4][256]) after sorting the first `count`
        elements the way the reference's four passes do.  Inputs are not modified."""
        k = _as_u32(keys).copy()
        v = _as_u32(values).copy() if values is not None else None
        n = k.size if count is None else int(count)
        assert n <= k.size and (v is None or v.size >= n)
        hist = np.zeros(4 * 256, dtype=np.uint32)
        rc = self._lib.vrdx_oracle_sort(k.ctypes.data, v.ctypes.data if v is not None else None, n,
                                        hist.ctypes.data)
        if rc != 0:
            raise MemoryError("vrdx_oracle_sort")
        return k, v, hist.reshape(4, 256)

    def digit_counts(self, keys, count: Optional[int] = None) -> np.ndarray:
        k = _as_u32(keys)
        out = np.zeros(4 * 256, dtype=np.uint32)
        self._lib.vrdx_oracle_digit_counts(k, k.size if count is None else count, out)
        return out.reshape(4, 256)

    def hash(self, data) -> int:
        d = _as_u32(data)
        return int(self._lib.vrdx_oracle_hash(d, d.size))

    # -- port of CpuBenchmark / DataGenerator ----------------------------------------------
    def port_sort_keys(self, keys) -> Tuple[np.ndarray, int]:
        k = _as_u32(keys).copy()
        ns = self._lib.vrdx_port_sort_keys(k, k.size)
        return k, int(ns)

    def port_sort_key_value(self, keys, values) -> Tuple[np.ndarray, np.ndarray, int]:
        k = _as_u32(keys).copy()
        v = _as_u32(values).copy()
        ns = self._lib.vrdx_port_sort_key_value(k, v, k.size)
        return k, v, int(ns)

    def generate(self, seed: int, size: int, bits: int = 32, with_values: bool = True):
        k = np.empty(size, dtype=np.uint32)
        v = np.empty(size, dtype=np.uint32) if with_values else None
        self._lib.vrdx_port_generate(seed, size, bits, k.ctypes.data,
                                     v.ctypes.data if v is not None else None)
        return k, v


class Reference:
    """The reference's own CPU backend (bench/cpu_benchmark.cc, bench/data_generator.cc)."""

    def __init__(self, path: str):
        lib = ctypes.CDLL(path)
        lib.vrdx_ref_sort_keys.restype = ctypes.c_int64
        lib.vrdx_ref_sort_keys.argtypes = [_U32P, ctypes.c_uint64]
        lib.vrdx_ref_sort_key_value.restype = ctypes.c_int64
        lib.vrdx_ref_sort_key_value.argtypes = [_U32P, _U32P, ctypes.c_uint64]
        lib.vrdx_ref_generate.restype = None
        lib.vrdx_ref_generate.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p]
        self._lib = lib

    def sort_keys(self, keys) -> Tuple[np.ndarray, int]:
        k = _as_u32(keys).copy()
        ns = self._lib.vrdx_ref_sort_keys(k, k.size)
        return k, int(ns)

    def sort_key_value(self, keys, values) -> Tuple[np.ndarray, np.ndarray, int]:
        k = _as_u32(keys).copy()
        v = _as_u32(values).copy()
        ns = self._lib.vrdx_ref_sort_key_value(k, v, k.size)
        return k, v, int(ns)

    def generate(self, seed: int, size: int, bits: int = 32):
        k = np.empty(size, dtype=np.uint32)
        v = np.empty(size, dtype=np.uint32)
        self._lib.vrdx_ref_generate(seed, size, bits, k.ctypes.data, v.ctypes.data)
        return k, v


def load_oracle(autobuild: bool = True) -> Oracle:
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path) and autobuild:
        build()
    return Oracle(path)


def load_reference() -> Optional[Reference]:
    """None when oracle/_ref was never built (no /root/reference and no prebuilt file)."""
    path = os.path.join(_HERE, "_ref", "libvrdx_ref.so")
    if not os.path.exists(path):
        return None
    return Reference(path)
