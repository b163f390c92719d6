"""The single-header distribution: one generated `vk_radix_sort.h` with the device
code embedded as a gfx950 code object, activated by VRDX_IMPLEMENTATION in one translation unit -- the
reference's model (/root/reference/src/vk_radix_sort.h.in:85-98, tools/generate_header.py:5-35).

The consumer side is rebuilt HERE with plain g++ (no hipcc) and linked against libamdhip64 only (plus the
oracle, which is the checker): tests/native/selftest_single_header.cpp = `#define VRDX_IMPLEMENTATION`,
the generated header, and the native parity battery; it must pass the same `quick` battery as the .so."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "build", "single_header", "vk_radix_sort.h")
NATIVE = os.path.join(ROOT, "tests", "native")


def _header():
    if not os.path.exists(HEADER):  # packaging step: needs hipcc (the image has it on both boxes)
        subprocess.run(["python3", os.path.join(ROOT, "tools", "generate_single_header.py"), "-o", HEADER], check=True)
    return HEADER


def test_single_header_is_self_contained_and_builds_with_a_host_compiler_only(tmp_path):
    """CPU: the generated header compiles as C++17 with g++ both ways -- declarations only, and with
    VRDX_IMPLEMENTATION -- and the implementation unit exports the eight reference entry points."""
    header = _header()
    text = open(header).read()
    assert "static const unsigned char kVrdxCodeObject[]" in text and "#ifdef VRDX_IMPLEMENTATION" in text
    assert '#include "' not in text.split("#ifdef VRDX_IMPLEMENTATION", 1)[1], "the implementation section includes a local file"
    shutil.copy(header, tmp_path / "vk_radix_sort.h")
    (tmp_path / "decl.cc").write_text('#include "vk_radix_sort.h"\nint main() { return VRDX_VERSION_MAJOR < 0; }\n')
    (tmp_path / "impl.cc").write_text('#define VRDX_IMPLEMENTATION\n#include "vk_radix_sort.h"\n')
    gxx = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + str(tmp_path)]
    subprocess.run(gxx + ["-c", str(tmp_path / "decl.cc"), "-o", str(tmp_path / "decl.o")], check=True)
    subprocess.run(gxx + ["-c", str(tmp_path / "impl.cc"), "-o", str(tmp_path / "impl.o")], check=True)
    nm = subprocess.run(["nm", "-g", "--defined-only", str(tmp_path / "impl.o")], capture_output=True, text=True, check=True).stdout
    for name in ("vrdxCreateSorter", "vrdxDestroySorter", "vrdxGetSorterStorageRequirements",
                 "vrdxGetSorterKeyValueStorageRequirements", "vrdxCmdSort", "vrdxCmdSortIndirect", "vrdxCmdSortKeyValue",
                 "vrdxCmdSortKeyValueIndirect"):
        assert f" T {name}\n" in nm, name


@pytest.mark.gpu
def test_single_header_passes_the_native_parity_battery():
    _header()
    exe = os.path.join(NATIVE, "selftest_single_header")
    if os.path.exists(exe):
        os.remove(exe)
    subprocess.run(["make", "-C", NATIVE, "selftest_single_header"], check=True, capture_output=True)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True, check=True).stdout
    assert "libvrdx_hip" not in ldd and "libamdhip64" in ldd
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_single_header_passes_the_msd_battery():
    """The MSD plan through the single header's own launcher (vrdx_module_launch.inc: hipModuleLaunchKernel by mangled name,
    its grids and LDS sizes restated there) -- the sizes the quick battery stops short of: both bucket kernels, both
    windows, the plan's launches in their second role as passes 0 and 1 (keys-only and key+value), every verdict."""
    _header()
    exe = os.path.join(NATIVE, "selftest_single_header")
    subprocess.run(["make", "-C", NATIVE, "selftest_single_header"], check=True, capture_output=True)
    r = subprocess.run([exe, "msd"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
