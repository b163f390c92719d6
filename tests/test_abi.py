"""CPU-only: the C-ABI boundary.  The library must load without a GPU, export every symbol that
include/vk_radix_sort.h declares, keep the reference's struct layouts, and fail LOUDLY (never fall
back to a CPU path) when there is no gfx950 device."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vk_radix_sort.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(vrdx[A-Z]\w+)\s*\(", text)))


def test_header_declares_the_reference_entry_points():
    names = _declared_functions()
    for required in ("vrdxCreateSorter", "vrdxDestroySorter", "vrdxGetSorterStorageRequirements",
                     "vrdxGetSorterKeyValueStorageRequirements", "vrdxCmdSort", "vrdxCmdSortIndirect",
                     "vrdxCmdSortKeyValue", "vrdxCmdSortKeyValueIndirect"):
        assert required in names


def test_library_exports_every_declared_symbol():
    import vulkan_radix_sort_amd as vrdx
    lib = vrdx.load_library()
    declared = _declared_functions()
    assert set(declared) == set(vrdx.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None, name
    assert "gfx950" in vrdx.version_string()


def test_struct_layouts_match_the_reference():
    from vulkan_radix_sort_amd.api import VrdxSorterCreateInfo, VrdxSorterStorageRequirements
    # src/vk_radix_sort.h.in:18-22 (3 handles) and :28-31 (VkDeviceSize + VkBufferUsageFlags)
    assert ctypes.sizeof(VrdxSorterCreateInfo) == 24
    assert ctypes.sizeof(VrdxSorterStorageRequirements) == 16
    assert VrdxSorterStorageRequirements.usage.offset == 8


def test_create_sorter_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vulkan_radix_sort_amd as vrdx
    with pytest.raises(vrdx.VrdxError) as e:
        vrdx.Sorter()
    assert e.value.result == -3  # VK_ERROR_INITIALIZATION_FAILED
    # destroy is NULL-safe like the reference (src/vk_radix_sort.h.in:268)
    vrdx.load_library().vrdxDestroySorter(None)


def test_product_never_touches_the_oracle():
    # the judge's rule: nothing under vulkan_radix_sort_amd/ may import, link or call oracle/
    pkg = os.path.join(ROOT, "vulkan_radix_sort_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, name), errors="ignore").read()
                assert "liboracle" not in text and "vrdx_oracle" not in text, name
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), name
    out = subprocess.run(["ldd", os.path.join(pkg, "libvrdx_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "vrdx_ref" not in out


@pytest.mark.parametrize("compiler,lang", [("gcc", "c"), ("g++", "c++")])
def test_header_compiles_as_c_and_cpp(tmp_path, compiler, lang):
    src = tmp_path / ("t.c" if lang == "c" else "t.cc")
    src.write_text(
        '#define VRDX_IMPLEMENTATION\n#include "vk_radix_sort.h"\n'
        "int main(void) {\n"
        "  VrdxSorterCreateInfo info = {0};\n  VrdxSorterStorageRequirements req;\n  VrdxSorter s = 0;\n"
        "  (void)info; (void)req; (void)s;\n"
        "  return (sizeof(VrdxSorterStorageRequirements) == 16 && sizeof(VrdxSorterCreateInfo) == 24\n"
        "          && VRDX_VERSION == ((0 << 22) | (4 << 12) | 0)) ? 0 : 1;\n}\n")
    exe = tmp_path / "t"
    subprocess.run([compiler, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_bench_driver_cpu_backend_and_csv_format(tmp_path):
    """bench/bench (the reference's bench/bench.cc protocol): cpu backend runs without a GPU and the
    CSV keeps the reference's seven columns (tools/plot.py reads them) plus the two roofline ones."""
    exe = os.path.join(ROOT, "bench", "bench")
    if not os.path.exists(exe):
        pytest.skip("bench/bench not built (needs hipcc for the rocPRIM comparator)")
    out = tmp_path / "r.csv"
    subprocess.run([exe, "cpu", "--points", "3", "--min-log2n", "10", "--max-log2n", "13", "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL)
    lines = [l for l in out.read_text().splitlines() if not l.startswith("#")]
    assert lines[0] == "backend,n,sort,gpu_ms,cpu_ms,gpu_gitems_s,cpu_gitems_s,achieved_GBps,hbm_fraction"
    rows = [l.split(",") for l in lines[1:]]
    assert [r[1] for r in rows] == ["1024", "1024", "4608", "4608", "8192", "8192"]
    assert [r[2] for r in rows] == ["keys", "kv"] * 3
    assert all(r[0] == "cpu" and float(r[3]) > 0 for r in rows)


def test_storage_layout_invariants():
    """vrdx_layout.h: status regions and tickets stay inside the reference's partition-histogram area
    for every N up to the ceiling and every tile size, and the totals equal the oracle's (CPU only)."""
    native = os.path.join(ROOT, "tests", "native")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    subprocess.run(["make", "-C", native, "layout_check"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(native, "layout_check")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and ", 0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_py_launcher_refuses_more_gpus_than_present():
    """`python bench.py --gpus N` without a launcher starts its own ranks in a child process -- unless
    the node has fewer GPUs than asked for (here: none), which is an error, never a silent N = 1."""
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("enough GPUs")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 2 and "exposes" in r.stderr and not r.stdout.strip()


def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    """ASan + UBSan build of everything that runs on the host without a GPU: vrdx_layout.h, the oracle
    (vrdx_oracle.c, cpu_sort.cc) and the bench driver's cpu backend (GPU sanitizers do not exist on this pool)."""
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    obj = tmp_path / "vrdx_oracle.o"
    subprocess.run(["gcc", "-std=c99", "-c", os.path.join(ROOT, "oracle", "vrdx_oracle.c"), "-o", str(obj)] + san, check=True)
    exe = tmp_path / "sanitized_host_check"
    subprocess.run(["g++", "-std=c++17", os.path.join(ROOT, "tests", "native", "sanitized_host_check.cpp"),
                    os.path.join(ROOT, "oracle", "cpu_sort.cc"), str(obj), "-o", str(exe)] + san, check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_no_kernel_uses_scratch_memory():
    """A spill is not an option on this path (120 bytes of scratch per lane made a pass 30x slower, DESIGN.md section
    4.2): the compiler's resource report (make resources) must show ScratchSize 0 for EVERY kernel of the code object,
    the ballot-ranking forms included -- whichever ranking mode vrdxCreateSorter selects, no selectable kernel spills."""
    import re
    import subprocess
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "vulkan_radix_sort_amd", "csrc"), "resources"],
                         capture_output=True, text=True, timeout=900)
    report = out.stdout + out.stderr
    names = re.findall(r"Function Name: (\S+)", report)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", report)]
    assert len(names) >= 20 and len(names) == len(scratch), report[-2000:]
    spilled = [(n, s) for n, s in zip(names, scratch) if s != 0]
    assert not spilled, spilled
