#!/usr/bin/env python3
"""Generates the single-header distribution of the HIP backend: ONE file, `vk_radix_sort.h`, that holds the
public interface and -- behind `#define VRDX_IMPLEMENTATION` in exactly one translation unit -- the whole
implementation with the device code EMBEDDED as a gfx950 code object.

This is the reference's distribution model: its `include/vk_radix_sort.h` is the
template `src/vk_radix_sort.h.in` with the compiled SPIR-V of the four shaders spliced in as uint32 arrays
(/root/reference/tools/generate_header.py:5-35, tools/slangc_to_header.py:43-57), activated by
VRDX_IMPLEMENTATION (src/vk_radix_sort.h.in:85-98).  Here:

    vrdx_kernels.hip --(hipcc --genco --offload-arch=gfx950)--> code object --> unsigned char array
    include/vk_radix_sort.h  +  vrdx_layout.h  +  vrdx_kernels.h  +  embedded array
                             +  vrdx_module_launch.inc (hipModuleLoadData / hipModuleLaunchKernel)
                             +  vrdx_api.cpp (the host recorder, unchanged)

A consumer needs a host C++ compiler and libamdhip64 only:

    // one .cc file
    #define VRDX_IMPLEMENTATION
    #include "vk_radix_sort.h"
    // g++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include app.cc -L/opt/rocm/lib -lamdhip64

usage: generate_single_header.py [-o build/single_header/vk_radix_sort.h] [--hipcc /opt/rocm/bin/hipcc]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vulkan_radix_sort_amd", "csrc")

# every kernel the launcher (vrdx_module_launch.inc) can ask for, by mangled name
def expected_kernels():
    names = ["_ZN4vrdx22lds_order_check_kernelEPjS0_", "_ZN4vrdx11spin_kernelEPyj"]
    names += ["_ZN4vrdx16histogram_kernelILj%uEEEvPKjjS2_PjS3_PDv4_jj" % c for c in (8, 32)]
    for threads, kpt, sub in ((1024, 8, 1), (1024, 16, 1), (1024, 32, 1), (1024, 32, 2)):
        for kv in (0, 1):
            for atomic in (0, 1):
                # last template argument: the form with run-time slot counts (the 32-keys-per-thread geometries)
                if sub == 2:  # the two-sub-tile kernel: keys-only, one-atomic ranking
                    if not kv and atomic:
                        for even in (0, 1):
                            names.append("_ZN4vrdx20onesweep_pair_kernelILi%dELi%dELb%dEEEvNS_12OnesweepArgsE" % (threads, kpt, even))
                    continue
                for even in ((0, 1) if kpt == 32 else (0,)):
                    names.append("_ZN4vrdx15onesweep_kernelILi%dELi%dELb%dELb%dELb%dEEEvNS_12OnesweepArgsE"
                                 % (threads, kpt, kv, atomic, even))
    # the MSD plan: histogram with per-tile counts, spine, scatter, two-pass bucket sort; the packed-counter order check
    names.append("_ZN4vrdx29lds_order_check_packed_kernelEPjS0_")
    names += ["_ZN4vrdx24bucket_sort2_half_kernelILj10ELb%dEEEvNS_7MsdArgsE" % kv for kv in (0, 1)]
    for bits in (10, 11):
        names.append("_ZN4vrdx16spine_msd_kernelILj%dEEEvNS_7MsdArgsE" % bits)
        names += ["_ZN4vrdx20histogram_msd_kernelILj%dELj%dEEEvNS_7MsdArgsE" % (c, bits) for c in (8, 32)]
        names += ["_ZN4vrdx18scatter_msd_kernelILj%dELb%dEEEvNS_7MsdArgsE" % (bits, kv) for kv in (0, 1)]
        names += ["_ZN4vrdx19bucket_sort2_kernelILj%dELi36ELb%dEEEvNS_7MsdArgsE" % (bits, kv) for kv in (0, 1)]
        for kernel in ("27msd_scatter_or_pass0_kernel", "27msd_buckets_or_pass1_kernel"):  # the plan's launch or a pass of its fallback
            names += ["_ZN4vrdx%sILj%dELb%dELb%dEEEvNS_7MsdArgsENS_12OnesweepArgsE" % (kernel, bits, kv, dyn) for kv in (0, 1) for dyn in (0, 1)]
    for kv in (0, 1):  # 32768-element buckets: one-atomic ranking only
        names.append("_ZN4vrdx18bucket_sort_kernelILi1024ELi32ELb%dELb1EEEvNS_14BucketSortArgsE" % kv)
    for kpt in (4, 8, 16):
        for kv in (0, 1):
            for atomic in (0, 1):
                names.append("_ZN4vrdx18bucket_sort_kernelILi1024ELi%dELb%dELb%dEEEvNS_14BucketSortArgsE" % (kpt, kv, atomic))
    for threads in (256, 1024):
        for kv in (0, 1):
            for atomic in (0, 1):
                names.append("_ZN4vrdx17small_sort_kernelILi%dELi16ELb%dELb%dEEEvPjS1_jPKjS1_" % (threads, kv, atomic))
    return names


def strip_local_includes(text):
    """The concatenated sources include each other by relative path; in the single header they simply follow
    each other."""
    return re.sub(r'^#include "(\.\./\.\./include/vk_radix_sort\.h|vrdx_kernels\.h|vrdx_layout\.h)"\n', "", text, flags=re.M)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output", default=os.path.join(ROOT, "build", "single_header", "vk_radix_sort.h"))
    ap.add_argument("--hipcc", default="/opt/rocm/bin/hipcc")
    args = ap.parse_args()

    with tempfile.TemporaryDirectory() as tmp:
        hsaco = os.path.join(tmp, "vrdx_kernels.hsaco")
        subprocess.run([args.hipcc, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17",
                        os.path.join(CSRC, "vrdx_kernels.hip"), "-o", hsaco], check=True, cwd=CSRC)
        blob = open(hsaco, "rb").read()
    missing = [n for n in expected_kernels() if n.encode() not in blob]
    if missing:
        sys.exit("kernels missing from the code object (mangling changed?):\n  " + "\n  ".join(missing))

    public = open(os.path.join(ROOT, "include", "vk_radix_sort.h")).read()
    marker = "/* The reference is a single-header library activated with VRDX_IMPLEMENTATION"
    cut = public.index(marker)
    head = public[:cut]

    parts = [head]
    parts.append("""/* ======================================================================================
 * GENERATED by tools/generate_single_header.py -- the single-header distribution.
 * Define VRDX_IMPLEMENTATION in exactly ONE translation unit before including this file
 * (like the reference: src/vk_radix_sort.h.in:85-98, bench/vrdx_impl.cc:1-4); that unit then
 * contains the host recorder and the embedded gfx950 code object (%d bytes) and needs only a
 * host C++17 compiler and libamdhip64.
 * ====================================================================================== */
#ifdef VRDX_IMPLEMENTATION
#undef VRDX_IMPLEMENTATION
#ifndef VRDX_SINGLE_HEADER_IMPLEMENTATION_INCLUDED
#define VRDX_SINGLE_HEADER_IMPLEMENTATION_INCLUDED
""" % len(blob))
    parts.append("// ---- vulkan_radix_sort_amd/csrc/vrdx_layout.h ----\n" + open(os.path.join(CSRC, "vrdx_layout.h")).read())
    parts.append("// ---- vulkan_radix_sort_amd/csrc/vrdx_kernels.h ----\n" + strip_local_includes(open(os.path.join(CSRC, "vrdx_kernels.h")).read()))
    rows = []
    for i in range(0, len(blob), 32):
        rows.append(",".join(str(b) for b in blob[i:i + 32]))
    parts.append("// ---- the kernels of vrdx_kernels.hip, compiled for gfx950 (hipcc --genco) ----\n"
                 "static const unsigned char kVrdxCodeObject[] = {\n" + ",\n".join(rows) + "\n};\n"
                 "static const unsigned long kVrdxCodeObjectSize = sizeof(kVrdxCodeObject);\n")
    parts.append("// ---- vulkan_radix_sort_amd/csrc/vrdx_module_launch.inc ----\n" + open(os.path.join(CSRC, "vrdx_module_launch.inc")).read())
    parts.append("// ---- vulkan_radix_sort_amd/csrc/vrdx_api.cpp ----\n" + strip_local_includes(open(os.path.join(CSRC, "vrdx_api.cpp")).read()))
    parts.append("#endif /* VRDX_SINGLE_HEADER_IMPLEMENTATION_INCLUDED */\n#endif /* VRDX_IMPLEMENTATION */\n")

    os.makedirs(os.path.dirname(args.output), exist_ok=True)
    with open(args.output, "w") as f:
        f.write("\n".join(parts))
    print("wrote %s (%d bytes, code object %d bytes, %d kernels)" % (args.output, os.path.getsize(args.output), len(blob), len(expected_kernels())))


if __name__ == "__main__":
    main()
