// The native parity battery (vrdx_selftest.cpp) built against the SINGLE-HEADER distribution instead of
// libvrdx_hip.so: this one translation unit defines VRDX_IMPLEMENTATION, so it holds the host recorder
// and the embedded gfx950 code object (tools/generate_single_header.py).  Compiled with plain g++ and
// linked against libamdhip64 only -- no hipcc, no libvrdx_hip.so (reference:
// /root/reference/src/vk_radix_sort.h.in:85-98, bench/vrdx_impl.cc:1-4).
#define VRDX_IMPLEMENTATION
#include "../../build/single_header/vk_radix_sort.h"

#include "vrdx_selftest.cpp"
