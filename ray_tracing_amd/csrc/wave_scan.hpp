// Inclusive scans over the 64 lanes of a wave with DPP row shifts and row
// broadcasts (gfx9 family): seven cross-lane steps, no LDS.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rt {

// The value a lane receives from the lane the DPP control selects, 0 when that
// lane does not exist or this lane is masked out by row_mask / bank_mask.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr3 = 0x113, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143;

// Lane i gets x[0] + ... + x[i].  All 64 lanes must be active.
// Seven adds that take one operand through the DPP network (written as assembly: left to the compiler,
// each step is a DPP move into a zeroed register plus an add).  A lane whose source does not exist, or that
// row_mask / bank_mask leave out, does not write and keeps its sum.  A DPP operand written by the
// instruction before needs two wait states, which inline assembly has to provide itself.
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    uint32_t s = x;
    asm("s_nop 1\n\t"                                                    // x may have been written just before
        "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %0, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %0, %1, %0 row_shr:3 row_mask:0xf bank_mask:0xf\n\t"   // 4-lane windows inside each row of 16
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"   // lanes 4..15 of a row add the window 4 below
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"   // lanes 8..15 add the prefix 8 below
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  // rows 1 and 3 add the total of the row before
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"       // rows 2 and 3 add the total of rows 0 and 1
        : "+&v"(s)  // early clobber: s starts as a copy of x and must not share its register
        : "v"(x));
    return s;
}

// Lane i gets max(x[0], ..., x[i]) (unsigned; 0 is the identity).
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    uint32_t s = x;
    s = mx(s, dpp_or_zero<kRowShr1, 0xf, 0xf>(x));
    s = mx(s, dpp_or_zero<kRowShr2, 0xf, 0xf>(x));
    s = mx(s, dpp_or_zero<kRowShr3, 0xf, 0xf>(x));
    s = mx(s, dpp_or_zero<kRowShr4, 0xf, 0xe>(s));
    s = mx(s, dpp_or_zero<kRowShr8, 0xf, 0xc>(s));
    s = mx(s, dpp_or_zero<kRowBcast15, 0xa, 0xf>(s));
    s = mx(s, dpp_or_zero<kRowBcast31, 0xc, 0xf>(s));
    return s;
}

}  // namespace rt
