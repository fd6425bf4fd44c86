// The exact three-way bf16 split of an fp32 number (a = a0 + a1 + a2, every plane a truncation: gemm_x6.hip's header has the
// derivation) shared by the x6 GEMM, the panel kernels and the weight packer: ONE definition, so that a weight split ahead of
// time (hg_panel_pack) and an activation split inside a kernel follow the same arithmetic.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t fbits(float x) { return __float_as_uint(x); }

// (x0, x1) -> their three bf16 planes, packed as bf16x2 (x0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
#ifdef GX_ABLATE_SPLIT
    p0 = fbits(x0); p1 = fbits(x1); p2 = fbits(x0) ^ fbits(x1);
    return;
#endif
    const uint32_t u0 = fbits(x0), u1 = fbits(x1);
    p0 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = fbits(r0), v1 = fbits(r1);
    p1 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(fbits(s1), fbits(s0), 0x07060302u);
}

}  // namespace
