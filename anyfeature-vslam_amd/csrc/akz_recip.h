// akz_recip.h - the reciprocal of k_akz_fed_gauss's conductivity, shared with the program that proves it (tools/akaze_recip_check.hip)
#pragma once
// 1 / d for d >= 1, correctly rounded: v_rcp_f32 (1 ulp) + one Newton step in fused arithmetic.  Equal to the IEEE quotient for EVERY
// float in [1, 2^96) on gfx950 - tools/akaze_recip_check enumerates them (tests/test_gpu_akaze.py runs it) - at 4 instructions instead
// of the 11 of the division sequence; the conductivity's denominator 1 + |grad|^2 / k^2 is always in that range.
__device__ __forceinline__ float akz_recip_ge1(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
