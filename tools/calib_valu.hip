// calib_valu.hip — measured integer-VALU issue peak of this GPU for the instruction classes the pipeline uses
// (v_add_u32, v_pk_min_i16, v_pk_mad_u16, v_dot4_u32_u8, v_bcnt_u32_b32).  Every wave runs ITER x 32 independent
// ops on 8 accumulators; 8 waves/SIMD resident.  Prints wave-instructions/s per class.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2v __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    uint32_t b = seed | 1;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] + a[(i + 3) & 7];  // not foldable into one multiply-add
                if (OP == 1) { short2v x = __builtin_bit_cast(short2v, a[i]), y = __builtin_bit_cast(short2v, b); a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y)) + 1u; }
                if (OP == 2) { short2v x = __builtin_bit_cast(short2v, a[i]), y = __builtin_bit_cast(short2v, b); a[i] = __builtin_bit_cast(uint32_t, x * y + y); }
                if (OP == 3) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);
                if (OP == 4) a[i] = __builtin_popcount(a[i] ^ b) + a[i];
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    if (s == 0x12345) out[0] = s;
}
template <int OP> static double run(const char *name, double ops_per_elem) {
    uint32_t *d;
    hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8 * 4;  // 8 workgroups of 4 waves per CU, 4 rounds
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 7u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * ITER * 32 * ops_per_elem;
    printf("%-16s %.3f ms  %.3e wave-instr/s\n", name, ms, winst / (ms * 1e-3));
    return winst / (ms * 1e-3);
}
int main() {
    run<0>("v_add_u32", 1);
    run<1>("v_pk_min_i16+add", 2);
    run<2>("v_pk_mad_u16", 1);
    run<3>("v_dot4_u32_u8", 1);
    run<4>("xor+bcnt", 2);
    return 0;
}
