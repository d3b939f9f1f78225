// probe_mfma_victim_valu.hip - round 6: does an MFMA-dense kernel disturb the VECTOR ARITHMETIC of another kernel that shares its SIMDs?
// (k_describe's self-checking build evaluated (x, x) * (ca, sb) + (y, y) * (-sb, ca) + 1.5 * 2^23 twice from bit-identical operands and got
// two different results in lanes 48..63 while the column-sliced k_match_topk_mfma of another context was running.)
// victim: every lane evaluates the same expression from the same operands over and over and compares with its first result.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma_victim_valu.hip -o tools/probes/probe_mfma_victim_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ unsigned eval(float x, float y, float ca, float sb, unsigned u) {
    if (KIND == 0) {  // packed fp32, as k_describe
        f32x2 xx = {x, x}, yy = {y, y}, a = {ca, sb}, b = {-sb, ca}, mg = {12582912.0f, 12582912.0f};
        asm volatile("" : "+v"(xx), "+v"(yy), "+v"(a), "+v"(b));
        const f32x2 r = xx * a + yy * b + mg;
        return __float_as_uint(r.x) * 31u + __float_as_uint(r.y);
    } else if (KIND == 1) {  // plain fp32
        asm volatile("" : "+v"(x), "+v"(y), "+v"(ca), "+v"(sb));
        const float r0 = x * ca + y * -sb + 12582912.0f, r1 = x * sb + y * ca + 12582912.0f;
        return __float_as_uint(r0) * 31u + __float_as_uint(r1);
    } else if (KIND == 2) {  // packed i16
        s16x2 p = __builtin_bit_cast(s16x2, u), q = __builtin_bit_cast(s16x2, u * 2654435761u);
        asm volatile("" : "+v"(p), "+v"(q));
        const s16x2 r = __builtin_elementwise_max(__builtin_elementwise_min(p, q), p - q) + q;
        return __builtin_bit_cast(unsigned, r);
    } else {  // integer
        unsigned p = u, q = u * 2654435761u;
        asm volatile("" : "+v"(p), "+v"(q));
        return (p + q) * 3u + (p ^ (q >> 3));
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_victim(int iters, unsigned *__restrict__ bad /* [4] lane quarters */) {
    const int lane = threadIdx.x & 63;
    const float x = (float)((int)(threadIdx.x * 7 % 27) - 13), y = (float)((int)(threadIdx.x * 5 % 27) - 13);
    const float ca = 0.7071f + 0.001f * (blockIdx.x & 63), sb = 0.6931f - 0.002f * (blockIdx.x & 31);
    const unsigned u = threadIdx.x * 40503u + blockIdx.x;
    const unsigned first = eval<KIND>(x, y, ca, sb, u);
    unsigned b = 0;
    for (int it = 0; it < iters; ++it) b += eval<KIND>(x, y, ca, sb, u) != first;
    if (b) atomicAdd(&bad[lane >> 4], b);
}

__global__ __launch_bounds__(256, 2) void k_aggr(int mode, int iters, int *__restrict__ scratch) {
    __shared__ v4i s_a[1024];
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    s_a[threadIdx.x] = a; s_a[threadIdx.x + 256] = b; s_a[threadIdx.x + 512] = a; s_a[threadIdx.x + 768] = b;
    __syncthreads();
    int k0 = 0x7fffffff, k1 = 0x7fffffff;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) {  // bare MFMA chain
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        } else {  // the shape of k_match_topk_mfma: LDS fragments, 9 MFMAs from a zero accumulator, 16 x (min + med3) on the result
            v4i f[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) f[k] = s_a[(threadIdx.x + 64 * k + it) & 1023];
            v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 9; ++k) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[k], b, c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = c[r];
                k1 = max(min(k0, k1), min(max(k0, k1), key));
                k0 = min(k0, key);
            }
        }
    }
    scratch[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[15] + k0 + k1;
}

template <int KIND>
static void run(const char *vname, hipStream_t sv, hipStream_t sa, unsigned *d_bad, int *d_scratch, int rounds) {
    const char *anames[] = {"none", "bare MFMA i8 32x32x32 chain", "LDS fragments + 9 MFMA + min / med3 (k_match_topk_mfma's shape)"};
    for (int mode = 0; mode <= 2; ++mode) {
        CHK(hipMemset(d_bad, 0, 16));
        CHK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r) {
            if (mode) hipLaunchKernelGGL(k_aggr, dim3(64), dim3(256), 0, sa, mode, 300, d_scratch);   // few workgroups, as a one-pair match
            hipLaunchKernelGGL(k_victim<KIND>, dim3(1024), dim3(256), 0, sv, 2000, d_bad);
        }
        CHK(hipDeviceSynchronize());
        unsigned b[4];
        CHK(hipMemcpy(b, d_bad, 16, hipMemcpyDeviceToHost));
        printf("victim %-12s aggressor %-70s | wrong results by lane quarter %u %u %u %u\n", vname, anames[mode], b[0], b[1], b[2], b[3]);
    }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    unsigned *d_bad;
    int *d_scratch;
    CHK(hipMalloc(&d_bad, 16));
    CHK(hipMalloc(&d_scratch, 1 << 20));
    hipStream_t sv, sa;
    CHK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    run<0>("packed fp32", sv, sa, d_bad, d_scratch, rounds);
    run<1>("plain fp32", sv, sa, d_bad, d_scratch, rounds);
    run<2>("packed i16", sv, sa, d_bad, d_scratch, rounds);
    run<3>("integer", sv, sa, d_bad, d_scratch, rounds);
    return 0;
}
