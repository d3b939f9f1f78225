// probe_pk_war.hip - round 6: is "packed-fp32 instruction reads v[n:n+1] with op_sel, the NEXT instruction overwrites v[n]" safe on gfx950
// when another kernel keeps the matrix pipe busy?  The compiler emits exactly this pair in k_describe's rBRIEF rotation:
//     v_pk_mul_f32 v[18:19], v[4:5], v[14:15] op_sel_hi:[1,0]      ; (ca, sb) * (x1, x1)
//     v_mov_b32    v14, v15                                        ; y1 over x1
// and the lanes 48..63 of the product were seen to use y1 (tests/test_gpu_match.py::test_three_threads_three_contexts, DESIGN_LOG round 6).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_pk_war.hip -o tools/probes/probe_pk_war && tools/probes/probe_pk_war
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// NOPS wait states between the packed multiply and the overwrite of its source
template <int NOPS>
__global__ __launch_bounds__(256) void k_victim(int iters, unsigned *__restrict__ bad /* [4] lane quarters */) {
    const int lane = threadIdx.x & 63;
    const float ca = 0.75f, sb = 0.5f;
    unsigned b = 0;
    for (int it = 0; it < iters; ++it) {
        float x_in = (float)(lane + 1), y_in = (float)(-lane - 100), rx, ry;
        asm volatile("" : "+v"(x_in), "+v"(y_in));
#define PK_WAR(NOP)                                                                                                   \
    asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v22, %4\n\tv_mov_b32 v23, %5\n\ts_nop 4\n\t"       \
                 "v_pk_mul_f32 v[24:25], v[22:23], v[20:21] op_sel_hi:[1,0]\n\t" NOP "v_mov_b32 v20, v21\n\ts_nop 4\n\t"   \
                 "v_mov_b32 %0, v24\n\tv_mov_b32 %1, v25"                                                             \
                 : "=v"(rx), "=v"(ry) : "v"(x_in), "v"(y_in), "v"(ca), "v"(sb) : "v20", "v21", "v22", "v23", "v24", "v25")
        if (NOPS == 0) PK_WAR("");
        else if (NOPS == 1) PK_WAR("s_nop 0\n\t");
        else PK_WAR("s_nop 3\n\t");
        f32x2 r = {rx, ry};
        // expected: (ca * x, sb * x) with x = lane + 1
        const float x = (float)(lane + 1);
        b += (r.x != ca * x || r.y != sb * x) ? 1u : 0u;
    }
    if (b) atomicAdd(&bad[lane >> 4], b);
}

__global__ __launch_bounds__(256, 2) void k_aggr(int iters, int *__restrict__ scratch) {
    __shared__ v4i s_a[1024];
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    s_a[threadIdx.x] = a; s_a[threadIdx.x + 256] = b; s_a[threadIdx.x + 512] = a; s_a[threadIdx.x + 768] = b;
    __syncthreads();
    int k0 = 0x7fffffff, k1 = 0x7fffffff;
    for (int it = 0; it < iters; ++it) {
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
    scratch[blockIdx.x * 256 + threadIdx.x] = k0 + k1;
}

template <int NOPS>
static void run(hipStream_t sv, hipStream_t sa, unsigned *d_bad, int *d_scratch, int rounds) {
    for (int aggr = 0; aggr <= 1; ++aggr) {
        CHK(hipMemset(d_bad, 0, 16));
        CHK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r) {
            if (aggr) hipLaunchKernelGGL(k_aggr, dim3(256), dim3(256), 0, sa, 400, d_scratch);
            hipLaunchKernelGGL(k_victim<NOPS>, dim3(1024), dim3(256), 0, sv, 4000, d_bad);
        }
        CHK(hipDeviceSynchronize());
        unsigned b[4];
        CHK(hipMemcpy(b, d_bad, 16, hipMemcpyDeviceToHost));
        printf("wait states between v_pk_mul_f32 and the overwrite of its source: %d | MFMA kernel beside it: %s | wrong products by lane quarter %u %u %u %u\n",
               NOPS == 0 ? 0 : NOPS == 1 ? 1 : 4, aggr ? "yes" : "no ", b[0], b[1], b[2], b[3]);
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
    run<0>(sv, sa, d_bad, d_scratch, rounds);
    run<1>(sv, sa, d_bad, d_scratch, rounds);
    run<2>(sv, sa, d_bad, d_scratch, rounds);
    return 0;
}
