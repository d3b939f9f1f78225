// probe_fence_victim.hip - round 6: what does a kernel that executes agent-scope fences (buffer_wbl2 sc1 / buffer_inv sc1), MFMA or
// atomics do to the loads, LDS and registers of ANOTHER kernel that shares the chip with it?
// (tests/test_gpu_match.py::test_three_threads_three_contexts: k_describe of one context produced wrong bits for the lanes 48..63 of a
// BRIEF group while another context ran the column-sliced k_match_topk_mfma.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_fence_victim.hip -o tools/probes/probe_fence_victim && tools/probes/probe_fence_victim
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned table_word(unsigned i) { return i * 2654435761u + 12345u; }

// victim: every wave re-reads a 4 KB table (one dwordx4 per lane and group, as k_describe reads its pattern), keeps a copy of the first
// read in registers and in LDS, and counts per lane quarter how often a later read / the LDS copy / the register copy disagrees
__global__ __launch_bounds__(256) void k_victim(const uint4 *__restrict__ table, int iters, unsigned *__restrict__ bad /* [3][4] */, int spread) {
    __shared__ uint4 s_copy[256 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    uint4 first[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        first[g] = table[g * 64 + lane];
        s_copy[tid * 4 + g] = first[g];
    }
    unsigned bl = 0, bs = 0, br = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // a different copy of the table every iteration (spread copies of 4 KB): loads miss L1 / L2 now and then
            const uint4 *vp = &table[(size_t)(it % spread) * 256 + g * 64 + lane];
            asm volatile("" : "+v"(vp));  // opaque: the load is not hoisted out of the loop
            const uint4 v = *vp;
            const unsigned i0 = (g * 64 + lane) * 4;
            if (v.x != table_word(i0) || v.y != table_word(i0 + 1) || v.z != table_word(i0 + 2) || v.w != table_word(i0 + 3)) ++bl;
            const uint4 s = s_copy[tid * 4 + g];
            if (s.x != table_word(i0) || s.w != table_word(i0 + 3)) ++bs;
            if (first[g].x != table_word(i0) || first[g].w != table_word(i0 + 3)) ++br;
            asm volatile("" : "+v"(first[g].x), "+v"(first[g].w));
        }
    }
    if (bl) atomicAdd(&bad[0 * 4 + (lane >> 4)], bl);
    if (bs) atomicAdd(&bad[1 * 4 + (lane >> 4)], bs);
    if (br) atomicAdd(&bad[2 * 4 + (lane >> 4)], br);
}

// aggressors
__global__ __launch_bounds__(256) void k_aggr(int mode, int iters, int *__restrict__ scratch) {
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // buffer_inv sc1
        else if (mode == 2) { scratch[blockIdx.x * 256 + threadIdx.x] = it; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }  // buffer_wbl2 sc1
        else if (mode == 3) { scratch[blockIdx.x * 256 + threadIdx.x] = it; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); }
        else if (mode == 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
        } else if (mode == 5) atomicAdd(&scratch[(blockIdx.x * 7 + it) & 1023], 1);
        else if (mode == 6) {  // one lane per workgroup fences (what the ticket hand-off does)
            if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); atomicAdd(&scratch[blockIdx.x & 63], 1); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
            __syncthreads();
        }
    }
    if (mode == 4) scratch[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[15];
}

int main(int argc, char **argv) {
    const int spread = 64, rounds = argc > 1 ? atoi(argv[1]) : 40;
    std::vector<unsigned> h((size_t)spread * 1024);
    for (int c = 0; c < spread; ++c)
        for (unsigned i = 0; i < 1024; ++i) h[(size_t)c * 1024 + i] = i * 2654435761u + 12345u;
    uint4 *d_table;
    unsigned *d_bad;
    int *d_scratch;
    CHK(hipMalloc(&d_table, h.size() * 4));
    CHK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_bad, 12 * 4));
    CHK(hipMalloc(&d_scratch, 4 << 20));
    CHK(hipMemset(d_scratch, 0, 4 << 20));
    hipStream_t sv, sa;
    CHK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    const char *names[] = {"none", "buffer_inv sc1 (acquire) every wave", "store + buffer_wbl2 sc1 (release) every wave", "store + acq_rel every wave", "MFMA i8 32x32x32 loop",
                           "atomics", "one lane per workgroup: release, atomic, acquire (the ticket hand-off)"};
    for (int mode = 0; mode <= 6; ++mode) {
        CHK(hipMemset(d_bad, 0, 48));
        CHK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r) {
            // many short launches of both (the real scene is launch-sized, not steady-state): victim 512 workgroups, aggressor 512
            if (mode) hipLaunchKernelGGL(k_aggr, dim3(512), dim3(256), 0, sa, mode, 200, d_scratch);
            hipLaunchKernelGGL(k_victim, dim3(512), dim3(256), 0, sv, d_table, 50, d_bad, spread);
        }
        CHK(hipDeviceSynchronize());
        unsigned b[12];
        CHK(hipMemcpy(b, d_bad, 48, hipMemcpyDeviceToHost));
        printf("aggressor %-75s | bad loads by lane quarter %u %u %u %u | bad LDS %u %u %u %u | bad registers %u %u %u %u\n", names[mode], b[0], b[1], b[2], b[3], b[4], b[5],
               b[6], b[7], b[8], b[9], b[10], b[11]);
    }
    return 0;
}
