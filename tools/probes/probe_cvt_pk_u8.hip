// probe: does v_cvt_pk_u8_f32 round to nearest even and saturate?  (candidate for the blur's round-half-even(S / 65536))
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_cvt_pk_u8.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *S, uint32_t *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float f = (float)S[i] * (1.0f / 65536.0f);
    uint32_t r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(f));
    out[i] = r;
}
static uint32_t ref(uint32_t S) {
    uint32_t q = S >> 16, r = S & 0xffff;
    q += (r > 32768) || (r == 32768 && (q & 1));
    return q > 255 ? 255 : q;
}
int main() {
    const int n = 257 * 257 * 255 + 1;  // every value the 7x7 filter can produce
    uint32_t *h = (uint32_t *)malloc(n * 4), *o = (uint32_t *)malloc(n * 4), *dS, *dO;
    for (int i = 0; i < n; ++i) h[i] = i;
    hipMalloc(&dS, n * 4); hipMalloc(&dO, n * 4);
    hipMemcpy(dS, h, n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dS, dO, n);
    hipMemcpy(o, dO, n * 4, hipMemcpyDeviceToHost);
    long bad = 0; int first = -1;
    for (int i = 0; i < n; ++i) if (o[i] != ref(h[i])) { if (first < 0) first = i; ++bad; }
    printf("cvt_pk_u8_f32 vs round-half-even+saturate over %d values: %ld mismatches", n, bad);
    if (first >= 0) printf(" (first S=%d got %u want %u)", first, o[first], ref(first));
    printf("\n");
    return 0;
}
