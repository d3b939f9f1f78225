// measurement (round 5): WHY does a strip-shaped 1-read : 3-write float stream (k_akz_dhess's shape) run at 3.4 - 3.8 TB/s when full
// rows reach 5.3?  Candidates: idle lanes (fewer bytes per memory instruction), partial cache lines completed by another workgroup on
// another XCD, or neither.  Variants over strips of OW output columns x 64 rows, one wavefront per strip, 4 per workgroup:
//   base      lanes >= OW idle (probe_stream13's form)
//   xcd       the same, workgroups renumbered so that every XCD owns a contiguous range of strips
//   halo      all 64 lanes load (columns x0 - (64 - OW) / 2 ..), OW lanes store: the real kernel's access shape
//   halo+xcd
//   lds       the 4 wavefronts of a workgroup stage 8 rows of their outputs in LDS and the workgroup stores full 256-byte lines
//             (4 x 48 = 192 columns = 3 lines per row and plane)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROWS 64
template <int OW, bool HALO, bool XCD>
__global__ __launch_bounds__(256) void k_strips(const float *__restrict__ in, float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int w, int h, int nframes) {
    const int lane = threadIdx.x & 63;
    const int nstr = (w + OW - 1) / OW, nband = (h + ROWS - 1) / ROWS;
    int blk = blockIdx.x;
    if (XCD) { const int per = gridDim.x / 8; blk = (blk & 7) * per + (blk >> 3); }   // gridDim.x is a multiple of 8
    int id = blk * 4 + (threadIdx.x >> 6);
    if (id >= nstr * nband * nframes) return;
    const int f = id / (nstr * nband); id -= f * nstr * nband;
    const int band = id / nstr, x0 = (id - band * nstr) * OW, y0 = band * ROWS;
    constexpr int H = (64 - OW) / 2;
    const int x = HALO ? x0 - H + lane : x0 + lane;
    const bool out = HALO ? (lane >= H && lane < H + OW && x < w) : (lane < OW && x < w);
    if (!HALO && !out) return;
    const int xl = min(max(x, 0), w - 1);
    const size_t fo = (size_t)f * w * h;
    const int y1 = min(y0 + ROWS, h);
#pragma unroll 4
    for (int y = y0; y < y1; ++y) {
        const float v = in[fo + (size_t)y * w + xl];
        if (out) { const size_t o = fo + (size_t)y * w + x; a[o] = v * 2.0f; b[o] = v + 1.0f; c[o] = v * v; }
    }
}
// 4 wavefronts = 4 strips of 48 outputs (64 loaded) -> 192 columns; 8 rows staged per round
__global__ __launch_bounds__(256) void k_lds(const float *__restrict__ in, float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int w, int h, int nframes, int xcd) {
    constexpr int OW = 48, H = 8, GW = 192, R = 8;
    __shared__ float s[3][R][GW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ngrp = (w + GW - 1) / GW, nband = (h + ROWS - 1) / ROWS;
    int blk = blockIdx.x;
    if (xcd) { const int per = gridDim.x / 8; blk = (blk & 7) * per + (blk >> 3); }
    if (blk >= ngrp * nband * nframes) return;
    const int f = blk / (ngrp * nband); blk -= f * ngrp * nband;
    const int band = blk / ngrp, gx0 = (blk - band * ngrp) * GW, y0 = band * ROWS;
    const int x = gx0 + wv * OW - H + lane;
    const bool out = lane >= H && lane < H + OW;
    const int xl = min(max(x, 0), w - 1);
    const size_t fo = (size_t)f * w * h;
    const int y1 = min(y0 + ROWS, h);
    for (int yb = y0; yb < y1; yb += R) {
        float v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = in[fo + (size_t)min(yb + r, h - 1) * w + xl];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (out) { const int cx = wv * OW + lane - H; s[0][r][cx] = v[r] * 2.0f; s[1][r][cx] = v[r] + 1.0f; s[2][r][cx] = v[r] * v[r]; }
        __syncthreads();
        // 3 planes x R rows x 3 lines of 64 floats = 72 line stores, 18 per wavefront
        for (int i = wv; i < 3 * R * 3; i += 4) {
            const int pl = i / (R * 3), rr = (i / 3) % R, seg = i % 3;
            const int gx = gx0 + seg * 64 + lane, gy = yb + rr;
            if (gx < w && gy < y1) {
                float *dst = pl == 0 ? a : pl == 1 ? b : c;
                dst[fo + (size_t)gy * w + gx] = s[pl][rr][seg * 64 + lane];
            }
        }
        __syncthreads();
    }
}
int main() {
    const int w = 1280, h = 720, nf = 64; const size_t n = (size_t)w * h * nf;
    float *in, *a, *b, *c;
    hipMalloc(&in, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMemset(in, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.1f us  %.2f TB/s (16 B per pixel)\n", name, ms * 100, n * 16.0 / (ms / 10 * 1e-3) / 1e12);
    };
#define RUN(OW, HALO, XCD, NAME) time(NAME, [&] { const int ns = ((w + OW - 1) / OW) * ((h + ROWS - 1) / ROWS) * nf; const int g = ((ns + 3) / 4 + 7) / 8 * 8; \
        hipLaunchKernelGGL((k_strips<OW, HALO, XCD>), dim3(g), dim3(256), 0, 0, in, a, b, c, w, h, nf); });
    RUN(64, false, false, "64 columns base")
    RUN(64, false, true, "64 columns xcd")
    RUN(56, false, false, "56 columns base")
    RUN(56, false, true, "56 columns xcd")
    RUN(56, true, false, "56 columns halo")
    RUN(56, true, true, "56 columns halo+xcd")
    RUN(48, false, false, "48 columns base")
    RUN(48, false, true, "48 columns xcd")
    RUN(48, true, false, "48 columns halo")
    RUN(48, true, true, "48 columns halo+xcd")
    RUN(32, true, false, "32 columns halo")
    for (int xcd = 0; xcd < 2; ++xcd)
        time(xcd ? "lds-staged 4 x 48 -> 3 lines, xcd" : "lds-staged 4 x 48 -> 3 lines", [&] { const int ng = ((w + 191) / 192) * ((h + ROWS - 1) / ROWS) * nf; const int g = (ng + 7) / 8 * 8;
            hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 0, 0, in, a, b, c, w, h, nf, xcd); });
    return 0;
}
