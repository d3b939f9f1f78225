// k_akaze.hip — SURVEY §8f rank 4 (config #5): AKAZE nonlinear scale space + determinant-of-Hessian response.
//
// Replaces libAKAZE's Create_Nonlinear_Scale_Space and Compute_Determinant_Hessian_Response as driven by
// FeatureExtractor_akaze61::initializeExtractor / detectKeypoints (Feature_akaze61.cpp:26-47).  The library is an absent
// fork (fontan::akaze); the arithmetic follows upstream libAKAZE 1.5 with the explicit operation order written down in
// oracle/akaze.c (parity unpinned against the fork, bit-exact against that restatement).
//
// All of it is float stencil work on planes of w x h x 4 B — the HBM-bound part of this path.  Every kernel owns a
// 64 x 32 output tile per 256-thread workgroup, stages its inputs (+ halo, with the border rule of the OpenCV call it
// replaces baked into the staging coordinates) in LDS with coalesced row reads, and writes each output once.
// Batch layout: plane[frame][y][x], frame stride = w * h of the level.
#include "afv_device.h"
#include "akz_jobs.h"
#include "akz_recip.h"

#define AT_W 64
#define AT_H 32
#define AKZ_T 256

// XCD-aware tile order: the launch is 1-D, block b runs on XCD b % 8 (afv_device.h) and takes work item
// (b % 8) * ceil(n / 8) + b / 8 of the (frame, tile row, tile column) list, so that tiles sharing halo rows stay in one L2.
#define AKZ_TILE(TW, TH)                                                                        \
    const int tiles_x_ = (w + (TW) - 1) / (TW), tiles_y_ = (h + (TH) - 1) / (TH);               \
    const int total_ = tiles_x_ * tiles_y_ * nframes;                                           \
    const int work_ = afv_xcd_remap(blockIdx.x, total_);                                        \
    if (work_ >= total_) return;                                                                \
    const int f = work_ / (tiles_x_ * tiles_y_), t_ = work_ - f * (tiles_x_ * tiles_y_);        \
    const int y0 = (t_ / tiles_x_) * (TH), x0 = (t_ - (t_ / tiles_x_) * tiles_x_) * (TW);

__device__ __forceinline__ int akz_clamp(int v, int n) { return min(max(v, 0), n - 1); }
__device__ __forceinline__ int akz_reflect(int p, int n) {  // BORDER_REFLECT_101, any distance
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// BORDER_REFLECT_101 for -n < p < 2n - 1 (every row k_akz_dhess USES: its strip +- 2S rows, levels of at least 20 rows): one fold, no
// loop.  Rows it only prefetches past the end of a strip can lie further out: clamped, so that the address stays inside the plane.
__device__ __forceinline__ int akz_reflect1(int p, int n) {
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * n - 2 - p : p;
    return min(max(p, 0), n - 1);
}

// ---- helpers of the lane = column kernels (k_akz_contrast_modg, k_akz_fed_gauss) ----
typedef float akz_f2 __attribute__((ext_vector_type(2)));
typedef float akz_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float akz_from_left(float v) {   // lane i <- lane i - 1 (DPP wave_shr:1; lane 0 reads 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float akz_from_right(float v) {  // lane i <- lane i + 1 (DPP wave_shl:1; lane 63 reads 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// NOUT outputs of the symmetric 5-tap filter from a window of NOUT + 4 values, two at a time: out[k] = k0 v[k+2] + k1 (v[k+3] + v[k+1]) + k2 (v[k+4] + v[k])
template <int NOUT>
__device__ __forceinline__ void akz_gauss5_window(const float *v, float k0, float k1, float k2, float *out) {
#pragma unroll
    for (int k = 0; k < NOUT; k += 2) {
        const akz_f2 c = {v[k + 2], v[k + 3]}, p1 = {v[k + 3], v[k + 4]}, m1 = {v[k + 1], v[k + 2]}, p2 = {v[k + 4], v[k + 5]}, m2 = {v[k], v[k + 1]};
        akz_f2 a = k0 * c;
        a += k1 * (p1 + m1);
        a += k2 * (p2 + m2);
        out[k] = a.x;
        out[k + 1] = a.y;
    }
}

// stage a (AT_W + 2R) x (AT_H + 2R) float tile around (x0, y0); REFLECT: reflect-101 coordinates, else clamped (replicate)
template <int R, bool REFLECT>
__device__ __forceinline__ void akz_stage(const float *__restrict__ src, int w, int h, int x0, int y0, float *lds) {
    constexpr int LW = AT_W + 2 * R, LH = AT_H + 2 * R;
    for (int i = threadIdx.x; i < LW * LH; i += AKZ_T) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = REFLECT ? akz_reflect(x0 - R + lx, w) : akz_clamp(x0 - R + lx, w);
        const int gy = REFLECT ? akz_reflect(y0 - R + ly, h) : akz_clamp(y0 - R + ly, h);
        lds[i] = src[(size_t)gy * w + gx];
    }
}

// ---- convertTo(CV_32F, 1/255) + GaussianBlur(ksize, sigma, BORDER_REPLICATE) ----
// U8: source is the gray image (converted on the fly); else a float plane.  Symmetric separable filter, rows then columns,
// s = k[r] * c + sum_j k[r + j] * (S[+j] + S[-j]).
template <int R, bool U8>
__global__ __launch_bounds__(AKZ_T) void k_akz_gauss(const void *__restrict__ src_v, int src_stride, size_t src_frame_stride, int w, int h,
                                                     int nframes, const float *__restrict__ taps, float *__restrict__ dst) {
    constexpr int LW = AT_W + 2 * R, LH = AT_H + 2 * R;
    constexpr int LP = LW | 1;  // odd LDS pitch: the row pass walks down rows lane by lane
    constexpr int RUN = 8;      // outputs per register window: a tap is read from LDS once per run, not once per output
    __shared__ float s_in[LP * LH];
    __shared__ float s_row[AT_W * LH];
    AKZ_TILE(AT_W, AT_H)
    float k[R + 1];  // k[j] = tap at distance j
#pragma unroll
    for (int j = 0; j <= R; ++j) k[j] = taps[R + j];
    if (U8) {
        const uint8_t *src = reinterpret_cast<const uint8_t *>(src_v) + (size_t)f * src_frame_stride;
        const float a = (float)(1.0 / 255.0);
        if (LW % 4 == 0 && x0 - R >= 0 && x0 - R + LW <= w) {
            // no column of the tile is clamped: four pixels per load (LW = 64 + 2R is a multiple of 4 for even R)
            for (int i = threadIdx.x; i < (LW / 4) * LH; i += AKZ_T) {
                const int ly = i / (LW / 4), q = i - ly * (LW / 4);
                const int gy = akz_clamp(y0 - R + ly, h);
                uint32_t v;
                __builtin_memcpy(&v, src + (size_t)gy * src_stride + (x0 - R + 4 * q), 4);
                float *d = &s_in[ly * LP + 4 * q];
                d[0] = (float)(v & 0xffu) * a;
                d[1] = (float)((v >> 8) & 0xffu) * a;
                d[2] = (float)((v >> 16) & 0xffu) * a;
                d[3] = (float)(v >> 24) * a;
            }
        } else {
            for (int i = threadIdx.x; i < LW * LH; i += AKZ_T) {
                const int ly = i / LW, lx = i - ly * LW;
                const int gx = akz_clamp(x0 - R + lx, w), gy = akz_clamp(y0 - R + ly, h);
                s_in[ly * LP + lx] = (float)src[(size_t)gy * src_stride + gx] * a;
            }
        }
    } else {
        const float *src = reinterpret_cast<const float *>(src_v) + (size_t)f * src_frame_stride;
        for (int i = threadIdx.x; i < LW * LH; i += AKZ_T) {
            const int ly = i / LW, lx = i - ly * LW;
            const int gx = akz_clamp(x0 - R + lx, w), gy = akz_clamp(y0 - R + ly, h);
            s_in[ly * LP + lx] = src[(size_t)gy * w + gx];
        }
    }
    __syncthreads();
    // rows: item = (tile row, run of RUN columns)
    for (int i = threadIdx.x; i < LH * (AT_W / RUN); i += AKZ_T) {
        const int g = i / LH, ly = i - g * LH;
        const float *c = &s_in[ly * LP + g * RUN];
        float v[RUN + 2 * R];
#pragma unroll
        for (int t = 0; t < RUN + 2 * R; ++t) v[t] = c[t];
#pragma unroll
        for (int t = 0; t < RUN; ++t) {
            float a = k[0] * v[t + R];
#pragma unroll
            for (int j = 1; j <= R; ++j) a += k[j] * (v[t + R + j] + v[t + R - j]);
            s_row[ly * AT_W + g * RUN + t] = a;
        }
    }
    __syncthreads();
    // columns: item = (tile column, run of RUN rows): AT_W x AT_H / RUN items == AKZ_T
    float *out = dst + (size_t)f * w * h;
    for (int i = threadIdx.x; i < AT_W * (AT_H / RUN); i += AKZ_T) {
        const int g = i / AT_W, lx = i - g * AT_W;
        const float *c = &s_row[g * RUN * AT_W + lx];
        float v[RUN + 2 * R];
#pragma unroll
        for (int t = 0; t < RUN + 2 * R; ++t) v[t] = c[t * AT_W];
        const int gx = x0 + lx;
#pragma unroll
        for (int t = 0; t < RUN; ++t) {
            float a = k[0] * v[t + R];
#pragma unroll
            for (int j = 1; j <= R; ++j) a += k[j] * (v[t + R + j] + v[t + R - j]);
            const int gy = y0 + g * RUN + t;
            if (gx < w && gy < h) out[(size_t)gy * w + gx] = a;
        }
    }
}

// cv::Scharr 3x3 on an LDS tile with pitch LW, centre pointer c
__device__ __forceinline__ float akz_scharr_x(const float *c, int LW) {
    const float t0 = c[-LW + 1] - c[-LW - 1], t1 = c[1] - c[-1], t2 = c[LW + 1] - c[LW - 1];
    return 10.0f * t1 + 3.0f * (t0 + t2);
}
__device__ __forceinline__ float akz_scharr_y(const float *c, int LW) {
    const float u0 = 10.0f * c[-LW] + 3.0f * (c[-LW - 1] + c[-LW + 1]);
    const float u2 = 10.0f * c[LW] + 3.0f * (c[LW - 1] + c[LW + 1]);
    return u2 - u0;
}

// ---- compute_k_percentile, pass 1: gradient magnitude of the sigma = 1 smoothed image over the interior; frame maximum ----
__global__ __launch_bounds__(AKZ_T) void k_akz_modg(const float *__restrict__ gsm, int w, int h, int nframes, float *__restrict__ modg,
                                                    unsigned int *__restrict__ hmax_bits) {
    constexpr int LW = AT_W + 2, LH = AT_H + 2;
    __shared__ float s_in[LW * LH];
    __shared__ unsigned int s_max;
    AKZ_TILE(AT_W, AT_H)
    if (threadIdx.x == 0) s_max = 0;
    akz_stage<1, true>(gsm + (size_t)f * w * h, w, h, x0, y0, s_in);
    __syncthreads();
    float *out = modg + (size_t)f * w * h;
    unsigned int mx = 0;
    for (int i = threadIdx.x; i < AT_W * AT_H; i += AKZ_T) {
        const int ly = i / AT_W, lx = i - ly * AT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            float m = 0.0f;
            if (gx >= 1 && gx < w - 1 && gy >= 1 && gy < h - 1) {  // the histogram skips the 1 px border
                const float *c = &s_in[(ly + 1) * LW + lx + 1];
                const float lxv = akz_scharr_x(c, LW), lyv = akz_scharr_y(c, LW);
                m = sqrtf(lxv * lxv + lyv * lyv);
            }
            out[(size_t)gy * w + gx] = m;
            mx = max(mx, __float_as_uint(m));  // m >= 0: the bit pattern orders like the value
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x == 0 && s_max) atomicMax(&hmax_bits[f], s_max);
}

// pass 1 without the detour through HBM: the sigma = 1 image (k_akz_gauss<2, true>'s expressions: convertTo(1/255) + 5 x 5 Gaussian,
// BORDER_REPLICATE) of the tile and one ring around it never leaves the workgroup, k_akz_modg's expressions run on it.  Only pixels whose
// whole 3 x 3 neighbourhood is inside the image get a magnitude, so nothing of the ring that lies outside the image is ever used.
// Shape (round 4, after k_akz_fed_gauss): 64 x 32 tile, lane = tile column, a wavefront owns a band of 8 rows; the row pass works on
// (source row, run of 16 columns) items with 128-bit LDS reads and packed fp32 arithmetic and writes its result TRANSPOSED, so that the
// column pass is four 128-bit reads per lane and leaves the 10 smoothed rows a band's Scharr stencils need in registers; the left /
// right neighbours are DPP shifts, the two ring columns (tile columns -1 and 64) a small job of 20 lanes per wavefront.  (The first
// form staged, filtered and differentiated through three LDS planes with run-time item mappings: 183 us per 64 frames of 1280 x 720.)
#define AKZ_CM_PS 72   // source tile pitch in BYTES (columns x0 - 4 .. x0 + 67 as 18 dwords)
#define AKZ_CM_PT 44   // transposed row-pass plane: [column 0 .. 65][source row 0 .. 37]; 44 mod 32 = 12 as above
#define AKZ_CM_ROWS 38
__global__ __launch_bounds__(AKZ_T) void k_akz_contrast_modg(const uint8_t *__restrict__ gray, int src_stride, size_t src_frame_stride, int w, int h,
                                                             int nframes, const float *__restrict__ taps, float *__restrict__ modg,
                                                             unsigned int *__restrict__ hmax_bits) {
    constexpr int PS = AKZ_CM_PS, PT = AKZ_CM_PT, SR = AKZ_CM_ROWS;
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SR * PS];  // source rows y0 - 3 .., byte column = image column - (x0 - 4) (replicate coordinates)
    __shared__ __attribute__((aligned(16))) float s_t[66 * PT];      // row-pass result of tile column c - 1 at [c][source row]
    __shared__ float s_ring[4][2][10];
    __shared__ unsigned int s_max;
    AKZ_TILE(AT_W, AT_H)
    const int tid = threadIdx.x, tx = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6) & 3;
    const int r0 = wv * 8;
    if (tid == 0) s_max = 0;
    const float k0 = taps[2], k1 = taps[3], k2 = taps[4];
    const uint8_t *src = gray + (size_t)f * src_frame_stride;
    const float a = (float)(1.0 / 255.0);
    // ---- staging: the u8 pixels as they are (2.7 KB instead of 11.5 KB of floats: more workgroups per CU); converted where they are read
    if (x0 - 4 >= 0 && x0 + 68 <= w) {  // no column clamped (all but the first and last tile column): 38 rows x 18 aligned dwords from x0 - 4
        uint32_t v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * AKZ_T;
            const int py = min(i / 18, SR - 1), q = i - (i / 18) * 18;
            v[k] = *reinterpret_cast<const uint32_t *>(src + (size_t)akz_clamp(y0 - 3 + py, h) * src_stride + (x0 - 4 + 4 * q));
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * AKZ_T;
            const int py = i / 18, q = i - py * 18;
            if (py < SR) *reinterpret_cast<uint32_t *>(&s_src[py * PS + 4 * q]) = v[k];
        }
    } else {  // replicate coordinates: byte gathers, columns x0 - 3 .. x0 + 66 at byte columns 1 .. 70
        const unsigned cxb = (unsigned)akz_clamp(x0 - 3 + tx, w);
        uint8_t v[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int py = wv + 4 * k;
            if (py < SR) v[k] = (src + (size_t)akz_clamp(y0 - 3 + py, h) * src_stride)[cxb];
        }
        uint8_t e = 0;
        const int ey = tid / 6, ex = tid - ey * 6;  // the six columns 64 .. 69: one element per thread (228 of them)
        if (tid < SR * 6) e = src[(size_t)akz_clamp(y0 - 3 + ey, h) * src_stride + akz_clamp(x0 - 3 + 64 + ex, w)];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int py = wv + 4 * k;
            if (py < SR) s_src[py * PS + 1 + tx] = v[k];
        }
        if (tid < SR * 6) s_src[ey * PS + 1 + 64 + ex] = e;
    }
    __syncthreads();
    // ---- row pass: items 0 .. 151 = (source row, run of 16 output columns 16 g .. 16 g + 15), items 152 .. 189 = the columns 64, 65 of a row.
    //      Window value j = source column 16 g + j = byte 16 g + j + 1 of the row: convertTo(CV_32F, 1 / 255) as (float)byte * a
    if (tid < SR * 4) {
        const int g = tid / SR, py = tid - g * SR;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(&s_src[py * PS + 16 * g]);
        uint32_t d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = q[k];
        float v[22], o[16];
#pragma unroll
        for (int j = 0; j < 20; ++j) v[j] = (float)((d[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu) * a;
        v[20] = v[21] = 0.0f;
        akz_gauss5_window<16>(v, k0, k1, k2, o);
        float *dst = &s_t[(16 * g) * PT + py];
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[k * PT] = o[k];
    } else if (tid < SR * 5) {
        const int py = tid - SR * 4;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(&s_src[py * PS + 64]);
        const uint32_t d0 = q[0], d1 = q[1];  // bytes 64 .. 71: source columns 63 .. 70
        float v[8], o[2];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = (float)(((j + 1 < 4 ? d0 : d1) >> (8 * ((j + 1) & 3))) & 0xffu) * a;
        v[6] = v[7] = 0.0f;
        akz_gauss5_window<2>(v, k0, k1, k2, o);
        s_t[64 * PT + py] = o[0];
        s_t[65 * PT + py] = o[1];
    }
    __syncthreads();
    // ---- column pass: smoothed rows r0 - 1 .. r0 + 8 of tile column tx (row-pass column tx + 1; source rows r0 .. r0 + 13)
    float vc[10], vl[10], vr[10];
    {
        const akz_f4 *q = reinterpret_cast<const akz_f4 *>(&s_t[(tx + 1) * PT + r0]);
        float v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const akz_f4 t = q[k];
            v[4 * k] = t.x, v[4 * k + 1] = t.y, v[4 * k + 2] = t.z, v[4 * k + 3] = t.w;
        }
        akz_gauss5_window<10>(v, k0, k1, k2, vc);
    }
    if (tx < 20) {  // the ring columns -1 (row-pass column 0) and 64 (column 65), ten rows each
        const int side = tx >= 10, m = tx - 10 * side;
        const float *c = &s_t[(side ? 65 : 0) * PT + r0 + m];
        float rr = k0 * c[2];
        rr += k1 * (c[3] + c[1]);
        rr += k2 * (c[4] + c[0]);
        s_ring[wv][side][m] = rr;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // same wavefront: LDS operations execute in order, the compiler must not reorder
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        const float dl = akz_from_left(vc[m]), dr = akz_from_right(vc[m]);
        vl[m] = tx == 0 ? s_ring[wv][0][m] : dl;
        vr[m] = tx == 63 ? s_ring[wv][1][m] : dr;
    }
    // ---- Scharr + magnitude of the band rows (k_akz_modg's expressions), frame maximum
    float *out = modg + (size_t)f * w * h;
    const int gx = x0 + tx;
    const bool col_ok = gx >= 1 && gx < w - 1;
    unsigned int mx = 0;
    float t[10], u[10];
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        t[m] = vr[m] - vl[m];
        u[m] = 10.0f * vc[m] + 3.0f * (vl[m] + vr[m]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int gy = y0 + r0 + j;
        if (gx < w && gy < h) {
            float mg = 0.0f;
            if (col_ok && gy >= 1 && gy < h - 1) {  // the histogram skips the 1 px border
                const float lxv = 10.0f * t[j + 1] + 3.0f * (t[j] + t[j + 2]);
                const float lyv = u[j + 2] - u[j];
                mg = sqrtf(lxv * lxv + lyv * lyv);
            }
            out[(size_t)gy * w + gx] = mg;
            mx = max(mx, __float_as_uint(mg));  // mg >= 0: the bit pattern orders like the value
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64));
    if (tx == 0) atomicMax(&s_max, mx);
    __syncthreads();
    if (tid == 0 && s_max) atomicMax(&hmax_bits[f], s_max);
}

// pass 2: histogram of the non-zero magnitudes (bins depend on the frame maximum)
__global__ __launch_bounds__(AKZ_T) void k_akz_hist(const float *__restrict__ modg, int w, int h, const unsigned int *__restrict__ hmax_bits,
                                                    int nbins, int *__restrict__ hist /* [frame][nbins + 1]: bins, then npoints */) {
    extern __shared__ int s_hist[];
    const int f = blockIdx.y;
    for (int i = threadIdx.x; i <= nbins; i += AKZ_T) s_hist[i] = 0;
    __syncthreads();
    const float hmax = __uint_as_float(hmax_bits[f]);
    const float *src = modg + (size_t)f * w * h;
    const int n = w * h;
    int npoints = 0;  // counted in a register: one LDS atomic per wavefront at the end instead of one per pixel
    // four loads in flight per thread: the pass waits on memory (one dependent load per iteration ran at 2.5 TB/s)
    const int stride = (int)gridDim.x * AKZ_T;
    for (int i0 = blockIdx.x * AKZ_T + threadIdx.x; i0 < n; i0 += 4 * stride) {
        float m4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m4[k] = i0 + k * stride < n ? src[i0 + k * stride] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float m = m4[k];
            if (m != 0.0f) {
                int nbin = (int)floorf((float)nbins * (m / hmax));
                if (nbin == nbins) nbin--;
                atomicAdd(&s_hist[nbin], 1);
                ++npoints;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) npoints += __shfl_xor(npoints, o, 64);
    if ((threadIdx.x & 63) == 0 && npoints) atomicAdd(&s_hist[nbins], npoints);
    __syncthreads();
    for (int i = threadIdx.x; i <= nbins; i += AKZ_T)
        if (s_hist[i]) atomicAdd(&hist[(size_t)f * (nbins + 1) + i], s_hist[i]);
}

// percentile walk (one thread per frame; 300 bins)
__global__ void k_akz_kperc(const int *__restrict__ hist, const unsigned int *__restrict__ hmax_bits, int nbins, float perc, int nframes,
                            float *__restrict__ kcontrast) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    const int *hh = hist + (size_t)f * (nbins + 1);
    const float hmax = __uint_as_float(hmax_bits[f]);
    const int nthreshold = (int)((float)hh[nbins] * perc);
    int k = 0, nelements = 0;
    for (k = 0; nelements < nthreshold && k < nbins; ++k) nelements += hh[k];
    kcontrast[f] = (nelements < nthreshold || hmax == 0.0f) ? 0.03f : hmax * ((float)k / (float)nbins);
}

// ---- halfsample_image (INTER_AREA, exact factor 2) ----
__global__ __launch_bounds__(AKZ_T) void k_akz_halfsample(const float *__restrict__ src, int w, int h, float *__restrict__ dst, int dw, int dh) {
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const float *r0 = src + (size_t)f * w * h + (size_t)(2 * y) * w + 2 * x, *r1 = r0 + w;
    const float2 a = *reinterpret_cast<const float2 *>(r0), b = *reinterpret_cast<const float2 *>(r1);
    dst[(size_t)f * dw * dh + (size_t)y * dw + x] = ((a.x + a.y) + (b.x + b.y)) * 0.25f;
}

// ---- image_derivatives_scharr x 2 + pm_g2 ----
__global__ __launch_bounds__(AKZ_T) void k_akz_flow(const float *__restrict__ lsm, int w, int h, int nframes, const float *__restrict__ kcontrast,
                                                    int octave, float *__restrict__ flow) {
    constexpr int LW = AT_W + 2, LH = AT_H + 2;
    __shared__ float s_in[LW * LH];
    AKZ_TILE(AT_W, AT_H)
    akz_stage<1, true>(lsm + (size_t)f * w * h, w, h, x0, y0, s_in);
    __syncthreads();
    float k = kcontrast[f];
    for (int i = 0; i < octave; ++i) k = k * 0.75f;  // `options_.kcontrast *= 0.75` once per octave change
    const float k2inv = 1.0f / (k * k);
    float *out = flow + (size_t)f * w * h;
    for (int i = threadIdx.x; i < AT_W * AT_H; i += AKZ_T) {
        const int ly = i / AT_W, lx = i - ly * AT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const float *c = &s_in[(ly + 1) * LW + lx + 1];
            const float lxv = akz_scharr_x(c, LW), lyv = akz_scharr_y(c, LW);
            out[(size_t)gy * w + gx] = 1.0f / (1.0f + (lxv * lxv + lyv * lyv) * k2inv);
        }
    }
}

// ---- nld_step_scalar: one explicit diffusion step, zero flux across the image border ----
__global__ __launch_bounds__(AKZ_T) void k_akz_nld_step(const float *__restrict__ Lt, const float *__restrict__ flow, int w, int h, int nframes, float tau,
                                                        float *__restrict__ out) {
    constexpr int LW = AT_W + 2, LH = AT_H + 2;
    __shared__ float s_l[LW * LH], s_c[LW * LH];
    AKZ_TILE(AT_W, AT_H)
    akz_stage<1, false>(Lt + (size_t)f * w * h, w, h, x0, y0, s_l);
    akz_stage<1, false>(flow + (size_t)f * w * h, w, h, x0, y0, s_c);
    __syncthreads();
    const double hs = 0.5 * (double)tau;
    float *o = out + (size_t)f * w * h;
    for (int i = threadIdx.x; i < AT_W * AT_H; i += AKZ_T) {
        const int ly = i / AT_W, lx = i - ly * AT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const int p = (ly + 1) * LW + lx + 1;
            const float L = s_l[p], c = s_c[p];
            const float xpos = gx + 1 < w ? (c + s_c[p + 1]) * (s_l[p + 1] - L) : 0.0f;
            const float xneg = gx > 0 ? (s_c[p - 1] + c) * (L - s_l[p - 1]) : 0.0f;
            const float ypos = gy + 1 < h ? (c + s_c[p + LW]) * (s_l[p + LW] - L) : 0.0f;
            const float yneg = gy > 0 ? (s_c[p - LW] + c) * (L - s_l[p - LW]) : 0.0f;
            const float sum = ((xpos - xneg) + ypos) - yneg;
            o[(size_t)gy * w + gx] = L + (float)(hs * (double)sum);
        }
    }
}

// ---- fused level update: Lsmooth (5-tap Gaussian) + pm_g2 conductivity + the whole FED cycle (N <= 8 steps) in one launch ----
#define AKZ_FED_MAX 8
struct AkzTau {
    float t[AKZ_FED_MAX];
};
#define AKZ_FT 512

// Shaped by what rocprofv3 said about its first form (round 3's k_akz_fed_fused: a (64 - 2N) x 48 tile, the Gaussian as LDS-tiled row /
// column passes with run-time item mappings, the conductivity and the FED steps per band of 8 rows in registers): that one was bound by
// the vector ALU, and most of what it issued was not arithmetic: 5.0 wavefront instructions per pixel
// (322 lane operations) for about 100 of filter + conductivity + FED arithmetic - per-element clamps and 64-bit addresses in the
// staging loop, run-time divisions in the item mappings, nine predicated stores per run, a six-term predicate per Lsmooth store.
// This form has the step count as a template parameter and keeps everything per-row on the scalar unit:
//   * tile = (62 - 2N) x (64 - 2N) outputs; lane = tile column (lanes 0 and 63 only carry the Lsmooth ring), wavefront = band of 8 rows,
//     every wavefront busy in every phase;
//   * staging: a wavefront fetches whole rows (row clamp on the scalar unit, one column clamp per lane), all loads in flight before the
//     first LDS write;
//   * row pass: item = (source row, run of 16 columns): five 128-bit LDS reads, packed fp32 arithmetic (two outputs per instruction,
//     same expression order), written TRANSPOSED so that
//   * the column pass is four 128-bit LDS reads per lane and leaves the 12 Lsmooth rows a band's conductivity needs in registers - no
//     Lsmooth plane in LDS, no reflect fix-up pass (the two rows / columns just outside the image are patched in registers);
//   * conductivity and FED steps on row pairs (packed fp32); barriers wait for LDS only, so the Lsmooth stores drain behind the FED cycle.
// Every pixel sees exactly the per-step arithmetic of k_akz_gauss / k_akz_flow / k_akz_nld_step (zero flux across the image border;
// out-of-image halo cells never reach an output), so the result is bit-identical to the step-by-step path (-ffp-contract=off); only the
// traffic changes: 12 B/px per level instead of 8 + 8 + 12 B/px per step.
#define AKZ_G_PS 76  // source tile pitch: 68 columns used; 76 mod 32 = 12 spreads the 128-bit reads of lanes that differ in the row
#define AKZ_G_PT 76  // transposed row-pass plane: [column][source row + 1], 72 entries used
#define AKZ_G_ROWS 70
#define AKZ_G_LDS_FLOATS (AKZ_G_ROWS * AKZ_G_PS + 64 * AKZ_G_PT)  // 40736 B: four workgroups per CU
#define AKZ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int N>
__global__ __launch_bounds__(AKZ_FT, 8) void k_akz_fed_gauss(const float *__restrict__ Lt_in, float *__restrict__ lsm, const float *__restrict__ taps, int w,
                                                          int h, int nframes, const float *__restrict__ kcontrast, int octave, AkzTau tau,
                                                          float *__restrict__ Lt_out) {
    extern __shared__ float s_fed[];
    constexpr int OW = 62 - 2 * N, FH = 64 - 2 * N, PS = AKZ_G_PS, PT = AKZ_G_PT, SR = AKZ_G_ROWS;
    float *s_src = s_fed;            // [70][PS]: source rows y0 - N - 3 .., columns x0 - N - 3 .. (replicate coordinates)
    float *s_t = s_fed + SR * PS;    // [64][PT]: row-pass result of tile column c, source row py at [c][py + 1]
    float *s_x = s_src;              // band boundary rows: [2 parities][top, bottom][8 waves][64], on top of the source tile (dead by then)
    AKZ_TILE(OW, FH)
    const int tid = threadIdx.x, tx = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6) & 7;  // & 7: a known range removes the row tests below
    const int r0 = wv * 8;
    const int gx = x0 - N - 1 + tx;  // image column of tile column tx
    const int ty0 = y0 - N;          // image row of tile row 0
    const float *pin = Lt_in + (size_t)f * w * h;
    {   // ---- staging: 70 rows x 68 columns
        const unsigned cxb = (unsigned)akz_clamp(x0 - N - 3 + tx, w) * 4u;  // byte offset inside a row: 32 bits, zero-extended by the load
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int py = wv + 8 * k;
            if (py < SR)  // row pointer on the scalar unit + lane offset
                v[k] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(pin + (size_t)akz_clamp(ty0 - 3 + py, h) * w) + cxb);
        }
        float e = 0.0f;
        if (tid < SR * 4) e = pin[(size_t)akz_clamp(ty0 - 3 + (tid >> 2), h) * w + akz_clamp(x0 - N - 3 + 64 + (tid & 3), w)];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int py = wv + 8 * k;
            if (py < SR) s_src[py * PS + tx] = v[k];
        }
        if (tid < SR * 4) s_src[(tid >> 2) * PS + 64 + (tid & 3)] = e;
    }
    const float k0 = taps[2], k1 = taps[3], k2 = taps[4];
    float kc = kcontrast[f];
    for (int i = 0; i < octave; ++i) kc = kc * 0.75f;
    const float k2inv = 1.0f / (kc * kc);
    AKZ_LDS_BARRIER();
    float L[8];  // the band's own pixels of Lt: tile row r0 + j = source row r0 + j + 3, tile column tx = source column tx + 2
#pragma unroll
    for (int j = 0; j < 8; ++j) L[j] = s_src[(r0 + j + 3) * PS + tx + 2];
    if (tid < SR * 4) {  // ---- row pass: item = (source row, run of 16 tile columns)
        const int g = tid / SR, py = tid - g * SR;
        const akz_f4 *q = reinterpret_cast<const akz_f4 *>(&s_src[py * PS + 16 * g]);
        float v[22], o[16];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const akz_f4 t = q[k];
            v[4 * k] = t.x, v[4 * k + 1] = t.y, v[4 * k + 2] = t.z, v[4 * k + 3] = t.w;
        }
        v[20] = v[21] = 0.0f;
        akz_gauss5_window<16>(v, k0, k1, k2, o);
        float *d = &s_t[(16 * g) * PT + py + 1];
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k * PT] = o[k];
    }
    AKZ_LDS_BARRIER();
    // ---- column pass: Lsmooth of tile rows r0 - 2 .. r0 + 9, tile column tx
    float vc[12], vl[12], vr[12];
    {
        const akz_f4 *q = reinterpret_cast<const akz_f4 *>(&s_t[tx * PT + r0]);
        float v[18];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const akz_f4 t = q[k];
            v[4 * k] = t.x, v[4 * k + 1] = t.y, v[4 * k + 2] = t.z, v[4 * k + 3] = t.w;
        }
        v[16] = v[17] = 0.0f;
        akz_gauss5_window<12>(v, k0, k1, k2, vc);
    }
    const bool lane_out = tx > N && tx <= N + OW && gx < w;
    {   // this tile's Lsmooth goes out for the derivative kernels
        float *pl = lsm + (size_t)f * w * h + gx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j, gy = ty0 + r;
            if (r >= N && r < N + FH && gy < h && lane_out) pl[(size_t)gy * w] = vc[j + 2];
        }
    }
    // the conductivity stencil reads Lsmooth at BORDER_REFLECT_101 coordinates: a pixel inside the image reaches one row / column
    // outside it, whose mirror image is two rows / columns further in (held by the same lane / the neighbour on the other side)
    const int m_top = -1 - (ty0 + r0 - 2), m_bot = h - (ty0 + r0 - 2);  // window index of image rows -1 and h
    if (m_top >= 0 || m_bot < 12) {
#pragma unroll
        for (int m = 0; m < 10; ++m)
            if (m == m_top) vc[m] = vc[m + 2];
#pragma unroll
        for (int m = 2; m < 12; ++m)
            if (m == m_bot) vc[m] = vc[m - 2];
    }
#pragma unroll
    for (int m = 0; m < 12; ++m) {
        vl[m] = akz_from_left(vc[m]);
        vr[m] = akz_from_right(vc[m]);
    }
    if (x0 - N - 1 <= 0 || x0 - N - 1 + 64 >= w) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            const float a = vl[m], b = vr[m];
            vl[m] = gx == 0 ? b : a;
            vr[m] = gx == w - 1 ? a : b;
        }
    }
    // ---- conductivity of tile rows r0 - 1 .. r0 + 8 (pm_g2), Scharr on the 12-row window
    float c[10];
    {
        float t[12], u[12];
#pragma unroll
        for (int m = 0; m < 12; m += 2) {
            const akz_f2 l2 = {vl[m], vl[m + 1]}, r2 = {vr[m], vr[m + 1]}, c2 = {vc[m], vc[m + 1]};
            const akz_f2 t2 = r2 - l2, u2 = 10.0f * c2 + 3.0f * (l2 + r2);
            t[m] = t2.x, t[m + 1] = t2.y, u[m] = u2.x, u[m + 1] = u2.y;
        }
#pragma unroll
        for (int q = 0; q < 10; q += 2) {
            const akz_f2 ta = {t[q], t[q + 1]}, tb = {t[q + 1], t[q + 2]}, tc = {t[q + 2], t[q + 3]};
            const akz_f2 ua = {u[q], u[q + 1]}, uc = {u[q + 2], u[q + 3]};
            const akz_f2 lxv = 10.0f * tb + 3.0f * (ta + tc);
            const akz_f2 lyv = uc - ua;
            const akz_f2 d = 1.0f + (lxv * lxv + lyv * lyv) * k2inv;
            c[q] = akz_recip_ge1(d.x);
            c[q + 1] = akz_recip_ge1(d.y);
        }
    }
    // Flux form of the step: xpos of a pixel is xneg of its right neighbour and ypos is yneg of the pixel below - the same product of
    // the same operands ((c_a + c_b) * (L_b - L_a)), so every edge is evaluated once: eX[j] = the conductivity sum of the edge to the
    // right of this lane in band row j, eY[k] = of the edge above band row k (k = 8: below row 7).  An edge that leaves the image has
    // coefficient 0 (that flux term is 0, as upstream); pixels outside the image and the tile's outermost lanes / rows pick up fluxes
    // they should not, which only makes values wrong that no output depends on (the halo argument of the step-by-step kernel).
    const bool x_in = gx >= 0 && gx < w;
    const bool x_edge = gx >= 0 && gx + 1 < w;
    float eX[8];
    akz_f2 eYa[4], eYb[4];  // {eY[2p], eY[2p + 1]}, {eY[2p + 1], eY[2p + 2]}
    {
        float eY[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int gyb = ty0 + r0 + k;  // image row below the edge
            eY[k] = x_in && gyb >= 1 && gyb < h ? c[k] + c[k + 1] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gy = ty0 + r0 + j;
            const float cc = c[j + 1];
            const float cr = cc + akz_from_right(cc);  // outside the select: a lane-crossing read must not sit under a divergent condition
            eX[j] = x_edge && gy >= 0 && gy < h ? cr : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) eYa[p] = akz_f2{eY[2 * p], eY[2 * p + 1]}, eYb[p] = akz_f2{eY[2 * p + 1], eY[2 * p + 2]};
    }
#pragma unroll
    for (int st = 0; st < N; ++st) {
        float *xb = s_x + (st & 1) * (2 * 8 * 64);
        xb[wv * 64 + tx] = L[0];
        xb[8 * 64 + wv * 64 + tx] = L[7];
        AKZ_LDS_BARRIER();
        const float up_halo = xb[8 * 64 + max(wv - 1, 0) * 64 + tx];  // bands 0 / 7: their outer rows are never valid
        const float dn_halo = xb[min(wv + 1, 7) * 64 + tx];
        // upstream forms 0.5 * stepsize * sum in double; 0.5 * tau is exact in float and a float x float product rounded once from
        // double equals the IEEE float product, so the float multiply below is bit-identical (the tests compare with the double
        // formulation of the step-by-step kernel and of the oracle)
        const float hs = 0.5f * tau.t[st];
        float xd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float fx = eX[j] * (akz_from_right(L[j]) - L[j]);  // xpos
            xd[j] = fx - akz_from_left(fx);                           // xpos - xneg
        }
        float nl[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int j = 2 * p;
            const akz_f2 Lc = {L[j], L[j + 1]}, U2 = {j > 0 ? L[j - 1] : up_halo, L[j]}, D2 = {L[j + 1], j < 6 ? L[j + 2] : dn_halo};
            const akz_f2 ga = eYa[p] * (Lc - U2);  // yneg of rows j, j + 1
            const akz_f2 gb = eYb[p] * (D2 - Lc);  // ypos
            const akz_f2 x2 = {xd[j], xd[j + 1]};
            const akz_f2 sum = (x2 + gb) - ga;
            const akz_f2 n2 = Lc + hs * sum;
            nl[j] = n2.x, nl[j + 1] = n2.y;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) L[j] = nl[j];
    }
    if (lane_out) {
        float *o = Lt_out + (size_t)f * w * h + gx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = r0 + j, gy = ty0 + r;
            if (r >= N && r < N + FH && gy < h) o[(size_t)gy * w] = L[j];
        }
    }
}

// ---- Compute_Multiscale_Derivatives, first derivatives (unscaled): sparse 3-tap Scharr at distance s ----
#define AKZ_MAX_S 8
__global__ __launch_bounds__(AKZ_T) void k_akz_deriv1(const float *__restrict__ lsm, int w, int h, int nframes, int s, float *__restrict__ dx,
                                                      float *__restrict__ dy) {
    extern __shared__ float s_in[];  // (AT_W + 2s) x (AT_H + 2s)
    const int LW = AT_W + 2 * s, LH = AT_H + 2 * s;
    AKZ_TILE(AT_W, AT_H)
    const float *src = lsm + (size_t)f * w * h;
    for (int i = threadIdx.x; i < LW * LH; i += AKZ_T) {
        const int ly = i / LW, lx = i - ly * LW;
        s_in[i] = src[(size_t)akz_reflect(y0 - s + ly, h) * w + akz_reflect(x0 - s + lx, w)];
    }
    __syncthreads();
    const float wgt = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f)), mid = wgt * norm;
    for (int i = threadIdx.x; i < AT_W * AT_H; i += AKZ_T) {
        const int ly = i / AT_W, lx = i - ly * AT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const float *c = &s_in[(ly + s) * LW + lx + s];
            // Lx: row derivative, column smoothing
            const float t0 = c[-s * LW + s] - c[-s * LW - s], t1 = c[s] - c[-s], t2 = c[s * LW + s] - c[s * LW - s];
            const float vx = mid * t1 + norm * (t0 + t2);
            // Ly: row smoothing, column derivative
            const float u0 = mid * c[-s * LW] + norm * (c[-s * LW - s] + c[-s * LW + s]);
            const float u2 = mid * c[s * LW] + norm * (c[s * LW - s] + c[s * LW + s]);
            const size_t o = (size_t)f * w * h + (size_t)gy * w + gx;
            dx[o] = vx;
            dy[o] = u2 - u0;
        }
    }
}

// second derivatives from the unscaled first ones, the sigma_size normalisation and the determinant.  Upstream scales Lx / Ly by
// sigma_size in place AFTER this (Compute_Multiscale_Derivatives); here the per-level planes keep the UNSCALED first derivatives
// and every consumer multiplies by sigma_size as it reads (the same single float product) — the scaled copies were 8 B/px of
// pure re-write traffic.
__global__ __launch_bounds__(AKZ_T) void k_akz_hessian(const float *__restrict__ dx, const float *__restrict__ dy, int w, int h, int nframes, int s,
                                                       float *__restrict__ Ldet) {
    extern __shared__ float s_mem[];  // two (AT_W + 2s) x (AT_H + 2s) planes
    const int LW = AT_W + 2 * s, LH = AT_H + 2 * s;
    float *s_x = s_mem, *s_y = s_mem + LW * LH;
    AKZ_TILE(AT_W, AT_H)
    const float *px = dx + (size_t)f * w * h, *py = dy + (size_t)f * w * h;
    for (int i = threadIdx.x; i < LW * LH; i += AKZ_T) {
        const int ly = i / LW, lx = i - ly * LW;
        const size_t g = (size_t)akz_reflect(y0 - s + ly, h) * w + akz_reflect(x0 - s + lx, w);
        s_x[i] = px[g];
        s_y[i] = py[g];
    }
    __syncthreads();
    const float wgt = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f)), mid = wgt * norm;
    const float fs2 = (float)(s * s);
    for (int i = threadIdx.x; i < AT_W * AT_H; i += AKZ_T) {
        const int ly = i / AT_W, lx = i - ly * AT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < w && gy < h) {
            const int p = (ly + s) * LW + lx + s;
            const float *cx = &s_x[p], *cy = &s_y[p];
            // Lxx = d/dx of Lx: row derivative, column smoothing
            const float t0 = cx[-s * LW + s] - cx[-s * LW - s], t1 = cx[s] - cx[-s], t2 = cx[s * LW + s] - cx[s * LW - s];
            const float lxx = (mid * t1 + norm * (t0 + t2)) * fs2;
            // Lyy = d/dy of Ly: row smoothing, column derivative
            const float u0 = mid * cy[-s * LW] + norm * (cy[-s * LW - s] + cy[-s * LW + s]);
            const float u2 = mid * cy[s * LW] + norm * (cy[s * LW - s] + cy[s * LW + s]);
            const float lyy = (u2 - u0) * fs2;
            // Lxy = d/dy of Lx
            const float v0 = mid * cx[-s * LW] + norm * (cx[-s * LW - s] + cx[-s * LW + s]);
            const float v2 = mid * cx[s * LW] + norm * (cx[s * LW - s] + cx[s * LW + s]);
            const float lxy = (v2 - v0) * fs2;
            const size_t o = (size_t)f * w * h + (size_t)gy * w + gx;
            Ldet[o] = lxx * lyy - lxy * lxy;
        }
    }
}

// ---- first derivatives + Hessian determinant in one pass (sigma_size 2 .. 4: every level of the akaze61 settings) ----
// One wavefront per strip: lane = column, the rows slide through registers.  Every row of Lsmooth is loaded once (full lines), the
// +-s column neighbours come from ds_bpermute, the +-s row neighbours from register rings of depth 2s + 1 (the row loop is unrolled
// by the ring depth, so every ring index is a compile-time constant).  Four stages per loaded row v:
//   1. D[v] = S[x+s] - S[x-s], U[v] = mid * S[x] + norm * (S[x-s] + S[x+s])
//   2. row q = v - s:  Lx = mid * D[q] + norm * (D[q-s] + D[q+s]),  Ly = U[q+s] - U[q-s]            -> stored (unscaled)
//   3. the same two filters over Lx / Ly of row q (DX, UX, UY)
//   4. row p = q - s:  Lxx, Lyy, Lxy, Ldet                                                        -> stored
// which are k_akz_deriv1's and k_akz_hessian's expressions term for term (the two-kernel form stays for other sigma sizes and as
// the test reference).  Borders: those kernels read their inputs at BORDER_REFLECT_101 coordinates.  A lane on a column outside the
// image holds S at the reflected column, so its neighbours deliver exactly the reflected taps; what it computes there is the
// derivative "at a virtual position", and since the filters are odd / even under the reflection, Lx at a virtual COLUMN is exactly
// -Lx at the reflected column and Ly at a virtual ROW exactly -Ly at the reflected row (a - b == -(b - a), sums commute): two sign
// flips give stage 3 the values k_akz_hessian reads.  HBM bytes per pixel: 4 read (+ halo) + 12 written, against 12 + 12.
// What bounds it (round 4, all measured): the SHAPE of its memory accesses.  A 1-read : 3-write stream in strips of 56 / 52 / 48
// columns reaches 3.8 / 3.5 / 3.7 TB/s on this chip (248 / 271 / 255 us for one level of 64 frames) where full 256-byte rows reach
// 5.1 (184 us; tools/probes/probe_stream13.hip) - and this kernel takes 230 / 265-285 / 260 us.  The vector ALU is ~30 % busy; more
// wavefronts per SIMD (__launch_bounds__(256, 5) builds: 89 -> 70 and 111 -> 92 VGPRs without spills) and DPP shift chains instead of the
// crossbar change nothing.  A variant with 64-ALIGNED strips in which every lane loads its five taps x - 2S .. x + 2S itself and
// evaluates the first-derivative filters at x - S, x, x + S (no lane crossing, full-line stores, bit-identical) was built and is
// SLOWER: 285 / 305-320 / 366 us - five loads per row instead of one put the time into cache round trips.  Dropped.
#ifndef AKZ_DH_ROWS
#define AKZ_DH_ROWS 64  // rows per strip (+ 4S rows of run-in).  Round 4 sweep, 64 frames of 1280 x 720, all eight levels: 64 rows 1.36 ms,
                       // 96: 1.39, 128: 1.42, 180: 1.50, 240: 1.74, 360: 2.08 (fewer, longer wavefronts lose more than the run-in costs)
#endif
// SD: also store the (unscaled) first derivatives.  The pipeline does not (round 5): their only reader is the descriptor stage, which
// evaluates them at its sample positions from Lsmooth (k_akaze_desc.hip) - 8 of this kernel's 16 B per pixel were written for a plane
// of which under a tenth is ever read.  SD = true serves afv_akaze_get_plane (and keeps the fused Lx / Ly under the plane tests).
// At least 5 wavefronts per SIMD (round 5, late): left alone the register allocator takes 90 / 158 VGPRs for S = 3 / 4 (5 / 3 wavefronts), and
// the SQ counters had the S = 4 instantiation at 0.58 vector-busy with 2.7 resident wavefronts.  With the bound it takes 68 / 96 (7 / 5
// wavefronts; 20 bytes of scratch at S = 4): 97 -> 91 and 134 -> 108 us per level of 64 frames.  6 and 7 spill more than they hide
// (130 and 630 us at S = 4).
template <int S, bool SD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_akz_dhess(const float *__restrict__ lsm, int w, int h, int nframes, int band_rows, float *__restrict__ dx,
                                                   float *__restrict__ dy, float *__restrict__ Ldet) {
    constexpr int P = 2 * S + 1;     // ring depth
    constexpr int OW = 64 - 4 * S;   // output columns per strip
    const int lane = threadIdx.x & 63;
    const int nstr = (w + OW - 1) / OW, nband = (h + band_rows - 1) / band_rows;
    // the strip is a property of the wavefront: said so, its rows' addresses (reflection included) are computed on the scalar unit
    int id = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (id >= nstr * nband * nframes) return;
    const int f = id / (nstr * nband);
    id -= f * nstr * nband;
    const int band = id / nstr, x0 = (id - band * nstr) * OW, y0 = band * band_rows;
    const int gx = x0 - 2 * S + lane;
    const bool colv = gx < 0 || gx >= w;  // virtual column
    const int cx = akz_reflect(gx, w);
    const bool out_lane = lane >= 2 * S && lane < 64 - 2 * S && gx < w;
    const int am = ((lane - S) & 63) * 4, ap = ((lane + S) & 63) * 4;
    const float *src = lsm + (size_t)f * w * h;
    const size_t fo = (size_t)f * w * h;
    const float wgt = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)S * (wgt + 2.0f)), mid = wgt * norm;
    const float fs2 = (float)(S * S);
    const int rows = min(band_rows, h - y0);
    const int T = rows + 4 * S;  // Lsmooth rows y0 - 2S .. y0 + rows - 1 + 2S
    float D[P], U[P], DX[P], UX[P], UY[P];
#pragma unroll
    for (int k = 0; k < P; ++k) D[k] = U[k] = DX[k] = UX[k] = UY[k] = 0.0f;
    auto shl = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(am, __builtin_bit_cast(int, v))); };  // value of lane - S
    auto shr = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ap, __builtin_bit_cast(int, v))); };  // value of lane + S
    // the rows of the NEXT group of P are fetched while this group is worked on: P full lines in flight per wavefront
    float cur[P], nxt[P];
#pragma unroll
    for (int k = 0; k < P; ++k) cur[k] = src[(size_t)akz_reflect1(y0 - 2 * S + k, h) * w + cx];
    for (int t0 = 0; t0 < T; t0 += P) {
#pragma unroll
        for (int k = 0; k < P; ++k) nxt[k] = src[(size_t)akz_reflect1(y0 - 2 * S + t0 + P + k, h) * w + cx];
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int t = t0 + k;
            if (t >= T) break;
            const int km = (k + P - S) % P, kmm = (k + P - 2 * S) % P;  // ring slots of the rows S and 2S back
            const int v = y0 - 2 * S + t;
            // stage 1
            const float sc = cur[k];
            const float sm = shl(sc), sp = shr(sc);
            D[k] = sp - sm;
            U[k] = mid * sc + norm * (sm + sp);
            if (t < 2 * S) continue;
            // stage 2: row q
            const int q = v - S;
            const float lx = mid * D[km] + norm * (D[kmm] + D[k]);
            const float ly = U[k] - U[kmm];
            if (SD && q >= y0 && q < y0 + rows && out_lane) {
                const size_t o = fo + (size_t)q * w + gx;
                dx[o] = lx;
                dy[o] = ly;
            }
            const float lxf = colv ? -lx : lx;
            const float lyf = (q < 0 || q >= h) ? -ly : ly;
            // stage 3
            const float xm = shl(lxf), xp = shr(lxf), ym = shl(lyf), yp = shr(lyf);
            DX[k] = xp - xm;
            UX[k] = mid * lxf + norm * (xm + xp);
            UY[k] = mid * lyf + norm * (ym + yp);
            if (t < 4 * S) continue;
            // stage 4: row p
            const int pr = q - S;
            const float lxx = (mid * DX[km] + norm * (DX[kmm] + DX[k])) * fs2;
            const float lyy = (UY[k] - UY[kmm]) * fs2;
            const float lxy = (UX[k] - UX[kmm]) * fs2;
            if (out_lane) Ldet[fo + (size_t)pr * w + gx] = lxx * lyy - lxy * lxy;
        }
#pragma unroll
        for (int k = 0; k < P; ++k) cur[k] = nxt[k];
    }
}

// ---------------- launchers ----------------
static inline dim3 akz_grid1(int w, int h, int nframes, int tw, int th) {  // 1-D, padded to a multiple of 8 (XCD remap)
    const int total = ((w + tw - 1) / tw) * ((h + th - 1) / th) * nframes;
    return dim3((total + 7) / 8 * 8);
}
static inline dim3 akz_grid(int w, int h, int nframes) { return akz_grid1(w, h, nframes, AT_W, AT_H); }

extern "C" int afv_akz_launch_gauss(const void *src, int is_u8, int src_stride, size_t src_frame_stride, int w, int h, int nframes,
                                    const float *taps, int ksize, float *dst, hipStream_t st) {
    const dim3 g = akz_grid(w, h, nframes);
    const int r = ksize / 2;
#define AKZ_GAUSS_CASE(R)                                                                                                   \
    case R:                                                                                                                 \
        if (is_u8) hipLaunchKernelGGL((k_akz_gauss<R, true>), g, dim3(AKZ_T), 0, st, src, src_stride, src_frame_stride, w, h, nframes, taps, dst); \
        else hipLaunchKernelGGL((k_akz_gauss<R, false>), g, dim3(AKZ_T), 0, st, src, src_stride, src_frame_stride, w, h, nframes, taps, dst);      \
        return 0;
    switch (r) {
        AKZ_GAUSS_CASE(1) AKZ_GAUSS_CASE(2) AKZ_GAUSS_CASE(3) AKZ_GAUSS_CASE(4) AKZ_GAUSS_CASE(5) AKZ_GAUSS_CASE(6)
        default: return -1;
    }
#undef AKZ_GAUSS_CASE
}

// gsm != nullptr: the sigma = 1 image was written by k_akz_gauss; else it is formed inside the magnitude kernel from the gray frame
// (5-tap Gaussian `taps`)
extern "C" void afv_akz_launch_kcontrast(const float *gsm, const uint8_t *gray, int src_stride, size_t src_frame_stride, const float *taps, int w,
                                         int h, int nframes, float *modg, unsigned int *hmax_bits, int *hist, int nbins, float perc,
                                         float *kcontrast, hipStream_t st) {
    if (gsm) hipLaunchKernelGGL(k_akz_modg, akz_grid(w, h, nframes), dim3(AKZ_T), 0, st, gsm, w, h, nframes, modg, hmax_bits);
    else hipLaunchKernelGGL(k_akz_contrast_modg, akz_grid(w, h, nframes), dim3(AKZ_T), 0, st, gray, src_stride, src_frame_stride, w, h, nframes, taps,
                            modg, hmax_bits);
    hipLaunchKernelGGL(k_akz_hist, dim3(64, nframes), dim3(AKZ_T), (size_t)(nbins + 1) * sizeof(int), st, modg, w, h, hmax_bits, nbins, hist);
    hipLaunchKernelGGL(k_akz_kperc, dim3((nframes + 63) / 64), dim3(64), 0, st, hist, hmax_bits, nbins, perc, nframes, kcontrast);
}

extern "C" void afv_akz_launch_halfsample(const float *src, int w, int h, float *dst, int dw, int dh, int nframes, hipStream_t st) {
    hipLaunchKernelGGL(k_akz_halfsample, dim3((dw + 63) / 64, (dh + 3) / 4, nframes), dim3(AKZ_T), 0, st, src, w, h, dst, dw, dh);
}

extern "C" void afv_akz_launch_flow(const float *lsm, int w, int h, int nframes, const float *kcontrast, int octave, float *flow,
                                    hipStream_t st) {
    hipLaunchKernelGGL(k_akz_flow, akz_grid(w, h, nframes), dim3(AKZ_T), 0, st, lsm, w, h, nframes, kcontrast, octave, flow);
}

extern "C" void afv_akz_launch_nld_step(const float *Lt, const float *flow, int w, int h, int nframes, float tau, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_akz_nld_step, akz_grid(w, h, nframes), dim3(AKZ_T), 0, st, Lt, flow, w, h, nframes, tau, out);
}

// Lsmooth (the 5 taps of the level's Gaussian; written to lsm for the derivative kernels) + conductivity + FED cycle in one launch.
// Returns 0 when the level does not fit the fused kernel - more than AKZ_FED_MAX steps, or an image too small for the register patch of
// the rows / columns just outside it - and the caller then steps through k_akz_gauss + k_akz_flow + k_akz_nld_step.
extern "C" int afv_akz_launch_fed_gauss(const float *Lt_in, float *lsm, const float *taps, int w, int h, int nframes, const float *kcontrast,
                                        int octave, int nsteps, const float *tau, float *Lt_out, hipStream_t st) {
    if (nsteps < 1 || nsteps > AKZ_FED_MAX || w < 4 || h < 4) return 0;
    AkzTau t{};
    for (int i = 0; i < nsteps; ++i) t.t[i] = tau[i];
    const size_t lds = (size_t)AKZ_G_LDS_FLOATS * sizeof(float);
    const dim3 g = akz_grid1(w, h, nframes, 62 - 2 * nsteps, 64 - 2 * nsteps);
#define AKZ_FG_CASE(NN)                                                                                                                  \
    case NN:                                                                                                                             \
        hipLaunchKernelGGL(k_akz_fed_gauss<NN>, g, dim3(AKZ_FT), lds, st, Lt_in, lsm, taps, w, h, nframes, kcontrast, octave, t, Lt_out); \
        break;
    switch (nsteps) {
        AKZ_FG_CASE(1) AKZ_FG_CASE(2) AKZ_FG_CASE(3) AKZ_FG_CASE(4) AKZ_FG_CASE(5) AKZ_FG_CASE(6) AKZ_FG_CASE(7) AKZ_FG_CASE(8)
    }
#undef AKZ_FG_CASE
    return 1;
}

// dx, dy: where the level's (unscaled) first derivatives go; nullptr (fused form only): they are not stored
extern "C" int afv_akz_launch_hessian(const float *lsm, int w, int h, int nframes, int s, int two_kernels, float *dx, float *dy, float *Ldet,
                                      hipStream_t st) {
    if (s < 1 || s > AKZ_MAX_S) return -1;
    if (s >= 2 && s <= 4 && !two_kernels) {
        const int ow = 64 - 4 * s;
        // a wavefront walks its strip row group by row group (a dependent load per group): a batch wants long strips (64 rows: fewest run-in
        // rows), a single frame short ones - 300 strips of 64 rows leave the chip idle for 25 us per level, strips of 16 rows take 9
        const int band_rows = nframes >= 8 ? AKZ_DH_ROWS : 16;
        const int strips = ((w + ow - 1) / ow) * ((h + band_rows - 1) / band_rows) * nframes;
        const dim3 g((strips + 3) / 4);
        if (dx && dy) {
            if (s == 2) hipLaunchKernelGGL((k_akz_dhess<2, true>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
            else if (s == 3) hipLaunchKernelGGL((k_akz_dhess<3, true>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
            else hipLaunchKernelGGL((k_akz_dhess<4, true>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
        } else {
            if (s == 2) hipLaunchKernelGGL((k_akz_dhess<2, false>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
            else if (s == 3) hipLaunchKernelGGL((k_akz_dhess<3, false>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
            else hipLaunchKernelGGL((k_akz_dhess<4, false>), g, dim3(256), 0, st, lsm, w, h, nframes, band_rows, dx, dy, Ldet);
        }
        return 0;
    }
    if (!dx || !dy) return -1;  // the two-kernel form hands the first derivatives over in memory
    const size_t plane = (size_t)(AT_W + 2 * s) * (AT_H + 2 * s) * sizeof(float);
    hipLaunchKernelGGL(k_akz_deriv1, akz_grid(w, h, nframes), dim3(AKZ_T), plane, st, lsm, w, h, nframes, s, dx, dy);
    hipLaunchKernelGGL(k_akz_hessian, akz_grid(w, h, nframes), dim3(AKZ_T), 2 * plane, st, dx, dy, w, h, nframes, s, Ldet);
    return 0;
}
