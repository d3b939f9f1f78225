// k_akaze_desc.hip — AKAZE61 plugin tail: per-level quadtree filter + Compute_Descriptors.
//
//  k_akz_select   = FeatureExtractor_akaze61::detectKeypoints' bucketing by class_id (Feature_akaze61.cpp:43-46) +
//                   filterKeypoints -> FeatureExtractor::filterKeypoints_notScaled -> DistributeOctTree
//                   (Feature_akaze61.cpp:63-65, FeatureExtractor.cpp:276-284, ORBextractor.cc:239-458): one workgroup per
//                   (frame, level), quadtree core shared with the ORB path (afv_quadtree.h).
//  k_akz_describe = libAKAZE Compute_Main_Orientation + Get_MLDB_Full_Descriptor (486 bits, 3 channels), one wavefront per
//                   surviving keypoint, in the order mergeKeypointLevels produces (levels ascending, list order inside).
// Float sums are order sensitive, so every accumulation below runs in upstream's loop order inside one lane: the 42
// orientation windows and the 29 MLDB grid cells are spread over lanes, their inner sums stay sequential.
#include <type_traits>
#include "afv_device.h"
#include "akz_jobs.h"
#include "../../include/afv_hip.h"
#include "afv_quadtree.h"
#include "akaze_tables.inc"



struct AksPts {
    const afv_keypoint *k;
    const int *idx;
    __device__ __forceinline__ float x(int p) const { return k[idx[p]].x; }
    __device__ __forceinline__ float y(int p) const { return k[idx[p]].y; }
};

__global__ __launch_bounds__(QT_T) void k_akz_select(AksParams P, const afv_keypoint *__restrict__ kps, const int *__restrict__ kp_count,
                                                     int *__restrict__ lvl_idx, uint16_t *__restrict__ lvl_node, int *__restrict__ sel,
                                                     int *__restrict__ sel_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int level = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const QtScratch S = qt_carve(smem, P.M);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(smem + qt_lds_bytes(P.M));
    constexpr int GU = 4, GW = QT_T / 64;  // keypoints per thread and gather step; wavefronts
    __shared__ int s_wsum[2][GU * GW];
    const afv_keypoint *k = kps + (size_t)f * P.kp_cap;
    int *idx = lvl_idx + ((size_t)f * P.nlevels + level) * P.kp_cap;
    uint16_t *kn = lvl_node + ((size_t)f * P.nlevels + level) * P.kp_cap;
    const int n = kp_count[f];
    // ordered gather of this level's keypoints (keypoints_level[class_id].push_back in detection order).  Every workgroup of a frame
    // walks all of the frame's keypoints (a replaced entry keeps its slot, so a level's points are not contiguous): GU x QT_T of them
    // per step, ONE barrier per step (the wave counts alternate between two LDS rows, the running total lives in registers)
#ifdef AFV_AKS_STATS
    const long long st0 = wall_clock64();
#endif
    int m2 = 0;
    for (int i0 = 0, it = 0; i0 < n; i0 += GU * QT_T, ++it) {
        unsigned long long m[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = i0 + u * QT_T + tid;
            m[u] = __ballot(i < n && k[i].class_id == level);
            if (lane == 0) s_wsum[it & 1][u * GW + (tid >> 6)] = __popcll(m[u]);
        }
        __syncthreads();
        int base = m2;
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            int off = base;
#pragma unroll
            for (int w = 0; w < GW; ++w) {
                const int cw = s_wsum[it & 1][u * GW + w];
                off += w < (tid >> 6) ? cw : 0;
                base += cw;
            }
            if ((m[u] >> lane) & 1ull) idx[off + __popcll(m[u] & ((1ull << lane) - 1ull))] = i0 + u * QT_T + tid;
        }
        m2 = base;
    }
    __threadfence_block();
    __syncthreads();
#ifdef AFV_AKS_STATS
    const long long st1 = wall_clock64();
#endif
    AksPts pts{k, idx};
    const int size = qt_build(pts, m2, P.quota[level], P.n_ini, P.h_x, P.H, kn, S);
#ifdef AFV_AKS_STATS
    const long long st2 = wall_clock64();
#endif
    // survivor of each node: max response, first in input order on ties (ORBextractor.cc:446-453)
    for (int i = tid; i < size; i += QT_T) best[i] = 0ull;
    __syncthreads();
    for (int p = tid; p < m2; p += QT_T) {
        const unsigned long long key = ((unsigned long long)qt_float_key(k[idx[p]].response) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)p);
        atomicMax(&best[kn[p]], key);
    }
    __syncthreads();
    const int nout = min(size, P.sel_cap);
    int *out = sel + ((size_t)f * P.nlevels + level) * P.sel_cap;
    for (int i = tid; i < nout; i += QT_T) out[i] = idx[0xffffffffu - (uint32_t)(best[i] & 0xffffffffu)];
    if (tid == 0) sel_count[f * AKS_MAX_LEVELS + level] = nout;
#ifdef AFV_AKS_STATS
    if (tid == 0 && f == 0) printf("akz_select level %d: n %d m2 %d size %d | x10 ns: gather %lld build %lld pick %lld\n", level, n, m2, size, st1 - st0, st2 - st1, wall_clock64() - st2);
#endif
}

// ---------------- Compute_Descriptors ----------------

#define AKZ_PI_D 3.14159265358979323846

__device__ __forceinline__ int akd_fround(float x) { return (int)(x + 0.5f); }
__device__ __forceinline__ int akd_iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// atan(z), z >= 0, same explicit algorithm as oracle/akaze.c atan_pos_f64
__device__ __forceinline__ double akd_atan_pos(double z) {
    const double pio2 = 1.5707963267948966, pio4 = 0.78539816339744831;
    double base, t;
    if (z > 2.414213562373095) { base = pio2; t = -1.0 / z; }
    else if (z > 0.4142135623730950) { base = pio4; t = (z - 1.0) / (z + 1.0); }
    else { base = 0.0; t = z; }
    const double w = t * t;
    double s = 1.0 / 23.0;
    s = 1.0 / 21.0 - w * s;
    s = 1.0 / 19.0 - w * s;
    s = 1.0 / 17.0 - w * s;
    s = 1.0 / 15.0 - w * s;
    s = 1.0 / 13.0 - w * s;
    s = 1.0 / 11.0 - w * s;
    s = 1.0 / 9.0 - w * s;
    s = 1.0 / 7.0 - w * s;
    s = 1.0 / 5.0 - w * s;
    s = 1.0 / 3.0 - w * s;
    s = 1.0 - w * s;
    return base + t * s;
}
__device__ __forceinline__ float akd_atanf(float z) { return (float)akd_atan_pos((double)z); }
__device__ __forceinline__ float akd_get_angle(float x, float y) {
    if (x >= 0 && y >= 0) return akd_atanf(y / x);
    if (x < 0 && y >= 0) return (float)(AKZ_PI_D - (double)akd_atanf(-y / x));
    if (x < 0 && y < 0) return (float)(AKZ_PI_D + (double)akd_atanf(y / x));
    if (x >= 0 && y < 0) return (float)(2.0 * AKZ_PI_D - (double)akd_atanf(-y / x));
    return 0.0f;
}
__device__ __forceinline__ void akd_sincos(double t, double *c_out, double *s_out) {  // as k_describe.hip / oracle
    const double two_over_pi = 0.63661977236758138;
    const double pio2_hi = 1.5707963267341256e+00, pio2_lo = 6.0771005065061922e-11;
    const double kd = floor(t * two_over_pi + 0.5);
    const int k = (int)kd;
    const double r = (t - kd * pio2_hi) - kd * pio2_lo;
    const double z = r * r;
    const double sp = 1.0 + z * (-1.6666666666666666e-01 + z * (8.3333333333333332e-03 + z * (-1.9841269841269841e-04 +
                      z * (2.7557319223985893e-06 + z * (-2.5052108385441720e-08 + z * (1.6059043836821613e-10 +
                      z * (-7.6471637318198164e-13)))))));
    const double s = r * sp;
    const double c = 1.0 + z * (-0.5 + z * (4.1666666666666664e-02 + z * (-1.3888888888888889e-03 + z * (2.4801587301587302e-05 +
                     z * (-2.7557319223985888e-07 + z * (2.0876756987868100e-09 + z * (-1.1470745597729725e-11 +
                     z * (4.7794773323873853e-14))))))));
    switch (k & 3) {
        case 0: *c_out = c; *s_out = s; break;
        case 1: *c_out = -s; *s_out = c; break;
        case 2: *c_out = -c; *s_out = -s; break;
        default: *c_out = s; *s_out = -c; break;
    }
}

__constant__ float k_gauss25[7][7] = {
    {0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
    {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
    {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
    {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
    {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
    {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
    {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};

// ---- first derivatives on demand ----
// Upstream computes Lx / Ly for every pixel of every level (Compute_Multiscale_Derivatives) and the descriptor stage reads them at ~550
// positions per keypoint: on a 1280 x 720 frame with 850 keypoints less than a tenth of the 37 MB of derivative planes is ever read.
// Round 5 stopped storing them (k_akz_dhess keeps them in registers for the Hessian only): a sample position (x, y) gets its
// derivatives from the eight Lsmooth taps around it with exactly the expressions of k_akz_deriv1 (k_akaze.hip; the Scharr pair at
// tap distance s, inputs at BORDER_REFLECT_101 coordinates), i.e. the same float bits the planes held.
__device__ __forceinline__ int akd_reflect(int p, int n) {  // BORDER_REFLECT_101 for -n < p < 2n - 1 (tap distance <= 8, levels >= 20 pixels): one fold
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}
struct AkdTaps {
    float a, b, c, d, e, f, g, h;  // rows y - s, y, y + s x columns x - s, x, x + s without the centre
};
__device__ __forceinline__ AkdTaps akd_load_taps(const float *__restrict__ S, int w, int h, int s, int x, int y) {
    const int xm = akd_reflect(x - s, w), xp = akd_reflect(x + s, w);
    // 32-bit BYTE offsets into the frame's plane (S is wave-uniform: scalar base + vector offset is what the load takes; rows and widths
    // are far below 2^24, a plane far below 4 GB): 24-bit multiplies instead of three 64-bit multiply-adds (quarter rate) per sample
    const uint32_t rm = __umul24((uint32_t)akd_reflect(y - s, h), (uint32_t)w), r0 = __umul24((uint32_t)y, (uint32_t)w),
                   rp = __umul24((uint32_t)akd_reflect(y + s, h), (uint32_t)w);
    const char *B = reinterpret_cast<const char *>(S);
    auto at = [&](uint32_t row, int col) { return *reinterpret_cast<const float *>(B + ((row + (uint32_t)col) << 2)); };
    AkdTaps t;
    t.a = at(rm, xm); t.b = at(rm, x); t.c = at(rm, xp);
    t.d = at(r0, xm);                  t.e = at(r0, xp);
    t.f = at(rp, xm); t.g = at(rp, x); t.h = at(rp, xp);
    return t;
}
__device__ __forceinline__ void akd_deriv(const AkdTaps &t, float mid, float norm, float *vx, float *vy) {
    const float t0 = t.c - t.a, t1 = t.e - t.d, t2 = t.h - t.f;   // Lx: row derivative, column smoothing
    *vx = mid * t1 + norm * (t0 + t2);
    const float u0 = mid * t.b + norm * (t.a + t.c);              // Ly: row smoothing, column derivative
    const float u2 = mid * t.g + norm * (t.f + t.h);
    *vy = u2 - u0;
}

// window start angles of Compute_Main_Orientation: upstream's loop variable ang1 += 0.15f, i.e. lane t holds the float sum of t additions
struct AkdAngles {
    float a[64];
};
constexpr AkdAngles akd_make_angles() {
    AkdAngles r{};
    float v = 0.0f;
    for (int i = 0; i < 64; ++i) {
        r.a[i] = v;
        v += 0.15f;
    }
    return r;
}
__constant__ AkdAngles k_ang1 = akd_make_angles();

#define AKD_LDS_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

__global__ __launch_bounds__(256) void k_akz_describe(AkdDescParams P, const afv_keypoint *__restrict__ kps, const int *__restrict__ sel,
                                                      const int *__restrict__ sel_count, afv_keypoint *__restrict__ out_kps,
                                                      uint8_t *__restrict__ out_desc, int *__restrict__ out_count, int *__restrict__ status) {
    __shared__ float s_val[4][96];
    // MLDB samples of the 21 x 21 pattern positions as three planes {Lt, rotated Lx, rotated Ly} of 441 floats (a float4 per sample
    // with an unused lane cost 7 KB per workgroup and, at 29.8 KB, two workgroups of occupancy per CU); before that, the first 436
    // floats hold the 109 orientation samples {angle, weighted Lx, weighted Ly, angle < 2 pi (as 1 / 0)} as float4
    __shared__ __attribute__((aligned(16))) float s_smp[4][3 * 441 + 1];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, f = blockIdx.y;  // the keypoint is the wavefront's
    const int slot = blockIdx.x * 4 + wv;
    // slot -> (level, position): levels ascending (mergeKeypointLevels, FeatureExtractor.cpp:296-308)
    int level = -1, pos = slot, total = 0;
    for (int l = 0; l < P.nlevels; ++l) {
        const int c = sel_count[f * AKS_MAX_LEVELS + l];
        if (level < 0 && pos < c) level = l;
        if (level < 0) pos -= c;
        total += c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_count[f] = min(total, P.out_cap);
        if (total > P.out_cap) atomicExch(status, 4);
    }
    if (level < 0 || slot >= P.out_cap) return;
    afv_keypoint kp = kps[(size_t)f * P.kp_cap + sel[((size_t)f * P.nlevels + level) * P.sel_cap + pos]];
    const AkdLevelPlanes L = P.lv[level];
    const size_t fo = (size_t)f * L.w * L.h;
    const float *Lt = L.lt + fo, *Ls = L.lsm + fo;
    const float d_wgt = 10.0f / 3.0f, d_norm = 1.0f / (2.0f * (float)L.s * (d_wgt + 2.0f)), d_mid = d_wgt * d_norm;  // as k_akz_deriv1
    const float ratio = (float)(1 << L.octave);
    const float xf = kp.x / ratio, yf = kp.y / ratio;
    float4 *ori = reinterpret_cast<float4 *>(s_smp[wv]);
    float *val = s_val[wv];
    // ---- Compute_Main_Orientation ----
    {
        const int s = akd_fround((float)(0.5 * (double)kp.size / (double)ratio));
        AkdTaps ot[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // the gathers of both samples of a lane in flight together
            const int idx = min(lane + 64 * it, 108);
            const int i = k_ori_ij[idx][0], j = k_ori_ij[idx][1];
            const int iy = akd_iclamp(akd_fround(yf + (float)(j * s)), 0, L.h - 1), ix = akd_iclamp(akd_fround(xf + (float)(i * s)), 0, L.w - 1);
            ot[it] = akd_load_taps(Ls, L.w, L.h, L.s, ix, iy);
        }
        float ox[2], oy[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) akd_deriv(ot[it], d_mid, d_norm, &ox[it], &oy[it]);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = lane + 64 * it;
            if (idx < 109) {
                const int i = k_ori_ij[idx][0], j = k_ori_ij[idx][1];
                const float gw = k_gauss25[i < 0 ? -i : i][j < 0 ? -j : j];
                const float vx = gw * (ox[it] * L.fs), vy = gw * (oy[it] * L.fs);
                const float ang = akd_get_angle(vx, vy);
                ori[idx] = make_float4(ang, vx, vy, (double)ang < 2.0 * AKZ_PI_D ? 1.0f : 0.0f);
            }
        }
        AKD_LDS_SYNC();
        // 42 sliding windows (ang1 = 0, 0.15, ... accumulated in float like upstream's loop variable), one per lane
        const float ang1 = k_ang1.a[lane];
        const bool live = (double)ang1 < 2.0 * AKZ_PI_D;
        float sumX = 0.f, sumY = 0.f;
        {
            const float ang2 =
                (float)((double)ang1 + AKZ_PI_D / 3.0 > 2.0 * AKZ_PI_D ? (double)ang1 - 5.0 * AKZ_PI_D / 3.0 : (double)ang1 + AKZ_PI_D / 3.0);
            // upstream: if (ang1 < ang2 && ang1 < ang && ang < ang2) add; else if (ang2 < ang1 && ((ang > 0 && ang < ang2) ||
            // (ang > ang1 && ang < 2 pi))) add - the window either wraps or it does not, which is a property of the lane.
            // Eight samples' broadcast reads in flight, the test as mask arithmetic (as short-circuit code every sample was an LDS round
            // trip followed by a ladder of exec-mask branches: most of this kernel's time); every window still adds ITS samples in order.
            const bool nowrap = ang1 < ang2, wrap = ang2 < ang1;
            for (int k0 = 0; k0 < 109; k0 += 8) {
                float4 o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = ori[min(k0 + u, 108)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool a1 = ang1 < o[u].x, a2 = o[u].x < ang2;
                    const bool in = (k0 + u < 109) & ((nowrap & a1 & a2) | (!nowrap & wrap & (((o[u].x > 0) & a2) | (a1 & (o[u].w != 0.0f)))));
                    sumX = in ? sumX + o[u].y : sumX;
                    sumY = in ? sumY + o[u].z : sumY;
                }
            }
        }
        const float mag = live ? sumX * sumX + sumY * sumY : 0.0f;
        // first window with the largest magnitude wins (upstream updates on strict >); magnitude 0 never wins
        unsigned long long key = ((unsigned long long)__float_as_uint(mag) << 32) | (unsigned)(63 - lane);
        if (!(mag > 0.0f)) key = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = __shfl_xor(key, o, 64);
            key = t > key ? t : key;
        }
        if (key != 0) {
            const int win = 63 - (int)(key & 0xffffffffu);
            const float wx = __shfl(sumX, win, 64), wy = __shfl(sumY, win, 64);
            kp.angle = akd_get_angle(wx, wy);
        }
    }
    // ---- Get_MLDB_Full_Descriptor ----
    {
        const float scale = (float)akd_fround(0.5f * kp.size / ratio);
        double cd, sd;
        akd_sincos((double)kp.angle, &cd, &sd);
        const float co = (float)cd, si = (float)sd;
        // The 2 x 2, 3 x 3 and 4 x 4 grids all sum samples at the integer pattern positions (k, l) in [-10, 10]^2, and a sample
        // depends on (k, l) only: the 441 positions are fetched once, all lanes in parallel (1241 dependent gathers per keypoint
        // if every cell fetches its own), and every cell then adds ITS samples in upstream's (k, l) loop order from LDS.
        AKD_LDS_SYNC();  // every lane is through with the orientation samples that share this wavefront's block
        float *smp = s_smp[wv];
        auto mldb_samples = [&](auto first, auto count) {  // the 9 gathers of each of `count` samples of a lane in flight together
            constexpr int IT0 = decltype(first)::value, N = decltype(count)::value;
            float ri[N];
            AkdTaps tp[N];
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int p = min(lane + 64 * (IT0 + n), 440);
                const int k = p / 21 - 10, l = p - (p / 21) * 21 - 10;
                const float sample_y = yf + ((float)l * co * scale + (float)k * si * scale);
                const float sample_x = xf + (-(float)l * si * scale + (float)k * co * scale);
                const int y1 = akd_iclamp(akd_fround(sample_y), 0, L.h - 1), x1 = akd_iclamp(akd_fround(sample_x), 0, L.w - 1);
                ri[n] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(Lt) + ((__umul24((uint32_t)y1, (uint32_t)L.w) + (uint32_t)x1) << 2));
                tp[n] = akd_load_taps(Ls, L.w, L.h, L.s, x1, y1);
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int p = lane + 64 * (IT0 + n);
                if (p < 441) {
                    float gx, gy;
                    akd_deriv(tp[n], d_mid, d_norm, &gx, &gy);
                    const float vx = gx * L.fs, vy = gy * L.fs;
                    smp[p] = ri[n];
                    smp[441 + p] = -vx * si + vy * co;  // rrx
                    smp[882 + p] = vx * co + vy * si;   // rry
                }
            }
        };
        mldb_samples(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
        mldb_samples(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
        AKD_LDS_SYNC();
        if (lane < 29) {
            const int i0 = k_mldb_cell[lane][1], j0 = k_mldb_cell[lane][2], step = k_mldb_cell[lane][3];
            float di = 0.f, dx = 0.f, dy = 0.f;
            // the cell's samples in upstream's (k, l) order as one flat walk, four reads in flight (a 10 x 10 cell is a chain of 100 additions
            // per channel: as nested loops every addition waited for its own LDS round trip)
            const float *q = smp + (i0 + 10) * 21 + (j0 + 10);
            const int nn = step * step;
            int kk = 0, ll = 0;
            for (int t0 = 0; t0 < nn; t0 += 4) {
                float3 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = min(kk, step - 1) * 21 + ll;
                    v[u] = make_float3(q[o], q[441 + o], q[882 + o]);
                    if (++ll == step) {
                        ll = 0;
                        ++kk;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t0 + u < nn) {
                        di += v[u].x;
                        dx += v[u].y;
                        dy += v[u].z;
                    }
            }
            const float ns = (float)(step * step);
            val[3 * lane] = di / ns;
            val[3 * lane + 1] = dx / ns;
            val[3 * lane + 2] = dy / ns;
        }
        AKD_LDS_SYNC();
        unsigned long long m[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int b = t * 64 + lane;
            bool bit = false;
            if (b < 486) {
                int va = __float_as_int(val[k_mldb_bits[b][0]]), vb = __float_as_int(val[k_mldb_bits[b][1]]);
                va ^= (va < 0 ? 0x7fffffff : 0);  // CV_TOGGLE_FLT
                vb ^= (vb < 0 ? 0x7fffffff : 0);
                bit = va > vb;
            }
            m[t] = __ballot(bit);
        }
        if (lane < 61) {
            unsigned long long word = m[0];
#pragma unroll
            for (int t = 1; t < 8; ++t)
                if ((lane >> 3) == t) word = m[t];
            out_desc[((size_t)f * P.out_cap + slot) * P.desc_pitch + lane] = (uint8_t)(word >> (8 * (lane & 7)));
        }
    }
    if (lane == 0) out_kps[(size_t)f * P.out_cap + slot] = kp;
}

extern "C" size_t afv_akz_select_lds_bytes(int M) { return qt_lds_bytes(M) + (size_t)M * 8; }

extern "C" void afv_akz_launch_select(const AksParams *P, int nframes, const afv_keypoint *kps, const int *kp_count, int *lvl_idx,
                                      uint16_t *lvl_node, int *sel, int *sel_count, hipStream_t st) {
    hipLaunchKernelGGL(k_akz_select, dim3(P->nlevels, nframes), dim3(QT_T), afv_akz_select_lds_bytes(P->M), st, *P, kps, kp_count, lvl_idx,
                       lvl_node, sel, sel_count);
}

extern "C" void afv_akz_launch_describe(const AkdDescParams *P, int nframes, int max_out, const afv_keypoint *kps, const int *sel,
                                        const int *sel_count, afv_keypoint *out_kps, uint8_t *out_desc, int *out_count, int *status,
                                        hipStream_t st) {
    hipLaunchKernelGGL(k_akz_describe, dim3((max_out + 3) / 4, nframes), dim3(256), 0, st, *P, kps, sel, sel_count, out_kps, out_desc, out_count,
                       status);
}
