// k_describe.hip — E6 + E10 + E11 fused: IC angle and rotated BRIEF, one wavefront per selected keypoint.
//
// Replaces ICAngles (cv::ORB::detect, Feature_orb32.cpp:34 == IC_Angle ORBextractor.cc:143-170), computeOrbDescriptors inside each
// cv::ORB::compute call (Feature_orb32.cpp:48; same arithmetic as computeOrbDescriptor FeatureExtractor.h:178-217) and
// mergeKeypointLevels (FeatureExtractor.cpp:296-308).
//
// Inputs are the two apron planes k_blur.hip writes per (frame, level): `blur` (ROI blurred 7x7, apron = unblurred reflect-101: the
// memory image OpenCV samples) and `raw` (unblurred + apron).  With a 20 px apron no keypoint has a border case: every wavefront
//   1. loads the 37 x 37 window of the blur plane around the BRIEF centre (7 aligned dword loads per lane, 6 rows per wave
//      instruction) and the radius-15 disc of the raw plane (lane = (row, half): 5 aligned dwords straight from global memory, no
//      LDS round trip), all loads issued together;
//   2. intensity-centroid moments by v_dot4_u32_u8 on masked dwords + DPP wave sums, cv::fastAtan2, explicit f64 sincos;
//   3. the 512 rotated tests: each lane gathers 8 bytes of the blurred window from LDS (4 tests), ballot-assembled descriptor.
#include "afv_device.h"

// the 256 test pairs (x0, y0, x1, y1) as floats (FeatureExtractor.h:219-477): one 16-byte load per test, no conversions
__device__ __constant__ __attribute__((aligned(16))) const float k_brief_pattern[1024] = {
#include "brief_pattern.inc"
};
// umax[v], v = 0..15: last column of row v of the radius-15 disc (orb.cpp / ORBextractor.cc:124-139)
__device__ __constant__ const signed char k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

#define WR 18        // window radius: BRIEF reach (13 * sqrt 2 rounded)
#define WS 37        // window side
#define WP 44        // LDS pitch of the window rows: 11 dwords (odd => row strides spread over the banks)
#define KP_PER_BLOCK 4

// cv::fastAtan2 (OpenCV mathfuncs_core atan_f32), degrees
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg, p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg, p7 = -0.04432655554792128f * rad2deg;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.2204460492503131e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.2204460492503131e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// cos/sin of angle_deg*(pi/180) (float product, as FeatureExtractor.h:181): explicit double algorithm (Cody-Waite by
// pi/2 + Taylor), rounded once to float — bit-identical to the oracle's restatement, independent of any libm.
__device__ __forceinline__ void sincos_deg(float angle_deg, float &c_out, float &s_out) {
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f);
    const double t = (double)(angle_deg * factor_pi);
    const double kd = floor(t * 0.63661977236758138 + 0.5);
    const int k = (int)kd;
    const double r = (t - kd * 1.5707963267341256e+00) - kd * 6.0771005065061922e-11;
    const double z = r * r;
    const double sp = 1.0 + z * (-1.6666666666666666e-01 + z * (8.3333333333333332e-03 + z * (-1.9841269841269841e-04 +
                      z * (2.7557319223985893e-06 + z * (-2.5052108385441720e-08 + z * (1.6059043836821613e-10 +
                      z * (-7.6471637318198164e-13)))))));
    const double s = r * sp;
    const double c = 1.0 + z * (-0.5 + z * (4.1666666666666664e-02 + z * (-1.3888888888888889e-03 + z * (2.4801587301587302e-05 +
                     z * (-2.7557319223985888e-07 + z * (2.0876756987868100e-09 + z * (-1.1470745597729725e-11 +
                     z * (4.7794773323873853e-14))))))));
    double co, si;
    switch (k & 3) {
    case 0: co = c; si = s; break;
    case 1: co = -s; si = c; break;
    case 2: co = -c; si = -s; break;
    default: co = s; si = -c; break;
    }
    c_out = (float)co;
    s_out = (float)si;
}

// LDS hand-off between lanes of ONE wavefront: DS operations of a wave execute in order, so only the compiler has
// to be kept from moving the reads above the writes.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 64 lanes (every lane receives it): DPP row reductions, then the last lane's total through a scalar register
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}

__global__ __launch_bounds__(64 * KP_PER_BLOCK) void k_describe(const Geo *__restrict__ geo_p, const uint8_t *__restrict__ blur,
                                                                const uint8_t *__restrict__ raw, const SelPoint *__restrict__ sel,
                                                                const int *__restrict__ sel_count, afv_keypoint *__restrict__ kps,
                                                                uint8_t *__restrict__ desc, int cap_per_frame, int *__restrict__ n_out,
                                                                int *__restrict__ status, int frame_base, int per_frame, int total_blocks) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[KP_PER_BLOCK][WS * WP + 4];

    const Geo &geo = *geo_p;
    // XCD-aware placement: all keypoints of a frame are described on one XCD (their windows share L2 lines)
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    // a frame's blocks: level after level, ceil(sel_cap / KP_PER_BLOCK) blocks each (per_frame in total)
    const int fl = work / per_frame;
    const int blk = work - fl * per_frame;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && blk >= geo.lv[i].desc_blk_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int kblk = blk - L.desc_blk_base;
    const int f = frame_base + fl;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = kblk * KP_PER_BLOCK + wv;

    // frame-level bookkeeping: counts of the eight levels (two scalar loads), this level's first output slot, total
    const int4 *scp = reinterpret_cast<const int4 *>(sel_count + f * AFV_MAX_LEVELS);
    const int4 ca_ = scp[0], cb_ = scp[1];
    const int cnt[AFV_MAX_LEVELS] = {ca_.x, ca_.y, ca_.z, ca_.w, cb_.x, cb_.y, cb_.z, cb_.w};
    int level_base = 0, total = 0, mine = 0;
#pragma unroll
    for (int i = 0; i < AFV_MAX_LEVELS; ++i) {
        const int c = i < geo.nlevels ? cnt[i] : 0;
        level_base += i < l ? c : 0;
        mine = i == l ? c : mine;
        total += c;
    }
    if (blk == 0 && threadIdx.x == 0) {
        n_out[f] = min(total, cap_per_frame);
        if (status && total > cap_per_frame) atomicMin(status, AFV_ECAPACITY);
    }
    if (idx >= mine) return;  // wave-uniform
    const int out_idx = level_base + idx;
    if (out_idx >= cap_per_frame) return;

    const SelPoint sp = sel[(size_t)f * geo.sel_per_frame + L.sel_base + idx];
    const int cx = sp.x, cy = sp.y;
    const int bpitch = L.bpitch;
    const size_t plane = L.b_off + (size_t)f * L.b_frame_stride;
    // BRIEF centre = cvRound(pt * (1/scale)) with pt = level coordinate * scale (orb.cpp computeOrbDescriptors); equals (cx, cy) in
    // practice, kept literal
    const float ptx = (float)cx * L.scale, pty = (float)cy * L.scale;
    const int bx = (int)rintf(ptx * L.inv_scale), by = (int)rintf(pty * L.inv_scale);
    uint8_t *P = s_win[wv];

    // ---- 1. loads: the blurred window (plane coordinates: + AFV_APRON) and this lane's part of the IC disc ----
    const int wx0 = bx + AFV_APRON - WR, wy0 = by + AFV_APRON - WR;
    const int a = wx0 & 3;  // column of window column 0 inside the LDS row
    // rows rr, rr + 6, ..., rr + 36 of 10 dwords: lanes 60..63 shadow lane 59 and row 36 is loaded by every row slot (same value to the
    // same LDS address): no predication anywhere
    uint32_t wv_[7];
    const int ln = min(lane, 59);
    const int rr = (int)(__umul24((unsigned)ln, 6554u) >> 16), rq = ln - rr * 10;  // ln / 10, ln % 10
    {
        const uint8_t *gp = blur + plane + (size_t)(wy0 + rr) * bpitch + (wx0 - a) + rq * 4;
#pragma unroll
        for (int k = 0; k < 7; ++k) wv_[k] = *reinterpret_cast<const uint32_t *>(gp + (size_t)(k < 6 ? k * 6 : 36 - rr) * bpitch);
    }
    // IC disc: lane = (row v = (lane >> 1) - 15, half): half 0 covers u in [-16, -1], half 1 covers u in [0, 15]; five aligned dwords
    const int v = (lane >> 1) - 15, half = lane & 1;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    int sh = 0;
    if (v <= 15) {
        const int A = (cx + AFV_APRON) + (half ? 0 : -16);
        const uint32_t *wp = reinterpret_cast<const uint32_t *>(raw + plane + (size_t)(cy + AFV_APRON + v) * bpitch + (A & ~3));
        sh = (A & 3) * 8;
        w0 = wp[0]; w1 = wp[1]; w2 = wp[2]; w3 = wp[3]; w4 = wp[4];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) *reinterpret_cast<uint32_t *>(&P[(k < 6 ? rr + 6 * k : 36) * WP + rq * 4]) = wv_[k];

    // ---- 2. intensity centroid over the radius-15 disc ----
    int m10 = 0, m01 = 0;
    if (v <= 15) {
        const int av = v < 0 ? -v : v;
        const int d = k_umax[av];
        // half 0: u in [-d, -1] = bytes j = 16-d .. 15 of the 16 bytes starting at u = -16; half 1: u in [0, d] = bytes j = 0 .. d of the 16
        // bytes starting at u = 0.  Funnel-shifted to the start byte, bytes outside the disc masked to zero, then sum(val) and
        // sum(j * val) by v_dot4_u32_u8.
        uint32_t b[4] = {__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh), __builtin_amdgcn_alignbit(w3, w2, sh),
                         __builtin_amdgcn_alignbit(w4, w3, sh)};
        uint32_t s1 = 0, sj = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t m;
            if (half) {
                const int nb = min(max(d + 1 - 4 * q, 0), 4);                  // valid leading bytes
                m = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
            } else {
                const int nz = min(max(16 - d - 4 * q, 0), 4);                 // invalid leading bytes
                m = nz >= 4 ? 0u : (0xffffffffu << (8 * nz));
            }
            const uint32_t x = b[q] & m;
            s1 = __builtin_amdgcn_udot4(x, 0x01010101u, s1, false);
            sj = __builtin_amdgcn_udot4(x, 0x03020100u + 0x04040404u * (uint32_t)q, sj, false);
        }
        m10 = half ? (int)sj : (int)sj - 16 * (int)s1;
        m01 = v * (int)s1;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- 3. rotated BRIEF on the blurred window: lane handles tests lane, lane+64, lane+128, lane+192 ----
    float ca, sb;
    sincos_deg(angle, ca, sb);
    wave_sync();  // window complete (each wave owns its LDS slice: no workgroup barrier anywhere in this kernel)
    const uint8_t *C = &P[WR * WP + WR + a];  // window centre
    uint32_t words[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = g * 64 + lane;
        const float4 pt = reinterpret_cast<const float4 *>(k_brief_pattern)[t];
        const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
        const int ix0 = (int)rintf(x0 * ca - y0 * sb), iy0 = (int)rintf(x0 * sb + y0 * ca);
        const int ix1 = (int)rintf(x1 * ca - y1 * sb), iy1 = (int)rintf(x1 * sb + y1 * ca);
        const int t0 = C[iy0 * WP + ix0], t1 = C[iy1 * WP + ix1];
        const unsigned long long m = __ballot(t0 < t1);
        words[2 * g] = (uint32_t)m;
        words[2 * g + 1] = (uint32_t)(m >> 32);
    }
    // ---- 4. outputs (E11 merge: ascending level, list order inside a level) ----
    const size_t o = (size_t)f * cap_per_frame + out_idx;
    if (lane < 8) {
        uint32_t w = words[0];
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (lane == i) w = words[i];
        reinterpret_cast<uint32_t *>(desc + o * 32)[lane] = w;
    }
    if (lane == 0) {
        afv_keypoint k;
        k.x = ptx;
        k.y = pty;
        k.size = 31 * L.scale;
        k.angle = angle;
        k.response = sp.response;
        k.octave = l;
        k.class_id = -1;
        kps[o] = k;
    }
}

extern "C" int afv_describe_blocks_per_frame(const Geo *g) {
    int n = 0;
    for (int l = 0; l < g->nlevels; ++l) n += (g->lv[l].sel_cap + KP_PER_BLOCK - 1) / KP_PER_BLOCK;
    return n;
}

extern "C" void afv_launch_describe(const Geo *geo_dev, int blocks_per_frame, const uint8_t *blur, const uint8_t *raw, const SelPoint *sel,
                                    const int *sel_count, afv_keypoint *kps, uint8_t *desc, int cap_per_frame, int *n_out, int *status,
                                    int frame_base, int nframes, hipStream_t stream) {
    const int total = blocks_per_frame * nframes;
    dim3 grid((total + 7) / 8 * 8);
    hipLaunchKernelGGL(k_describe, grid, dim3(64 * KP_PER_BLOCK), 0, stream, geo_dev, blur, raw, sel, sel_count, kps, desc, cap_per_frame,
                       n_out, status, frame_base, blocks_per_frame, total);
}
