// k_describe.hip — E6 + E9 + E10 + E11 fused: IC angle, 7x7 Gaussian blur and rotated BRIEF, one wavefront
// per selected keypoint; plus the standalone level blur used by afv_debug_blur_level.
//
// Replaces ICAngles (cv::ORB::detect, Feature_orb32.cpp:34 == IC_Angle ORBextractor.cc:143-170), the
// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level and computeOrbDescriptors inside each
// cv::ORB::compute call (Feature_orb32.cpp:48; same arithmetic as computeOrbDescriptor FeatureExtractor.h:178-217),
// and mergeKeypointLevels (FeatureExtractor.cpp:296-308).
//
// The reference blurs whole levels (36 level-blurs per frame because compute() is called once per level).  Only the
// 512 rotated test locations inside the 37x37 neighbourhood of a keypoint are ever sampled, so each wavefront stages
// the 43x43 UNBLURRED patch around its keypoint in LDS (reflect-101 at the image edge = cv::ORB's apron), computes
// the intensity-centroid moments from it, runs the ROW pass of the blur over the patch once (integer taps
// [18,34,49,55,49,34,18], two v_dot4_u32_u8 per output on funnel-shifted dwords, exact in 16 bits) and evaluates the COLUMN
// pass + round-half-even(S/65536) only at the sampled locations.  A test that falls outside the level ROI reads the unblurred apron pixel, as in OpenCV
// where only the ROI is blurred in place.  No blurred image ever touches HBM.
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

// the 256 test pairs (x0, y0, x1, y1) as floats (FeatureExtractor.h:219-477): one 16-byte load per test, no conversions
__device__ __constant__ __attribute__((aligned(16))) const float k_brief_pattern[1024] = {
#include "brief_pattern.inc"
};

// IC stage: lane = (row v = (lane >> 1) - 15, half): the four byte masks of the lane's 16-byte run (which bytes lie inside the disc) depend
// on the lane only - a table instead of ~24 vector instructions of min / max / shift per keypoint
struct IcMasks {
    uint32_t m[64][4];
};
constexpr IcMasks ic_make_masks() {
    IcMasks t{};
    // umax[v], v = 0..15: last column of row v of the radius-15 disc (orb.cpp / ORBextractor.cc:124-139)
    constexpr int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int lane = 0; lane < 64; ++lane) {
        const int v = (lane >> 1) - 15, half = lane & 1;
        for (int q = 0; q < 4; ++q) {
            uint32_t m = 0;
            if (v <= 15) {
                const int d = umax[v < 0 ? -v : v];
                if (half) {
                    int nb = d + 1 - 4 * q;  // valid leading bytes
                    nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);
                    m = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
                } else {
                    int nz = 16 - d - 4 * q;  // invalid leading bytes
                    nz = nz < 0 ? 0 : (nz > 4 ? 4 : nz);
                    m = nz >= 4 ? 0u : (0xffffffffu << (8 * nz));
                }
            }
            t.m[lane][q] = m;
        }
    }
    return t;
}
__device__ __constant__ __attribute__((aligned(16))) const IcMasks k_ic_masks = ic_make_masks();

#define PR 21        // patch radius: 18 (BRIEF reach) + 3 (blur)
#define PS 43        // patch side
#define PP 52        // LDS pitch of the patch rows: 13 dwords (odd => conflict-free row strides)
#define KP_PER_BLOCK AFV_KP_PER_BLOCK  // afv_device.h (the host plan of the block list divides by it too)

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

// round-half-even(S / 65536) saturated to 255.  S < 2^24 whenever the result is not saturated, so u32 -> f32 is exact, the scaling by
// 2^-16 is exact, and v_cvt_pk_u8_f32 rounds to nearest even and clamps: identical to the integer rule for every S the filter can
// produce (all 16 842 496 values checked on the device: tools/probes/probe_cvt_pk_u8.hip)
__device__ __forceinline__ uint32_t blur_round(uint32_t S) { return __builtin_amdgcn_cvt_pk_u8_f32((float)S * (1.0f / 65536.0f), 0, 0u); }

// Row pass of the separable 8U filter over the whole staged patch, once per keypoint: H[r][xh] = sum_k taps[k] * P[r][xh + k]
// for xh = 0..39 (<= 257 * 255 = 65535: exact in 16 bits).  One lane per group of 4 adjacent outputs: 3 aligned dwords in, two
// v_dot4_u32_u8 per output on funnel-shifted dwords, 4 x u16 out.
#ifndef HP
#define HP 40  // u16 pitch of the row-filtered plane (80 bytes: 8-byte aligned rows, no padding: 44 cost a workgroup of occupancy per CU)
#endif
__device__ __forceinline__ void blur_rows(const uint8_t *P, uint16_t *H, int lane) {
    // output k of a group = taps over bytes k .. k + 6 of the 12-byte window (d0, d1, d2): instead of shifting the DATA to the taps
    // (two funnel shifts per output) the TAPS are laid out at the byte positions of every k - eleven constants in scalar registers,
    // 2 + 3 + 3 + 3 dot products and no shifts for the four outputs
    constexpr unsigned long long TAPS = 18ull | (34ull << 8) | (49ull << 16) | (55ull << 24) | (49ull << 32) | (34ull << 40) | (18ull << 48);
    for (int i = lane; i < PS * 10; i += 64) {
        const int r = i / 10, g = i - r * 10;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(P + r * PP + g * 4);
        const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the 7 taps shifted up by k bytes inside a 96-bit field: its three dwords
            const unsigned __int128 F = (unsigned __int128)TAPS << (8 * k);
            const uint32_t t0 = (uint32_t)F, t1 = (uint32_t)(F >> 32), t2 = (uint32_t)(F >> 64);
            uint32_t a = __builtin_amdgcn_udot4(d1, t1, __builtin_amdgcn_udot4(d0, t0, 0u, false), false);
            if (t2) a = __builtin_amdgcn_udot4(d2, t2, a, false);
            o[k] = a;
        }
        uint2 w;
        w.x = o[0] | (o[1] << 16);
        w.y = o[2] | (o[3] << 16);
        *reinterpret_cast<uint2 *>(&H[r * HP + g * 4]) = w;
    }
}

// column pass + rounding at patch position (row y, column x), taps centred: exact integer arithmetic of the separable filter,
// then round-half-even(S / 65536)
typedef unsigned short ushort2d __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int blur_at(const uint16_t *H, int x, int y) {
    const uint16_t *c = H + (__umul24((uint32_t)(y - 3), (uint32_t)HP) + (uint32_t)(x - 3));  // 24-bit multiply (y - 3 is 0 .. 36): a plain int product is a quarter-rate v_mul_lo_u32
    // symmetric taps: rows k and 6 - k share a weight -> three v_dot2_u32_u16 on (row k, row 6 - k) pairs + the centre row
    ushort2d p0, p1, p2, t0, t1, t2;
    p0.x = c[0];
    p0.y = c[6 * HP];
    p1.x = c[HP];
    p1.y = c[5 * HP];
    p2.x = c[2 * HP];
    p2.y = c[4 * HP];
    t0.x = t0.y = 18;
    t1.x = t1.y = 34;
    t2.x = t2.y = 49;
    uint32_t S = 55u * (uint32_t)c[3 * HP];
    S = __builtin_amdgcn_udot2(p0, t0, S, false);
    S = __builtin_amdgcn_udot2(p1, t1, S, false);
    S = __builtin_amdgcn_udot2(p2, t2, S, false);
    return (int)blur_round(S);
}

// MODE 0: the extraction pipeline (keypoints of the quadtree survivors + their descriptors: detectAndCompute);
// MODE 1: keypoints only - position, IC angle, response (afv_orb_detect: detectKeypoints + filterKeypoints, Feature_orb32.cpp:26-40, :63-65);
// MODE 2: descriptors of CALLER-GIVEN keypoints at their own angle (afv_orb_compute: computeDescriptors = cv::ORB::compute, :42-53):
//         `given` holds total_blocks * KP_PER_BLOCK >= n keypoints of frame `frame_base`, `cap_per_frame` = n, descriptor i to desc[i]
template <int MODE>
__device__ __forceinline__ void describe_body(const Geo *__restrict__ geo_p, const FrameSrc src0, const uint8_t *__restrict__ pyr,
                                              const SelPoint *__restrict__ sel, const int *__restrict__ sel_count, afv_keypoint *__restrict__ kps,
                                              uint8_t *__restrict__ desc, int cap_per_frame, int *__restrict__ n_out, int *__restrict__ status,
                                              int frame_base, int per_frame, int total_blocks, const DescribeMirror mir,
                                              const afv_keypoint *__restrict__ given) {
    // One LDS slice per wavefront holds BOTH the staged patch (u8, rows PP = 52 bytes apart: 13 dwords, odd -> rows spread over all
    // LDS banks) and the row-filtered plane H (u16, rows 2 HP = 80 bytes apart) that is computed from it: H starts at byte 0, the patch
    // at byte SLICE - PS * PP, and H row r ends at or before patch row r + 1 begins (80 r + 80 <= P_OFF + 52 (r + 1) for r <= 42), so the
    // ascending row pass only ever overwrites patch rows it has consumed (rows sharing a wave iteration are read before anything of
    // that iteration is written: DS operations of a wave execute in order).  3440 instead of 6028 bytes per keypoint: 8 instead of 6
    // workgroups per CU.  The patch is dead after the row pass: the IC moments are taken first, and the rare test of a border keypoint
    // that falls outside the level reads its (unblurred, reflected) pixel from the level image itself.
    constexpr int SLICE = PS * HP * 2, P_OFF = SLICE - PS * PP;
    static_assert(P_OFF % 4 == 0 && P_OFF >= 0 && 2 * HP * PS <= SLICE, "slice layout");
    static_assert(2 * HP * (PS - 1) + 2 * HP <= P_OFF + PP * PS, "H row r must end before patch row r + 1 begins");
    __shared__ __attribute__((aligned(16))) uint8_t s_slice[KP_PER_BLOCK][SLICE];

    const Geo &geo = *geo_p;
    // the keypoint is a property of the wavefront: said so, its record, its patch origin and every per-keypoint scalar live on the scalar unit
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    int l = 0, f, out_idx, cx, cy;
    float given_angle = 0.f, resp = 0.f;
    if constexpr (MODE == 2) {
        // caller-given keypoints: wave i describes keypoint i.  BRIEF centre = cvRound(pt * (1 / scale)) in the keypoint's OWN octave
        // (orb.cpp computeOrbDescriptors; the host checked 0 <= octave < nlevels)
        f = frame_base;
        out_idx = (int)blockIdx.x * KP_PER_BLOCK + wv;
        if (out_idx >= cap_per_frame) return;  // wave-uniform
        const afv_keypoint g = given[out_idx];
        l = __builtin_amdgcn_readfirstlane(g.octave);
        const float inv = geo.lv[l].inv_scale;
        cx = __builtin_amdgcn_readfirstlane((int)rintf(g.x * inv));
        cy = __builtin_amdgcn_readfirstlane((int)rintf(g.y * inv));
        given_angle = g.angle;
    } else {
        // XCD-aware placement: all keypoints of a frame are described on one XCD (their patches share L2 lines)
        const int work = afv_xcd_remap(blockIdx.x, total_blocks);
        if (work >= total_blocks) return;
        // a frame's blocks: level after level, ceil(sel_cap / KP_PER_BLOCK) blocks each (per_frame in total)
        const int fl = (int)afv_udiv((uint32_t)work, geo.dv_desc_per_frame);
        const int blk = work - fl * per_frame;
#pragma unroll
        for (int i = 1; i < AFV_MAX_LEVELS; ++i)
            if (i < geo.nlevels && blk >= geo.lv[i].desc_blk_base) l = i;
        const int kblk = blk - geo.lv[l].desc_blk_base;
        f = frame_base + fl;
        const int idx = kblk * KP_PER_BLOCK + wv;

        // frame-level bookkeeping: counts of the eight levels (scalar loads), this level's first output slot, total
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
            if (mir.n) mir.n[f] = min(total, cap_per_frame);
            if (status && total > cap_per_frame) atomicMin(status, AFV_ECAPACITY);
        }
        if (idx >= mine) return;  // wave-uniform
        out_idx = level_base + idx;
        if (out_idx >= cap_per_frame) return;

        const SelPoint sp = sel[(size_t)f * geo.sel_per_frame + geo.lv[l].sel_base + idx];
        cx = sp.x;
        cy = sp.y;
        resp = sp.response;
    }
    const LevelGeo &L = geo.lv[l];
    const int lw = L.w, lh = L.h;
    const uint8_t *img;
    int pitch;
    if (l == 0) {
        img = src0.base + (size_t)f * src0.frame_stride;
        pitch = src0.stride;
    } else {
        img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        pitch = L.pitch;
    }
    uint8_t *P = s_slice[wv] + P_OFF;
    uint16_t *H = reinterpret_cast<uint16_t *>(s_slice[wv]);

    // ---- 1. stage the 43x43 unblurred patch; P[r][a + c] = level(cx-21+c, cy-21+r) with reflect-101 ----
    const int px0 = cx - PR, py0 = cy - PR;
    int a;  // column offset of patch column 0 inside the LDS row
    const bool interior = px0 >= 0 && py0 >= 0 && px0 + 48 <= lw && py0 + PS <= lh;  // wave-uniform
    if (interior) {
        // interior: 12 aligned dwords per row, 5 rows per wave instruction
        a = px0 & 3;
        const int ax0 = px0 - a;
        const int rr = lane / 12, rq = lane - rr * 12;
        if (lane < 60) {
            // 32-bit byte offsets into the level image (a level is far below 4 GB; scalar base + vector offset is what the load takes):
            // the patch origin on the scalar unit, the lane's part by a 24-bit multiply, the row steps by scalar adds - no 64-bit
            // vector multiply-adds (quarter rate) in the address chain
            uint32_t o_patch = (uint32_t)py0 * (uint32_t)pitch + (uint32_t)ax0, o_step = 5u * (uint32_t)pitch;
            asm("" : "+s"(o_patch), "+s"(o_step));  // opaque scalars: the compiler would re-form (py0 + rr + 5 k) * pitch per row
            uint32_t o = o_patch + __umul24((uint32_t)rr, (uint32_t)pitch) + (uint32_t)(rq * 4);
            uint8_t *lp = &P[rr * PP + rq * 4];
            uint32_t v[9];  // rows rr, rr + 5, ..., rr + 40 (PS = 43: the last one exists for rr < 3): loads first, then the LDS writes
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k < 8 || rr < PS - 40) v[k] = *reinterpret_cast<const uint32_t *>(img + o);
                o += o_step;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (k < 8 || rr < PS - 40) *reinterpret_cast<uint32_t *>(lp + k * 5 * PP) = v[k];
        }
    } else {
        // border: lane = patch column (reflected once), rows reflected per iteration (wave-uniform)
        a = 0;
        if (lane < PS) {
            const int x = afv_reflect101(px0 + lane, lw);
            for (int r = 0; r < PS; ++r) {
                const int y = afv_reflect101(py0 + r, lh);  // wave-uniform
                P[r * PP + lane] = img[(uint32_t)y * (uint32_t)pitch + (uint32_t)x];
            }
        }
    }
    wave_sync();  // each wave owns its LDS slice: no workgroup barrier anywhere in this kernel

    // ---- 2. intensity centroid over the radius-15 disc: lane = (row, half) ----
    int m10 = 0, m01 = 0;
    if constexpr (MODE != 2) {
        const int v = (lane >> 1) - 15;  // -15..16
        if (v <= 15) {
            // half 0: u in [-d, -1] = bytes j = 16-d .. 15 of the 16 bytes starting at u = -16; half 1: u in [0, d] = bytes
            // j = 0 .. d of the 16 bytes starting at u = 0 (d = umax[|v|]).  Five aligned dwords, funnel-shifted to the start byte, bytes
            // outside the disc masked to zero (k_ic_masks), then sum(val) and sum(j * val) by v_dot4_u32_u8.
            const int half = lane & 1;
            const int A = (PR + v) * PP + PR + a + (half ? 0 : -16);
            const uint32_t *wp = reinterpret_cast<const uint32_t *>(P + (A & ~3));
            const int sh = (A & 3) * 8;
            const uint4 mk = *reinterpret_cast<const uint4 *>(k_ic_masks.m[lane]);
            const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4];
            const uint32_t b[4] = {__builtin_amdgcn_alignbit(w1, w0, sh) & mk.x, __builtin_amdgcn_alignbit(w2, w1, sh) & mk.y,
                                   __builtin_amdgcn_alignbit(w3, w2, sh) & mk.z, __builtin_amdgcn_alignbit(w4, w3, sh) & mk.w};
            uint32_t s1 = 0, sj = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1 = __builtin_amdgcn_udot4(b[q], 0x01010101u, s1, false);
                sj = __builtin_amdgcn_udot4(b[q], 0x03020100u + 0x04040404u * (uint32_t)q, sj, false);
            }
            m10 = half ? (int)sj : (int)sj - 16 * (int)s1;
            m01 = v * (int)s1;
        }
    }
    if constexpr (MODE != 1) blur_rows(P, H, lane);  // overwrites the patch (see the slice layout above): everything that reads P comes before this line
    float angle = given_angle;
    if constexpr (MODE != 2) {
        m10 = wave_sum(m10);
        m01 = wave_sum(m01);
        angle = fast_atan2_deg((float)m01, (float)m10);
    }
    const float ptx = (float)cx * L.scale, pty = (float)cy * L.scale;
    if constexpr (MODE == 1) {  // keypoints only
        if (lane == 0) {
            afv_keypoint k;
            k.x = ptx;
            k.y = pty;
            k.size = 31 * L.scale;
            k.angle = angle;
            k.response = resp;
            k.octave = l;
            k.class_id = -1;
            kps[(size_t)f * cap_per_frame + out_idx] = k;
        }
        return;
    }

    // ---- 3+4. rotated BRIEF on the blurred patch: lane handles tests lane, lane+64, lane+128, lane+192; the blur is
    // evaluated only where a test samples it (512 of the 1369 patch positions) ----
    float ca, sb;
    sincos_deg(angle, ca, sb);
    wave_sync();  // row-filtered plane complete
    // BRIEF centre = cvRound(pt * (1/scale)) with pt = level coordinate * scale (orb.cpp computeOrbDescriptors); a given keypoint's
    // centre IS that rounding already
    const int bx = MODE == 2 ? cx : (int)rintf(ptx * L.inv_scale), by = MODE == 2 ? cy : (int)rintf(pty * L.inv_scale);
    const int ox = bx - cx, oy = by - cy;  // 0 in practice; kept literal
    const int kox = ox - 0x4B400000, koy = oy - 0x4B400000;  // scalar: the mantissa offset of the rounding trick below and the (0) centre offset
    uint32_t words[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int t = g * 64 + lane;
        const float4 pt = reinterpret_cast<const float4 *>(k_brief_pattern)[t];
        const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
        // (x ca - y sb, x sb + y ca) in PLAIN fp32, one rounding per operator (a - b = a + (-b), (-s) y = -(s y); -ffp-contract=off).
        // Round 5 held this as three packed instructions per point ((x, x) * (ca, sb) + (y, y) * (-sb, ca)); round 6 found that
        // v_pk_mul_f32 with a broadcast of its second operand's register (op_sel / op_sel_hi) returns the product of the OTHER register
        // in lanes 48..63 now and then while a wavefront of an MFMA kernel of another queue shares the SIMD (one wrong descriptor in a
        // few hundred frames beside k_match_topk_mfma; tools/probes/probe_pk_real.hip reproduces it in seconds, DESIGN_LOG round 6).
        // The whole library is built without packed fp32 instructions (build.py); this kernel does not ask for them either.
        const float rx0 = x0 * ca + y0 * -sb, ry0 = x0 * sb + y0 * ca, rx1 = x1 * ca + y1 * -sb, ry1 = x1 * sb + y1 * ca;
        // cvRound = round half to even: x + 1.5 * 2^23 leaves the rounded integer in the mantissa (|x| < 2^22; the addition rounds to
        // nearest even exactly where rintf does), one add per coordinate instead of v_rndne + v_cvt; the integer offset 0x4B400000
        // folds into the address constants below
        const float rmag = 12582912.0f;
        const int ix0 = __float_as_int(rx0 + rmag) + kox, iy0 = __float_as_int(ry0 + rmag) + koy;
        const int ix1 = __float_as_int(rx1 + rmag) + kox, iy1 = __float_as_int(ry1 + rmag) + koy;
        // inside the ROI -> blurred, outside -> unblurred apron (a patch that lies inside the level has no outside samples)
        int t0, t1;
        if (interior) {
            t0 = blur_at(H, PR + a + ix0, PR + iy0);
            t1 = blur_at(H, PR + a + ix1, PR + iy1);
        } else {
            const int gx0 = cx + ix0, gy0 = cy + iy0, gx1 = cx + ix1, gy1 = cy + iy1;
            t0 = (gx0 >= 0 && gx0 < lw && gy0 >= 0 && gy0 < lh) ? blur_at(H, PR + a + ix0, PR + iy0)
                                                                 : (int)img[__umul24((uint32_t)afv_reflect101(gy0, lh), (uint32_t)pitch) + (uint32_t)afv_reflect101(gx0, lw)];
            t1 = (gx1 >= 0 && gx1 < lw && gy1 >= 0 && gy1 < lh) ? blur_at(H, PR + a + ix1, PR + iy1)
                                                                 : (int)img[__umul24((uint32_t)afv_reflect101(gy1, lh), (uint32_t)pitch) + (uint32_t)afv_reflect101(gx1, lw)];
        }
        const unsigned long long m = __ballot(t0 < t1);
        words[2 * g] = (uint32_t)m;
        words[2 * g + 1] = (uint32_t)(m >> 32);
    }
    // ---- 5. outputs (E11 merge: ascending level, list order inside a level) ----
    const size_t o = MODE == 2 ? (size_t)out_idx : (size_t)f * cap_per_frame + out_idx;
    if (lane < 8) {
        uint32_t w = words[0];
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (lane == i) w = words[i];
        reinterpret_cast<uint32_t *>(desc + o * 32)[lane] = w;
        if (mir.desc) reinterpret_cast<uint32_t *>(mir.desc + o * 32)[lane] = w;  // afv_frame_extract: the frame's device copy
    }
    if (MODE == 0 && lane == 0) {
        afv_keypoint k;
        k.x = ptx;
        k.y = pty;
        k.size = 31 * L.scale;
        k.angle = angle;
        k.response = resp;
        k.octave = l;
        k.class_id = -1;
        kps[o] = k;
        if (mir.kps) mir.kps[o] = k;
    }
}

__global__ __launch_bounds__(64 * KP_PER_BLOCK) void k_describe(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                                const SelPoint *__restrict__ sel, const int *__restrict__ sel_count,
                                                                afv_keypoint *__restrict__ kps, uint8_t *__restrict__ desc, int cap_per_frame,
                                                                int *__restrict__ n_out, int *__restrict__ status, int frame_base, int per_frame,
                                                                int total_blocks, DescribeMirror mir) {
    describe_body<0>(geo_p, src0, pyr, sel, sel_count, kps, desc, cap_per_frame, n_out, status, frame_base, per_frame, total_blocks, mir, nullptr);
}
__global__ __launch_bounds__(64 * KP_PER_BLOCK) void k_describe_angles(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                                       const SelPoint *__restrict__ sel, const int *__restrict__ sel_count,
                                                                       afv_keypoint *__restrict__ kps, int cap_per_frame, int *__restrict__ n_out,
                                                                       int *__restrict__ status, int frame_base, int per_frame, int total_blocks) {
    describe_body<1>(geo_p, src0, pyr, sel, sel_count, kps, nullptr, cap_per_frame, n_out, status, frame_base, per_frame, total_blocks,
                     DescribeMirror{nullptr, nullptr, nullptr}, nullptr);
}
__global__ __launch_bounds__(64 * KP_PER_BLOCK) void k_describe_given(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                                      const afv_keypoint *__restrict__ given, int n, uint8_t *__restrict__ desc,
                                                                      int frame) {
    describe_body<2>(geo_p, src0, pyr, nullptr, nullptr, nullptr, desc, n, nullptr, nullptr, frame, 0, 0, DescribeMirror{nullptr, nullptr, nullptr}, given);
}

extern "C" int afv_describe_blocks_per_frame(const Geo *g) {
    int n = 0;
    for (int l = 0; l < g->nlevels; ++l) n += (g->lv[l].sel_cap + KP_PER_BLOCK - 1) / KP_PER_BLOCK;
    return n;
}

// `desc` == nullptr: keypoints only (afv_orb_detect)
extern "C" void afv_launch_describe(const Geo *geo_dev, int blocks_per_frame, const FrameSrc *src0, const uint8_t *pyr,
                                    const SelPoint *sel, const int *sel_count, afv_keypoint *kps, uint8_t *desc,
                                    int cap_per_frame, int *n_out, int *status, int frame_base, int nframes, const DescribeMirror *mirror,
                                    hipStream_t stream) {
    const int total = blocks_per_frame * nframes;
    dim3 grid((total + 7) / 8 * 8);
    if (!desc) {
        hipLaunchKernelGGL(k_describe_angles, grid, dim3(64 * KP_PER_BLOCK), 0, stream, geo_dev, *src0, pyr, sel, sel_count, kps, cap_per_frame, n_out,
                           status, frame_base, blocks_per_frame, total);
        return;
    }
    const DescribeMirror mir = mirror ? *mirror : DescribeMirror{nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_describe, grid, dim3(64 * KP_PER_BLOCK), 0, stream, geo_dev, *src0, pyr, sel, sel_count, kps, desc,
                       cap_per_frame, n_out, status, frame_base, blocks_per_frame, total, mir);
}

// descriptors of n caller-given keypoints (device array) on the pyramid of frame `frame`
extern "C" void afv_launch_describe_given(const Geo *geo_dev, const FrameSrc *src0, const uint8_t *pyr, const afv_keypoint *given, int n, uint8_t *desc,
                                          int frame, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_describe_given, dim3((n + KP_PER_BLOCK - 1) / KP_PER_BLOCK), dim3(64 * KP_PER_BLOCK), 0, stream, geo_dev, *src0, pyr, given, n, desc,
                       frame);
}

// ---------------- standalone E9: blur one level of one frame (debug / parity of the blur arithmetic) ----------------
__global__ __launch_bounds__(256) void k_blur_level(const uint8_t *__restrict__ img, int w, int h, int pitch,
                                                    uint8_t *__restrict__ out) {
    __shared__ uint8_t t[(16 + 6) * 72];
    __shared__ uint16_t hh[(16 + 6) * 64];
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 22 * 70; i += 256) {
        const int r = i / 70, c = i - r * 70;
        const int y = min(max(afv_reflect101(y0 + r - 3, h), 0), h - 1), x = min(max(afv_reflect101(x0 + c - 3, w), 0), w - 1);
        t[r * 72 + c] = img[(size_t)y * pitch + x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 22 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t *p = &t[r * 72 + c];
        hh[r * 64 + c] = (uint16_t)(18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint16_t *q = &hh[r * 64 + c];
        const int S = 18 * ((int)q[0] + q[6 * 64]) + 34 * ((int)q[64] + q[5 * 64]) + 49 * ((int)q[2 * 64] + q[4 * 64]) + 55 * (int)q[3 * 64];
        if (x0 + c < w && y0 + r < h) out[(size_t)(y0 + r) * w + x0 + c] = (uint8_t)blur_round((uint32_t)S);
    }
}

extern "C" void afv_launch_blur_level(const uint8_t *img, int w, int h, int pitch, uint8_t *out, hipStream_t stream) {
    dim3 grid((w + 63) / 64, (h + 15) / 16);
    hipLaunchKernelGGL(k_blur_level, grid, dim3(256), 0, stream, img, w, h, pitch, out);
}
