// k_fast.hip — E3: FAST-9/16 score and 3x3 non-max suppression.
//
// Replaces, for every pyramid level of every frame of the batch, the FastFeatureDetector(20, true) call inside cv::ORB::detect
// (Feature_orb32.cpp:34; OpenCV fast.cpp FAST_t<16> + cornerScore<16>).  The Harris responses cv::ORB computes next are
// k_harris.hip's job: they are only needed for the candidates that survive retainBest on the FAST score, which is known once
// every tile of a level is done.  One 256-thread workgroup owns a 64x32 tile:
//   1. the tile plus a 4 px halo (ring radius 3 + the NMS neighbour) is staged in LDS with coalesced dword loads;
//   2a. OpenCV's necessary pre-test (each of the 4 even antipodal ring pairs must hold a brighter / a darker pixel) on
//      the 66x34 ring-extended tile, TWO horizontally adjacent pixels per lane as packed u16 pairs: per polarity
//      min over the pairs of max(a, b) - v > t  /  v - max over the pairs of min(a, b) > t.  A thread's 5 rows share their
//      operands (a row's centre pair is the vertical ring pair three rows up and down, ...): 16 LDS reads of aligned dwords
//      per thread.  Survivors go to two LDS lists (bright / dark), one entry per (pixel, polarity);
//   2b. the exact corner score = largest threshold for which the pixel is still a 9-arc corner, two list entries per
//      lane (the packed halves now carry two different pixels of one polarity each): 16 ring differences,
//      van-Herk prefix/suffix minima over the two ring halves (59 packed min/max), corner iff score > threshold; scores go
//      to an LDS score plane (a pixel is a corner in at most one polarity: two 9-arcs of a 16-ring overlap);
//   3. strict 3x3 maxima, dense over the score plane (again two pixels per lane, packed max; a thread owns 4 consecutive rows
//      and reduces each score row once), compacted into an LDS list (<= 512 per tile);
//   4. one global atomic per tile reserves output slots in the (frame, level) candidate array.
// Candidates leave the kernel unordered; everything downstream is order-independent (ties are broken by the
// raster index), see DESIGN.md "canonical order".
#include <type_traits>

#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2v pkmin(short2v a, short2v b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2v pkmax(short2v a, short2v b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ uint32_t as_u32(short2v v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_shr1(uint32_t v) {  // v_pk_lshrrev_b16: both 16-bit halves >> 1
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ushort2v, v) >> (unsigned short)1);
}
__device__ __forceinline__ short2v pk_sar15(short2v v) { return v >> (short)15; }  // v_pk_ashrrev_i16: 0xffff where negative

// ---- cross-lane helpers on DPP (no LDS round trips) ----
#define DPP_QUAD_XOR1 0xB1       // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E       // quad_perm:[2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
// inclusive prefix sum over the 64 lanes of a wavefront: 4 Hillis-Steele steps inside each row of 16 lanes, then the row
// totals are carried over with row_bcast15 (rows 1, 3) and row_bcast31 (rows 2, 3)
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}
// one lane's LDS fetch-add as ONE instruction (the compiler's wave aggregation of atomics - mbcnt, compare, popcount, multiply - is dead
// weight when a single lane is active by construction)
__device__ __forceinline__ int lds_add_rtn_one_lane(int *p, int v) {
    int r;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int *)p;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory");
    return r;
}
// LDS row pitch of the tile and of the score plane: the 72 staged bytes of a row + 4 bytes of padding.  19 dwords per row instead of 18
// skews the bank pattern of the scattered byte reads of stage 2b (round 4, tools/experiments.py with -DFT_PITCH=n: 72 -> 76 bytes
// = 5 % fewer bank-conflict cycles, kernel -2.2 %; 80 and 84 are worse than 72).  Must be a multiple of 4.
#ifndef FT_PITCH
#define FT_PITCH (FT_LW + 4)
#endif
static_assert(FT_PITCH >= FT_LW && FT_PITCH % 4 == 0, "tile pitch");
#define RING_OFF(dx, dy) ((dy) * FT_PITCH + (dx))
#define PRE_ROWS (FT_H + 2)             // LDS rows 3 .. 36 (tile pixels -1 .. 32)
#define PRE_PAIRS ((FT_W + 4) / 2)      // LDS columns 2 .. 69 as 34 pixel pairs at EVEN columns (aligned 16-bit LDS reads); columns
                                        // 2 and 69 are outside the ring-extended tile and masked out
#define PRE_GROUPS (256 / PRE_PAIRS)      // 7 row groups of 34 threads
#define PRE_ITERS ((PRE_ROWS + PRE_GROUPS - 1) / PRE_GROUPS)  // rows rg, rg + 7, ... : 5 iterations
#define PRE_MAX (2 * (PRE_GROUPS * PRE_ITERS) * 2 * PRE_PAIRS)  // one entry per (pixel, polarity) of every row the row groups compute (PRE_ROWS + the padding row: its pixels can make entries since step 2a stopped testing rows)
#define PRE_DARK0 (PRE_MAX / 2)  // first entry of the dark list (both lists grow upwards, each can hold every pixel)
#define PRE_PAD_ROWS (PRE_GROUPS * PRE_ITERS - PRE_ROWS)  // rows past the tile the masked iterations of the last row group touch (1 at FT_H = 32)
#define ST_GROUPS 14                                       // staging: 18 dword columns x 14 row groups = 252 threads
#define ST_ITERS ((FT_LH + ST_GROUPS - 1) / ST_GROUPS)     // rows rs, rs + 14, ... (3 at FT_H = 32)
static_assert(PRE_ITERS <= 16, "pre-test flags live in 16-bit halves");
static_assert(ST_ITERS >= 2 && ST_GROUPS * (ST_ITERS - 1) <= FT_LH - 1, "staging rows");

// two horizontally adjacent tile bytes as a packed u16 pair: one 16-bit LDS read (any alignment) + one byte permute
__device__ __forceinline__ short2v ld_pair(const uint8_t *c, int off) {
    unsigned short w;
    __builtin_memcpy(&w, c + off, 2);
    return as_s2(__builtin_amdgcn_perm(0u, (uint32_t)w, 0x0c010c00u));
}
// the same ring position of two different pixels
__device__ __forceinline__ short2v ld_two(const uint8_t *c0, const uint8_t *c1, int off) {
    short2v r;
    r.x = (short)c0[off];
    r.y = (short)c1[off];
    return r;
}

// score = max over the 16 arcs of 9 contiguous ring pixels of min(d) where d = ring - v (bright list) or v - ring (dark
// list); the two halves of every register belong to two different list entries of the same polarity.  v is the same for all 16
// ring pixels, so it leaves the network: bright  max_arcs min_arc (r - v) = (max_arcs min_arc r) - v,  dark  max_arcs min_arc (v - r) =
// v - (min_arcs max_arc r) - the network runs on the RAW ring values (0 .. 255 in the u16 halves) with min / max swapped for the dark
// list, and v is subtracted once at the end instead of once per ring pixel (round 5: 16 packed subtractions per lane and list pair less).
#ifndef FT_NO_MIN3  // (-DFT_NO_MIN3: the van Herk network on two-input packed i16 min / max of rounds 3-5, kept for A / B runs)
// Round 6: the ring values 0 .. 255 in u16 halves are, read as f16 bit patterns, ordered subnormals, so gfx950's
// three-input packed v_pk_minimum3_f16 / v_pk_maximum3_f16 compute the same packed min / max as the i16 forms (f16 denormals are preserved
// in the kernel's MODE; no NaN can occur: the exponent field is 0).  Triples t_k = m3(d_k, d_k+1, d_k+2), windows w_k = m3(t_k, t_k+3,
// t_k+6) = the 9 ring pixels from k, outer reduction by threes: 16 + 16 + 8 = 40 packed instructions instead of van Herk's 59.
// Measured (A / B on one box, twice): vector instructions of the kernel - 6.1 %, 311.4 -> 298.5 us per launch (- 4.2 %), bit-exact.
__device__ __forceinline__ short2v pkmin3(short2v a, short2v b, short2v c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(as_u32(a)), "v"(as_u32(b)), "v"(as_u32(c)));
    return as_s2(r);
}
__device__ __forceinline__ short2v pkmax3(short2v a, short2v b, short2v c) {
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(as_u32(a)), "v"(as_u32(b)), "v"(as_u32(c)));
    return as_s2(r);
}
template <bool DARK>
__device__ __forceinline__ short2v fast_score2(const uint8_t *c0, const uint8_t *c1, short2v V) {
    auto inner3 = [](short2v a, short2v b, short2v c) { return DARK ? pkmax3(a, b, c) : pkmin3(a, b, c); };
    auto outer3 = [](short2v a, short2v b, short2v c) { return DARK ? pkmin3(a, b, c) : pkmax3(a, b, c); };
    short2v d[16], t[16], w[16];
#define RD(k, dx, dy) d[k] = ld_two(c0, c1, RING_OFF(dx, dy))
    RD(0, 0, 3); RD(1, 1, 3); RD(2, 2, 2); RD(3, 3, 1); RD(4, 3, 0); RD(5, 3, -1); RD(6, 2, -2); RD(7, 1, -3);
    RD(8, 0, -3); RD(9, -1, -3); RD(10, -2, -2); RD(11, -3, -1); RD(12, -3, 0); RD(13, -3, 1); RD(14, -2, 2); RD(15, -1, 3);
#undef RD
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = inner3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = inner3(t[k], t[(k + 3) & 15], t[(k + 6) & 15]);
    short2v o[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = outer3(w[3 * i], w[3 * i + 1], w[3 * i + 2]);
    const short2v best = outer3(outer3(o[0], o[1], o[2]), outer3(o[3], o[4], w[15]), w[15]);
    return DARK ? (V - best) : (best - V);
}
#else
template <bool DARK>
__device__ __forceinline__ short2v fast_score2(const uint8_t *c0, const uint8_t *c1, short2v V) {
    auto inner = [](short2v a, short2v b) { return DARK ? pkmax(a, b) : pkmin(a, b); };  // over the pixels of an arc
    auto outer = [](short2v a, short2v b) { return DARK ? pkmin(a, b) : pkmax(a, b); };  // over the arcs
    short2v d[16];
#define RD(k, dx, dy) d[k] = ld_two(c0, c1, RING_OFF(dx, dy))
    RD(0, 0, 3); RD(1, 1, 3); RD(2, 2, 2); RD(3, 3, 1); RD(4, 3, 0); RD(5, 3, -1); RD(6, 2, -2); RD(7, 1, -3);
    RD(8, 0, -3); RD(9, -1, -3); RD(10, -2, -2); RD(11, -3, -1); RD(12, -3, 0); RD(13, -3, 1); RD(14, -2, 2); RD(15, -1, 3);
#undef RD
    // van Herk: window k (9 long, circular) = suffix of its ring half starting at k + prefix of the other half
    short2v suf0[8], pre0[8], suf1[8], pre1[8];
    suf0[7] = d[7];
    suf1[7] = d[15];
    pre0[0] = d[0];
    pre1[0] = d[8];
#pragma unroll
    for (int k = 6; k >= 0; --k) {
        suf0[k] = inner(d[k], suf0[k + 1]);
        suf1[k] = inner(d[8 + k], suf1[k + 1]);
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        pre0[k] = inner(d[k], pre0[k - 1]);
        pre1[k] = inner(d[8 + k], pre1[k - 1]);
    }
    short2v best = inner(suf0[0], pre1[0]);
#pragma unroll
    for (int k = 1; k < 8; ++k) best = outer(best, inner(suf0[k], pre1[k]));
#pragma unroll
    for (int k = 0; k < 8; ++k) best = outer(best, inner(suf1[k], pre0[k]));
    return DARK ? (V - best) : (best - V);
}
#endif

__global__ __launch_bounds__(256) void k_fast_nms(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                  uint32_t *__restrict__ cand_packed, int *__restrict__ cand_count, int total_blocks,
                                                  int frame_base) {
    // + pad: masked pre-test positions (the rows the last row group computes beyond PRE_ROWS, column 69) read up to PRE_PAD_ROWS rows + 2 bytes past the last row
    __shared__ __attribute__((aligned(16))) uint8_t tile[(FT_LH + PRE_PAD_ROWS) * FT_PITCH + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sc[FT_LH * FT_PITCH];  // same geometry as `tile`
    __shared__ __attribute__((aligned(4))) unsigned short pre[PRE_MAX];  // bright entries from the front, dark entries from the back
    __shared__ int list_n, out_base, pre_nb, pre_nd;
    uint32_t *list = reinterpret_cast<uint32_t *>(pre);  // NMS survivors (<= 512): reuses `pre`, which is dead after step 2b

    const Geo &geo = *geo_p;
    // XCD-aware placement: every XCD works on whole frames, so the halo / cache-line sharing between neighbouring
    // tiles stays inside one L2
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int f0 = (int)afv_udiv((uint32_t)work, geo.dv_total_tiles), tile_id = work - f0 * geo.total_tiles;
    const int f = frame_base + f0;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && tile_id >= geo.lv[i].tile_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int t = tile_id - L.tile_base;
    const int tyi = (int)afv_udiv((uint32_t)t, L.dv_tiles_x), txi = t - tyi * L.tiles_x;
    const int gx0 = txi * FT_W - FT_HALO, gy0 = tyi * FT_H - FT_HALO;
    const int lw = L.w, lh = L.h;
    const int tid = threadIdx.x, lane = tid & 63;

    const uint8_t *img;
    int pitch;
    if (l == 0) {
        img = src0.base + (size_t)f * src0.frame_stride;
        pitch = src0.stride;
    } else {
        img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        pitch = L.pitch;
    }
    if (tid == 0) {
        list_n = 0;
        pre_nb = 0;
        pre_nd = 0;
    }
    // the padding row(s) behind the staged window: read by the last row group's padding iteration (its entries land where step 3 never
    // looks) - cleared so that not even the COUNT of entries depends on what an earlier workgroup left in LDS
    if (tid < PRE_PAD_ROWS * (FT_PITCH / 4)) reinterpret_cast<uint32_t *>(&tile[FT_LH * FT_PITCH])[tid] = 0u;

    // 1. stage 72x40 bytes and clear the score plane.  Thread (rq, rs) = (tid % 18, tid / 18) owns the dword column rq of the rows
    //    rs, rs + 14, rs + 28 (252 of the 256 threads): the column part of the address (and its reflect-101 handling at the left /
    //    right image edge) is computed once, each row costs one reflect + one 32-bit load + two LDS writes.
    {
        const int rs = (int)(__umul24((unsigned)tid, 3641u) >> 16);  // tid / 18 (exact for tid < 256)
        const int rq = tid - rs * (FT_LW / 4);
        const int gx = gx0 + rq * 4;
        const bool interior = gx >= 0 && gx + 3 < lw;
        if (rs < ST_GROUPS) {
            // the row offsets first, then the loads back to back (one global round trip), then the LDS writes;
            // the last row of a thread may not exist (FT_H = 32: row 28 + rs only for rs < 12): its load is redirected to the last row and its write dropped
            uint32_t roff[ST_ITERS], v[ST_ITERS];
            // (24-bit multiplies: rows and pitches are far below 2^24 and v_mul_lo_u32 is a quarter-rate instruction)
            if (gy0 >= 0 && gy0 + FT_LH <= lh) {  // tile-uniform: no row of the staged window leaves the level (all but the top / bottom tile rows)
                uint32_t r0 = (uint32_t)gy0 * (uint32_t)pitch;  // scalar unit
                asm("" : "+s"(r0));                             // opaque: keeps (gy0 + ry) * pitch from being re-formed as one 32-bit vector multiply
                uint32_t pg = (uint32_t)(ST_GROUPS * pitch);  // scalar, opaque for the same reason
                asm("" : "+s"(pg));
                roff[0] = r0 + __umul24((uint32_t)rs, (uint32_t)pitch);
#pragma unroll
                for (int k3 = 1; k3 < ST_ITERS - 1; ++k3) roff[k3] = roff[k3 - 1] + pg;  // rs + ST_GROUPS * k3 <= FT_LH - 1 for these
                roff[ST_ITERS - 1] = r0 + __umul24((uint32_t)min(rs + ST_GROUPS * (ST_ITERS - 1), FT_LH - 1), (uint32_t)pitch);
            } else {
#pragma unroll
                for (int k3 = 0; k3 < ST_ITERS; ++k3) {
                    const int ry = min(rs + ST_GROUPS * k3, FT_LH - 1);
                    const int gy = min(max(afv_reflect101(gy0 + ry, lh), 0), lh - 1);
                    roff[k3] = __umul24((uint32_t)gy, (uint32_t)pitch);
                }
            }
            if (interior) {
#pragma unroll
                for (int k3 = 0; k3 < ST_ITERS; ++k3) v[k3] = *reinterpret_cast<const uint32_t *>(img + (roff[k3] + (uint32_t)gx));
            } else {  // left / right image edge: four reflected byte gathers per row
                uint32_t xs[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) xs[k] = (uint32_t)min(max(afv_reflect101(gx + k, lw), 0), lw - 1);
#pragma unroll
                for (int k3 = 0; k3 < ST_ITERS; ++k3) {
                    const uint8_t *row = img + roff[k3];
                    v[k3] = (uint32_t)row[xs[0]] | ((uint32_t)row[xs[1]] << 8) | ((uint32_t)row[xs[2]] << 16) | ((uint32_t)row[xs[3]] << 24);
                }
            }
#pragma unroll
            for (int k3 = 0; k3 < ST_ITERS; ++k3) {
                const int ry = rs + ST_GROUPS * k3;
                if (ry < FT_LH) {
                    *reinterpret_cast<uint32_t *>(&tile[ry * FT_PITCH + rq * 4]) = v[k3];
                    *reinterpret_cast<uint32_t *>(&sc[ry * FT_PITCH + rq * 4]) = 0u;
                }
            }
        }
    }
    __syncthreads();

#if defined(AFV_FAST_STOP) && AFV_FAST_STOP == 1
    if (tile[tid] == 255 && sc[tid] == 77) cand_count[0] = 1;  // keep the staging alive
    return;
#endif
    // 2a. pre-test on LDS rows 3..36 x columns 3..68 (= tile pixels -1..32 x -1..64).  Thread (pr, rg) = (tid % 34, tid / 34)
    //     owns the pixel pair at columns 2 + 2 pr, 3 + 2 pr of the rows 5 rg .. 5 rg + 4 (238 of the 256 threads; threads 238 .. 255
    //     repeat row group 6 with every pixel declared invalid): the column never changes, so every LDS read is base + immediate.
    //     The pass flags are the sign bits of (v + t) - (min over pairs of max(a, b)) [bright] and (max over pairs of min(a, b)) -
    //     (v - t) [dark].  Validity costs nothing inside the loop (round 5): a pixel outside the FAST border of the level in x (or outside
    //     the ring-extended tile) gets the threshold 0x3fff in ITS half of T0 and can never pass; rows are not tested here at all - rows
    //     outside the FAST border in y exist in the top / bottom tiles of a level only and their scores are cleared by step 3 before
    //     anything looks at them, and the one row the last row group computes past the ring-extended tile (LDS row 38) lies where
    //     step 3 never reads.
    const int thr = geo.fast_threshold;
    {
        const int rg0 = (int)(__umul24((unsigned)tid, 1928u) >> 16);  // tid / 34 (exact for tid < 256)
        const int rg = min(rg0, PRE_GROUPS - 1);
        const int pr = tid - rg0 * PRE_PAIRS;  // threads >= 238: 0 .. 17 (in range, unused)
        const int col = 2 + 2 * pr, gx = gx0 + col;
        short2v T0;
        T0.x = (short)((rg0 < PRE_GROUPS && pr > 0 && gx >= 3 && gx < lw - 3) ? thr : 0x3fff);
        T0.y = (short)((rg0 < PRE_GROUPS && pr < PRE_PAIRS - 1 && gx + 1 >= 3 && gx + 1 < lw - 3) ? thr : 0x3fff);
        const int p00 = (rg * PRE_ITERS + 3) * FT_PITCH + col;
        const uint8_t *c = &tile[p00];
        const int sh = (col - 4) & 3;  // 0 or 2
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(c - 4 - sh);
        // (one product, then additions of literals: a multiply-add per selector needs its constant in a register first)
        uint32_t shm = (uint32_t)sh * 0x00010001u;
        asm("" : "+v"(shm));  // opaque: the compiler would fold every `literal + sh * 0x10001` back into a multiply-add with the literal moved into a register first
        const uint32_t shm2 = shm ^ 0x00020002u;  // sh2 = sh ^ 2 in both halves
        const uint32_t sel_b = 0x0c020c01u + shm;  // (col-3, col-2) out of (w1:w0)
        const uint32_t sel_v = 0x0c050c04u + shm;  // (col,   col+1) out of (w1:w0)
        const uint32_t sel_a = 0x0c040c03u + shm;  // (col+3, col+4) out of (w2:w1)
        // Operand fetch, LDS instructions counted (the LDS pipe is as busy as the vector ALU in this kernel): the thread's rows r = -3 ..
        // PRE_ITERS + 2 around its PRE_ITERS centre rows.  The centre pair of row r is the vertical ring pair of rows r - 3 and r + 3, the
        // pairs at columns -2 / +2 of row r serve rows r - 2 and r + 2, and all of them lie in the aligned dwords read for the row anyway:
        // centre rows = 3 dwords (bytes col-4-sh .. col+7-sh, two instructions), the two rows above / below = 2 dwords (bytes col-2-sh2 ..
        // col+5-sh2, one instruction), rows -3 and PRE_ITERS + 2 = the centre pair alone.  16 LDS instructions per thread (35 when every
        // pair was read by itself); every pair is cut out with one byte permute whose selector depends on col & 2 only.
        const int sh2 = sh ^ 2;
        const uint32_t *cw2 = reinterpret_cast<const uint32_t *>(c - 2 - sh2);
        const uint32_t sel_l = 0x0c030c02u + shm;    // (col-2, col-1) out of (w1:w0); (col+2, col+3) out of (w2:w1)
        const uint32_t sel2_l = 0x0c010c00u + shm2;  // two-dword rows: (col-2, col-1)
        const uint32_t sel2_v = 0x0c030c02u + shm2;  //                 (col,   col+1)
        const uint32_t sel2_r = 0x0c050c04u + shm2;  //                 (col+2, col+3)
        short2v Vc[PRE_ITERS + 6], R2[PRE_ITERS + 4], L2[PRE_ITERS + 4], A4[PRE_ITERS], B4[PRE_ITERS];  // Vc[k]: row k - 3; R2 / L2[k]: row k - 2
        Vc[0] = ld_pair(c, -3 * FT_PITCH);
        Vc[PRE_ITERS + 5] = ld_pair(c, (PRE_ITERS + 2) * FT_PITCH);
#pragma unroll
        for (int r = -2; r < PRE_ITERS + 2; ++r) {
            if (r >= 0 && r < PRE_ITERS) {
                const int o = r * FT_PITCH;
                const uint32_t w0 = cw[o / 4], w1 = cw[o / 4 + 1], w2 = cw[o / 4 + 2];
                Vc[r + 3] = as_s2(__builtin_amdgcn_perm(w1, w0, sel_v));
                A4[r] = as_s2(__builtin_amdgcn_perm(w2, w1, sel_a));
                B4[r] = as_s2(__builtin_amdgcn_perm(w1, w0, sel_b));
                L2[r + 2] = as_s2(__builtin_amdgcn_perm(w1, w0, sel_l));
                R2[r + 2] = as_s2(__builtin_amdgcn_perm(w2, w1, sel_l));
            } else {
                const int o = r * FT_PITCH;  // FT_PITCH is a multiple of 4
                const uint32_t w0 = cw2[o / 4], w1 = cw2[o / 4 + 1];
                L2[r + 2] = as_s2(__builtin_amdgcn_perm(w1, w0, sel2_l));
                Vc[r + 3] = as_s2(__builtin_amdgcn_perm(w1, w0, sel2_v));
                R2[r + 2] = as_s2(__builtin_amdgcn_perm(w1, w0, sel2_r));
            }
        }
        // flags as bits of two registers (bit 16 - PRE_ITERS + it = pixel 0, bit 32 - PRE_ITERS + it = pixel 1 of iteration it), compaction:
        // exclusive prefix of the per-lane survivor counts (both polarities packed in one register, one DPP scan), two LDS atomics per
        // wavefront, then every lane writes its own <= 10 + 10 entries with predicated stores (below).  A lane's entries are a 2 x 5 pixel block and
        // neighbouring lanes hold neighbouring column pairs, so list neighbours stay image neighbours and the scattered ring reads of
        // step 2b hit few LDS bank windows.
        uint32_t bb = 0, bd = 0;
#pragma unroll
        for (int it = 0; it < PRE_ITERS; ++it) {
            const short2v V = Vc[it + 3], a0 = Vc[it + 6], b0 = Vc[it];
            const short2v a2 = R2[it + 4], b2 = L2[it], a6 = R2[it], b6 = L2[it + 4];
            const short2v a4 = A4[it], b4 = B4[it];
#if !defined(FT_NO_MIN3) && !defined(FT_PRE_MIN2)  // the outer reduction over the four pairs with one three-input instruction: 12 instead of 14 per pixel pair
            const short2v hi = pkmin3(pkmax(a0, b0), pkmax(a2, b2), pkmin(pkmax(a4, b4), pkmax(a6, b6)));
            const short2v lo = pkmax3(pkmin(a0, b0), pkmin(a2, b2), pkmax(pkmin(a4, b4), pkmin(a6, b6)));
#else
            const short2v hi = pkmin(pkmin(pkmax(a0, b0), pkmax(a2, b2)), pkmin(pkmax(a4, b4), pkmax(a6, b6)));
            const short2v lo = pkmax(pkmax(pkmin(a0, b0), pkmin(a2, b2)), pkmax(pkmin(a4, b4), pkmin(a6, b6)));
#endif
            const uint32_t bit = 0x00010001u << (16 - PRE_ITERS + it);
            bb |= as_u32(pk_sar15((V + T0) - hi)) & bit;
            bd |= as_u32(pk_sar15(lo - (V - T0))) & bit;
        }
        const int cnt = __popc(bb) | (__popc(bd) << 16);
        const int incl = wave_incl_scan(cnt);
        int base_b = 0, base_d = 0;
        if (lane == 63) {
            base_b = lds_add_rtn_one_lane(&pre_nb, incl & 0xffff);
            base_d = lds_add_rtn_one_lane(&pre_nd, incl >> 16);
        }
        const int excl = incl - cnt;
        unsigned short *wb = pre + __builtin_amdgcn_readlane(base_b, 63) + (excl & 0xffff);
        unsigned short *wd = pre + PRE_DARK0 + __builtin_amdgcn_readlane(base_d, 63) + (excl >> 16);
        // the flags leave their registers through the carry: x + x shifts the next flag out of bit 31 (v_add_co_u32: shift AND test in one
        // instruction), the store runs under that carry, the pointer advances by it - three vector instructions per slot instead of
        // four (mask, compare, extract, add).  Slot order: pixel 1 of rows PRE_ITERS - 1 .. 0, then pixel 0 (its flags are brought to
        // the top by one shift): a lane's entries are still its 2 x 5 pixel block.
        {
            uint32_t sb = bb, sd = bd;
#pragma unroll
            for (int px = 1; px >= 0; --px) {
                if (px == 0) {
                    sb <<= 16 - PRE_ITERS;
                    sd <<= 16 - PRE_ITERS;
                }
#pragma unroll
                for (int it = PRE_ITERS - 1; it >= 0; --it) {
                    const unsigned short val = (unsigned short)(p00 + it * FT_PITCH + px);
                    const bool cb = __builtin_uadd_overflow(sb, sb, &sb);
                    if (cb) *wb = val;
                    wb += cb;
                    const bool cd = __builtin_uadd_overflow(sd, sd, &sd);
                    if (cd) *wd = val;
                    wd += cd;
                }
            }
        }
    }
    __syncthreads();
#if defined(AFV_FAST_STOP) && AFV_FAST_STOP == 2
    if (pre[tid] == 0xffff) cand_count[0] = pre_nb + pre_nd;
    return;
#endif
    // 2b. exact corner score, two list entries of one polarity per lane
    auto score_list = [&](auto dark_tag, int nlist) {
        constexpr bool DARK = decltype(dark_tag)::value;
        for (int i0 = 0; 2 * i0 < nlist; i0 += 256) {
            const int i = i0 + tid;
            if (2 * i < nlist) {
                const bool two = 2 * i + 1 < nlist;
                // entries 2i, 2i+1 of the bright list (from entry 0) or of the dark list (from entry PRE_DARK0): one aligned dword
                const uint32_t e2 = *reinterpret_cast<const uint32_t *>(&pre[(DARK ? PRE_DARK0 : 0) + 2 * i]);
                const int p0 = (int)(e2 & 0xffffu);
                const int p1 = two ? (int)(e2 >> 16) : p0;
                const uint8_t *c0 = &tile[p0], *c1 = &tile[p1];
                const short2v best = fast_score2<DARK>(c0, c1, ld_two(c0, c1, 0));
                const int s0 = best.x, s1 = best.y;
                // a pixel is a corner in at most one polarity (two 9-arcs of a 16-ring overlap): no write conflicts
                if (s0 > thr) sc[p0] = (uint8_t)(s0 - 1);
                if (s1 > thr) sc[p1] = (uint8_t)(s1 - 1);
            }
        }
    };
    score_list(std::false_type{}, pre_nb);
    score_list(std::true_type{}, pre_nd);
    __syncthreads();

#if defined(AFV_FAST_STOP) && AFV_FAST_STOP == 3
    if (sc[tid] == 255) cand_count[0] = 1;
    return;
#endif
    // 3. strict 3x3 maxima, dense over the score plane: thread (pair, rr) = (tid & 31, tid >> 5) owns the pixel pair at LDS columns
    //    c = 4 + 2 pair, c + 1 of the NR = FT_H / 8 CONSECUTIVE tile rows NR rr .. NR rr + NR - 1.  Per score row the packed pairs
    //    (s[c-1], s[c]), (s[c], s[c+1]), (s[c+1], s[c+2]) hold left neighbour / self / right neighbour of pixel 0 in the low and of
    //    pixel 1 in the high halves; the horizontal maxima of a row serve the pixel above and the pixel below, so each of the NR + 2
    //    rows is read (two aligned dwords, one LDS instruction) and reduced ONCE - 3 permutes + 2 packed max - and an output costs two
    //    more packed max (round 5; rows 8 apart, every row of every 3x3 window fetched by itself: 114 -> 86 vector, 40 -> 10 LDS
    //    instructions per thread).  keep <=> own score > every neighbour (a zero score never is).  Keep flags -> one compaction per wavefront.
    {
        const int c = 4 + 2 * (tid & 31), rr = tid >> 5;
        uint32_t kb = 0;
        constexpr int NR = FT_H / 8;
        const uint8_t *q0 = &sc[(FT_HALO + NR * rr) * FT_PITCH + c];
        short2v Hm[NR + 2], Sd[NR + 2], Sf[NR + 2];  // max of (left, self, right); max of (left, right); self
        // the four scores c-1 .. c+2 of a row lie in two aligned dwords (one LDS instruction); selectors depend on c & 2 only
        const int nsh = (c - 1) & 3;  // 3 or 1
        const uint32_t *qw = reinterpret_cast<const uint32_t *>(q0 - 1 - nsh);
        uint32_t nshm = (uint32_t)nsh * 0x00010001u;
        asm("" : "+v"(nshm));  // as in step 2a
        const uint32_t nsel_a = 0x0c010c00u + nshm;  // (c-1, c)
        const uint32_t nsel_b = 0x0c020c01u + nshm;  // (c,   c+1)
        const uint32_t nsel_c = 0x0c030c02u + nshm;  // (c+1, c+2)
#pragma unroll
        for (int k = 0; k < NR + 2; ++k) {
            const uint32_t w0 = qw[(k - 1) * (FT_PITCH / 4)], w1 = qw[(k - 1) * (FT_PITCH / 4) + 1];
            Sf[k] = as_s2(__builtin_amdgcn_perm(w1, w0, nsel_b));
            Sd[k] = pkmax(as_s2(__builtin_amdgcn_perm(w1, w0, nsel_a)), as_s2(__builtin_amdgcn_perm(w1, w0, nsel_c)));
            Hm[k] = pkmax(Sd[k], Sf[k]);
        }
        // top / bottom tiles of a level: scores of rows outside the FAST border (rows < 3 or >= h - 3; step 2a does not test rows) are
        // not scores - cleared here, in registers, before anything compares with them
        if (gy0 < 0 || gy0 + FT_HALO + FT_H >= lh - 3) {
#pragma unroll
            for (int k = 0; k < NR + 2; ++k) {
                const int gy = gy0 + FT_HALO + NR * rr + k - 1;
                if (gy < 3 || gy >= lh - 3) {
                    Hm[k] = as_s2(0u);
                    Sd[k] = as_s2(0u);
                    Sf[k] = as_s2(0u);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const uint32_t x = as_u32(pkmax(pkmax(Hm[it], Hm[it + 2]), Sd[it + 1]) - Sf[it + 1]);  // sign bit set <=> self > every neighbour
            kb = pk_shr1(kb) | (x & 0x80008000u);
        }
        kb &= ((0xffffu << (16 - FT_H / 8)) & 0xffffu) * 0x00010001u;
        const int cnt = __popc(kb);
        const int incl = wave_incl_scan(cnt);
        int base = 0;
        if (lane == 63) base = lds_add_rtn_one_lane(&list_n, incl);
        // `list` aliases `pre`: every wavefront must be done with the score lists before anyone writes (barrier above)
        uint32_t *wl = list + (__builtin_amdgcn_readlane(base, 63) + incl - cnt);
        while (kb) {
            const int qb = __builtin_ctz(kb);
            kb &= kb - 1;
            const int px = c - FT_HALO + (qb >> 4), py = NR * rr + ((qb & 15) - (16 - NR));
            *wl++ = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)sc[(py + FT_HALO) * FT_PITCH + px + FT_HALO] << 16);
        }
    }
    __syncthreads();
#if defined(AFV_FAST_STOP) && AFV_FAST_STOP == 4
    if (list[tid] == 0xffffffffu) cand_count[0] = list_n;
    return;
#endif
    const int n = list_n;
    if (n == 0) return;
    if (tid == 0) out_base = atomicAdd(&cand_count[f * AFV_MAX_LEVELS + l], n);
    __syncthreads();

    // 4. one global atomic per tile reserved the output slots; the survivors leave as (x | y << 12 | score << 24) in level coordinates.
    //    Their Harris responses are computed later (k_harris.hip) and only for the ones that survive retainBest on the score.
    const size_t obase = L.cand_off + (size_t)f * L.cand_frame_stride + (size_t)out_base;
    for (int ci = tid; ci < n; ci += 256) {
        const uint32_t e = list[ci];
        const int gx = gx0 + FT_HALO + (int)(e & 255u), gy = gy0 + FT_HALO + (int)((e >> 8) & 255u);
        cand_packed[obase + ci] = (uint32_t)gx | ((uint32_t)gy << 12) | ((e >> 16) << 24);
    }
}

// `geo` is the DEVICE copy of the geometry
extern "C" void afv_launch_fast_nms(const Geo *geo, int total_tiles, const FrameSrc *src0, const uint8_t *pyr, uint32_t *cand_packed,
                                    int *cand_count, int frame_base, int nframes, hipStream_t stream) {
    const int total = total_tiles * nframes;
    dim3 grid((total + 7) / 8 * 8);
    hipLaunchKernelGGL(k_fast_nms, grid, dim3(256), 0, stream, geo, *src0, pyr, cand_packed, cand_count, total, frame_base);
}
