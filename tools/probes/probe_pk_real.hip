// probe_pk_real.hip - round 6 reproducer: k_describe's packed-fp32 rotation beside the REAL column-sliced MFMA matcher of another context.
// Two host threads call afv_match_bow (1000 x 1000 descriptors, the per-frame plugin shape: k_match_topk_mfma<true>) on their own contexts;
// the main thread launches a victim kernel that evaluates the rBRIEF rotation of random points in the packed and in the plain form and
// counts, per lane quarter, where the two disagree.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/probes/probe_pk_real.hip -Iinclude -Lanyfeature-vslam_amd -lafv_hip \
//         -Wl,-rpath,$PWD/anyfeature-vslam_amd -lpthread -o tools/probes/probe_pk_real
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "afv_hip.h"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: the packed form of k_describe.hip; 1: packed, operands held as natural pairs (no op_sel broadcast of an odd register);
// 2: plain fp32 against plain fp32 (control)
template <int MODE>
__global__ __launch_bounds__(256) void k_victim(const float4 *__restrict__ table, int iters, unsigned seed, unsigned *__restrict__ bad /* [4 groups][4 quarters] */) {
    __shared__ uint8_t s_pad[13760];  // k_describe's LDS footprint (same residency beside the matcher)
    const int lane = threadIdx.x & 63;
    if (seed == 0xffffffffu) s_pad[threadIdx.x] = 1;
    unsigned st = seed ^ (blockIdx.x * 2654435761u) ^ ((threadIdx.x >> 6) * 40503u);
    unsigned b[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        st = st * 1664525u + 1013904223u;  // wave-uniform
        const float ca = (float)((int)(st >> 8 & 2047) - 1024) * (1.0f / 1024.0f), sb = (float)((int)(st >> 20 & 2047) - 1024) * (1.0f / 1024.0f);
        const f32x2 cs_a = {ca, sb}, cs_b = {-sb, ca};
        const int kox = -0x4B400000;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 pt = table[((it & 15) * 4 + g) * 64 + lane];
            const float x0 = pt.x, y0 = pt.y, x1 = pt.z, y1 = pt.w;
            int ix0, iy0, ix1, iy1;
            const f32x2 rmag = {12582912.0f, 12582912.0f};
            if (MODE == 0) {
                const f32x2 rp0 = f32x2{x0, x0} * cs_a + f32x2{y0, y0} * cs_b, rp1 = f32x2{x1, x1} * cs_a + f32x2{y1, y1} * cs_b;
                const f32x2 rq0 = rp0 + rmag, rq1 = rp1 + rmag;
                ix0 = __float_as_int(rq0.x) + kox; iy0 = __float_as_int(rq0.y) + kox; ix1 = __float_as_int(rq1.x) + kox; iy1 = __float_as_int(rq1.y) + kox;
            } else if (MODE == 1) {
                // over the two POINTS: X = (x0, x1) ca - (y0, y1) sb, Y = (x0, x1) sb + (y0, y1) ca
                const f32x2 xs = {x0, x1}, ys = {y0, y1};
                const f32x2 X = xs * f32x2{ca, ca} + ys * f32x2{-sb, -sb} + rmag, Y = xs * f32x2{sb, sb} + ys * f32x2{ca, ca} + rmag;
                ix0 = __float_as_int(X.x) + kox; ix1 = __float_as_int(X.y) + kox; iy0 = __float_as_int(Y.x) + kox; iy1 = __float_as_int(Y.y) + kox;
            } else if (MODE >= 16) {
                // packed 16-bit forms the kernels rely on (k_fast_nms: v_pk_minimum3_f16 / v_pk_maximum3_f16 on u16 values 0 .. 255, v_pk_min_i16,
                // v_pk_sub_i16; k_harris: v_pk_mul_lo_u16 / v_pk_add_u16): result against the scalar evaluation of the same selection
                const unsigned a = (unsigned)(int)x0 & 0xffu, bb = (unsigned)(int)y0 & 0xffu, c = (unsigned)(int)x1 & 0xffu, d = (unsigned)(int)y1 & 0xffu;
                const unsigned pa = a | (bb << 16), pb = c | (d << 16), pc = d | (a << 16);
                unsigned o, e;
                auto mn = [](unsigned u, unsigned v) { return u < v ? u : v; };
                auto mx = [](unsigned u, unsigned v) { return u > v ? u : v; };
                if (MODE == 16) { asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(pa), "v"(pb), "v"(pc)); e = mn(mn(a, c), d) | (mn(mn(bb, d), a) << 16); }
                else if (MODE == 17) { asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(pa), "v"(pb), "v"(pc)); e = mx(mx(a, c), d) | (mx(mx(bb, d), a) << 16); }
                else if (MODE == 18) { asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(o) : "v"(pa), "v"(pb)); e = mn(a, c) | (mn(bb, d) << 16); }
                else if (MODE == 19) { asm volatile("v_pk_sub_i16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(o) : "v"(pa), "v"(pb)); e = ((bb - c) & 0xffffu) | (((a - d) & 0xffffu) << 16); }
                else { asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(o) : "v"(pa), "v"(pb), "v"(pc)); e = ((a * c + d) & 0xffffu) | (((bb * d + a) & 0xffffu) << 16); }
                asm volatile("" : "+v"(e));
                b[g] += o != e ? 1u : 0u;
                continue;
            } else if (MODE >= 9) {
                // single instructions: which forms are hit?  result (o0, o1) against what the same selection gives with v_mov / plain ops
                float o0, o1, e0, e1;
#define ONE(TXT) asm volatile("v_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\tv_mov_b32 v14, %4\n\tv_mov_b32 v15, %5\n\ts_nop 4\n\t" TXT "\n\ts_nop 4\n\tv_mov_b32 %0, v20\n\tv_mov_b32 %1, v21" \
                              : "=v"(o0), "=v"(o1) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "v12", "v13", "v14", "v15", "v20", "v21")
                if (MODE == 9) { ONE("v_pk_mov_b32 v[20:21], v[12:13], v[12:13] op_sel:[0,1]"); e0 = x0; e1 = y0; }            // the compiler's 64-bit copy
                else if (MODE == 10) { ONE("v_pk_mov_b32 v[20:21], v[12:13], v[14:15] op_sel:[1,0]"); e0 = y0; e1 = x1; }      // a shuffle
                else if (MODE == 11) { ONE("v_pk_mul_f32 v[20:21], v[12:13], v[14:15] op_sel_hi:[1,0]"); e0 = x0 * x1; e1 = y0 * x1; }   // second operand: low register twice
                else if (MODE == 12) { ONE("v_pk_mul_f32 v[20:21], v[12:13], v[14:15] op_sel:[0,1]"); e0 = x0 * y1; e1 = y0 * y1; }      // second operand: high register twice
                else if (MODE == 13) { ONE("v_pk_mul_f32 v[20:21], v[12:13], v[14:15] op_sel_hi:[0,1]"); e0 = x0 * x1; e1 = x0 * y1; }   // FIRST operand: low register twice
                else if (MODE == 14) { ONE("v_pk_mul_f32 v[20:21], v[12:13], v[14:15]"); e0 = x0 * x1; e1 = y0 * y1; }                   // no selection
                else { ONE("v_pk_add_f32 v[20:21], v[12:13], v[14:15] op_sel_hi:[1,0]"); e0 = x0 + x1; e1 = y0 + x1; }                   // 15: the add
                asm volatile("" : "+v"(e0), "+v"(e1));
                b[g] += (__float_as_int(o0) != __float_as_int(e0) || __float_as_int(o1) != __float_as_int(e1)) ? 1u : 0u;
                continue;
            } else if (MODE >= 3) {
                // the compiler's own block for MODE 0, as text, with variations (see main)
                float o0, o1, o2, o3;
#define BLK(I1, I2, I3, I4, I5)                                                                                                   \
    asm volatile("v_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, %6\n\tv_mov_b32 v15, %7\n\t"                          \
                 "v_mov_b32 v4, %8\n\tv_mov_b32 v5, %9\n\tv_mov_b32 v6, %10\n\tv_mov_b32 v7, %8\n\ts_nop 4\n\t" I1 I2 I3 I4 I5    \
                 "v_pk_add_f32 v[12:13], v[16:17], v[12:13]\n\tv_pk_add_f32 v[14:15], v[18:19], v[14:15]\n\t"                      \
                 "v_pk_add_f32 v[12:13], v[12:13], %11 op_sel_hi:[1,0]\n\tv_pk_add_f32 v[16:17], v[14:15], %11 op_sel_hi:[1,0]\n\t" \
                 "s_nop 4\n\tv_mov_b32 %0, v12\n\tv_mov_b32 %1, v13\n\tv_mov_b32 %2, v16\n\tv_mov_b32 %3, v17"                    \
                 : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3)                                                                         \
                 : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(ca), "v"(sb), "v"(-sb), "s"(rmag)                                      \
                 : "v4", "v5", "v6", "v7", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21")
#define M1 "v_pk_mul_f32 v[18:19], v[4:5], v[14:15] op_sel_hi:[1,0]\n\t"
#define MV "v_mov_b32 v14, v15\n\t"
#define M2 "v_pk_mul_f32 v[16:17], v[4:5], v[12:13] op_sel_hi:[1,0]\n\t"
#define M3 "v_pk_mul_f32 v[12:13], v[6:7], v[12:13] op_sel:[0,1]\n\t"
#define M4 "v_pk_mul_f32 v[14:15], v[6:7], v[14:15] op_sel_hi:[1,0]\n\t"
#define NP "s_nop 1\n\t"
                if (MODE == 3) BLK(M1, MV, M2, M3, M4);                                   // as compiled
                else if (MODE == 4) BLK(M1 NP, MV NP, M2 NP, M3 NP, M4 NP);               // two wait states behind every instruction
                else if (MODE == 5) BLK(M1 NP, MV, M2, M3, M4);                           // ... only between the first multiply and the v_mov over its source
                else if (MODE == 6) BLK(M1, MV, M2 NP, M3, M4);                           // ... only between the multiply that reads v12 and the one that writes it
                else if (MODE == 7) BLK(M1, MV, M2, "v_pk_mul_f32 v[20:21], v[6:7], v[12:13] op_sel:[0,1]\n\tv_mov_b32 v12, v20\n\tv_mov_b32 v13, v21\n\t", M4);  // the op_sel:[0,1] multiply not in place
                else BLK(M1, MV, M2, M3 NP, M4);                                          // 8: wait states behind the in-place op_sel:[0,1] multiply
                // v7 must hold ca: fixed below (the block above loads v7 from operand 8 = ca)
                ix0 = __float_as_int(o0) + kox; iy0 = __float_as_int(o1) + kox; ix1 = __float_as_int(o2) + kox; iy1 = __float_as_int(o3) + kox;
            } else {
                float a0 = x0 * ca, a1 = y0 * -sb, a2 = x0 * sb, a3 = y0 * ca, c0 = x1 * ca, c1 = y1 * -sb, c2 = x1 * sb, c3 = y1 * ca;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
                ix0 = __float_as_int(a0 + a1 + 12582912.0f) + kox; iy0 = __float_as_int(a2 + a3 + 12582912.0f) + kox;
                ix1 = __float_as_int(c0 + c1 + 12582912.0f) + kox; iy1 = __float_as_int(c2 + c3 + 12582912.0f) + kox;
            }
            // reference: plain fp32, every product and sum behind a barrier the vectoriser cannot see through
            float p0 = x0 * ca, p1 = y0 * -sb, p2 = x0 * sb, p3 = y0 * ca, q0 = x1 * ca, q1 = y1 * -sb, q2 = x1 * sb, q3 = y1 * ca;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            float s0 = p0 + p1, s1 = p2 + p3, s2 = q0 + q1, s3 = q2 + q3;
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
            const int jx0 = __float_as_int(s0 + 12582912.0f) + kox, jy0 = __float_as_int(s1 + 12582912.0f) + kox;
            const int jx1 = __float_as_int(s2 + 12582912.0f) + kox, jy1 = __float_as_int(s3 + 12582912.0f) + kox;
            b[g] += (ix0 != jx0 || iy0 != jy0 || ix1 != jx1 || iy1 != jy1) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (b[g]) atomicAdd(&bad[g * 4 + (lane >> 4)], b[g]);
}

static std::atomic<bool> g_stop{false};
static std::atomic<long> g_calls{0};

static void matcher_thread(int seed) {
    afv_orb_params p;
    afv_default_orb_params(&p);
    afv_ctx *c = nullptr;
    if (afv_create(0, &p, &c) != 0) { printf("afv_create failed\n"); return; }
    const int n = 1000;
    std::vector<uint8_t> d1((size_t)n * 32), d2((size_t)n * 32);
    unsigned st = 12345u + seed;
    for (auto &v : d1) { st = st * 1664525u + 1013904223u; v = (uint8_t)(st >> 24); }
    d2 = d1;
    for (size_t i = 0; i < d2.size(); i += 7) d2[i] ^= 0x10;
    std::vector<int32_t> out(n), nm(1);
    afv_match_job j;
    std::memset(&j, 0, sizeof j);
    j.desc1 = d1.data(); j.n1 = n; j.desc2 = d2.data(); j.n2 = n; j.desc_bytes = 32;
    j.th_low = 75.f; j.nnratio = 0.7f; j.check_orientation = 0; j.mode = AFV_MATCH_KF_KF;
    while (!g_stop.load()) {
        const int rc = afv_match_bow(c, &j, 1, out.data(), nm.data());
        if (rc) { printf("afv_match_bow: %d %s\n", rc, afv_last_error(c)); break; }
        g_calls++;
    }
    afv_destroy(c);
}

template <int MODE>
static void run(const char *name, const float4 *d_table, unsigned *d_bad, hipStream_t s, double seconds, bool with_matchers) {
    CHK(hipMemset(d_bad, 0, 64));
    CHK(hipDeviceSynchronize());
    g_stop = false;
    g_calls = 0;
    std::vector<std::thread> th;
    if (with_matchers)
        for (int i = 0; i < 2; ++i) th.emplace_back(matcher_thread, i);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    unsigned seed = 1;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(k_victim<MODE>, dim3(256), dim3(256), 0, s, d_table, 64, seed++, d_bad);  // one frame's worth of keypoints x 64
        CHK(hipStreamSynchronize(s));
        ++launches;
    }
    g_stop = true;
    for (auto &t : th) t.join();
    unsigned b[16];
    CHK(hipMemcpy(b, d_bad, 64, hipMemcpyDeviceToHost));
    printf("%-44s matchers beside it: %s (%ld match calls, %ld victim launches) | disagreements [group][lane quarter]:", name, with_matchers ? "yes" : "no ", g_calls.load(), launches);
    for (int g = 0; g < 4; ++g) printf("  %u %u %u %u", b[g * 4], b[g * 4 + 1], b[g * 4 + 2], b[g * 4 + 3]);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    std::vector<float> h(16 * 4 * 64 * 4);
    unsigned st = 99;
    for (auto &v : h) { st = st * 1664525u + 1013904223u; v = (float)((int)(st >> 16) % 14); }
    float4 *d_table;
    unsigned *d_bad;
    CHK(hipMalloc(&d_table, h.size() * 4));
    CHK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_bad, 64));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    run<0>("packed fp32 as in k_describe (op_sel)", d_table, d_bad, s, 2.0, false);
    run<0>("packed fp32 as in k_describe (op_sel)", d_table, d_bad, s, seconds, true);
    run<1>("packed fp32 over natural pairs", d_table, d_bad, s, seconds, true);
    run<2>("plain fp32", d_table, d_bad, s, seconds, true);
    if (argc > 2) {  // single-instruction forms only
        run<9>("v_pk_mov_b32 op_sel:[0,1] (64-bit copy)", d_table, d_bad, s, seconds, true);
        run<10>("v_pk_mov_b32 op_sel:[1,0] (shuffle)", d_table, d_bad, s, seconds, true);
        run<11>("v_pk_mul_f32 op_sel_hi:[1,0]", d_table, d_bad, s, seconds, true);
        run<12>("v_pk_mul_f32 op_sel:[0,1]", d_table, d_bad, s, seconds, true);
        run<13>("v_pk_mul_f32 op_sel_hi:[0,1]", d_table, d_bad, s, seconds, true);
        run<14>("v_pk_mul_f32 (no selection)", d_table, d_bad, s, seconds, true);
        run<15>("v_pk_add_f32 op_sel_hi:[1,0]", d_table, d_bad, s, seconds, true);
        run<16>("v_pk_minimum3_f16 (u16 values 0..255)", d_table, d_bad, s, seconds, true);
        run<17>("v_pk_maximum3_f16", d_table, d_bad, s, seconds, true);
        run<18>("v_pk_min_i16", d_table, d_bad, s, seconds, true);
        run<19>("v_pk_sub_i16 op_sel:[1,0] op_sel_hi:[0,1]", d_table, d_bad, s, seconds, true);
        run<20>("v_pk_mad_u16", d_table, d_bad, s, seconds, true);
        run<16>("v_pk_minimum3_f16 (u16 values 0..255)", d_table, d_bad, s, 2.0, false);
        run<19>("v_pk_sub_i16 op_sel:[1,0] op_sel_hi:[0,1]", d_table, d_bad, s, 2.0, false);
        run<9>("v_pk_mov_b32 op_sel:[0,1] (64-bit copy)", d_table, d_bad, s, 2.0, false);
        run<12>("v_pk_mul_f32 op_sel:[0,1]", d_table, d_bad, s, 2.0, false);
        return 0;
    }
    run<3>("the compiler's block as text", d_table, d_bad, s, seconds, true);
    run<4>("... s_nop 1 behind every multiply / v_mov", d_table, d_bad, s, seconds, true);
    run<5>("... s_nop 1 only before the v_mov over v14", d_table, d_bad, s, seconds, true);
    run<6>("... s_nop 1 only before the write of v12", d_table, d_bad, s, seconds, true);
    run<7>("... op_sel:[0,1] multiply not in place", d_table, d_bad, s, seconds, true);
    run<8>("... s_nop 1 behind the in-place multiply", d_table, d_bad, s, seconds, true);
    return 0;
}
