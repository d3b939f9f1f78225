// host_latency.cpp - what the reference's call costs end to end through the C-ABI, measured from C++ (no Python in the loop):
// afv_orb_extract (FeatureExtractor::operator(), one host image in, host vectors out) and afv_match_bow (brute-force SearchByBoW of two
// frames through host buffers), each called back to back on one context.  Build: g++ -O2 -std=c++17 tools/host_latency.cpp -I include
// -L anyfeature-vslam_amd -lafv_hip -Wl,-rpath,$PWD/anyfeature-vslam_amd -Wl,-rpath,/opt/rocm/lib -L /opt/rocm/lib -lamdhip64 -o tools/host_latency
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "afv_hip.h"

static uint32_t lcg(uint32_t &x) { return x = x * 1664525u + 1013904223u; }

// the "corners" frame of anyfeature-vslam_amd/synth.py (8 x 8 LCG tiles, 3 x 3 box blur, +-4 noise)
static void corners_frame(uint32_t seed, int w, int h, std::vector<uint8_t> &out) {
    const int bw = (w + 7) / 8, bh = (h + 7) / 8;
    uint32_t x = seed;
    std::vector<int> tiles((size_t)bw * bh), img((size_t)w * h);
    for (auto &t : tiles) t = (int)((lcg(x) >> 8) & 255u);
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < w; ++c) img[(size_t)y * w + c] = tiles[(size_t)(y / 8) * bw + c / 8];
    out.resize((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int c = 0; c < w; ++c) {
            int s = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = std::min(std::max(y + dy, 0), h - 1), xx = std::min(std::max(c + dx, 0), w - 1);
                    s += img[(size_t)yy * w + xx];
                }
            const int noise = (int)((lcg(x) >> 8) % 9u) - 4;
            out[(size_t)y * w + c] = (uint8_t)std::min(std::max((s + 4) / 9 + noise, 0), 255);
        }
}

template <class F>
static double time_us(int reps, F &&f) {
    for (int i = 0; i < 10; ++i) f();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) f();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main(int argc, char **argv) {
    const int W = 640, H = 480, reps = argc > 1 ? atoi(argv[1]) : 300;
    afv_orb_params p;
    afv_default_orb_params(&p);
    afv_ctx *ctx = nullptr;
    if (afv_create(0, &p, &ctx)) {
        fprintf(stderr, "afv_create failed\n");
        return 1;
    }
    std::vector<uint8_t> f1, f2;
    corners_frame(7001, W, H, f1);
    corners_frame(7002, W, H, f2);
    const int cap = afv_max_keypoints_per_frame(ctx);
    std::vector<afv_keypoint> k1(cap), k2(cap);
    std::vector<uint8_t> d1((size_t)cap * 32), d2((size_t)cap * 32);
    int n1 = 0, n2 = 0;
    afv_orb_extract(ctx, f1.data(), W, H, W, k1.data(), d1.data(), cap, &n1);
    const double t_pageable = time_us(reps, [&] { afv_orb_extract(ctx, f2.data(), W, H, W, k2.data(), d2.data(), cap, &n2); });
    uint8_t *pin = nullptr;
    double t_pinned = -1;
    if (hipHostMalloc(reinterpret_cast<void **>(&pin), (size_t)W * H, hipHostMallocDefault) == hipSuccess) {
        std::memcpy(pin, f2.data(), (size_t)W * H);
        t_pinned = time_us(reps, [&] { afv_orb_extract(ctx, pin, W, H, W, k2.data(), d2.data(), cap, &n2); });
        (void)hipHostFree(pin);
    }
    // brute-force SearchByBoW(KF, KF) of the two frames through host buffers (Tracking's matcher calls, FeatureMatcher.cc:561-660)
    std::vector<float> a1(n1), a2(n2);
    for (int i = 0; i < n1; ++i) a1[i] = k1[i].angle;
    for (int i = 0; i < n2; ++i) a2[i] = k2[i].angle;
    afv_match_job j;
    std::memset(&j, 0, sizeof(j));
    j.desc1 = d2.data();
    j.n1 = n2;
    j.desc2 = d1.data();
    j.n2 = n1;
    j.desc_bytes = 32;
    j.angle1 = a2.data();
    j.angle2 = a1.data();
    j.th_low = 75.f;
    j.nnratio = 0.6f;
    j.check_orientation = 1;
    j.mode = AFV_MATCH_KF_KF;
    std::vector<int32_t> out(cap);
    int32_t nm = 0;
    const double t_match = time_us(reps, [&] { afv_match_bow(ctx, &j, 1, out.data(), &nm); });
    printf("{\"afv_orb_extract_us\": %.1f, \"afv_orb_extract_pinned_input_us\": %.1f, \"afv_match_bow_us\": %.1f, \"keypoints\": %d, \"matches\": %d, \"reps\": %d}\n",
           t_pageable, t_pinned, t_match, n2, nm, reps);
    afv_destroy(ctx);
    return 0;
}
