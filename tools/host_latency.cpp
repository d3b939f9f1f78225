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
    // ---- the tracking step of one frame through a device-resident Frame (round 5): Frame::Frame (extract + grid) -> ComputeBoW ->
    // SearchByProjection(cur, last) (1000 queries, Tracking.cc:747) -> SearchByProjection(F, local map) (2000 queries, :1026) -> promotion to
    // a keyframe (KeyFrame.cc:36), each stage a synchronous C call with host results, against the same chain through the host-array
    // entry points (what round 4 offered: every call re-uploads the frame and rebuilds its grid) ----
    double t_chain = -1, t_chain_ref = -1, t_fextract = -1, t_fbow = -1, t_fproj1 = -1, t_fproj1_ref = -1, t_fproj2 = -1, t_fpromote = -1, t_old_chain = -1,
           t_old_bow = -1, t_old_proj1 = -1, t_old_proj2 = -1, t_init_old = -1, t_init_frame = -1, t_fbowmatch = -1, t_ffuse = -1;
    int nm_p1 = 0, nm_p2 = 0, nm_init = 0, nm_bf = 0, vocab_nodes = 0, nm_fuse = 0;
    {
        // a vocabulary of the shipped shape (k = 10, L = 6: createVocabulary.py:39-42), LCG descriptors
        const int K = 10, L = 6;
        std::vector<int32_t> child_ptr, child_idx;
        int nnodes = 1;
        {
            long level = 1;
            for (int l = 1; l <= L; ++l) {
                level *= K;
                nnodes += (int)level;
            }
            child_ptr.assign((size_t)nnodes + 1, 0);
            child_idx.resize((size_t)nnodes - 1);
            const int inner = nnodes - (int)level;  // nodes of levels 0 .. L-1 have K children each, ids in breadth-first order
            for (int i = 0; i < nnodes; ++i) child_ptr[i + 1] = child_ptr[i] + (i < inner ? K : 0);
            for (int i = 0; i < nnodes - 1; ++i) child_idx[i] = i + 1;
        }
        vocab_nodes = nnodes;
        std::vector<uint8_t> vdesc((size_t)nnodes * 32);
        uint32_t x = 99;
        for (auto &b : vdesc) b = (uint8_t)((lcg(x) >> 8) & 255u);
        afv_vocab *voc = nullptr;
        afv_table *table = nullptr;
        afv_frame *cur = nullptr, *lastf = nullptr;
        afv_frame_params fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.struct_size = sizeof(fp);
        fp.min_x = 0; fp.min_y = 0; fp.max_x = (float)W; fp.max_y = (float)H;
        fp.grid_cols = 64; fp.grid_rows = 48;
        int rc = afv_vocab_create(ctx, K, L, nnodes, child_ptr.data(), child_idx.data(), vdesc.data(), 32, &voc);
        if (!rc) rc = afv_table_create(ctx, 8, cap, &table);
        if (!rc) rc = afv_frame_create(ctx, &fp, &cur);
        if (!rc) rc = afv_frame_create(ctx, &fp, &lastf);
        if (rc) {
            fprintf(stderr, "tracking chain set-up failed: %d %s\n", rc, afv_last_error(ctx));
        } else {
            std::vector<uint8_t> f3;
            corners_frame(7003, W, H, f3);
            // the current frame of the chain: the last frame moved by 3 px (consecutive video frames: the projection searches find their matches)
            std::vector<uint8_t> f2s((size_t)W * H);
            for (int y = 0; y < H; ++y)
                for (int c = 0; c < W; ++c) f2s[(size_t)y * W + c] = f1[(size_t)y * W + (c + W - 3) % W];
            std::vector<afv_keypoint> k3(cap);
            std::vector<uint8_t> d3((size_t)cap * 32);
            int n3 = 0;
            afv_orb_extract(ctx, f3.data(), W, H, W, k3.data(), d3.data(), cap, &n3);
            // last frame = f1 (k1, d1), resident and promoted into slot 0; a second keyframe (f3) in slot 1
            afv_frame_extract(lastf, f1.data(), W, H, W, nullptr, nullptr, 0, nullptr);
            afv_frame_bow_transform(lastf, voc, 4, nullptr, nullptr, nullptr);
            afv_table_set_from_frame(table, 0, lastf);
            afv_frame_extract(cur, f3.data(), W, H, W, nullptr, nullptr, 0, nullptr);
            afv_frame_bow_transform(cur, voc, 4, nullptr, nullptr, nullptr);
            afv_table_set_from_frame(table, 1, cur);
            // queries: the last frame's features "projected" with the identity motion (1000), then those plus the other keyframe's (2000)
            std::vector<float> sz1(n1), sg1(n1), in1(n1), sz3(n3), sg3(n3), in3(n3);
            afv_orb_size_sigma(ctx, k1.data(), n1, sz1.data(), sg1.data(), in1.data());
            afv_orb_size_sigma(ctx, k3.data(), n3, sz3.data(), sg3.data(), in3.data());
            const int nq1 = n1, nq2 = n1 + n3;
            std::vector<float> qu(nq2), qv(nq2), qr(nq2), qmn(nq2), qmx(nq2), qang(nq2);
            std::vector<uint8_t> qd((size_t)nq2 * 32);
            std::vector<int32_t> qslot(nq2), qidx(nq2);
            for (int i = 0; i < nq2; ++i) {
                const bool a = i < n1;
                const afv_keypoint &k = a ? k1[i] : k3[i - n1];
                const float s = a ? sz1[i] : sz3[i - n1];
                qu[i] = k.x; qv[i] = k.y; qr[i] = 15.0f * s; qmn[i] = s / 1.2f; qmx[i] = s * 1.2f; qang[i] = k.angle;
                std::memcpy(&qd[(size_t)i * 32], a ? &d1[(size_t)i * 32] : &d3[(size_t)(i - n1) * 32], 32);
                qslot[i] = a ? 0 : 1;
                qidx[i] = a ? i : i - n1;
            }
            afv_proj_queries Q1, Q2;
            std::memset(&Q1, 0, sizeof(Q1));
            Q1.struct_size = sizeof(Q1);
            Q1.nq = nq1; Q1.qdesc = qd.data(); Q1.desc_bytes = 32;
            Q1.qu = qu.data(); Q1.qv = qv.data(); Q1.qr = qr.data(); Q1.qmin_size = qmn.data(); Q1.qmax_size = qmx.data(); Q1.qangle = qang.data();
            Q1.th_high = 75.f; Q1.nnratio = 0.9f; Q1.check_orientation = 1; Q1.mode = AFV_PROJ_LASTFRAME;
            Q2 = Q1;
            Q2.nq = nq2; Q2.nnratio = 0.8f; Q2.check_orientation = 0; Q2.mode = AFV_PROJ_LOCALMAP;
            afv_proj_queries Q1r = Q1, Q2r = Q2;  // the same queries with their descriptors named as rows of the keyframe table
            Q1r.qdesc = nullptr; Q1r.qref_table = table; Q1r.qref_slot = qslot.data(); Q1r.qref_idx = qidx.data();
            Q2r.qdesc = nullptr; Q2r.qref_table = table; Q2r.qref_slot = qslot.data(); Q2r.qref_idx = qidx.data();
            std::vector<int32_t> assign(cap), leaf(cap), nid(cap);
            int32_t nm1 = 0, nm2 = 0, nnodes_fv = 0, nmb = 0;
            int slot = 2;
            auto chain = [&](bool by_ref) {
                afv_frame_extract(cur, f2s.data(), W, H, W, k2.data(), d2.data(), cap, &n2);
                afv_frame_bow_transform(cur, voc, 4, leaf.data(), nid.data(), &nnodes_fv);
                afv_frame_match_projection(cur, by_ref ? &Q1r : &Q1, assign.data(), &nm1);
                afv_frame_match_projection(cur, by_ref ? &Q2r : &Q2, assign.data(), &nm2);
                afv_table_set_from_frame(table, slot, cur);
                slot = 2 + (slot - 1) % 6;
            };
            t_chain = time_us(reps, [&] { chain(false); });
            t_chain_ref = time_us(reps, [&] { chain(true); });
            nm_p1 = nm1; nm_p2 = nm2;
            t_fextract = time_us(reps, [&] { afv_frame_extract(cur, f2s.data(), W, H, W, k2.data(), d2.data(), cap, &n2); });
            t_fbow = time_us(reps, [&] { afv_frame_bow_transform(cur, voc, 4, leaf.data(), nid.data(), &nnodes_fv); });
            t_fproj1 = time_us(reps, [&] { afv_frame_match_projection(cur, &Q1, assign.data(), &nm1); });
            t_fproj1_ref = time_us(reps, [&] { afv_frame_match_projection(cur, &Q1r, assign.data(), &nm1); });
            t_fproj2 = time_us(reps, [&] { afv_frame_match_projection(cur, &Q2, assign.data(), &nm2); });
            {   // matching core of Fuse(pKF, vpMapPoints) (FeatureMatcher.cc:794-940) with the frame in the keyframe's role: 2000 map points
                std::vector<int32_t> best(nq2);
                int32_t nf = 0;
                afv_proj_queries QF = Q2;
                QF.th_high = 75.f;
                t_ffuse = time_us(reps, [&] { afv_frame_match_fuse(cur, &QF, 1, best.data(), &nf); });
                nm_fuse = nf;
            }
            t_fpromote = time_us(reps, [&] {
                afv_table_set_from_frame(table, 2, cur);
                (void)hipStreamSynchronize((hipStream_t)afv_stream(ctx));
            });
            {   // SearchByBoW(KF, F) of TrackReferenceKeyFrame (Tracking.cc:626-629) with the frame side resident
                const int32_t sl = 0;
                std::vector<int32_t> mf(cap);
                t_fbowmatch = time_us(reps, [&] { afv_table_match_bow_frame_h(table, &sl, 1, cur, 75.f, 0.7f, 1, mf.data(), &nmb); });
                nm_bf = nmb;
            }
            // the same chain through the host-array entry points
            std::vector<float> x2(cap), y2(cap), s2(cap), sg2(cap), i2(cap), a2f(cap);
            afv_proj_job J1, J2;
            auto fill = [&](afv_proj_job &J, const afv_proj_queries &Q) {
                std::memset(&J, 0, sizeof(J));
                J.struct_size = sizeof(J);
                J.desc = d2.data(); J.n = n2; J.desc_bytes = 32;
                J.x = x2.data(); J.y = y2.data(); J.size = s2.data(); J.angle = a2f.data();
                J.min_x = 0; J.min_y = 0; J.grid_inv_w = 64.0f / (float)W; J.grid_inv_h = 48.0f / (float)H; J.grid_cols = 64; J.grid_rows = 48;
                J.nq = Q.nq; J.qdesc = Q.qdesc; J.qu = Q.qu; J.qv = Q.qv; J.qr = Q.qr; J.qmin_size = Q.qmin_size; J.qmax_size = Q.qmax_size;
                J.qangle = Q.qangle; J.th_high = Q.th_high; J.nnratio = Q.nnratio; J.size_tol = 1.2f; J.inv_size_tol = 1.0f / 1.2f;
                J.check_orientation = Q.check_orientation; J.mode = Q.mode;
            };
            auto old_prep = [&] {
                afv_orb_size_sigma(ctx, k2.data(), n2, s2.data(), sg2.data(), i2.data());
                for (int i = 0; i < n2; ++i) {
                    x2[i] = k2[i].x; y2[i] = k2[i].y; a2f[i] = k2[i].angle;
                }
                fill(J1, Q1);
                fill(J2, Q2);
            };
            auto old_chain = [&] {
                afv_orb_extract(ctx, f2s.data(), W, H, W, k2.data(), d2.data(), cap, &n2);
                afv_bow_transform(ctx, voc, d2.data(), n2, 4, leaf.data(), nid.data());
                old_prep();
                afv_match_projection(ctx, &J1, 1, assign.data(), &nm1);
                afv_match_projection(ctx, &J2, 1, assign.data(), &nm2);
                afv_table_set(table, 2, d2.data(), a2f.data(), n2);
            };
            t_old_chain = time_us(reps, old_chain);
            if (nm1 != nm_p1 || nm2 != nm_p2) fprintf(stderr, "WARNING: the two chains disagree: %d/%d vs %d/%d\n", nm1, nm2, nm_p1, nm_p2);
            t_old_bow = time_us(reps, [&] { afv_bow_transform(ctx, voc, d2.data(), n2, 4, leaf.data(), nid.data()); });
            t_old_proj1 = time_us(reps, [&] { afv_match_projection(ctx, &J1, 1, assign.data(), &nm1); });
            t_old_proj2 = time_us(reps, [&] { afv_match_projection(ctx, &J2, 1, assign.data(), &nm2); });
            // SearchForInitialization(F1 = last, F2 = cur): host arrays vs two resident frames
            {
                std::vector<float> px(n1), py(n1), wr(n1, 100.0f), zero(n1, 0.0f), mx(n1, sz1.empty() ? 1.0f : 3.5831808f), ang1(n1);
                std::vector<uint8_t> oct0(n1);
                for (int i = 0; i < n1; ++i) {
                    px[i] = k1[i].x; py[i] = k1[i].y; ang1[i] = k1[i].angle; oct0[i] = k1[i].octave == 0;
                }
                afv_proj_job JI;
                std::memset(&JI, 0, sizeof(JI));
                JI.struct_size = sizeof(JI);
                JI.desc = d2.data(); JI.n = n2; JI.desc_bytes = 32; JI.x = x2.data(); JI.y = y2.data(); JI.size = s2.data(); JI.angle = a2f.data();
                JI.grid_inv_w = 64.0f / (float)W; JI.grid_inv_h = 48.0f / (float)H; JI.grid_cols = 64; JI.grid_rows = 48;
                JI.nq = n1; JI.qdesc = d1.data(); JI.qvalid = oct0.data(); JI.qu = px.data(); JI.qv = py.data(); JI.qr = wr.data();
                JI.qmin_size = zero.data(); JI.qmax_size = mx.data(); JI.qangle = ang1.data();
                JI.th_high = 75.f; JI.nnratio = 0.9f; JI.size_tol = 1.2f; JI.inv_size_tol = 1.0f / 1.2f; JI.check_orientation = 1;
                std::vector<int32_t> m12(cap);
                int32_t nmi = 0, nmi2 = 0;
                t_init_old = time_us(reps, [&] { afv_match_initialization(ctx, &JI, 1, m12.data(), &nmi); });
                afv_frame_extract(cur, f2s.data(), W, H, W, nullptr, nullptr, 0, nullptr);
                t_init_frame = time_us(reps, [&] { afv_frame_match_initialization(lastf, cur, px.data(), py.data(), 100.0f, 75.f, 0.9f, 1, m12.data(), &nmi2); });
                nm_init = nmi2;
                if (nmi != nmi2) fprintf(stderr, "WARNING: SearchForInitialization disagrees: %d vs %d\n", nmi, nmi2);
            }
        }
        if (cur) afv_frame_destroy(cur);
        if (lastf) afv_frame_destroy(lastf);
        if (table) afv_table_destroy(table);
        if (voc) afv_vocab_destroy(ctx, voc);
    }
    printf("{\"afv_orb_extract_us\": %.1f, \"afv_orb_extract_pinned_input_us\": %.1f, \"afv_match_bow_us\": %.1f, \"keypoints\": %d, \"matches\": %d, \"reps\": %d, "
           "\"tracking_frame\": {\"chain_us\": %.1f, \"chain_queries_by_reference_us\": %.1f, \"frame_extract_us\": %.1f, \"frame_bow_transform_us\": %.1f, "
           "\"frame_projection_lastframe_1000q_us\": %.1f, \"frame_projection_lastframe_1000q_by_reference_us\": %.1f, "
           "\"frame_projection_localmap_2000q_us\": %.1f, \"frame_fuse_2000q_us\": %.1f, \"matches_fuse\": %d, \"promote_us\": %.1f, \"frame_search_by_bow_kf_f_us\": %.1f, "
           "\"host_array_chain_us\": %.1f, \"host_array_bow_transform_us\": %.1f, \"host_array_projection_1000q_us\": %.1f, "
           "\"host_array_projection_2000q_us\": %.1f, \"initialization_host_arrays_us\": %.1f, \"initialization_resident_frames_us\": %.1f, "
           "\"matches_lastframe\": %d, \"matches_localmap\": %d, \"matches_initialization\": %d, \"matches_bow_kf_f\": %d, \"vocabulary_nodes\": %d, "
           "\"stages\": \"extract+grid -> ComputeBoW (k=10, L=6) -> SearchByProjection(cur,last) 1000 q -> SearchByProjection(F, local map) 2000 q -> promote; every stage a synchronous C call with host results\"}}\n",
           t_pageable, t_pinned, t_match, n2, nm, reps, t_chain, t_chain_ref, t_fextract, t_fbow, t_fproj1, t_fproj1_ref, t_fproj2, t_ffuse, nm_fuse, t_fpromote, t_fbowmatch,
           t_old_chain, t_old_bow, t_old_proj1, t_old_proj2, t_init_old, t_init_frame, nm_p1, nm_p2, nm_init, nm_bf, vocab_nodes);
    afv_destroy(ctx);
    return 0;
}
