// adapter_selftest.cpp — compile check (always) and run check (GPU box) of the C++ host adapter.
//   adapter_selftest <frame.raw> <w> <h> <out_prefix>   extracts with FeatureExtractor_orb32_hip and matches the frame
//   against itself shifted by 4 px; writes <out_prefix>.kps / .desc / .match for the python test to compare with the
//   oracle.  Exit code 0 on success, 3 when no HIP device is present (afv_create -> AFV_ENODEV).
#include <cstdio>
#include <cstring>
#include <fstream>

#include "afv_adapter.hpp"

static bool dump(const std::string &path, const void *p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(p), (std::streamsize)n);
    return (bool)f;
}

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s frame.raw w h out_prefix\n", argv[0]);
        return 2;
    }
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const std::string out = argv[4];
    afv::Image img, img2;
    img.grayImg.create(h, w);
    {
        std::ifstream f(argv[1], std::ios::binary);
        f.read(reinterpret_cast<char *>(img.grayImg.ptr()), (std::streamsize)w * h);
        if (!f) return 2;
    }
    img2.grayImg.create(h, w);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img2.grayImg.ptr(y)[x] = img.grayImg.ptr(y)[(x + w - 4) % w];  // np.roll(img, 4, axis=1)

    {   // probe for a device first: the adapter itself terminates on failure, like the reference's plugins
        afv_orb_params p;
        afv_default_orb_params(&p);
        afv_ctx *probe = nullptr;
        const int rc = afv_create(0, &p, &probe);
        if (rc == AFV_ENODEV) {
            std::fprintf(stderr, "no HIP device\n");
            return 3;
        }
        afv_destroy(probe);
    }
    auto settings = std::make_shared<afv::FeatureExtractorSettings>();
    afv::FeatureExtractor_orb32_hip extractor(1000, settings, 0, w, h);
    std::vector<afv::KeyPoint> k1, k2;
    afv::Mat8 d1, d2;
    std::vector<afv::Mat2f> s2, inf;
    std::vector<float> size;
    extractor(img, k1, d1, s2, inf, size);  // 6-arg operator()
    {   // the virtuals one by one (FeatureExtractor.h:123-128): detectKeypoints -> filterKeypoints -> computeDescriptors -> merge == detectAndCompute
        std::map<int, std::vector<afv::KeyPoint>> kl;
        std::map<int, afv::Mat8> dl;
        extractor.detectKeypoints(kl, img, settings->detectTh, settings->nOctaves);
        extractor.filterKeypoints(kl, img.grayImg, img.grayImg);
        extractor.computeDescriptors(dl, kl, img);
        std::vector<afv::KeyPoint> km;
        std::vector<uint8_t> dm;
        for (auto &lk : kl) {
            km.insert(km.end(), lk.second.begin(), lk.second.end());
            dm.insert(dm.end(), dl[lk.first].data.begin(), dl[lk.first].data.end());
        }
        if (km.size() != k1.size() || std::memcmp(km.data(), k1.data(), km.size() * sizeof(afv::KeyPoint)) != 0) return 7;
        if (dm.size() != d1.data.size() || std::memcmp(dm.data(), d1.ptr(), dm.size()) != 0) return 7;
    }
    extractor(img2, k2, d2);                // 3-arg operator()
    if (settings->ON_automaticTuning || k1.empty() || (int)size.size() != (int)k1.size()) return 4;

    afv::FeatureMatcherHip::setDescriptorDistanceThresholds(75.0f);
    afv::FeatureMatcherHip matcher(extractor.context(), 0.6f, true);
    std::vector<float> a1(k1.size()), a2(k2.size());
    for (size_t i = 0; i < k1.size(); ++i) a1[i] = k1[i].angle;
    for (size_t i = 0; i < k2.size(); ++i) a2[i] = k2[i].angle;
    afv::FeatureView v1, v2;
    v1.descriptors = d1.ptr(); v1.N = (int)k1.size(); v1.angles = a1.data();
    v2.descriptors = d2.ptr(); v2.N = (int)k2.size(); v2.angles = a2.data();
    std::vector<int> m21;
    const int nm = matcher.SearchByBoW(v2, v1, m21);
    std::printf("%zu %zu %d\n", k1.size(), k2.size(), nm);
    bool ok = dump(out + ".kps1", k1.data(), k1.size() * sizeof(afv::KeyPoint)) && dump(out + ".desc1", d1.ptr(), d1.data.size()) &&
              dump(out + ".kps2", k2.data(), k2.size() * sizeof(afv::KeyPoint)) && dump(out + ".desc2", d2.ptr(), d2.data.size()) &&
              dump(out + ".match21", m21.data(), m21.size() * sizeof(int)) && dump(out + ".size1", size.data(), size.size() * 4);
    if (!ok) return 5;
    {   // mvImagePyramid of the last frame (img2): levels dumped for the Python side to compare with the oracle's pyramid
        std::vector<afv::Mat8> pyr;
        extractor.ImagePyramid(pyr);
        if ((int)pyr.size() != settings->nOctaves || pyr[0].rows != h || pyr[0].cols != w) return 6;
        std::vector<uint8_t> flat;
        for (const afv::Mat8 &m : pyr) flat.insert(flat.end(), m.data.begin(), m.data.end());
        if (!dump(out + ".pyramid2", flat.data(), flat.size())) return 6;
    }

    // ---- Vocabulary::transform + SearchByBoW over the feature vectors (KeyFrame::ComputeBoW -> SearchByBoW(KF, KF)) ----
    {
        // deterministic 2-level tree, k = 6: node descriptors from an LCG, weights 1 (0 for every 7th leaf)
        const int k = 6, L = 2, n = 1 + k + k * k;
        std::vector<int> parent((size_t)n, 0);
        std::vector<uint8_t> leaf((size_t)n, 0), nd((size_t)n * 32, 0);
        std::vector<double> wgt((size_t)n, 1.0);
        for (int i = 1 + k; i < n; ++i) { parent[i] = 1 + (i - 1 - k) / k; leaf[i] = 1; if (i % 7 == 0) wgt[i] = 0.0; }
        uint32_t x = 12345u;
        for (size_t i = 32; i < nd.size(); ++i) { x = x * 1664525u + 1013904223u; nd[i] = (uint8_t)((x >> 8) & 255u); }
        wgt[0] = 0.0;
        afv::VocabularyHip voc(extractor.context(), k, L, parent, leaf, nd, wgt);
        afv::BowVector bow1, bow2;
        afv::FeatureVector fv1, fv2;
        voc.transform(d1.ptr(), (int)k1.size(), bow1, fv1, 1);
        voc.transform(d2.ptr(), (int)k2.size(), bow2, fv2, 1);
        afv::FeatureView b1 = v1, b2 = v2;
        b1.featVec = &fv1; b2.featVec = &fv2;
        std::vector<int> m12;
        const int nb = matcher.SearchByBoW(b1, b2, m12);
        std::printf("bow %zu %zu %d\n", fv1.size(), fv2.size(), nb);
        std::vector<int> flat;  // (node, feature) pairs of fv1
        for (const auto &kv : fv1) for (unsigned f : kv.second) { flat.push_back((int)kv.first); flat.push_back((int)f); }
        ok = dump(out + ".bowmatch12", m12.data(), m12.size() * sizeof(int)) && dump(out + ".fv1", flat.data(), flat.size() * sizeof(int)) &&
             dump(out + ".vocdesc", nd.data(), nd.size());
        if (!ok) return 5;
    }
    // ---- projection-guided searches through the FeatureMatcherHip wrappers (the python test rebuilds the same views from the
    // dumped keypoints and compares every output with the oracle) ----
    {
        std::vector<afv::KeyPoint> k2b;
        afv::Mat8 d2b;
        std::vector<afv::Mat2f> s2b, infb;
        std::vector<float> size2;
        extractor(img2, k2b, d2b, s2b, infb, size2);
        if (k2b.size() != k2.size()) return 7;
        const size_t n1 = k1.size(), n2 = k2.size();
        std::vector<float> x1(n1), y1(n1), inf1(n1), x2(n2), y2(n2), inf2(n2);
        for (size_t i = 0; i < n1; ++i) { x1[i] = k1[i].pt.x; y1[i] = k1[i].pt.y; inf1[i] = inf[i].m[0][0]; }
        for (size_t i = 0; i < n2; ++i) { x2[i] = k2[i].pt.x; y2[i] = k2[i].pt.y; inf2[i] = infb[i].m[0][0]; }
        using M = afv::FeatureMatcherHip;
        auto grid = [&](const afv::Mat8 &d, size_t n, const float *x, const float *y, const float *sz, const float *ang, const float *inf_) {
            M::FrameGridView F;
            F.descriptors = d.ptr(); F.N = (int)n; F.x = x; F.y = y; F.size = sz; F.angle = ang; F.inf = inf_;
            F.mfGridElementWidthInv = 64.0f / ((float)w - 0.0f);
            F.mfGridElementHeightInv = 48.0f / ((float)h - 0.0f);
            return F;
        };
        const M::FrameGridView F1 = grid(d1, n1, x1.data(), y1.data(), size.data(), a1.data(), inf1.data());
        const M::FrameGridView F2 = grid(d2, n2, x2.data(), y2.data(), size2.data(), a2.data(), inf2.data());
        // queries: the other frame's keypoints "projected" with the known 4 px shift; window 15 * size; size band /1.2 .. *1.2
        std::vector<float> u2(n2), r2(n2), lo2(n2), hi2(n2), u1(n1), r1(n1), lo1(n1), hi1(n1), zero1(n1, 0.f), big1(n1, 3.6f), win1(n1, 100.f);
        std::vector<uint8_t> oct0(n1);
        for (size_t i = 0; i < n2; ++i) { u2[i] = x2[i] - 4.0f; r2[i] = 15.0f * size2[i]; lo2[i] = size2[i] / 1.2f; hi2[i] = size2[i] * 1.2f; }
        for (size_t i = 0; i < n1; ++i) { u1[i] = x1[i] + 4.0f; r1[i] = 15.0f * size[i]; lo1[i] = size[i] / 1.2f; hi1[i] = size[i] * 1.2f; oct0[i] = k1[i].octave == 0; }
        M::ProjectionQueries Q2;  // frame-2 keypoints searched in frame 1
        Q2.descriptors = d2.ptr(); Q2.n = (int)n2; Q2.u = u2.data(); Q2.v = y2.data(); Q2.r = r2.data(); Q2.min_size = lo2.data();
        Q2.max_size = hi2.data(); Q2.angle = a2.data();
        M::ProjectionQueries Q1;  // frame-1 keypoints searched in frame 2
        Q1.descriptors = d1.ptr(); Q1.n = (int)n1; Q1.u = u1.data(); Q1.v = y1.data(); Q1.r = r1.data(); Q1.min_size = lo1.data();
        Q1.max_size = hi1.data(); Q1.angle = a1.data();
        M::ProjectionQueries QI;  // SearchForInitialization: F1's level-0 features, vbPrevMatched = own position, window 100
        QI.descriptors = d1.ptr(); QI.n = (int)n1; QI.u = x1.data(); QI.v = y1.data(); QI.r = win1.data(); QI.min_size = zero1.data();
        QI.max_size = big1.data(); QI.valid = oct0.data(); QI.angle = a1.data();
        afv::FeatureMatcherHip pm(extractor.context(), 0.8f, true);
        std::vector<int> o_local, o_last, o_reloc, o_sim3p, o_fuse, o_fuse3, o_sim3, o_init;
        const int c0 = pm.SearchByProjection(F1, Q2, o_local);
        const int c1 = pm.SearchByProjection_LastFrame(F1, Q2, o_last);
        const int c2 = pm.SearchByProjection_Reloc(F1, Q2, 60.0f, o_reloc);
        const int c3 = pm.SearchByProjection_Sim3(F1, Q2, o_sim3p);
        const int c4 = pm.Fuse(F1, Q2, o_fuse);
        M::FrameGridView F1n = F1;
        F1n.inf = nullptr;
        const int c5 = pm.Fuse(F1n, Q2, o_fuse3);
        const int c6 = pm.SearchBySim3(F1, Q1, F2, Q2, o_sim3);
        afv::FeatureMatcherHip im(extractor.context(), 0.9f, true);
        const int c7 = im.SearchForInitialization(F2, QI, o_init);
        std::printf("proj %d %d %d %d %d %d %d %d\n", c0, c1, c2, c3, c4, c5, c6, c7);
        auto dumpv = [&](const char *name, const std::vector<int> &v) { return dump(out + name, v.data(), v.size() * sizeof(int)); };
        ok = dumpv(".p_local", o_local) && dumpv(".p_last", o_last) && dumpv(".p_reloc", o_reloc) && dumpv(".p_sim3p", o_sim3p) &&
             dumpv(".p_fuse", o_fuse) && dumpv(".p_fuse3", o_fuse3) && dumpv(".p_sim3", o_sim3) && dumpv(".p_init", o_init) &&
             dump(out + ".size2", size2.data(), size2.size() * 4) && dump(out + ".inf1", inf1.data(), inf1.size() * 4);
        if (!ok) return 5;
        // ---- the same searches through DeviceFrame (resident frames: nothing of the frame is uploaded again): must reproduce the host-array
        // results above element for element ----
        {
            afv::DeviceFrame D1(extractor.context(), 0.0f, 0.0f, (float)w, (float)h), D2(extractor.context(), 0.0f, 0.0f, (float)w, (float)h);
            std::vector<afv::KeyPoint> kd1, kd2;
            afv::Mat8 dd1, dd2;
            D1.Extract(img, kd1, dd1);
            D2.Extract(img2, kd2, dd2);
            if (kd1.size() != n1 || kd2.size() != n2 || std::memcmp(kd1.data(), k1.data(), n1 * sizeof(afv::KeyPoint)) != 0 || dd1.data != d1.data ||
                dd2.data != d2.data)
                return 8;
            std::vector<int> f_local, f_last, f_reloc, f_fuse, f_fuse3, f_init;
            const int e0 = pm.SearchByProjection(D1, Q2, nullptr, f_local);
            const int e1 = pm.SearchByProjection_LastFrame(D1, Q2, nullptr, f_last);
            const int e2 = pm.SearchByProjection_Reloc(D1, Q2, nullptr, 60.0f, f_reloc);
            const int e4 = pm.Fuse(D1, Q2, true, f_fuse);
            const int e5 = pm.Fuse(D1, Q2, false, f_fuse3);
            std::vector<float> px(x1), py(y1);
            const int e7 = im.SearchForInitialization(D1, D2, px, py, 100, f_init);
            if (e0 != c0 || e1 != c1 || e2 != c2 || e4 != c4 || e5 != c5 || e7 != c7 || f_local != o_local || f_last != o_last || f_reloc != o_reloc ||
                f_fuse != o_fuse || f_fuse3 != o_fuse3 || f_init != o_init) {
                std::fprintf(stderr, "resident-frame searches differ from the host-array ones: %d/%d %d/%d %d/%d %d/%d %d/%d %d/%d\n", e0, c0, e1, c1, e2,
                             c2, e4, c4, e5, c5, e7, c7);
                return 8;
            }
            std::printf("frame %d %d %d %d %d %d\n", e0, e1, e2, e4, e5, e7);
        }
    }
    // ---- float descriptors through the matcher's dispatch (FeatureMatcher.cc:1508-1531): SearchByBoW on 128-float rows (one node = brute
    //      force) must equal afv_match_l2, the float matcher of config #3, called directly ----
    {
        const int n1 = (int)k1.size(), n2 = (int)k2.size(), dim = 128;
        auto floaten = [&](const afv::Mat8 &d, int n) {
            std::vector<float> f((size_t)n * dim);
            for (int i = 0; i < n; ++i)
                for (int b = 0; b < dim; ++b)
                    f[(size_t)i * dim + b] = ((d.ptr(i)[b >> 3] >> (7 - (b & 7))) & 1) * 0.75f + (float)((i * 131 + b * 29) % 97) * (0.2f / 97.0f);
            return f;
        };
        const std::vector<float> f1 = floaten(d1, n1), f2 = floaten(d2, n2);
        afv::FeatureMatcherHip::setDescriptorDistanceThresholds(20.0f);
        afv::FeatureMatcherHip fm(extractor.context(), 0.8f, true);
        afv::FeatureView fa, fb;
        fa.descriptors = reinterpret_cast<const uint8_t *>(f1.data()); fa.N = n1; fa.float_dim = dim;
        fb.descriptors = reinterpret_cast<const uint8_t *>(f2.data()); fb.N = n2; fb.float_dim = dim;
        std::vector<int> m12;
        const int nf = fm.SearchByBoW(fa, fb, m12);
        std::vector<int32_t> want((size_t)std::max(n1, 1), -1);
        int32_t nw = 0;
        if (afv_match_l2(extractor.context(), f1.data(), n1, f2.data(), n2, dim, nullptr, nullptr, 20.0f, 0.8f, want.data(), &nw) != AFV_OK) return 9;
        std::printf("float128 %d %d\n", nf, (int)nw);
        if (nf != nw || nf < 50) return 9;
        for (int i = 0; i < n1; ++i)
            if (m12[(size_t)i] != want[(size_t)i]) return 9;
        afv::FeatureMatcherHip::setDescriptorDistanceThresholds(75.0f);
    }
    // ---- AKAZE61 plugin ----
    {
        auto s2 = std::make_shared<afv::FeatureExtractorSettings>();
        afv::FeatureExtractorSettings::numOctaves0 = 8;
        afv::FeatureExtractorSettings::scaleFactor0 = 1.1892f;
        s2->detectTh = 0.0005f;
        afv::FeatureExtractor_akaze61_hip akz(1000, s2, 0, w, h);
        std::vector<afv::KeyPoint> ka;
        afv::Mat8 da;
        akz.detectAndCompute(img, ka, da);
        std::printf("akaze %zu %d\n", ka.size(), da.cols);
        if (ka.empty() || da.cols != 61 || akz.GetKeypointOctave(ka[0]) != ka[0].class_id) return 6;
        ok = dump(out + ".akz_kps", ka.data(), ka.size() * sizeof(afv::KeyPoint)) && dump(out + ".akz_desc", da.ptr(), da.data.size());
    }
    return ok ? 0 : 5;
}
