// afv_api.hip — host runtime behind the C-ABI of include/afv_hip.h.
// Owns the HIP stream, the per-geometry tables (level sizes, resize coefficients, quotas) and the device scratch
// of one context; stages host-pointer calls; enqueues the kernel pipeline
//   resize x (nlevels-1) -> FAST+NMS+Harris -> retainBest x2 + quadtree -> IC + blur + rBRIEF.
// No CPU fallback exists: without a HIP device afv_create fails with AFV_ENODEV.
#include <chrono>

#include "afv_runtime.h"

static const char *k_errors[] = {"ok", "invalid argument", "no usable HIP device", "out of memory", "HIP runtime error",
                                 "output capacity too small", "unsupported", "device-side wait timed out"};


extern "C" void afv_default_orb_params(afv_orb_params *p) {
    if (!p) return;
    p->nfeatures = 1000;
    p->nlevels = 8;
    p->scale_factor = 1.2f;
    p->fast_threshold = 20;
    p->max_width = 640;
    p->max_height = 480;
    p->max_batch = 1;
}

extern "C" int afv_abi_version(void) { return AFV_ABI_VERSION; }

extern "C" const char *afv_strerror(int code) {
    const int i = -code;
    if (i < 0 || i > 7) return "unknown error";
    return k_errors[i];
}
extern "C" const char *afv_last_error(const afv_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }
extern "C" void *afv_stream(afv_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// ---- geometry (host).  Mirrors cv::ORB's level sizes / quotas and FeatureExtractor.cpp:97-108 ----
static int build_geometry(const afv_orb_params &p, int w, int h, int max_batch, Geo &g) {
    std::memset(&g, 0, sizeof(g));
    if (p.nlevels < 1 || p.nlevels > AFV_MAX_LEVELS || w < 1 || h < 1 || w > 4095 || h > 4095) return AFV_EINVAL;
    g.nlevels = p.nlevels;
    g.width = w;
    g.height = h;
    g.fast_threshold = std::min(std::max(p.fast_threshold, 0), 255);
    {
        const float scale = 1.f / ((1 << 2) * 7 * 255.f);
        g.harris_scale4 = scale * scale * scale * scale;
    }
    g.n_ini = (int)roundf((float)w / (float)h);  // ORBextractor.cc:243
    if (g.n_ini < 1 || g.n_ini > 16) return AFV_EUNSUPPORTED;
    g.h_x = (float)w / (float)g.n_ini;
    // quadtree quotas (FeatureExtractor.cpp:97-108)
    int quota[AFV_MAX_LEVELS], cvq[AFV_MAX_LEVELS];
    {
        const float factor = 1.0f / p.scale_factor;
        float desired = (float)p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
        int sum = 0;
        for (int l = 0; l < p.nlevels - 1; ++l) {
            quota[l] = cv_round(desired);
            sum += quota[l];
            desired *= factor;
        }
        quota[p.nlevels - 1] = std::max(p.nfeatures - sum, 0);
    }
    {   // cv::ORB computeKeyPoints with nfeatures*10 (Feature_orb32.cpp:22), scaleFactor held as double
        const int nf = p.nfeatures * 10;
        const float factor = (float)(1.0 / (double)p.scale_factor);
        float desired = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
        int sum = 0;
        for (int l = 0; l < p.nlevels - 1; ++l) {
            cvq[l] = cv_round(desired);
            sum += cvq[l];
            desired *= factor;
        }
        cvq[p.nlevels - 1] = std::max(nf - sum, 0);
    }
    size_t pyr_off = 0, cand_off = 0;
    int tile_base = 0, sel_base = 0, desc_blk_base = 0;
    for (int l = 0; l < p.nlevels; ++l) {
        LevelGeo &L = g.lv[l];
        L.scale = (float)std::pow((double)p.scale_factor, (double)l);
        L.inv_scale = 1.0f / L.scale;
        L.w = cv_round((float)w * L.inv_scale);
        L.h = cv_round((float)h * L.inv_scale);
        if (L.w < 32 || L.h < 32) return AFV_EUNSUPPORTED;  // single-reflection apron needs >= 32 px levels
        if (l > 0 && !afv_resize_window_ok(g.lv[l - 1].w, g.lv[l - 1].h, L.w, L.h)) return AFV_EUNSUPPORTED;  // level ratio > ~2.37
        L.pitch = (int)align_up((size_t)L.w, 64);
        L.tiles_x = (L.w + FT_W - 1) / FT_W;
        L.tiles_y = (L.h + FT_H - 1) / FT_H;
        L.dv_tiles_x = afv_div_magic((uint32_t)L.tiles_x);
        L.tile_base = tile_base;
        tile_base += L.tiles_x * L.tiles_y;
        L.quota = quota[l];
        L.cv_quota = cvq[l];
        L.cand_cap = ((L.w + 1) / 2) * ((L.h + 1) / 2);
        // DistributeOctTree returns quota .. quota+2 nodes when saturated, but the first split round is unconditional
        // (ORBextractor.cc:283-366) and leaves up to 4 * n_ini nodes even when the quota is smaller
        L.sel_cap = std::max(L.quota + 3, 4 * g.n_ini);
        L.sel_base = sel_base;
        sel_base += L.sel_cap;
        L.desc_blk_base = desc_blk_base;
        desc_blk_base += (L.sel_cap + AFV_KP_PER_BLOCK - 1) / AFV_KP_PER_BLOCK;  // keypoints per k_describe block
        L.pyr_frame_stride = align_up((size_t)L.h * L.pitch + 64, 256);
        L.pyr_off = pyr_off;
        if (l > 0) pyr_off += L.pyr_frame_stride * (size_t)max_batch;
        L.cand_frame_stride = (size_t)L.cand_cap;
        L.cand_off = cand_off;
        cand_off += L.cand_frame_stride * (size_t)max_batch;
    }
    g.total_tiles = tile_base;
    g.dv_total_tiles = afv_div_magic((uint32_t)tile_base);
    g.sel_per_frame = sel_base;
    g.dv_desc_per_frame = afv_div_magic((uint32_t)afv_describe_blocks_per_frame(&g));
    return AFV_OK;
}

static size_t geo_pyr_bytes(const Geo &g, int max_batch) {
    const LevelGeo &L = g.lv[g.nlevels - 1];
    return g.nlevels > 1 ? L.pyr_off + L.pyr_frame_stride * (size_t)max_batch : 256;
}
static size_t geo_cand_elems(const Geo &g, int max_batch) {
    const LevelGeo &L = g.lv[g.nlevels - 1];
    return L.cand_off + L.cand_frame_stride * (size_t)max_batch;
}

// OpenCV resize_bitExact / interpolation_linear<uchar>::getCoeffs: per destination index {offset, weight of right tap}
static void resize_table(int src, int dst, short2 *t) {
    const double inv_scale = (double)dst / (double)src;
    const double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        const double f = scale * ((double)d + 0.5) - 0.5;
        const int i = (int)std::floor(f);
        short2 e;
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                e.x = (short)i;
                e.y = (short)lrint((f - (double)i) * 256.0);
            } else {
                e.x = (short)(src - 1);
                e.y = 0;
            }
        } else {
            e.x = 0;
            e.y = 0;
        }
        t[d] = e;
    }
}

// Region bookkeeping of k_pyramid_fused (k_pyramid.hip): per level and per tile index of the TOP level, the range a workgroup
// computes (need) and the range it stores (own), x and y separately.  own: the top level is cut into tiles of tw x th; one level
// down a tile owns what lies between the source offsets of its own and of its right / lower neighbour's first pixel (monotone
// tables: a disjoint cover).  need: the owned range (x: widened to whole dwords) united with the source span of the level above.
static int ceil_log2(int v) {
    int lg = 0;
    while ((1 << lg) < v) ++lg;
    return lg;
}
static bool plan_pyr_fuse(const Geo &g, const short2 *tab, const size_t *tab_off_x, const size_t *tab_off_y, int TW, int TH,
                          std::vector<short4> &reg, PyrFusePlan &P) {
    const int NL = g.nlevels, L = NL - 1;
    if (NL < 2 || TW < 4 || (TW & 3) || TH < 1) return false;
    const int ntx = (g.lv[L].w + TW - 1) / TW, nty = (g.lv[L].h + TH - 1) / TH;
    reg.assign((size_t)NL * (ntx + nty), short4{0, 0, 0, 0});
    int maxlen[2][AFV_MAX_LEVELS] = {};
    for (int ax = 0; ax < 2; ++ax) {
        const bool is_x = ax == 0;
        const int nt = is_x ? ntx : nty, T = is_x ? TW : TH;
        auto dim = [&](int l) { return is_x ? g.lv[l].w : g.lv[l].h; };
        auto offs = [&](int l) { return tab + (is_x ? tab_off_x[l] : tab_off_y[l]); };  // table of level l: offsets into level l - 1
        short4 *out = reg.data() + (is_x ? 0 : (size_t)NL * ntx);
        for (int t = 0; t < nt; ++t) {
            int own_lo[AFV_MAX_LEVELS], own_hi[AFV_MAX_LEVELS], lo[AFV_MAX_LEVELS], hi[AFV_MAX_LEVELS];
            own_lo[L] = t * T;
            own_hi[L] = std::min((t + 1) * T, dim(L));
            for (int l = L - 1; l >= 1; --l) {
                own_lo[l] = t == 0 ? 0 : (own_lo[l + 1] < dim(l + 1) ? offs(l + 1)[own_lo[l + 1]].x : dim(l));
                own_hi[l] = t == nt - 1 ? dim(l) : (own_hi[l + 1] < dim(l + 1) ? offs(l + 1)[own_hi[l + 1]].x : dim(l));
            }
            for (int l = L; l >= 1; --l) {
                int a = own_lo[l], b = own_hi[l];
                if (is_x) {
                    a &= ~3;
                    b = (b + 3) & ~3;
                }
                int lo_ = a, hi_ = b - 1;
                if (l < L) {
                    const int n0 = lo[l + 1], n1 = std::min(hi[l + 1], dim(l + 1) - 1);
                    const int lo2 = offs(l + 1)[n0].x, hi2 = std::min(offs(l + 1)[n1].x + 1, dim(l) - 1);
                    if (b > a) {
                        lo_ = std::min(lo_, lo2);
                        hi_ = std::max(hi_, hi2);
                    } else {
                        lo_ = lo2;
                        hi_ = hi2;
                    }
                }
                if (is_x) {
                    lo_ &= ~3;
                    hi_ = lo_ + ((hi_ - lo_ + 4) & ~3) - 1;
                }
                lo[l] = lo_;
                hi[l] = hi_;
                out[(size_t)l * nt + t] = short4{(short)lo_, (short)hi_, (short)own_lo[l], (short)own_hi[l]};
                maxlen[ax][l] = std::max(maxlen[ax][l], hi_ - lo_ + 1);
            }
            {   // the level-0 window
                const int n0 = lo[1], n1 = std::min(hi[1], dim(1) - 1);
                int lo0 = offs(1)[n0].x;
                const int hi0 = std::min(offs(1)[n1].x + 1, dim(0) - 1);
                if (is_x) lo0 &= ~3;
                out[t] = short4{(short)lo0, (short)hi0, 0, 0};
                maxlen[ax][0] = std::max(maxlen[ax][0], hi0 - lo0 + 1);
            }
        }
    }
    std::memset(&P, 0, sizeof(P));
    P.nlevels = NL;
    P.ntx = ntx;
    P.nty = nty;
    for (int l = 0; l < NL; ++l) {
        P.pitch[l] = (int)align_up((size_t)maxlen[0][l], 4);
        P.maxh[l] = maxlen[1][l];
        P.lg_q[l] = ceil_log2(P.pitch[l] / 4);
        if (P.lg_q[l] > 10) return false;  // more dword slots per row than the workgroup has threads
        if (l > 0) {
            P.tabx[l] = (int)tab_off_x[l];
            P.taby[l] = (int)tab_off_y[l];
            // the dword-read form of the level loop: left tap of column c + 3 at most 6 bytes past that of column c
            const short2 *xt = tab + tab_off_x[l];
            bool narrow = true;
            for (int x = 0; x + 3 < g.lv[l].w && narrow; x += 4) narrow = xt[x + 3].x - xt[x].x <= 6;
            P.narrow[l] = narrow ? 1 : 0;
        }
    }
    return true;
}

// the device image of a plan (layout: afv_device.h) and the kernel's arguments / LDS size
static void pack_pyr_fuse(const Geo &g, const short2 *tab, const std::vector<short4> &reg, const PyrFusePlan &P, std::vector<uint8_t> &blob,
                          PyrFuseArgs &A, size_t &lds) {
    const int NL = P.nlevels;
    std::memset(&A, 0, sizeof(A));
    A.nlevels = NL;
    A.ntx = P.ntx;
    A.nty = P.nty;
    A.w0 = g.width;
    PfLevelC C[AFV_MAX_LEVELS];
    std::memset(C, 0, sizeof(C));
    size_t sx = 0, sy = 0;
    for (int l = 0; l < NL; ++l) {
        C[l].lds_pitch = P.pitch[l];
        C[l].lg_q = P.lg_q[l];
        C[l].narrow = P.narrow[l];
        C[l].gpitch = g.lv[l].pitch;
        C[l].pyr_off = g.lv[l].pyr_off;
        C[l].fstride = g.lv[l].pyr_frame_stride;
        C[l].x_rx = (int)sx;
        sx += 16;
        C[l].y_ry = (int)sy;
        sy += 16;
        if (l > 0) {
            C[l].x_xt = (int)sx;
            sx = align_up(sx + (size_t)P.pitch[l] * sizeof(short2), 16);
            C[l].y_yt = (int)sy;
            sy = align_up(sy + (size_t)P.maxh[l] * sizeof(short2), 16);
        }
    }
    A.sx = (int)sx;
    A.sy = (int)sy;
    A.off_x = (int)sizeof(C);
    A.off_y = A.off_x + (int)sx * P.ntx;
    blob.assign((size_t)A.off_y + sy * P.nty, 0);
    std::memcpy(blob.data(), C, sizeof(C));
    for (int ax = 0; ax < 2; ++ax) {
        const bool is_x = ax == 0;
        const int nt = is_x ? P.ntx : P.nty;
        const short4 *R = reg.data() + (is_x ? 0 : (size_t)NL * P.ntx);
        for (int t = 0; t < nt; ++t) {
            uint8_t *part = blob.data() + (is_x ? (size_t)A.off_x + sx * t : (size_t)A.off_y + sy * t);
            for (int l = 0; l < NL; ++l) {
                const short4 r = R[(size_t)l * nt + t];
                std::memcpy(part + (is_x ? C[l].x_rx : C[l].y_ry), &r, sizeof(r));
                if (l == 0) continue;
                const short4 rp = R[(size_t)(l - 1) * nt + t];
                const short2 *tb = tab + (is_x ? P.tabx[l] : P.taby[l]);
                const int dim = is_x ? g.lv[l].w : g.lv[l].h;
                short2 *out = reinterpret_cast<short2 *>(part + (is_x ? C[l].x_xt : C[l].y_yt));
                for (int i = 0; i <= r.y - r.x; ++i) {
                    short2 e{0, 0};  // columns past the level's width (dword padding): offset 0, weight 0
                    if (r.x + i < dim) {
                        e = tb[r.x + i];
                        e.x = (short)(e.x - rp.x);  // relative to the source region
                    }
                    out[i] = e;
                }
            }
        }
    }
    size_t off = sizeof(C);
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 16);
        return (int)o;
    };
    A.lds_x = take(sx);
    A.lds_y = take(sy);
    size_t buf[2] = {0, 0};
    for (int l = 0; l < NL; ++l) buf[l & 1] = std::max(buf[l & 1], (size_t)P.pitch[l] * P.maxh[l]);
    A.off_buf[0] = take(buf[0] + 16);  // + slack: the dword-read form fetches up to 12 bytes past a row's last tap
    A.off_buf[1] = take(buf[1] + 16);
    lds = off;
}
// the plan for the context's geometry
static int build_pyr_fuse(afv_ctx *c, const Geo &g, const short2 *tab) {
    std::vector<short4> reg;
    PyrFusePlan P;
    c->pf_ok = false;
    if (!plan_pyr_fuse(g, tab, c->tab_off_x, c->tab_off_y, c->pf_tw, c->pf_th, reg, P)) return AFV_OK;
    std::vector<uint8_t> blob;
    pack_pyr_fuse(g, tab, reg, P, blob, c->pf, c->pf_lds);
    if ((size_t)(sizeof(PfLevelC) * AFV_MAX_LEVELS + c->pf.sx + c->pf.sy) / 4 > 2 * 1024) return AFV_OK;  // the prologue copies <= 2 dwords per thread
    if (!afv_pyramid_fused_prepare(c->pf_lds)) return AFV_OK;
    if (blob.size() > c->pf_blob_cap) {
        if (c->d_pf_blob) (void)hipFree(c->d_pf_blob);
        c->d_pf_blob = nullptr;
        c->pf_blob_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_pf_blob, blob.size()));
        c->pf_blob_cap = blob.size();
    }
    HIPCHK(c, hipMemcpy(c->d_pf_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    c->pf.blob = c->d_pf_blob;
    c->pf_ok = true;
    return AFV_OK;
}

// host-only view of the plan and of the coefficient tables (no device needed): tests/test_host_logic.py replays the one-launch pyramid
// on the CPU from exactly these numbers and compares it with the level-by-level resize
extern "C" int afv_debug_pyramid_plan(const afv_orb_params *p, int width, int height, int tile_w, int tile_h, int16_t *regions, int regions_cap,
                                      int32_t *info, int16_t *tables, int tables_cap) {
    if (!p || !regions || !info) return AFV_EINVAL;
    Geo g;
    const int rc = build_geometry(*p, width, height, 1, g);
    if (rc) return rc;
    size_t tox[AFV_MAX_LEVELS] = {}, toy[AFV_MAX_LEVELS] = {}, n = 0;
    for (int l = 1; l < g.nlevels; ++l) n += (size_t)g.lv[l].w + (size_t)g.lv[l].h;
    std::vector<short2> tab(std::max<size_t>(n, 1));
    size_t off = 0;
    for (int l = 1; l < g.nlevels; ++l) {
        tox[l] = off;
        resize_table(g.lv[l - 1].w, g.lv[l].w, tab.data() + off);
        off += (size_t)g.lv[l].w;
        toy[l] = off;
        resize_table(g.lv[l - 1].h, g.lv[l].h, tab.data() + off);
        off += (size_t)g.lv[l].h;
    }
    std::vector<short4> reg;
    PyrFusePlan A;
    if (!plan_pyr_fuse(g, tab.data(), tox, toy, tile_w, tile_h, reg, A)) return AFV_EUNSUPPORTED;
    std::vector<uint8_t> blob;
    PyrFuseArgs args;
    size_t lds = 0;
    pack_pyr_fuse(g, tab.data(), reg, A, blob, args, lds);
    if ((int)reg.size() * 4 > regions_cap || (tables && (int)n * 2 > tables_cap)) return AFV_ECAPACITY;
    std::memcpy(regions, reg.data(), reg.size() * sizeof(short4));
    if (tables) std::memcpy(tables, tab.data(), n * sizeof(short2));
    int k = 0;
    info[k++] = g.nlevels;
    info[k++] = A.ntx;
    info[k++] = A.nty;
    info[k++] = (int)lds;
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = g.lv[l].w;
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = g.lv[l].h;
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = A.pitch[l];
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = A.lg_q[l];
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = A.tabx[l];
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) info[k++] = A.taby[l];
    return AFV_OK;  // info: 4 + 6 * AFV_MAX_LEVELS ints
}

static int set_geometry(afv_ctx *c, int w, int h) {
    if (c->geo_valid && c->geo.width == w && c->geo.height == h) return AFV_OK;
    if (w > c->p.max_width || h > c->p.max_height) return AFV_EINVAL;
    Geo g;
    const int rc = build_geometry(c->p, w, h, c->p.max_batch, g);
    if (rc) return rc;
    // allocation layout always follows the capacity geometry so buffers never move
    if (g.sel_per_frame > c->cap_geo.sel_per_frame) return AFV_EINVAL;  // wider aspect ratio than the capacity geometry
    for (int l = 0; l < g.nlevels; ++l) {
        if (g.lv[l].cand_cap > c->cap_geo.lv[l].cand_cap || g.lv[l].pyr_frame_stride > c->cap_geo.lv[l].pyr_frame_stride)
            return AFV_EINVAL;
        g.lv[l].pyr_off = c->cap_geo.lv[l].pyr_off;
        g.lv[l].pyr_frame_stride = c->cap_geo.lv[l].pyr_frame_stride;
        g.lv[l].cand_off = c->cap_geo.lv[l].cand_off;
        g.lv[l].cand_frame_stride = c->cap_geo.lv[l].cand_frame_stride;
    }
    std::vector<short2> tab(c->tab_elems);
    size_t off = 0;
    for (int l = 1; l < g.nlevels; ++l) {
        c->tab_off_x[l] = off;
        resize_table(g.lv[l - 1].w, g.lv[l].w, tab.data() + off);
        off += (size_t)g.lv[l].w;
        c->tab_off_y[l] = off;
        resize_table(g.lv[l - 1].h, g.lv[l].h, tab.data() + off);
        off += (size_t)g.lv[l].h;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->d_tab, tab.data(), off * sizeof(short2), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_geo, &g, sizeof(Geo), hipMemcpyHostToDevice));
    {   // the one-launch pyramid of the small-batch path
        const int rc_pf = build_pyr_fuse(c, g, tab.data());
        if (rc_pf) return rc_pf;
    }
    c->geo = g;
    c->geo_valid = true;
    return AFV_OK;
}

extern "C" int afv_max_keypoints_per_frame(const afv_ctx *c) {
    if (!c) return AFV_EINVAL;
    int s = 0;
    for (int l = 0; l < c->cap_geo.nlevels; ++l) s += std::max(c->cap_geo.lv[l].quota + 2, 4 * c->cap_geo.n_ini);
    return s;
}

extern "C" void afv_destroy(afv_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    afv_table_release_all(c);
    afv_frame_release_all(c);
    void *ptrs[] = {c->d_geo, c->d_tab, c->d_pyr, c->d_cand_packed, c->d_kept_xy, c->d_l1, c->d_l1_resp, c->d_l1_count, c->d_hq, c->d_hq_n, c->d_kept_resp,
                    c->d_kept_node, c->d_cand_count, c->d_sel_count, c->d_sel, c->d_frames, c->d_out_block,
                    c->d_status, c->d_match, c->d_topk, c->d_slice, c->d_tickets, c->d_pf_blob, c->d_l2_scratch, c->d_proj_ticket};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->h_stage) {
        if (c->stage_pinned) (void)hipHostFree(c->h_stage);
        else std::free(c->h_stage);
    }
    for (auto &v : c->prof_ev)
        for (hipEvent_t e : v) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->pipe_ev) (void)hipEventDestroy(e);
    if (c->stream_copy) (void)hipStreamDestroy(c->stream_copy);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int afv_create(int device, const afv_orb_params *params, afv_ctx **out) {
    if (!params || !out) return AFV_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AFV_ENODEV;
    if (device < 0 || device >= ndev) return AFV_ENODEV;
    if (params->nfeatures < 1 || params->nfeatures > 4000 || params->max_batch < 1 || params->max_batch > 65535 ||
        params->scale_factor <= 1.0f)
        return AFV_EINVAL;
    afv_ctx *c = new (std::nothrow) afv_ctx();
    if (!c) return AFV_ENOMEM;
    c->device = device;
    c->p = *params;
    int rc = build_geometry(c->p, params->max_width, params->max_height, params->max_batch, c->cap_geo);
    if (rc) {
        delete c;
        return rc;
    }
    const Geo &g = c->cap_geo;
    const int B = params->max_batch;
#define CREATE_CHK(call)                                                         \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "afv_create: %s: %s\n", #call, hipGetErrorString(e_)); \
            afv_destroy(c);                                                      \
            return e_ == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;            \
        }                                                                        \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_CHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    CREATE_CHK(hipMalloc(&c->d_geo, sizeof(Geo)));
    c->tab_elems = 0;
    for (int l = 1; l < g.nlevels; ++l) c->tab_elems += (size_t)g.lv[l].w + (size_t)g.lv[l].h;
    CREATE_CHK(hipMalloc(&c->d_tab, std::max<size_t>(c->tab_elems, 1) * sizeof(short2)));
    CREATE_CHK(hipMalloc(&c->d_pyr, geo_pyr_bytes(g, B)));
    const size_t ce = geo_cand_elems(g, B);
    CREATE_CHK(hipMalloc(&c->d_cand_packed, ce * 4));
    CREATE_CHK(hipMalloc(&c->d_l1, ce * 4));
    CREATE_CHK(hipMalloc(&c->d_l1_resp, ce * 4));
    CREATE_CHK(hipMalloc(&c->d_l1_count, (size_t)B * AFV_MAX_LEVELS * sizeof(int)));
    c->hq_per_frame = afv_harris_queue_per_frame(&g);
    CREATE_CHK(hipMalloc(&c->d_hq, c->hq_per_frame * (size_t)B * sizeof(uint2)));
    CREATE_CHK(hipMalloc(&c->d_hq_n, (size_t)B * sizeof(int)));
    CREATE_CHK(hipMalloc(&c->d_kept_xy, ce * 4));
    CREATE_CHK(hipMalloc(&c->d_kept_resp, ce * 4));
    CREATE_CHK(hipMalloc(&c->d_kept_node, ce * 2));
    CREATE_CHK(hipMalloc(&c->d_cand_count, (size_t)B * AFV_MAX_LEVELS * sizeof(int)));
    CREATE_CHK(hipMalloc(&c->d_sel_count, (size_t)B * AFV_MAX_LEVELS * sizeof(int)));
    CREATE_CHK(hipMalloc(&c->d_sel, (size_t)B * g.sel_per_frame * sizeof(SelPoint)));
    CREATE_CHK(afv_fill(c, c->d_sel_count, 0, (size_t)B * AFV_MAX_LEVELS * sizeof(int)));
    // staging for host-pointer calls
    c->frames_pitch = align_up((size_t)params->max_width, 64);
    c->frames_stride = align_up(c->frames_pitch * (size_t)params->max_height, 256);
    CREATE_CHK(hipMalloc(&c->d_frames, c->frames_stride * (size_t)B + 256));  // frames back to back, one guard at the very end
    c->stage_cap = afv_max_keypoints_per_frame(c);
    {   // counts, keypoints and descriptors of the host-pointer calls in ONE block: [n][kps][desc] - with max_batch 1 (the plugin
        // context) the results of a frame are one contiguous range, i.e. one device-to-host copy
        c->out_kps_off = align_up((size_t)B * sizeof(int), 256);
        c->out_desc_off = c->out_kps_off + align_up((size_t)B * c->stage_cap * sizeof(afv_keypoint), 256);
        c->out_bytes = c->out_desc_off + (size_t)B * c->stage_cap * AFV_DESC_BYTES;
        CREATE_CHK(hipMalloc(&c->d_out_block, c->out_bytes));
        c->d_n = reinterpret_cast<int *>(c->d_out_block);
        c->d_kps = reinterpret_cast<afv_keypoint *>(c->d_out_block + c->out_kps_off);
        c->d_desc = c->d_out_block + c->out_desc_off;
    }
    CREATE_CHK(hipMalloc(&c->d_status, sizeof(int)));
    // quadtree node capacity: alive nodes <= max(quota + 3, 4 * n_ini)
    int M = 64;
    for (int l = 0; l < g.nlevels; ++l) M = std::max(M, g.lv[l].quota + 8);
    M = std::max(M, 4 * 16 + 8);
    M = (int)align_up((size_t)M, 64);
    c->select_M = M;
    if (afv_select_lds_bytes(M) > 160 * 1024) {
        afv_destroy(c);
        return AFV_EUNSUPPORTED;
    }
#undef CREATE_CHK
    // kernels that ask for more dynamic LDS than the default: raised once per context, on its device, checked
    if (hipMalloc(reinterpret_cast<void **>(&c->d_proj_ticket), 256) != hipSuccess || afv_fill(c, c->d_proj_ticket, 0, 256) != hipSuccess) {
        (void)hipGetLastError();
        if (c->d_proj_ticket) (void)hipFree(c->d_proj_ticket);
        c->d_proj_ticket = nullptr;  // the searches then take two launches
    }
    c->proj_wg_lds_max = afv_project_prepare();
    c->frame_lds_max = afv_frame_prepare();
    (void)afv_match_prepare();
    c->select_wide_ok = afv_select_prepare(c->select_M) != 0;
    *out = c;
    return AFV_OK;
}

static void profile_drain(afv_ctx *c) {
    for (int st = 0; st < AFV_NUM_STAGES; ++st) {
        for (size_t i = 0; i + 1 < c->prof_used[st]; i += 2) {
            float ms = 0.f;
            if (hipEventSynchronize(c->prof_ev[st][i + 1]) == hipSuccess &&
                hipEventElapsedTime(&ms, c->prof_ev[st][i], c->prof_ev[st][i + 1]) == hipSuccess) {
                c->prof_ms[st] += ms;
                c->prof_launches[st] += 1;
            }
        }
        c->prof_used[st] = 0;
    }
}

extern "C" int afv_set_split_threshold(afv_ctx *c, int min_frames) {
    if (!c || min_frames < 2) return AFV_EINVAL;
    c->split_min_frames = min_frames;
    return AFV_OK;
}
extern "C" int afv_set_pipeline_chunk(afv_ctx *c, int frames, int chunks_ahead) {
    if (!c || frames < 1 || chunks_ahead < 1 || chunks_ahead > 64) return AFV_EINVAL;
    c->pipe_chunk = frames;
    c->pipe_ahead = chunks_ahead;
    return AFV_OK;
}
extern "C" int afv_set_match_engine(afv_ctx *c, int engine) {
    if (!c || (engine != AFV_MATCH_ENGINE_POPCOUNT && engine != AFV_MATCH_ENGINE_MFMA)) return AFV_EINVAL;
    c->match_engine = engine;
    return AFV_OK;
}

extern "C" int afv_set_l2_chunk_pairs(afv_ctx *c, int pairs) {
    if (!c || pairs < 1 || pairs > 65535) return AFV_EINVAL;
    c->l2_chunk_pairs = pairs;
    return AFV_OK;
}

extern "C" int afv_set_match_resolve(afv_ctx *c, int engine) {
    if (!c || engine < 0 || engine > 2) return AFV_EINVAL;
    c->resolve_engine = engine;
    return AFV_OK;
}

extern "C" int afv_set_small_batch_path(afv_ctx *c, int mode, int max_frames) {
    if (!c || mode < 0 || mode > 2 || max_frames < 0) return AFV_EINVAL;
    c->small_mode = mode;
    if (max_frames > 0) c->small_max_frames = max_frames;
    return AFV_OK;
}

extern "C" int afv_set_split_chunks(afv_ctx *c, int chunks) {
    if (!c || (chunks != 0 && chunks < 2) || chunks > 64) return AFV_EINVAL;
    c->split_chunks = chunks;
    return AFV_OK;
}

extern "C" int afv_profile_enable(afv_ctx *c, int enable) {
    if (!c) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    profile_drain(c);
    c->prof = false;
    c->prof_every = enable > 0 ? enable : 0;
    c->prof_tick_extract = c->prof_tick_match = 0;
    if (enable)
        for (int st = 0; st < AFV_NUM_STAGES; ++st) {
            c->prof_ms[st] = 0.f;
            c->prof_launches[st] = 0;
            c->prof_units[st] = 0;
        }
    return AFV_OK;
}

extern "C" int afv_num_stages(void) { return AFV_NUM_STAGES; }

extern "C" int afv_profile_read(afv_ctx *c, int32_t *launches, float *total_ms, int64_t *units) {
    if (!c || !launches || !total_ms) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    profile_drain(c);
    for (int st = 0; st < AFV_NUM_STAGES; ++st) {
        launches[st] = c->prof_launches[st];
        total_ms[st] = c->prof_ms[st];
        if (units) units[st] = c->prof_units[st];
    }
    return AFV_OK;
}

extern "C" int afv_get_geometry(const afv_ctx *c, afv_geometry *g) {
    if (!c || !g) return AFV_EINVAL;
    const Geo &s = c->geo_valid ? c->geo : c->cap_geo;
    std::memset(g, 0, sizeof(*g));
    g->nlevels = s.nlevels;
    g->width = s.width;
    g->height = s.height;
    for (int l = 0; l < s.nlevels; ++l) {
        g->lw[l] = s.lv[l].w;
        g->lh[l] = s.lv[l].h;
        g->lscale[l] = s.lv[l].scale;
        g->quota[l] = s.lv[l].quota;
        g->cv_quota[l] = s.lv[l].cv_quota;
        g->cand_cap[l] = s.lv[l].cand_cap;
    }
    return AFV_OK;
}

// phase 2 of a brute-force pair call.  The workgroup-wide fixed point is the faster form while every pair has a CU to itself (one pair of
// unrelated frames 23 -> 10 us, of overlapping video frames 61 -> 28 us; 256 overlapping pairs per call: 0.16 against 0.19 ms of
// resolve tail), but it evaluates every live row in every pass: a batch that fills the chip several times over is throughput-bound
// and faster with the one-wavefront walk (10 000 jobs: 3.52 M jobs/s against 3.30 M)
static int resolve_engine_for(const afv_ctx *c, int npairs) {
    return c->resolve_engine == 2 ? (npairs <= c->resolve_wg_max_pairs ? 1 : 0) : c->resolve_engine;
}

// ---- the pipeline ----
// The small-batch ("latency") path: kernels shaped for one or a few frames (Tracking extracts ONE frame per call, Frame.cc:186) -
// the whole pyramid in one launch, ... - same results bit for bit.  afv_set_small_batch_path: 0 = never, 1 = batches of at most
// `small_max_frames` frames (default), 2 = always (parity tests).
static bool small_batch_path(const afv_ctx *c, int nf) {
    return c->small_mode == 2 || (c->small_mode == 1 && nf <= c->small_max_frames);
}

// kernels of one contiguous frame range [f0, f0 + nf) on stream s
// clear_status: this range is the whole call, *d_status is cleared ahead of its kernels (by the one-launch pyramid when there is one)
// d_desc == nullptr: keypoints only (afv_orb_detect); pyramid_only: nothing behind the pyramid (afv_orb_compute describes given keypoints on it)
static void enqueue_range(afv_ctx *c, const FrameSrc &src, int f0, int nf, afv_keypoint *d_kps, uint8_t *d_desc, int cap, int *d_n,
                          int *d_status, hipStream_t s, bool clear_status = false, const DescribeMirror *mirror = nullptr, bool pyramid_only = false) {
    const Geo &g = c->geo;
    {   // work lists are indexed with afv_udiv (exact below AFV_MAX_WORK items): longer ranges go out in pieces
        const int per = std::max(std::max(g.total_tiles, afv_describe_blocks_per_frame(&g)), 1);
        const int max_nf = std::max(1, (AFV_MAX_WORK - 8) / per);
        if (nf > max_nf) {
            if (clear_status && d_status) (void)hipMemsetAsync(d_status, 0, sizeof(int), s);
            for (int b = 0; b < nf; b += max_nf) enqueue_range(c, src, f0 + b, std::min(max_nf, nf - b), d_kps, d_desc, cap, d_n, d_status, s, false, mirror, pyramid_only);
            return;
        }
    }
    int *cnt0 = c->d_cand_count + (size_t)f0 * AFV_MAX_LEVELS;
    if (g.nlevels < 2) {  // no pyramid launch to carry the clears
        (void)hipMemsetAsync(cnt0, 0, (size_t)nf * AFV_MAX_LEVELS * sizeof(int), s);
        (void)hipMemsetAsync(c->d_hq_n + f0, 0, sizeof(int), s);
    }
    const bool small = small_batch_path(c, nf);
    if (clear_status && d_status && !(small && c->pf_ok)) (void)hipMemsetAsync(d_status, 0, sizeof(int), s);
    if (small && c->pf_ok) {
        StageTimer t_(c, AFV_STAGE_PYRAMID, s, nf);
        PyrFuseArgs A = c->pf;
        A.zero_counts = cnt0;
        A.n_zero = nf * AFV_MAX_LEVELS;
        A.zero_one = c->d_hq_n + f0;
        A.zero_two = clear_status ? d_status : nullptr;
        afv_launch_pyramid_fused(&src, c->d_pyr, &A, c->pf_lds, f0, nf, s);
    } else {
        StageTimer t_(c, AFV_STAGE_PYRAMID, s, nf);
        for (int l = 1; l < g.nlevels; ++l) {
            const LevelGeo &S = g.lv[l - 1], &D = g.lv[l];
            const uint8_t *sp = (l == 1) ? src.base : c->d_pyr + S.pyr_off;
            const int spitch = (l == 1) ? src.stride : S.pitch;
            const size_t sframe = (l == 1) ? src.frame_stride : S.pyr_frame_stride;
            afv_launch_resize(sp, S.w, S.h, spitch, sframe, c->d_pyr + D.pyr_off, D.w, D.h, D.pitch, D.pyr_frame_stride,
                              c->d_tab + c->tab_off_x[l], c->d_tab + c->tab_off_y[l], f0, nf, l == 1 ? cnt0 : nullptr, nf * AFV_MAX_LEVELS,
                              l == 1 ? c->d_hq_n + f0 : nullptr, s);
        }
    }
    if (pyramid_only) return;
    {
        StageTimer t_(c, AFV_STAGE_FAST_NMS, s, nf);
        afv_launch_fast_nms(c->d_geo, g.total_tiles, &src, c->d_pyr, c->d_cand_packed, c->d_cand_count, f0, nf, s);
    }
    {
        StageTimer t_(c, AFV_STAGE_HARRIS, s, nf);
        afv_launch_retain_harris(c->d_geo, g.nlevels, &src, c->d_pyr, c->d_cand_packed, c->d_cand_count, c->d_l1, c->d_l1_count,
                                 c->d_l1_resp, c->d_hq + (size_t)f0 * c->hq_per_frame, c->d_hq_n + f0, f0, nf, small ? 1 : 0, s);
    }
    {
        StageTimer t_(c, AFV_STAGE_SELECT, s, nf);
        afv_launch_select(c->d_geo, g.nlevels, c->d_l1, c->d_l1_resp, c->d_l1_count, c->d_kept_xy, c->d_kept_resp,
                          c->d_kept_node, c->d_sel, c->d_sel_count, c->select_M, f0, nf, (small && c->select_wide_ok) ? 1 : 0, s);
    }
    {
        StageTimer t_(c, AFV_STAGE_DESCRIBE, s, nf);
        afv_launch_describe(c->d_geo, afv_describe_blocks_per_frame(&g), &src, c->d_pyr, c->d_sel, c->d_sel_count, d_kps, d_desc, cap, d_n,
                            d_status, f0, nf, mirror, s);
    }
}

// ---- the pipeline ----
static int enqueue_extract(afv_ctx *c, const FrameSrc &src, int nframes, afv_keypoint *d_kps, uint8_t *d_desc, int cap,
                           int *d_n, int *d_status, hipStream_t s) {
    c->prof = c->prof_every && (c->prof_tick_extract++ % (unsigned)c->prof_every) == 0;
    if (nframes >= c->split_min_frames) {
        if (d_status) HIPCHK(c, hipMemsetAsync(d_status, 0, sizeof(int), s));
        // two halves on two streams: the select / describe tail of one half overlaps the FAST kernel of the other
        HIPCHK(c, hipEventRecord(c->ev_fork, s));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        // automatic: chunks of ~85 frames (measured optimum at 640x480: large enough to fill the chip, small enough that the
        // latency-bound kernels of one chunk hide behind the VALU-bound ones of the next), and an EVEN number of them: an odd
        // last chunk runs with nothing beside it (256 frames: 2 chunks 1.55 ms, 3 chunks 1.61; 384: 4 chunks 2.25, 5 chunks 2.33)
        const int K = c->split_chunks ? c->split_chunks : std::min(64, 2 * std::max(1, (int)((float)nframes / 170.0f + 0.45f)));
        for (int k = 0; k < K; ++k) {  // chunk k on stream k % 2
            const int b = (int)((long)nframes * k / K), e = (int)((long)nframes * (k + 1) / K);
            if (e > b) enqueue_range(c, src, b, e - b, d_kps, d_desc, cap, d_n, d_status, (k & 1) ? c->stream2 : s);
        }
        HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0));
    } else {
        enqueue_range(c, src, 0, nframes, d_kps, d_desc, cap, d_n, d_status, s, true);
    }
    HIPCHK(c, hipGetLastError());
    c->last_src = src;
    c->last_nframes = nframes;
    return AFV_OK;
}

extern "C" int afv_orb_extract_batch_device(afv_ctx *c, const uint8_t *d_frames, int nframes, int width, int height,
                                            int stride_bytes, size_t frame_stride_bytes, afv_keypoint *d_kps,
                                            uint8_t *d_desc32, int cap_per_frame, int32_t *d_n_out, int32_t *d_status_out,
                                            void *stream) {
    if (!c || !d_frames || !d_kps || !d_desc32 || !d_n_out) return AFV_EINVAL;
    if (nframes < 1 || nframes > c->p.max_batch || cap_per_frame < 1 || stride_bytes < width) return AFV_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_frames) & 3) || (stride_bytes & 3) || (frame_stride_bytes & 3)) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = set_geometry(c, width, height);
    if (rc) return rc;
    FrameSrc src{d_frames, stride_bytes, frame_stride_bytes};
    return enqueue_extract(c, src, nframes, d_kps, d_desc32, cap_per_frame, d_n_out, d_status_out,
                           stream ? (hipStream_t)stream : c->stream);
}

// true when `p` is page-locked host memory the DMA engines can reach directly (hipHostMalloc / hipHostRegister / torch pin_memory)
static bool is_pinned_host(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

static int ensure_events(afv_ctx *c, size_t n) {
    while (c->pipe_ev.size() < n) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->pipe_ev.push_back(e);
    }
    return AFV_OK;
}

// Host-buffer batch (the vocabulary-builder shape, createVocabulary.cpp:161-174), software-pipelined over chunks of frames:
//   copy lane     ... D2H(k-1), H2D(k+1), D2H(k), H2D(k+2) ...   (one stream: the two directions do not overlap each other)
//   compute       pyramid .. describe of chunk k (alternating over the context's two streams)
// Page-locked caller memory is DMA'd in place (frames in, keypoints / descriptors out); pageable memory goes through the
// context's pinned arena with one CPU memcpy each way.  Results are identical to the serial path (same kernels, same chunks).
extern "C" int afv_orb_extract_batch(afv_ctx *c, const uint8_t *const *frames, int nframes, int width, int height,
                                     int stride_bytes, afv_keypoint *kps, uint8_t *desc32, int cap_per_frame, int *n_out) {
    if (!c || !frames || !kps || !desc32 || !n_out) return AFV_EINVAL;
    if (nframes < 1 || nframes > c->p.max_batch || cap_per_frame < 1 || stride_bytes < width) return AFV_EINVAL;
    for (int f = 0; f < nframes; ++f)
        if (!frames[f]) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = set_geometry(c, width, height);
    if (rc) return rc;
    c->prof = c->prof_every && (c->prof_tick_extract++ % (unsigned)c->prof_every) == 0;
    return guarded(c, [&]() -> int {
        // whatever path leaves this function, no DMA may still be in flight into the caller's buffers or the staging arena
        struct Quiesce {
            afv_ctx *c;
            ~Quiesce() {
                (void)hipStreamSynchronize(c->stream);
                (void)hipStreamSynchronize(c->stream2);
                if (c->stream_copy) (void)hipStreamSynchronize(c->stream_copy);
            }
        } quiesce{c};
        // device staging layout for THIS geometry: frames back to back when the row pitch allows it (one DMA per chunk)
        const size_t pitch = align_up((size_t)width, 64);
        const size_t fstride = align_up(pitch * (size_t)height, 256);
        FrameSrc src{c->d_frames, (int)pitch, fstride};
        const int CH = nframes >= 2 * c->pipe_chunk ? c->pipe_chunk : nframes;  // small batches: one chunk (plugin path: 1 frame)
        const int nchunks = (nframes + CH - 1) / CH;
        rc = ensure_events(c, (size_t)nchunks * 3);
        if (rc) return rc;
        // in-place DMA needs the WHOLE output arrays page-locked: first and last byte are probed (one registration covers a buffer)
        const bool out_direct = is_pinned_host(kps) && is_pinned_host(desc32) &&
                                is_pinned_host(reinterpret_cast<const uint8_t *>(kps + (size_t)nframes * cap_per_frame) - 1) &&
                                is_pinned_host(desc32 + (size_t)nframes * cap_per_frame * AFV_DESC_BYTES - 1);
        const int ocap = std::min(cap_per_frame, c->stage_cap);
        // pinned arena: [pageable frames of the chunks in flight][n][kps][desc] (only what is not DMA'd in place)
        // all or nothing: frames are DMA'd in place only when the first and the last frame of EVERY chunk are page-locked (a probe per
        // frame would cost as much as the extraction of a small batch); a mix of pinned and pageable frames inside a chunk is not supported
        bool all_pinned_in = true;
        for (int f = 0; f < nframes && all_pinned_in; f += CH)
            all_pinned_in = is_pinned_host(frames[f]) && is_pinned_host(frames[std::min(f + CH, nframes) - 1] + (size_t)(height - 1) * stride_bytes + width - 1);
        HostImage arena{c};
        const size_t frame_bytes = (size_t)width * height;
        const size_t in_off = 0, in_bytes = all_pinned_in ? 0 : frame_bytes * (size_t)nframes;
        const size_t n_off = align_up(in_off + in_bytes, 64), n_bytes = (size_t)nframes * sizeof(int);
        const size_t k_off = align_up(n_off + n_bytes, 64), k_bytes = out_direct ? 0 : (size_t)nframes * c->stage_cap * sizeof(afv_keypoint);
        const size_t d_off = align_up(k_off + k_bytes, 64), d_bytes = out_direct ? 0 : (size_t)nframes * c->stage_cap * AFV_DESC_BYTES;
        arena.resize(d_off + d_bytes, false);
        uint8_t *hb = arena.data();
        // one chunk (the single-frame plugin path): everything on the context's stream, no cross-stream hand-offs
        hipStream_t s_copy = c->stream, s_back = c->stream;
        if (nchunks > 1) {
            if (!c->stream_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
            s_copy = s_back = c->stream_copy;
        }
        if (nchunks > 1) {
            HIPCHK(c, hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
            HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            HIPCHK(c, hipStreamWaitEvent(s_copy, c->ev_fork, 0));  // earlier work of this context that still reads d_frames
        }
        // H2D of chunk k (on the copy lane; the single-chunk case runs everything on the context's stream)
        auto upload = [&](int k) -> int {
            const int f0 = k * CH, nf = std::min(CH, nframes - f0);
            bool contiguous = (size_t)stride_bytes == (size_t)width && pitch == (size_t)width && fstride == frame_bytes;
            for (int f = f0 + 1; f < f0 + nf && contiguous; ++f) contiguous = frames[f] == frames[f - 1] + frame_bytes;
            const bool pinned_in = all_pinned_in;
            if (!pinned_in) {  // pageable source: gather the chunk into the pinned arena (tight rows), then DMA from there
                for (int f = f0; f < f0 + nf; ++f)
                    for (int y = 0; y < height; ++y)
                        std::memcpy(hb + in_off + (size_t)f * frame_bytes + (size_t)y * width, frames[f] + (size_t)y * stride_bytes, (size_t)width);
            }
            if (pinned_in && contiguous) {
                HIPCHK(c, hipMemcpyAsync(c->d_frames + (size_t)f0 * fstride, frames[f0], frame_bytes * (size_t)nf, hipMemcpyHostToDevice, s_copy));
            } else if (!pinned_in && pitch == (size_t)width && fstride == frame_bytes) {
                HIPCHK(c, hipMemcpyAsync(c->d_frames + (size_t)f0 * fstride, hb + in_off + (size_t)f0 * frame_bytes, frame_bytes * (size_t)nf,
                                         hipMemcpyHostToDevice, s_copy));
            } else {
                for (int f = f0; f < f0 + nf; ++f) {
                    const uint8_t *sp = pinned_in ? frames[f] : hb + in_off + (size_t)f * frame_bytes;
                    const size_t spitch = pinned_in ? (size_t)stride_bytes : (size_t)width;
                    HIPCHK(c, hipMemcpy2DAsync(c->d_frames + (size_t)f * fstride, pitch, sp, spitch, (size_t)width, (size_t)height,
                                               hipMemcpyHostToDevice, s_copy));
                }
            }
            if (nchunks > 1) HIPCHK(c, hipEventRecord(c->pipe_ev[3 * k], s_copy));
            return AFV_OK;
        };
        // Both directions share ONE copy lane: concurrent H2D + D2H halves each direction on this platform (measured 56 GB/s one
        // way, 24 + 24 GB/s both ways), so the lane carries H2D(k+AHEAD) behind D2H(k): uploads stay AHEAD chunks ahead of the compute.
        const int AHEAD = std::max(c->pipe_ahead, 1);
        for (int k = 0; k < std::min(AHEAD, nchunks); ++k) {
            rc = upload(k);
            if (rc) return rc;
        }
        for (int k = 0; k < nchunks; ++k) {
            const int f0 = k * CH, nf = std::min(CH, nframes - f0);
            hipEvent_t e_in = c->pipe_ev[3 * k], e_done = c->pipe_ev[3 * k + 1], e_out = c->pipe_ev[3 * k + 2];
            // ---- compute ----
            hipStream_t cs = (k & 1) ? c->stream2 : c->stream;
            if (nchunks > 1) HIPCHK(c, hipStreamWaitEvent(cs, e_in, 0));
            enqueue_range(c, src, f0, nf, c->d_kps, c->d_desc, c->stage_cap, c->d_n, c->d_status, cs, nchunks == 1);
            // ---- D2H ----
            if (nchunks > 1) {
                HIPCHK(c, hipEventRecord(e_done, cs));
                HIPCHK(c, hipStreamWaitEvent(s_back, e_done, 0));
            }
            HIPCHK(c, hipMemcpyAsync(hb + n_off + (size_t)f0 * sizeof(int), c->d_n + f0, (size_t)nf * sizeof(int), hipMemcpyDeviceToHost, s_back));
            if (out_direct && cap_per_frame == c->stage_cap) {  // same row length on both sides: two plain DMA transfers
                HIPCHK(c, hipMemcpyAsync(kps + (size_t)f0 * cap_per_frame, c->d_kps + (size_t)f0 * c->stage_cap,
                                         (size_t)nf * c->stage_cap * sizeof(afv_keypoint), hipMemcpyDeviceToHost, s_back));
                HIPCHK(c, hipMemcpyAsync(desc32 + (size_t)f0 * cap_per_frame * AFV_DESC_BYTES, c->d_desc + (size_t)f0 * c->stage_cap * AFV_DESC_BYTES,
                                         (size_t)nf * c->stage_cap * AFV_DESC_BYTES, hipMemcpyDeviceToHost, s_back));
            } else if (out_direct) {
                HIPCHK(c, hipMemcpy2DAsync(kps + (size_t)f0 * cap_per_frame, (size_t)cap_per_frame * sizeof(afv_keypoint),
                                           c->d_kps + (size_t)f0 * c->stage_cap, (size_t)c->stage_cap * sizeof(afv_keypoint),
                                           (size_t)ocap * sizeof(afv_keypoint), (size_t)nf, hipMemcpyDeviceToHost, s_back));
                HIPCHK(c, hipMemcpy2DAsync(desc32 + (size_t)f0 * cap_per_frame * AFV_DESC_BYTES, (size_t)cap_per_frame * AFV_DESC_BYTES,
                                           c->d_desc + (size_t)f0 * c->stage_cap * AFV_DESC_BYTES, (size_t)c->stage_cap * AFV_DESC_BYTES,
                                           (size_t)ocap * AFV_DESC_BYTES, (size_t)nf, hipMemcpyDeviceToHost, s_back));
            } else {
                HIPCHK(c, hipMemcpyAsync(hb + k_off + (size_t)f0 * c->stage_cap * sizeof(afv_keypoint), c->d_kps + (size_t)f0 * c->stage_cap,
                                         (size_t)nf * c->stage_cap * sizeof(afv_keypoint), hipMemcpyDeviceToHost, s_back));
                HIPCHK(c, hipMemcpyAsync(hb + d_off + (size_t)f0 * c->stage_cap * AFV_DESC_BYTES, c->d_desc + (size_t)f0 * c->stage_cap * AFV_DESC_BYTES,
                                         (size_t)nf * c->stage_cap * AFV_DESC_BYTES, hipMemcpyDeviceToHost, s_back));
            }
            HIPCHK(c, hipEventRecord(e_out, s_back));
            if (k + AHEAD < nchunks) {
                rc = upload(k + AHEAD);
                if (rc) return rc;
            }
        }
        HIPCHK(c, hipGetLastError());
        c->last_src = src;
        c->last_nframes = nframes;
        // ---- hand the chunks over as they complete ----
        int result = AFV_OK;
        for (int k = 0; k < nchunks; ++k) {
            const int f0 = k * CH, nf = std::min(CH, nframes - f0);
            HIPCHK(c, hipEventSynchronize(c->pipe_ev[3 * k + 2]));
            const int *hn = reinterpret_cast<const int *>(hb + n_off);
            for (int f = f0; f < f0 + nf; ++f) {
                int n = hn[f];
                if (n > cap_per_frame) {
                    n = cap_per_frame;
                    result = AFV_ECAPACITY;
                }
                n_out[f] = n;
                if (!out_direct) {
                    std::memcpy(kps + (size_t)f * cap_per_frame, hb + k_off + (size_t)f * c->stage_cap * sizeof(afv_keypoint), (size_t)n * sizeof(afv_keypoint));
                    std::memcpy(desc32 + (size_t)f * cap_per_frame * AFV_DESC_BYTES, hb + d_off + (size_t)f * c->stage_cap * AFV_DESC_BYTES,
                                (size_t)n * AFV_DESC_BYTES);
                }
            }
        }
        // the context's streams are idle again (the next call may overwrite d_frames)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        return result;
    });
}

// The plugin call (FeatureExtractor::operator(), FeatureExtractor.cpp:111-129: ONE host image in, host vectors out) without the batch
// pipeline's bookkeeping: no events, no output probes, everything on the context's stream.  A pageable image goes through the pinned
// arena in four strips (the CPU copies strip k + 1 while strip k is on the link); the results come back in one copy when the context
// was created for one frame (counts, keypoints and descriptors are then one contiguous range), else in three.
// `frame` (afv_frame_extract): the describe kernel also writes the frame's device arrays, and k_frame_grid (per-feature arrays + grid) follows
// on the stream before the host waits.
// `keypoints_only` (afv_orb_detect): the last kernel writes keypoints (position, angle, response) and no descriptors.
static int extract_one(afv_ctx *c, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps, uint8_t *desc32, int cap,
                       int *n_out, afv_frame *frame = nullptr, bool keypoints_only = false) {
    struct Quiesce {  // whatever path leaves this function, no DMA may still be in flight into the arena
        afv_ctx *c;
        bool armed = true;
        ~Quiesce() {
            if (armed) (void)hipStreamSynchronize(c->stream);
        }
    } quiesce{c};
    // AFV_TRACE_HOST=1: where the host spends a call (averages over 100 calls on stderr) - a measurement aid, off by default
    static const bool trace = std::getenv("AFV_TRACE_HOST") != nullptr;
    static thread_local double acc[8] = {};
    static thread_local int ncalls = 0;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ts[8] = {};
    if (trace) ts[0] = now();
    const size_t pitch = align_up((size_t)width, 64);
    const size_t fstride = align_up(pitch * (size_t)height, 256);
    FrameSrc src{c->d_frames, (int)pitch, fstride};
    hipStream_t s = c->stream;
    const bool pinned_in = is_pinned_host(gray) && is_pinned_host(gray + (size_t)(height - 1) * stride_bytes + width - 1);
    const size_t frame_bytes = (size_t)width * height;
    // results: the describe kernel writes counts, keypoints and descriptors STRAIGHT into the pinned arena (device-visible host memory:
    // 60 KB of posted writes over the link) - no copy engine between the last kernel and the host (that hop cost ~7 us)
    const size_t kps_off = 256, desc_off = kps_off + align_up((size_t)c->stage_cap * sizeof(afv_keypoint), 256);
    const size_t res_bytes = desc_off + (size_t)c->stage_cap * AFV_DESC_BYTES;
    HostImage arena{c};
    const size_t res_off = align_up(pinned_in ? 0 : frame_bytes, 256);
    arena.resize(res_off + res_bytes, false);
    uint8_t *hb = arena.data();
    const bool zero_copy_out = c->stage_pinned;  // (a pageable arena only exists when pinning failed: results then take the copy engine)
    if (trace) ts[1] = now();
    if (pinned_in) {
        HIPCHK(c, hipMemcpy2DAsync(c->d_frames, pitch, gray, (size_t)stride_bytes, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s));
    } else {
        // one copy into the arena, one DMA: strips (CPU copy of strip k + 1 beside the DMA of strip k) lose more to the per-transfer
        // latency of the copy engine (~6 us each, serialised) than the overlap wins (measured: 4 strips +16 us)
        if ((size_t)stride_bytes == (size_t)width) {
            std::memcpy(hb, gray, frame_bytes);
        } else {
            for (int y = 0; y < height; ++y) std::memcpy(hb + (size_t)y * width, gray + (size_t)y * stride_bytes, (size_t)width);
        }
        if (pitch == (size_t)width) {
            HIPCHK(c, hipMemcpyAsync(c->d_frames, hb, frame_bytes, hipMemcpyHostToDevice, s));
        } else {
            HIPCHK(c, hipMemcpy2DAsync(c->d_frames, pitch, hb, (size_t)width, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s));
        }
    }
    uint8_t *hres = hb + res_off;
    if (trace) ts[2] = now();
    DescribeMirror mir{nullptr, nullptr, nullptr};
    if (frame) mir = DescribeMirror{frame->d_kps, frame->d_desc, frame->d_n};
    // (a frame smaller than the staging capacity: slots beyond frame->cap would be written past its arrays - afv_frame_create sizes frames
    // for the context's capacity, afv_frame_extract checks)
    if (zero_copy_out) {
        enqueue_range(c, src, 0, 1, reinterpret_cast<afv_keypoint *>(hres + kps_off), keypoints_only ? nullptr : hres + desc_off, c->stage_cap,
                      reinterpret_cast<int *>(hres), c->d_status, s, true, frame ? &mir : nullptr);
    } else {
        enqueue_range(c, src, 0, 1, c->d_kps, keypoints_only ? nullptr : c->d_desc, c->stage_cap, c->d_n, c->d_status, s, true, frame ? &mir : nullptr);
    }
    // a resident frame: the host vectors are complete when the describe kernel is (it writes them straight into the pinned arena);
    // k_frame_grid, which only feeds later device-side consumers on the same stream, runs on while the call returns
    bool wait_event = false;
    if (frame && zero_copy_out) {
        wait_event = hipEventRecord(c->ev_fork, s) == hipSuccess;
        if (!wait_event) (void)hipGetLastError();
    }
    if (frame) {
        const int rc_grid = afv_frame_after_extract(frame, s);  // (a failed launch must not leave has_grid set over the zeroed memset: ADVICE r5)
        if (rc_grid) return rc_grid;
    }
    HIPCHK(c, hipGetLastError());
    if (trace) ts[3] = now();
    if (!zero_copy_out) {
        HIPCHK(c, hipMemcpyAsync(hres, c->d_n, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(hres + kps_off, c->d_kps, (size_t)c->stage_cap * sizeof(afv_keypoint), hipMemcpyDeviceToHost, s));
        if (!keypoints_only) HIPCHK(c, hipMemcpyAsync(hres + desc_off, c->d_desc, (size_t)c->stage_cap * AFV_DESC_BYTES, hipMemcpyDeviceToHost, s));
    }
    const uint8_t *h_kps = hres + kps_off, *h_desc = hres + desc_off;
    c->last_src = src;
    c->last_nframes = 1;
    if (trace) ts[4] = now();
    if (wait_event) HIPCHK(c, hipEventSynchronize(c->ev_fork));
    else HIPCHK(c, hipStreamSynchronize(s));
    if (trace) ts[5] = now();
    quiesce.armed = false;
    int n = *reinterpret_cast<const int *>(hres), result = AFV_OK;
    if (n > cap) {
        n = cap;
        result = AFV_ECAPACITY;
    }
    n = std::max(n, 0);
    if (frame) frame->n = std::min(std::max(*reinterpret_cast<const int *>(hres), 0), frame->cap);
    if (n_out) *n_out = n;
    if (kps) std::memcpy(kps, h_kps, (size_t)n * sizeof(afv_keypoint));
    if (desc32 && !keypoints_only) std::memcpy(desc32, h_desc, (size_t)n * AFV_DESC_BYTES);
    if (trace) {
        ts[6] = now();
        for (int i = 0; i < 6; ++i) acc[i] += ts[i + 1] - ts[i];
        if (++ncalls % 100 == 0) {
            fprintf(stderr, "afv_orb_extract (us, mean of 100): probes + arena %.1f | upload enqueue %.1f | kernel launches %.1f | download enqueue %.1f | wait %.1f | hand-over %.1f\n",
                    acc[0] / 100, acc[1] / 100, acc[2] / 100, acc[3] / 100, acc[4] / 100, acc[5] / 100);
            for (double &v : acc) v = 0;
        }
    }
    return result;
}

extern "C" int afv_orb_extract(afv_ctx *c, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps,
                               uint8_t *desc32, int cap, int *n_out) {
    if (!c || !gray || !kps || !desc32 || !n_out) return AFV_EINVAL;
    if (cap < 1 || stride_bytes < width) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = set_geometry(c, width, height);
    if (rc) return rc;
    c->prof = c->prof_every && (c->prof_tick_extract++ % (unsigned)c->prof_every) == 0;
    return guarded(c, [&]() -> int { return extract_one(c, gray, width, height, stride_bytes, kps, desc32, cap, n_out); });
}

// ---- the two halves of the plugin call as entry points of their own (FeatureExtractor.h:123-124, Feature_orb32.cpp:26-53) ----
// detectKeypoints + filterKeypoints (E1-E7): what afv_orb_extract returns as keypoints, before any descriptor is computed
extern "C" int afv_orb_detect(afv_ctx *c, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps, int cap, int *n_out) {
    if (!c || !gray || !kps || !n_out) return AFV_EINVAL;
    if (cap < 1 || stride_bytes < width) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = set_geometry(c, width, height);
    if (rc) return rc;
    c->prof = c->prof_every && (c->prof_tick_extract++ % (unsigned)c->prof_every) == 0;
    return guarded(c, [&]() -> int { return extract_one(c, gray, width, height, stride_bytes, kps, nullptr, cap, n_out, nullptr, true); });
}

// computeDescriptors = cv::ORB::compute at caller-given keypoints (E8-E10): for every keypoint the rBRIEF descriptor at
// cvRound(pt / scale(octave)) of its OWN octave and at its OWN angle, on the blurred level with the unblurred apron - the pyramid is
// rebuilt from the image as cv::ORB::compute rebuilds levels 0 .. max octave (here: all levels, in one launch).  Keypoints are taken as
// they are (orb.cpp runByImageBorder with edgeThreshold 0 removes nothing); a keypoint whose octave is not a level of this context or
// whose centre does not lie on its level image is refused (AFV_EINVAL), nothing is described then.
extern "C" int afv_orb_compute(afv_ctx *c, const uint8_t *gray, int width, int height, int stride_bytes, const afv_keypoint *kps, int n, uint8_t *desc32) {
    if (!c || !gray || n < 0 || (n > 0 && (!kps || !desc32)) || stride_bytes < width) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = set_geometry(c, width, height);
    if (rc) return rc;
    const Geo &g = c->geo;
    for (int i = 0; i < n; ++i) {
        const afv_keypoint &k = kps[i];
        if (k.octave < 0 || k.octave >= g.nlevels) return AFV_EINVAL;
        const LevelGeo &L = g.lv[k.octave];
        const float cx = rintf(k.x * L.inv_scale), cy = rintf(k.y * L.inv_scale);
        if (!(cx >= 0.f && cx <= (float)L.w && cy >= 0.f && cy <= (float)L.h)) return AFV_EINVAL;  // (also refuses NaN)
    }
    if (n == 0) return AFV_OK;
    return guarded(c, [&]() -> int {
        struct Quiesce {
            afv_ctx *c;
            ~Quiesce() { (void)hipStreamSynchronize(c->stream); }
        } quiesce{c};
        hipStream_t s = c->stream;
        const size_t pitch = align_up((size_t)width, 64), fstride = align_up(pitch * (size_t)height, 256), frame_bytes = (size_t)width * height;
        FrameSrc src{c->d_frames, (int)pitch, fstride};
        // arena: [image][keypoints][descriptors]; device side of the keypoints / descriptors: the matcher staging buffer
        const size_t k_off = align_up(frame_bytes, 256), k_bytes = (size_t)n * sizeof(afv_keypoint);
        const size_t d_off = align_up(k_off + k_bytes, 256), d_bytes = (size_t)n * AFV_DESC_BYTES;
        HostImage arena{c};
        arena.resize(d_off + d_bytes, false);
        uint8_t *hb = arena.data();
        int rc2 = ensure_match_buffer(c, d_off + d_bytes);
        if (rc2) return rc2;
        if ((size_t)stride_bytes == (size_t)width) std::memcpy(hb, gray, frame_bytes);
        else
            for (int y = 0; y < height; ++y) std::memcpy(hb + (size_t)y * width, gray + (size_t)y * stride_bytes, (size_t)width);
        std::memcpy(hb + k_off, kps, k_bytes);
        if (pitch == (size_t)width) HIPCHK(c, hipMemcpyAsync(c->d_frames, hb, frame_bytes, hipMemcpyHostToDevice, s));
        else HIPCHK(c, hipMemcpy2DAsync(c->d_frames, pitch, hb, (size_t)width, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_match + k_off, hb + k_off, k_bytes, hipMemcpyHostToDevice, s));
        enqueue_range(c, src, 0, 1, nullptr, nullptr, 0, nullptr, nullptr, s, false, nullptr, true);
        afv_launch_describe_given(c->d_geo, &src, c->d_pyr, reinterpret_cast<const afv_keypoint *>(c->d_match + k_off), n, c->d_match + d_off, 0, s);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hb + d_off, c->d_match + d_off, d_bytes, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        std::memcpy(desc32, hb + d_off, d_bytes);
        c->last_src = src;
        c->last_nframes = 1;
        return AFV_OK;
    });
}

// afv_frame_extract (afv_frame.hip): the same call with a frame attached
int afv_extract_into_frame(afv_ctx *c, afv_frame *f, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps,
                           uint8_t *desc32, int cap, int *n_out) {
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = set_geometry(c, width, height);
    if (rc) return rc;
    if (f->cap < c->stage_cap) return AFV_ECAPACITY;  // the describe kernel writes up to stage_cap slots of the mirror
    c->prof = c->prof_every && (c->prof_tick_extract++ % (unsigned)c->prof_every) == 0;
    return guarded(c, [&]() -> int { return extract_one(c, gray, width, height, stride_bytes, kps, desc32, cap, n_out, f); });
}

// E12 (FeatureExtractor.cpp:132-172, settings FeatureExtractor.cpp:52-55)
float afv_size_of_octave(const afv_ctx *c, int octave) {
    const float scale_factor_orb = 1.2f;
    const float max_size0 = powf(scale_factor_orb, float(8 - 1.0));
    const float max_size = max_size0, min_size = 1.0f;
    const float s = powf(c->p.scale_factor, float(octave));  // GetKeypointSize Feature_orb32.cpp:59-61
    float norm = max_size;
    if (max_size > min_size) norm = 1.0f + (s - min_size) * (max_size0 - 1.0f) / (max_size - min_size);
    return norm;
}
extern "C" int afv_orb_size_sigma(const afv_ctx *c, const afv_keypoint *kps, int n, float *size, float *sigma2, float *inf) {
    if (!c || (n > 0 && (!kps || !size || !sigma2 || !inf)) || n < 0) return AFV_EINVAL;
    for (int i = 0; i < n; ++i) {
        const float norm = afv_size_of_octave(c, kps[i].octave);
        size[i] = norm;
        const float s2 = norm * norm;
        sigma2[i] = s2;
        inf[i] = 1.0f / s2;
    }
    return AFV_OK;
}

extern "C" int afv_hamming256(const uint8_t *a, const uint8_t *b) {
    int d = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        d += __builtin_popcount(x ^ y);
    }
    return d;
}

// ---- debug getters ----
static const uint8_t *level_ptr(const afv_ctx *c, int frame, int level, int *pitch) {
    const LevelGeo &L = c->geo.lv[level];
    if (level == 0) {
        *pitch = c->last_src.stride;
        return c->last_src.base + (size_t)frame * c->last_src.frame_stride;
    }
    *pitch = L.pitch;
    return c->d_pyr + L.pyr_off + (size_t)frame * L.pyr_frame_stride;
}

extern "C" int afv_debug_get_level(afv_ctx *c, int frame, int level, uint8_t *out) {
    if (!c || !out || !c->geo_valid || frame < 0 || frame >= c->last_nframes || level < 0 || level >= c->geo.nlevels) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    int pitch;
    const uint8_t *p = level_ptr(c, frame, level, &pitch);
    const LevelGeo &L = c->geo.lv[level];
    HIPCHK(c, hipMemcpy2D(out, (size_t)L.w, p, (size_t)pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
    return AFV_OK;
}

extern "C" int afv_debug_get_candidates(afv_ctx *c, int frame, int level, uint32_t *packed, float *response, int cap, int *n_out) {
    if (!c || !packed || !response || !n_out || !c->geo_valid || frame < 0 || frame >= c->last_nframes || level < 0 ||
        level >= c->geo.nlevels)
        return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_cand_count + frame * AFV_MAX_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost));
    const LevelGeo &L = c->geo.lv[level];
    n = std::min(n, L.cand_cap);
    *n_out = n;
    const int m = std::min(n, cap);
    const size_t base = L.cand_off + (size_t)frame * L.cand_frame_stride;
    if (m > 0) {
        HIPCHK(c, hipMemcpy(packed, c->d_cand_packed + base, (size_t)m * 4, hipMemcpyDeviceToHost));
        // the Harris response exists for the candidates that survived retainBest on the score (the level's l1 list); 0 for the rest
        int n1 = 0;
        HIPCHK(c, hipMemcpy(&n1, c->d_l1_count + frame * AFV_MAX_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost));
        n1 = std::min(std::max(n1, 0), L.cand_cap);
        std::vector<uint32_t> l1(std::max(n1, 1));
        std::vector<float> r1(std::max(n1, 1));
        if (n1 > 0) {
            HIPCHK(c, hipMemcpy(l1.data(), c->d_l1 + base, (size_t)n1 * 4, hipMemcpyDeviceToHost));
            HIPCHK(c, hipMemcpy(r1.data(), c->d_l1_resp + base, (size_t)n1 * 4, hipMemcpyDeviceToHost));
        }
        std::unordered_map<uint32_t, float> by_pos;
        by_pos.reserve((size_t)n1 * 2);
        for (int i = 0; i < n1; ++i) by_pos[l1[i] & 0x00ffffffu] = r1[i];
        for (int i = 0; i < m; ++i) {
            auto it = by_pos.find(packed[i] & 0x00ffffffu);
            response[i] = it == by_pos.end() ? 0.f : it->second;
        }
    }
    return n > cap ? AFV_ECAPACITY : AFV_OK;
}

extern "C" int afv_debug_get_selected(afv_ctx *c, int frame, int level, int32_t *x, int32_t *y, float *response, int cap, int *n_out) {
    if (!c || !x || !y || !response || !n_out || !c->geo_valid || frame < 0 || frame >= c->last_nframes || level < 0 ||
        level >= c->geo.nlevels)
        return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_sel_count + frame * AFV_MAX_LEVELS + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    const int m = std::min(n, cap);
    std::vector<SelPoint> tmp(std::max(m, 1));
    if (m > 0)
        HIPCHK(c, hipMemcpy(tmp.data(), c->d_sel + (size_t)frame * c->geo.sel_per_frame + c->geo.lv[level].sel_base,
                            (size_t)m * sizeof(SelPoint), hipMemcpyDeviceToHost));
    for (int i = 0; i < m; ++i) {
        x[i] = tmp[i].x;
        y[i] = tmp[i].y;
        response[i] = tmp[i].response;
    }
    return n > cap ? AFV_ECAPACITY : AFV_OK;
}

extern "C" int afv_debug_blur_level(afv_ctx *c, int frame, int level, uint8_t *out) {
    if (!c || !out || !c->geo_valid || frame < 0 || frame >= c->last_nframes || level < 0 || level >= c->geo.nlevels) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    int pitch;
    const uint8_t *p = level_ptr(c, frame, level, &pitch);
    const LevelGeo &L = c->geo.lv[level];
    uint8_t *d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_out, (size_t)L.w * L.h));
    afv_launch_blur_level(p, L.w, L.h, pitch, d_out, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)L.w * L.h, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    HIPCHK(c, e);
    return AFV_OK;
}


// merge-join of the two FeatureVectors (FeatureMatcher.cc:205-276): list of (range1, range2) for shared node ids
void afv_shared_segments(const afv_match_job &j, std::vector<Seg> &segs) {
    segs.clear();
    if (j.nnodes1 == 0 || j.nnodes2 == 0) {
        segs.push_back(Seg{0, j.n1, 0, j.n2});
        return;
    }
    int a = 0, b = 0;
    while (a < j.nnodes1 && b < j.nnodes2) {
        if (j.node_id1[a] == j.node_id2[b]) {
            segs.push_back(Seg{j.seg_ptr1[a], j.seg_ptr1[a + 1] - j.seg_ptr1[a], j.seg_ptr2[b], j.seg_ptr2[b + 1] - j.seg_ptr2[b]});
            ++a;
            ++b;
        } else if (j.node_id1[a] < j.node_id2[b]) {
            ++a;
        } else {
            ++b;
        }
    }
}

struct JobOffsets {
    size_t d1, d2, segs, idx1, idx2, v1, v2, a1, a2, out, nm;
    int nseg, words, nout, fdim;
    bool has_idx, has_v1, has_v2, has_ang;
};

static int validate_job(const afv_match_job &j, bool need_angles) {
    if (j.n1 < 0 || j.n2 < 0 || j.n1 > AFV_MAX_SIDE || j.n2 > AFV_MAX_SIDE) return AFV_EINVAL;
    if ((j.n1 > 0 && !j.desc1) || (j.n2 > 0 && !j.desc2)) return AFV_EINVAL;
    if (j.mode & AFV_MATCH_FLOAT32) {  // float rows: desc_bytes = 4 * dim, rows 16-byte aligned in the staging blob
        if (j.desc_bytes < 16 || j.desc_bytes > 4096 || (j.desc_bytes & 15)) return AFV_EINVAL;
    } else if (j.desc_bytes < 1 || j.desc_bytes > 64) {
        return AFV_EINVAL;
    }
    if (j.nnodes1 < 0 || j.nnodes2 < 0) return AFV_EINVAL;
    if (j.nnodes1 > 0 && (!j.node_id1 || !j.seg_ptr1 || !j.seg_idx1)) return AFV_EINVAL;
    if (j.nnodes2 > 0 && (!j.node_id2 || !j.seg_ptr2 || !j.seg_idx2)) return AFV_EINVAL;
    if (need_angles && j.check_orientation && (!j.angle1 || !j.angle2)) return AFV_EINVAL;
    // the CSR FeatureVectors size host copies and index descriptors on the device: check them here, O(n)
    const int32_t *ptrs[2] = {j.seg_ptr1, j.seg_ptr2}, *idxs[2] = {j.seg_idx1, j.seg_idx2}, *ids[2] = {j.node_id1, j.node_id2};
    const int nn[2] = {j.nnodes1, j.nnodes2}, nf[2] = {j.n1, j.n2};
    for (int s = 0; s < 2; ++s) {
        if (nn[s] == 0) continue;
        if (ptrs[s][0] != 0) return AFV_EINVAL;
        for (int i = 0; i < nn[s]; ++i) {
            if (ptrs[s][i + 1] < ptrs[s][i]) return AFV_EINVAL;
            if (i > 0 && ids[s][i] <= ids[s][i - 1]) return AFV_EINVAL;  // std::map order: strictly ascending node ids
        }
        const int total = ptrs[s][nn[s]];
        if (total > nf[s]) return AFV_EINVAL;  // a feature sits in exactly one node
        for (int i = 0; i < total; ++i)
            if (idxs[s][i] < 0 || idxs[s][i] >= nf[s]) return AFV_EINVAL;
    }
    return AFV_OK;
}

static void stage_job(Blob &b, const afv_match_job &j, bool tri, JobOffsets &o) {
    o.fdim = (j.mode & AFV_MATCH_FLOAT32) ? j.desc_bytes / 4 : 0;
    const int kind = j.mode & ~AFV_MATCH_FLOAT32;
    o.words = o.fdim ? o.fdim : (j.desc_bytes <= 32 ? 8 : 16);
    o.d1 = o.fdim ? b.put(j.desc1, (size_t)j.n1 * j.desc_bytes) : put_desc(b, j.desc1, j.n1, j.desc_bytes, o.words);
    o.d2 = o.fdim ? b.put(j.desc2, (size_t)j.n2 * j.desc_bytes) : put_desc(b, j.desc2, j.n2, j.desc_bytes, o.words);
    std::vector<Seg> segs;
    afv_shared_segments(j, segs);
    o.nseg = (int)segs.size();
    o.segs = b.put(segs.data(), segs.size() * sizeof(Seg));
    o.has_idx = j.nnodes1 > 0 && j.nnodes2 > 0;
    if (o.has_idx) {
        o.idx1 = b.put(j.seg_idx1, (size_t)j.seg_ptr1[j.nnodes1] * 4);
        o.idx2 = b.put(j.seg_idx2, (size_t)j.seg_ptr2[j.nnodes2] * 4);
    }
    o.has_v1 = j.valid1 != nullptr;
    o.has_v2 = j.valid2 != nullptr && (tri || kind != AFV_MATCH_KF_FRAME);
    if (o.has_v1) o.v1 = b.put(j.valid1, (size_t)j.n1);
    if (o.has_v2) o.v2 = b.put(j.valid2, (size_t)j.n2);
    o.has_ang = !tri && j.check_orientation;
    if (o.has_ang) {
        o.a1 = b.put(j.angle1, (size_t)j.n1 * 4);
        o.a2 = b.put(j.angle2, (size_t)j.n2 * 4);
    }
    o.nout = (!tri && kind == AFV_MATCH_KF_FRAME) ? j.n2 : j.n1;
}

static void fill_dev_job(DevMatchJob &d, const afv_match_job &j, const JobOffsets &o, uint8_t *base, bool tri) {
    d.d1 = reinterpret_cast<const uint32_t *>(base + o.d1);
    d.d2 = reinterpret_cast<const uint32_t *>(base + o.d2);
    d.n1 = j.n1;
    d.n2 = j.n2;
    d.words = o.fdim ? 0 : o.words;
    d.fdim = o.fdim;
    d.segs = reinterpret_cast<const Seg *>(base + o.segs);
    d.nseg = o.nseg;
    d.idx1 = o.has_idx ? reinterpret_cast<const int *>(base + o.idx1) : nullptr;
    d.idx2 = o.has_idx ? reinterpret_cast<const int *>(base + o.idx2) : nullptr;
    d.valid1 = o.has_v1 ? base + o.v1 : nullptr;
    d.valid2 = o.has_v2 ? base + o.v2 : nullptr;
    d.ang1 = o.has_ang ? reinterpret_cast<const float *>(base + o.a1) : nullptr;
    d.ang2 = o.has_ang ? reinterpret_cast<const float *>(base + o.a2) : nullptr;
    d.ang_stride = 1;
    d.th = j.th_low;
    d.ratio = j.nnratio;
    d.check_ori = tri ? 0 : (j.check_orientation != 0);
    d.mode = tri ? AFV_MATCH_KF_KF : (j.mode & ~AFV_MATCH_FLOAT32);
    d.out = reinterpret_cast<int *>(base + o.out);
    d.nmatches = reinterpret_cast<int *>(base + o.nm);
}

static int afv_match_bow_impl(afv_ctx *c, const afv_match_job *jobs, int njobs, int32_t *out, int32_t *nmatches) {
    if (!c || !jobs || njobs < 1 || !out || !nmatches) return AFV_EINVAL;
    for (int i = 0; i < njobs; ++i) {
        const int rc = validate_job(jobs[i], true);
        if (rc) return rc;
        const int kind = jobs[i].mode & ~AFV_MATCH_FLOAT32;
        if (kind != AFV_MATCH_KF_KF && kind != AFV_MATCH_KF_FRAME) return AFV_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->device));
    {
        // plain brute-force KF-KF jobs over 32-byte descriptors take the two-phase path of the device pipeline
        // (parallel top-4 + ordered resolve): stage them as a descriptor table of 2 sets per job
        bool eligible = true;
        int cap = 1;
        for (int i = 0; i < njobs; ++i) {
            const afv_match_job &j = jobs[i];
            eligible = eligible && (j.nnodes1 == 0 || j.nnodes2 == 0) && j.mode == AFV_MATCH_KF_KF && j.desc_bytes == 32 &&
                       !j.valid1 && !j.valid2 && j.n1 <= 4096 && j.n2 <= 4096;
            cap = std::max(cap, std::max(j.n1, j.n2));
        }
        if (eligible) {
            Blob b(c);
            const int nsets = 2 * njobs;
            const size_t desc_off = b.reserve((size_t)nsets * cap * 32);
            const size_t n_off = b.reserve((size_t)nsets * 4);
            const size_t pa_off = b.reserve((size_t)njobs * 4), pb_off = b.reserve((size_t)njobs * 4);
            bool any_ori = false;
            for (int i = 0; i < njobs; ++i) any_ori = any_ori || jobs[i].check_orientation;
            const size_t ang_off = any_ori ? b.reserve((size_t)nsets * cap * sizeof(float)) : 0;
            for (int i = 0; i < njobs; ++i) {
                const afv_match_job &j = jobs[i];
                if (j.n1) std::memcpy(b.h.data() + desc_off + (size_t)(2 * i) * cap * 32, j.desc1, (size_t)j.n1 * 32);
                if (j.n2) std::memcpy(b.h.data() + desc_off + (size_t)(2 * i + 1) * cap * 32, j.desc2, (size_t)j.n2 * 32);
                int32_t *n = reinterpret_cast<int32_t *>(b.h.data() + n_off);
                n[2 * i] = j.n1;
                n[2 * i + 1] = j.n2;
                reinterpret_cast<int32_t *>(b.h.data() + pa_off)[i] = 2 * i;
                reinterpret_cast<int32_t *>(b.h.data() + pb_off)[i] = 2 * i + 1;
                if (any_ori && j.check_orientation) {
                    float *a1 = reinterpret_cast<float *>(b.h.data() + ang_off) + (size_t)(2 * i) * cap;
                    std::memcpy(a1, j.angle1, (size_t)j.n1 * sizeof(float));
                    std::memcpy(a1 + cap, j.angle2, (size_t)j.n2 * sizeof(float));
                }
            }
            const size_t match_off = b.reserve((size_t)njobs * cap * 4), nm_off = b.reserve((size_t)njobs * 4);
            const int nslices = small_batch_path(c, njobs) ? afv_match_topk_slices(cap, c->match_engine, ((cap + 63) / 64 + 1) / 2) : 1;
            const size_t topk_off = b.reserve_scratch((size_t)njobs * cap * 32);
            {
                const int rc_ = ensure_slice_scratch(c, njobs, cap, nslices);
                if (rc_) return rc_;
            }
            int rc = ensure_match_buffer(c, b.h.size());
            if (rc) return rc;
            HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), match_off, hipMemcpyHostToDevice, c->stream));  // inputs only
            // jobs may differ in mbCheckOrientation / thresholds: launch runs of identical settings
            int i0 = 0;
            while (i0 < njobs) {
                int i1 = i0 + 1;
                while (i1 < njobs && jobs[i1].th_low == jobs[i0].th_low && jobs[i1].nnratio == jobs[i0].nnratio &&
                       (jobs[i1].check_orientation != 0) == (jobs[i0].check_orientation != 0))
                    ++i1;
                const float *angp = any_ori ? reinterpret_cast<const float *>(c->d_match + ang_off) : nullptr;
                const int *np_ = reinterpret_cast<const int *>(c->d_match + n_off);
                const int *pa_ = reinterpret_cast<const int *>(c->d_match + pa_off), *pb_ = reinterpret_cast<const int *>(c->d_match + pb_off);
                afv_launch_match_topk(c->d_match + desc_off, np_, cap, pa_, pb_, i1 - i0, c->d_match + topk_off, i0, c->match_engine, nslices, c->d_slice, c->d_tickets, c->stream);
                afv_launch_match_resolve(c->d_match + desc_off, angp, 1, np_, cap, pa_, pb_, i1 - i0, jobs[i0].th_low, jobs[i0].nnratio,
                                         jobs[i0].check_orientation != 0, reinterpret_cast<int *>(c->d_match + match_off),
                                         reinterpret_cast<int *>(c->d_match + nm_off), c->d_match + topk_off, i0, resolve_engine_for(c, njobs), c->stream);
                i0 = i1;
            }
            HIPCHK(c, hipGetLastError());
            size_t acc = 0;
            for (int i = 0; i < njobs; ++i) {
                HIPCHK(c, b.fetch(out + acc, match_off + (size_t)i * cap * 4, (size_t)jobs[i].n1 * 4, c->stream));
                acc += (size_t)jobs[i].n1;
            }
            HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)njobs * 4, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            b.finish();
            return afv_check_resolve_guard(c, nmatches, njobs);
        }
    }
    Blob b(c);
    std::vector<JobOffsets> offs(njobs);
    for (int i = 0; i < njobs; ++i) stage_job(b, jobs[i], false, offs[i]);
    size_t total_out = 0;
    for (int i = 0; i < njobs; ++i) total_out += (size_t)offs[i].nout;
    // BoW-guided jobs (more than one shared node) run one wavefront per node; a single-segment job (brute force) keeps
    // the ordered workgroup-per-job kernel
    bool per_node = false;
    for (int i = 0; i < njobs; ++i) per_node = per_node || offs[i].nseg > 1;
    const size_t out_off = b.reserve(std::max<size_t>(total_out, 1) * 4);
    const size_t nm_off = b.reserve((size_t)njobs * 4);
    const size_t jobs_off = b.reserve((size_t)njobs * sizeof(DevMatchJob));
    std::vector<SegTask> tasks;
    std::vector<int> bin_off(njobs, 0);
    size_t tasks_off = 0, hist_off = 0, bins_off = 0, binoff_off = 0;
    bool any_ori = false;
    if (per_node) {
        size_t acc = 0;
        for (int i = 0; i < njobs; ++i) {
            for (int sgi = 0; sgi < offs[i].nseg; ++sgi) tasks.push_back(SegTask{i, sgi});
            bin_off[i] = (int)acc;
            acc += (size_t)offs[i].nout;
            any_ori = any_ori || jobs[i].check_orientation;
        }
        tasks_off = b.put(tasks.data(), tasks.size() * sizeof(SegTask));
        hist_off = b.reserve((size_t)njobs * 32 * 4);
        bins_off = b.reserve(std::max<size_t>(acc, 1));
        binoff_off = b.put(bin_off.data(), (size_t)njobs * 4);
    }
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    size_t acc = 0;
    for (int i = 0; i < njobs; ++i) {
        offs[i].out = out_off + acc * 4;
        offs[i].nm = nm_off + (size_t)i * 4;
        acc += (size_t)offs[i].nout;
        fill_dev_job(reinterpret_cast<DevMatchJob *>(b.h.data() + jobs_off)[i], jobs[i], offs[i], c->d_match, false);
    }
    if (per_node) {  // the per-node kernel accumulates: outputs start at -1, counters at 0 (the blob is zero-filled)
        int32_t *o = reinterpret_cast<int32_t *>(b.h.data() + out_off);
        for (size_t i = 0; i < total_out; ++i) o[i] = -1;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), b.h.size(), hipMemcpyHostToDevice, c->stream));
    if (per_node)
        afv_launch_match_bow_seg(reinterpret_cast<const DevMatchJob *>(c->d_match + jobs_off), njobs, c->d_match + tasks_off,
                                 (int)tasks.size(), reinterpret_cast<int *>(c->d_match + hist_off), c->d_match + bins_off,
                                 reinterpret_cast<const int *>(c->d_match + binoff_off), any_ori ? 1 : 0, c->stream);
    else
        afv_launch_match_bow(reinterpret_cast<const DevMatchJob *>(c->d_match + jobs_off), njobs, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, b.fetch(out, out_off, total_out * 4, c->stream));
    HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)njobs * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}
extern "C" int afv_match_bow(afv_ctx *c, const afv_match_job *jobs, int njobs, int32_t *out, int32_t *nmatches) {
    return guarded(c, [&] { return afv_match_bow_impl(c, jobs, njobs, out, nmatches); });
}

static int afv_match_triangulation_impl(afv_ctx *c, const afv_tri_job *caller_jobs, int njobs, int32_t *match12, int32_t *nmatches) {
    if (!c || !caller_jobs || njobs < 1 || !match12 || !nmatches) return AFV_EINVAL;
    std::vector<afv_tri_job> loaded;
    if (!afv_load_jobs(caller_jobs, njobs, offsetof(afv_tri_job, u_right1), loaded)) return AFV_EINVAL;
    const afv_tri_job *jobs = loaded.data();
    for (int i = 0; i < njobs; ++i) {
        const int rc = validate_job(jobs[i].bow, false);
        if (rc) return rc;
        const afv_tri_job &t = jobs[i];
        if ((t.bow.n1 > 0 && (!t.x1 || !t.y1)) || (t.bow.n2 > 0 && (!t.x2 || !t.y2 || !t.sigma2_2))) return AFV_EINVAL;
        if (t.only_stereo != 0 && t.only_stereo != 1) return AFV_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->device));
    Blob b(c);
    std::vector<JobOffsets> offs(njobs);
    std::vector<size_t> geo_off(njobs * 5), rowseg_off(njobs), ur_off(njobs * 2);
    for (int i = 0; i < njobs; ++i) {
        stage_job(b, jobs[i].bow, true, offs[i]);
        const afv_tri_job &t = jobs[i];
        geo_off[5 * i + 0] = b.put(t.x1, (size_t)t.bow.n1 * 4);
        geo_off[5 * i + 1] = b.put(t.y1, (size_t)t.bow.n1 * 4);
        geo_off[5 * i + 2] = b.put(t.x2, (size_t)t.bow.n2 * 4);
        geo_off[5 * i + 3] = b.put(t.y2, (size_t)t.bow.n2 * 4);
        geo_off[5 * i + 4] = b.put(t.sigma2_2, (size_t)t.bow.n2 * 4);
        ur_off[2 * i + 0] = t.u_right1 ? b.put(t.u_right1, (size_t)t.bow.n1 * 4) : 0;  // stereo keyframes (FeatureMatcher.cc:705, :727)
        ur_off[2 * i + 1] = t.u_right2 ? b.put(t.u_right2, (size_t)t.bow.n2 * 4) : 0;
        // feature -> shared node (a feature sits in exactly one node of its FeatureVector)
        std::vector<Seg> segs;
        afv_shared_segments(t.bow, segs);
        std::vector<int> row_seg((size_t)std::max(t.bow.n1, 1), -1);
        const bool has_idx = t.bow.nnodes1 > 0 && t.bow.nnodes2 > 0;
        for (size_t sgi = 0; sgi < segs.size(); ++sgi)
            for (int r = 0; r < segs[sgi].n1; ++r) {
                const int f = has_idx ? t.bow.seg_idx1[segs[sgi].s1 + r] : segs[sgi].s1 + r;
                if (f >= 0 && f < t.bow.n1) row_seg[f] = (int)sgi;
            }
        rowseg_off[i] = b.put(row_seg.data(), row_seg.size() * 4);
    }
    size_t total_out = 0;
    for (int i = 0; i < njobs; ++i) total_out += (size_t)jobs[i].bow.n1;
    const size_t out_off = b.reserve(std::max<size_t>(total_out, 1) * 4);
    const size_t nm_off = b.reserve((size_t)njobs * 4);
    const size_t jobs_off = b.reserve((size_t)njobs * sizeof(DevTriJob));
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    size_t acc = 0;
    for (int i = 0; i < njobs; ++i) {
        offs[i].out = out_off + acc * 4;
        offs[i].nm = nm_off + (size_t)i * 4;
        acc += (size_t)jobs[i].bow.n1;
        DevTriJob &d = reinterpret_cast<DevTriJob *>(b.h.data() + jobs_off)[i];
        fill_dev_job(d.m, jobs[i].bow, offs[i], c->d_match, true);
        d.x1 = reinterpret_cast<const float *>(c->d_match + geo_off[5 * i + 0]);
        d.y1 = reinterpret_cast<const float *>(c->d_match + geo_off[5 * i + 1]);
        d.x2 = reinterpret_cast<const float *>(c->d_match + geo_off[5 * i + 2]);
        d.y2 = reinterpret_cast<const float *>(c->d_match + geo_off[5 * i + 3]);
        d.sigma2_2 = reinterpret_cast<const float *>(c->d_match + geo_off[5 * i + 4]);
        std::memcpy(d.F, jobs[i].F12, sizeof(d.F));
        d.ex = jobs[i].ex;
        d.ey = jobs[i].ey;
        d.row_seg = reinterpret_cast<const int *>(c->d_match + rowseg_off[i]);
        d.u_right1 = jobs[i].u_right1 ? reinterpret_cast<const float *>(c->d_match + ur_off[2 * i + 0]) : nullptr;
        d.u_right2 = jobs[i].u_right2 ? reinterpret_cast<const float *>(c->d_match + ur_off[2 * i + 1]) : nullptr;
        d.only_stereo = jobs[i].only_stereo;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), b.h.size(), hipMemcpyHostToDevice, c->stream));
    int max_n1 = 0;
    for (int i = 0; i < njobs; ++i) max_n1 = std::max(max_n1, jobs[i].bow.n1);
    afv_launch_match_tri(reinterpret_cast<const DevTriJob *>(c->d_match + jobs_off), njobs, max_n1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, b.fetch(match12, out_off, total_out * 4, c->stream));
    HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)njobs * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}
extern "C" int afv_match_triangulation(afv_ctx *c, const afv_tri_job *jobs, int njobs, int32_t *match12, int32_t *nmatches) {
    return guarded(c, [&] { return afv_match_triangulation_impl(c, jobs, njobs, match12, nmatches); });
}

// a pair whose fixed point hit its pass guard carries nmatches = -0x7fffffff (k_match_resolve_wg): entry points that hand results to the
// host report it (never observed outside the test hook afv_debug_pass_cap)
int afv_check_resolve_guard(afv_ctx *c, const int32_t *nmatches, int n) {
    for (int i = 0; i < n; ++i)
        if (nmatches[i] == -0x7fffffff) {
            c->last_error = "pair matcher: the fixed point of pair " + std::to_string(i) + " hit its pass guard; afv_set_match_resolve(ctx, 0) selects the ordered walk";
            return AFV_EHIP;
        }
    return AFV_OK;
}

// core of the device-resident brute-force batch; angles as a strided float array (see afv_launch_match_resolve)
int afv_match_pairs_core(afv_ctx *c, const uint8_t *d_desc, const float *d_ang, int ang_stride, const int32_t *d_n, int cap,
                         const int32_t *d_pair_a, const int32_t *d_pair_b, int npairs, float th_low, float nnratio,
                         int check_orientation, int32_t *d_match, int32_t *d_nmatches, hipStream_t s) {
    c->prof = c->prof_every && (c->prof_tick_match++ % (unsigned)c->prof_every) == 0;
    // the small-batch path deals the column tiles of phase 1 to several workgroups per row tile (two 64-column tiles each): a single
    // pair then runs on 32 workgroups instead of 4, and the resolve kernel merges the slices' key records
    const int nslices = small_batch_path(c, npairs) ? afv_match_topk_slices(cap, c->match_engine, ((cap + 63) / 64 + 1) / 2) : 1;
    {
        const int rc = ensure_slice_scratch(c, npairs, cap, nslices);
        if (rc) return rc;
    }
    const size_t need = (size_t)npairs * cap * 32;  // one 2 x int4 key record per row
    if (need > c->topk_bytes) {  // grow-only scratch (first call / larger batch): implies a device sync
        HIPCHK(c, hipDeviceSynchronize());
        if (c->d_topk) (void)hipFree(c->d_topk);
        c->d_topk = nullptr;
        c->topk_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_topk, need));
        c->topk_bytes = need;
    }
    if (npairs >= c->split_min_frames) {
        // grid.y carries the pair index (<= 65535 per launch) and the two streams overlap the latency-bound ordered
        // resolve of one chunk with the VALU-bound top-k of the other (more, smaller chunks were measured and are slower: 256 frames,
        // 4 / 6 / 8 chunks: -4 / -9 / -13 %)
        HIPCHK(c, hipEventRecord(c->ev_fork, s));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        const int K = std::max(2, (npairs + 32767) / 32768 * 2);
        for (int k = 0; k < K; ++k) {
            const int b0 = (int)((long)npairs * k / K), e0 = (int)((long)npairs * (k + 1) / K);
            if (e0 <= b0) continue;
            hipStream_t ks = (k & 1) ? c->stream2 : s;
            {
                StageTimer t_(c, AFV_STAGE_MATCH, ks, e0 - b0);
                afv_launch_match_topk(d_desc, d_n, cap, d_pair_a, d_pair_b, e0 - b0, c->d_topk, b0, c->match_engine, nslices, c->d_slice, c->d_tickets, ks);
            }
            StageTimer t_(c, AFV_STAGE_MATCH_RESOLVE, ks, e0 - b0);
            afv_launch_match_resolve(d_desc, d_ang, ang_stride, d_n, cap, d_pair_a, d_pair_b, e0 - b0, th_low, nnratio, check_orientation,
                                     d_match, d_nmatches, c->d_topk, b0, resolve_engine_for(c, npairs), ks);
        }
        HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_join, 0));
    } else {
        {
            StageTimer t_(c, AFV_STAGE_MATCH, s, npairs);
            afv_launch_match_topk(d_desc, d_n, cap, d_pair_a, d_pair_b, npairs, c->d_topk, 0, c->match_engine, nslices, c->d_slice, c->d_tickets, s);
        }
        StageTimer t_(c, AFV_STAGE_MATCH_RESOLVE, s, npairs);
        afv_launch_match_resolve(d_desc, d_ang, ang_stride, d_n, cap, d_pair_a, d_pair_b, npairs, th_low, nnratio, check_orientation, d_match,
                                 d_nmatches, c->d_topk, 0, resolve_engine_for(c, npairs), s);
    }
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            // a sliced launch that did not go out (or went out half) may leave row-tile tickets behind: zero them, so that the next
            // small-batch call on this context starts from rest instead of merging early or never
            if (nslices > 1 && c->d_tickets) (void)hipMemsetAsync(c->d_tickets, 0, c->tickets_n * sizeof(int), s);
            c->last_error = std::string("pair matcher launch: ") + hipGetErrorString(e);
            return AFV_EHIP;
        }
    }
    return AFV_OK;
}

extern "C" int afv_match_bruteforce_pairs_device(afv_ctx *c, const uint8_t *d_desc, const afv_keypoint *d_kps,
                                                 const int32_t *d_n, int nsets, int cap, const int32_t *d_pair_a,
                                                 const int32_t *d_pair_b, int npairs, float th_low, float nnratio,
                                                 int check_orientation, int32_t *d_match, int32_t *d_nmatches, void *stream) {
    if (!c || !d_desc || !d_n || !d_pair_a || !d_pair_b || !d_match || !d_nmatches) return AFV_EINVAL;
    if (nsets < 1 || npairs < 1 || cap < 1 || cap > 4096) return AFV_EINVAL;  // PAIR_MAX_SIDE in k_match.hip
    if (check_orientation && !d_kps) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    return afv_match_pairs_core(c, d_desc, d_kps ? &d_kps->angle : nullptr, (int)(sizeof(afv_keypoint) / sizeof(float)), d_n, cap,
                                d_pair_a, d_pair_b, npairs, th_low, nnratio, check_orientation, d_match, d_nmatches,
                                stream ? (hipStream_t)stream : c->stream);
}

static int afv_match_l2_impl(afv_ctx *c, const float *desc1, int n1, const float *desc2, int n2, int dim, const uint8_t *valid1,
                            const uint8_t *valid2, float th_low, float nnratio, int32_t *match12, int32_t *nmatches) {
    if (!c || !match12 || !nmatches || n1 < 0 || n2 < 0 || n1 > AFV_MAX_SIDE || n2 > AFV_MAX_SIDE || dim < 1 || dim > 1024)
        return AFV_EINVAL;
    if ((n1 > 0 && !desc1) || (n2 > 0 && !desc2)) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    Blob b(c);
    const size_t o1 = b.put(desc1, (size_t)n1 * dim * 4), o2 = b.put(desc2, (size_t)n2 * dim * 4);
    const size_t ov1 = valid1 ? b.put(valid1, (size_t)n1) : 0, ov2 = valid2 ? b.put(valid2, (size_t)n2) : 0;
    const size_t in_bytes = b.h.size();
    const size_t oo = b.reserve((size_t)std::max(n1, 1) * 4), on = b.reserve(4);
    int ntiles = 1, cols_per_tile = 32;
    const size_t ok = b.reserve_scratch(afv_match_l2_scratch_bytes(n1, n2, &ntiles, &cols_per_tile));
    const int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
    const float *p1 = reinterpret_cast<const float *>(c->d_match + o1), *p2 = reinterpret_cast<const float *>(c->d_match + o2);
    const uint8_t *pv1 = valid1 ? c->d_match + ov1 : nullptr, *pv2 = valid2 ? c->d_match + ov2 : nullptr;
    int *pout = reinterpret_cast<int *>(c->d_match + oo), *pn = reinterpret_cast<int *>(c->d_match + on);
    if (!afv_launch_match_l2_tiled(p1, n1, p2, n2, dim, pv1, pv2, th_low, nnratio, pout, pn, c->d_match + ok, ntiles, cols_per_tile, c->stream))
        afv_launch_match_l2(p1, n1, p2, n2, dim, pv1, pv2, th_low, nnratio, pout, pn, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, b.fetch(match12, oo, (size_t)n1 * 4, c->stream));
    HIPCHK(c, b.fetch(nmatches, on, 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}
extern "C" int afv_match_l2(afv_ctx *c, const float *desc1, int n1, const float *desc2, int n2, int dim, const uint8_t *valid1,
                            const uint8_t *valid2, float th_low, float nnratio, int32_t *match12, int32_t *nmatches) {
    return guarded(c, [&] { return afv_match_l2_impl(c, desc1, n1, desc2, n2, dim, valid1, valid2, th_low, nnratio, match12, nmatches); });
}

// device-resident batch of the float-descriptor matcher (config #3 as a throughput path, like afv_match_bruteforce_pairs_device for
// ORB32): the key scratch is the Hamming path's grow-only buffer (32 B per row there as well)
extern "C" int afv_match_l2_pairs_device(afv_ctx *c, const float *d_desc, const int32_t *d_n, int cap, int dim, const int32_t *d_pair_a,
                                         const int32_t *d_pair_b, int npairs, float th_low, float nnratio, int32_t *d_match,
                                         int32_t *d_nmatches, void *stream) {
    return guarded(c, [&]() -> int {
        if (!c || !d_desc || !d_n || !d_pair_a || !d_pair_b || !d_match || !d_nmatches || npairs < 0 || cap < 1 || cap > AFV_MAX_SIDE)
            return AFV_EINVAL;
        if (dim != 64 && dim != 128) return AFV_EUNSUPPORTED;
        if (npairs == 0) return AFV_OK;
        HIPCHK(c, hipSetDevice(c->device));
        hipStream_t s = stream ? (hipStream_t)stream : c->stream;
        const int chunk = std::min(npairs, c->l2_chunk_pairs);  // pairs per launch: grid.y and the scratch stay bounded
        // the float matcher's own key scratch: the Hamming pair calls keep theirs (d_topk) busy on the context's streams, and a caller
        // may run the two kinds on different streams of one context
        const size_t need = (size_t)chunk * cap * 32;
        if (need > c->l2_bytes) {  // grow-only (first call / larger batch): implies a device sync
            HIPCHK(c, hipDeviceSynchronize());
            if (c->d_l2_scratch) (void)hipFree(c->d_l2_scratch);
            c->d_l2_scratch = nullptr;
            c->l2_bytes = 0;
            HIPCHK(c, hipMalloc(&c->d_l2_scratch, need));
            c->l2_bytes = need;
        }
        // chunks reuse the scratch one after the other: same stream, so chunk k + 1 starts after chunk k has read its keys
        for (int b0 = 0; b0 < npairs; b0 += chunk)
            if (!afv_launch_match_l2_pairs(d_desc, d_n, cap, dim, d_pair_a, d_pair_b, std::min(chunk, npairs - b0), b0, th_low, nnratio, d_match,
                                           d_nmatches, c->d_l2_scratch, s))
                return AFV_EUNSUPPORTED;
        HIPCHK(c, hipGetLastError());
        return AFV_OK;
    });
}

// ---- projection-guided matching ----
// One implementation behind afv_match_projection / _fuse / _initialization / _sim3 (host arrays on both sides) and the afv_frame_* forms
// (afv_frame.hip: the feature side, its grid and possibly the queries' descriptors are already on the device: `dev`).  The grid of
// Frame::AssignFeaturesToGrid is built ON THE DEVICE in both cases (k_frame_grid): the host-array form uploads x / y / size and runs the
// same kernel a resident frame ran when it was extracted.
int afv_match_projection_core(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *assign, int32_t *nmatches, int kind,
                              const ProjFeatureSide *dev) {
    const bool fuse = kind == AFV_KIND_FUSE, per_query = kind != AFV_KIND_PROJ;
    if (!c || !jobs || njobs < 1 || !assign || !nmatches) return AFV_EINVAL;
    if (dev && njobs != 1) return AFV_EINVAL;
    for (int i = 0; i < njobs; ++i) {
        const afv_proj_job &j = jobs[i];
        if (j.n < 0 || j.n > AFV_MAX_SIDE || j.nq < 0 || j.nq > 65535) return AFV_EINVAL;
        if (j.float_dim != 0) {  // float descriptors (L2^2): rows of float_dim floats
            if (j.float_dim < 4 || j.float_dim > 1024 || (j.float_dim & 3)) return AFV_EINVAL;
        } else if (j.desc_bytes < 1 || j.desc_bytes > 64) {
            return AFV_EINVAL;
        }
        if (j.grid_cols < 1 || j.grid_rows < 1 || (long)j.grid_cols * j.grid_rows > 8192) return AFV_EINVAL;
        if (!dev && j.n > 0 && (!j.desc || !j.x || !j.y || !j.size)) return AFV_EINVAL;
        if (j.nq > 0 && ((!j.qdesc && !(dev && (dev->qdesc_dev || dev->qref_table))) || !j.qu || !j.qv || !j.qr || !j.qmin_size || !j.qmax_size)) return AFV_EINVAL;
        const bool has_angle = dev ? dev->angle != nullptr : j.angle != nullptr;
        const bool has_qangle = j.qangle != nullptr || (dev && dev->qangle_dev);
        if (kind == AFV_KIND_INIT && j.check_orientation && ((j.n > 0 && !has_angle) || (j.nq > 0 && !has_qangle))) return AFV_EINVAL;
        if (kind == AFV_KIND_PROJ && j.mode != AFV_PROJ_LOCALMAP && j.mode != AFV_PROJ_LASTFRAME) return AFV_EINVAL;
        if (kind == AFV_KIND_PROJ && j.mode == AFV_PROJ_LASTFRAME && j.check_orientation && ((j.n > 0 && !has_angle) || (j.nq > 0 && !has_qangle)))
            return AFV_EINVAL;
        // stereo frames: the queries' right-image coordinate (and, for the projection searches, their gate) come with mvuRight
        if (j.u_right && (kind == AFV_KIND_PROJ || kind == AFV_KIND_FUSE) && j.nq > 0 && (!j.q_ur || (kind == AFV_KIND_PROJ && !j.q_er_max)))
            return AFV_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->device));
    Blob b(c);
    struct Off { size_t fd, x, y, size, angle, occ, inf, cptr, cent, qd, qrs, qri, qvalid, qu, qv, qr, qmin, qmax, qang, qocc, keys, ncand, ori, ur, qur, qer; int words; bool stereo; };
    std::vector<Off> offs(njobs);
    size_t total_out = 0;
    int max_nq = 0, max_n = 0;
    size_t grid_lds = 0;
    bool any_float = false;
    for (int i = 0; i < njobs; ++i) {
        const afv_proj_job &j = jobs[i];
        Off &o = offs[i];
        o = Off{};
        o.words = j.float_dim ? j.float_dim : (j.desc_bytes <= 32 ? 8 : 16);  // dwords of one row
        any_float = any_float || j.float_dim != 0;
        if (dev && dev->words != o.words) return AFV_EINVAL;
        // mvuRight branches: FeatureMatcher.cc:114-119, :1367-1372, :880-894.  A resident frame always carries the plane (-1 = monocular);
        // it takes part when the caller sends the queries' side of the gate
        const bool ur_here = dev ? (dev->u_right != nullptr && j.q_ur != nullptr) : j.u_right != nullptr;
        o.stereo = ur_here && (kind == AFV_KIND_PROJ || kind == AFV_KIND_FUSE);
        if (o.stereo && kind == AFV_KIND_PROJ && j.nq > 0 && !j.q_er_max) return AFV_EINVAL;
        if (!dev) {
            o.fd = j.float_dim ? b.put(j.desc, (size_t)j.n * j.float_dim * 4) : put_desc(b, j.desc, j.n, j.desc_bytes, o.words);
            o.x = b.put(j.x, (size_t)j.n * 4); o.y = b.put(j.y, (size_t)j.n * 4); o.size = b.put(j.size, (size_t)j.n * 4);
            o.angle = j.angle ? b.put(j.angle, (size_t)j.n * 4) : 0;
            o.inf = (fuse && j.inf) ? b.put(j.inf, (size_t)j.n * 4) : 0;
            o.ur = o.stereo ? b.put(j.u_right, (size_t)j.n * 4) : 0;
            grid_lds = std::max(grid_lds, afv_frame_grid_lds(j.grid_cols, j.grid_rows, std::max(j.n, 1)));
        }
        o.occ = (j.occupied && kind != AFV_KIND_INIT) ? b.put(j.occupied, (size_t)j.n) : 0;
        o.qur = o.stereo ? b.put(j.q_ur, (size_t)j.nq * 4) : 0;
        o.qer = (o.stereo && kind == AFV_KIND_PROJ) ? b.put(j.q_er_max, (size_t)j.nq * 4) : 0;
        const bool by_ref = dev && dev->qref_table && !dev->qdesc_dev;
        if (by_ref) {
            // MapPoint descriptors by reference (rows of a keyframe table): checked here, gathered on the device behind the upload
            const afv_table *qt = dev->qref_table;
            if (qt->c != c || !dev->qref_slot || !dev->qref_idx || o.words != 8) return AFV_EINVAL;
            for (int q = 0; q < j.nq; ++q) {
                const int sl = dev->qref_slot[q];
                if (sl < 0 || sl >= qt->nsets || dev->qref_idx[q] < 0 || dev->qref_idx[q] >= qt->h_n[sl]) return AFV_EINVAL;
            }
            o.qrs = b.put(dev->qref_slot, (size_t)j.nq * 4);
            o.qri = b.put(dev->qref_idx, (size_t)j.nq * 4);
        } else if (!(dev && dev->qdesc_dev)) {
            o.qd = j.float_dim ? b.put(j.qdesc, (size_t)j.nq * j.float_dim * 4) : put_desc(b, j.qdesc, j.nq, j.desc_bytes, o.words);
        }
        o.qvalid = (j.qvalid && !(dev && dev->qvalid_dev)) ? b.put(j.qvalid, (size_t)j.nq) : 0;
        o.qu = b.put(j.qu, (size_t)j.nq * 4); o.qv = b.put(j.qv, (size_t)j.nq * 4); o.qr = b.put(j.qr, (size_t)j.nq * 4);
        o.qmin = b.put(j.qmin_size, (size_t)j.nq * 4); o.qmax = b.put(j.qmax_size, (size_t)j.nq * 4);
        o.qang = (j.qangle && !(dev && dev->qangle_dev)) ? b.put(j.qangle, (size_t)j.nq * 4) : 0;
        o.qocc = j.qoccupies ? b.put(j.qoccupies, (size_t)j.nq) : 0;
        total_out += (size_t)(per_query ? j.nq : j.n);
        max_nq = std::max(max_nq, j.nq);
        max_n = std::max(max_n, j.n);
    }
    // records the kernels read: the search jobs and, for staged feature sides, the grid jobs (uploaded with the inputs)
    const size_t jobs_off = b.reserve((size_t)njobs * sizeof(DevProjJob));
    const size_t gjobs_off = dev ? 0 : b.reserve((size_t)njobs * sizeof(DevGridJob));
    const size_t nm_off = b.reserve((size_t)njobs * 4);
    const size_t in_bytes = b.h.size();
    for (int i = 0; i < njobs; ++i) {  // device-only scratch
        const afv_proj_job &j = jobs[i];
        offs[i].keys = b.reserve_scratch((size_t)std::max(j.nq, 1) * 64);  // 64-byte record / 8 keys per query
        offs[i].ncand = b.reserve_scratch((size_t)std::max(j.nq, 1) * 4);
        offs[i].ori = b.reserve_scratch((size_t)std::max(j.nq, 1) * 8);
        if (dev && dev->qref_table && !dev->qdesc_dev) offs[i].qd = b.reserve_scratch((size_t)std::max(j.nq, 1) * 32);
        if (!dev) {
            offs[i].cptr = b.reserve_scratch(((size_t)j.grid_cols * j.grid_rows + 1) * 4);
            offs[i].cent = b.reserve_scratch((size_t)std::max(j.n, 1) * 16);
        }
    }
    const size_t out_off = b.reserve_scratch(std::max<size_t>(total_out, 1) * 4);
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    // ordered phase: the workgroup fixed point when the largest job's tables fit the LDS it may use
    size_t wg_lds = 0;
    // (float descriptors: the projection searches' fixed point carries float distances; SearchForInitialization's packs them in 16 bits and
    // float jobs take its ordered walk)
    if (!fuse && !(any_float && kind == AFV_KIND_INIT) && c->proj_engine != 0 && c->proj_wg_lds_max > 0 && (kind != AFV_KIND_INIT || max_nq <= 32767)) {
        for (int i = 0; i < njobs; ++i) wg_lds = std::max(wg_lds, afv_project_wg_lds(kind == AFV_KIND_INIT, jobs[i].n, jobs[i].nq, any_float ? 1 : 0));
        if (wg_lds > (size_t)c->proj_wg_lds_max) wg_lds = 0;
    }
    // results straight into the pinned arena (device-visible host memory) when the kernels write them once and never read them back
    const bool zero_copy = c->stage_pinned && (fuse || wg_lds != 0);
    // ... and, for ONE job against a resident frame, the inputs straight out of it: the job record is the kernel argument, the queries
    // (a few KB per array, read once by the ranking kernel) come over the link without a copy-engine hop ahead of the launch
    // (not with an occupancy mask: that one is gathered per candidate, which belongs in device memory)
    // (nor with float rows: a query row is 4 * dim bytes and is read once per CANDIDATE - that belongs in device memory too)
    const bool zero_copy_in = zero_copy && dev && njobs == 1 && !any_float && !(jobs[0].occupied && kind != AFV_KIND_INIT);
    uint8_t *B = c->d_match, *H = b.h.data();
    uint8_t *IN = zero_copy_in ? H : B;  // where the kernels find the staged inputs
    size_t acc = 0;
    for (int i = 0; i < njobs; ++i) {
        const afv_proj_job &j = jobs[i];
        const Off &o = offs[i];
        DevProjJob &d = reinterpret_cast<DevProjJob *>(H + jobs_off)[i];
        d = DevProjJob{};
        d.n = j.n; d.words = j.float_dim ? 0 : o.words; d.fdim = j.float_dim;
        if (dev) {
            d.fdesc = dev->fdesc; d.x = dev->x; d.y = dev->y; d.size = dev->size; d.angle = dev->angle;
            d.inf = fuse ? dev->inf : nullptr;
            d.u_right = o.stereo ? dev->u_right : nullptr;
            d.cell_ptr = dev->cell_ptr; d.cell_ent = dev->cell_ent;
        } else {
            d.fdesc = reinterpret_cast<const uint32_t *>(B + o.fd);
            d.x = reinterpret_cast<const float *>(B + o.x); d.y = reinterpret_cast<const float *>(B + o.y);
            d.size = reinterpret_cast<const float *>(B + o.size);
            d.angle = j.angle ? reinterpret_cast<const float *>(B + o.angle) : nullptr;
            d.inf = (fuse && j.inf) ? reinterpret_cast<const float *>(B + o.inf) : nullptr;
            d.u_right = o.stereo ? reinterpret_cast<const float *>(B + o.ur) : nullptr;
            d.cell_ptr = reinterpret_cast<const int *>(B + o.cptr); d.cell_ent = reinterpret_cast<const int4 *>(B + o.cent);
            DevGridJob &g = reinterpret_cast<DevGridJob *>(H + gjobs_off)[i];
            g = DevGridJob{};
            g.n = j.n; g.cap = std::max(j.n, 1);
            g.x = const_cast<float *>(d.x); g.y = const_cast<float *>(d.y); g.size = const_cast<float *>(d.size);
            g.min_x = j.min_x; g.min_y = j.min_y; g.inv_w = j.grid_inv_w; g.inv_h = j.grid_inv_h; g.cols = j.grid_cols; g.rows = j.grid_rows;
            g.cell_ptr = reinterpret_cast<int *>(B + o.cptr); g.cell_ent = reinterpret_cast<int4 *>(B + o.cent);
        }
        d.occupied = (j.occupied && kind != AFV_KIND_INIT) ? IN + o.occ : nullptr;
        d.min_x = j.min_x; d.min_y = j.min_y; d.inv_w = j.grid_inv_w; d.inv_h = j.grid_inv_h; d.cols = j.grid_cols; d.rows = j.grid_rows;
        d.nq = j.nq;
        d.qdesc = (dev && dev->qdesc_dev) ? dev->qdesc_dev
                                          : reinterpret_cast<const uint32_t *>(((dev && dev->qref_table) ? B : IN) + o.qd);
        d.qvalid = (dev && dev->qvalid_dev) ? dev->qvalid_dev : (j.qvalid ? IN + o.qvalid : nullptr);
        d.qu = reinterpret_cast<const float *>(IN + o.qu); d.qv = reinterpret_cast<const float *>(IN + o.qv);
        d.qr = reinterpret_cast<const float *>(IN + o.qr); d.qmin = reinterpret_cast<const float *>(IN + o.qmin);
        d.qmax = reinterpret_cast<const float *>(IN + o.qmax);
        d.qangle = (dev && dev->qangle_dev) ? dev->qangle_dev : (j.qangle ? reinterpret_cast<const float *>(IN + o.qang) : nullptr);
        d.qocc = j.qoccupies ? IN + o.qocc : nullptr;
        d.th = j.th_high; d.ratio = j.nnratio; d.tol = j.size_tol; d.inv_tol = j.inv_size_tol;
        d.check_ori = j.check_orientation != 0; d.mode = j.mode;
        d.pass_cap = afv_debug_pass_cap;
        d.keys = reinterpret_cast<unsigned long long *>(B + o.keys); d.ncand = reinterpret_cast<int *>(B + o.ncand);
        d.orilist = reinterpret_cast<int *>(B + o.ori);
        d.q_ur = o.stereo ? reinterpret_cast<const float *>(IN + o.qur) : nullptr;
        d.q_er = (o.stereo && kind == AFV_KIND_PROJ) ? reinterpret_cast<const float *>(IN + o.qer) : nullptr;
        d.stereo_gate = (o.stereo && kind == AFV_KIND_PROJ) ? 1 : 0;
        uint8_t *R = zero_copy ? H : B;
        d.assign = reinterpret_cast<int *>(R + out_off + acc * 4); d.nmatches = reinterpret_cast<int *>(R + nm_off + (size_t)i * 4);
        acc += (size_t)(per_query ? j.nq : j.n);
    }
    if (!zero_copy_in) HIPCHK(c, hipMemcpyAsync(B, H, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (!dev) {
        if (grid_lds > (size_t)c->frame_lds_max) {
            c->last_error = "projection search: the grid of the feature side does not fit the LDS of one workgroup (cells x features too large)";
            return AFV_EUNSUPPORTED;
        }
        afv_launch_frame_grid(reinterpret_cast<const DevGridJob *>(B + gjobs_off), njobs, grid_lds, c->stream);
    }
    if (dev && dev->qref_table && !dev->qdesc_dev) {
        const afv_table *qt = dev->qref_table;
        afv_launch_frame_gather(qt->d_desc, qt->d_n, qt->nsets, qt->cap, reinterpret_cast<const int *>(IN + offs[0].qrs),
                                reinterpret_cast<const int *>(IN + offs[0].qri), jobs[0].nq, B + offs[0].qd, nullptr, c->stream);
    }
    const DevProjJob *dj = reinterpret_cast<const DevProjJob *>(B + jobs_off);
    const DevProjJob *one = zero_copy_in ? reinterpret_cast<const DevProjJob *>(H + jobs_off) : nullptr;
    // one launch for ranking + ordered phase: the projection searches (a few candidates per query).  SearchForInitialization keeps two: its
    // ranking walks 100-pixel windows (hundreds of cells per query) and is better off on 250 four-wave workgroups than on 63 sixteen-wave
    // ones (measured: 59.6 us against 72.5 host to host)
    int *ticket = (one && wg_lds && c->proj_fuse && kind == AFV_KIND_PROJ && !any_float) ? c->d_proj_ticket : nullptr;  // (the one-launch kernel is binary-only)
    if (fuse) afv_launch_match_fuse(dj, njobs, max_nq, one, c->stream);
    else if (kind == AFV_KIND_INIT) afv_launch_match_init(dj, njobs, max_nq, wg_lds, one, ticket, c->stream);
    else afv_launch_match_projection(dj, njobs, max_nq, wg_lds, one, ticket, c->stream);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            if (ticket) (void)hipMemsetAsync(ticket, 0, sizeof(int), c->stream);  // a launch that did not go out must not leave the ticket armed
            c->last_error = std::string("projection search launch: ") + hipGetErrorString(e);
            return AFV_EHIP;
        }
    }
    if (!zero_copy) {
        HIPCHK(c, b.fetch(assign, out_off, total_out * 4, c->stream));
        if (!fuse) HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)njobs * 4, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (zero_copy) {
        std::memcpy(assign, H + out_off, total_out * 4);
        if (!fuse) std::memcpy(nmatches, H + nm_off, (size_t)njobs * 4);
    } else {
        b.finish();
    }
    if (fuse) {  // independent queries: the count is just the number of hits
        size_t at = 0;
        for (int i = 0; i < njobs; ++i) {
            int found = 0;
            for (int q = 0; q < jobs[i].nq; ++q) found += assign[at + q] >= 0;
            nmatches[i] = found;
            at += (size_t)jobs[i].nq;
        }
    } else {
        for (int i = 0; i < njobs; ++i)
            if (nmatches[i] == -0x7fffffff) {  // PW_GUARD: the fixed point did not settle within its pass guard (never observed)
                c->last_error = "projection search: the fixed point hit its pass guard; afv_set_projection_resolve(ctx, 0) selects the ordered walk";
                return AFV_EHIP;
            }
    }
    return AFV_OK;
}

// job arrays arrive with the layout the caller was compiled against (struct_size): bring them to the current one
static int proj_jobs_entry(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *out, int32_t *nm, int kind) {
    return guarded(c, [&]() -> int {
        std::vector<afv_proj_job> J;
        if (!afv_load_jobs(jobs, njobs, offsetof(afv_proj_job, u_right), J)) return AFV_EINVAL;
        return afv_match_projection_core(c, J.data(), njobs, out, nm, kind, nullptr);
    });
}
extern "C" int afv_match_projection(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *assign, int32_t *nmatches) {
    return proj_jobs_entry(c, jobs, njobs, assign, nmatches, AFV_KIND_PROJ);
}
extern "C" int afv_match_fuse(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *best, int32_t *nfound) {
    return proj_jobs_entry(c, jobs, njobs, best, nfound, AFV_KIND_FUSE);
}
extern "C" int afv_match_initialization(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *match12, int32_t *nmatches) {
    return proj_jobs_entry(c, jobs, njobs, match12, nmatches, AFV_KIND_INIT);
}
static int afv_match_sim3_impl(afv_ctx *c, const afv_proj_job *j12, const afv_proj_job *j21, int32_t *match12, int32_t *nfound) {
    if (!c || !j12 || !j21 || !match12 || !nfound) return AFV_EINVAL;
    std::vector<afv_proj_job> A, Bv;
    if (!afv_load_jobs(j12, 1, offsetof(afv_proj_job, u_right), A) || !afv_load_jobs(j21, 1, offsetof(afv_proj_job, u_right), Bv)) return AFV_EINVAL;
    if (A[0].nq != Bv[0].n || Bv[0].nq != A[0].n) return AFV_EINVAL;
    afv_proj_job jobs[2] = {A[0], Bv[0]};
    jobs[0].inf = nullptr;  // no reprojection gate in SearchBySim3
    jobs[1].inf = nullptr;
    jobs[0].u_right = jobs[1].u_right = nullptr;  // ... and no stereo branch (FeatureMatcher.cc:1066-1287)
    std::vector<int32_t> best((size_t)jobs[0].nq + (size_t)jobs[1].nq + 1);
    int32_t nf[2];
    const int rc = afv_match_projection_core(c, jobs, 2, best.data(), nf, AFV_KIND_FUSE, nullptr);
    if (rc) return rc;
    const int32_t *m1 = best.data(), *m2 = best.data() + jobs[0].nq;
    int found = 0;
    for (int i1 = 0; i1 < jobs[0].nq; ++i1) {  // FeatureMatcher.cc:1268-1284
        const int idx2 = m1[i1];
        const bool agree = idx2 >= 0 && m2[idx2] == i1;
        match12[i1] = agree ? idx2 : -1;
        found += agree;
    }
    *nfound = found;
    return AFV_OK;
}
extern "C" int afv_match_sim3(afv_ctx *c, const afv_proj_job *j12, const afv_proj_job *j21, int32_t *match12, int32_t *nfound) {
    return guarded(c, [&] { return afv_match_sim3_impl(c, j12, j21, match12, nfound); });
}
extern "C" int afv_set_projection_resolve(afv_ctx *c, int engine) {
    if (!c || engine < 0 || engine > 3) return AFV_EINVAL;
    c->proj_fuse = engine != 3;          // 3 = the fixed point as two launches (ranking, then ordered phase): the A / B of the one-launch form
    c->proj_engine = engine == 3 ? 1 : engine;
    return AFV_OK;
}

// ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:279-349) for a batch of map points ----
extern "C" int afv_distinctive_descriptors(afv_ctx *c, const uint8_t *desc, int desc_bytes, const int32_t *set_ptr, int nsets, int32_t *best_idx,
                                           int32_t *best_median) {
    if (!c || !set_ptr || nsets < 0 || desc_bytes < 1 || desc_bytes > 64 || (nsets > 0 && !best_idx)) return AFV_EINVAL;
    if (nsets == 0) return AFV_OK;
    if (set_ptr[0] != 0) return AFV_EINVAL;
    for (int s = 0; s < nsets; ++s)
        if (set_ptr[s + 1] < set_ptr[s] || set_ptr[s + 1] - set_ptr[s] > 65535) return AFV_EINVAL;  // (the kernel's key holds the row in 16 bits)
    const int total = set_ptr[nsets];
    if (total > 0 && !desc) return AFV_EINVAL;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        Blob b(c);
        const int words = desc_bytes <= 32 ? 8 : 16;
        const size_t d_off = put_desc(b, desc, total, desc_bytes, words);
        const size_t p_off = b.put(set_ptr, (size_t)(nsets + 1) * 4);
        const size_t in_bytes = b.h.size();
        const size_t bi_off = b.reserve((size_t)nsets * 4), bm_off = b.reserve((size_t)nsets * 4);
        const int rc = ensure_match_buffer(c, b.h.size());
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
        afv_launch_distinctive(reinterpret_cast<const uint32_t *>(c->d_match + d_off), reinterpret_cast<const int *>(c->d_match + p_off), nsets, words,
                               desc_bytes, reinterpret_cast<int *>(c->d_match + bi_off), reinterpret_cast<int *>(c->d_match + bm_off), c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, b.fetch(best_idx, bi_off, (size_t)nsets * 4, c->stream));
        if (best_median) HIPCHK(c, b.fetch(best_median, bm_off, (size_t)nsets * 4, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        b.finish();
        return AFV_OK;
    });
}

extern "C" int afv_distinctive_descriptors_f32(afv_ctx *c, const float *desc, int dim, const int32_t *set_ptr, int nsets, int32_t *best_idx,
                                               float *best_median) {
    if (!c || !set_ptr || nsets < 0 || dim < 1 || dim > 1024 || (nsets > 0 && !best_idx)) return AFV_EINVAL;
    if (nsets == 0) return AFV_OK;
    if (set_ptr[0] != 0) return AFV_EINVAL;
    for (int s = 0; s < nsets; ++s)
        if (set_ptr[s + 1] < set_ptr[s] || set_ptr[s + 1] - set_ptr[s] > 65535) return AFV_EINVAL;
    const int total = set_ptr[nsets];
    if (total > 0 && !desc) return AFV_EINVAL;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        Blob b(c);
        const size_t d_off = b.put(desc, (size_t)total * dim * 4);
        const size_t p_off = b.put(set_ptr, (size_t)(nsets + 1) * 4);
        const size_t in_bytes = b.h.size();
        const size_t bi_off = b.reserve((size_t)nsets * 4), bm_off = b.reserve((size_t)nsets * 4);
        const int rc = ensure_match_buffer(c, b.h.size());
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
        afv_launch_distinctive_f32(reinterpret_cast<const float *>(c->d_match + d_off), dim, reinterpret_cast<const int *>(c->d_match + p_off), nsets,
                                   reinterpret_cast<int *>(c->d_match + bi_off), reinterpret_cast<float *>(c->d_match + bm_off), c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, b.fetch(best_idx, bi_off, (size_t)nsets * 4, c->stream));
        if (best_median) HIPCHK(c, b.fetch(best_median, bm_off, (size_t)nsets * 4, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        b.finish();
        return AFV_OK;
    });
}

// ---- BoW quantisation ----
// the tree image of k_bow.hip for node descriptors of `words` dwords each (binary: 8 or 16, zero-padded; float: the dimension)
static int vocab_create_impl(afv_ctx *c, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx, const uint8_t *desc, int desc_bytes,
                             int words, int float_dim, afv_vocab **out);

extern "C" int afv_vocab_create(afv_ctx *c, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx,
                                const uint8_t *desc, int desc_bytes, afv_vocab **out) {
    if (!c || !out || !child_ptr || !child_idx || !desc || k < 1 || L < 1 || nnodes < 1 || desc_bytes < 1 || desc_bytes > 64)
        return AFV_EINVAL;
    return vocab_create_impl(c, k, L, nnodes, child_ptr, child_idx, desc, desc_bytes, desc_bytes <= 32 ? 8 : 16, 0, out);
}

// float node descriptors (the non-binary cases of Vocabulary::transform, Vocabulary.cpp:158-187): dim = 64 (SURF64 / KAZE64), 128 (SIFT128 / R2D2) or 256
extern "C" int afv_vocab_create_f32(afv_ctx *c, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx, const float *desc, int dim,
                                    afv_vocab **out) {
    if (!c || !out || !child_ptr || !child_idx || !desc || k < 1 || L < 1 || nnodes < 1 || (dim != 64 && dim != 128 && dim != 256)) return AFV_EINVAL;
    return vocab_create_impl(c, k, L, nnodes, child_ptr, child_idx, reinterpret_cast<const uint8_t *>(desc), dim * 4, dim, dim, out);
}

static int vocab_create_impl(afv_ctx *c, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx, const uint8_t *desc, int desc_bytes,
                             int words, int float_dim, afv_vocab **out) {
    *out = nullptr;
    const int nchild = child_ptr[nnodes];
    if (child_ptr[0] != 0 || nchild < 0 || nchild > nnodes) return AFV_EINVAL;
    for (int i = 0; i < nnodes; ++i)
        if (child_ptr[i + 1] < child_ptr[i]) return AFV_EINVAL;
    {   // a tree: every non-root node is the child of exactly one node (no duplicates => no cycles reachable from the
        // root, so the descent of k_bow_transform terminates)
        std::vector<uint8_t> seen((size_t)nnodes, 0);
        for (int i = 0; i < nchild; ++i) {
            if (child_idx[i] <= 0 || child_idx[i] >= nnodes || seen[child_idx[i]]) return AFV_EINVAL;
            seen[child_idx[i]] = 1;
        }
        // ... and no node is its own ancestor: walk the levels from the root, at most L of them may have children
        std::vector<int> frontier{0}, next;
        for (int depth = 0; !frontier.empty(); ++depth) {
            next.clear();
            for (int nd : frontier)
                for (int q = child_ptr[nd]; q < child_ptr[nd + 1]; ++q) next.push_back(child_idx[q]);
            if (!next.empty() && depth >= L) return AFV_EINVAL;
            frontier.swap(next);
        }
    }
    for (int i = 0; i < nnodes; ++i)
        if (child_ptr[i + 1] - child_ptr[i] > 65535) return AFV_EINVAL;  // the descent's key carries the child position in 16 bits
    HIPCHK(c, hipSetDevice(c->device));
    afv_vocab *v = new (std::nothrow) afv_vocab();
    if (!v) return AFV_ENOMEM;
    v->desc_bytes = desc_bytes;
    v->float_dim = float_dim;
    const int RD = words + 4;
    // device image (k_bow.hip): nodes renumbered breadth first, the children of a node consecutive and in DBoW2 order; record =
    // descriptor | first child record | #children | DBoW2 id | 0.  Nodes the root does not reach keep no record.
    std::vector<int> order;  // record -> DBoW2 id
    order.reserve((size_t)nnodes);
    order.push_back(0);
    std::vector<uint32_t> rec((size_t)nnodes * RD, 0);
    std::vector<int> depth((size_t)nnodes, -1);  // by DBoW2 id; -1: not reachable from the root
    depth[0] = 0;
    for (size_t r = 0; r < order.size(); ++r) {
        const int id = order[r];
        for (int q = child_ptr[id]; q < child_ptr[id + 1]; ++q) depth[child_idx[q]] = depth[id] + 1;
        uint32_t *R = rec.data() + r * RD;
        std::memcpy(R, desc + (size_t)id * desc_bytes, (size_t)desc_bytes);
        const int nc = child_ptr[id + 1] - child_ptr[id];
        R[words] = (uint32_t)order.size();
        R[words + 1] = (uint32_t)nc;
        R[words + 2] = (uint32_t)id;
        for (int q = child_ptr[id]; q < child_ptr[id + 1]; ++q) order.push_back(child_idx[q]);
    }
    {   // rank of every node among the nodes of its depth, ascending DBoW2 id: the FeatureVector's sort key (k_featvec_build)
        std::vector<int> rank_of((size_t)nnodes, 0);
        v->depth_width.clear();
        for (int id = 0; id < nnodes; ++id) {
            if (depth[id] < 0) continue;
            if ((size_t)depth[id] >= v->depth_width.size()) v->depth_width.resize((size_t)depth[id] + 1, 0);
            rank_of[id] = v->depth_width[(size_t)depth[id]]++;
        }
        for (size_t r = 0; r < order.size(); ++r) rec[r * RD + words + 3] = (uint32_t)rank_of[order[r]];
    }
    hipError_t e = hipMalloc(&v->d_rec, rec.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(v->d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        c->last_error = std::string("afv_vocab_create: ") + hipGetErrorString(e);
        afv_vocab_destroy(c, v);
        return e == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;
    }
    v->dev = DevVocab{k, L, nnodes, words, RD, (const uint32_t *)v->d_rec, nullptr};
    *out = v;
    return AFV_OK;
}

extern "C" int afv_vocab_set_stopped(afv_ctx *c, afv_vocab *v, const uint8_t *stopped) {
    if (!c || !v) return AFV_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!stopped) {
        v->dev.stopped = nullptr;
        v->h_stopped.clear();
        return AFV_OK;
    }
    try {
        v->h_stopped.assign(stopped, stopped + v->dev.nnodes);
    } catch (...) {
        return AFV_ENOMEM;
    }
    if (!v->d_stopped) HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&v->d_stopped), (size_t)v->dev.nnodes));
    HIPCHK(c, hipMemcpy(v->d_stopped, stopped, (size_t)v->dev.nnodes, hipMemcpyHostToDevice));
    v->dev.stopped = v->d_stopped;
    return AFV_OK;
}

extern "C" void afv_vocab_destroy(afv_ctx *c, afv_vocab *v) {
    if (!v) return;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    if (v->d_rec) (void)hipFree(v->d_rec);
    if (v->d_stopped) (void)hipFree(v->d_stopped);
    delete v;
}

static int afv_bow_transform_impl(afv_ctx *c, const afv_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *leaf_node,
                                 int32_t *node_at_level) {
    if (!c || !v || n < 0 || (n > 0 && (!desc || !leaf_node || !node_at_level))) return AFV_EINVAL;
    if (v->float_dim) return AFV_EINVAL;  // a float vocabulary: afv_bow_transform_f32
    if (n == 0) return AFV_OK;
    HIPCHK(c, hipSetDevice(c->device));
    Blob b(c);
    const size_t d_off = put_desc(b, desc, n, v->desc_bytes, v->dev.words);
    const size_t in_bytes = b.h.size();
    const size_t leaf_off = b.reserve((size_t)n * 4), nid_off = b.reserve((size_t)n * 4);
    const int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
    afv_launch_bow_transform(&v->dev, reinterpret_cast<const uint32_t *>(c->d_match + d_off), n, levelsup,
                             reinterpret_cast<int *>(c->d_match + leaf_off), reinterpret_cast<int *>(c->d_match + nid_off), nullptr, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, b.fetch(leaf_node, leaf_off, (size_t)n * 4, c->stream));
    HIPCHK(c, b.fetch(node_at_level, nid_off, (size_t)n * 4, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}
extern "C" int afv_bow_transform(afv_ctx *c, const afv_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *leaf_node,
                                 int32_t *node_at_level) {
    return guarded(c, [&] { return afv_bow_transform_impl(c, v, desc, n, levelsup, leaf_node, node_at_level); });
}

extern "C" int afv_bow_transform_f32(afv_ctx *c, const afv_vocab *v, const float *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level) {
    if (!c || !v || n < 0 || (n > 0 && (!desc || !leaf_node || !node_at_level))) return AFV_EINVAL;
    if (!v->float_dim) return AFV_EINVAL;  // a binary vocabulary: afv_bow_transform
    if (n == 0) return AFV_OK;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        Blob b(c);
        const size_t d_off = b.put(desc, (size_t)n * v->float_dim * 4);
        const size_t in_bytes = b.h.size();
        const size_t leaf_off = b.reserve((size_t)n * 4), nid_off = b.reserve((size_t)n * 4);
        const int rc = ensure_match_buffer(c, b.h.size());
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
        if (!afv_launch_bow_transform_f32(&v->dev, reinterpret_cast<const float *>(c->d_match + d_off), n, v->float_dim, levelsup,
                                          reinterpret_cast<int *>(c->d_match + leaf_off), reinterpret_cast<int *>(c->d_match + nid_off), nullptr, c->stream))
            return AFV_EUNSUPPORTED;
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, b.fetch(leaf_node, leaf_off, (size_t)n * 4, c->stream));
        HIPCHK(c, b.fetch(node_at_level, nid_off, (size_t)n * 4, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        b.finish();
        return AFV_OK;
    });
}
