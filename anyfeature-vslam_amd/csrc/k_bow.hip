// k_bow.hip — SURVEY §8f rank 2: BoW quantisation, the step right before SearchByBoW / SearchForTriangulation.
//
// Replaces DBoW2's TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) descent (upstream DBoW2; the
// reference calls it through Vocabulary::transform, src/Vocabulary.cpp:156-206, with levelsup = 4; Frame::ComputeBoW src/Frame.cc:397-401,
// KeyFrame::ComputeBoW src/KeyFrame.cc:65-73).  At each level the Hamming distance to every child of the current node, first minimum wins
// (strict <), until a node without children is reached.
//
// Round 5 shape (the round-1 kernel was a thread per descriptor with a serial child loop behind three dependent loads per level: child_ptr
// -> child_idx -> descriptor; 1000 descriptors = 4 workgroups on a 256-CU chip):
//   * a DPP row of 16 lanes per descriptor, four descriptors per wavefront, ONE wavefront per workgroup: 1000 descriptors = 250 workgroups,
//     a CU each;
//   * the k children of a node sit on the lanes of the row: their loads are in flight together, the argmin is four row_ror steps on DPP
//     (key = distance << 16 | child position: the first minimum wins);
//   * the tree is stored breadth first with the children of a node as CONSECUTIVE records, and a child's record carries, behind its
//     descriptor, {first child, number of children, DBoW2 id, rank of the id among the nodes of its depth}: what the next level needs arrives with the distance data, so the descent is
//     ONE dependent memory round trip per level (L = 6: six) instead of three.  The winner's triple moves to the row with three
//     ds_bpermute (a data-dependent lane: not expressible on DPP; one crossbar trip per level against a memory round trip).
// HBM-latency bound: 6 levels x (48 B x 10 children) = 2.9 KB per descriptor, the upper three levels (1 110 nodes, 53 KB) stay in L2, the
// lower ones come from the 53 MB image (k = 10, L = 6) through MALL / HBM.
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch
#include "afv_jobs.h"

#define BOW_NONE 0xffffffffu

__device__ __forceinline__ unsigned row_min_u32(unsigned v) {  // minimum over the 16 lanes of a DPP row, in every lane of the row
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false));
    return v;
}

template <int W>
__global__ __launch_bounds__(64) void k_bow_transform(DevVocab v, const uint32_t *__restrict__ desc, int n, int levelsup,
                                                      int *__restrict__ leaf_node, int *__restrict__ node_at_level, int *__restrict__ rank_at_level) {
    const int lane = threadIdx.x, sub = lane & 15, rowbase = lane & 48;
    const int i = blockIdx.x * 4 + (lane >> 4);
    const bool active = i < n;
    uint32_t q[W];
    {
        const uint4 *qp = reinterpret_cast<const uint4 *>(desc + (size_t)min(i, n - 1) * W);
#pragma unroll
        for (int w = 0; w < W / 4; ++w) {
            const uint4 t = qp[w];
            q[4 * w] = t.x, q[4 * w + 1] = t.y, q[4 * w + 2] = t.z, q[4 * w + 3] = t.w;
        }
    }
    const int nid_level = v.L - levelsup;
    const int RD = v.rec_dwords;
    const uint4 root = *reinterpret_cast<const uint4 *>(v.rec + W);  // {first child, #children, id, 0} of record 0
    int cur_base = (int)root.x, cur_n = active ? (int)root.y : 0;
    int final_id = 0, nid = 0, nrank = -1, level = 0;
    while (__any(cur_n > 0)) {
        unsigned best = BOW_NONE;
        int nb = 0, nn = 0, nd = 0, nr = 0;
        for (int c0 = 0; __any(c0 < cur_n); c0 += 16) {
            const int c = c0 + sub;
            const bool has = c < cur_n;
            unsigned key = BOW_NONE;
            uint4 info = make_uint4(0, 0, 0, 0);
            if (has) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(v.rec + (size_t)(cur_base + c) * RD);
                uint4 d4[W / 4];
#pragma unroll
                for (int w = 0; w < W / 4; ++w) d4[w] = rp[w];
                info = rp[W / 4];
                int d = 0;
#pragma unroll
                for (int w = 0; w < W / 4; ++w)
                    d += __popc(q[4 * w] ^ d4[w].x) + __popc(q[4 * w + 1] ^ d4[w].y) + __popc(q[4 * w + 2] ^ d4[w].z) + __popc(q[4 * w + 3] ^ d4[w].w);
                key = ((unsigned)d << 16) | (unsigned)c;
            }
            const unsigned m = row_min_u32(key);
            // the winner of this chunk sits on lane rowbase + (its position - c0); rows without a child here read lane rowbase (ignored)
            const int src = (rowbase + (m == BOW_NONE ? 0 : (int)(m & 0xffffu) - c0)) * 4;
            const int wb = __builtin_amdgcn_ds_bpermute(src, (int)info.x);
            const int wn = __builtin_amdgcn_ds_bpermute(src, (int)info.y);
            const int wd = __builtin_amdgcn_ds_bpermute(src, (int)info.z);
            const int wr = __builtin_amdgcn_ds_bpermute(src, (int)info.w);
            if (m < best) {  // keys carry the child position: a later chunk only wins with a strictly smaller distance
                best = m;
                nb = wb;
                nn = wn;
                nd = wd;
                nr = wr;
            }
        }
        if (cur_n > 0) {
            ++level;
            final_id = nd;
            if (level == nid_level) {
                nid = nd;
                nrank = nr;
            }
            cur_base = nb;
            cur_n = nn;
        }
    }
    if (active && sub == 0) {
        leaf_node[i] = final_id;
        node_at_level[i] = nid_level <= 0 ? 0 : nid;
        // sort key of the FeatureVector: 0 = the root (levelsup >= L, or a leaf above that depth: DBoW2 leaves *nid = 0), else 1 + the rank
        // of the node among the nodes of its depth by DBoW2 id
        if (rank_at_level) rank_at_level[i] = nid_level <= 0 ? 0 : nrank + 1;
    }
}

// ---------------- float descriptors (SIFT128, SURF64, KAZE64, R2D2 ...: the non-binary cases of Vocabulary::transform, Vocabulary.cpp:158-187) ----------------
// Same tree image, the node descriptor as `dim` floats.  Distance = DBoW2's float descriptor classes (upstream FSurf64::distance): the
// squared differences, each evaluated in FLOAT ((a - b) * (a - b)), accumulated in DOUBLE in index order - so the sum of a child is one
// lane's sequential loop (splitting it over lanes would change the roundings), the children of a node sit on the 16 lanes of a row as in
// the binary kernel, and the four query descriptors of a wavefront are read from LDS (a row reads one address: a broadcast).
// First minimum wins (strict <).  Parity: unpinned (the reference's DBoW2 fork with the float classes is an empty submodule).
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v) {
#define AFV_ROWMIN64(ctrl)                                                                                                  \
    {                                                                                                                       \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)v, (int)(unsigned)v, ctrl, 0xf, 0xf, false);               \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(v >> 32), (int)(unsigned)(v >> 32), ctrl, 0xf, 0xf, false); \
        const unsigned long long t_ = ((unsigned long long)hi_ << 32) | lo_;                                                \
        v = t_ < v ? t_ : v;                                                                                                \
    }
    AFV_ROWMIN64(0x121) AFV_ROWMIN64(0x122) AFV_ROWMIN64(0x124) AFV_ROWMIN64(0x128)
#undef AFV_ROWMIN64
    return v;
}

template <int DIM>
__global__ __launch_bounds__(64) void k_bow_transform_f32(DevVocab v, const float *__restrict__ desc, int n, int levelsup, int *__restrict__ leaf_node,
                                                          int *__restrict__ node_at_level, int *__restrict__ rank_at_level) {
    __shared__ __attribute__((aligned(16))) float s_q[4][DIM];
    const int lane = threadIdx.x, sub = lane & 15, row = lane >> 4, rowbase = lane & 48;
    const int i = blockIdx.x * 4 + row;
    const bool active = i < n;
    for (int e = sub; e < DIM; e += 16) s_q[row][e] = desc[(size_t)min(i, n - 1) * DIM + e];
    __syncthreads();
    const int nid_level = v.L - levelsup;
    const int RD = v.rec_dwords;  // DIM + 4
    const uint4 root = *reinterpret_cast<const uint4 *>(v.rec + DIM);
    int cur_base = (int)root.x, cur_n = active ? (int)root.y : 0;
    int final_id = 0, nid = 0, nrank = -1, level = 0;
    while (__any(cur_n > 0)) {
        unsigned long long best = ~0ull;
        int best_c = 0x7fffffff, nb = 0, nn = 0, nd = 0, nr = 0;
        for (int c0 = 0; __any(c0 < cur_n); c0 += 16) {
            const int c = c0 + sub;
            const bool has = c < cur_n;
            unsigned long long key = ~0ull;
            uint4 info = make_uint4(0, 0, 0, 0);
            if (has) {
                const float4 *rp = reinterpret_cast<const float4 *>(v.rec + (size_t)(cur_base + c) * RD);
                double acc = 0.0;
                for (int e = 0; e < DIM / 4; ++e) {
                    const float4 b = rp[e];
                    const float4 a = *reinterpret_cast<const float4 *>(&s_q[row][4 * e]);
                    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
                    acc += (double)(d0 * d0);
                    acc += (double)(d1 * d1);
                    acc += (double)(d2 * d2);
                    acc += (double)(d3 * d3);
                }
                info = *reinterpret_cast<const uint4 *>(v.rec + (size_t)(cur_base + c) * RD + DIM);
                key = (unsigned long long)__double_as_longlong(acc);  // a sum of squares: non-negative, its bit pattern orders like its value
            }
            const unsigned long long m = row_min_u64(key);
            // the FIRST child of this chunk with that distance: smallest position among the lanes that hold it
            const unsigned first = row_min_u32((has && key == m) ? (unsigned)c : BOW_NONE);
            const int src = (rowbase + (first == BOW_NONE ? 0 : (int)first - c0)) * 4;
            const int wb = __builtin_amdgcn_ds_bpermute(src, (int)info.x);
            const int wn = __builtin_amdgcn_ds_bpermute(src, (int)info.y);
            const int wd = __builtin_amdgcn_ds_bpermute(src, (int)info.z);
            const int wr = __builtin_amdgcn_ds_bpermute(src, (int)info.w);
            if (first != BOW_NONE && m < best) {  // a later chunk only wins with a strictly smaller distance
                best = m;
                best_c = (int)first;
                nb = wb;
                nn = wn;
                nd = wd;
                nr = wr;
            }
        }
        (void)best_c;
        if (cur_n > 0) {
            ++level;
            final_id = nd;
            if (level == nid_level) {
                nid = nd;
                nrank = nr;
            }
            cur_base = nb;
            cur_n = nn;
        }
    }
    if (active && sub == 0) {
        leaf_node[i] = final_id;
        node_at_level[i] = nid_level <= 0 ? 0 : nid;
        if (rank_at_level) rank_at_level[i] = nid_level <= 0 ? 0 : nrank + 1;
    }
}

extern "C" int afv_launch_bow_transform_f32(const DevVocab *v, const float *desc, int n, int dim, int levelsup, int *leaf_node, int *node_at_level,
                                            int *rank_at_level, hipStream_t stream) {
    if (n <= 0) return 1;
    dim3 grid((n + 3) / 4);
    if (dim == 128) hipLaunchKernelGGL(k_bow_transform_f32<128>, grid, dim3(64), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level, rank_at_level);
    else if (dim == 64) hipLaunchKernelGGL(k_bow_transform_f32<64>, grid, dim3(64), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level, rank_at_level);
    else if (dim == 256) hipLaunchKernelGGL(k_bow_transform_f32<256>, grid, dim3(64), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level, rank_at_level);
    else return 0;
    return 1;
}

extern "C" void afv_launch_bow_transform(const DevVocab *v, const uint32_t *desc, int n, int levelsup, int *leaf_node,
                                         int *node_at_level, int *rank_at_level, hipStream_t stream) {
    if (n <= 0) return;
    dim3 grid((n + 3) / 4);
    if (v->words == 8) hipLaunchKernelGGL(k_bow_transform<8>, grid, dim3(64), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level, rank_at_level);
    else hipLaunchKernelGGL(k_bow_transform<16>, grid, dim3(64), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level, rank_at_level);
}
