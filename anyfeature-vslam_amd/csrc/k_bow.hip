// k_bow.hip — SURVEY §8f rank 2: BoW quantisation, the step right before SearchByBoW / SearchForTriangulation.
//
// Replaces DBoW2's TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) descent (upstream DBoW2; the
// reference calls it through Vocabulary::transform, src/Vocabulary.cpp:156-206, with levelsup = 4).  One thread per
// descriptor: at each level the Hamming distance to every child of the current node (8 xor + 8 v_bcnt per child), first
// minimum wins (strict <), until a node without children is reached.  Embarrassingly parallel, read-only tree.
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch
#include "afv_jobs.h"


template <int W>
__global__ __launch_bounds__(256) void k_bow_transform(DevVocab v, const uint32_t *__restrict__ desc, int n, int levelsup,
                                                       int *__restrict__ leaf_node, int *__restrict__ node_at_level) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t q[W];
#pragma unroll
    for (int w = 0; w < W; ++w) q[w] = desc[(size_t)i * W + w];
    const int nid_level = v.L - levelsup;
    int final_id = 0, level = 0, nid = 0;
    int b = v.child_ptr[0], e = v.child_ptr[1];
    while (e > b) {
        ++level;
        int best = -1, best_d = 0x7fffffff;
        for (int c = b; c < e; ++c) {
            const int id = v.child_idx[c];
            const uint32_t *nd = v.desc + (size_t)id * W;
            int d = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) d += __popc(q[w] ^ nd[w]);
            if (d < best_d) {
                best_d = d;
                best = id;
            }
        }
        final_id = best;
        if (level == nid_level) nid = final_id;
        b = v.child_ptr[final_id];
        e = v.child_ptr[final_id + 1];
    }
    leaf_node[i] = final_id;
    node_at_level[i] = nid_level <= 0 ? 0 : nid;
}

extern "C" void afv_launch_bow_transform(const DevVocab *v, const uint32_t *desc, int n, int levelsup, int *leaf_node,
                                         int *node_at_level, hipStream_t stream) {
    if (n <= 0) return;
    dim3 grid((n + 255) / 256);
    if (v->words == 8) hipLaunchKernelGGL(k_bow_transform<8>, grid, dim3(256), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level);
    else hipLaunchKernelGGL(k_bow_transform<16>, grid, dim3(256), 0, stream, *v, desc, n, levelsup, leaf_node, node_at_level);
}
