// k_match_l2.hip — SURVEY §8f rank 3 / §8a M8: float-descriptor matcher (SIFT128, SURF64, KAZE64, R2D2-128 ...).
//
// Distance = cv::norm(a, b, NORM_L2SQR) as the reference calls it (Feature_sift128.cpp:132-134): float differences, squares
// and 4-way partial sums in double, one rounding to float at the end (OpenCV normL2Sqr<float,double>; OpenCV is absent:
// parity unpinned).  The summation order is part of the result, so this is NOT reformulated as a -2ab GEMM (no MFMA).
// Control flow = SearchByBoW(KF,KF) on a single node (FeatureMatcher.cc:561-660): rows in order, greedy.
//
// Two phases like the Hamming matcher (k_match.hip):
//  1. k_l2_topk — grid (row tiles of 64) x (column tiles): lane = row (its 128 floats live in registers), the block's four
//     waves split the column tile, columns are staged through LDS and read as wave-uniform ds_read_b128 broadcasts; every
//     (row, column tile) keeps its 4 best (distance bits, column) keys.  All pairs are touched exactly once: n1*n2*dim
//     float subs + 2*n1*n2*dim double ops, spread over >= 1024 waves.
//  2. k_l2_merge folds the tiles' keys per row (one wave per row); k_l2_resolve — one wave replays the rows in order in speculative
//     64-row rounds (claim table, stop at the first conflict); a row whose 4 keys are used up is rescanned exactly
//     (64 lanes over the columns).
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

#define L2T 256
#define L2K 4
#define L2_CHUNK 32
#define L2_NO_KEY 0xffffffffffffffffull
#define L2_MAX_SIDE 8192

#define L2_WAVE_SYNC()                                         \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

__device__ __forceinline__ void l2_insert(unsigned long long (&k)[L2K], unsigned long long key) {
#pragma unroll
    for (int s = 0; s < L2K; ++s) {
        if (key < k[s]) {
            const unsigned long long t = k[s];
            k[s] = key;
            key = t;
        }
    }
}

// row tile bx (64 rows) x column tile by of one job
template <int DIM>
__device__ __forceinline__ void l2_topk_body(const float *__restrict__ d1, int n1, const float *__restrict__ d2, int n2,
                                             const uint8_t *__restrict__ valid2, int cols_per_tile, int ntiles,
                                             unsigned long long *__restrict__ keys, int bx, int by) {
    __shared__ float4 s_col[L2_CHUNK * DIM / 4];
    __shared__ unsigned long long s_keys[L2T / 64][64][L2K];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = bx * 64 + lane;
    const int rowc = min(row, n1 - 1);
    float4 a[DIM / 4];
    const float4 *ap = reinterpret_cast<const float4 *>(d1 + (size_t)rowc * DIM);
#pragma unroll
    for (int i = 0; i < DIM / 4; ++i) a[i] = ap[i];
    const int c_begin = by * cols_per_tile, c_end = min(n2, c_begin + cols_per_tile);
    unsigned long long k[L2K] = {L2_NO_KEY, L2_NO_KEY, L2_NO_KEY, L2_NO_KEY};
    for (int cb = c_begin; cb < c_end; cb += L2_CHUNK) {
        const int nc = min(L2_CHUNK, c_end - cb);
        __syncthreads();
        const float4 *src = reinterpret_cast<const float4 *>(d2 + (size_t)cb * DIM);
        for (int i = tid; i < nc * (DIM / 4); i += L2T) s_col[i] = src[i];
        __syncthreads();
        for (int c = wv; c < nc; c += L2T / 64) {
            const int col = cb + c;
            if (valid2 && !valid2[col]) continue;  // wave-uniform
            double s = 0;
#pragma unroll
            for (int q = 0; q < DIM / 4; ++q) {
                const float4 b = s_col[c * (DIM / 4) + q];
                const double v0 = (double)(a[q].x - b.x), v1 = (double)(a[q].y - b.y), v2 = (double)(a[q].z - b.z),
                             v3 = (double)(a[q].w - b.w);
                s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
            }
            l2_insert(k, ((unsigned long long)__float_as_uint((float)s) << 32) | (unsigned)col);
        }
    }
#pragma unroll
    for (int s = 0; s < L2K; ++s) s_keys[wv][lane][s] = k[s];
    __syncthreads();
    if (wv == 0 && row < n1) {
#pragma unroll
        for (int w = 1; w < L2T / 64; ++w)
#pragma unroll
            for (int s = 0; s < L2K; ++s) l2_insert(k, s_keys[w][lane][s]);
#pragma unroll
        for (int s = 0; s < L2K; ++s) keys[((size_t)row * ntiles + by) * L2K + s] = k[s];
    }
}
template <int DIM>
__global__ __launch_bounds__(L2T) void k_l2_topk(const float *__restrict__ d1, int n1, const float *__restrict__ d2, int n2,
                                                 const uint8_t *__restrict__ valid2, int cols_per_tile, int ntiles,
                                                 unsigned long long *__restrict__ keys) {
    l2_topk_body<DIM>(d1, n1, d2, n2, valid2, cols_per_tile, ntiles, keys, blockIdx.x, blockIdx.y);
}
// batch form over a device-resident table desc[set][cap][DIM] with counts nset[set]: blockIdx.y = pair; one column tile per pair
// (the pairs fill the chip), so the keys need no merge
template <int DIM>
__global__ __launch_bounds__(L2T) void k_l2_topk_pairs(const float *__restrict__ desc, const int *__restrict__ nset, int cap,
                                                       const int *__restrict__ pa, const int *__restrict__ pb, int pair_base,
                                                       unsigned long long *__restrict__ keys) {
    const int p = pair_base + blockIdx.y, sa = pa[p], sb = pb[p];
    const int n1 = min(nset[sa], cap), n2 = min(nset[sb], cap);
    if ((int)blockIdx.x * 64 >= n1) return;  // uniform
    const int cols = ((max(n2, 1) + L2_CHUNK - 1) / L2_CHUNK) * L2_CHUNK;
    l2_topk_body<DIM>(desc + (size_t)sa * cap * DIM, n1, desc + (size_t)sb * cap * DIM, n2, nullptr, cols, 1,
                      keys + (size_t)blockIdx.y * cap * L2K, blockIdx.x, 0);
}

// one wave per row: fold the column tiles' keys into the row's 4 best (written over the row's first slot)
__global__ __launch_bounds__(L2T) void k_l2_merge(int n1, int ntiles, unsigned long long *__restrict__ keys) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * (L2T / 64) + (threadIdx.x >> 6);
    if (row >= n1) return;
    unsigned long long *rk = keys + (size_t)row * ntiles * L2K;
    unsigned long long k[L2K] = {L2_NO_KEY, L2_NO_KEY, L2_NO_KEY, L2_NO_KEY};
    for (int i = lane; i < ntiles * L2K; i += 64) l2_insert(k, rk[i]);
    unsigned long long best[L2K];
#pragma unroll
    for (int s = 0; s < L2K; ++s) {
        unsigned long long m = k[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = __shfl_xor(m, o, 64);
            m = t < m ? t : m;
        }
        best[s] = m;
        if (k[0] == m && m != L2_NO_KEY) {  // keys are unique (the column is part of the key): exactly one lane pops
            k[0] = k[1];
            k[1] = k[2];
            k[2] = k[3];
            k[3] = L2_NO_KEY;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < L2K; ++s) rk[s] = best[s];
    }
}

// exact distance from global memory (rescan path), same operation order as above
__device__ __forceinline__ float l2_exact(const float *a, const float *b, int dim) {
    double s = 0;
    for (int i = 0; i < dim; i += 4) {
        const double v0 = (double)(a[i] - b[i]), v1 = (double)(a[i + 1] - b[i + 1]), v2 = (double)(a[i + 2] - b[i + 2]),
                     v3 = (double)(a[i + 3] - b[i + 3]);
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    return (float)s;
}

__device__ __forceinline__ void l2_resolve_body(const float *__restrict__ d1, int n1, const float *__restrict__ d2, int n2,
                                                int dim, const uint8_t *__restrict__ valid1,
                                                const uint8_t *__restrict__ valid2, float th, float ratio, int ntiles,
                                                const unsigned long long *__restrict__ keys, int *__restrict__ out,
                                                int *__restrict__ nmatches) {
    __shared__ uint32_t s_matched[L2_MAX_SIDE / 32];
    __shared__ int s_claim[L2_MAX_SIDE];
    __shared__ int s_nvalid2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float FMAX = 3.402823466e+38f;
    for (int i = tid; i < (n2 + 31) / 32; i += L2T) s_matched[i] = 0;
    for (int i = tid; i < n2; i += L2T) s_claim[i] = 0x7fffffff;
    if (tid == 0) s_nvalid2 = 0;
    __syncthreads();
    int nv = 0;
    for (int i = tid; i < n2; i += L2T) nv += !valid2 || valid2[i];
    if (nv) atomicAdd(&s_nvalid2, nv);
    for (int row = tid; row < n1; row += L2T) out[row] = -1;
    __threadfence_block();
    __syncthreads();
    if (wv != 0) return;
    const bool lists_complete = s_nvalid2 <= L2K;
    int nm = 0, pos = 0;
    while (pos < n1) {
        const int q = pos + lane;
        const bool act = q < n1 && (!valid1 || valid1[q]);
        unsigned long long k[L2K] = {L2_NO_KEY, L2_NO_KEY, L2_NO_KEY, L2_NO_KEY};
        if (act) {
#pragma unroll
            for (int s = 0; s < L2K; ++s) k[s] = keys[(size_t)q * ntiles * L2K + s];
        }
        int e0 = -1, e1 = -1;
        float d0 = FMAX, d1v = FMAX, dlast = FMAX;
        bool open = act, exhausted = act;
#pragma unroll
        for (int s = 0; s < L2K; ++s) {
            if (open) {
                if (k[s] == L2_NO_KEY) {
                    open = false;
                    exhausted = false;
                } else {
                    const int idx = (int)(k[s] & 0xffffffffu);
                    dlast = __uint_as_float((uint32_t)(k[s] >> 32));
                    if (!((s_matched[idx >> 5] >> (idx & 31)) & 1u)) {
                        if (e0 < 0) {
                            e0 = idx;
                            d0 = dlast;
                        } else {
                            e1 = idx;
                            d1v = dlast;
                            open = false;
                            exhausted = false;
                        }
                    }
                }
            }
        }
        if (lists_complete) exhausted = false;
        int type = 0;  // 0 no match, 1 accept e0, 2 exact rescan
        if (act) {
            if (e0 >= 0 && !(d0 < th)) {
                e0 = -1;  // best unmatched column fails TH_LOW: final (the second cannot help)
                e1 = -1;
            } else if (exhausted) {
                // second best unknown but >= the last key's distance: accept early when the ratio test passes against that bound
                if (e0 >= 0 && d0 < ratio * dlast) type = 1;
                else type = 2;
            } else if (e0 >= 0) {
                if (d0 < ratio * d1v) type = 1;
            }
        }
        if (type == 1) atomicMin(&s_claim[e0], lane);
        L2_WAVE_SYNC();
        bool stopper = type == 2;
        if (act && type != 2) {
            if (e0 >= 0 && s_claim[e0] < lane) stopper = true;
            if (e1 >= 0 && s_claim[e1] < lane) stopper = true;
        }
        const unsigned long long sm = __ballot(stopper);
        const int stop = sm ? (int)__builtin_ctzll(sm) : 64;
        const bool commit = type == 1 && lane < stop;
        if (commit) {
            out[q] = e0;
            atomicOr(&s_matched[e0 >> 5], 1u << (e0 & 31));
        }
        nm += __popcll(__ballot(commit));
        if (type == 1) s_claim[e0] = 0x7fffffff;
        L2_WAVE_SYNC();
        if (stop == 0) {
            // exact rescan of row `pos` against the current matched set, 64 lanes over the columns
            const int q0 = pos;
            float bd = FMAX, bs = FMAX;
            int bp = 0x7fffffff;
            for (int c = lane; c < n2; c += 64) {
                if ((s_matched[c >> 5] >> (c & 31)) & 1u) continue;
                if (valid2 && !valid2[c]) continue;
                const float d = l2_exact(d1 + (size_t)q0 * dim, d2 + (size_t)c * dim, dim);
                if (d < bd || (d == bd && c < bp)) {
                    bs = bd;
                    bd = d;
                    bp = c;
                } else if (d < bs) {
                    bs = d;
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float od = __shfl_xor(bd, m, 64), os = __shfl_xor(bs, m, 64);
                const int op = __shfl_xor(bp, m, 64);
                if (od < bd || (od == bd && op < bp)) {
                    bs = fminf(os, bd);
                    bd = od;
                    bp = op;
                } else {
                    bs = fminf(bs, od);
                }
            }
            if (bp != 0x7fffffff && bd < th && bd < ratio * bs) {
                if (lane == 0) {
                    out[q0] = bp;
                    s_matched[bp >> 5] |= 1u << (bp & 31);
                }
                nm += 1;
            }
            L2_WAVE_SYNC();
            pos += 1;
        } else {
            pos += stop;
        }
    }
    if (lane == 0) *nmatches = nm;
}
__global__ __launch_bounds__(L2T) void k_l2_resolve(const float *__restrict__ d1, int n1, const float *__restrict__ d2, int n2,
                                                    int dim, const uint8_t *__restrict__ valid1,
                                                    const uint8_t *__restrict__ valid2, float th, float ratio, int ntiles,
                                                    unsigned long long *__restrict__ keys, int *__restrict__ out,
                                                    int *__restrict__ nmatches) {
    l2_resolve_body(d1, n1, d2, n2, dim, valid1, valid2, th, ratio, ntiles, keys, out, nmatches);
}
// one workgroup per pair; out[pair][cap] (rows >= n1 of a pair are set to -1 as well)
__global__ __launch_bounds__(L2T) void k_l2_resolve_pairs(const float *__restrict__ desc, const int *__restrict__ nset, int cap, int dim,
                                                          const int *__restrict__ pa, const int *__restrict__ pb, int pair_base, float th,
                                                          float ratio, const unsigned long long *__restrict__ keys, int *__restrict__ out,
                                                          int *__restrict__ nmatches) {
    const int p = pair_base + blockIdx.x, sa = pa[p], sb = pb[p];
    const int n1 = min(nset[sa], cap), n2 = min(nset[sb], cap);
    int *o = out + (size_t)p * cap;
    for (int row = n1 + (int)threadIdx.x; row < cap; row += L2T) o[row] = -1;
    l2_resolve_body(desc + (size_t)sa * cap * dim, n1, desc + (size_t)sb * cap * dim, n2, dim, nullptr, nullptr, th, ratio, 1,
                    keys + (size_t)blockIdx.x * cap * L2K, o, nmatches + p);
}

extern "C" size_t afv_match_l2_scratch_bytes(int n1, int n2, int *ntiles_out, int *cols_per_tile_out) {
    // enough (row tile, column tile) blocks to cover the chip: 256 CUs x 4 waves per block
    const int row_tiles = (n1 + 63) / 64;
    int ntiles = row_tiles > 0 ? (512 + row_tiles - 1) / row_tiles : 1;
    const int max_tiles = (n2 + L2_CHUNK - 1) / L2_CHUNK;
    if (ntiles > max_tiles) ntiles = max_tiles;
    if (ntiles < 1) ntiles = 1;
    int cols = (n2 + ntiles - 1) / ntiles;
    cols = ((cols + L2_CHUNK - 1) / L2_CHUNK) * L2_CHUNK;
    if (cols < L2_CHUNK) cols = L2_CHUNK;
    ntiles = n2 > 0 ? (n2 + cols - 1) / cols : 1;
    *ntiles_out = ntiles;
    *cols_per_tile_out = cols;
    return (size_t)(n1 > 0 ? n1 : 1) * ntiles * L2K * sizeof(unsigned long long);
}

// returns 0 when the tiled path does not apply (dim not 64 / 128): the caller falls back to the generic kernel
extern "C" int afv_launch_match_l2_tiled(const float *d1, int n1, const float *d2, int n2, int dim, const uint8_t *v1, const uint8_t *v2,
                                         float th, float ratio, int *out, int *nmatches, void *scratch, int ntiles, int cols_per_tile,
                                         hipStream_t stream) {
    if (dim != 64 && dim != 128) return 0;
    if (n1 <= 0 || n2 <= 0) return 0;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(scratch);
    dim3 grid((n1 + 63) / 64, ntiles);
    if (dim == 128) hipLaunchKernelGGL(k_l2_topk<128>, grid, dim3(L2T), 0, stream, d1, n1, d2, n2, v2, cols_per_tile, ntiles, keys);
    else hipLaunchKernelGGL(k_l2_topk<64>, grid, dim3(L2T), 0, stream, d1, n1, d2, n2, v2, cols_per_tile, ntiles, keys);
    hipLaunchKernelGGL(k_l2_merge, dim3((n1 + L2T / 64 - 1) / (L2T / 64)), dim3(L2T), 0, stream, n1, ntiles, keys);
    hipLaunchKernelGGL(k_l2_resolve, dim3(1), dim3(L2T), 0, stream, d1, n1, d2, n2, dim, v1, v2, th, ratio, ntiles, keys, out, nmatches);
    return 1;
}

// batch over a device-resident table (dim 64 / 128, cap <= L2_MAX_SIDE): pairs [pair_base, pair_base + npairs); `scratch` holds
// npairs * cap * L2K keys.  Returns 0 if the shape is not supported.
extern "C" int afv_launch_match_l2_pairs(const float *desc, const int *nset, int cap, int dim, const int *pa, const int *pb, int npairs,
                                         int pair_base, float th, float ratio, int *out, int *nmatches, void *scratch, hipStream_t stream) {
    if ((dim != 64 && dim != 128) || cap < 1 || cap > L2_MAX_SIDE || npairs < 1) return 0;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(scratch);
    const dim3 grid((cap + 63) / 64, npairs);
    if (dim == 128) hipLaunchKernelGGL(k_l2_topk_pairs<128>, grid, dim3(L2T), 0, stream, desc, nset, cap, pa, pb, pair_base, keys);
    else hipLaunchKernelGGL(k_l2_topk_pairs<64>, grid, dim3(L2T), 0, stream, desc, nset, cap, pa, pb, pair_base, keys);
    hipLaunchKernelGGL(k_l2_resolve_pairs, dim3(npairs), dim3(L2T), 0, stream, desc, nset, cap, dim, pa, pb, pair_base, th, ratio, keys, out,
                       nmatches);
    return 1;
}
