// k_match_mfma.hip — M1/M5 phase 1 on the matrix cores: the 256-bit Hamming distance as an EXACT integer contraction.
//
// Replaces the distance loop of the brute-force matchers (DescriptorDistance_orb32, Feature_orb32.cpp:67-84, called n1 x n2 times per
// keyframe pair from FeatureMatcher.cc:587-641) for whole descriptor sets; same output as k_match_topk (k_match.hip): every row's four
// smallest keys  distance << 16 | column — and, for free, up to three more: the record phase 2 reads holds the first seven keys of the
// merged top-4 lists of the two lane halves + how many of them are exact (4..7).
//
// popcount(a ^ b) over 256 bits is a dot product in disguise: with the bits of the train descriptor as +-64 and the bits of the query
// as -+64 (opposite signs), sum_k A_k B_k = 4096 * (#differing - #equal) = 8192 * d - 2^20.  i8 x i8 -> i32 is exact, so
// v_mfma_i32_32x32x32_i8 (8 instructions cover the 256 bits of 32 train x 32 query descriptors) delivers 1024 distances per 8
// instructions, and — because a ninth instruction contracts two more k-slots, (m & 63, m >> 6) of the train row against (1, 64) —
// directly the ORDERING KEY  8192 * d - 2^20 + m  (m < 8192 = the train descriptor's index: distance first, then position, exactly
// the order of  d << 16 | m).  The accumulator chain starts from the inline constant 0: no vector instruction prepares it.  In the C/D register layout of
// the 32x32 forms a lane owns ONE column (= one query) and 16 rows (= 16 train descriptors), so the top-4 insertion is the same
// lane-private  v_min + 3 v_med3  per value as in k_match_topk, with no cross-lane traffic; the VALU work per descriptor pair drops
// from 8 xor + 8 v_bcnt + 5 to 4 and runs beside the MFMA pipe.
//
// Workgroup = 4 wavefronts = 256 queries (64 per wavefront: two 32-column blocks, their B fragments live in registers for the whole
// kernel).  Train descriptors are expanded 64 at a time into LDS (bit -> +-64 byte through a 256-entry byte -> 8-byte table that also
// lives in LDS), double-buffered, one barrier per tile; every wavefront reads its A fragments with ds_read_b128 (row pitch 304 B =
// 256 expanded bytes + the 32 index slots + 16: 16 consecutive rows start in 16 different bank quads).
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

#define MQ_T 256
#define NO_KEY 0x7fffffff
#define T_TILE 64
#define A_PITCH 304

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// median of three, written as the min / max form the backend selects v_med3_i32 for.  NOT inline assembly: the hazard recogniser has to
// see these instructions to place the wait states a VALU read of a fresh MFMA result needs.
__device__ __forceinline__ int mq_med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// sorted insert into k[0] <= k[1] <= k[2] <= k[3]; the largest of the five falls out
__device__ __forceinline__ void mq_insert(int (&k)[4], int key) {
    const int n3 = mq_med3(k[2], k[3], key), n2 = mq_med3(k[1], k[2], key), n1 = mq_med3(k[0], k[1], key);
    k[0] = min(k[0], key);
    k[1] = n1;
    k[2] = n2;
    k[3] = n3;
}

// expand 64 train descriptors (rows tile_row0 .. +63, clamped to n2 - 1) into one LDS buffer
__device__ __forceinline__ void mq_stage(const uint32_t *__restrict__ train, int n2, int tile_row0, const uint2 *lut, uint8_t *buf, int tid) {
    uint32_t w[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + MQ_T * k, row = idx >> 3, wd = idx & 7;
        w[k] = train[(size_t)min(tile_row0 + row, n2 - 1) * 8 + wd];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + MQ_T * k, row = idx >> 3, wd = idx & 7;
        const uint2 e0 = lut[w[k] & 255u], e1 = lut[(w[k] >> 8) & 255u], e2 = lut[(w[k] >> 16) & 255u], e3 = lut[w[k] >> 24];
        uint4 *dst = reinterpret_cast<uint4 *>(buf + row * A_PITCH + wd * 32);
        dst[0] = make_uint4(e0.x, e0.y, e1.x, e1.y);
        dst[1] = make_uint4(e2.x, e2.y, e3.x, e3.y);
        if (wd == 0) {  // the index slots: k-slots 256, 257 = (m & 63, m >> 6), the other 30 zero
            const uint32_t mi = (uint32_t)(tile_row0 + row);
            uint4 *ix = reinterpret_cast<uint4 *>(buf + row * A_PITCH + 256);
            ix[0] = make_uint4((mi & 63u) | ((mi >> 6) << 8), 0u, 0u, 0u);
            ix[1] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

#define RKEYS 7  // key slots of a record (int4 x 2: seven keys + the exact-prefix length), as in k_match.hip
// Column slices (nslices > 1): a row has one record per slice, each over its own columns; the row's record is their merge - the seven
// smallest keys of the union, exact as far as the smallest of the slices' bounds (a slice's bound = its last exact key: every column
// of the slice that is not among its exact keys lies beyond it; a slice whose list ends inside the exact prefix has no other columns
// and no bound).  Same rule as the merge of the two lane halves at the end of the kernel.
__device__ __forceinline__ void mq_merge_record(int (&m)[RKEYS], int &bound, const int4 a, const int4 b) {
    const int k[RKEYS] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
    const int nk = min(max(b.w, 1), RKEYS);
    int last = k[0];
    bool complete = false;
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) {
        last = q < nk ? k[q] : last;
        complete = complete || (q < nk && k[q] == NO_KEY);
    }
    bound = min(bound, complete ? NO_KEY : last);
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) {  // sorted insert, the largest falls out
        const int key = k[q];
#pragma unroll
        for (int j = RKEYS - 1; j >= 1; --j) m[j] = mq_med3(m[j - 1], m[j], key);
        m[0] = min(m[0], key);
    }
}

template <bool PARTIAL>
__device__ __forceinline__ void mq_compute(const uint8_t *buf, const v4i (&bq)[2][8], const v4i &bidx, int (&kk)[2][4], int tile_row0, int n2,
                                           int lane) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        v4i a[9];
        const uint8_t *ap = buf + (sub * 32 + (lane & 31)) * A_PITCH + (lane >> 5) * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = *reinterpret_cast<const v4i *>(ap + t * 32);
        const int off = tile_row0 + sub * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[8], bidx, acc, 0, 0, 0);  // + m
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t], bq[qb][t], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int key = acc[r];
                // C/D register r holds train row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-row block
                if (PARTIAL) key = ((r & 3) + 8 * (r >> 2) + off < n2) ? key : NO_KEY;
                mq_insert(kk[qb], key);
            }
        }
    }
}

// SLICED = false: the batch form (one workgroup per row tile walks all column tiles; nslices is 1 and the slice arguments are unused) -
// kept as its own instantiation so that the column-slice logic costs the throughput-bound launches nothing
template <bool SLICED>
__global__ __launch_bounds__(MQ_T, 2) void k_match_topk_mfma(const uint8_t *__restrict__ desc, const int *__restrict__ nset, int cap,
                                                             const int *__restrict__ pair_a, const int *__restrict__ pair_b,
                                                             int4 *__restrict__ topk, int pair_base, int nslices_arg, int4 *__restrict__ slice_rec,
                                                             int *__restrict__ tickets) {
    const int nslices = SLICED ? nslices_arg : 1;
    __shared__ __attribute__((aligned(16))) uint2 s_lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_a[2][T_TILE * A_PITCH];
    const int p = pair_base + blockIdx.y;
    const int sa = pair_a[p], sb = pair_b[p];
    const int n1 = min(nset[sa], cap), n2 = min(nset[sb], cap);
    // nslices > 1 (one or a few pairs, the per-frame plugin call): the 64-column tiles of the train set are dealt to `nslices`
    // workgroups per row tile; every slice writes its own record per row, and the workgroup that finishes LAST for a row tile (a
    // ticket per pair and row tile) merges them into the record k_match_resolve reads
    const int rt = SLICED ? (int)blockIdx.x / nslices : (int)blockIdx.x, slice = SLICED ? (int)blockIdx.x - rt * nslices : 0;
    if (rt * MQ_T >= n1) return;  // uniform
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {   // byte -> eight +-64 bytes: bit j of the byte -> byte j (0x40 if set, 0xC0 = -64 if clear)
        const uint32_t lo = ((uint32_t)(tid & 15) * 0x00204081u) & 0x01010101u, hi = ((uint32_t)(tid >> 4) * 0x00204081u) & 0x01010101u;
        s_lut[tid] = make_uint2(0xC0C0C0C0u - lo * 0x80u, 0xC0C0C0C0u - hi * 0x80u);
    }
    int kk[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 4; ++i) kk[qb][i] = NO_KEY;
    const int row0 = rt * MQ_T + wv * 64;  // first query of this wavefront
    const int ntiles_all = (n2 + T_TILE - 1) / T_TILE, per_slice = (ntiles_all + nslices - 1) / nslices;
    const int tile0 = slice * per_slice, tile1 = min(tile0 + per_slice, ntiles_all);
    if (tile0 < tile1) {
        __syncthreads();  // table ready
        // B fragments: lane (n = lane & 31, g = lane >> 5) holds, for instruction t, the k-slots 32 t + 16 g .. + 15 = descriptor bytes
        // 4 t + 2 g, 4 t + 2 g + 1 of query row0 + 32 qb + n, signs flipped
        v4i bq[2][8];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const int q = min(row0 + 32 * qb + (lane & 31), n1 - 1);
            const uint4 *qp = reinterpret_cast<const uint4 *>(desc + ((size_t)sa * cap + q) * 32);
            const uint4 q0 = qp[0], q1 = qp[1];
            const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t h = (lane >> 5) ? (w[t] >> 16) : (w[t] & 0xffffu);
                const uint2 e0 = s_lut[h & 255u], e1 = s_lut[h >> 8];
                v4i v;
                v[0] = (int)(e0.x ^ 0x80808080u);
                v[1] = (int)(e0.y ^ 0x80808080u);
                v[2] = (int)(e1.x ^ 0x80808080u);
                v[3] = (int)(e1.y ^ 0x80808080u);
                bq[qb][t] = v;
            }
        }
        // B fragment of the index instruction: k-slots 256, 257 (lanes 0..31 hold them) carry the weights (1, 64)
        v4i bidx = {0, 0, 0, 0};
        if (lane < 32) bidx[0] = 1 | (64 << 8);
        const uint32_t *train = reinterpret_cast<const uint32_t *>(desc + (size_t)sb * cap * 32);
        const bool wave_has_rows = row0 < n1;
        mq_stage(train, n2, tile0 * T_TILE, s_lut, s_a[tile0 & 1], tid);
        __syncthreads();
        for (int tile = tile0; tile < tile1; ++tile) {
            if (tile + 1 < tile1) mq_stage(train, n2, (tile + 1) * T_TILE, s_lut, s_a[(tile + 1) & 1], tid);
            if (wave_has_rows) {
                if ((tile + 1) * T_TILE <= n2) mq_compute<false>(s_a[tile & 1], bq, bidx, kk, tile * T_TILE, n2, lane);
                else mq_compute<true>(s_a[tile & 1], bq, bidx, kk, tile * T_TILE, n2, lane);
            }
            __syncthreads();
        }
    }
    // The two lane halves hold the top-4 over DISJOINT halves of the train rows of the same queries.  Their merged list is exact as far
    // as the smaller of the two fourth keys t (every unseen column of a half is farther than that half's fourth key): the record is the
    // first seven merged keys + nk = the number of them <= t (4..7; a half with fewer than four columns has no unseen ones).
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(kk[qb][i], 32, 64);
        const int t = min(kk[qb][3], o[3]);
        int m[8] = {kk[qb][0], kk[qb][1], kk[qb][2], kk[qb][3], NO_KEY, NO_KEY, NO_KEY, NO_KEY};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = o[i];
#pragma unroll
            for (int j = 7; j >= 1; --j) m[j] = mq_med3(m[j - 1], m[j], key);
            m[0] = min(m[0], key);
        }
        int nk = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) nk += (m[j] != NO_KEY && m[j] <= t) ? 1 : 0;
        if (t == NO_KEY) nk = 7;  // both halves complete: every valid key is exact, the NO_KEY slots end the list
        const int row = row0 + 32 * qb + (lane & 31);
        if ((lane >> 5) == qb && row < n1) {
            int s[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int D = m[i] + (1 << 20);
                s[i] = m[i] == NO_KEY ? NO_KEY : (((D >> 13) << 16) | (D & 8191));
            }
            int4 *rec = SLICED ? slice_rec + (((size_t)p * cap + row) * nslices + slice) * 2 : topk + ((size_t)p * cap + row) * 2;
            rec[0] = make_int4(s[0], s[1], s[2], s[3]);
            rec[1] = make_int4(s[4], s[5], s[6], max(nk, 1));
        }
    }
    if (!SLICED) return;
    // ---- the last workgroup of the row tile merges the slices (cdna_hip_programming.md Guideline 16: every wave's stores drained by
    // the barrier -> ONE lane: agent-scope release, ticket; the last arriver: ONE agent-scope acquire -> barrier -> plain loads) ----
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        int *tk = tickets + (size_t)p * gridDim.x / nslices + rt;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // restated where the compiler cannot drop it (ROCm 7.2)
        const int old = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == nslices - 1;
        if (last) {
            __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch over this scratch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int row = rt * MQ_T + tid;
    if (row >= n1) return;
    int m[RKEYS];
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) m[q] = NO_KEY;
    int bound = NO_KEY;
    const int4 *rp = slice_rec + ((size_t)p * cap + row) * nslices * 2;
    for (int s0 = 0; s0 < nslices; s0 += 4) {  // four slices per step: their eight 16-byte loads are in flight together
        int4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = min(s0 + u, nslices - 1);
            a[u] = rp[2 * sl];
            b[u] = rp[2 * sl + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (s0 + u < nslices) mq_merge_record(m, bound, a[u], b[u]);
    }
    int nk = 0;
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) nk += (m[q] != NO_KEY && m[q] <= bound) ? 1 : 0;
    if (bound == NO_KEY) nk = RKEYS;  // every slice complete: all columns are in the list, the NO_KEY slots end it
    int4 *rec = topk + ((size_t)p * cap + row) * 2;
    rec[0] = make_int4(m[0], m[1], m[2], m[3]);
    rec[1] = make_int4(m[4], m[5], m[6], max(nk, 1));
}

extern "C" void afv_launch_match_topk_mfma(const uint8_t *desc, const int *nset, int cap, const int *pa, const int *pb, int npairs,
                                           void *topk_scratch, int pair_base, int nslices, void *slice_scratch, void *tickets, hipStream_t stream) {
    int4 *topk = reinterpret_cast<int4 *>(topk_scratch);
    if (nslices > 1)
        hipLaunchKernelGGL(k_match_topk_mfma<true>, dim3((cap + MQ_T - 1) / MQ_T * nslices, npairs), dim3(MQ_T), 0, stream, desc, nset, cap, pa, pb, topk,
                           pair_base, nslices, reinterpret_cast<int4 *>(slice_scratch), reinterpret_cast<int *>(tickets));
    else
        hipLaunchKernelGGL(k_match_topk_mfma<false>, dim3((cap + MQ_T - 1) / MQ_T, npairs), dim3(MQ_T), 0, stream, desc, nset, cap, pa, pb, topk,
                           pair_base, 1, static_cast<int4 *>(nullptr), static_cast<int *>(nullptr));
}
