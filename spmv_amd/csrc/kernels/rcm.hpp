// rcm.hpp -- reverse Cuthill-McKee on the device (round 4).
//
// The reference's reordering is METIS on the host (HyperGraphInterface.cpp:60-147, compiled out by default); round 1 put a host BFS RCM behind
// the same handle->index protocol (reorder/rcm.c) -- 3.3 s for 2e6 rows, ~20 s at config-2 size, which nobody would pay.  This is the same
// ordering idea as device kernels over the resident CSR:
//   graph      A + A^T without the diagonal; A^T's pattern is built once (column histogram -> scan -> fill), so the BFS can PUSH along both;
//   BFS        level-synchronous, queue-based.  Frontiers of at most kRcmSmall vertices -- every level of a banded matrix, the first and last
//              levels of a power-law one -- are expanded by ONE persistent 1024-thread workgroup that walks level after level inside a single
//              launch (a workgroup barrier per level instead of a kernel boundary: a scrambled band has ~m / bandwidth levels, 60 000 at 2e6
//              rows); it returns when the queue is empty or the frontier has outgrown it, and then whole-grid launches take over one level at a
//              time.  A vertex is claimed by atomicCAS on level[], so the LEVEL of every vertex is the BFS level whatever the claiming order;
//   start      a vertex of least degree, then (George-Liu, one round) a least-degree vertex of the last level of the BFS from it;
//   order      Cuthill-McKee's "neighbours by increasing degree" becomes the level-set form: vertices sorted by (level, degree, id) -- one
//              global bitonic sort of (64-bit key, id) pairs -- and the order reversed.  A function of the matrix alone: the permutation, and with it every
//              bit of every later product, is the same from handle to handle;
//   components the next unvisited vertex of least degree starts the next component (levels keep counting); isolated vertices all at once;
//   P A P^T    row lengths -> scan -> one wave per new row copies its entries and renames their columns.
#pragma once
#include "common.hpp"

namespace spmv {

constexpr int kRcmSmall = 8192;   // frontier sizes the one-workgroup kernel keeps for itself
constexpr int kRcmThreads = 1024;

// degree in A + A^T (diagonal and duplicates counted as they are stored: only the ORDER of degrees matters)
static __global__ __launch_bounds__(kBlock) void rcm_col_count_kernel(long long nnz, int m, const int *__restrict__ colidx, int *__restrict__ cnt)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long p = (long long) blockIdx.x * kBlock + threadIdx.x; p < nnz; p += stride) {
        const int c = colidx[p];
        if (c < m) atomicAdd(&cnt[c], 1);
    }
}

// A^T's pattern: trow[cursor[c]++] = r for every entry (r, c), c < m.  The order inside a column's list is whatever the atomics make it;
// nothing below depends on it (levels are BFS levels, the final order is sorted).
static __global__ __launch_bounds__(kBlock) void rcm_transpose_fill_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, int *__restrict__ cursor,
                                                                          int *__restrict__ trow)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave = ((long long) blockIdx.x * kBlock + threadIdx.x) / kWave, waves = (long long) gridDim.x * (kBlock / kWave);
    for (long long r = wave; r < m; r += waves)
        for (int p = rowptr[r] + lane; p < rowptr[r + 1]; p += kWave) {
            const int c = colidx[p];
            if (c < m) trow[atomicAdd(&cursor[c], 1)] = (int) r;
        }
}

// best = min over the unvisited vertices of (degree << 32 | id); degree = row length + column count
static __global__ __launch_bounds__(kBlock) void rcm_min_degree_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ tptr, const int *__restrict__ level,
                                                                      unsigned long long *__restrict__ best)
{
    unsigned long long b = ~0ull;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long v = (long long) blockIdx.x * kBlock + threadIdx.x; v < m; v += stride)
        if (level[v] < 0) {
            const unsigned long long d = (unsigned long long) (rowptr[v + 1] - rowptr[v]) + (unsigned long long) (tptr[v + 1] - tptr[v]);
            const unsigned long long k = (d << 32) | (unsigned long long) v;
            b = k < b ? k : b;
        }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(b, o, kWave); b = t < b ? t : b; }
    if ((threadIdx.x & (kWave - 1)) == 0 && b != ~0ull) atomicMin(best, b);
}

// the vertex of the LAST level (level == last) with least degree: the George-Liu restart
static __global__ __launch_bounds__(kBlock) void rcm_last_level_kernel(int m, int last, const int *__restrict__ rowptr, const int *__restrict__ tptr, const int *__restrict__ level,
                                                                      unsigned long long *__restrict__ best)
{
    unsigned long long b = ~0ull;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long v = (long long) blockIdx.x * kBlock + threadIdx.x; v < m; v += stride)
        if (level[v] == last) {
            const unsigned long long d = (unsigned long long) (rowptr[v + 1] - rowptr[v]) + (unsigned long long) (tptr[v + 1] - tptr[v]);
            const unsigned long long k = (d << 32) | (unsigned long long) v;
            b = k < b ? k : b;
        }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(b, o, kWave); b = t < b ? t : b; }
    if ((threadIdx.x & (kWave - 1)) == 0 && b != ~0ull) atomicMin(best, b);
}

static __global__ __launch_bounds__(kBlock) void rcm_fill_level_kernel(int m, int *__restrict__ level, int from_ge, int value)
{
    // level[v] >= from_ge -> value (from_ge = 0, value = -1: forget a BFS); from_ge = INT_MIN: every unvisited vertex of degree 0 ... see host
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long v = (long long) blockIdx.x * kBlock + threadIdx.x; v < m; v += stride)
        if (level[v] >= from_ge) level[v] = value;
}

// every still-unvisited vertex gets `value` (isolated vertices, or what is left when the component budget is spent); out[0] += how many
static __global__ __launch_bounds__(kBlock) void rcm_claim_rest_kernel(int m, int *__restrict__ level, int value, int only_isolated, const int *__restrict__ rowptr,
                                                                      const int *__restrict__ tptr, int *__restrict__ out)
{
    int n = 0;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long v = (long long) blockIdx.x * kBlock + threadIdx.x; v < m; v += stride)
        if (level[v] < 0 && (!only_isolated || (rowptr[v + 1] == rowptr[v] && tptr[v + 1] == tptr[v]))) { level[v] = value; ++n; }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) n += __shfl_xor(n, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && n) atomicAdd(out, n);
}

struct RcmState { // host <-> device hand-over of the BFS
    int count;    // vertices in the current frontier (queue `cur`)
    int lvl;      // their level
    int cur;      // which of the two queues holds them
    int visited;  // vertices with a level so far
};

// one wave per frontier vertex: claim its unvisited neighbours (A's row and A^T's row) for level lvl + 1 and append them to `next`
__device__ __forceinline__ void rcm_expand(int v, int lane, int lvl, const int *__restrict__ rowptr, const int *__restrict__ colidx, const int *__restrict__ tptr,
                                           const int *__restrict__ trow, int m, int *__restrict__ level, int *__restrict__ next, int *counter)
{
    for (int pass = 0; pass < 2; ++pass) {
        const int *__restrict__ ptr = pass ? tptr : rowptr;
        const int *__restrict__ idx = pass ? trow : colidx;
        for (int p = ptr[v] + lane; p < ptr[v + 1]; p += kWave) {
            const int c = idx[p];
            if (c < m && level[c] < 0 && atomicCAS(&level[c], -1, lvl + 1) == -1) next[atomicAdd(counter, 1)] = c;
        }
    }
}

// Small frontiers: ONE workgroup, level after level inside one launch.  Leaves when the frontier is empty (component done) or larger than
// kRcmSmall (the whole-grid kernel's turn).  Every wave leaves through the same test of a value read behind a barrier.
static __global__ __launch_bounds__(kRcmThreads) void rcm_bfs_small_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, const int *__restrict__ tptr,
                                                                          const int *__restrict__ trow, int *__restrict__ level, int *__restrict__ q0, int *__restrict__ q1,
                                                                          RcmState *__restrict__ st)
{
    __shared__ int s_next;
    int count = st->count, lvl = st->lvl, cur = st->cur, visited = st->visited;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    while (count > 0 && count <= kRcmSmall) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
        const int *__restrict__ q = cur ? q1 : q0;
        int *__restrict__ nq = cur ? q0 : q1;
        for (int i = wave; i < count; i += kRcmThreads / kWave) rcm_expand(q[i], lane, lvl, rowptr, colidx, tptr, trow, m, level, nq, &s_next);
        __threadfence_block();
        __syncthreads();
        count = s_next; // the same for every thread
        visited += count;
        lvl += 1;
        cur ^= 1;
        __syncthreads(); // s_next is reset at the top of the next round
    }
    if (threadIdx.x == 0) { st->count = count; st->lvl = lvl; st->cur = cur; st->visited = visited; }
}

// Large frontiers: the whole grid expands ONE level; the host reads the new count.
static __global__ __launch_bounds__(kBlock) void rcm_bfs_wide_kernel(int m, int count, int lvl, const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                                    const int *__restrict__ tptr, const int *__restrict__ trow, int *__restrict__ level,
                                                                    const int *__restrict__ q, int *__restrict__ nq, int *__restrict__ next_count)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave = ((long long) blockIdx.x * kBlock + threadIdx.x) / kWave, waves = (long long) gridDim.x * (kBlock / kWave);
    for (long long i = wave; i < count; i += waves) rcm_expand(q[i], lane, lvl, rowptr, colidx, tptr, trow, m, level, nq, next_count);
}

// sort key of vertex v: (level, degree) in 64 bits and the vertex number beside it; padding (v >= m) sorts behind everything
static __global__ __launch_bounds__(kBlock) void rcm_keys_kernel(long long n2, int m, const int *__restrict__ rowptr, const int *__restrict__ tptr, const int *__restrict__ level,
                                                                unsigned long long *__restrict__ key, unsigned *__restrict__ id)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long v = (long long) blockIdx.x * kBlock + threadIdx.x; v < n2; v += stride) {
        unsigned long long k = ~0ull;
        if (v < m) {
            const unsigned long long d = (unsigned long long) (rowptr[v + 1] - rowptr[v]) + (unsigned long long) (tptr[v + 1] - tptr[v]);
            k = ((unsigned long long) (unsigned) level[v] << 32) | (d > 0xffffffffull ? 0xffffffffull : d);
        }
        key[v] = k;
        id[v] = (unsigned) v;
    }
}

// one compare-exchange pass of the bitonic network over n2 = 2^k (key, id) pairs, ascending by key, then id
static __global__ __launch_bounds__(kBlock) void rcm_bitonic_kernel(long long n2, long long kk, long long jj, unsigned long long *__restrict__ key, unsigned *__restrict__ id)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
        const long long l = i ^ jj;
        if (l > i) {
            const unsigned long long a = key[i], b = key[l];
            const unsigned ia = id[i], ib = id[l];
            const bool gt = a > b || (a == b && ia > ib);
            const bool up = (i & kk) == 0;
            if (gt == up) { key[i] = b; key[l] = a; id[i] = ib; id[l] = ia; }
        }
    }
}

// perm[i] = vertex at sorted position m - 1 - i (the REVERSE of Cuthill-McKee); inv[perm[i]] = i; newlen[i] = its row length
static __global__ __launch_bounds__(kBlock) void rcm_perm_kernel(int m, const unsigned *__restrict__ id, const int *__restrict__ rowptr, int *__restrict__ perm,
                                                                int *__restrict__ inv, int *__restrict__ newlen)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
        const int v = (int) id[m - 1 - i];
        perm[i] = v;
        inv[v] = (int) i;
        newlen[i] = rowptr[v + 1] - rowptr[v];
    }
}

// row i of P A P^T = row perm[i] of A with its columns renamed (entry order kept); one wave per row
template <typename T>
__global__ __launch_bounds__(kBlock) void rcm_permute_kernel(int m, const int *__restrict__ perm, const int *__restrict__ inv, const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx, const T *__restrict__ val, const int *__restrict__ rp2,
                                                            int *__restrict__ ci2, T *__restrict__ va2)
{
    const int lane = threadIdx.x & (kWave - 1);
    const long long wave = ((long long) blockIdx.x * kBlock + threadIdx.x) / kWave, waves = (long long) gridDim.x * (kBlock / kWave);
    for (long long i = wave; i < m; i += waves) {
        const int src = rowptr[perm[i]], len = rowptr[perm[i] + 1] - src, dst = rp2[i];
        for (int k = lane; k < len; k += kWave) {
            const int c = colidx[src + k];
            ci2[dst + k] = c < m ? inv[c] : c;
            va2[dst + k] = val[src + k];
        }
    }
}

} // namespace spmv
