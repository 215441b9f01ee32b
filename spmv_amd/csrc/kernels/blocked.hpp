// blocked.hpp -- row blocks x column slabs: the executor of the nnz-split family for matrices whose
// columns have NO locality (uniformly random / social-graph structure).
//
// Why.  When no x window of a tile group fits LDS (xwindows.hpp), every gather of x is a scattered
// 8-byte read.  x itself stays resident in the 256 MiB Infinity Cache, but each gather pulls a whole
// line across the fabric into an XCD's L2: 3.2e8 gathers = ~41 GB of line traffic, ~6 ms for a
// matrix whose own stream is 3.8 GB.  The reference meets the same wall on the CPU
// and does nothing about it; its Balanced2 workers own consecutive non-zeros
// (parallel_balanced2_spmv.c:41-53) and gather x wherever the columns point.
//
// Here, for that case only:
//   inspector   rows are cut into blocks of R rows, columns into slabs of W columns -- as narrow as a
//               table of 2^25 (block, slab) cells allows, down to 32 columns: the sweep over x is what
//               keeps x in L2, and narrow slabs additionally put gathers from one cache line into
//               neighbouring lanes, which merge into one L2 request.  The entries of a
//               row block are stored sorted by SLAB (counting sort on the device: histogram of
//               (block, slab) cells, scans, scatter), as three streams: value, global column, and
//               the 16-bit row number inside the block.  Block regions start at multiples of 8
//               entries so every load is a 16-byte load.
//   executor    one workgroup per row block.  y of the block lives in LDS (R * sizeof(T) = 64 KiB: two
//               workgroups per CU -- more resident workgroups drift apart in their slab position and
//               thrash L2: 32 KiB blocks ran 3.9 ms where 64 KiB blocks run 2.2 ms),
//               the workgroup walks the block's entries in stored order -- i.e. slab after slab, and
//               since workgroups are dispatched in order and blocks hold similar work, all
//               workgroups of an XCD gather from the same one or two slabs of x at a time, which
//               therefore stay in L2 -- and adds every product into y's LDS copy with an LDS
//               floating-point atomic.  At the end the block's y is written once, coalesced: no
//               partial sums, no carries, no read-modify-write of y in HBM.
//
// Measured (config 2 with uniformly random columns, fp64): 84 % of the L2 requests hit (5 % for the tile
// executors), 5.9 -> 2.0 ms; Orkut-style stand-in (74 nnz/row, 3e6 columns): 3.5 -> 1.13 ms.  What bounds it now is the L2's rate of random requests (~1.4e11/s over
// the chip; the LDS atomics are free: removing them changes nothing).
//
// The order in which the atomics of one row arrive is not fixed, so results are reproducible
// bit for bit only for exactly-representable data (the "eighths" fixtures); otherwise they vary in
// the last bits from run to run, within the parity tolerance.  That is why this executor is used
// only where it pays by a large factor (option "cache_block", default 1 = automatic) and never for
// matrices whose x windows can be staged.
#pragma once
#include <climits>
#include "common.hpp"

namespace spmv {

constexpr int kBlkThreads = 256;

// cell histogram: cnt[(r / R) * K + (c >> wshift)] += 1 for every entry; 16 lanes sweep a row
__global__ __launch_bounds__(kBlock) void blk_count_kernel(int m, int R, int K, int wshift, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, int *__restrict__ cnt)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < m; r += stride) {
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        const long long cell0 = (r / R) * K;
        for (int p = p0 + l; p < p1; p += 16) atomicAdd(&cnt[cell0 + (colidx[p] >> wshift)], 1);
    }
}

// tot[b] = entries of block b, rounded up to 8 (block regions start 16-byte aligned in every stream)
__global__ __launch_bounds__(kBlock) void blk_totals_kernel(int B, int K, const int *__restrict__ cnt, int *__restrict__ tot)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    int s = 0;
    for (int k = 0; k < K; ++k) s += cnt[(long long) b * K + k];
    tot[b] = (s + 7) & ~7;
}

// cursor[b][k] = first position of cell (b, k); end[b] = one past the block's last real entry
__global__ __launch_bounds__(kBlock) void blk_cells_kernel(int B, int K, const int *__restrict__ cnt, const long long *__restrict__ start,
                                                           long long *__restrict__ cursor, long long *__restrict__ end)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    long long p = start[b];
    for (int k = 0; k < K; ++k) {
        cursor[(long long) b * K + k] = p;
        p += cnt[(long long) b * K + k];
    }
    end[b] = p;
}

// scatter every entry to its cell (the order inside a cell is the order the atomics arrive in)
template <typename T>
__global__ __launch_bounds__(kBlock) void blk_fill_kernel(int m, int R, int K, int wshift, const int *__restrict__ rowptr,
                                                          const int *__restrict__ colidx, const T *__restrict__ val,
                                                          unsigned long long *__restrict__ cursor, T *__restrict__ bval,
                                                          int *__restrict__ bcol, unsigned short *__restrict__ brow)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < m; r += stride) {
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        const long long cell0 = (r / R) * K;
        const unsigned short rl = (unsigned short) (r % R);
        for (int p = p0 + l; p < p1; p += 16) {
            const int c = colidx[p];
            const unsigned long long pos = atomicAdd(&cursor[cell0 + (c >> wshift)], 1ull);
            bval[pos] = val[p];
            bcol[pos] = c;
            brow[pos] = rl;
        }
    }
}

__device__ __forceinline__ void lds_add(float *p, float v) { (void) unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void lds_add(double *p, double v) { (void) unsafeAtomicAdd(p, v); }

// Executor.  Dynamic LDS: R * sizeof(T) bytes (the block's y).
template <typename T, int NT = kBlkThreads>
__global__ __launch_bounds__(NT) void blk_kernel(int m, int R, const long long *__restrict__ start, const long long *__restrict__ end,
                                                          const T *__restrict__ bval, const int *__restrict__ bcol,
                                                          const unsigned short *__restrict__ brow, const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    T *ys = reinterpret_cast<T *>(blk_y_lds);
    constexpr int EPL = 16 / (int) sizeof(T); // entries per 16-byte value load
    constexpr int UN = 4;                     // load groups in flight per thread (2, 8, 16 measured no better)
    for (int i = threadIdx.x; i < R; i += NT) ys[i] = T(0);
    __syncthreads();
    const long long s = start[blockIdx.x], e = end[blockIdx.x];
    for (long long base = s + (long long) threadIdx.x * EPL; base < e; base += (long long) NT * EPL * UN) {
        T v[UN][EPL];
        int c[UN][EPL];
        unsigned rw[UN][EPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) { // every instruction reads one contiguous run over the wave
            const long long p = base + (long long) u * NT * EPL;
            if (p < e) {
                if constexpr (EPL == 2) {
                    const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(bval + p));
                    const i32x2 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(bcol + p));
                    const unsigned w = (unsigned) __builtin_nontemporal_load(reinterpret_cast<const int *>(brow + p));
                    v[u][0] = q.x; v[u][1] = q.y; c[u][0] = cc.x; c[u][1] = cc.y; rw[u][0] = w & 0xffffu; rw[u][1] = w >> 16;
                } else {
                    const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bval + p));
                    const i32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(bcol + p));
                    const i32x2 w = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(brow + p));
                    v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
                    c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
                    rw[u][0] = (unsigned) w.x & 0xffffu; rw[u][1] = (unsigned) w.x >> 16; rw[u][2] = (unsigned) w.y & 0xffffu; rw[u][3] = (unsigned) w.y >> 16;
                }
            }
        }
        T xv[UN][EPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = base + (long long) u * NT * EPL;
#pragma unroll
            for (int j = 0; j < EPL; ++j) xv[u][j] = p + j < e ? x[c[u][j]] : T(0); // cached loads: the slab stays in L2
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = base + (long long) u * NT * EPL;
#pragma unroll
            for (int j = 0; j < EPL; ++j)
                if (p + j < e) lds_add(&ys[rw[u][j]], v[u][j] * xv[u][j]);
        }
    }
    __syncthreads();
    const long long r0 = (long long) blockIdx.x * R;
    for (int i = threadIdx.x; i < R; i += NT)
        if (r0 + i < m) y[r0 + i] = ys[i];
}

} // namespace spmv
