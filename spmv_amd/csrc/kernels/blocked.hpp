// blocked.hpp -- row blocks x column slabs: the executor for matrices whose columns have NO locality
// (uniformly random / social-graph structure), whatever the method asked for.
//
// Why.  When no x window of a tile group fits LDS (xwindows.hpp), every gather of x is a scattered
// 8-byte read.  x itself stays resident in the 256 MiB Infinity Cache, but each gather pulls a whole
// line across the fabric into an XCD's L2: 3.2e8 gathers = ~21 GB of line traffic, ~6 ms for a
// matrix whose own stream is 3.8 GB.  The reference meets the same wall on the CPU
// and does nothing about it; its Balanced2 workers own consecutive non-zeros
// (parallel_balanced2_spmv.c:41-53) and gather x wherever the columns point.
//
// Here, for that case only:
//   inspector   rows are cut into blocks of R rows, columns into slabs of W columns -- as narrow as a
//               table of 2^25 (block, slab) cells allows, down to 32 columns: the sweep over x is what
//               keeps x in L2, and narrow slabs additionally put gathers from one cache line into
//               neighbouring lanes, which merge into one L2 request.  The entries of a row block are
//               stored sorted by SLAB, and inside a (block, slab) cell in CSR order -- a STABLE counting
//               sort, done by ONE wavefront per block that walks the block's entries in CSR order and
//               ranks the 64 entries of a batch against each other with ballots (blk_fill_kernel), so the
//               stored order is a function of the matrix alone.  Three streams: value, global column,
//               16-bit row number inside the block; block regions start at multiples of 8 entries so
//               every load is a 16-byte load.
//   executor    ONE WAVEFRONT per row block (a 64-thread workgroup).  y of the block lives in LDS
//               (R doubles = 64 KiB, for fp32 values too -- see lds_add: two blocks per CU -- more resident blocks drift apart in their
//               slab position and thrash L2), the wave walks the block's entries in stored order -- slab
//               after slab; blocks are dispatched in order and hold similar work, so the blocks of an XCD
//               gather from the same few slabs of x at a time, which therefore stay in L2 -- with 16
//               gathers in flight per lane, and adds every product into y's LDS copy (ds_add).  At the end
//               the block's y is written once, coalesced: no partial sums, no carries, no read-modify-write
//               of y in HBM.
//   determinism every row is touched by exactly one wavefront, whose additions happen in program order
//               over a stream whose order is fixed by the inspector: the result is reproducible bit for
//               bit, run to run and handle to handle (the first version of this executor used a
//               256-thread workgroup per block, whose four waves raced on the rows: kept as variant 23).
//
// What bounds it is the L2's rate of random requests (one request per gather that is not merged with a
// neighbour's; ~1.4-1.7e11/s over the chip), not HBM: DESIGN.md 3.7 has the counters.
#pragma once
#include <climits>
#include "common.hpp"

namespace spmv {


// Row blocks of EQUAL WORK: block b holds rows [row0[b], row0[b + 1]).  All blocks of a round are resident together and the
// round lasts as long as its heaviest block (Orkut-style stand-in, fixed 5997-row blocks: heaviest / mean = 1.18; 4e6 rows of
// 2.6 entries: 1.40), so the cut points follow the entries, not the rows: work(r) = RowPtr[r] + c r (c entries of fixed cost per
// row, so that stretches of empty rows still end a block), each block takes 1 / (blocks left) of the work left, never more than
// rcap rows (its accumulators live in LDS).  One thread: B binary searches over RowPtr.  out[0] = blocks made (>= btarget when
// the row cap cut some short), out[1] = most rows in a block.
__global__ __launch_bounds__(kBlock) void blk_partition_kernel(int m, const int *__restrict__ rowptr, int btarget, int rcap, long long c,
                                                               int *__restrict__ cut /* [btarget + 1] scratch */, int *__restrict__ row0, int *__restrict__ out)
{
    // one workgroup.  Cut points of equal work, found independently: cut[b] = first row whose work reaches b / btarget of the total
    const long long total = (long long) rowptr[m] + c * m;
    for (int b = threadIdx.x; b <= btarget; b += kBlock) {
        const long long target = b == btarget ? total : (total / btarget) * b + ((total % btarget) * b) / btarget;
        int lo = 0, hi = m; // smallest r in [0, m] with work(r) >= target
        while (lo < hi) {
            const int mid = lo + ((hi - lo) >> 1);
            if ((long long) rowptr[mid] + c * mid >= target) hi = mid; else lo = mid + 1;
        }
        cut[b] = b == 0 ? 0 : (b == btarget ? m : lo);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // one thread: drop empty blocks (a single row heavier than a share), split blocks above the row cap
    int nb = 0, maxr = 0;
    for (int b = 0; b < btarget; ++b) {
        int start = cut[b];
        const int end = cut[b + 1];
        while (end - start > rcap) {
            row0[nb++] = start;
            maxr = rcap;
            start += rcap;
        }
        if (end > start) {
            row0[nb++] = start;
            maxr = end - start > maxr ? end - start : maxr;
        }
    }
    row0[nb] = m;
    out[0] = nb;
    out[1] = maxr;
}

// block of row r: the last b with row0[b] <= r
__device__ __forceinline__ int blk_of_row(const int *__restrict__ row0, int B, int r)
{
    int lo = 0, hi = B; // row0[lo] <= r < row0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row0[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// cell histogram: cnt[block(r) * K + (c >> wshift)] += 1 for every entry; 16 lanes sweep a row
__global__ __launch_bounds__(kBlock) void blk_count_kernel(int m, int B, const int *__restrict__ row0, int K, int wshift, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, int *__restrict__ cnt)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < m; r += stride) {
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        if (p0 == p1) continue;
        const long long cell0 = (long long) blk_of_row(row0, B, (int) r) * K;
        for (int p = p0 + l; p < p1; p += 16) atomicAdd(&cnt[cell0 + (colidx[p] >> wshift)], 1);
    }
}

// exclusive prefix sum of one value per lane over the wavefront
__device__ __forceinline__ long long wave_excl_scan(long long v, int lane, long long *total)
{
    long long inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const long long t = __shfl_up(inc, o, kWave);
        if (lane >= o) inc += t;
    }
    *total = __shfl(inc, kWave - 1, kWave);
    return inc - v;
}

// tot[b] = entries of block b, rounded up to 8 (block regions start 16-byte aligned in every stream); one wave per block
__global__ __launch_bounds__(kWave) void blk_totals_kernel(int B, int K, const int *__restrict__ cnt, int *__restrict__ tot)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    long long s = 0;
    for (int k = lane; k < K; k += kWave) s += cnt[(long long) b * K + k];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
    if (lane == 0) tot[b] = (int) ((s + 7) & ~7ll);
}

// cursor[b][k] = first position of cell (b, k); end[b] = one past the block's last real entry; one wave per block,
// every lane owns a contiguous run of cells
__global__ __launch_bounds__(kWave) void blk_cells_kernel(int B, int K, const int *__restrict__ cnt, const long long *__restrict__ start,
                                                          long long *__restrict__ cursor, long long *__restrict__ end)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int per = (K + kWave - 1) / kWave, k0 = lane * per, k1 = k0 + per < K ? k0 + per : K;
    long long mine = 0, total;
    for (int k = k0; k < k1; ++k) mine += cnt[(long long) b * K + k];
    long long p = start[b] + wave_excl_scan(mine, lane, &total);
    for (int k = k0; k < k1; ++k) {
        cursor[(long long) b * K + k] = p;
        p += cnt[(long long) b * K + k];
    }
    if (lane == 0) end[b] = start[b] + total;
}

// Stable rank of the 64 entries of a batch among the entries with the same key: rank = number of lower lanes with
// my key, cnt = lanes with my key, leader = the lowest of them.  One pass per distinct key of the batch.
__device__ __forceinline__ void batch_rank(bool valid, int key, int lane, int &rank, int &cnt, int &leader)
{
    unsigned long long todo = __ballot(valid);
    rank = 0; cnt = 0; leader = lane;
    while (todo) {
        const int l0 = __ffsll((long long) todo) - 1;
        const int k0 = __shfl(key, l0, kWave);
        const bool mine = valid && key == k0;
        const unsigned long long mk = __ballot(mine);
        if (mine) {
            rank = __popcll(mk & ((1ull << lane) - 1ull));
            cnt = __popcll(mk);
            leader = l0;
        }
        todo &= ~mk;
    }
}

// row of CSR position p inside the block whose row pointers are rp[0..nr]: the largest r with rp[r] <= p, searched
// upwards from `row` (p only grows along a lane)
__device__ __forceinline__ int row_of_position(const int *rp, int nr, int row, int p)
{
    int lo = row, hi = nr; // rp[lo] <= p < rp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rp[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// Stable scatter: ONE wavefront per row block walks the block's entries in CSR order, 64 at a time.  Inside a
// batch, entries of the same cell are ranked by lane (batch_rank); the first of each cell advances the cell's
// cursor by the cell's count in this batch.  Only this wave touches the cursors of its block and it does so batch
// after batch, so position = cell start + number of earlier entries (CSR order) of the same cell: the stored order
// does not depend on timing.  VALUES_ONLY: re-permute new values into the same positions
// (spmv_hip_update_values).  Dynamic LDS: (R + 1) ints, the block's row pointers.
// (Tried and dropped: a pre-pass that sorts the entries of a cell by x cache line, so that neighbouring lanes
// gather from the same line -- no change on any shape: 2.32 vs 2.31 ms, 1.178 vs 1.176 ms, +40 ms of inspector.
// The L1/TA path merges lanes of an instruction that hit the same line wherever they sit in the wave, and a cell
// is only a few lines wide.)
template <typename T, bool VALUES_ONLY>
__global__ __launch_bounds__(kWave) void blk_fill_kernel(const int *__restrict__ row0, int K, int wshift, const int *__restrict__ rowptr,
                                                         const int *__restrict__ colidx, const T *__restrict__ val,
                                                         unsigned long long *__restrict__ cursor, T *__restrict__ bval,
                                                         int *__restrict__ bcol, unsigned short *__restrict__ brow)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_fill_lds[];
    int *rp = reinterpret_cast<int *>(blk_fill_lds);
    const int lane = threadIdx.x;
    const long long r0 = row0[blockIdx.x];
    const int nr = row0[blockIdx.x + 1] - (int) r0;
    for (int i = lane; i <= nr; i += kWave) rp[i] = rowptr[r0 + i];
    __syncthreads();
    const int p0 = rp[0], p1 = rp[nr];
    unsigned long long *cur = cursor + (long long) blockIdx.x * K;
    int row = 0;
    for (int q = p0; q < p1; q += kWave) {
        const int p = q + lane;
        const bool valid = p < p1;
        const int c = valid ? colidx[p] : 0;
        const int cell = c >> wshift;
        if (valid) row = row_of_position(rp, nr, row, p);
        int rank, cnt, leader;
        batch_rank(valid, cell, lane, rank, cnt, leader);
        unsigned long long base = 0;
        if (valid && rank == 0) base = atomicAdd(&cur[cell], (unsigned long long) cnt); // L2 atomic: coherent batch to batch
        base = __shfl(base, leader, kWave);
        if (valid) {
            const unsigned long long pos = base + (unsigned long long) rank;
            bval[pos] = val[p];
            if constexpr (!VALUES_ONLY) {
                bcol[pos] = c;
                brow[pos] = (unsigned short) row;
            }
        }
    }
}

constexpr int kBlkPad = 65536; // zero entries behind the last block: two executor steps of the widest form (256 threads x 4 entries x 16 groups x 2)

// The block's y is accumulated in DOUBLE for both value types: ds_add_f32 runs at 2.0e11 adds/s over the chip whatever the
// bank pattern, ds_add_f64 at 7-9e11/s (tools/gbench lds, profiles/r02_gbench.txt) -- with float accumulators every fp32
// matrix ran at the LDS-atomic rate (0.51 ms for 9.6e7 entries, local columns or not).  fp32 products are rounded to
// float first (as every other executor forms them) and summed in double, which is at least as accurate as a float sum.
__device__ __forceinline__ void lds_add(double *p, double v) { (void) unsafeAtomicAdd(p, v); }

// Executor.  Dynamic LDS: R * 8 bytes (the block's y, in double).  NT = 64: one wavefront per block (deterministic);
// UN load groups of 16 bytes of values in flight per lane, and the NEXT step's stream loads are issued before
// this step's gathers are waited for.
//
// No load sits behind a branch: the streams are padded past the last block (blocked_fill) and a load group that
// starts past this block's end simply reads the next block's entries, whose products are masked at the LDS add.
// With "if (p < e) load" the compiler had put a wait after every load -- one gather in flight per wave, 0.8 us
// per 64 entries, and the kernel looked "L2-request bound" at a rate that was really one memory latency per
// instruction (round 1: 2.0 ms on config 2 with random columns).
template <typename T, int EPL>
struct BlkGroup {
    T v[EPL];
    int c[EPL];
    unsigned r[EPL];
};

template <typename T, int EPL>
__device__ __forceinline__ void blk_load(long long p, const T *__restrict__ bval, const int *__restrict__ bcol,
                                         const unsigned short *__restrict__ brow, BlkGroup<T, EPL> &g)
{
    if constexpr (EPL == 2) {
        const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(bval + p));
        const i32x2 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(bcol + p));
        const unsigned w = (unsigned) __builtin_nontemporal_load(reinterpret_cast<const int *>(brow + p));
        g.v[0] = q.x; g.v[1] = q.y; g.c[0] = cc.x; g.c[1] = cc.y; g.r[0] = w & 0xffffu; g.r[1] = w >> 16;
    } else {
        const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bval + p));
        const i32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(bcol + p));
        const i32x2 w = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(brow + p));
        g.v[0] = q.x; g.v[1] = q.y; g.v[2] = q.z; g.v[3] = q.w;
        g.c[0] = cc.x; g.c[1] = cc.y; g.c[2] = cc.z; g.c[3] = cc.w;
        g.r[0] = (unsigned) w.x & 0xffffu; g.r[1] = (unsigned) w.x >> 16; g.r[2] = (unsigned) w.y & 0xffffu; g.r[3] = (unsigned) w.y >> 16;
    }
}

template <typename T, int NT = kWave, int UN = 4>
__global__ __launch_bounds__(NT) void blk_kernel(const int *__restrict__ row0, int R, const long long *__restrict__ start, const long long *__restrict__ end,
                                                 const T *__restrict__ bval, const int *__restrict__ bcol,
                                                 const unsigned short *__restrict__ brow, const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    double *ys = reinterpret_cast<double *>(blk_y_lds);
    constexpr int EPL = 16 / (int) sizeof(T); // entries per 16-byte value load
    constexpr int STEP = NT * EPL;            // entries one load group covers over the workgroup
    for (int i = threadIdx.x; i < R; i += NT) ys[i] = 0.0; // R = the most rows of any block: masked entries of the next block may add 0 anywhere below it
    __syncthreads();
    const long long s = start[blockIdx.x], e = end[blockIdx.x];
    const long long lane0 = s + (long long) threadIdx.x * EPL;
    BlkGroup<T, EPL> cur[UN], nxt[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) blk_load<T, EPL>(lane0 + (long long) u * STEP, bval, bcol, brow, cur[u]);
    for (long long it = s; it < e; it += (long long) STEP * UN) { // wave-uniform trip count
        T xv[UN][EPL];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < EPL; ++j) xv[u][j] = x[cur[u].c[j]]; // cached loads: the slab stays in L2
        const long long pn = lane0 + (it - s) + (long long) STEP * UN;
#pragma unroll
        for (int u = 0; u < UN; ++u) blk_load<T, EPL>(pn + (long long) u * STEP, bval, bcol, brow, nxt[u]); // next step's stream (padded: always in bounds)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = lane0 + (it - s) + (long long) u * STEP;
#pragma unroll
            for (int j = 0; j < EPL; ++j) // unconditional: a masked entry adds 0 to a row of this block (its row number is the next block's or padding's, < R) -- a branch here makes the compiler sink gathers into it and wait for the whole queue
                lds_add(&ys[cur[u].r[j]], p + j < e ? (double) (cur[u].v[j] * xv[u][j]) : 0.0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) cur[u] = nxt[u];
    }
    __syncthreads();
    const long long r0 = row0[blockIdx.x];
    const int nr = row0[blockIdx.x + 1] - (int) r0;
    for (int i = threadIdx.x; i < nr; i += NT) y[r0 + i] = (T) ys[i];
}

// Three-stage form of the executor (the default): in step t the wave ISSUES the stream loads of step t + 2, ISSUES the
// gathers of step t + 1 (whose columns arrived during step t - 1 .. t) and ADDS step t (whose x values were gathered during
// step t - 1).  Nothing is waited for in the step that issued it: blk_kernel above gathers and adds in the same step and
// copies cur = nxt at its end, i.e. every step costs a gather round trip plus the rest of a stream round trip for 512
// entries per wave -- with two or three waves per CU that, not a bandwidth, was its rate.  Three stream register sets
// and two x sets are used in rotation (the loop body is written out for six consecutive steps), so no loaded register is
// ever copied.  Same additions in the same order as blk_kernel: bit-identical results.
template <typename T, int UN = 4>
__global__ __launch_bounds__(kWave) void blk_kernel3(const int *__restrict__ row0, int R, const long long *__restrict__ start, const long long *__restrict__ end,
                                                     const T *__restrict__ bval, const int *__restrict__ bcol,
                                                     const unsigned short *__restrict__ brow, const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    double *ys = reinterpret_cast<double *>(blk_y_lds);
    constexpr int EPL = 16 / (int) sizeof(T);
    constexpr int STEP = kWave * EPL;
    constexpr long long S = (long long) STEP * UN; // entries per step
    for (int i = threadIdx.x; i < R; i += kWave) ys[i] = 0.0;
    __syncthreads();
    const long long s = start[blockIdx.x], e = end[blockIdx.x];
    const long long lane0 = s + (long long) threadIdx.x * EPL;
    const int nsteps = (int) ((e - s + S - 1) / S); // wave-uniform
    BlkGroup<T, EPL> g0[UN], g1[UN], g2[UN];
    T x0[UN][EPL], x1[UN][EPL];
    auto load = [&](int t, BlkGroup<T, EPL>(&g)[UN]) { // stream of step t (padded: in bounds up to three steps past the last block)
#pragma unroll
        for (int u = 0; u < UN; ++u) blk_load<T, EPL>(lane0 + (long long) t * S + (long long) u * STEP, bval, bcol, brow, g[u]);
    };
    auto gather = [&](const BlkGroup<T, EPL>(&g)[UN], T(&xv)[UN][EPL]) { // columns past this block's end are the next block's or padding's: valid
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < EPL; ++j) xv[u][j] = x[g[u].c[j]];
    };
    auto add = [&](int t, const BlkGroup<T, EPL>(&g)[UN], const T(&xv)[UN][EPL]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = lane0 + (long long) t * S + (long long) u * STEP;
#pragma unroll
            for (int j = 0; j < EPL; ++j) lds_add(&ys[g[u].r[j]], p + j < e ? (double) (g[u].v[j] * xv[u][j]) : 0.0);
        }
    };
    if (nsteps > 0) {
        load(0, g0);
        load(1, g1);
        gather(g0, x0);
#define SPMV_BLK_PHASE(ga, gb, gc, xa, xb)                                                                              \
        load(t + 2, gc);                                                                                                \
        gather(gb, xb);                                                                                                 \
        add(t, ga, xa);                                                                                                 \
        if (++t >= nsteps) break;
        for (int t = 0;;) {
            SPMV_BLK_PHASE(g0, g1, g2, x0, x1)
            SPMV_BLK_PHASE(g1, g2, g0, x1, x0)
            SPMV_BLK_PHASE(g2, g0, g1, x0, x1)
            SPMV_BLK_PHASE(g0, g1, g2, x1, x0)
            SPMV_BLK_PHASE(g1, g2, g0, x0, x1)
            SPMV_BLK_PHASE(g2, g0, g1, x1, x0)
        }
#undef SPMV_BLK_PHASE
    }
    __syncthreads();
    const long long r0 = row0[blockIdx.x];
    const int nr = row0[blockIdx.x + 1] - (int) r0;
    for (int i = threadIdx.x; i < nr; i += kWave) y[r0 + i] = (T) ys[i];
}

} // namespace spmv
