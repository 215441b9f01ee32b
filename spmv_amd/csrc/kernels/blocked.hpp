// blocked.hpp -- row blocks x column slabs: the executor for matrices whose columns have NO locality
// (uniformly random / social-graph structure), whatever the method asked for.
//
// Why.  When no x window of a tile group fits LDS (xwindows.hpp), every gather of x is a scattered read.  x itself
// stays resident in the 256 MiB Infinity Cache, but each gather pulls a whole line across the fabric into an XCD's
// L2: 3.2e8 gathers = ~21 GB of line traffic, ~6 ms for a matrix whose own stream is 3.8 GB.  The
// reference meets the same wall on the CPU and does nothing about it; its Balanced2 workers own consecutive
// non-zeros (parallel_balanced2_spmv.c:41-53) and gather x wherever the columns point.
//
// Here, for that case only (format of round 3; DESIGN.md 3.7 has the measurements):
//   layout      rows are cut into blocks of at most R rows of EQUAL WORK (blk_partition_kernel), columns into slabs of
//               2^wshift columns (128 unless that would make more than 32768 slabs).  The entries of a row block are
//               stored sorted by slab, in GROUPS of 64 lanes x 16 bytes of values (128 fp64 / 256 fp32 entries), as two
//               streams: the value and a 32-bit word (16-bit column offset | 16-bit row inside the block) -- 12 bytes per
//               fp64 entry, the bytes of plain CSR.  A block's region has two parts:
//                 dense   the (block, slab) cells with at least one group's worth of entries, back to back.  A group
//                         touches at most two cells, so its header names two slabs (A: the slab of its first entry, B:
//                         the next dense slab); the executor reads both slabs of x with COALESCED loads into LDS and
//                         the entries pick their x there -- the column field is (A or B, offset in the slab).  No
//                         scattered load at all: a scattered load occupies the CU's vector-memory path lane by lane
//                         and the matrix stream queues behind it, a coalesced one for a few cycles.
//                 sparse  everything else, super-slab (65536 columns) after super-slab, each super-slab's run padded
//                         to whole groups, so that a group lies in ONE super-slab: its header is the super-slab's
//                         first column (wave-uniform: a scalar load, the gather is base + 16-bit offset) and x is
//                         gathered through L2 as before -- the sweep over the slabs, in step over the blocks of an XCD,
//                         is what keeps x in L2.
//   inspector   (device) one 1024-thread workgroup per block counts the block's cells in LDS (blk_count_kernel); a
//               layout pass turns counts into part offsets; then the fill (blk_fill_kernel): the cells of a block are
//               split into up to 16 column ranges and ONE WAVEFRONT per (block, range) scans the block's entries in CSR
//               order, keeps those of its range (compacted through a small LDS queue, order kept) and gives each the
//               next position of its cell from a cursor in LDS.  Only that wave touches those cursors and it does so in
//               CSR order, so the stored order -- (part, slab, CSR order) -- is a function of the matrix alone.
//   executor    ONE WAVEFRONT per row block (a 64-thread workgroup, two blocks per CU).  y of the block lives in LDS
//               as doubles (for fp32 values too -- see lds_add); the wave walks first the dense groups, then the sparse
//               ones, three steps in flight (stream loads of step t + 2, x of step t + 1, additions of step t), and
//               adds every product into y's LDS copy (ds_add_f64).  At the end the block's y is written once,
//               coalesced: no partial sums, no carries, no read-modify-write of y in HBM.  No load sits behind a
//               branch: groups past a part's end are read (the streams are padded) and their products go to a junk
//               accumulator.
//   determinism every row is touched by exactly one wavefront, whose additions happen in program order over a stream
//               whose order is fixed by the inspector: the result is reproducible bit for bit, run to run and handle
//               to handle.
#pragma once
#include <climits>
#include "common.hpp"

namespace spmv {

constexpr int kBlkSlabShift = 7;     // dense cells: slabs of 128 columns (1 KiB of fp64 x, two 512-byte load instructions)
constexpr int kBlkSuperShift = 16;   // sparse entries: 16-bit column offsets inside super-slabs of 65536 columns
constexpr int kBlkMaxCells = 32768;  // cells of one row block at most (the count kernel's histogram: 128 KiB of LDS)
constexpr int kBlkParts = 16;        // column ranges the fill of one block is split over (one wavefront each)
constexpr int kBlkCountThreads = 1024;
constexpr int kBlkPadGroups = 128;   // zero groups behind the last block: three executor steps of the widest form, and header loads reach 64 groups ahead
constexpr int kBlkDenseUn = 8;       // groups per step of the dense loop (LDS: 2 slabs x 128 columns each)

struct BlkDir {      // one row block's region: groups [g0, g0 + nd) dense, [g0 + nd, g0 + nd + ns) sparse
    long long g0;
    int nd, ns;
};

// Row blocks of EQUAL WORK: block b holds rows [row0[b], row0[b + 1]).  All blocks of a round are resident together and the
// round lasts as long as its heaviest block (Orkut-style stand-in, fixed 5997-row blocks: heaviest / mean = 1.18; 4e6 rows of
// 2.6 entries: 1.40), so the cut points follow the entries, not the rows: work(r) = RowPtr[r] + c r (c entries of fixed cost per
// row, so that stretches of empty rows still end a block), each block takes 1 / (blocks left) of the work left, never more than
// rcap rows (its accumulators live in LDS).  One workgroup: B binary searches over RowPtr.  out[0] = blocks made (>= btarget when
// the row cap cut some short), out[1] = most rows in a block.
__global__ __launch_bounds__(kBlock) void blk_partition_kernel(int m, const int *__restrict__ rowptr, int btarget, int rcap, long long c,
                                                               int *__restrict__ cut /* [btarget + 1] scratch */, int *__restrict__ row0, int *__restrict__ out)
{
    // cut points of equal work, found independently: cut[b] = first row whose work reaches b / btarget of the total
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

// Sum over the wavefront, every lane gets the total.
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// exclusive prefix sum of one value per lane over the wavefront
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total)
{
    int inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(inc, o, kWave);
        if (lane >= o) inc += t;
    }
    *total = __shfl(inc, kWave - 1, kWave);
    return inc - v;
}

// super-slabs [s0, s1) of column range `part` out of `nparts` (S super-slabs in all)
__device__ __forceinline__ void blk_part_range(int S, int nparts, int part, int &s0, int &s1)
{
    s0 = (int) (((long long) S * part) / nparts);
    s1 = (int) (((long long) S * (part + 1)) / nparts);
}

// Inspector pass 1: the cells of row block blockIdx.x, counted in LDS (K counters), written to cnt[b][K]; and per
// column range (part) of the block the entries in dense cells (count >= dense_min) and the entries in sparse cells,
// the latter with every super-slab's run rounded up to whole groups of 2^ge entries -- parts[b][part] = (dense, sparse).
__global__ __launch_bounds__(kBlkCountThreads) void blk_count_kernel(const int *__restrict__ row0, int K, int wshift, int dense_min, int ge, int nparts,
                                                                     int S, const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                                                     int *__restrict__ cnt, i32x2 *__restrict__ parts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_count_lds[];
    unsigned *cells = reinterpret_cast<unsigned *>(blk_count_lds);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < K; i += kBlkCountThreads) cells[i] = 0u;
    __syncthreads();
    const int p0 = rowptr[row0[b]], p1 = rowptr[row0[b + 1]];
    for (int p = p0 + (int) threadIdx.x; p < p1; p += kBlkCountThreads) atomicAdd(&cells[ld_stream(colidx + p) >> wshift], 1u);
    __syncthreads();
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int cs = kBlkSuperShift - wshift; // cells per super-slab = 2^cs
    int *out = cnt + (long long) b * K;
    for (int part = wave; part < nparts; part += kBlkCountThreads / kWave) {
        int s0, s1;
        blk_part_range(S, nparts, part, s0, s1);
        int dsum = 0, spad = 0;
        for (int s = s0; s < s1; ++s) {
            const int ka = s << cs, kb = min((s + 1) << cs, K);
            int d = 0, sp = 0;
            for (int k = ka + lane; k < kb; k += kWave) {
                const int c = (int) cells[k];
                out[k] = c;
                if (c >= dense_min) d += c; else sp += c;
            }
            dsum += wave_sum(d);
            spad += ((wave_sum(sp) + (1 << ge) - 1) >> ge) << ge;
        }
        if (lane == 0) parts[(long long) b * kBlkParts + part] = i32x2{dsum, spad};
    }
}

// Inspector pass 2: per block, where each part's dense and sparse entries start inside the block's region (in entries:
// dense parts back to back from 0, the dense total rounded up to a whole group, then the sparse parts), and the block's
// group counts.  One thread per block.
__global__ __launch_bounds__(kBlock) void blk_layout_kernel(int B, int nparts, int ge, const i32x2 *__restrict__ parts, i32x2 *__restrict__ part_off,
                                                            int *__restrict__ groups, int *__restrict__ dense_groups)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    int d = 0;
    for (int w = 0; w < nparts; ++w) {
        part_off[(long long) b * kBlkParts + w].x = d;
        d += parts[(long long) b * kBlkParts + w].x;
    }
    const int dp = ((d + (1 << ge) - 1) >> ge) << ge;
    int s = dp;
    for (int w = 0; w < nparts; ++w) {
        part_off[(long long) b * kBlkParts + w].y = s;
        s += parts[(long long) b * kBlkParts + w].y;
    }
    groups[b] = s >> ge;
    dense_groups[b] = dp >> ge;
}

// Inspector pass 3, the stable fill: ONE wavefront per (row block, column range).  It turns the range's cell counts into
// cursors (LDS) and group headers, then scans the block's entries in CSR order, 8 batches of 64 per round, appends the
// entries of its range to a small LDS queue (ballot + popcount: order kept) and, whenever 64 are queued, hands each the next
// position of its cell (LDS atomic with return: one wave, one instruction at a time, so positions follow the queue order
// batch after batch; inside a batch the LDS unit serialises the lanes that hit one cursor in a fixed order) and stores
// value and (column field | row << 16) there.  VALUES_ONLY: re-permute new values into the same positions
// (spmv_hip_update_values).  Dynamic LDS: 2 x (most cells of a range) + 128 ints.
template <typename T, bool VALUES_ONLY>
__global__ __launch_bounds__(kWave) void blk_fill_kernel(const int *__restrict__ row0, int K, int wshift, int dense_min, int ge, int nparts, int S, int range_cells,
                                                         const int *__restrict__ rowptr, const int *__restrict__ colidx, const T *__restrict__ val,
                                                         const int *__restrict__ cnt, const i32x2 *__restrict__ part_off, const long long *__restrict__ gstart,
                                                         const int *__restrict__ dense_groups, T *__restrict__ bval, unsigned *__restrict__ bmeta,
                                                         int *__restrict__ hdr_a, int *__restrict__ hdr_b, BlkDir *__restrict__ dir)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_fill_lds[];
    unsigned *cur = reinterpret_cast<unsigned *>(blk_fill_lds); // next position of every cell of the range (entries from the block region's start)
    unsigned *cst = cur + range_cells;                          // the cell's first position | dense << 31
    int *queue = reinterpret_cast<int *>(cst + range_cells);    // ring of 128 CSR positions
    const int lane = threadIdx.x;
    const int b = blockIdx.x / nparts, part = blockIdx.x % nparts;
    const int cs = kBlkSuperShift - wshift, gsize = 1 << ge;
    int s0, s1;
    blk_part_range(S, nparts, part, s0, s1);
    const int k0 = s0 << cs, k1 = min(s1 << cs, K);
    const long long g0 = gstart[b];
    if (!VALUES_ONLY && part == 0 && lane == 0) dir[b] = BlkDir{g0, dense_groups[b], (int) (gstart[b + 1] - g0) - dense_groups[b]};
    const int *mycnt = cnt + (long long) b * K;
    { // cursors and headers of this range: super-slab after super-slab, every lane a contiguous run of the super-slab's cells
        const i32x2 off = part_off[(long long) b * kBlkParts + part];
        int doff = off.x, soff = off.y;
        for (int s = s0; s < s1; ++s) {
            const int ka = s << cs, kb = min((s + 1) << cs, K);
            const int per = (kb - ka + kWave - 1) / kWave;
            const int a = min(ka + lane * per, kb), e = min(a + per, kb);
            int d = 0, sp = 0;
            for (int k = a; k < e; ++k) {
                const int c = mycnt[k];
                if (c >= dense_min) d += c; else sp += c;
            }
            int dtot, stot;
            int dpos = doff + wave_excl_scan(d, lane, &dtot), spos = soff + wave_excl_scan(sp, lane, &stot);
            for (int k = a; k < e; ++k) {
                const int c = mycnt[k];
                if (c >= dense_min) {
                    cur[k - k0] = (unsigned) dpos;
                    cst[k - k0] = (unsigned) dpos | 0x80000000u;
                    if (!VALUES_ONLY) { // this cell is slab A of every group that STARTS inside it, slab B of the group it starts inside of
                        for (int g = (dpos + gsize - 1) >> ge; g <= (dpos + c - 1) >> ge; ++g) hdr_a[g0 + g] = k;
                        if (dpos & (gsize - 1)) hdr_b[g0 + (dpos >> ge)] = k;
                    }
                    dpos += c;
                } else {
                    cur[k - k0] = (unsigned) spos;
                    cst[k - k0] = (unsigned) spos;
                    spos += c;
                }
            }
            doff += dtot;
            const int send = ((soff + stot + gsize - 1) >> ge) << ge; // the super-slab's run, in whole groups
            if (!VALUES_ONLY)
                for (int g = (soff >> ge) + lane; g < (send >> ge); g += kWave) hdr_a[g0 + g] = s << kBlkSuperShift; // first column of the super-slab
            soff = send;
        }
    }
    __syncthreads();
    const int r0 = row0[b], nr = row0[b + 1] - r0;
    const int p0 = rowptr[r0], p1 = rowptr[r0 + nr];
    const long long e0 = g0 << ge; // first entry of the block's region
    int qh = 0, qn = 0;            // queue head, entries queued (wave-uniform)
    auto drain = [&](int count) {  // the first `count` (<= 64) queued entries, in queue order
        const bool on = lane < count;
        const int p = on ? queue[(qh + lane) & 127] : p0;
        const int c = colidx[p];
        const T v = val[p];
        if (on) {
            const int kk = (c >> wshift) - k0;
            const unsigned pos = atomicAdd(&cur[kk], 1u);
            // lane l of the executor loads 16 bytes = the entries l, 64 + l (fp32: .., 128 + l, 192 + l) of the group's sorted order, so that ONE gather
            // instruction covers 64 CONSECUTIVE entries of that order: the entries of a cell then sit in one instruction and their lanes merge per cache line
            const unsigned epl = 16u / (unsigned) sizeof(T), gi = pos & ((1u << ge) - 1u);
            const long long spos = e0 + (long long) (pos - gi) + (long long) ((gi & (kWave - 1)) * epl + (gi >> 6));
            bval[spos] = v;
            if constexpr (!VALUES_ONLY) {
                const unsigned st = cst[kk];
                int lo = 0, hi = nr; // row of CSR position p inside the block: rowptr[r0 + lo] <= p < rowptr[r0 + hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rowptr[r0 + mid] <= p) lo = mid; else hi = mid;
                }
                unsigned col;
                if (st >> 31) { // dense: slab A of my group if my cell started at or before the group's first entry
                    const unsigned gfirst = (pos >> ge) << ge;
                    col = ((st & 0x7fffffffu) > gfirst ? 1u << kBlkSlabShift : 0u) | ((unsigned) c & ((1u << kBlkSlabShift) - 1u));
                } else {
                    col = (unsigned) c & ((1u << kBlkSuperShift) - 1u);
                }
                bmeta[spos] = col | ((unsigned) lo << 16);
            }
        }
        qh = (qh + count) & 127;
        qn -= count;
    };
    constexpr int NB = 8; // batches of 64 entries loaded per round
    for (int q = p0; q < p1; q += NB * kWave) {
        int c[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) { // padded ColIdx: in bounds
            const int p = q + i * kWave + lane;
            c[i] = p < p1 ? ld_stream(colidx + p) : -1;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int k = c[i] >> wshift; // -1 for lanes past the block's end
            const bool mine = k >= k0 && k < k1;
            const unsigned long long mk = __ballot(mine);
            if (mine) queue[(qh + qn + __popcll(mk & ((1ull << lane) - 1ull))) & 127] = q + i * kWave + lane;
            qn += __popcll(mk);
            if (qn >= kWave) {
                __syncthreads();
                drain(kWave);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (qn > 0) drain(qn);
}

// The block's y is accumulated in DOUBLE for both value types: ds_add_f32 runs at 2.0e11 adds/s over the chip whatever the
// bank pattern, ds_add_f64 at 7-9e11/s (tools/gbench lds, profiles/r02_gbench.txt) -- with float accumulators every fp32
// matrix ran at the LDS-atomic rate (0.51 ms for 9.6e7 entries, local columns or not).  fp32 products are rounded to
// float first (as every other executor forms them) and summed in double, which is at least as accurate as a float sum.
__device__ __forceinline__ void lds_add(double *p, double v) { (void) unsafeAtomicAdd(p, v); }

template <typename T, int UN>
struct BlkStep { // the stream of one step: UN groups, per lane 16 bytes of values and their (column field | row << 16) words ...
    T v[UN][16 / sizeof(T)];
    unsigned w[UN][16 / sizeof(T)];
    int ha, hb;  // ... and, in lane u < UN, the header words of group u of some step (see the loops: which step)
};

// stream of the UN groups from group g on, header words of the groups from hg on (all relative to the block's region)
template <typename T, int UN, bool DENSE>
__device__ __forceinline__ void blk_load_step(int g, int hg, int lane, const T *__restrict__ bv, const unsigned *__restrict__ bm, const int *__restrict__ ha,
                                              const int *__restrict__ hb, BlkStep<T, UN> &s)
{
    constexpr int EPL = 16 / (int) sizeof(T);
    s.ha = ha[hg + lane]; // one coalesced load for all groups of a step (the arrays are padded by kBlkPadGroups); a scalar load per group would have
    if constexpr (DENSE) s.hb = hb[hg + lane]; // to be waited for with lgkmcnt(0), i.e. together with every LDS operation in flight
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int p = ((g + u) * kWave + lane) * EPL;
        if constexpr (EPL == 2) {
            const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(bv + p));
            const i32x2 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(bm + p));
            s.v[u][0] = q.x; s.v[u][1] = q.y;
            s.w[u][0] = (unsigned) cc.x; s.w[u][1] = (unsigned) cc.y;
        } else {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(bv + p));
            const i32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(bm + p));
            s.v[u][0] = q.x; s.v[u][1] = q.y; s.v[u][2] = q.z; s.v[u][3] = q.w;
            s.w[u][0] = (unsigned) cc.x; s.w[u][1] = (unsigned) cc.y; s.w[u][2] = (unsigned) cc.z; s.w[u][3] = (unsigned) cc.w;
        }
    }
}

// Executor.  Dynamic LDS: (R + 2) doubles (the block's y; slot R takes the products of padding entries and of groups past a
// part's end), then -- only when some block has dense groups -- kBlkDenseUn x 256 values of x (two slabs per group of a step).
// UN = groups per step of the sparse loop (8 or 12 loads of values in flight per lane and step).
//
// Both loops run three steps deep: in step t the wave ISSUES the stream loads of step t + 2, brings in the x of step t + 1
// (dense: writes the slabs loaded during step t - 1 to LDS and issues the slab loads of step t + 2; sparse: issues the gathers,
// whose columns arrived during step t - 1 .. t) and ADDS step t.  Nothing is waited for in the step that issued it.  Three
// stream register sets (and two x sets in the sparse loop) are used in rotation -- the loop bodies are written out for
// three / six consecutive steps -- so no loaded register is ever copied.  Group headers travel with the stream sets as ONE
// vector load per step (lane u = group u) and are broadcast with v_readlane when needed, a step after they were loaded.
template <typename T, int UN, int DBG = 0> // DBG (tools only, wrong results): 1 = coalesced x reads instead of gathers, 2 = no LDS adds, 3 = both
__global__ __launch_bounds__(kWave) void blk_kernel(const int *__restrict__ row0, int R, const BlkDir *__restrict__ dir, const T *__restrict__ bval,
                                                    const unsigned *__restrict__ bmeta, const int *__restrict__ hdr_a, const int *__restrict__ hdr_b,
                                                    const T *__restrict__ x, int n, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    double *ys = reinterpret_cast<double *>(blk_y_lds);
    T *xb = reinterpret_cast<T *>(ys + ((R + 2) & ~1)); // slab buffers of the dense loop
    constexpr int EPL = 16 / (int) sizeof(T);
    constexpr int DU = kBlkDenseUn;
    constexpr int SLAB = 1 << kBlkSlabShift;
    const int lane = threadIdx.x;
    for (int i = lane; i <= R; i += kWave) ys[i] = 0.0;
    __syncthreads();
    const BlkDir d = dir[blockIdx.x];
    const unsigned junk = (unsigned) R;
    const long long e0 = d.g0 * (long long) (kWave * EPL);
    const T *__restrict__ bv = bval + e0;           // the block's region: groups [0, nd) dense, [nd, nd + ns) sparse
    const unsigned *__restrict__ bm = bmeta + e0;
    const int *__restrict__ ha = hdr_a + d.g0, *__restrict__ hb = hdr_b + d.g0;

    if (d.nd > 0) { // ------------------------------------------------------------------ dense groups: x through LDS
        const int nd = d.nd, nsteps = (nd + DU - 1) / DU;
        BlkStep<T, DU> g0, g1, g2; // the set of step s carries the headers of step s + 1
        T sl[DU][4];               // the two slabs of every group of a step: A[lane], A[64 + lane], B[lane], B[64 + lane]
        auto load_slabs = [&](int s, int hva, int hvb) { // slabs of step s, whose header words are in (hva, hvb)
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const bool ok = s * DU + u < nd; // wave-uniform
                int a = __builtin_amdgcn_readlane(hva, u), bb = __builtin_amdgcn_readlane(hvb, u);
                a = ok ? a : 0;
                bb = ok && bb >= 0 ? bb : a;     // no second cell in this group: stage slab A twice
                const int ca = a << kBlkSlabShift, cb = bb << kBlkSlabShift;
                sl[u][0] = x[min(ca + lane, n - 1)];
                sl[u][1] = x[min(ca + kWave + lane, n - 1)];
                sl[u][2] = x[min(cb + lane, n - 1)];
                sl[u][3] = x[min(cb + kWave + lane, n - 1)];
            }
        };
        auto write_slabs = [&]() {
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                xb[u * 2 * SLAB + lane] = sl[u][0];
                xb[u * 2 * SLAB + kWave + lane] = sl[u][1];
                xb[u * 2 * SLAB + SLAB + lane] = sl[u][2];
                xb[u * 2 * SLAB + SLAB + kWave + lane] = sl[u][3];
            }
        };
        load_slabs(0, ha[lane], hb[lane]);
        blk_load_step<T, DU, true>(0, DU, lane, bv, bm, ha, hb, g0);
        blk_load_step<T, DU, true>(DU, 2 * DU, lane, bv, bm, ha, hb, g1);
        write_slabs();  // slabs of step 0
        load_slabs(1, g0.ha, g0.hb);
#define SPMV_BLK_DENSE_PHASE(ga, gb, gc)                                                                                   \
        {                                                                                                                   \
            blk_load_step<T, DU, true>((t + 2) * DU, (t + 3) * DU, lane, bv, bm, ha, hb, gc);                               \
            T xv[DU][EPL];                                                                                                  \
            _Pragma("unroll") for (int u = 0; u < DU; ++u)                                                                  \
                _Pragma("unroll") for (int j = 0; j < EPL; ++j) xv[u][j] = xb[u * 2 * SLAB + (ga.w[u][j] & (2 * SLAB - 1))]; \
            write_slabs();     /* slabs of step t + 1, loaded during step t - 1 .. t; DS operations of a wave execute in order: the reads above come first */ \
            load_slabs(t + 2, gb.ha, gb.hb);                                                                                \
            _Pragma("unroll") for (int u = 0; u < DU; ++u) {                                                                \
                const bool ok = t * DU + u < nd;                                                                            \
                _Pragma("unroll") for (int j = 0; j < EPL; ++j)                                                             \
                    lds_add(&ys[ok ? ga.w[u][j] >> 16 : junk], (double) (ga.v[u][j] * xv[u][j]));                           \
            }                                                                                                               \
            if (++t >= nsteps) break;                                                                                       \
        }
        for (int t = 0;;) {
            SPMV_BLK_DENSE_PHASE(g0, g1, g2)
            SPMV_BLK_DENSE_PHASE(g1, g2, g0)
            SPMV_BLK_DENSE_PHASE(g2, g0, g1)
        }
#undef SPMV_BLK_DENSE_PHASE
    }

    if (d.ns > 0) { // ------------------------------------------------------------------ sparse groups: x gathered through L2
        const int gs = d.nd, ns = d.ns, nsteps = (ns + UN - 1) / UN;
        BlkStep<T, UN> g0, g1, g2; // the set of step s carries the headers of step s
        T x0[UN][EPL], x1[UN][EPL];
        auto gather = [&](int s, const BlkStep<T, UN> &g, T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                int base = __builtin_amdgcn_readlane(g.ha, u);
                base = s * UN + u < ns ? base : 0; // wave-uniform; a group past the part's end gathers x[16-bit offset]: in bounds (offset < min(n, 65536))
                const T *__restrict__ xs = x + base;
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[u][j] = (DBG & 1) ? x[lane * EPL + j + u * kWave * EPL] : xs[g.w[u][j] & 0xffffu];
            }
        };
        double dbg_acc = 0.0;
        auto add = [&](int s, const BlkStep<T, UN> &g, const T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool ok = s * UN + u < ns;
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    if constexpr (DBG & 2) dbg_acc += (double) (g.v[u][j] * xv[u][j]) * (double) (ok ? g.w[u][j] >> 16 : junk);
                    else lds_add(&ys[ok ? g.w[u][j] >> 16 : junk], (double) (g.v[u][j] * xv[u][j]));
                }
            }
        };
        blk_load_step<T, UN, false>(gs, gs, lane, bv, bm, ha, hb, g0);
        blk_load_step<T, UN, false>(gs + UN, gs + UN, lane, bv, bm, ha, hb, g1);
        gather(0, g0, x0);
#define SPMV_BLK_PHASE(ga, gb, gc, xa, xbb)                                                                                \
        blk_load_step<T, UN, false>(gs + (t + 2) * UN, gs + (t + 2) * UN, lane, bv, bm, ha, hb, gc);                       \
        gather(t + 1, gb, xbb);                                                                                            \
        add(t, ga, xa);                                                                                                    \
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
        if constexpr (DBG & 2) ys[lane] = dbg_acc;
    }
    __syncthreads();
    const long long r0 = row0[blockIdx.x];
    const int nr = row0[blockIdx.x + 1] - (int) r0;
    for (int i = lane; i < nr; i += kWave) y[r0 + i] = (T) ys[i];
}

} // namespace spmv
