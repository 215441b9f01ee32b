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
//               2^wshift columns (128 unless that would make more than 12288 slabs).  The entries of a row block are
//               stored sorted by slab -- the sweep over the slabs, in step over the blocks of an XCD, is what keeps x in L2 --
//               in GROUPS of 64 lanes x 16 bytes of values (128 fp64 / 256 fp32 entries), as two streams: the value and a
//               32-bit word (16-bit column offset | 16-bit row inside the block) -- 12 bytes per fp64 entry, the bytes of
//               plain CSR.  16-bit column offsets count from the first column of a SUPER-SLAB of 65536 columns; every
//               super-slab's run of a block is padded to whole groups, so a group lies in ONE super-slab and its header is
//               that super-slab's first column (wave-uniform; the gather is base + offset).  Inside a group the sorted order
//               is dealt to the lanes so that ONE gather instruction covers 64 CONSECUTIVE entries of it (lane l holds
//               entries l and 64 + l): the entries of a (block, slab) cell then sit in one instruction and the lanes that
//               hit the same cache line merge into one L2 request (with lane l holding 2l and 2l + 1 a cell was split over
//               two instructions: 0.63 instead of ~0.45 L2 requests per entry on uniformly random columns).
//   inspector   (device, three kernels over the resident CSR) blk_rows_kernel: the row-in-block of every entry;
//               blk_count_kernel: one 1024-thread workgroup per block counts the block's cells in LDS and reports the
//               block's group count; blk_fill_kernel: the same workgroup counts again, one wave turns the counts into cell
//               cursors (LDS) and writes the group headers, then the 16 waves walk the block's entries in CSR order, 4096
//               per round, wave w owning entries [256 w, 256 w + 256) of the round, and take TURNS (a barrier apiece) at
//               advancing the cursors with LDS atomics: cursor updates therefore happen in CSR order, wave after wave,
//               instruction after instruction (inside one instruction the LDS unit serialises the lanes that hit one
//               cursor in a fixed order), so an entry's rank inside its cell is its CSR rank and the stored order -- (slab, then
//               blk_spread(CSR rank): the entries of a large cell dealt out so that one row's neighbours do not share an
//               instruction) -- is a function of the matrix alone.
//   executor    ONE WAVEFRONT per row block (a 64-thread workgroup, two blocks per CU).  y of the block lives in LDS
//               as doubles (for fp32 values too -- see lds_add); the wave walks the block's groups three steps deep
//               (stream loads of step t + 2, gathers of step t + 1, additions of step t) and adds every product into y's
//               LDS copy (ds_add_f64).  At the end the block's y is written once, coalesced: no partial sums, no
//               carries, no read-modify-write of y in HBM.  No load sits behind a branch: groups past the block's end are
//               read (the streams are padded) and their products go to a junk accumulator.
//   determinism every row is touched by exactly one wavefront, whose additions happen in program order over a stream
//               whose order is fixed by the inspector: the result is reproducible bit for bit, run to run and handle
//               to handle.
//   not here    DENSE (block, slab) cells read through LDS (two coalesced slabs of x per group, the column field naming
//               slab and offset): built and measured in round 3 (commit "dense cells through LDS as an experiment") --
//               slower on every shape (Orkut-style R-MAT 0.65 vs 0.57 ms, web-like 4e6 x 24 0.31 vs 0.24): the cells that
//               qualify are exactly those whose gathers already merge into a handful of L2 requests or hit L1, and the
//               slab traffic (2 KiB per 128 entries through the vector-memory path and LDS) costs more than it saves.
#pragma once
#include <climits>
#include "common.hpp"

namespace spmv {

constexpr int kBlkSlabShift = 7;     // slabs of 128 columns unless that makes too many cells
constexpr int kBlkSuperShift = 16;   // 16-bit column offsets inside super-slabs of 65536 columns
constexpr int kBlkMaxCells = 12288;  // cells of one row block at most (the inspector keeps count, start and cursor of every cell in LDS: 144 KiB)
constexpr int kBlkThreads = 1024;    // inspector workgroups: 16 waves per row block
constexpr int kBlkTurn = 4;          // batches of 64 entries a wave places per turn
constexpr int kBlkStepGroups = 24;   // a block's region is padded to a multiple of this: whole executor steps of either form (8 or 12 groups), nothing to mask
constexpr int kBlkPadGroups = 384;   // zero groups behind the last block: the look-ahead of the widest form (wide: 3 x 8 waves x 12 groups), and header loads reach 64 groups ahead

struct BlkDir {      // one row block's region: groups [g0, g0 + ns)
    long long g0;
    int ns, pad;
};

// Row blocks of EQUAL WORK: block b holds rows [row0[b], row0[b + 1]).  All blocks of a round are resident together and the
// round lasts as long as its heaviest block (Orkut-style stand-in, fixed 5997-row blocks: heaviest / mean = 1.18; 4e6 rows of
// 2.6 entries: 1.40), so the cut points follow the entries, not the rows: work(r) = RowPtr[r] + c r (c entries of fixed cost per
// row, so that stretches of empty rows still end a block), each block takes 1 / (blocks left) of the work left, never more than
// rcap rows (its accumulators live in LDS).  One workgroup: B binary searches over RowPtr.  out[0] = blocks made (>= btarget when
// the row cap cut some short), out[1] = most rows in a block.
static __global__ __launch_bounds__(kBlock) void blk_partition_kernel(int m, const int *__restrict__ rowptr, int btarget, int rcap, long long c,
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

// rowin[p] = row of entry p inside its row block, for every entry of block blockIdx.x: the block's row pointers in LDS
// ((R + 1) ints), one binary search per entry.
static __global__ __launch_bounds__(kBlkThreads) void blk_rows_kernel(const int *__restrict__ row0, const int *__restrict__ rowptr, unsigned short *__restrict__ rowin)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_rows_lds[];
    int *rp = reinterpret_cast<int *>(blk_rows_lds);
    const int r0 = row0[blockIdx.x], nr = row0[blockIdx.x + 1] - r0;
    for (int i = threadIdx.x; i <= nr; i += kBlkThreads) rp[i] = rowptr[r0 + i];
    __syncthreads();
    const int p0 = rp[0], p1 = rp[nr];
    for (int p = p0 + (int) threadIdx.x; p < p1; p += kBlkThreads) {
        int lo = 0, hi = nr; // rp[lo] <= p < rp[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rp[mid] <= p) lo = mid; else hi = mid;
        }
        rowin[p] = (unsigned short) lo;
    }
}

// cells[k] = entries of row block b in slab k (LDS histogram, all threads of the workgroup)
__device__ __forceinline__ void blk_count_cells(unsigned *cells, int K, int wshift, int p0, int p1, const int *__restrict__ colidx)
{
    for (int i = threadIdx.x; i < K; i += kBlkThreads) cells[i] = 0u;
    __syncthreads();
    for (int p = p0 + (int) threadIdx.x; p < p1; p += kBlkThreads) atomicAdd(&cells[ld_stream(colidx + p) >> wshift], 1u);
    __syncthreads();
}

// Inspector pass 1: groups[b] = groups of 2^ge entries row block b needs -- its cells counted in LDS (K counters), every
// super-slab's run rounded up to whole groups.
static __global__ __launch_bounds__(kBlkThreads) void blk_count_kernel(const int *__restrict__ row0, int K, int wshift, int ge, const int *__restrict__ rowptr,
                                                                const int *__restrict__ colidx, int *__restrict__ groups, int *__restrict__ occupied)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_count_lds[];
    unsigned *cells = reinterpret_cast<unsigned *>(blk_count_lds);
    __shared__ int total, occ;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) total = occ = 0;
    blk_count_cells(cells, K, wshift, rowptr[row0[b]], rowptr[row0[b + 1]], colidx);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int cs = kBlkSuperShift - wshift, S = (K + (1 << cs) - 1) >> cs; // cells per super-slab = 2^cs
    int mine = 0, nz = 0;
    for (int s = wave; s < S; s += kBlkThreads / kWave) {
        const int ka = s << cs, kb = min((s + 1) << cs, K);
        int c = 0;
        for (int k = ka + lane; k < kb; k += kWave) {
            c += (int) cells[k];
            nz += cells[k] != 0u;
        }
        mine += (wave_sum(c) + (1 << ge) - 1) >> ge;
    }
    nz = wave_sum(nz);
    if (lane == 0 && mine) atomicAdd(&total, mine);
    if (lane == 0 && nz) atomicAdd(&occ, nz);
    __syncthreads();
    if (threadIdx.x == 0) {
        groups[b] = (total + kBlkStepGroups - 1) / kBlkStepGroups * kBlkStepGroups; // whole steps (padding groups: junk row, value 0)
        occupied[b] = occ;                                                          // cells with entries: how scattered the block's gathers are
    }
}

// position of entry i of a group's sorted order inside the group's 2^ge stored entries: lane (i mod 64) holds it as its
// (i / 64)-th entry, and a lane's EPL entries are consecutive in memory (one 16-byte load)
__device__ __forceinline__ unsigned blk_stored_pos(unsigned pos, int ge, unsigned epl)
{
    const unsigned gi = pos & ((1u << ge) - 1u);
    return (pos - gi) + (gi & (kWave - 1)) * epl + (gi >> 6);
}

// Where the q-th entry (CSR order) of a cell of c entries goes inside the cell.  Entries of ONE ROW that fall into the same cell
// are neighbours in CSR order; left side by side they would sit in neighbouring lanes of one executor instruction and their LDS
// adds would hit one accumulator -- the LDS unit serialises same-address atomics (a block of banded rows ran 2.5 x as long as a block
// of random ones, and with two blocks per slot ten such blocks cost the whole launch a third round: 1.71 vs 1.39 ms).  So a cell is
// dealt out in chunks of 64 s entries (s = min(32, c / 64)) transposed s x 64: positions 64 a .. 64 a + 63 of a chunk hold its
// entries a, a + s, a + 2 s, ...: a run of up to s entries of one row meets one lane per instruction, and for banded rows the 64
// lanes of an instruction hold the same entry of 64 consecutive rows -- consecutive columns, one coalesced gather.  Cells under
// 128 entries (the sparse case) keep CSR order.  A pure function of (q, c): the stored order stays a function of the matrix alone.
__device__ __forceinline__ unsigned blk_spread(unsigned q, unsigned c)
{
    unsigned s = c >> 6;
    if (s < 2u) return q;
    if (s > 32u) s = 32u;
    const unsigned chunk = s << 6, base = q - q % chunk, rem = c - base; // entries from this chunk's start to the cell's end
    if (rem < chunk) {        // the cell's last, partial chunk: a narrower transpose, its own last < 64 entries in place
        s = rem >> 6;
        if (s < 2u || q - base >= (s << 6)) return q;
    }
    const unsigned i = q - base;
    return base + (i % s) * 64u + i / s;
}

// Inspector pass 2, the stable fill (see the header).  Dynamic LDS: 3 K words (count, start and cursor of every cell).  VALUES_ONLY:
// re-permute new values into the same positions (spmv_hip_update_values).
template <typename T, bool VALUES_ONLY>
__global__ __launch_bounds__(kBlkThreads) void blk_fill_kernel(const int *__restrict__ row0, int K, int wshift, int ge, const int *__restrict__ rowptr,
                                                               const int *__restrict__ colidx, const T *__restrict__ val, const unsigned short *__restrict__ rowin,
                                                               const long long *__restrict__ gstart, T *__restrict__ bval, unsigned *__restrict__ bmeta,
                                                               int *__restrict__ hdr, BlkDir *__restrict__ dir, int subsort, unsigned short *__restrict__ kscr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_fill_lds[];
    unsigned *cnt = reinterpret_cast<unsigned *>(blk_fill_lds); // entries of every cell
    unsigned *cst = cnt + K;                                     // first position of the cell (entries from the block region's start)
    unsigned *cur = cst + K;                                     // entries of the cell placed so far
    const int b = blockIdx.x;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int p0 = rowptr[row0[b]], p1 = rowptr[row0[b + 1]];
    const long long g0 = gstart[b];
    blk_count_cells(cnt, K, wshift, p0, p1, colidx);
    for (int i = threadIdx.x; i < K; i += kBlkThreads) cur[i] = 0u;
    if (wave == 0) { // counts -> cell starts, super-slab after super-slab (every lane a contiguous run of the super-slab's cells); group headers
        const int cs = kBlkSuperShift - wshift, S = (K + (1 << cs) - 1) >> cs, gsize = 1 << ge;
        int off = 0;
        for (int s = 0; s < S; ++s) {
            const int ka = s << cs, kb = min((s + 1) << cs, K);
            const int per = (kb - ka + kWave - 1) / kWave;
            const int a = min(ka + lane * per, kb), e = min(a + per, kb);
            int c = 0;
            for (int k = a; k < e; ++k) c += (int) cnt[k];
            int tot;
            int pos = off + wave_excl_scan(c, lane, &tot);
            for (int k = a; k < e; ++k) {
                cst[k] = (unsigned) pos;
                pos += (int) cnt[k];
            }
            const int end = ((off + tot + gsize - 1) >> ge) << ge; // the super-slab's run, in whole groups
            if (!VALUES_ONLY)
                for (int g = (off >> ge) + lane; g < (end >> ge); g += kWave) hdr[g0 + g] = s << kBlkSuperShift; // first column of the super-slab
            off = end;
        }
        if (!VALUES_ONLY && lane == 0) dir[b] = BlkDir{g0, ((off >> ge) + kBlkStepGroups - 1) / kBlkStepGroups * kBlkStepGroups, 0};
    }
    __syncthreads();
    const long long e0 = g0 << ge; // first entry of the block's region
    constexpr unsigned EPL = 16u / (unsigned) sizeof(T);
    constexpr int ROUND = kBlkThreads * kBlkTurn; // entries per round; wave w owns [w * 64 * kBlkTurn, ...) of it: turns in wave order = CSR order
    for (int q = p0; q < p1; q += ROUND) {
        int c[kBlkTurn], r[kBlkTurn];
        T v[kBlkTurn];
        unsigned pos[kBlkTurn];
#pragma unroll
        for (int i = 0; i < kBlkTurn; ++i) {
            const int p = q + (wave * kBlkTurn + i) * kWave + lane;
            const bool on = p < p1;
            c[i] = on ? ld_stream(colidx + p) : -1;
            v[i] = on ? ld_stream(val + p) : T(0);
            if constexpr (!VALUES_ONLY) r[i] = on ? (int) ld_stream(rowin + p) : 0;
        }
        for (int turn = 0; turn < kBlkThreads / kWave; ++turn) {
            if (wave == turn) {
#pragma unroll
                for (int i = 0; i < kBlkTurn; ++i)
                    if (c[i] >= 0) pos[i] = atomicAdd(&cur[c[i] >> wshift], 1u); // rank of the entry inside its cell, CSR order
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < kBlkTurn; ++i)
            if (c[i] >= 0) {
                const int k = c[i] >> wshift;
                const long long sp = e0 + blk_stored_pos(cst[k] + blk_spread(pos[i], cnt[k]), ge, EPL);
                bval[sp] = v[i];
                if constexpr (!VALUES_ONLY) bmeta[sp] = ((unsigned) c[i] & ((1u << kBlkSuperShift) - 1u)) | ((unsigned) r[i] << 16);
                else if (subsort) kscr[sp] = (unsigned short) ((unsigned) c[i] & ((1u << kBlkSuperShift) - 1u));
            }
    }
    if (!subsort) return;
    // Sparse cells -- fewer than 128 entries, the ones blk_spread leaves in CSR order -- are sorted by COLUMN (ties: CSR order).  One
    // executor gather instruction covers 64 consecutive entries of the stored order; with a cell in CSR order an instruction that ends
    // inside the cell takes a RANDOM subset of its entries, spread over all of the slab's cache lines, and the neighbour instruction
    // fetches the same lines again.  Sorted, an instruction's entries cover a contiguous column range and everything the block has in
    // those lines: (1 - exp(-l)) / l lines per entry for l entries per line and block instead of the average over the cut position
    // (20 k-row blocks, 32 random columns per row of 1e7: 0.63 instead of 0.74).  A wave per cell; rank = entries with a smaller
    // (column, CSR rank) key, counted with v_readlane over the cell's <= 128 keys; every load of the cell precedes its first store.
    __syncthreads(); // the block's entries are in place (written by all waves)
    for (int k = wave; k < K; k += kBlkThreads / kWave) {
        const unsigned cc = (unsigned) __builtin_amdgcn_readfirstlane((int) cnt[k]); // wave-uniform: k is
        if (cc < 2u || cc >= 128u) continue;
        const unsigned base = (unsigned) __builtin_amdgcn_readfirstlane((int) cst[k]);
        unsigned key[2], mw[2] = {0u, 0u};
        long long sp[2];
        T vv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned i = (unsigned) lane + 64u * j;
            const bool on = i < cc;
            sp[j] = e0 + blk_stored_pos(base + (on ? i : 0u), ge, EPL);
            vv[j] = bval[sp[j]];
            if constexpr (VALUES_ONLY) key[j] = kscr[sp[j]];
            else { mw[j] = bmeta[sp[j]]; key[j] = mw[j] & 0xffffu; }
            key[j] = on ? (key[j] << 7) | i : 0xffffffffu;
        }
        unsigned rank[2] = {0u, 0u};
        const int c0 = cc < 64u ? (int) cc : 64, c1 = (int) cc - 64;
        for (int q = 0; q < c0; ++q) {
            const unsigned kq = (unsigned) __builtin_amdgcn_readlane((int) key[0], q);
            rank[0] += kq < key[0];
            rank[1] += kq < key[1];
        }
        for (int q = 0; q < c1; ++q) {
            const unsigned kq = (unsigned) __builtin_amdgcn_readlane((int) key[1], q);
            rank[0] += kq < key[0];
            rank[1] += kq < key[1];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every load of the cell before its first store
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if ((unsigned) lane + 64u * j < cc) {
                const long long np = e0 + blk_stored_pos(base + rank[j], ge, EPL);
                bval[np] = vv[j];
                if constexpr (!VALUES_ONLY) bmeta[np] = mw[j];
            }
    }
}

// The block's y is accumulated in DOUBLE for both value types: ds_add_f32 runs at 2.0e11 adds/s over the chip whatever the
// bank pattern, ds_add_f64 at 7-9e11/s (tools/gbench lds, profiles/r02_gbench.txt) -- with float accumulators every fp32
// matrix ran at the LDS-atomic rate (0.51 ms for 9.6e7 entries, local columns or not).  fp32 products are rounded to
// float first (as every other executor forms them) and summed in double, which is at least as accurate as a float sum.
__device__ __forceinline__ void lds_add(double *p, double v) { (void) unsafeAtomicAdd(p, v); }

template <typename T, int UN>
struct BlkStep { // the stream of one step: UN groups, per lane 16 bytes of values and their (column offset | row << 16) words ...
    T v[UN][16 / sizeof(T)];
    unsigned w[UN][16 / sizeof(T)];
    int h;       // ... and, in lane u < UN, the header (first column of the super-slab) of group u of the step
};

// stream and headers of the UN groups from group g on (relative to the block's region).  Addresses are a wave-uniform base (scalar
// registers: the step's first byte of each stream) + the lane's constant byte offset + an immediate per group: no vector arithmetic per
// load (indexing the arrays with a 32-bit element index cost three VALU instructions per load, a quarter of the loop's vector work).
template <typename T, int UN>
__device__ __forceinline__ void blk_load_step(int g, int lane, const T *__restrict__ bv, const unsigned *__restrict__ bm, const int *__restrict__ hd, BlkStep<T, UN> &s)
{
    constexpr int EPL = 16 / (int) sizeof(T);
    const char *pv = reinterpret_cast<const char *>(bv) + (size_t) g * (kWave * 16);               // 1 KiB of values per group
    const char *pm = reinterpret_cast<const char *>(bm) + (size_t) g * (kWave * EPL * 4);          // 512 B (fp64) / 1 KiB (fp32) of words per group
    const unsigned lv = (unsigned) lane * 16u, lm = (unsigned) lane * (EPL * 4u);
    s.h = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(hd) + (size_t) g * 4 + (unsigned) lane * 4u);
    // ^ one coalesced load for all groups of a step (the array is padded by kBlkPadGroups); a scalar load per group would have to be
    //   waited for with lgkmcnt(0), i.e. together with every LDS operation in flight
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        if constexpr (EPL == 2) {
            const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(pv + lv + u * (kWave * 16)));
            const i32x2 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(pm + lm + u * (kWave * 8)));
            s.v[u][0] = q.x; s.v[u][1] = q.y;
            s.w[u][0] = (unsigned) cc.x; s.w[u][1] = (unsigned) cc.y;
        } else {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pv + lv + u * (kWave * 16)));
            const i32x4 cc = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(pm + lm + u * (kWave * 16)));
            s.v[u][0] = q.x; s.v[u][1] = q.y; s.v[u][2] = q.z; s.v[u][3] = q.w;
            s.w[u][0] = (unsigned) cc.x; s.w[u][1] = (unsigned) cc.y; s.w[u][2] = (unsigned) cc.z; s.w[u][3] = (unsigned) cc.w;
        }
    }
}

// Executor.  Dynamic LDS: (R + 2) doubles (the block's y; slot R takes the products of padding entries and of groups past the
// block's end).  UN = groups per step (8 / 12 loads of values in flight per lane and step).
//
// Three steps deep: in step t the wave ISSUES the stream loads of step t + 2, ISSUES the gathers of step t + 1 (whose columns
// arrived during step t - 1 .. t) and ADDS step t (whose x values were gathered during step t - 1).  Nothing is waited for in
// the step that issued it.  Three stream register sets and two x sets are used in rotation (the loop body is written out for
// six consecutive steps), so no loaded register is ever copied.  Group headers travel with the stream sets as ONE vector load
// per step (lane u = group u) and are broadcast with v_readlane when the gathers are issued, a step after they were loaded.
#ifdef SPMV_BLK_DEBUG_FORMS
__device__ unsigned long long blk_dbg_times[4 * 8192]; // per workgroup: start, end (s_memrealtime, 100 MHz), XCC id, block (tools: SPMV_BLK_DEBUG_FORMS builds only)
#endif

template <typename T, int UN, int DBG = 0> // DBG (tools only, wrong results): 1 = coalesced x reads instead of gathers, 2 = no LDS adds, 3 = both
__global__ __launch_bounds__(kWave) void blk_kernel(const int *__restrict__ row0, int R, const BlkDir *__restrict__ dir, const T *__restrict__ bval,
                                                    const unsigned *__restrict__ bmeta, const int *__restrict__ hdr, const int *__restrict__ order,
                                                    const T *__restrict__ x, T *__restrict__ y, int accumulate /* y += (the far half of a split matrix, shim/split.hpp) instead of y = */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    double *ys = reinterpret_cast<double *>(blk_y_lds);
    constexpr int EPL = 16 / (int) sizeof(T);
    const int lane = threadIdx.x;
    // Launch order (blocked_fill): the blocks whose entries are scattered over the most cells first, blocks of few cells (local rows: quick) behind
    // them, empty ones last; row order inside a class.  The blocks resident on an XCD sweep the column slabs in step -- that is what keeps the
    // slabs of x they gather from in L2 -- and a block that ends early hands its slot to a successor that then runs out of step with everyone
    // else for the rest of the launch: 1 % of quick leading blocks cost 25 % (1e7 x 32 random columns behind a 1 % banded prefix: 1.71 vs 1.39 ms).
    const int blk = order[blockIdx.x];
#ifdef SPMV_BLK_DEBUG_FORMS
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const BlkDir d = dir[blk];
    if (accumulate && d.ns == 0) return; // y += 0: nothing to do for a block without entries (the far half of a split matrix has many)
    for (int i = lane; i <= R; i += kWave) ys[i] = 0.0;
    __syncthreads();
    const long long e0 = d.g0 * (long long) (kWave * EPL);
    const T *__restrict__ bv = bval + e0;           // the block's region: groups [0, ns)
    const unsigned *__restrict__ bm = bmeta + e0;
    const int *__restrict__ hd = hdr + d.g0;
    static_assert(kBlkStepGroups % UN == 0, "a block's region is a whole number of steps");
    const int ns = d.ns, nsteps = ns / UN; // whole steps: the region is padded (junk-row entries) to a multiple of kBlkStepGroups groups
    if (ns > 0) {
        BlkStep<T, UN> g0, g1, g2;
        T x0[UN][EPL], x1[UN][EPL];
        // Steps past the block's end are LOADED (streams and headers of the next block, or the padding behind the last one) and GATHERED
        // (a consistent header / offset pair: a valid column) by the pipeline's look-ahead, never added.
        auto gather = [&](const BlkStep<T, UN> &g, T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const T *__restrict__ xs = x + __builtin_amdgcn_readlane(g.h, u);
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[u][j] = (DBG & 1) ? x[lane * EPL + j + u * kWave * EPL] : xs[g.w[u][j] & 0xffffu];
            }
        };
        double dbg_acc = 0.0;
        auto add = [&](const BlkStep<T, UN> &g, const T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    if constexpr (DBG & 2) dbg_acc += (double) (g.v[u][j] * xv[u][j]) * (double) (g.w[u][j] >> 16);
                    else lds_add(&ys[g.w[u][j] >> 16], (double) (g.v[u][j] * xv[u][j]));
                }
        };
        blk_load_step<T, UN>(0, lane, bv, bm, hd, g0);
        blk_load_step<T, UN>(UN, lane, bv, bm, hd, g1);
        gather(g0, x0);
#define SPMV_BLK_PHASE(ga, gb, gc, xa, xbb)                                                                                \
        blk_load_step<T, UN>((t + 2) * UN, lane, bv, bm, hd, gc);                                                          \
        gather(gb, xbb);                                                                                                   \
        add(ga, xa);                                                                                                       \
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
    const long long r0 = row0[blk];
    const int nr = row0[blk + 1] - (int) r0;
    if (accumulate) for (int i = lane; i < nr; i += kWave) y[r0 + i] += (T) ys[i];
    else for (int i = lane; i < nr; i += kWave) y[r0 + i] = (T) ys[i];
#ifdef SPMV_BLK_DEBUG_FORMS
    if (lane == 0 && blockIdx.x < 8192) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        blk_dbg_times[4 * blockIdx.x] = dbg_t0;
        blk_dbg_times[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        blk_dbg_times[4 * blockIdx.x + 2] = (xcc & 0xf) | ((unsigned long long) hw << 8);
        blk_dbg_times[4 * blockIdx.x + 3] = (unsigned long long) blk | ((unsigned long long) d.ns << 32);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The WIDE form (round 4): ONE row block per CU -- up to ~20 k rows, 159 KiB of double accumulators -- walked by W wavefronts of one
// workgroup that SHARE the accumulators.  Why: what bounds the executor on columns without locality is the number of L2 lines
// its gathers request (one 128-byte line per gathered double that does not merge with a neighbour in its
// instruction); the lanes of one gather instruction hold 64 consecutive entries of the block's column-sorted order, so the
// lines per entry are (1 - exp(-l)) / l with l = entries per line and block = rows x nnz-per-row x 16 / n -- 0.79 at 10 k rows
// (two blocks per CU), 0.63 at 20 k.  More rows per accumulator set is the only lever, and one wavefront cannot keep a CU's
// memory pipeline full alone (a block alone on its CU: 546 us against 690 us for two side by side, tools/blk_timeline.py).
//   work split   wave w takes the block's steps w, w + W, w + 2 W, ... (a step = UN groups): all waves sweep the column slabs
//                together, the XCD's blocks stay in step as before.
//   ORD = 0      every wave adds its products as they arrive (ds_add_f64 on shared accumulators): the order in which two
//                waves' additions reach one row is decided by the hardware -- results are correct to rounding but NOT
//                reproducible bit for bit (option "deterministic" = 0).
//   ORD = 1      the waves take TURNS at adding: in every phase (one step per wave) wave 0 adds, waits for its LDS
//                operations, barrier, wave 1 adds, ... -- additions reach every row in (phase, wave, instruction, lane)
//                order, a function of the stored stream alone: bit-reproducible like the one-wave form.  Loads and gathers
//                are not ordered (they stay in flight across the barriers: s_barrier without the vmcnt drain of
//                __syncthreads); the LDS unit executes one wave's adds at a time either way.
template <typename T, int UN, int W, bool ORD>
__global__ __launch_bounds__(kWave * W) void blk_wide_kernel(const int *__restrict__ row0, int R, const BlkDir *__restrict__ dir, const T *__restrict__ bval,
                                                             const unsigned *__restrict__ bmeta, const int *__restrict__ hdr, const int *__restrict__ order,
                                                             const T *__restrict__ x, T *__restrict__ y, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blk_y_lds[];
    double *ys = reinterpret_cast<double *>(blk_y_lds);
    constexpr int EPL = 16 / (int) sizeof(T);
    constexpr int NT = kWave * W;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = order[blockIdx.x];
#ifdef SPMV_BLK_DEBUG_FORMS
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const BlkDir d = dir[blk];
    if (accumulate && d.ns == 0) return;
    for (int i = tid; i <= R; i += NT) ys[i] = 0.0;
    __syncthreads();
    const long long e0 = d.g0 * (long long) (kWave * EPL);
    const T *__restrict__ bv = bval + e0;
    const unsigned *__restrict__ bm = bmeta + e0;
    const int *__restrict__ hd = hdr + d.g0;
    static_assert(kBlkStepGroups % UN == 0, "a block's region is a whole number of steps");
    const int nsteps = d.ns / UN;
    const int mine = nsteps > wave ? (nsteps - wave + W - 1) / W : 0; // steps wave, wave + W, ...
    const int phases = (nsteps + W - 1) / W;
    if (ORD ? phases > 0 : mine > 0) {
        BlkStep<T, UN> g0, g1, g2;
        T x0[UN][EPL], x1[UN][EPL];
        auto gather = [&](const BlkStep<T, UN> &g, T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const T *__restrict__ xs = x + __builtin_amdgcn_readlane(g.h, u);
#pragma unroll
                for (int j = 0; j < EPL; ++j) xv[u][j] = xs[g.w[u][j] & 0xffffu];
            }
        };
        auto add = [&](const BlkStep<T, UN> &g, const T(&xv)[UN][EPL]) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int j = 0; j < EPL; ++j) lds_add(&ys[g.w[u][j] >> 16], (double) (g.v[u][j] * xv[u][j]));
        };
        // step t of this wave = groups [(t W + wave) UN, ... + UN) of the region; steps past the block's end are loaded and gathered
        // (the next block's region or the padding behind the last one: valid columns), never added
        const int gw = wave * UN, gs = W * UN;
        blk_load_step<T, UN>(gw, lane, bv, bm, hd, g0);
        blk_load_step<T, UN>(gw + gs, lane, bv, bm, hd, g1);
        gather(g0, x0);
        const int last = ORD ? phases : mine;
#define SPMV_BLK_WPHASE(ga, gb, gc, xa, xbb)                                                                               \
        blk_load_step<T, UN>(gw + (t + 2) * gs, lane, bv, bm, hd, gc);                                                     \
        gather(gb, xbb);                                                                                                   \
        if constexpr (ORD) {                                                                                               \
            for (int w = 0; w < W; ++w) {                                                                                  \
                if (w == wave && t < mine) add(ga, xa);                                                                    \
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                            \
            }                                                                                                              \
        } else {                                                                                                           \
            add(ga, xa);                                                                                                   \
        }                                                                                                                  \
        if (++t >= last) break;
        for (int t = 0;;) {
            SPMV_BLK_WPHASE(g0, g1, g2, x0, x1)
            SPMV_BLK_WPHASE(g1, g2, g0, x1, x0)
            SPMV_BLK_WPHASE(g2, g0, g1, x0, x1)
            SPMV_BLK_WPHASE(g0, g1, g2, x1, x0)
            SPMV_BLK_WPHASE(g1, g2, g0, x0, x1)
            SPMV_BLK_WPHASE(g2, g0, g1, x1, x0)
        }
#undef SPMV_BLK_WPHASE
    }
    __syncthreads();
    const long long r0 = row0[blk];
    const int nr = row0[blk + 1] - (int) r0;
    if (accumulate) for (int i = tid; i < nr; i += NT) y[r0 + i] += (T) ys[i];
    else for (int i = tid; i < nr; i += NT) y[r0 + i] = (T) ys[i];
#ifdef SPMV_BLK_DEBUG_FORMS
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        blk_dbg_times[4 * blockIdx.x] = dbg_t0;
        blk_dbg_times[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        blk_dbg_times[4 * blockIdx.x + 2] = (xcc & 0xf) | ((unsigned long long) hw << 8);
        blk_dbg_times[4 * blockIdx.x + 3] = (unsigned long long) blk | ((unsigned long long) d.ns << 32);
    }
#endif
}

} // namespace spmv
