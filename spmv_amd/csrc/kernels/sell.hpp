// sell.hpp -- SELL-C-sigma with C = 64 (one wavefront per chunk, lane = row).
//
// GPU schedule of Method_SellCSigma.  Reference: sell_C_Sigma_spmv.c -- inspector :61-130 and
// :141-247 (sigma-window sort by row length with qsort, per-chunk column-major ValT/ColIndex with
// -1 padding, ld[] prefix of chunk widths, RowIndex permutation), executor :249-352 and the chunk
// kernel basic_d_lineProductGather_avx2 (inner_spmv.h:448-477).  The reference is called with
// C = 4 and sigma ~ m/nthreads (common.c:139-140); here C = 64 so that element j of all rows of a
// chunk is ONE contiguous 256 B (fp32) / 512 B (fp64) line, and sigma = 1024 (16 chunks).
//
// HBM layout (global arrays instead of the reference's per-window mallocs):
//   perm      int32[nchunks*64]   original row of each sorted slot, -1 = no row (padding slot,
//                                 or a LONG row handled by the long-row path, kernels/long_rows.hpp)
//   chunk_ptr int64[nchunks+1]    prefix sum of chunk widths (in columns; element offset = *64)
//   col / val [chunk_ptr[nchunks]*64]  column-major inside a chunk: (row slot l, j) at
//                                 (chunk_ptr[c] + j)*64 + l;  padding: col = -1, val = 0
//   long rows                     rows longer than `long_thr` are excluded from the slabs (they would
//                                 pad their whole chunk to their length) and computed by
//                                 the long-row path (their CSR5 sub-matrix, kernels/long_rows.hpp) -- the
//                                 analogue of the reference's CSR remainder loop
//                                 (sell_C_Sigma_spmv.c:289-298).
// Rows are sorted DESCENDING by length inside a window (the reference sorts ascending; either is
// fine) so a chunk's width is the length of its first slot.
#pragma once
#include <climits>
#include "common.hpp"
#include "xwindows.hpp"

namespace spmv {

constexpr int kSellC = kWave;

// One workgroup per sigma window: bitonic sort of (length, slot) keys in LDS.
static __global__ __launch_bounds__(kBlock) void sell_sort_kernel(int m, int sigma, int long_thr,
                                                           const int *__restrict__ rowptr,
                                                           int *__restrict__ perm,
                                                           int *__restrict__ chunk_width)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sell_lds[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(sell_lds);
    const long long w0 = (long long) blockIdx.x * sigma;
    for (int i = threadIdx.x; i < sigma; i += kBlock) {
        const long long row = w0 + i;
        unsigned len = 0, valid = 0;
        if (row < m) {
            len = (unsigned) (rowptr[row + 1] - rowptr[row]);
            valid = 1;
            if ((int) len > long_thr) { // handled by kernels/long_rows.hpp, not by the slabs
                len = 0;
                valid = 0;
            }
        }
        keys[i] = ((unsigned long long) len << 32) | ((unsigned long long) (sigma - 1 - i) << 1) | valid;
    }
    __syncthreads();
    for (int k = 2; k <= sigma; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < sigma; i += kBlock) {
                const int o = i ^ j;
                if (o > i) {
                    const unsigned long long a = keys[i], b = keys[o];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[o] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < sigma; i += kBlock) {
        const unsigned long long key = keys[i];
        const int slot = sigma - 1 - (int) ((key & 0xFFFFFFFFull) >> 1);
        perm[w0 + i] = (key & 1ull) ? (int) (w0 + slot) : -1;
        if ((i & (kSellC - 1)) == 0) chunk_width[(w0 + i) / kSellC] = (int) (key >> 32);
    }
}

// Exclusive prefix sum int32 -> int64 over n values (+ the total at out[n]); one workgroup walks
// the array in 256-element steps.  n is the chunk count (1.6e5 for 1e7 rows): inspector only.
static __global__ __launch_bounds__(kBlock) void scan_i32_to_i64_kernel(int n, const int *__restrict__ in,
                                                                 long long *__restrict__ out)
{
    __shared__ long long wave_tot[kBlock / kWave];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kBlock) {
        const int i = base + threadIdx.x;
        const long long v = i < n ? (long long) in[i] : 0;
        long long inc = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const long long o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        long long off = carry_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < n) out[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

// One wavefront per chunk: copy each slot's CSR row into the column-major slab.
template <typename T>
__global__ __launch_bounds__(kBlock) void sell_fill_kernel(int nchunks, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx,
                                                           const T *__restrict__ val,
                                                           const int *__restrict__ perm,
                                                           const long long *__restrict__ chunk_ptr,
                                                           int *__restrict__ scol, T *__restrict__ sval)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (c >= nchunks) return;
    const long long c0 = chunk_ptr[c];
    const int width = (int) (chunk_ptr[c + 1] - c0);
    const int row = perm[(long long) c * kSellC + lane];
    int p0 = 0, len = 0;
    if (row >= 0) { p0 = rowptr[row]; len = rowptr[row + 1] - p0; }
    for (int j = 0; j < width; ++j) {
        const size_t o = (size_t) (c0 + j) * kSellC + lane;
        const bool in = j < len;
        if (scol) scol[o] = in ? colidx[p0 + j] : -1; // NULL: values only (spmv_hip_update_values)
        sval[o] = in ? val[p0 + j] : T(0);
    }
}

// Executor: one wavefront per chunk, lane = row slot; every load of the slab is a full line.
template <typename T>
__global__ __launch_bounds__(kBlock) void sell_kernel(int nchunks, const long long *__restrict__ chunk_ptr,
                                                      const int *__restrict__ scol,
                                                      const T *__restrict__ sval,
                                                      const int *__restrict__ perm,
                                                      const T *__restrict__ x, T *__restrict__ y)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int waves_total = gridDim.x * (kBlock / kWave);
    for (int c = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; c < nchunks; c += waves_total) {
        const long long c0 = chunk_ptr[c];
        const int width = (int) (chunk_ptr[c + 1] - c0);
        const int *pc = scol + (size_t) c0 * kSellC + lane;
        const T *pv = sval + (size_t) c0 * kSellC + lane;
        T sum = 0;
        int j = 0;
        for (; j + 4 <= width; j += 4) { // 4 lines of each stream in flight per wave
            int cc[4];
            T vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cc[u] = ld_stream(pc + (size_t) (j + u) * kSellC);
                vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (cc[u] >= 0) sum = fmadd(vv[u], x[cc[u]], sum); // padding never touches x
        }
        for (; j < width; ++j) {
            const int cc = ld_stream(pc + (size_t) j * kSellC);
            const T vv = ld_stream(pv + (size_t) j * kSellC);
            if (cc >= 0) sum = fmadd(vv, x[cc], sum);
        }
        const int row = perm[(long long) c * kSellC + lane];
        if (row >= 0) y[row] = sum;
    }
}

// ---- LDS-staged x tiles (north_star: "SELL-C-sigma ... LDS-staged x tiles") ---------------------
// Sorting rows by length inside a sigma window puts rows that are up to sigma apart into adjacent
// lanes, so one gather instruction touches up to 64 different cache lines of x -- on a GPU that, not
// the padding, is what makes SELL slow on skewed matrices (measured 0.7 TB/s on config 4).  But the
// window as a whole only references x[lo, lo+span): when that span fits in LDS the workgroup stages
// it once (coalesced) and every gather becomes an LDS read.
//
// Inspector: range_windows_kernel (xwindows.hpp) over each sigma window's slab entries, in place on scol.
constexpr int kSellWinThreads = 512; // 8 wavefronts per sigma window

template <typename T, bool STAGED>
__device__ __forceinline__ void sell_chunk(const int *__restrict__ pc, const unsigned short *__restrict__ pc16,
                                           const T *__restrict__ pv, int width,
                                           const T *__restrict__ xs, const T *__restrict__ x, T &sum)
{
    // STAGED: the column stream is pc16, 16-bit LDS slots; padding points at the zero slot (0 * 0)
    constexpr int U = 8;
    int j = 0;
    for (; j + U <= width; j += U) {
        int cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (STAGED) cc[u] = ld_stream(pc16 + (size_t) (j + u) * kSellC);
            else cc[u] = ld_stream(pc + (size_t) (j + u) * kSellC);
            vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (STAGED) sum = fmadd(vv[u], xs[cc[u]], sum);
            else if (cc[u] >= 0) sum = fmadd(vv[u], x[cc[u]], sum);
        }
    }
    // the last 1..7 elements: one block whose loads are all issued before the first use (the chunk width is
    // wave-uniform, so the guards are scalar branches) -- a chunk of 5-entry rows costs one load round, not five
    const int r = __builtin_amdgcn_readfirstlane(width - j);
    if (r > 0) {
        int cc[U - 1];
        T vv[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) {
                if (STAGED) cc[u] = ld_stream(pc16 + (size_t) (j + u) * kSellC);
                else cc[u] = ld_stream(pc + (size_t) (j + u) * kSellC);
                vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
            }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) {
                if (STAGED) sum = fmadd(vv[u], xs[cc[u]], sum);
                else if (cc[u] >= 0) sum = fmadd(vv[u], x[cc[u]], sum);
            }
    }
}

// One workgroup per GROUP of consecutive sigma windows (chunks_per_win chunks); the group's x windows are
// staged in LDS when they fit.  A group is one sigma window unless staging is expensive next to the
// group's own stream (build_sell widens it then).
//
// y.  A group covers rows_per_group CONSECUTIVE original rows (sorting happens inside a sigma window), so the row sums are
// collected in LDS (ys, behind the x windows) and written by one coalesced sweep -- round 2 scattered them through `perm`, 64
// different lines per store instruction (0.67 of the HBM roofline on config 4's slabs).  Rows of the group that are not in
// the slabs -- the long rows, which the CSR5 sub-matrix launched BEHIND this kernel overwrites -- get the 0 the sweep finds.
// Chunks.  Sorted by width inside a sigma window, so the 8 waves take them from a counter in LDS, widest first (longest
// processing time first): a wave that drew a wide chunk simply draws fewer (round 2 dealt them by index, mirrored every other
// pass: 38 vs 23 columns per window between the busiest and the idlest wave before the mirroring, ~10 % after).
// RUN groups (as csr_vector_tile.hpp's RUN tiles, for the row-granular SELL slabs): when every row of a staged window group references ONE
// run of consecutive columns -- slot[j] = slot[0] + j for j < len -- the 16-bit slot slab is not read at all: a word per row slot
// (sell_run: first slot | row length << 16) replaces it, entries past the row's length read the zero slot as before (0 * 0).
template <typename T>
__device__ __forceinline__ void sell_chunk_run(unsigned run, const T *__restrict__ pv, int width, const T *__restrict__ xs, unsigned zslot, T &sum)
{
    constexpr int U = 8;
    const unsigned s0 = run & 0xffffu, len = run >> 16;
    int j = 0;
    for (; j + U <= width; j += U) {
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
#pragma unroll
        for (int u = 0; u < U; ++u) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + (unsigned) (j + u) : zslot], sum);
    }
    const int r = __builtin_amdgcn_readfirstlane(width - j);
    if (r > 0) {
        T vv[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + (unsigned) (j + u) : zslot], sum);
    }
}

// TEMPLATE groups (round 4; csr_vector_tile.hpp's TEMPLATE tiles for the SELL slabs): the rows of a staged window group use at most kSellTmplCount
// lists of slot offsets from their first entry (stencils: every interior row the same 27 offsets): no slot slab either; the word per row slot is
// first slot | row length << 16 | list number << 24, the lists (kSellTmplMax offsets each) sit in LDS.
constexpr int kSellTmplMax = 64, kSellTmplCount = 8;
template <typename T>
__device__ __forceinline__ void sell_chunk_tmpl(unsigned run, const T *__restrict__ pv, int width, const T *__restrict__ xs, unsigned zslot,
                                                const unsigned short *__restrict__ tm, T &sum)
{
    constexpr int U = 8;
    const unsigned s0 = run & 0xffffu, len = (run >> 16) & 0xffu;
    const unsigned short *__restrict__ list = tm + (run >> 24) * kSellTmplMax;
    int j = 0;
    for (; j + U <= width; j += U) {
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
#pragma unroll
        for (int u = 0; u < U; ++u) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + list[(j + u) & (kSellTmplMax - 1)] : zslot], sum);
    }
    const int r = __builtin_amdgcn_readfirstlane(width - j);
    if (r > 0) {
        T vv[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + list[(j + u) & (kSellTmplMax - 1)] : zslot], sum);
    }
}

// BYTE groups (round 4; csr_vector_tile.hpp's BYTE tiles for the SELL slabs): every row of a staged window group keeps its slots within 255 of its first one
// (bands with holes): an 8-bit slab of offsets from the row's first slot (scol8, same positions as the 16-bit slab) + the word per row slot
// (first slot | row length << 16) -- 1 B per stored entry instead of 2.
template <typename T>
__device__ __forceinline__ void sell_chunk_byte(unsigned run, const unsigned char *__restrict__ pc8, const T *__restrict__ pv, int width, const T *__restrict__ xs,
                                                unsigned zslot, T &sum)
{
    constexpr int U = 8;
    const unsigned s0 = run & 0xffffu, len = run >> 16;
    int j = 0;
    for (; j + U <= width; j += U) {
        unsigned cc[U];
        T vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cc[u] = ld_stream(pc8 + (size_t) (j + u) * kSellC);
            vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + cc[u] : zslot], sum);
    }
    const int r = __builtin_amdgcn_readfirstlane(width - j);
    if (r > 0) {
        unsigned cc[U - 1];
        T vv[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) {
                cc[u] = ld_stream(pc8 + (size_t) (j + u) * kSellC);
                vv[u] = ld_stream(pv + (size_t) (j + u) * kSellC);
            }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < r) sum = fmadd(vv[u], xs[(unsigned) (j + u) < len ? s0 + cc[u] : zslot], sum);
    }
}

// Inspector: is window group w a RUN group (wins[w].runs = 1), else a TEMPLATE group (3), else a BYTE group (2)?  One workgroup per group, a wave per chunk
// (lane = row slot) walks the row's slots in the slab.  counters[0] += groups, [1] += entries (real ones), [2] += stored slab entries, [3] += row slots of the
// RUN / TEMPLATE groups, [4] += entries of the TEMPLATE groups; BYTE groups: [5] += entries, [6] += stored slab entries, [7] += row slots.
static __global__ __launch_bounds__(kBlock) void sell_runs_kernel(int chunks_per_win, long long nchunks, const long long *__restrict__ chunk_ptr,
                                                           const unsigned short *__restrict__ scol16, const int *__restrict__ perm,
                                                           const int *__restrict__ rowptr, TileWindows *__restrict__ wins,
                                                           unsigned *__restrict__ sell_run, unsigned long long *__restrict__ counters,
                                                           unsigned short *__restrict__ tmpl /* NULL: no TEMPLATE groups; else kSellTmplCount lists of kSellTmplMax offsets per group */,
                                                           unsigned char *__restrict__ scol8 /* NULL: no BYTE groups; else the 8-bit slab */)
{
    __shared__ unsigned s_hash[kSellTmplCount];
    __shared__ int s_tlen[kSellTmplCount], s_first[kSellTmplCount], s_tm[kSellTmplCount][kSellTmplMax];
    const int w = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const TileWindows &tw = wins[w];
    int ok = tw.nwin > 0;
    unsigned long long entries = 0, stored = 0, slots = 0;
    if (ok)
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) {
            const long long c = (long long) w * chunks_per_win + k;
            if (c >= nchunks) break;
            const long long c0 = chunk_ptr[c];
            const int width = (int) (chunk_ptr[c + 1] - c0);
            const int row = perm[c * kSellC + lane];
            const int len = row >= 0 ? rowptr[row + 1] - rowptr[row] : 0; // rows kept out of the slabs (long rows) have perm = -1
            const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
            const unsigned s0 = len > 0 ? pc[0] : (unsigned) tw.total;
            for (int j = 1; j < len && j < width; ++j) ok &= pc[(size_t) j * kSellC] == s0 + (unsigned) j;
            ok &= len <= width && len < 65536;
            entries += (unsigned long long) len;
            if (lane == 0) { stored += (unsigned long long) width * kSellC; slots += kSellC; }
        }
    ok = __syncthreads_and(ok);
    if (!ok) {
        if (tw.nwin == 0) return;
        auto row_of = [&](int k, long long &c0, int &width, int &len) { // lane's row slot in chunk k of the group; false past the matrix
            const long long c = (long long) w * chunks_per_win + k;
            if (c >= nchunks) return false;
            c0 = chunk_ptr[c];
            width = (int) (chunk_ptr[c + 1] - c0);
            const int row = perm[c * kSellC + lane];
            len = row >= 0 ? rowptr[row + 1] - rowptr[row] : 0;
            return true;
        };
        // TEMPLATE group?  every row hashes (length, offsets from its first slot), claims or finds one of kSellTmplCount list numbers; list i is written
        // from the FIRST row slot (chunk, lane order) that carries it; every row is then compared with its list.
        auto template_stage = [&]() -> bool {
        for (int i = threadIdx.x; i < kSellTmplCount * kSellTmplMax; i += kBlock) s_tm[i / kSellTmplMax][i % kSellTmplMax] = 0;
        if (threadIdx.x < kSellTmplCount) { s_hash[threadIdx.x] = 0u; s_tlen[threadIdx.x] = 0; s_first[threadIdx.x] = INT_MAX; }
        __syncthreads();
        int okt = 1;
        unsigned long long tent = 0;
        auto hash_of = [&](const unsigned short *pc, int len, unsigned s0) {
            unsigned h = 0x9E3779B9u * (unsigned) len;
            for (int j = 0; j < len; ++j) h += ((unsigned) pc[(size_t) j * kSellC] - s0 + 0x7F4A7C15u) * (2u * (unsigned) j + 1u) * 0x85EBCA6Bu;
            return h | 1u;
        };
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) {
            long long c0; int width, len;
            if (!row_of(k, c0, width, len)) break;
            if (len == 0) continue;
            if (len > kSellTmplMax || len > width) { okt = 0; continue; }
            const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
            const unsigned s0 = pc[0], h = hash_of(pc, len, s0);
            int id = -1;
            for (int i = 0; i < kSellTmplCount && id < 0; ++i) {
                const unsigned old = atomicCAS(&s_hash[i], 0u, h);
                if (old == 0u || old == h) id = i;
            }
            if (id < 0) okt = 0;
            else atomicMin(&s_first[id], k * kSellC + lane);
            tent += (unsigned long long) len;
        }
        okt = __syncthreads_and(okt);
        if (!okt) return false;
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) { // the first carrier of each list writes it
            long long c0; int width, len;
            if (!row_of(k, c0, width, len)) break;
            if (len == 0) continue;
            const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
            const unsigned s0 = pc[0], h = hash_of(pc, len, s0);
            for (int i = 0; i < kSellTmplCount; ++i)
                if (s_hash[i] == h && s_first[i] == k * kSellC + lane) {
                    for (int j = 0; j < len; ++j) s_tm[i][j] = (int) pc[(size_t) j * kSellC] - (int) s0;
                    s_tlen[i] = len;
                }
        }
        __syncthreads();
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) { // every row against its list; its word
            long long c0; int width, len;
            if (!row_of(k, c0, width, len)) break;
            const long long c = (long long) w * chunks_per_win + k;
            unsigned word = (unsigned) tw.total;
            if (len > 0) {
                const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
                const unsigned s0 = pc[0], h = hash_of(pc, len, s0);
                int id = -1;
                for (int i = 0; i < kSellTmplCount; ++i) if (s_hash[i] == h) id = i;
                if (id < 0 || s_tlen[id] != len) okt = 0;
                else {
                    for (int j = 0; j < len; ++j) okt &= (int) pc[(size_t) j * kSellC] - (int) s0 == s_tm[id][j] && s_tm[id][j] >= 0;
                    word = s0 | ((unsigned) len << 16) | ((unsigned) id << 24);
                }
            }
            sell_run[c * kSellC + lane] = word;
        }
        okt = __syncthreads_and(okt);
        if (!okt) return false; // sell_run was scribbled on: harmless, only RUN / TEMPLATE / BYTE groups read it (and a BYTE group writes it again)
        for (int i = threadIdx.x; i < kSellTmplCount * kSellTmplMax; i += kBlock) tmpl[(size_t) w * (kSellTmplCount * kSellTmplMax) + i] = (unsigned short) s_tm[i / kSellTmplMax][i % kSellTmplMax];
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) { tent += __shfl_xor(tent, o, kWave); stored += __shfl_xor(stored, o, kWave); slots += __shfl_xor(slots, o, kWave); }
        if (lane == 0) { atomicAdd(counters + 1, tent); atomicAdd(counters + 2, stored); atomicAdd(counters + 3, slots); atomicAdd(counters + 4, tent); }
        if (threadIdx.x == 0) { wins[w].runs = 3; atomicAdd(counters, 1ull); }
        return true;
        };
        if (tmpl && template_stage()) return;
        if (!scol8) return;
        // BYTE group?  every slot of a row within 255 of the row's first (slots of a row ascend: sorted columns inside sorted windows; anything else fails the test)
        int okb = 1;
        unsigned long long bent = 0;
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) {
            long long c0; int width, len;
            if (!row_of(k, c0, width, len)) break;
            if (len == 0) continue;
            if (len > width || len >= 65536) { okb = 0; continue; }
            const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
            const unsigned s0 = pc[0];
            for (int j = 1; j < len; ++j) okb &= (unsigned) pc[(size_t) j * kSellC] - s0 < 256u;
            bent += (unsigned long long) len;
        }
        okb = __syncthreads_and(okb);
        if (!okb) return;
        for (int k = wave; k < chunks_per_win; k += kBlock / kWave) {
            long long c0; int width, len;
            if (!row_of(k, c0, width, len)) break;
            const long long c = (long long) w * chunks_per_win + k;
            const unsigned short *pc = scol16 + (size_t) c0 * kSellC + lane;
            unsigned char *p8 = scol8 + (size_t) c0 * kSellC + lane;
            const unsigned s0 = len > 0 ? pc[0] : (unsigned) tw.total;
            for (int j = 0; j < width; ++j) p8[(size_t) j * kSellC] = j < len ? (unsigned char) (pc[(size_t) j * kSellC] - s0) : (unsigned char) 0;
            sell_run[c * kSellC + lane] = s0 | ((unsigned) len << 16);
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) { bent += __shfl_xor(bent, o, kWave); stored += __shfl_xor(stored, o, kWave); slots += __shfl_xor(slots, o, kWave); }
        if (lane == 0) { atomicAdd(counters + 5, bent); atomicAdd(counters + 6, stored); atomicAdd(counters + 7, slots); }
        if (threadIdx.x == 0) { wins[w].runs = 2; atomicAdd(counters, 1ull); }
        return;
    }
    for (int k = wave; k < chunks_per_win; k += kBlock / kWave) {
        const long long c = (long long) w * chunks_per_win + k;
        if (c >= nchunks) break;
        const long long c0 = chunk_ptr[c];
        const int row = perm[c * kSellC + lane];
        const int len = row >= 0 ? rowptr[row + 1] - rowptr[row] : 0;
        const unsigned s0 = len > 0 ? scol16[(size_t) c0 * kSellC + lane] : (unsigned) tw.total;
        sell_run[c * kSellC + lane] = s0 | ((unsigned) len << 16);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { entries += __shfl_xor(entries, o, kWave); stored += __shfl_xor(stored, o, kWave); slots += __shfl_xor(slots, o, kWave); }
    if (lane == 0) { atomicAdd(counters + 1, entries); atomicAdd(counters + 2, stored); atomicAdd(counters + 3, slots); }
    if (threadIdx.x == 0) { wins[w].runs = 1; atomicAdd(counters, 1ull); }
}

template <typename T>
__global__ __launch_bounds__(kSellWinThreads) void sell_window_kernel(int chunks_per_win, long long nchunks, int m,
                                                                      const long long *__restrict__ chunk_ptr,
                                                                      const int *__restrict__ scol,
                                                                      const unsigned short *__restrict__ scol16,
                                                                      const T *__restrict__ sval,
                                                                      const int *__restrict__ perm,
                                                                      const TileWindows *__restrict__ wins,
                                                                      const unsigned *__restrict__ sell_run, const unsigned short *__restrict__ sell_tmpl,
                                                                      const unsigned char *__restrict__ scol8,
                                                                      const T *__restrict__ x, T *__restrict__ y, int ys_offset)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sell_x_lds[];
    __shared__ int next_chunk;
    __shared__ unsigned short tm_lds[kSellTmplCount * kSellTmplMax];
    T *xs = reinterpret_cast<T *>(sell_x_lds);
    T *ys = reinterpret_cast<T *>(sell_x_lds + ys_offset);
    const int w = blockIdx.x;
    const TileWindows &tw = wins[w];
    const bool staged = tw.nwin > 0, runs = tw.runs == 1, bytes = tw.runs == 2, templ = tw.runs == 3; // runs / bytes / templ imply staged
    if (templ) for (int i = threadIdx.x; i < kSellTmplCount * kSellTmplMax; i += kSellWinThreads) tm_lds[i] = sell_tmpl[(size_t) blockIdx.x * (kSellTmplCount * kSellTmplMax) + i];
    const int rows_per_group = chunks_per_win * kSellC;
    const long long row0 = (long long) w * rows_per_group;
    for (int i = threadIdx.x; i < rows_per_group; i += kSellWinThreads) ys[i] = T(0);
    if (threadIdx.x == 0) next_chunk = 0;
    stage_windows<kSellWinThreads, T>(tw, x, xs);
    if (staged && threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of padding entries
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1);
    for (;;) {
        int k = 0;
        if (lane == 0) k = atomicAdd(&next_chunk, 1);
        k = __builtin_amdgcn_readfirstlane(k);
        const long long c = (long long) w * chunks_per_win + k;
        if (k >= chunks_per_win || c >= nchunks) break;
        const long long c0 = chunk_ptr[c];
        const int width = (int) (chunk_ptr[c + 1] - c0);
        const int *pc = scol + (size_t) c0 * kSellC + lane;
        const unsigned short *pc16 = scol16 + (size_t) c0 * kSellC + lane;
        const T *pv = sval + (size_t) c0 * kSellC + lane;
        const int row = perm[c * kSellC + lane];
        T sum = 0;
        if (runs) sell_chunk_run<T>(sell_run[c * kSellC + lane], pv, width, xs, (unsigned) tw.total, sum);
        else if (templ) sell_chunk_tmpl<T>(sell_run[c * kSellC + lane], pv, width, xs, (unsigned) tw.total, tm_lds, sum);
        else if (bytes) sell_chunk_byte<T>(sell_run[c * kSellC + lane], scol8 + (size_t) c0 * kSellC + lane, pv, width, xs, (unsigned) tw.total, sum);
        else if (staged) sell_chunk<T, true>(pc, pc16, pv, width, xs, x, sum);
        else sell_chunk<T, false>(pc, pc16, pv, width, xs, x, sum);
        if (row >= 0) ys[row - row0] = sum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows_per_group; i += kSellWinThreads)
        if (row0 + i < m) y[row0 + i] = ys[i];
}

} // namespace spmv
