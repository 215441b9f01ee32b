// csr_vector_tile.hpp -- CSR-vector over row tiles with LDS-staged x windows.
//
// Same schedule as csr_vector_pipe_kernel (L lanes per row over plain CSR; reference:
// parallel_spmv.c:12-18 + inner_spmv.h:232-286), organised like the tile kernels that measured
// best on this chip:
//   - a 256-thread workgroup owns a tile of consecutive rows (256 for Method_Parallel, one
//     equal-nnz row block for Method_Balanced), a wave 64 of them at a time (L steps of 64/L rows);
//   - the 65 RowPtr values a wave needs are read by ONE coalesced load and handed to the lane
//     groups through LDS (no per-row dependent RowPtr load in front of the matrix stream);
//   - x WINDOWS (north_star: "LDS staging of x-vector tiles"): the inspector
//     (csr_tile_windows_kernel) finds, per tile, up to 16 column windows that cover every column
//     its rows reference -- one window when the plain span [min, max] fits the LDS budget (banded
//     matrices), several when the columns sit in a few far-apart bands (3-D stencils: 9 windows of
//     ~260 columns for a 27-point stencil) -- and writes a tile-local copy of ColIdx in which each
//     entry already IS the LDS slot of its column, stored in 16 BITS (a slot is < 48 KiB / sizeof(T)):
//     the staged tiles' column stream is 2 B/nnz instead of ColIdx's 4 B/nnz, i.e. 10 instead of
//     12 B/nnz for fp64.  The executor stages the windows once per tile (coalesced) and every
//     gather is xs[slot], with no index arithmetic at all.  Tiles whose columns do not fit read
//     the original ColIdx and gather from L1/L2;
//   - RUN tiles: a staged tile whose rows each reference ONE run of consecutive columns (banded matrices, 1-D stencils, dense row
//     blocks: col[p] = col[p0] + p - p0) needs no column stream at all: the LDS slot of a row's first column (16 bits per ROW,
//     row_slot) + the entry's position in the row IS the slot.  fp64: 8 B/nnz + 2 B/row instead of 10 B/nnz -- BASELINE config 2's
//     banded matrix streams 2.8 instead of 3.4 GB per multiply.  Same products, same order of additions: the same bits;
//   - matrix stream: 16 B lane loads, DEPTH steps in flight per wave;
//   - the 64 row sums of a wave are collected through LDS and written by ONE coalesced store.
// Rows longer than long_thr are left to the long-row path (a CSR5 plan over their sub-matrix, long_rows.hpp) and
// are excluded from the windows.
// Extra HBM held: col_local uint16[nnz] (read INSTEAD of ColIdx for staged tiles) and 200 B of window
// table per tile.
#pragma once
#include <climits>
#include <type_traits>
#include "common.hpp"
#include "csr_vector4.hpp"
#include "xwindows.hpp"

namespace spmv {

constexpr int kVecTileThreads = 256;           // 4 wavefronts per workgroup (512 measured no better)
constexpr int kVecTileRows = kVecTileThreads;  // rows per workgroup slab (64 per wave)
constexpr int kTmplMax = 64;                   // TEMPLATE tiles: most entries of a row (a power of two: positions are masked with kTmplMax - 1),
constexpr int kTmplCount = 8;                  // ... different offset lists per tile,
constexpr int kTmplRows = 1024;                // ... rows of a tile (the wide form's blocks; Balanced's equal-nnz blocks may hold more: those keep their stream)
// Row range of tile b: fixed 256-row tiles, or the equal-nnz blocks of `split` (Method_Balanced).
__device__ __forceinline__ void tile_rows(int b, int m, int rows_per_tile, const int *__restrict__ split, long long &r0, long long &r1)
{
    if (split) { r0 = split[b]; r1 = split[b + 1]; }
    else { r0 = (long long) b * rows_per_tile; r1 = r0 + rows_per_tile < m ? r0 + rows_per_tile : m; }
}

// Inspector: windows of one row tile + the tile-local 16-bit column stream (written for staged tiles
// only; unstaged tiles and long rows are computed from the original ColIdx).
static __global__ __launch_bounds__(kBlock) void csr_tile_windows_kernel(int m, int n, int rows_per_tile, int long_thr, int max_cols, int slot_bytes,
                                                                  const int *__restrict__ split,
                                                                  const int *__restrict__ rowptr,
                                                                  const int *__restrict__ colidx,
                                                                  TileWindows *__restrict__ wins,
                                                                  unsigned short *__restrict__ col_local,
                                                                  unsigned short *__restrict__ row_slot /* NULL: no run tiles */,
                                                                  unsigned char *__restrict__ col8 /* NULL: no byte tiles */,
                                                                  unsigned short *__restrict__ tmpl /* NULL: no template tiles; else kTmplCount lists of kTmplMax offsets per tile */,
                                                                  unsigned char *__restrict__ row_tid /* template tiles: the list number of every row */,
                                                                  int *__restrict__ staged /* [0] tiles staged, [1] max total, [2] run tiles, [3] their entries, [4] their rows, [5] byte tiles, [6] their entries, [7] their rows,
                                                                                              [8] template tiles, [9] their entries, [10] their rows */)
{
    long long r0, r1;
    tile_rows(blockIdx.x, m, rows_per_tile, split, r0, r1);
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16; // 16 lanes sweep a row
    auto loop = [&](auto body) {
        for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
            const int p0 = rowptr[r], p1 = rowptr[r + 1];
            if (p1 - p0 > long_thr) continue; // long rows are computed elsewhere, from the original ColIdx
            for (int p = p0 + l; p < p1; p += 16) body(colidx[p], (long long) p);
        }
    };
    auto store = [&](long long pos, int slot, int) { col_local[pos] = (unsigned short) (slot * slot_bytes); };
    build_windows(n, max_cols, loop, store, wins[blockIdx.x], staged, true);
    if (!row_slot) return;
    // RUN tile?  every (non-long) row one run of consecutive columns: then row_slot[r] = slot of the row's first column, in the column
    // stream's unit, and the executor never reads col_local for this tile
    __syncthreads(); // wins[blockIdx.x] as written by thread 0
    const TileWindows &tw = wins[blockIdx.x];
    const int nwin = tw.nwin;
    int ok = nwin > 0, entries = 0;
    if (nwin > 0)
        for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
            const int p0 = rowptr[r], p1 = rowptr[r + 1];
            if (p1 - p0 > long_thr || p1 == p0) continue;
            const int c0 = colidx[p0];
            for (int p = p0 + l; p < p1; p += 16) ok &= colidx[p] == c0 + (p - p0);
            if (l == 0) entries += p1 - p0;
        }
    ok = __syncthreads_and(ok);
    if (!ok) {
        // BYTE tile?  Every (non-long) row's slots lie within 255 slots of the row's smallest one -- banded matrices with holes, block rows,
        // anything whose rows span under 256 columns inside one window: the column stream is then ONE byte per entry (the slot's distance from
        // the row's smallest slot, col8) + 16 bits per row (that smallest slot, row_slot) instead of 16 bits per entry.  Every thread re-reads
        // the 16-bit slots it stored itself (build_windows ran the same loop), so no fence is needed in front of this.
        // TEMPLATE tile?  The tile's (non-long, non-empty) rows use at most kTmplCount different lists of slot offsets from their first entry (each of
        // at most kTmplMax entries) -- the interior of any stencil (a 27-point row: three windows, nine runs of three, the same 27 offsets in every
        // row) plus the few other lists of the rows at the grid's edges, block rows, anything assembled from a few element patterns.  Like a RUN tile
        // it reads no column stream at all: 16 bits (row_slot: the slot of the row's first entry) + 8 bits (row_tid: which list) per ROW and the
        // lists once per tile (tmpl); the slot of a row's k-th entry is row_slot + list[k].  RUN = the one list 0, 1, 2, ...
        if (tmpl && nwin > 0 && r1 - r0 <= kTmplRows) {
            __shared__ unsigned s_hash[kTmplCount];
            __shared__ int s_tlen[kTmplCount], s_first[kTmplCount], s_tm[kTmplCount][kTmplMax];
            __shared__ unsigned char s_ids[kTmplRows];
            for (int i = threadIdx.x; i < kTmplCount * kTmplMax; i += kBlock) s_tm[i / kTmplMax][i % kTmplMax] = 0;
            if (threadIdx.x < kTmplCount) { s_hash[threadIdx.x] = 0u; s_tlen[threadIdx.x] = 0; s_first[threadIdx.x] = INT_MAX; }
            __syncthreads();
            int okt = 1, nt = 0;
            auto on_row = [&](int len) { return len > 0 && len <= long_thr; };
            // A: every row hashes its list (length and offsets) and claims or finds one of the kTmplCount list numbers
            for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
                const int p0 = rowptr[r], len = rowptr[r + 1] - p0;
                int id = 0;
                if (on_row(len)) {
                    if (len > kTmplMax) okt = 0;
                    else {
                        const int s0 = col_local[p0];
                        unsigned h = 0x9E3779B9u * (unsigned) len;
                        for (int k = l; k < len; k += 16) {
                            const int dlt = (int) col_local[p0 + k] - s0;
                            okt &= dlt >= 0;
                            h += ((unsigned) dlt + 0x7F4A7C15u) * (2u * (unsigned) k + 1u) * 0x85EBCA6Bu;
                        }
#pragma unroll
                        for (int o = 8; o > 0; o >>= 1) h += __shfl_xor(h, o, 16);
                        h |= 1u; // 0 = a free list number
                        id = -1;
                        if (l == 0)
                            for (int i = 0; i < kTmplCount && id < 0; ++i) {
                                const unsigned old = atomicCAS(&s_hash[i], 0u, h);
                                if (old == 0u || old == h) id = i;
                            }
                        id = __shfl(id, 0, 16);
                        if (id < 0) { okt = 0; id = 0; } // more than kTmplCount different lists in this tile
                    }
                    if (l == 0) nt += len;
                }
                if (l == 0) s_ids[r - r0] = (unsigned char) id;
            }
            __syncthreads();
            // B, C: list i is written by the FIRST row (in row order) that carries it
            for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
                const int len = rowptr[r + 1] - rowptr[r];
                if (l == 0 && on_row(len) && len <= kTmplMax) atomicMin(&s_first[s_ids[r - r0]], (int) (r - r0));
            }
            __syncthreads();
            for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
                const int p0 = rowptr[r], len = rowptr[r + 1] - p0;
                if (!on_row(len) || len > kTmplMax) continue;
                const int id = s_ids[r - r0];
                if (s_first[id] == (int) (r - r0)) {
                    const int s0 = col_local[p0];
                    for (int k = l; k < len; k += 16) s_tm[id][k] = (int) col_local[p0 + k] - s0;
                    if (l == 0) s_tlen[id] = len;
                }
            }
            __syncthreads();
            // D: every row against its list (a hash is not a proof)
            for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
                const int p0 = rowptr[r], len = rowptr[r + 1] - p0;
                if (!on_row(len) || len > kTmplMax) continue;
                const int id = s_ids[r - r0], s0 = col_local[p0];
                okt &= s_tlen[id] == len;
                for (int k = l; k < len; k += 16) okt &= (int) col_local[p0 + k] - s0 == s_tm[id][k];
            }
            okt = __syncthreads_and(okt);
            if (okt) {
                for (long long r = r0 + sub * 16 + l; r < r1; r += kBlock) { // one thread per row
                    const int p0 = rowptr[r], len = rowptr[r + 1] - p0;
                    row_slot[r] = (unsigned short) (on_row(len) ? col_local[p0] : 0);
                    row_tid[r] = on_row(len) ? s_ids[r - r0] : (unsigned char) 0;
                }
                for (int i = threadIdx.x; i < kTmplCount * kTmplMax; i += kBlock)
                    tmpl[(size_t) blockIdx.x * (kTmplCount * kTmplMax) + i] = (unsigned short) s_tm[i / kTmplMax][i % kTmplMax];
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) nt += __shfl_xor(nt, o, kWave);
                if ((threadIdx.x & (kWave - 1)) == 0 && nt) atomicAdd(staged + 9, nt);
                if (threadIdx.x == 0) {
                    wins[blockIdx.x].runs = 3;
                    atomicAdd(staged + 8, 1);
                    atomicAdd(staged + 10, (int) (r1 - r0));
                }
                return;
            }
        }
        if (!col8 || nwin == 0) return;
        int okb = 1, nb = 0;
        for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
            const int p0 = rowptr[r], p1 = rowptr[r + 1];
            if (p1 - p0 > long_thr || p1 == p0) continue;
            int mn = INT_MAX, mx = 0;
            for (int p = p0 + l; p < p1; p += 16) { const int sl = col_local[p]; mn = sl < mn ? sl : mn; mx = sl > mx ? sl : mx; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 16)); mx = max(mx, __shfl_xor(mx, o, 16)); }
            okb &= mx - mn <= 255 * slot_bytes;
            if (l == 0) nb += p1 - p0;
        }
        okb = __syncthreads_and(okb);
        if (!okb) return;
        for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
            const int p0 = rowptr[r], p1 = rowptr[r + 1];
            int mn = INT_MAX;
            const bool on = p1 - p0 <= long_thr && p1 > p0;
            if (on) for (int p = p0 + l; p < p1; p += 16) { const int sl = col_local[p]; mn = sl < mn ? sl : mn; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 16));
            if (on) for (int p = p0 + l; p < p1; p += 16) col8[p] = (unsigned char) ((col_local[p] - mn) / slot_bytes);
            if (l == 0) row_slot[r] = (unsigned short) (on ? mn : 0);
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) nb += __shfl_xor(nb, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0 && nb) atomicAdd(staged + 6, nb);
        if (threadIdx.x == 0) {
            wins[blockIdx.x].runs = 2;
            atomicAdd(staged + 5, 1);
            atomicAdd(staged + 7, (int) (r1 - r0));
        }
        return;
    }
    for (long long r = r0 + sub * 16 + l; r < r1; r += kBlock) { // one thread per row now
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        int slot = 0;
        if (p1 > p0 && p1 - p0 <= long_thr) {
            const int c0 = colidx[p0];
            int w = 0;
            for (int k = 1; k < nwin; ++k) w = c0 >= tw.start[k] ? k : w;
            slot = (tw.base[w] + (c0 - tw.start[w])) * slot_bytes; // the whole run lies in window w: windows are maximal runs of touched 64-column segments
        }
        row_slot[r] = (unsigned short) slot;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) entries += __shfl_xor(entries, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && entries) atomicAdd(staged + 3, entries);
    if (threadIdx.x == 0) {
        wins[blockIdx.x].runs = 1;
        atomicAdd(staged + 2, 1);
        atomicAdd(staged + 4, (int) (r1 - r0));
    }
}

// Which four entries of a 4L-entry chunk a lane takes.  Every load instruction should cover one
// contiguous run of memory over the wave (a lane stride of 32 B -- four consecutive doubles per
// lane -- makes each 16-byte load touch twice the cache lines it uses).  So a lane takes 16 bytes of
// values per load: fp32 four consecutive entries (one load), fp64 two pairs, entries 2l, 2l+1 and
// 2L+2l, 2L+2l+1 (two loads, each contiguous over the L lanes of the row).
template <typename T, int L>
struct Lane4 {
    static constexpr int EPL = 16 / (int) sizeof(T); // entries per 16-byte load
    static __device__ __forceinline__ int pos(int l, int k) { return (k / EPL) * EPL * L + EPL * l + (k % EPL); }
    // v[0..3] = val[base + pos(l, k)]
    static __device__ __forceinline__ void load_val(const T *__restrict__ val, int base, int l, T (&v)[4])
    {
        if constexpr (EPL == 4) {
            ld_stream4(val + base + 4 * l, v);
        } else {
            const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + base + 2 * l));
            const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + base + 2 * L + 2 * l));
            v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
        }
    }
    // 16-bit slots, packed two per word in c[0], c[1] (see lds_slot)
    static __device__ __forceinline__ void load_col16(const unsigned short *__restrict__ col, int base, int l, int (&c)[4])
    {
        if constexpr (EPL == 4) {
            ld_stream4(col + base + 4 * l, c);
        } else {
            c[0] = __builtin_nontemporal_load(reinterpret_cast<const int *>(col + base + 2 * l));
            c[1] = __builtin_nontemporal_load(reinterpret_cast<const int *>(col + base + 2 * L + 2 * l));
        }
    }
    // BYTE tiles: one byte per entry (distance from the row's smallest slot); c[0] (fp32: four bytes) or c[0], c[1] (fp64: two bytes each)
    static __device__ __forceinline__ void load_col8(const unsigned char *__restrict__ col, int base, int l, int (&c)[4])
    {
        if constexpr (EPL == 4) {
            c[0] = __builtin_nontemporal_load(reinterpret_cast<const int *>(col + base + 4 * l));
        } else {
            c[0] = (int) __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(col + base + 2 * l));
            c[1] = (int) __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(col + base + 2 * L + 2 * l));
        }
    }
    template <int K> static __device__ __forceinline__ unsigned byte_of(const int (&c)[4])
    {
        if constexpr (EPL == 4) return ((unsigned) c[0] >> (8 * K)) & 0xffu;
        else return ((unsigned) c[K / 2] >> (8 * (K % 2))) & 0xffu;
    }
    static __device__ __forceinline__ void load_col32(const int *__restrict__ col, int base, int l, int (&c)[4])
    {
        if constexpr (EPL == 4) {
            ld_stream4(col + base + 4 * l, c);
        } else {
            const i32x2 a = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(col + base + 2 * l));
            const i32x2 b = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(col + base + 2 * L + 2 * l));
            c[0] = a.x; c[1] = a.y; c[2] = b.x; c[3] = b.y;
        }
    }
};

// One step of one lane group: lane l of the L lanes of row [p0, p1) multiplies its four entries
// (c, v: already loaded by Lane4 from the chunk starting at the 16 B-aligned position p0 & ~3) and, for
// rows longer than that chunk, walks on in chunks of 4L.  Returns the lane's partial sum (not yet reduced over the group).
// MODE 0: unstaged (global columns), 1: staged (16-bit slots from the column stream), 2: staged RUN tile (slots = rs + position in the row;
// cc0 unused), 3: staged BYTE tile (slots = rs + the entry's byte of the 8-bit column stream, packed in cc0), 4: staged TEMPLATE tile (slots = rs +
// tm[position in the row], tm = the tile's offset list in LDS; cc0 unused)
template <typename T, int L, int MODE, int SHIFT = 0>
__device__ __forceinline__ T csr_vector_step(int p0, int p1, int l, const int (&cc0)[4], const T (&vv0)[4],
                                             const int *__restrict__ colidx, const unsigned short *__restrict__ col_local,
                                             const T *__restrict__ val, const T *__restrict__ x,
                                             const unsigned char *__restrict__ xb, unsigned zoff, unsigned rs = 0, const unsigned char *__restrict__ col8 = nullptr,
                                             const unsigned short *__restrict__ tm = nullptr)
{
    constexpr bool STAGED = MODE != 0;
    constexpr unsigned INC = SHIFT ? 1u : (unsigned) sizeof(T); // slot unit of the stream: indices (wide form) or bytes
    // slot of entry e (e - p0 < row length) of a RUN tile's row
    auto run_slot = [&](int e) { return rs + (unsigned) (e - p0) * INC; };
    using LMB = Lane4<T, L>;
    // slot of the lane's k-th entry: from the row (RUN), the byte stream (BYTE) or the 16-bit stream
    auto slot_of = [&](auto kc, int e, const int (&cc)[4]) -> unsigned {
        constexpr int K = decltype(kc)::value;
        if constexpr (MODE == 2) return run_slot(e);
        else if constexpr (MODE == 4) return (rs & 0xffffu) + tm[(rs >> 16) + ((unsigned) (e - p0) & (unsigned) (kTmplMax - 1))]; // rs = slot | list number * kTmplMax << 16; out-of-row positions read a slot they never use
        else if constexpr (MODE == 3) return rs + LMB::template byte_of<K>(cc) * INC;
        else return lds_slot<K>(cc);
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    // SHIFT = 0: the stream holds LDS byte offsets; SHIFT = log2(sizeof(T)): slot indices (windows above 64 KiB)
    auto xat = [&](unsigned off) { return *reinterpret_cast<const T *>(xb + ((size_t) off << SHIFT)); };
    using LM = Lane4<T, L>;
    const int base = p0 & ~3;
    const int e0 = base + LM::pos(l, 0), e1 = base + LM::pos(l, 1), e2 = base + LM::pos(l, 2), e3 = base + LM::pos(l, 3); // ascending
    T sum = 0;
    if (STAGED) {
        const bool full = (e0 >= p0) & (e3 < p1), none = e0 >= p1;
        if (__all(full | none)) {
            if (full) {
                const T x0 = xat(slot_of(K0{}, e0, cc0)), x1 = xat(slot_of(K1{}, e1, cc0)), x2 = xat(slot_of(K2{}, e2, cc0)), x3 = xat(slot_of(K3{}, e3, cc0));
                sum = fmadd(vv0[0], x0, sum);
                sum = fmadd(vv0[1], x1, sum);
                sum = fmadd(vv0[2], x2, sum);
                sum = fmadd(vv0[3], x3, sum);
            }
        } else {
            const unsigned len = (unsigned) (p1 - p0); // entry e is in the row iff e - p0 < len (unsigned)
            const bool k0 = (unsigned) (e0 - p0) < len, k1 = (unsigned) (e1 - p0) < len, k2 = (unsigned) (e2 - p0) < len,
                       k3 = (unsigned) (e3 - p0) < len;
            const T x0 = xat(k0 ? slot_of(K0{}, e0, cc0) : zoff), x1 = xat(k1 ? slot_of(K1{}, e1, cc0) : zoff),
                    x2 = xat(k2 ? slot_of(K2{}, e2, cc0) : zoff), x3 = xat(k3 ? slot_of(K3{}, e3, cc0) : zoff);
            sum = fmadd(k0 ? vv0[0] : T(0), x0, sum);
            sum = fmadd(k1 ? vv0[1] : T(0), x1, sum);
            sum = fmadd(k2 ? vv0[2] : T(0), x2, sum);
            sum = fmadd(k3 ? vv0[3] : T(0), x3, sum);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + LM::pos(l, k);
            const bool ok = (e >= p0) & (e < p1);
            const int ci = ok ? cc0[k] : 0;
            const T xl = x[ci];
            sum = fmadd(ok ? vv0[k] : T(0), ok ? xl : T(0), sum);
        }
    }
    if (__any(base + 4 * L < p1)) { // some row of this step is longer than its first 4L-entry chunk
        for (int bb = base + 4 * L; __any(bb < p1); bb += 4 * L) {
            if (bb < p1) {
                int cc[4] = {0, 0, 0, 0};
                T v2[4];
                if (MODE == 1) LM::load_col16(col_local, bb, l, cc);
                else if (MODE == 0) LM::load_col32(colidx, bb, l, cc);
                else if (MODE == 3) LM::load_col8(col8, bb, l, cc);
                LM::load_val(val, bb, l, v2);
                if (MODE == 4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (bb + LM::pos(l, k) < p1) sum = fmadd(v2[k], xat((rs & 0xffffu) + tm[(rs >> 16) + ((unsigned) (bb + LM::pos(l, k) - p0) & (unsigned) (kTmplMax - 1))]), sum);
                } else if (MODE == 3) {
                    if (bb + LM::pos(l, 0) < p1) sum = fmadd(v2[0], xat(slot_of(K0{}, 0, cc)), sum);
                    if (bb + LM::pos(l, 1) < p1) sum = fmadd(v2[1], xat(slot_of(K1{}, 0, cc)), sum);
                    if (bb + LM::pos(l, 2) < p1) sum = fmadd(v2[2], xat(slot_of(K2{}, 0, cc)), sum);
                    if (bb + LM::pos(l, 3) < p1) sum = fmadd(v2[3], xat(slot_of(K3{}, 0, cc)), sum);
                } else if (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (bb + LM::pos(l, k) < p1) sum = fmadd(v2[k], xat(run_slot(bb + LM::pos(l, k))), sum);
                } else if (STAGED) {
                    if (bb + LM::pos(l, 0) < p1) sum = fmadd(v2[0], xat(lds_slot<0>(cc)), sum);
                    if (bb + LM::pos(l, 1) < p1) sum = fmadd(v2[1], xat(lds_slot<1>(cc)), sum);
                    if (bb + LM::pos(l, 2) < p1) sum = fmadd(v2[2], xat(lds_slot<2>(cc)), sum);
                    if (bb + LM::pos(l, 3) < p1) sum = fmadd(v2[3], xat(lds_slot<3>(cc)), sum);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (bb + LM::pos(l, k) < p1) sum = fmadd(v2[k], x[cc[k]], sum);
                }
            }
        }
    }
    return sum;
}

// The wave program.  STAGED: c[][0..1] hold four packed 16-bit BYTE offsets into xs (the inspector
// stores slot * sizeof(T)), so a gather is one extract + one ds_read with the LDS base as immediate;
// zoff is the byte offset of a slot that holds 0 (masked entries read it: 0 * 0, never x's NaN/Inf).
// A step whose lanes all hold four entries of their row, or none (regular matrices: every step),
// takes the unmasked path.  The order of the fused multiply-adds is the same on both paths.
template <typename T, int L, int MODE, int DEPTH, bool PRE = true, int SHIFT = 0>
__device__ __forceinline__ void csr_vector_tile_wave(long long row_end, int long_thr, long long rw0, int lane,
                                                     const int *__restrict__ rp_lds, const unsigned *__restrict__ rs_lds /* MODE >= 2: slot of each row's first column (MODE 4: | list number * kTmplMax << 16) */,
                                                     T *__restrict__ y_lds,
                                                     const int *__restrict__ colidx, const unsigned short *__restrict__ col_local,
                                                     const T *__restrict__ val,
                                                     const T *__restrict__ x, const T *__restrict__ xs, unsigned zoff,
                                                     T *__restrict__ y, const int (&c0)[4], const T (&v0)[4], const unsigned char *__restrict__ col8 = nullptr,
                                                     const unsigned short *__restrict__ tm = nullptr)
{
    constexpr int RW = kWave / L; // rows per step
    constexpr int D = DEPTH < L ? DEPTH : L; // steps of matrix stream in flight per wave
    const int l = lane % L, sub = lane / L;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(xs);
    int c[D][4];
    T v[D][4];
    int pp0[D], pp1[D];
    unsigned rs[D];
#pragma unroll
    for (int k = 0; k < D; ++k) rs[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[0][k] = c0[k]; v[0][k] = v0[k]; } // PRE: step 0 was issued before the barrier
    auto issue = [&](int s) { // RowPtr pair of step s from LDS, then its stream loads
        const int slot = s % D;
        pp0[slot] = rp_lds[s * RW + sub];
        pp1[slot] = rp_lds[s * RW + sub + 1];
        if (pp1[slot] - pp0[slot] > long_thr) pp1[slot] = pp0[slot];
        if (MODE >= 2) rs[slot] = rs_lds[s * RW + sub];
        if (s > 0 || !PRE) {
            const int an = pp0[slot] & ~3;
            if (MODE == 1) Lane4<T, L>::load_col16(col_local, an, l, c[slot]);
            else if (MODE == 0) Lane4<T, L>::load_col32(colidx, an, l, c[slot]);
            else if (MODE == 3) Lane4<T, L>::load_col8(col8, an, l, c[slot]);
            Lane4<T, L>::load_val(val, an, l, v[slot]);
        }
    };
#pragma unroll
    for (int s = 0; s < D && s < L; ++s) issue(s);
#pragma unroll
    for (int s = 0; s < L; ++s) {
        const int cur = s % D;
        const int p0 = pp0[cur], p1 = pp1[cur];
        T sum = csr_vector_step<T, L, MODE, SHIFT>(p0, p1, l, c[cur], v[cur], colidx, col_local, val, x, xb, zoff, rs[cur], col8, tm);
        sum = group_sum_dpp<L>(sum);
        if (l == 0) y_lds[s * RW + sub] = sum;
        if (s + D < L) issue(s + D); // refill the slot just consumed
    }
    wave_lds_sync();
    const long long row = rw0 + lane;
    const int len = rp_lds[lane + 1] - rp_lds[lane];
    if (row < row_end && len <= long_thr) y[row] = y_lds[lane]; // one coalesced 64-row store per wave
}

template <typename T, int L, int DEPTH = 4, bool PRE = true>
__global__ __launch_bounds__(kVecTileThreads) void csr_vector_tile_kernel(int m, int long_thr, const int *__restrict__ rowptr,
                                                                          const int *__restrict__ colidx,
                                                                          const unsigned short *__restrict__ col_local,
                                                                          const T *__restrict__ val,
                                                                          const TileWindows *__restrict__ wins,
                                                                          const unsigned short *__restrict__ row_slot,
                                                                          const unsigned char *__restrict__ col8,
                                                                          const unsigned short *__restrict__ tmpl, const unsigned char *__restrict__ row_tid,
                                                                          const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[]; // the tile's staged x
    __shared__ unsigned short tm_lds[kTmplCount * kTmplMax];
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
    __shared__ unsigned rs_lds[kVecTileThreads / kWave][kWave];
    __shared__ T y_lds[kVecTileThreads / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long rw0 = (long long) blockIdx.x * kVecTileRows + wave * kWave;
    // RowPtr of the wave's 64 rows (+1): one coalesced load, rows past m repeat RowPtr[m]
    long long r = rw0 + lane, re = rw0 + kWave;
    if (r > m) r = m;
    if (re > m) re = m;
    const int rp = rowptr[r];
    const int rpe = rowptr[re]; // wave-uniform
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0, runs = tw.runs == 1, bytes = tw.runs == 2, templ = tw.runs == 3; // runs / bytes / templ imply staged
    // step 0 of the matrix stream is issued straight from registers (RowPtr handed over by
    // shuffles), BEFORE the x staging and the barrier, so neither sits in front of the first loads
    int c0[4] = {0, 0, 0, 0};
    T v0[4] = {T(0), T(0), T(0), T(0)};
    if (PRE) {
        const int sub = lane / L, l = lane % L;
        const int q0 = __shfl(rp, sub, kWave);
        const int a = q0 & ~3;
        if (runs | templ) {}
        else if (bytes) Lane4<T, L>::load_col8(col8, a, l, c0);
        else if (staged) Lane4<T, L>::load_col16(col_local, a, l, c0);
        else Lane4<T, L>::load_col32(colidx, a, l, c0);
        Lane4<T, L>::load_val(val, a, l, v0);
    }
    rp_lds[wave][lane] = rp;
    if (lane == 0) rp_lds[wave][kWave] = rpe;
    if (runs | bytes) rs_lds[wave][lane] = row_slot[r]; // r <= m: the array has m + 1 entries
    if (templ) {
        rs_lds[wave][lane] = (unsigned) row_slot[r] | ((unsigned) row_tid[r] * kTmplMax << 16);
        for (int i = threadIdx.x; i < kTmplCount * kTmplMax; i += kVecTileThreads) tm_lds[i] = tmpl[(size_t) blockIdx.x * (kTmplCount * kTmplMax) + i];
    }
    stage_windows<kVecTileThreads, T>(tw, x, xs);
    if (threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of masked entries
    __syncthreads();
    if (rw0 >= m) return;
    const unsigned zoff = (unsigned) tw.total * (unsigned) sizeof(T);
    if (runs) csr_vector_tile_wave<T, L, 2, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
    else if (bytes) csr_vector_tile_wave<T, L, 3, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0, col8);
    else if (templ) csr_vector_tile_wave<T, L, 4, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0, col8, tm_lds);
    else if (staged) csr_vector_tile_wave<T, L, 1, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
    else csr_vector_tile_wave<T, L, 0, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
}

// Balanced form (Method_Balanced): the same wave program over EQUAL-NNZ row blocks.  Block b owns
// rows [split[b], split[b+1]) (init_csrSplitter_balanced2 semantics, parallel_balanced2_spmv.c:41-53,
// built by rowblock_split_kernel) and walks them in 256-row slabs, 64 rows per wave; the x windows
// of the whole block are staged once.
template <typename T, int L, int DEPTH = 4, bool WIDE = false>
__global__ __launch_bounds__(kVecTileThreads) void csr_vector_rows_kernel(int long_thr, const int *__restrict__ split,
                                                                          int rows_per_block, int m,
                                                                          const int *__restrict__ rowptr,
                                                                          const int *__restrict__ colidx,
                                                                          const unsigned short *__restrict__ col_local,
                                                                          const T *__restrict__ val,
                                                                          const TileWindows *__restrict__ wins,
                                                                          const unsigned short *__restrict__ row_slot,
                                                                          const unsigned char *__restrict__ col8,
                                                                          const unsigned short *__restrict__ tmpl, const unsigned char *__restrict__ row_tid,
                                                                          const T *__restrict__ x, T *__restrict__ y)
{
    __shared__ unsigned short tm_lds[kTmplCount * kTmplMax];
    // WIDE: x windows above 64 KiB (fp64 rows whose columns scatter over thousands of columns): the column
    // stream holds slot INDICES instead of byte offsets; blocks are `rows_per_block` consecutive rows
    // (split == NULL) so that one staging serves more rows.
    constexpr int SHIFT = WIDE ? (sizeof(T) == 8 ? 3 : 2) : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[];
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
    __shared__ unsigned rs_lds[kVecTileThreads / kWave][kWave];
    __shared__ T y_lds[kVecTileThreads / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    long long r_begin, r_end;
    tile_rows(blockIdx.x, m, rows_per_block, split, r_begin, r_end);
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0, runs = tw.runs == 1, bytes = tw.runs == 2, templ = tw.runs == 3;
    if (templ) for (int i = threadIdx.x; i < kTmplCount * kTmplMax; i += kVecTileThreads) tm_lds[i] = tmpl[(size_t) blockIdx.x * (kTmplCount * kTmplMax) + i];
    stage_windows<kVecTileThreads, T>(tw, x, xs);
    if (threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of masked entries
    __syncthreads();
    const unsigned zoff = WIDE ? (unsigned) tw.total : (unsigned) tw.total * (unsigned) sizeof(T);
    const int c0[4] = {0, 0, 0, 0};
    const T v0[4] = {T(0), T(0), T(0), T(0)};
    for (long long rw0 = r_begin + (long long) wave * kWave; rw0 < r_end; rw0 += kVecTileRows) {
        long long r = rw0 + lane, re = rw0 + kWave;
        if (r > r_end) r = r_end;
        if (re > r_end) re = r_end;
        rp_lds[wave][lane] = rowptr[r];
        if (lane == 0) rp_lds[wave][kWave] = rowptr[re];
        if (runs | bytes) rs_lds[wave][lane] = row_slot[r];
        if (templ) rs_lds[wave][lane] = (unsigned) row_slot[r] | ((unsigned) row_tid[r] * kTmplMax << 16);
        wave_lds_sync();
        if (runs) csr_vector_tile_wave<T, L, 2, DEPTH, false, SHIFT>(r_end, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
        else if (bytes) csr_vector_tile_wave<T, L, 3, DEPTH, false, SHIFT>(r_end, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0, col8);
        else if (templ) csr_vector_tile_wave<T, L, 4, DEPTH, false, SHIFT>(r_end, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0, col8, tm_lds);
        else if (staged) csr_vector_tile_wave<T, L, 1, DEPTH, false, SHIFT>(r_end, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
        else csr_vector_tile_wave<T, L, 0, DEPTH, false, SHIFT>(r_end, long_thr, rw0, lane, rp_lds[wave], rs_lds[wave], y_lds[wave], colidx, col_local, val, x, xs, zoff, y, c0, v0);
        wave_lds_sync(); // y_lds / rp_lds are reused by the next slab
    }
}

} // namespace spmv
