// nnz_split.hpp -- equal-nnz tiles with a segmented wavefront reduction and a carry fix-up.
//
// GPU schedule of Method_Balanced2 / Method_Balanced_Yid.  The reference's clean formulation is
// Balanced_Yid (parallel_balanced_Yid_spmv.c:16-53 inspector, :97-160 executor): worker i owns
// the nnz range [stride*i, stride*(i+1)); l = lower_bound(RowPtr, begin) is its first whole row,
// the piece in front of it belongs to row l-1 ("begin_val"), the last row may be cut
// ("end_val"), and a serial pass adds the pieces into y.  Here:
//
//   worker        = one 64-lane wavefront, tile = 64 lanes x 4 consecutive elements = 256 nnz,
//                   so every load of the matrix stream is a full 16 B/lane coalesced line;
//   inspector     = tile_first[t] = lower_bound(RowPtr, 256 t)   (4 B per tile, on the device);
//                   tile t OWNS the rows whose first nnz position lies in the tile: it zeroes
//                   the empty ones and starts a segment for the others;
//   executor      = per lane a 4-element running sum cut at row starts, then a backward
//                   segmented scan over the 64 lanes (ballot mask + 6 shuffles) joins the pieces
//                   of rows that span lanes; every row start writes y[row] exactly once;
//   carries       = the piece in front of the tile's first row start (the reference's
//                   begin_val) goes to carry[t]; nnz_fixup_kernel adds the carries of a row's
//                   continuation tiles in tile order -- deterministic, no float atomics.
//
// The reference's defects are not reproduced: row 0 receives its carries, leading /
// interior / trailing empty rows are written, T > nnz is fine.
//
// Extra HBM traffic on top of B_alg: 4 B (tile_first) + s B (carry) per 256 nnz  (< 0.5 %).
#pragma once
#include <climits>
#include "common.hpp"
#include "xwindows.hpp"

namespace spmv {

// consecutive elements per lane: 4 for fp64 (32 B of Val + 16 B of ColIdx per lane), 8 for fp32
// (32 B + 32 B) -- the same bytes in flight per wave for both types
template <typename T> struct SplitCfg { static constexpr int K = sizeof(T) == 4 ? 8 : 4; static constexpr int Tile = kWave * K; };
constexpr int kSplitK = 4;
constexpr int kSplitTile = kWave * kSplitK;   // fp64 tile (256 nnz); fp32 uses SplitCfg<float>::Tile = 512

// tile_first[t] = first row r with RowPtr[r] >= t*TILE, t = 0..ntiles-1; tile_first[ntiles] = m
// (the last tile also owns trailing empty rows).  *any_head is set when some row crosses a tile
// boundary, i.e. the fix-up pass has work.
__global__ __launch_bounds__(kBlock) void nnz_tile_first_kernel(int m, int ntiles, int tile_nnz, const int *__restrict__ rowptr,
                                                                int *__restrict__ tile_first,
                                                                int *__restrict__ any_head)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) { tile_first[t] = m; return; }
    const long long base = (long long) t * tile_nnz;
    const int r = lower_bound_dev(rowptr, m + 1, base);
    tile_first[t] = r;
    if (t > 0 && (long long) rowptr[r] > base) *any_head = 1; // row r-1 runs into this tile
}

// One tile by one wavefront.  seg: the wave's LDS mark array.  STAGED: x[lo, lo+span) is in LDS (xs).
template <typename T, bool STAGED>
__device__ __forceinline__ void nnz_tile(int t, int lane, int *__restrict__ seg, int nnz,
                                         const int *__restrict__ rowptr, const int *__restrict__ colidx,
                                         const unsigned short *__restrict__ col16,
                                         const T *__restrict__ val, const T *__restrict__ x,
                                         const T *__restrict__ xs, T *__restrict__ y,
                                         const int *__restrict__ tile_first, T *__restrict__ carry)
{
    constexpr int kSplitK = SplitCfg<T>::K;
    constexpr int kSplitTile = SplitCfg<T>::Tile;
    const long long base = (long long) t * kSplitTile;
    const long long p = base + lane * kSplitK;

    // 1. matrix stream: K consecutive elements per lane, 16 B loads
    int c[kSplitK];
    T v[kSplitK];
    if (base + kSplitTile <= nnz) {
#pragma unroll
        for (int q = 0; q < kSplitK; q += 4) {
            if (STAGED) { // 16-bit LDS slots, four per 8-byte load
                int w[4];
                ld_stream4(col16 + p + q, w);
                c[q] = (int) lds_slot<0>(w); c[q + 1] = (int) lds_slot<1>(w); c[q + 2] = (int) lds_slot<2>(w); c[q + 3] = (int) lds_slot<3>(w);
            } else {
                ld_stream4(colidx + p + q, *reinterpret_cast<int(*)[4]>(&c[q]));
            }
            ld_stream4(val + p + q, *reinterpret_cast<T(*)[4]>(&v[q]));
        }
    } else { // last, partial tile
#pragma unroll
        for (int k = 0; k < kSplitK; ++k) {
            const bool in = p + k < nnz;
            c[k] = in ? (STAGED ? (int) col16[p + k] : colidx[p + k]) : 0;
            v[k] = in ? val[p + k] : T(0);
        }
    }
    // 2. gather x (slots past nnz are masked in step 4, so x[0] / xs[0] can never leak)
    T xv[kSplitK];
#pragma unroll
    for (int k = 0; k < kSplitK; ++k) xv[k] = STAGED ? xs[c[k]] : x[c[k]]; // STAGED: colidx holds LDS slots

    // 3. rows owned by this tile: zero the empty ones, mark the start of the others
    const int rf = tile_first[t];
    const int rl = tile_first[t + 1];
    {
        const i32x4 minus1 = {-1, -1, -1, -1};
#pragma unroll
        for (int q = 0; q < kSplitK; q += 4) *reinterpret_cast<i32x4 *>(seg + lane * kSplitK + q) = minus1;
    }
    wave_lds_sync();
    for (int i = lane; i < rl - rf; i += kWave) {
        const int r = rf + i;
        const int s = rowptr[r], e = rowptr[r + 1];
        if (e == s) y[r] = T(0);
        else seg[s - (int) base] = i;
    }
    wave_lds_sync();
    int mk[kSplitK];
#pragma unroll
    for (int q = 0; q < kSplitK; q += 4) {
        const i32x4 mk4 = *reinterpret_cast<const i32x4 *>(seg + lane * kSplitK + q);
        mk[q] = mk4.x; mk[q + 1] = mk4.y; mk[q + 2] = mk4.z; mk[q + 3] = mk4.w;
    }
    wave_lds_sync(); // marks are in registers before the next tile clears the array

    // 4. per-lane running sum, cut at row starts
    T head = 0, acc = 0;
    int cur = -1; // row offset (from rf) of the segment open in this lane, -1: none started here
#pragma unroll
    for (int k = 0; k < kSplitK; ++k) {
        if (mk[k] >= 0) {
            if (cur >= 0) y[rf + cur] = acc; // segment began and ended inside this lane
            else head = acc;                 // piece of a segment begun in an earlier lane/tile
            cur = mk[k];
            acc = 0;
        }
        if (p + k < nnz) acc = fmadd(v[k], xv[k], acc);
    }
    if (cur < 0) { head = acc; acc = 0; }

    // 5. backward segmented scan of the heads: B[i] = head[i] + ... + head[j], j = first lane
    //    >= i that holds a row start (or 63)
    const unsigned long long starts = __ballot(cur >= 0);
    T B = head;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const T nb = __shfl_down(B, d, kWave);
        const bool cut = ((starts >> lane) & ((1ull << d) - 1ull)) != 0ull;
        if (!cut && lane + d < kWave) B += nb;
    }
    T next = __shfl_down(B, 1, kWave);
    if (lane == kWave - 1) next = 0;
    // 6. each lane's last row start owns everything up to the next start (or the tile end:
    //    then the following tiles' carries complete the row in the fix-up pass)
    if (cur >= 0) y[rf + cur] = acc + next;
    if (lane == 0) carry[t] = B; // piece in front of the tile's first row start (0 if none)
}

template <typename T>
__global__ __launch_bounds__(kBlock) void nnz_split_kernel(int m, int nnz, int ntiles,
                                                           const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx,
                                                           const T *__restrict__ val,
                                                           const T *__restrict__ x, T *__restrict__ y,
                                                           const int *__restrict__ tile_first,
                                                           T *__restrict__ carry)
{
    __shared__ __attribute__((aligned(16))) int seg_lds[kBlock / kWave][SplitCfg<T>::Tile];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int waves_total = gridDim.x * (kBlock / kWave);
    (void) m;
    for (int t = blockIdx.x * (kBlock / kWave) + wave; t < ntiles; t += waves_total)
        nnz_tile<T, false>(t, lane, seg_lds[wave], nnz, rowptr, colidx, nullptr, val, x, nullptr, y, tile_first, carry);
}

// ---- LDS-staged x windows (xwindows.hpp): a workgroup owns kSplitGroupTiles consecutive tiles; the
// columns of their nnz range are covered by up to 16 windows and `col16` holds the 16-bit LDS slots
// of the staged groups' entries (2 B/nnz instead of ColIdx's 4); other groups read ColIdx.
constexpr int kSplitGroupTiles = 16;

template <typename T>
__global__ __launch_bounds__(kBlock) void nnz_group_kernel(int nnz, int ntiles, const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, const unsigned short *__restrict__ col16,
                                                           const T *__restrict__ val,
                                                           const TileWindows *__restrict__ wins,
                                                           const T *__restrict__ x, T *__restrict__ y,
                                                           const int *__restrict__ tile_first, T *__restrict__ carry)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nnz_x_lds[];
    T *xs = reinterpret_cast<T *>(nnz_x_lds);
    __shared__ __attribute__((aligned(16))) int seg_lds[kBlock / kWave][SplitCfg<T>::Tile];
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0;
    stage_windows<kBlock, T>(tw, x, xs);
    if (staged) __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t0 = blockIdx.x * kSplitGroupTiles;
    for (int k = wave; k < kSplitGroupTiles; k += kBlock / kWave) {
        const int t = t0 + k;
        if (t >= ntiles) break;
        if (staged) nnz_tile<T, true>(t, lane, seg_lds[wave], nnz, rowptr, colidx, col16, val, x, xs, y, tile_first, carry);
        else nnz_tile<T, false>(t, lane, seg_lds[wave], nnz, rowptr, colidx, col16, val, x, xs, y, tile_first, carry);
    }
}

// One lane per tile t >= 1.  If row hr = tile_first[t]-1 started in tile t-1 and runs into tile t,
// this lane adds carry[t..te] (te = last tile the row reaches) into y[hr], in tile order.
template <typename T>
__global__ __launch_bounds__(kBlock) void nnz_fixup_kernel(int ntiles, const int *__restrict__ rowptr,
                                                           const int *__restrict__ tile_first,
                                                           const T *__restrict__ carry, T *__restrict__ y)
{
    constexpr int kSplitTile = SplitCfg<T>::Tile;
    const int t = blockIdx.x * kBlock + threadIdx.x + 1;
    if (t >= ntiles) return;
    const long long base = (long long) t * kSplitTile;
    const int hr = tile_first[t] - 1;             // >= 0: RowPtr[0] = 0 < base
    const long long hend = rowptr[hr + 1];
    if (hend <= base) return;                     // nothing in front of the first row start
    if ((long long) rowptr[hr] < base - kSplitTile) return; // row began earlier: not the run's first tile
    const int te = (int) ((hend - 1) / kSplitTile);
    T sum = 0;
    for (int u = t; u <= te; ++u) sum += carry[u];
    y[hr] += sum;
}

} // namespace spmv
