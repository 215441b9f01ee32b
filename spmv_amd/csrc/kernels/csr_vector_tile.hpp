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
//     entry already IS the LDS slot of its column.  The executor stages the windows once per tile
//     (coalesced) and every gather is xs[col_local], with no index arithmetic at all.  Tiles whose
//     columns do not fit keep global indices in col_local and gather from L1/L2;
//   - matrix stream: 16 B lane loads, DEPTH steps in flight per wave;
//   - the 64 row sums of a wave are collected through LDS and written by ONE coalesced store.
// Rows longer than long_thr are left to kernels/long_rows.hpp (which reads the ORIGINAL ColIdx) and
// are excluded from the windows.
// Extra HBM held: col_local int32[nnz] (read INSTEAD of ColIdx by this kernel: same traffic) and
// 200 B of window table per tile.
#pragma once
#include <climits>
#include "common.hpp"
#include "csr_vector4.hpp"
#include "xwindows.hpp"

namespace spmv {

constexpr int kVecTileThreads = 256;           // 4 wavefronts per workgroup (512 measured no better)
constexpr int kVecTileRows = kVecTileThreads;  // rows per workgroup slab (64 per wave)
// Row range of tile b: fixed 256-row tiles, or the equal-nnz blocks of `split` (Method_Balanced).
__device__ __forceinline__ void tile_rows(int b, int m, const int *__restrict__ split, long long &r0, long long &r1)
{
    if (split) { r0 = split[b]; r1 = split[b + 1]; }
    else { r0 = (long long) b * kVecTileRows; r1 = r0 + kVecTileRows < m ? r0 + kVecTileRows : m; }
}

// Inspector: windows of one row tile + the tile-local column copy.  col_local must be pre-filled
// with a copy of ColIdx (entries of unstaged tiles and of long rows keep their global value).
__global__ __launch_bounds__(kBlock) void csr_tile_windows_kernel(int m, int n, int long_thr, int max_cols,
                                                                  const int *__restrict__ split,
                                                                  const int *__restrict__ rowptr,
                                                                  const int *__restrict__ colidx,
                                                                  TileWindows *__restrict__ wins,
                                                                  int *__restrict__ col_local,
                                                                  int *__restrict__ staged /* [0] tiles staged, [1] max total */)
{
    long long r0, r1;
    tile_rows(blockIdx.x, m, split, r0, r1);
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16; // 16 lanes sweep a row
    auto loop = [&](auto body) {
        for (long long r = r0 + sub; r < r1; r += kBlock / 16) {
            const int p0 = rowptr[r], p1 = rowptr[r + 1];
            if (p1 - p0 > long_thr) continue; // long rows are computed elsewhere, from the original ColIdx
            for (int p = p0 + l; p < p1; p += 16) body(colidx[p], (long long) p);
        }
    };
    build_windows(n, max_cols, loop, col_local, wins[blockIdx.x], staged);
}

template <typename T, int L, bool STAGED, int DEPTH, bool PRE = true>
__device__ __forceinline__ void csr_vector_tile_wave(long long row_end, int long_thr, long long rw0, int lane,
                                                     const int *__restrict__ rp_lds, T *__restrict__ y_lds,
                                                     const int *__restrict__ col_local, const T *__restrict__ val,
                                                     const T *__restrict__ x, const T *__restrict__ xs,
                                                     T *__restrict__ y, const int (&c0)[4], const T (&v0)[4])
{
    constexpr int RW = kWave / L; // rows per step
    constexpr int D = DEPTH < L ? DEPTH : L; // steps of matrix stream in flight per wave
    const int l = lane % L, sub = lane / L;
    int c[D][4];
    T v[D][4];
    int pp0[D], pp1[D];
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[0][k] = c0[k]; v[0][k] = v0[k]; } // PRE: step 0 was issued before the barrier
    auto issue = [&](int s) { // RowPtr pair of step s from LDS, then its 16 B stream loads
        const int slot = s % D;
        pp0[slot] = rp_lds[s * RW + sub];
        pp1[slot] = rp_lds[s * RW + sub + 1];
        if (pp1[slot] - pp0[slot] > long_thr) pp1[slot] = pp0[slot];
        if (s > 0 || !PRE) {
            const int an = (pp0[slot] & ~3) + l * 4;
            ld_stream4(col_local + an, c[slot]);
            ld_stream4(val + an, v[slot]);
        }
    };
#pragma unroll
    for (int s = 0; s < D && s < L; ++s) issue(s);
#pragma unroll
    for (int s = 0; s < L; ++s) {
        const int cur = s % D;
        const int p0 = pp0[cur], p1 = pp1[cur];
        const int a = (p0 & ~3) + l * 4;
        T sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = (a + k >= p0) & (a + k < p1);
            const int ci = ok ? c[cur][k] : 0; // STAGED: an LDS slot; else a global column
            const T xl = STAGED ? xs[ci] : x[ci];
            sum = fmadd(ok ? v[cur][k] : T(0), ok ? xl : T(0), sum);
        }
        if (__any(a + 4 * L < p1)) { // some row of this step is longer than 4L
            for (int aa = a + 4 * L; __any(aa < p1); aa += 4 * L) {
                if (aa < p1) {
                    int cc[4];
                    T v2[4];
                    ld_stream4(col_local + aa, cc);
                    ld_stream4(val + aa, v2);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (aa + k < p1) sum = fmadd(v2[k], STAGED ? xs[cc[k]] : x[cc[k]], sum);
                }
            }
        }
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
                                                                          const int *__restrict__ col_local,
                                                                          const T *__restrict__ val,
                                                                          const TileWindows *__restrict__ wins,
                                                                          const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[]; // the tile's staged x
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
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
    // step 0 of the matrix stream is issued straight from registers (RowPtr handed over by
    // shuffles), BEFORE the x staging and the barrier, so neither sits in front of the first loads
    int c0[4] = {0, 0, 0, 0};
    T v0[4] = {T(0), T(0), T(0), T(0)};
    if (PRE) {
        const int sub = lane / L, l = lane % L;
        const int q0 = __shfl(rp, sub, kWave);
        const int a = (q0 & ~3) + l * 4;
        ld_stream4(col_local + a, c0);
        ld_stream4(val + a, v0);
    }
    rp_lds[wave][lane] = rp;
    if (lane == 0) rp_lds[wave][kWave] = rpe;
    const bool staged = tw.nwin > 0;
    stage_windows<kVecTileThreads, T>(tw, x, xs);
    __syncthreads();
    if (rw0 >= m) return;
    if (staged) csr_vector_tile_wave<T, L, true, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], col_local, val, x, xs, y, c0, v0);
    else csr_vector_tile_wave<T, L, false, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], col_local, val, x, xs, y, c0, v0);
}

// Balanced form (Method_Balanced): the same wave program over EQUAL-NNZ row blocks.  Block b owns
// rows [split[b], split[b+1]) (init_csrSplitter_balanced2 semantics, parallel_balanced2_spmv.c:41-53,
// built by rowblock_split_kernel) and walks them in 256-row slabs, 64 rows per wave; the x windows
// of the whole block are staged once.
template <typename T, int L, int DEPTH = 4>
__global__ __launch_bounds__(kVecTileThreads) void csr_vector_rows_kernel(int long_thr, const int *__restrict__ split,
                                                                          const int *__restrict__ rowptr,
                                                                          const int *__restrict__ col_local,
                                                                          const T *__restrict__ val,
                                                                          const TileWindows *__restrict__ wins,
                                                                          const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[];
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
    __shared__ T y_lds[kVecTileThreads / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long r_begin = split[blockIdx.x], r_end = split[blockIdx.x + 1];
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0;
    stage_windows<kVecTileThreads, T>(tw, x, xs);
    __syncthreads();
    const int c0[4] = {0, 0, 0, 0};
    const T v0[4] = {T(0), T(0), T(0), T(0)};
    for (long long rw0 = r_begin + (long long) wave * kWave; rw0 < r_end; rw0 += kVecTileRows) {
        long long r = rw0 + lane, re = rw0 + kWave;
        if (r > r_end) r = r_end;
        if (re > r_end) re = r_end;
        rp_lds[wave][lane] = rowptr[r];
        if (lane == 0) rp_lds[wave][kWave] = rowptr[re];
        wave_lds_sync();
        if (staged) csr_vector_tile_wave<T, L, true, DEPTH, false>(r_end, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], col_local, val, x, xs, y, c0, v0);
        else csr_vector_tile_wave<T, L, false, DEPTH, false>(r_end, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], col_local, val, x, xs, y, c0, v0);
        wave_lds_sync(); // y_lds / rp_lds are reused by the next slab
    }
}

} // namespace spmv
