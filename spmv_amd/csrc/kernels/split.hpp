// split.hpp -- A = A_near + A_far: the inspector kernels behind a handle whose matrix has locality in PART of its entries.
//
// The tile schedules stage the x windows of a tile group in LDS when the group's columns fit, and gather through L1/L2
// when they do not -- per GROUP.  A matrix that is banded except for every tenth row, or whose rows keep 90 % of their entries
// near the diagonal and send 10 % to hub columns (web graphs), stages nothing: one stray entry per tile is enough.  The
// row-block x column-slab executor (blocked.hpp) takes such a matrix whole, but pays for the local entries too: 12 B/nnz
// instead of 10, and runs of entries of one row inside a cell serialise its LDS adds (every tenth row random, 1e7 x 32: 1.67 ms
// against 1.38 ms for ALL rows random).  So the entries are split, once, at create:
//   near  entries whose column lies within +-half of their 256-row tile's centre column (the median of five entries sampled
//         across the tile: robust against the far entries themselves).  A_near keeps all m rows, every tile's span is at most
//         2 half <= the LDS budget by construction, so EVERY tile stages: the tile schedule runs at its banded-matrix rate;
//   far   the rest.  A_far keeps all m rows too and is multiplied by the blocked executor in ACCUMULATE mode (y += ...).
// y = A_near x (tile schedule, writes every row), then y += A_far x, on the handle's stream: deterministic.  The reference has no
// counterpart: its workers gather x wherever the columns point (parallel_balanced2_spmv.c:242-282).
#pragma once
#include "common.hpp"

namespace spmv {

constexpr int kSplitTileRows = 256;

__device__ __forceinline__ int median5(int a, int b, int c, int d, int e)
{
    int t;
#define SPMV_SORT2(x, y) if (x > y) { t = x; x = y; y = t; }
    SPMV_SORT2(a, b) SPMV_SORT2(d, e) SPMV_SORT2(a, c) SPMV_SORT2(b, c) SPMV_SORT2(a, d) SPMV_SORT2(c, d) SPMV_SORT2(b, e) SPMV_SORT2(b, c) SPMV_SORT2(c, d)
#undef SPMV_SORT2
    return c;
}

// centre[t] = median column of five entries spread over tile t's entries (-1: the tile has none)
static __global__ __launch_bounds__(kBlock) void split_center_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, int tiles, int *__restrict__ centre)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= tiles) return;
    const long long r0 = (long long) t * kSplitTileRows, r1 = min(r0 + kSplitTileRows, (long long) m);
    const long long p0 = rowptr[r0], p1 = rowptr[r1], len = p1 - p0;
    if (len <= 0) { centre[t] = -1; return; }
    centre[t] = median5(colidx[p0 + len / 6], colidx[p0 + len / 3], colidx[p0 + len / 2], colidx[p0 + 2 * len / 3], colidx[p0 + 5 * len / 6 < p1 ? p0 + 5 * len / 6 : p1 - 1]);
}

__device__ __forceinline__ bool split_is_near(int c, int centre, int half) { return centre >= 0 && c >= centre - half && c < centre + half; }

// near[r] = entries of row r inside its tile's window; 16 lanes sweep a row
static __global__ __launch_bounds__(kBlock) void split_count_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, const int *__restrict__ centre, int half,
                                                             int *__restrict__ near)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < m; r += stride) {
        const int ct = centre[r / kSplitTileRows];
        int c = 0;
        for (int p = rowptr[r] + l; p < rowptr[r + 1]; p += 16) c += split_is_near(ld_stream(colidx + p), ct, half);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 16);
        if (l == 0) near[r] = c;
    }
}

// out[i] = block_off[i / kScanTile] + exclusive prefix of in[] inside the 1024-element tile (third pass of the scan in csr5.hpp);
// out[n] = total is written by the caller.  Also far_rp[i] = rowptr[i] - out[i] when far_rp != NULL (the other half's row pointer).
static __global__ __launch_bounds__(kBlock) void scan_apply_kernel(long long n, const int *__restrict__ in, const int *__restrict__ block_off, int *__restrict__ out,
                                                            const int *__restrict__ rowptr, int *__restrict__ far_rp)
{
    __shared__ int wave_tot[kBlock / kWave];
    __shared__ int slab_base;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x == 0) slab_base = block_off[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const long long i = (long long) blockIdx.x * kScanTile + k * kBlock + threadIdx.x;
        const int v = i < n ? in[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int off = slab_base;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < n) {
            out[i] = off + inc - v;
            if (far_rp) far_rp[i] = rowptr[i] - (off + inc - v);
        }
        __syncthreads();
        if (threadIdx.x == kBlock - 1) slab_base = off + inc;
        __syncthreads();
    }
}

// Stable split of every row's entries into the near and the far CSR (16 lanes sweep a row, 16 entries per pass, positions by
// ballot + popcount).  col_* == NULL: values only (spmv_hip_update_values).
template <typename T>
__global__ __launch_bounds__(kBlock) void split_scatter_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, const T *__restrict__ val,
                                                               const int *__restrict__ centre, int half, const int *__restrict__ rp_near, const int *__restrict__ rp_far,
                                                               int *__restrict__ col_near, T *__restrict__ val_near, int *__restrict__ col_far, T *__restrict__ val_far)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16, lane = threadIdx.x & (kWave - 1);
    const int shift = (lane / 16) * 16; // this 16-lane group's bits in the wave's ballot
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    const long long rows_padded = ((long long) m + stride - 1) / stride * stride; // every lane runs the same trip count: the ballots need the whole wave
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < rows_padded; r += stride) {
        const bool row_ok = r < m;
        const int p0 = row_ok ? rowptr[r] : 0, p1 = row_ok ? rowptr[r + 1] : 0;
        const int ct = row_ok ? centre[r / kSplitTileRows] : -1;
        int on = row_ok ? rp_near[r] : 0, of = row_ok ? rp_far[r] : 0;
        int passes = (p1 - p0 + 15) / 16;
#pragma unroll
        for (int o = 16; o < kWave; o <<= 1) passes = max(passes, __shfl_xor(passes, o, kWave)); // the four rows of a wave: same number of ballots
        for (int k = 0; k < passes; ++k) {
            const int p = p0 + k * 16 + l;
            const bool in = p < p1;
            const int c = in ? colidx[p] : 0;
            const bool nr = in && split_is_near(c, ct, half);
            const unsigned mn = (unsigned) (__ballot(nr) >> shift) & 0xffffu, mf = (unsigned) (__ballot(in && !nr) >> shift) & 0xffffu;
            const unsigned below = (1u << l) - 1u;
            if (nr) {
                const int q = on + __popc(mn & below);
                if (col_near) col_near[q] = c;
                val_near[q] = val[p];
            } else if (in) {
                const int q = of + __popc(mf & below);
                if (col_far) col_far[q] = c;
                val_far[q] = val[p];
            }
            on += __popc(mn);
            of += __popc(mf);
        }
    }
}

// Sampled estimate of the near share (before anything is built): 64 windows of 4096 consecutive entries, the window's centre =
// median of five of its entries, near = within +-half of it.  out[0] += near entries, out[1] += entries looked at.
static __global__ __launch_bounds__(kBlock) void split_sample_kernel(long long nnz, int windows, int wlen, const int *__restrict__ colidx, int half, unsigned long long *__restrict__ out)
{
    const long long start = windows > 1 ? (nnz - wlen) / (windows - 1) * blockIdx.x : 0;
    int c = 0;
    for (int i0 = 0; i0 < wlen; i0 += 1024) { // sub-windows of 1024 entries: about the entries of a 256-row tile
        const long long s = start + i0;
        const int ct = median5(colidx[s + 170], colidx[s + 341], colidx[s + 512], colidx[s + 682], colidx[s + 853]);
        for (int i = threadIdx.x; i < 1024; i += kBlock) c += split_is_near(colidx[s + i], ct, half);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(out, (unsigned long long) c);
    if (threadIdx.x == 0) atomicAdd(out + 1, (unsigned long long) wlen);
}

} // namespace spmv
