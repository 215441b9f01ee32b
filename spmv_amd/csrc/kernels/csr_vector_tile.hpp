// csr_vector_tile.hpp -- CSR-vector over 256-row tiles with an LDS-staged x window.
//
// Same schedule as csr_vector_pipe_kernel (L lanes per row over plain CSR; reference:
// parallel_spmv.c:12-18 + inner_spmv.h:232-286), organised like the tile kernels that measured
// best on this chip:
//   - a 256-thread workgroup owns 256 consecutive rows, a wave 64 of them (L steps of 64/L rows);
//   - the 65 RowPtr values a wave needs are read by ONE coalesced load and handed to the lane
//     groups through LDS (no per-row dependent RowPtr load in front of the matrix stream);
//   - the columns those 256 rows reference span [lo, lo+span) (found once by the inspector,
//     csr_tile_span_kernel); when the span fits the LDS budget the workgroup stages x[lo..] once,
//     coalesced, and every gather is an LDS read -- north_star's "LDS staging of x-vector tiles";
//   - matrix stream: 16 B lane loads, two steps in flight (as in the pipe kernel);
//   - the 64 row sums of a wave are collected through LDS and written by ONE coalesced store.
// Rows longer than long_thr are left to kernels/long_rows.hpp and excluded from the span.
#pragma once
#include <climits>
#include "common.hpp"
#include "csr_vector4.hpp"

namespace spmv {

constexpr int kVecTileThreads = 256;           // 4 wavefronts per workgroup (512 measured no better)
constexpr int kVecTileRows = kVecTileThreads; // rows per workgroup (64 per wave)

// Inspector: per 256-row tile, min / max column over its rows with len <= long_thr.
__global__ __launch_bounds__(kBlock) void csr_tile_span_kernel(int m, int long_thr, int max_span,
                                                               const int *__restrict__ rowptr,
                                                               const int *__restrict__ colidx,
                                                               int *__restrict__ tile_lo, int *__restrict__ tile_span,
                                                               int *__restrict__ staged /* [0] count, [1] max span */)
{
    __shared__ int smin[kBlock / kWave], smax[kBlock / kWave];
    const long long r0 = (long long) blockIdx.x * kVecTileRows;
    int mn = INT_MAX, mx = -1;
    // lanes sweep the tile's rows cooperatively: 16 lanes per row keeps the reads mostly coalesced
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    for (int rr = sub; rr < kVecTileRows; rr += kBlock / 16) {
        const long long r = r0 + rr;
        if (r >= m) break;
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        if (p1 - p0 > long_thr) continue;
        for (int p = p0 + l; p < p1; p += 16) {
            const int c = colidx[p];
            mn = min(mn, c);
            mx = max(mx, c);
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, kWave));
        mx = max(mx, __shfl_xor(mx, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { smin[threadIdx.x / kWave] = mn; smax[threadIdx.x / kWave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / kWave; ++k) { mn = min(mn, smin[k]); mx = max(mx, smax[k]); }
        const long long span = mx >= mn ? (long long) mx - mn + 1 : 0;
        const bool ok = span > 0 && span <= max_span;
        tile_lo[blockIdx.x] = ok ? mn : 0;
        tile_span[blockIdx.x] = ok ? (int) span : 0;
        if (ok) { atomicAdd(staged, 1); atomicMax(staged + 1, (int) span); }
    }
}

template <typename T, int L, bool STAGED, int DEPTH, bool PRE = true>
__device__ __forceinline__ void csr_vector_tile_wave(long long row_end, int long_thr, long long rw0, int lane,
                                                     const int *__restrict__ rp_lds, T *__restrict__ y_lds,
                                                     const int *__restrict__ colidx, const T *__restrict__ val,
                                                     const T *__restrict__ x, const T *__restrict__ xs, int lo,
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
            ld_stream4(colidx + an, c[slot]);
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
            const int ci = ok ? c[cur][k] : (STAGED ? lo : 0);
            const T xl = STAGED ? xs[ci - lo] : x[ci];
            sum = fmadd(ok ? v[cur][k] : T(0), ok ? xl : T(0), sum);
        }
        if (__any(a + 4 * L < p1)) { // some row of this step is longer than 4L
            for (int aa = a + 4 * L; __any(aa < p1); aa += 4 * L) {
                if (aa < p1) {
                    int cc[4];
                    T v2[4];
                    ld_stream4(colidx + aa, cc);
                    ld_stream4(val + aa, v2);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (aa + k < p1) sum = fmadd(v2[k], STAGED ? xs[cc[k] - lo] : x[cc[k]], sum);
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
                                                                 const int *__restrict__ colidx,
                                                                 const T *__restrict__ val,
                                                                 const int *__restrict__ tile_lo,
                                                                 const int *__restrict__ tile_span,
                                                                 const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[]; // span elements of x
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
    __shared__ T y_lds[kVecTileThreads / kWave][kWave];
    constexpr int RW = kWave / L;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long rw0 = (long long) blockIdx.x * kVecTileRows + wave * kWave;
    // RowPtr of the wave's 64 rows (+1): one coalesced load, rows past m repeat RowPtr[m]
    long long r = rw0 + lane, re = rw0 + kWave;
    if (r > m) r = m;
    if (re > m) re = m;
    const int rp = rowptr[r];
    const int rpe = rowptr[re]; // wave-uniform
    const int lo = tile_lo[blockIdx.x], span = tile_span[blockIdx.x];
    // step 0 of the matrix stream is issued straight from registers (RowPtr handed over by
    // shuffles), BEFORE the x staging and the barrier, so neither sits in front of the first loads
    int c0[4] = {0, 0, 0, 0};
    T v0[4] = {T(0), T(0), T(0), T(0)};
    if (PRE) {
        const int sub = lane / L, l = lane % L;
        const int q0 = __shfl(rp, sub, kWave);
        const int nx = __shfl(rp, (sub + 1) & (kWave - 1), kWave);
        const int q1 = sub + 1 < kWave ? nx : rpe;
        const int a = (q0 & ~3) + l * 4;
        (void) q1;
        ld_stream4(colidx + a, c0);
        ld_stream4(val + a, v0);
    }
    rp_lds[wave][lane] = rp;
    if (lane == 0) rp_lds[wave][kWave] = rpe;
    for (int i = threadIdx.x; i < span; i += kVecTileThreads) xs[i] = x[lo + i];
    __syncthreads();
    if (rw0 >= m) return;
    if (span > 0) csr_vector_tile_wave<T, L, true, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], colidx, val, x, xs, lo, y, c0, v0);
    else csr_vector_tile_wave<T, L, false, DEPTH, PRE>((long long) m, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], colidx, val, x, xs, lo, y, c0, v0);
    (void) RW;
}

// Balanced form (Method_Balanced): the same wave program over EQUAL-NNZ row blocks.  Block b owns
// rows [split[b], split[b+1]) (init_csrSplitter_balanced2 semantics, parallel_balanced2_spmv.c:41-53,
// built by rowblock_split_kernel) and walks them in 256-row slabs, 64 rows per wave; the x span of
// the whole block is staged once.  Replaces the LDS-products kernel of rowblock.hpp as executor.
template <typename T, int L, int DEPTH = 4>
__global__ __launch_bounds__(kVecTileThreads) void csr_vector_rows_kernel(int long_thr, const int *__restrict__ split,
                                                                          const int *__restrict__ rowptr,
                                                                          const int *__restrict__ colidx,
                                                                          const T *__restrict__ val,
                                                                          const int *__restrict__ tile_lo,
                                                                          const int *__restrict__ tile_span,
                                                                          const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vec_x_lds[];
    T *xs = reinterpret_cast<T *>(vec_x_lds);
    __shared__ int rp_lds[kVecTileThreads / kWave][kWave + 2];
    __shared__ T y_lds[kVecTileThreads / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const long long r_begin = split[blockIdx.x], r_end = split[blockIdx.x + 1];
    const int lo = tile_lo[blockIdx.x], span = tile_span[blockIdx.x];
    for (int i = threadIdx.x; i < span; i += kVecTileThreads) xs[i] = x[lo + i];
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
        if (span > 0) csr_vector_tile_wave<T, L, true, DEPTH, false>(r_end, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], colidx, val, x, xs, lo, y, c0, v0);
        else csr_vector_tile_wave<T, L, false, DEPTH, false>(r_end, long_thr, rw0, lane, rp_lds[wave], y_lds[wave], colidx, val, x, xs, lo, y, c0, v0);
        wave_lds_sync(); // y_lds / rp_lds are reused by the next slab
    }
}

// Inspector for the balanced form: column span of rows [split[b], split[b+1]) with len <= long_thr.
__global__ __launch_bounds__(kBlock) void csr_rows_span_kernel(int long_thr, int max_span, const int *__restrict__ split,
                                                               const int *__restrict__ rowptr,
                                                               const int *__restrict__ colidx,
                                                               int *__restrict__ tile_lo, int *__restrict__ tile_span,
                                                               int *__restrict__ staged)
{
    __shared__ int smin[kBlock / kWave], smax[kBlock / kWave];
    const int r0 = split[blockIdx.x], r1 = split[blockIdx.x + 1];
    int mn = INT_MAX, mx = -1;
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    for (int r = r0 + sub; r < r1; r += kBlock / 16) {
        const int p0 = rowptr[r], p1 = rowptr[r + 1];
        if (p1 - p0 > long_thr) continue;
        for (int p = p0 + l; p < p1; p += 16) {
            const int c = colidx[p];
            mn = min(mn, c);
            mx = max(mx, c);
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, kWave));
        mx = max(mx, __shfl_xor(mx, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { smin[threadIdx.x / kWave] = mn; smax[threadIdx.x / kWave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / kWave; ++k) { mn = min(mn, smin[k]); mx = max(mx, smax[k]); }
        const long long span = mx >= mn ? (long long) mx - mn + 1 : 0;
        const bool ok = span > 0 && span <= max_span;
        tile_lo[blockIdx.x] = ok ? mn : 0;
        tile_span[blockIdx.x] = ok ? (int) span : 0;
        if (ok) { atomicAdd(staged, 1); atomicMax(staged + 1, (int) span); }
    }
}

} // namespace spmv
