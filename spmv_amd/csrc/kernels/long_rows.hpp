// long_rows.hpp -- rows far longer than the schedule's lane group: the sub-matrix gather of the default
// long-row path (a CSR5 plan over the long rows, see the section further down) and the first-round segment
// kernel kept as A/B variant 13.
//
// Used by CSR-vector (Method_Parallel) and SELL-C-sigma (Method_SellCSigma): a row much longer than
// L*4 would keep one lane group looping (CSR-vector) or pad a whole 64-row chunk to its length
// (SELL).  The reference has the same split in SELL -- rows outside the sigma windows go through
// the plain CSR dot product (sell_C_Sigma_spmv.c:289-298) -- and the intent in Balanced2 (rows
// longer than nnz/nthreads get extra workers, parallel_balanced2_spmv.c:72-198).  Here:
//
//   inspector   rows with len > thr are appended to long_rows[]; each is cut into segments of
//               kLongSeg = 4096 nnz (seg_start = prefix sum of segment counts, seg_lr[s] = which
//               long row segment s belongs to);
//   executor    one 256-thread workgroup per segment: every lane loads its 16 elements up front
//               (16 B loads), the workgroup reduces the segment's min / max column, stages
//               x[min..max] in LDS when that span fits (gathers then hit LDS instead of up to 64
//               different cache lines per instruction), multiplies and reduces in a fixed order;
//               a one-segment row writes y directly, otherwise partials go to part[s] and
//               long_rows_combine_kernel adds a row's partials in segment order (deterministic).
#pragma once
#include <climits>
#include "common.hpp"
#include "csr_vector4.hpp"

namespace spmv {

constexpr int kLongSeg = 4092; // one 256-thread workgroup covers 4096 elements from the 16 B-aligned start below the segment

__global__ __launch_bounds__(kBlock) void long_rows_collect_kernel(int m, int thr, const int *__restrict__ rowptr,
                                                                   int *__restrict__ long_rows, int *__restrict__ count)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride)
        if (rowptr[r + 1] - rowptr[r] > thr) long_rows[atomicAdd(count, 1)] = (int) r;
}

// --- long rows as a CSR5 sub-matrix (default path) ------------------------------------------------
// The segment kernel below gives a long row to whole workgroups, but a matrix whose long rows hold
// half the non-zeros (config 4's power-law tail) then runs half its traffic through a kernel that
// stages one row's x span at a time.  The default path instead gathers the long rows, in row order,
// into a contiguous sub-CSR and runs the CSR5 inspector/executor over it (csr5.hpp, row_map = the
// long-row list): perfectly balanced 64 x sigma tiles with grouped x windows.
__global__ __launch_bounds__(kBlock) void long_rows_flag_kernel(int m, int thr, const int *__restrict__ rowptr, int *__restrict__ flags)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride)
        flags[r] = rowptr[r + 1] - rowptr[r] > thr;
}

__global__ __launch_bounds__(kBlock) void long_rows_len_kernel(int nlong, const int *__restrict__ long_rows,
                                                               const int *__restrict__ rowptr, int *__restrict__ lens)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nlong) lens[i] = rowptr[long_rows[i] + 1] - rowptr[long_rows[i]];
}

__global__ __launch_bounds__(kBlock) void narrow_i64_kernel(int n, const long long *__restrict__ in, int *__restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = (int) in[i];
}

// one workgroup per long row: copy its column indices and values to the sub-CSR
template <typename T>
__global__ __launch_bounds__(kBlock) void long_rows_gather_kernel(const int *__restrict__ long_rows, const int *__restrict__ rowptr,
                                                                  const int *__restrict__ colidx, const T *__restrict__ val,
                                                                  const int *__restrict__ sub_rowptr, int *__restrict__ sub_col,
                                                                  T *__restrict__ sub_val)
{
    const int r = long_rows[blockIdx.x];
    const int src = rowptr[r], len = rowptr[r + 1] - src, dst = sub_rowptr[blockIdx.x];
    for (int k = threadIdx.x; k < len; k += kBlock) {
        sub_col[dst + k] = colidx[src + k];
        sub_val[dst + k] = val[src + k];
    }
}

__global__ __launch_bounds__(kBlock) void long_rows_segcount_kernel(int nlong, const int *__restrict__ long_rows,
                                                                    const int *__restrict__ rowptr, int *__restrict__ seg_cnt)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nlong) return;
    const int r = long_rows[i];
    seg_cnt[i] = (rowptr[r + 1] - rowptr[r] + kLongSeg - 1) / kLongSeg;
}

__global__ __launch_bounds__(kBlock) void long_rows_segfill_kernel(int nlong, const long long *__restrict__ seg_start,
                                                                   int *__restrict__ seg_lr)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nlong) return;
    for (long long s = seg_start[i]; s < seg_start[i + 1]; ++s) seg_lr[s] = i;
}

// Inspector: column range [lo, lo+span) of every segment, so the executor can start staging x at
// once instead of first reducing min / max over the columns it has just loaded.
__global__ __launch_bounds__(kBlock) void long_rows_segspan_kernel(const int *__restrict__ seg_lr,
                                                                   const long long *__restrict__ seg_start,
                                                                   const int *__restrict__ long_rows,
                                                                   const int *__restrict__ rowptr,
                                                                   const int *__restrict__ colidx,
                                                                   int *__restrict__ seg_lo, int *__restrict__ seg_span,
                                                                   int xcap, int *__restrict__ max_staged_span)
{
    __shared__ int s_mn[kBlock / kWave], s_mx[kBlock / kWave];
    const int s = blockIdx.x;
    const int i = seg_lr[s];
    const int row = long_rows[i];
    const int p0 = rowptr[row] + (int) (s - seg_start[i]) * kLongSeg;
    const int p1 = min(rowptr[row + 1], p0 + kLongSeg);
    int mn = INT_MAX, mx = -1;
    for (int p = p0 + threadIdx.x; p < p1; p += kBlock) { const int c = colidx[p]; mn = min(mn, c); mx = max(mx, c); }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, kWave));
        mx = max(mx, __shfl_xor(mx, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { s_mn[threadIdx.x / kWave] = mn; s_mx[threadIdx.x / kWave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        mx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
        seg_lo[s] = mn;
        seg_span[s] = mx - mn + 1; // p1 > p0 for every segment
        if (mx - mn + 1 <= xcap) atomicMax(max_staged_span, mx - mn + 1); // sizes the executor's LDS request
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void long_rows_kernel(int nsegs, int xcap, const int *__restrict__ seg_lr,
                                                           const long long *__restrict__ seg_start,
                                                           const int *__restrict__ long_rows,
                                                           const int *__restrict__ seg_lo, const int *__restrict__ seg_span,
                                                           const int *__restrict__ rowptr,
                                                           const int *__restrict__ colidx, const T *__restrict__ val,
                                                           const T *__restrict__ x, T *__restrict__ y, T *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char long_x_lds[]; // xcap elements of x
    T *xs = reinterpret_cast<T *>(long_x_lds);
    __shared__ T s_sum[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int s = blockIdx.x;
    const int i = seg_lr[s];
    const long long s0 = seg_start[i];
    const bool single = seg_start[i + 1] - s0 == 1;
    const int row = long_rows[i];
    const int mn = seg_lo[s];
    const int span = seg_span[s];
    const bool staged = span <= xcap;
    const int b = rowptr[row], e = rowptr[row + 1];
    const int p0 = b + (int) (s - s0) * kLongSeg;
    const int p1 = min(e, p0 + kLongSeg);
    const int a0 = (p0 & ~3) + threadIdx.x * 4;
    int c[4][4];
    T v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // all 4 steps in flight; tail reads stay inside the padded allocation
        const int a = a0 + k * kBlock * 4;
        if (a < p1) { ld_stream4(colidx + a, c[k]); ld_stream4(val + a, v[k]); }
    }
    if (staged) { // x window of the segment, issued right behind the matrix stream
        for (int t = threadIdx.x; t < span; t += kBlock) xs[t] = x[mn + t];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = a0 + k * kBlock * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = a + q >= p0 && a + q < p1;
            c[k][q] = ok ? c[k][q] : -1;
        }
    }
    if (staged) __syncthreads();
    T sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c[k][q] >= 0) sum = fmadd(v[k][q], staged ? xs[c[k][q] - mn] : x[c[k][q]], sum);
    }
    sum = group_sum_dpp<kWave>(sum);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const T tot = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        if (single) y[row] = tot;
        else part[s] = tot;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void long_rows_combine_kernel(int nlong, const long long *__restrict__ seg_start,
                                                                   const int *__restrict__ long_rows,
                                                                   const T *__restrict__ part, T *__restrict__ y)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nlong) return;
    const long long s0 = seg_start[i], s1 = seg_start[i + 1];
    if (s1 - s0 <= 1) return;
    T sum = 0;
    for (long long s = s0; s < s1; ++s) sum += part[s];
    y[long_rows[i]] = sum;
}

} // namespace spmv
