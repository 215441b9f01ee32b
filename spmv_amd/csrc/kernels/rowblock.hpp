// rowblock.hpp -- equal-nnz ROW blocks, products staged in LDS.
//
// GPU schedule of Method_Balanced.  Reference: init_csrSplitter_balanced2
// (parallel_balanced2_spmv.c:41-53): split[t] = upper_bound(RowPtr, min(t*stride, nnz)) - 1 with
// stride = ceil(nnz / T), worker t computes whole rows [split[t], split[t+1])
// (parallel_balanced_spmv.c:89-98).  Here a worker is a 256-thread workgroup and `stride` is the
// nnz share one workgroup stages in LDS (default 2048); T = ceil(nnz / stride).
//
//   phase 1   all 256 lanes stream the block's contiguous nnz range and write val*x[col] into
//             LDS -- coalesced whatever the row lengths are;
//   phase 2   G lanes per row (G = power of two near the block's mean row length) add each
//             row's LDS slice and butterfly-reduce; one store per row.
//
// A block holds < stride + max_row_len products, and the planner only picks this schedule when
// max_row_len <= stride (otherwise the handle becomes Method_Balanced2 = nnz-split, the same
// switch the reference makes at parallel_balanced2_spmv.c:72-92), so 2*stride LDS slots suffice.
// Fixes over the reference: split[0] = 0 and split[T] = m, so leading and trailing empty rows are
// written.
#pragma once
#include "common.hpp"

namespace spmv {

// split[b], b = 0..nblocks
__global__ __launch_bounds__(kBlock) void rowblock_split_kernel(int m, int nnz, int nblocks, int stride,
                                                                const int *__restrict__ rowptr,
                                                                int *__restrict__ split)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b > nblocks) return;
    if (b == 0) { split[0] = 0; return; }
    if (b == nblocks) { split[b] = m; return; }
    long long key = (long long) b * stride;
    if (key > nnz) key = nnz;
    split[b] = upper_bound_dev(rowptr, m + 1, key) - 1;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void rowblock_kernel(const int *__restrict__ split,
                                                          const int *__restrict__ rowptr,
                                                          const int *__restrict__ colidx,
                                                          const T *__restrict__ val,
                                                          const T *__restrict__ x, T *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rowblock_lds[]; // 2*stride*sizeof(T)
    T *prod = reinterpret_cast<T *>(rowblock_lds);
    const int r0 = split[blockIdx.x], r1 = split[blockIdx.x + 1];
    const int nrows = r1 - r0;
    if (nrows <= 0) return;
    const int p0 = rowptr[r0];
    const int p1 = rowptr[r1];
    const int cnt = p1 - p0; // < 2*stride by construction

    // phase 1: 16 B lane loads from the 16 B-aligned start below p0; two steps in flight
    for (int a = (p0 & ~3) + threadIdx.x * 4; a < p1; a += 2 * kBlock * 4) {
        int c0[4], c1[4];
        T v0[4], v1[4];
        const int a1 = a + kBlock * 4;
        const bool second = a1 < p1;
        ld_stream4(colidx + a, c0);
        ld_stream4(val + a, v0);
        if (second) { ld_stream4(colidx + a1, c1); ld_stream4(val + a1, v1); }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (a + k >= p0 && a + k < p1) prod[a + k - p0] = v0[k] * x[c0[k]];
        if (second) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (a1 + k < p1) prod[a1 + k - p0] = v1[k] * x[c1[k]];
        }
    }
    __syncthreads();

    // phase 2: G lanes per row, G = power of two >= mean row length of this block, in [1, 64]
    int G = 1;
    while (G < kWave && G * nrows < cnt) G <<= 1;
    const int rows_per_pass = kBlock / G;
    const int gl = threadIdx.x & (G - 1);
    const int gi = threadIdx.x / G;
    for (int rbase = 0; rbase < nrows; rbase += rows_per_pass) {
        const int r = rbase + gi;
        T sum = 0;
        if (r < nrows) {
            const int s = rowptr[r0 + r] - p0, e = rowptr[r0 + r + 1] - p0;
            for (int i = s + gl; i < e; i += G) sum += prod[i];
        }
        sum = group_sum_rt(sum, G);
        if (gl == 0 && r < nrows) y[r0 + r] = sum;
    }
}

} // namespace spmv
