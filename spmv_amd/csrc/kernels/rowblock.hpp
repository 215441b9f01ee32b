// rowblock.hpp -- the equal-nnz ROW-block splitter of Method_Balanced.
//
// Reference: init_csrSplitter_balanced2 (parallel_balanced2_spmv.c:41-53): split[t] =
// upper_bound(RowPtr, min(t*stride, nnz)) - 1 with stride = ceil(nnz / T), worker t computes whole rows
// [split[t], split[t+1]) (parallel_balanced_spmv.c:89-98).  Here a worker is a 256-thread workgroup that
// walks its rows with the CSR-vector wave program (csr_vector_rows_kernel, csr_vector_tile.hpp) and
// `stride` is the nnz share of one workgroup (the non-zeros of 256 mean-length rows); T = ceil(nnz / stride).
// The planner only picks this schedule when max_row_len <= stride (otherwise the handle becomes
// Method_Balanced2 = nnz-split, the same switch the reference makes at parallel_balanced2_spmv.c:72-92).
// Fixes over the reference: split[0] = 0 and split[T] = m, so leading and trailing empty rows are
// written.
#pragma once
#include "common.hpp"

namespace spmv {

// split[b], b = 0..nblocks
static __global__ __launch_bounds__(kBlock) void rowblock_split_kernel(int m, int nnz, int nblocks, int stride,
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

} // namespace spmv
