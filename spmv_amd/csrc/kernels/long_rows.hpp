// long_rows.hpp -- rows far longer than the schedule's lane group: gathered into a sub-matrix that the CSR5
// executor multiplies.
//
// Used by CSR-vector (Method_Parallel), Balanced and SELL-C-sigma: a row much longer than L*4 would keep one
// lane group looping (CSR-vector) or pad a whole 64-row chunk to its length (SELL).  The reference has the
// same split in SELL -- rows outside the sigma windows go through the plain CSR dot product
// (sell_C_Sigma_spmv.c:289-298) -- and the intent in Balanced2 (rows longer than nnz/nthreads get extra
// workers, parallel_balanced2_spmv.c:72-198).  Here the long rows are compacted in row order (flag -> scan
// -> scatter, no atomics), copied into a contiguous sub-CSR, and a CSR5 plan with row_map = the long-row
// list is built over it (csr5.hpp): perfectly balanced 64 x sigma tiles with grouped x windows, whatever
// share of the non-zeros the long rows hold (config 4's power-law tail: 46 %).  (The first-round form -- one
// workgroup per 4096-entry row segment -- measured 1.53 vs 1.21 ms on config 4 and was removed in round 2.)
#pragma once
#include <climits>
#include "common.hpp"
#include "csr_vector4.hpp"

namespace spmv {

static __global__ __launch_bounds__(kBlock) void long_rows_flag_kernel(int m, int thr, const int *__restrict__ rowptr, int *__restrict__ flags)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride)
        flags[r] = rowptr[r + 1] - rowptr[r] > thr;
}

static __global__ __launch_bounds__(kBlock) void long_rows_len_kernel(int nlong, const int *__restrict__ long_rows,
                                                               const int *__restrict__ rowptr, int *__restrict__ lens)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < nlong) lens[i] = rowptr[long_rows[i] + 1] - rowptr[long_rows[i]];
}

static __global__ __launch_bounds__(kBlock) void narrow_i64_kernel(int n, const long long *__restrict__ in, int *__restrict__ out)
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
        if (sub_col) sub_col[dst + k] = colidx[src + k]; // NULL: values only (spmv_hip_update_values)
        sub_val[dst + k] = val[src + k];
    }
}

} // namespace spmv
