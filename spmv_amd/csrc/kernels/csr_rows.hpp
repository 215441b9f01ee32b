// csr_rows.hpp -- row-parallel CSR kernels.
//
//   csr_scalar_kernel   one lane per row.   GPU schedule of Method_Serial: the plumbing / debug
//                       kernel (reference: serial_spmv.c:9-37 is "for i<m: y[i] = dot(row i)").
//   csr_vector_kernel   L lanes per row.    GPU schedule of Method_Parallel (reference:
//                       parallel_spmv.c:12-18 fans the same row loop over OpenMP threads; the
//                       row's dot product, inner_spmv.h:232-286, becomes a strided partial sum
//                       per lane + a wavefront butterfly).
//
// HBM layout: plain CSR, RowPtr int32[m+1], ColIdx int32[nnz], Val T[nnz]; x T[n]; y T[m].
// Algorithmic bytes per row of length k:  4 (RowPtr) + k(4+s) + s (y) + its share of x.
#pragma once
#include "common.hpp"

namespace spmv {

template <typename T>
__global__ __launch_bounds__(kBlock) void csr_scalar_kernel(int m, const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx,
                                                            const T *__restrict__ val,
                                                            const T *__restrict__ x, T *__restrict__ y)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long row = (long long) blockIdx.x * kBlock + threadIdx.x; row < m; row += stride) {
        const int p0 = rowptr[row], p1 = rowptr[row + 1];
        T sum = 0;
        for (int p = p0; p < p1; ++p) sum = fmadd(val[p], x[colidx[p]], sum);
        y[row] = sum; // empty rows get 0 (Method_Serial writes every row)
    }
}

// L lanes cooperate on one row; a 256-thread workgroup covers 256/L consecutive rows per pass and
// grid-strides over row groups.  Lane l of a row reads elements p0+l, p0+l+L, ... so one wave
// load instruction covers 64/L adjacent rows = one contiguous span of the matrix stream.
template <typename T, int L>
__global__ __launch_bounds__(kBlock) void csr_vector_kernel(int m, const int *__restrict__ rowptr,
                                                            const int *__restrict__ colidx,
                                                            const T *__restrict__ val,
                                                            const T *__restrict__ x, T *__restrict__ y)
{
    constexpr int kRows = kBlock / L;
    const int lane = threadIdx.x % L;
    const int sub = threadIdx.x / L;
    const long long groups = ((long long) m + kRows - 1) / kRows;
    for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long long row = g * kRows + sub;
        T sum = 0;
        if (row < m) {
            const int p0 = rowptr[row], p1 = rowptr[row + 1];
            for (int p = p0 + lane; p < p1; p += L) sum = fmadd(ld_stream(val + p), x[ld_stream(colidx + p)], sum);
        }
        sum = group_sum<L>(sum);
        if (lane == 0 && row < m) y[row] = sum;
    }
}

} // namespace spmv
