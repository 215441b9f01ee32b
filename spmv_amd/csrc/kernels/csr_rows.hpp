// csr_rows.hpp -- the row-per-lane CSR kernel.
//
//   csr_scalar_kernel   one lane per row.   GPU schedule of Method_Serial: the plumbing / debug
//                       kernel (reference: serial_spmv.c:9-37 is "for i<m: y[i] = dot(row i)").
//   (Method_Parallel's CSR-vector kernels live in csr_vector_tile.hpp / csr_vector4.hpp.)
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

} // namespace spmv
