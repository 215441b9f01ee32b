// csr_vector4.hpp -- CSR-vector with 16-byte lane loads: the production kernel of Method_Parallel.
//
// Same schedule as csr_vector_kernel (L lanes per row; reference: parallel_spmv.c:12-18 fans
// "y[i] = dot(row i)" over workers, the dot being inner_spmv.h:232-286), re-shaped for the CDNA4
// memory pipe:
//   - every lane reads FOUR consecutive elements per step: one 16 B load of ColIdx and 16 B (fp32)
//     / 2 x 16 B (fp64) of Val, starting at the row start rounded DOWN to a multiple of 4 elements,
//     so all matrix-stream loads are 16 B aligned whatever RowPtr holds; slots in front of the row
//     start or behind the row end are masked (and never touch x).  The library's HBM copy of
//     ColIdx/Val is padded so the rounded-up tail read stays inside the allocation;
//   - the lane sums are combined with DPP row operations (quad_perm / row_half_mirror /
//     row_mirror: register-to-register, no LDS crossbar traffic) instead of ds_bpermute shuffles;
//   - U independent row groups are in flight per wave and step (memory-level parallelism).
// A wave covers (64/L)*U rows per step; with L*4 = mean row length one step is one pass.
#pragma once
#include "common.hpp"

namespace spmv {

} // namespace spmv

namespace spmv {

// Elements a lane may touch beyond nnz when its 16 B step is rounded up: the library pads its HBM
// copies of ColIdx / Val by this many elements.
constexpr int kStreamPad = 4 * kWave + 8;

// Software-pipelined form: a wave owns NB consecutive row groups; RowPtr of all of them is read
// first, the 16 B stream loads of group j+1 are issued before group j is consumed, and y stores
// are never waited for inside the wave -- so a wave keeps two groups of matrix stream in flight for
// most of its life and only the last store's acknowledgement is exposed.  The hot path is straight
// line (masks by select, clamped row index) so the compiler can hoist every load; rows longer than
// 4L fall into the loop at the end of each group, rows longer than long_thr are left to the
// whole-wavefront long-row kernels (kernels/long_rows.hpp).
template <typename T, int L, int NB, bool NTSTORE = false>
__global__ __launch_bounds__(kBlock) void csr_vector_pipe_kernel(int m, int long_thr, const int *__restrict__ rowptr,
                                                                 const int *__restrict__ colidx,
                                                                 const T *__restrict__ val,
                                                                 const T *__restrict__ x, T *__restrict__ y)
{
    constexpr int kRows = kBlock / L;
    const int lane = threadIdx.x % L;
    const int sub = threadIdx.x / L;
    const long long row0 = (long long) blockIdx.x * (kRows * NB) + sub;
    int p0[NB], p1[NB];
    unsigned skip = 0; // bit j: row is past m, or longer than long_thr (kernels/long_rows.hpp computes it)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        long long r = row0 + (long long) j * kRows;
        if (r > m - 1) { r = m - 1; skip |= 1u << j; }
        p0[j] = rowptr[r];
        p1[j] = rowptr[r + 1];
        if (p1[j] - p0[j] > long_thr) { p1[j] = p0[j]; skip |= 1u << j; }
    }
    int c[2][4];
    T v[2][4];
    {
        const int a = (p0[0] & ~3) + lane * 4;
        ld_stream4(colidx + a, c[0]);
        ld_stream4(val + a, v[0]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int cur = j & 1;
        if (j + 1 < NB) {
            const int an = (p0[j + 1] & ~3) + lane * 4;
            ld_stream4(colidx + an, c[cur ^ 1]);
            ld_stream4(val + an, v[cur ^ 1]);
        }
        const int a = (p0[j] & ~3) + lane * 4;
        T xv[4], vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = (a + k >= p0[j]) & (a + k < p1[j]);
            const T xl = x[ok ? c[cur][k] : 0];
            xv[k] = ok ? xl : T(0); // a masked slot contributes 0 * 0, never 0 * x[0] (x[0] may be NaN/Inf)
            vv[k] = ok ? v[cur][k] : T(0);
        }
        T sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) sum = fmadd(vv[k], xv[k], sum);
        if (__any(a + 4 * L < p1[j])) { // some row of this group is longer than one step
            for (int aa = a + 4 * L; __any(aa < p1[j]); aa += 4 * L) {
                if (aa < p1[j]) {
                    int cc[4];
                    T v2[4];
                    ld_stream4(colidx + aa, cc);
                    ld_stream4(val + aa, v2);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (aa + k < p1[j]) sum = fmadd(v2[k], x[cc[k]], sum);
                }
            }
        }
        sum = group_sum_dpp<L>(sum);
        const long long row = row0 + (long long) j * kRows;
        if (lane == 0 && !((skip >> j) & 1u)) {
            if (NTSTORE) __builtin_nontemporal_store(sum, y + row);
            else y[row] = sum;
        }
    }
}

} // namespace spmv
