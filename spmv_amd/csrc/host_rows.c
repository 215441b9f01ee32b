/*
 * host_rows.c -- the library's ONE piece of host arithmetic: a plain-C CSR row loop for
 * VECTOR_NONE + Method_Serial / Method_Parallel (BASELINE config 1: "Method_Serial fp64, VECTOR_NONE,
 * 100k x 100k banded CSR on CPU -- reference plumbing, no GPU").
 *
 * Replaces, for that configuration only:
 *   spmv_serial_Selected      serial_spmv.c:9-55      for i: Y[i] = dot(row i), every row written
 *   spmv_parallel_Selected    parallel_spmv.c:5-51    the same loop under "omp parallel for"
 *   Dot_Product_Avx2_{d,s}    inner_spmv.h:232-354    one sparse row . dense x
 *   Dot_{d,s}_Products[]      spmv_Defines.h:84-91    the published tables of those dot products
 *
 * It is NOT a fallback: it runs only when the caller asked for VECTOR_NONE AND switched it on
 * explicitly (option "host_rows" / env SPMV_HIP_HOST_ROWS=1, see spmv_api.c).  Every other request
 * runs the HIP schedules or fails loudly (SPMV_HIP_E_NODEVICE); a missing GPU never silently lands here.
 *
 * Written here from the definition y[i] = sum_j val[j] * x[col[j]]; nothing is shared with oracle/ or
 * built from the reference's sources.  Summation order: four independent FMA chains over the entries
 * j = 0, 1, 2, 3 (mod 4), combined as (s0 + s1) + (s2 + s3), then the tail entries in order.  The
 * reference's AVX2 dot uses another order; results agree within rounding and bit for bit on
 * exactly representable data (tests/test_host_rows.py).
 */
#include <math.h>
#include <stddef.h>

#include "spmv_Defines.h"
#include "spmv_internal.h"

double spmv_host_dot_d(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const double *val, const double *x)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s;
    BASIC_INT_TYPE j = 0;
    for (; j + 4 <= len; j += 4) {
        s0 = fma(val[j], x[indx[j]], s0);
        s1 = fma(val[j + 1], x[indx[j + 1]], s1);
        s2 = fma(val[j + 2], x[indx[j + 2]], s2);
        s3 = fma(val[j + 3], x[indx[j + 3]], s3);
    }
    s = (s0 + s1) + (s2 + s3);
    for (; j < len; ++j) s = fma(val[j], x[indx[j]], s);
    return s;
}

float spmv_host_dot_s(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const float *val, const float *x)
{
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s;
    BASIC_INT_TYPE j = 0;
    for (; j + 4 <= len; j += 4) {
        s0 = fmaf(val[j], x[indx[j]], s0);
        s1 = fmaf(val[j + 1], x[indx[j + 1]], s1);
        s2 = fmaf(val[j + 2], x[indx[j + 2]], s2);
        s3 = fmaf(val[j + 3], x[indx[j + 3]], s3);
    }
    s = (s0 + s1) + (s2 + s3);
    for (; j < len; ++j) s = fmaf(val[j], x[indx[j]], s);
    return s;
}

/* the published tables (spmv_Defines.h); the reference fills slot 0 with a scalar loop and slots 1, 2 with
 * its AVX2 / AVX-512 forms -- all compute the same dot product, and so do these */
float (*const Dot_s_Products[])(BASIC_INT_TYPE, const BASIC_INT_TYPE *, const float *, const float *) = {
    spmv_host_dot_s, spmv_host_dot_s, spmv_host_dot_s, spmv_host_dot_s,
};
double (*const Dot_d_Products[])(BASIC_INT_TYPE, const BASIC_INT_TYPE *, const double *, const double *) = {
    spmv_host_dot_d, spmv_host_dot_d, spmv_host_dot_d, spmv_host_dot_d,
};

/* y = A x on the host.  threads <= 1: the serial loop (serial_spmv.c:16-21); otherwise rows are
 * divided statically over an OpenMP team of that size (parallel_spmv.c:12-18).  Every row is written,
 * empty rows get 0. */
void spmv_host_rows(BASIC_INT_TYPE m, const BASIC_INT_TYPE *rowptr, const BASIC_INT_TYPE *colidx, const void *val,
                    size_t value_size, const void *x, void *y, int threads)
{
    BASIC_INT_TYPE i;
    if (value_size == sizeof(double)) {
        const double *v = (const double *) val, *xx = (const double *) x;
        double *yy = (double *) y;
        if (threads <= 1) {
            for (i = 0; i < m; ++i) yy[i] = spmv_host_dot_d(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i], xx);
        } else {
#pragma omp parallel for num_threads(threads) schedule(static)
            for (i = 0; i < m; ++i) yy[i] = spmv_host_dot_d(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i], xx);
        }
    } else {
        const float *v = (const float *) val, *xx = (const float *) x;
        float *yy = (float *) y;
        if (threads <= 1) {
            for (i = 0; i < m; ++i) yy[i] = spmv_host_dot_s(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i], xx);
        } else {
#pragma omp parallel for num_threads(threads) schedule(static)
            for (i = 0; i < m; ++i) yy[i] = spmv_host_dot_s(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i], xx);
        }
    }
}
