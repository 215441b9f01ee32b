/*
 * oracle_spmv.c -- CPU restatement of the reference's Method_Serial / Method_Parallel path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's library (oracle/liboracle_spmv.so).  Nothing under
 * spmv_amd/ links, loads or calls it; the product path fails loudly without its HIP library.
 *
 * Parity status: PINNED.  oracle/pin_oracle.py runs every generator case through this file and
 * through the real reference (oracle/_ref/libmv_l2.so, built by oracle/Makefile from the
 * sources under /root/reference) and requires BIT-IDENTICAL y for fp64 and fp32; the cases and
 * the reference's outputs are committed under tests/golden/ and re-checked by
 * tests/test_oracle.py on every run.  (The reference ships no fixtures of its own: SURVEY 4.1.)
 *
 * What is restated, and from where (all paths relative to /root/reference):
 *   oracle_dot_f64      src/src_spmv/inner_spmv.h:232-286   Dot_Product_Avx2_d
 *   oracle_dot_f32      src/src_spmv/inner_spmv.h:288-354   Dot_Product_Avx2_s
 *   oracle_spmv_serial  src/src_spmv/serial_spmv.c:9-55     spmv_serial_Selected
 *   oracle_spmv_omp     src/src_spmv/parallel_spmv.c:5-51   spmv_parallel_Selected
 *   oracle_spmv_exact   src/samples/test_spmv.c:204-207     the harness' inline golden loop
 *
 * The reference kernel's summation ORDER is reproduced so that results agree bit for bit with
 * the reference build (gcc -O3 -mavx2 -mfma, where gcc also contracts the scalar remainder loop
 * into FMAs): lane j%4 (fp64) or j%8 (fp32) accumulates with one fused multiply-add per
 * element, the lanes are combined in the order of the reference's horizontal add, and the
 * remainder elements are folded in with fused multiply-adds in index order.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* inner_spmv.h:232-286 -- 4 FMA lanes, hadd as (l0+l1)+(l2+l3), then len%4 scalar FMAs. */
double oracle_dot_f64(int len, const int *indx, const double *val, const double *x)
{
    double lane[4] = {0.0, 0.0, 0.0, 0.0};
    const int k_iter = len / 4;
    const int k_rem = len % 4;
    double result = 0.0;
    int j, p = 0;
    for (j = 0; j < k_iter; ++j, p += 4) {
        lane[0] = fma(val[p + 0], x[indx[p + 0]], lane[0]);
        lane[1] = fma(val[p + 1], x[indx[p + 1]], lane[1]);
        lane[2] = fma(val[p + 2], x[indx[p + 2]], lane[2]);
        lane[3] = fma(val[p + 3], x[indx[p + 3]], lane[3]);
    }
    if (k_iter) {
        /* _mm256_hadd_pd: (l0+l1, l0+l1, l2+l3, l2+l3); then low half + high half */
        result = (lane[0] + lane[1]) + (lane[2] + lane[3]);
    }
    for (j = 0; j < k_rem; ++j, ++p) {
        result = fma(val[p], x[indx[p]], result);
    }
    return result;
}

/* inner_spmv.h:288-354 -- 8 FMA lanes; (x0+x4)+(x2+x6) and (x1+x5)+(x3+x7), then their sum. */
float oracle_dot_f32(int len, const int *indx, const float *val, const float *x)
{
    float lane[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int k_iter = len / 8;
    const int k_rem = len % 8;
    float result = 0.f;
    int j, l, p = 0;
    for (j = 0; j < k_iter; ++j, p += 8) {
        for (l = 0; l < 8; ++l) {
            lane[l] = fmaf(val[p + l], x[indx[p + l]], lane[l]);
        }
    }
    if (k_iter) {
        const float q0 = lane[0] + lane[4];
        const float q1 = lane[1] + lane[5];
        const float q2 = lane[2] + lane[6];
        const float q3 = lane[3] + lane[7];
        const float d0 = q0 + q2;
        const float d1 = q1 + q3;
        result = d0 + d1;
    }
    for (j = 0; j < k_rem; ++j, ++p) {
        result = fmaf(val[p], x[indx[p]], result);
    }
    return result;
}

/* serial_spmv.c:9-55 -- every row is written, empty rows get 0.  size != 8 means float. */
void oracle_spmv_serial(int m, const int *rowptr, const int *colidx, const void *val,
                        const void *x, void *y, unsigned long size)
{
    int i;
    if (size == sizeof(double)) {
        const double *v = (const double *) val;
        double *yy = (double *) y;
        for (i = 0; i < m; ++i) {
            yy[i] = oracle_dot_f64(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i],
                                   (const double *) x);
        }
    } else {
        const float *v = (const float *) val;
        float *yy = (float *) y;
        for (i = 0; i < m; ++i) {
            yy[i] = oracle_dot_f32(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i],
                                   (const float *) x);
        }
    }
}

/* parallel_spmv.c:5-51 -- the same loop under "omp parallel for" (default static schedule).
 * Used as bench.py's cpu_baseline when oracle/_ref is absent ("kind": "port"). */
void oracle_spmv_omp(int m, const int *rowptr, const int *colidx, const void *val,
                     const void *x, void *y, unsigned long size)
{
    int i;
    if (size == sizeof(double)) {
        const double *v = (const double *) val;
        double *yy = (double *) y;
#pragma omp parallel for
        for (i = 0; i < m; ++i) {
            yy[i] = oracle_dot_f64(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i],
                                   (const double *) x);
        }
    } else {
        const float *v = (const float *) val;
        float *yy = (float *) y;
#pragma omp parallel for
        for (i = 0; i < m; ++i) {
            yy[i] = oracle_dot_f32(rowptr[i + 1] - rowptr[i], colidx + rowptr[i], v + rowptr[i],
                                   (const float *) x);
        }
    }
}

/* test_spmv.c:204-207 -- the harness' golden: plain left-to-right "y += a*x" in the value type.
 * Also the long-double-free "independent" check used by the tolerance tests: accumulate in
 * double whatever the value type, return double. */
void oracle_spmv_exact(int m, const int *rowptr, const int *colidx, const void *val,
                       const void *x, double *y, unsigned long size)
{
    int i, j;
    for (i = 0; i < m; ++i) {
        double acc = 0.0;
        if (size == sizeof(double)) {
            for (j = rowptr[i]; j < rowptr[i + 1]; ++j)
                acc += ((const double *) val)[j] * ((const double *) x)[colidx[j]];
        } else {
            for (j = rowptr[i]; j < rowptr[i + 1]; ++j)
                acc += (double) ((const float *) val)[j] * (double) ((const float *) x)[colidx[j]];
        }
        y[i] = acc;
    }
}

/* Per-row magnitude sum S_i = sum_j |a_ij * x_j| (in double): the scale of the tolerance test
 * |y_i - yref_i| <= tol * S_i  (SURVEY 4.4 item 3). */
void oracle_row_abs_sum(int m, const int *rowptr, const int *colidx, const void *val,
                        const void *x, double *s, unsigned long size)
{
    int i, j;
    for (i = 0; i < m; ++i) {
        double acc = 0.0;
        if (size == sizeof(double)) {
            for (j = rowptr[i]; j < rowptr[i + 1]; ++j)
                acc += fabs(((const double *) val)[j] * ((const double *) x)[colidx[j]]);
        } else {
            for (j = rowptr[i]; j < rowptr[i + 1]; ++j)
                acc += fabs((double) ((const float *) val)[j] * (double) ((const float *) x)[colidx[j]]);
        }
        s[i] = acc;
    }
}

int oracle_abi_version(void) { return 1; }
