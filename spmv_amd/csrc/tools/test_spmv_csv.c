/*
 * test_spmv_csv.c -- test_spmv-compatible bench / check driver, plain C.
 *
 * Same command line, same protocol and same 10-column CSV line as the reference's harness
 * (src/samples/test_spmv.c:62-156, 158-209, 215-252):
 *
 *     test_spmv <matrix.mtx> <threads_begin> <threads_end>
 *
 *   - load the matrix (binary cache first, else .mtx and write the cache): test_spmv.c:166-185
 *   - overwrite the values with rand()%8*0.125 under srand(m), x = 1, inline golden: :199-207
 *   - per method (1..6, or the one in $TEST_METHOD -- the reference bakes it in with
 *     -DTEST_METHOD, CMakeLists.txt:58-99) and per thread count (doubling): create (timed),
 *     10 warm-up + 100 individually timed spmv() calls, RMSE vs golden, one CSV line:
 *       matrix,method,vectorized,threads,nnz,rmse,create_ms,mean_ms,GFLOPs_mean,GFLOPs_best
 * Differences, all deliberate:
 *   - x and y live in HBM by default (the rate the GPU library is built for); set
 *     SPMV_HOST_VECTORS=1 to pass host pointers exactly like the reference harness does;
 *   - 2*nnz is computed in double (the reference's int expression overflows above 2^30 nnz,
 *     test_spmv.c:126-127);
 *   - $VALUE_TYPE=float selects fp32 at run time (compile-time VALUE_TYPE in the reference).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <hip/hip_runtime_api.h>

#include "spmv.h"
#include "spmv_hip.h"
#include "spmv_io.h"

static double ms_between(const struct timeval *a, const struct timeval *b)
{
    return (b->tv_sec - a->tv_sec) * 1000.0 + (b->tv_usec - a->tv_usec) / 1000.0;
}

static double get(const void *v, size_t vs, long long i)
{
    return vs == sizeof(double) ? ((const double *) v)[i] : (double) ((const float *) v)[i];
}

static void put(void *v, size_t vs, long long i, double x)
{
    if (vs == sizeof(double)) ((double *) v)[i] = x;
    else ((float *) v)[i] = (float) x;
}

static int run_method(const char *name, int m, int n, int nnz, int *rowptr, int *colidx, void *val, size_t vs,
                      const void *x_host, const double *golden, unsigned long threads, SPMV_METHODS method,
                      VECTORIZED_WAY way, int host_vectors)
{
    struct timeval t1, t2;
    spmv_Handle_t h = NULL;
    void *xd = NULL, *yd = NULL, *yh = malloc(vs * (size_t) (m ? m : 1));
    void *x_perm = malloc(vs * (size_t) ((m > n ? m : n) + 1));
    const void *xa;
    void *ya;
    double create_ms, total = 0, best = 1e9, rmse = 0;
    int it;
    if (!yh) return 1;
    gettimeofday(&t1, NULL);
    spmv_create_handle_all_in_one(&h, m, n, rowptr, colidx, val, threads, method, vs, way, name);
    gettimeofday(&t2, NULL);
    create_ms = ms_between(&t1, &t2);
    if (spmv_hip_last_error()) { fprintf(stderr, "%s\n", spmv_hip_last_error_string()); spmv_destory_handle(h); free(yh); return 1; }
    if (h->index) { /* reordered handle: gather x like test_spmv.c:95-101 (x = 1 here, kept for form) */
        int i;
        for (i = 0; i < m; ++i) put(x_perm, vs, i, get(x_host, vs, h->index[i]));
        x_host = x_perm;
    }
    if (host_vectors) {
        xa = x_host;
        ya = yh;
    } else {
        if (hipMalloc(&xd, vs * (size_t) (n ? n : 1)) != hipSuccess || hipMalloc(&yd, vs * (size_t) (m ? m : 1)) != hipSuccess ||
            hipMemcpy(xd, x_host, vs * (size_t) n, hipMemcpyHostToDevice) != hipSuccess) {
            fprintf(stderr, "device vector allocation failed\n");
            return 1;
        }
        xa = xd;
        ya = yd;
    }
    for (it = 0; it < 10; ++it) spmv(h, m, rowptr, colidx, val, xa, ya);
    for (it = 0; it < 100; ++it) {
        double cur;
        gettimeofday(&t1, NULL);
        spmv(h, m, rowptr, colidx, val, xa, ya); /* synchronous: returns when y is complete */
        gettimeofday(&t2, NULL);
        cur = ms_between(&t1, &t2);
        total += cur;
        if (cur < best) best = cur;
    }
    total /= 100.0;
    if (!host_vectors) (void) hipMemcpy(yh, yd, vs * (size_t) m, hipMemcpyDeviceToHost);
    for (it = 0; it < m; ++it) { /* scatter back through index like test_spmv.c:130-137 */
        const double d = get(yh, vs, it) - golden[h->index ? h->index[it] : it];
        rmse += d / m * d;
    }
    rmse = sqrt(rmse);
    printf("%s,%s,%s,%lu,%d,%f,%f,%f,%f,%f\n", name, Methods_names[method], /* the REQUESTED method, as test_spmv.c:148 prints */
           Vectorized_names[way], threads, nnz, rmse, create_ms, total, 2.0 * nnz / total / 1e6, 2.0 * nnz / best / 1e6);
    spmv_destory_handle(h);
    if (xd) (void) hipFree(xd);
    if (yd) (void) hipFree(yd);
    free(yh);
    free(x_perm);
    return spmv_hip_last_error() != 0;
}

int main(int argc, char **argv)
{
    const char *file, *env;
    int tb, te, m, n, nnz, sym, cached, method_only = -1, host_vectors, i, j, bad = 0;
    int *rowptr, *colidx;
    void *val, *x;
    double *golden;
    size_t vs = sizeof(double);
    unsigned long t;
    if (argc != 4) { fprintf(stderr, "usage: %s <matrix.mtx> <threads_begin> <threads_end>\n", argv[0]); return 1; }
    file = argv[1];
    tb = atoi(argv[2]);
    te = atoi(argv[3]);
    if ((env = getenv("VALUE_TYPE")) && strcmp(env, "float") == 0) vs = sizeof(float);
    if ((env = getenv("TEST_METHOD")) && *env) method_only = atoi(env);
    host_vectors = (env = getenv("SPMV_HOST_VECTORS")) && atoi(env);
    i = spmv_io_load(file, vs, &m, &n, &nnz, &sym, &rowptr, &colidx, &val, &cached);
    if (i != SPMV_IO_OK) { fprintf(stderr, "cannot load %s (error %d)\n", file, i); return 1; }
    x = malloc(vs * (size_t) (n ? n : 1));
    golden = (double *) calloc((size_t) (m ? m : 1), sizeof(double));
    if (!x || !golden) return 1;
    srand((unsigned) m);                                              /* test_spmv.c:198-202 */
    for (i = 0; i < nnz; ++i) put(val, vs, i, rand() % 8 * 0.125);
    for (i = 0; i < n; ++i) put(x, vs, i, 1.0);
    for (i = 0; i < m; ++i)                                           /* test_spmv.c:204-207 */
        for (j = rowptr[i]; j < rowptr[i + 1]; ++j) golden[i] += get(val, vs, j) * get(x, vs, colidx[j]);
    if (tb < 1) tb = 1;
    for (i = method_only >= 0 ? method_only : 1; i < (method_only >= 0 ? method_only + 1 : (int) Method_Total_Size); ++i) {
        const unsigned long b = i == (int) Method_Serial ? 1ul : (unsigned long) tb;
        const unsigned long e = i == (int) Method_Serial ? 1ul : (unsigned long) te;
        for (t = b; t <= e; t <<= 1u)
            bad |= run_method(file, m, n, nnz, rowptr, colidx, val, vs, x, golden, t, (SPMV_METHODS) i, VECTOR_HIP, host_vectors);
    }
    spmv_io_free(rowptr); spmv_io_free(colidx); spmv_io_free(val);
    free(x); free(golden);
    return bad;
}
