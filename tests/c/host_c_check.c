/* host_c_check.c -- sanitizer driver for the host-side C of the library (tests/test_host_c.py
 * builds it with -fsanitize=address,undefined: SURVEY 5 "Race detection / sanitizers").
 *   host_c_check rcm                 RCM on a scrambled band matrix: permutation valid, band recovered
 *   host_c_check mtx <file> [...]    run the Matrix Market reader over files, print rc and sizes
 *   host_c_check bin <file>          write + re-read a cache file
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "spmv_io.h"
#include "reorder/rcm.h"

static int check_rcm(void)
{
    const int m = 20000, hb = 5;
    int *sc = malloc(sizeof(int) * m), *inv = malloc(sizeof(int) * m), *perm = malloc(sizeof(int) * m);
    int *rp = malloc(sizeof(int) * (m + 1)), *ci = malloc(sizeof(int) * (size_t) m * (2 * hb + 1));
    double *va = malloc(sizeof(double) * (size_t) m * (2 * hb + 1));
    char *seen = calloc(m, 1);
    int *rp2 = NULL, *ci2 = NULL, nnz = 0, rc = 0, i, r, c;
    void *v2 = NULL;
    long long before, after;
    for (i = 0; i < m; ++i) sc[i] = i;
    srand(1);
    for (i = m - 1; i > 0; --i) { int j = rand() % (i + 1), t = sc[i]; sc[i] = sc[j]; sc[j] = t; }
    for (i = 0; i < m; ++i) inv[sc[i]] = i;
    rp[0] = 0;
    for (r = 0; r < m; ++r) {
        const int o = sc[r];
        for (c = o - hb; c <= o + hb; ++c)
            if (c >= 0 && c < m) { ci[nnz] = inv[c]; va[nnz] = o * 1000.0 + c; ++nnz; }
        rp[r + 1] = nnz;
    }
    before = spmv_csr_bandwidth(m, rp, ci);
    if (spmv_rcm_order(m, rp, ci, perm) || spmv_permute_csr(m, rp, ci, va, 8, perm, &rp2, &ci2, &v2)) rc = 1;
    for (i = 0; i < m && !rc; ++i) { if (seen[perm[i]]) rc = 2; seen[perm[i]] = 1; }
    after = rc ? -1 : spmv_csr_bandwidth(m, rp2, ci2);
    printf("rcm before=%lld after=%lld nnz=%d rc=%d\n", before, after, rc ? -1 : rp2[m], rc);
    if (!rc && (after > 3 * hb || rp2[m] != nnz)) rc = 3;
    /* degenerate inputs */
    { int one[2] = {0, 0}, p1[1] = {7}; if (spmv_rcm_order(1, one, ci, p1) || p1[0] != 0) rc = 4; }
    { int z[1] = {0}; if (spmv_rcm_order(0, z, ci, perm)) rc = 5; }
    free(sc); free(inv); free(perm); free(rp); free(ci); free(va); free(seen); free(rp2); free(ci2); free(v2);
    return rc;
}

int main(int argc, char **argv)
{
    int i;
    if (argc >= 2 && strcmp(argv[1], "rcm") == 0) return check_rcm();
    if (argc >= 3 && strcmp(argv[1], "mtx") == 0) {
        for (i = 2; i < argc; ++i) {
            int m = -1, n = -1, nnz = -1, sym = -1, *rp = NULL, *ci = NULL;
            void *va = NULL;
            const int rc = spmv_io_read_mtx(argv[i], 8, &m, &n, &nnz, &sym, &rp, &ci, &va);
            printf("%s rc=%d m=%d n=%d nnz=%d sym=%d\n", argv[i], rc, m, n, nnz, sym);
            if (rc == 0) {
                long long s = 0;
                int k;
                for (k = 0; k < nnz; ++k) s += ci[k];       /* touch every entry under ASan */
                if (rp[m] != nnz) { printf("inconsistent rowptr\n"); return 9; }
                (void) s;
                spmv_io_free(rp); spmv_io_free(ci); spmv_io_free(va);
            }
        }
        return 0;
    }
    if (argc == 3 && strcmp(argv[1], "bin") == 0) {
        int rp[4] = {0, 2, 2, 3}, ci[3] = {0, 2, 1}, m, n, nnz, *rp2, *ci2, rc;
        float va[3] = {1.5f, -2.f, 3.f};
        void *v2;
        rc = spmv_io_write_bin(argv[2], 3, 3, 3, rp, ci, va, 4);
        if (rc) return 10;
        rc = spmv_io_read_bin(argv[2], 4, &m, &n, &nnz, &rp2, &ci2, &v2);
        if (rc || m != 3 || nnz != 3 || memcmp(rp, rp2, sizeof rp) || memcmp(va, v2, sizeof va)) return 11;
        spmv_io_free(rp2); spmv_io_free(ci2); spmv_io_free(v2);
        rc = spmv_io_read_bin(argv[2], 8, &m, &n, &nnz, &rp2, &ci2, &v2); /* wrong value size: short file */
        printf("bin wrong-size rc=%d\n", rc);
        return rc == 0 ? 12 : 0;
    }
    fprintf(stderr, "usage: host_c_check rcm | mtx <file>... | bin <file>\n");
    return 64;
}
