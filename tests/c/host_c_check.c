/* host_c_check.c -- sanitizer driver for the host-side C of the library (tests/test_host_c.py
 * builds it with -fsanitize=address,undefined: SURVEY 5 "Race detection / sanitizers").
 *   host_c_check rcm                 RCM on a scrambled band matrix: permutation valid, band recovered
 *   host_c_check mtx <file> [...]    run the Matrix Market reader over files, print rc and sizes
 *   host_c_check bin <file>          write + re-read a cache file
 *   host_c_check plan                the planner (spmv_plan.c): method -> schedule, CSR-vector shape from histograms
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "spmv_io.h"
#include "reorder/rcm.h"
#include "spmv_hip.h"
#include "spmv_internal.h"

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

/* spmv_plan.c reports illegal option values through the API's error channel */
void spmv_set_error(int code, const char *where, const char *what) { (void) code; (void) where; (void) what; }

static void hist_put(spmv_stats *st, int len, long long rows)
{
    int b = 0;
    while (b < SPMV_LEN_BUCKETS - 1 && len > (4 << b)) ++b;
    st->hist_rows[b] += rows;
    st->hist_nnz[b] += rows * len;
    st->m += (int) rows;
    st->nnz += rows * len;
    if (len > st->max_row_len) st->max_row_len = len;
    if (len < st->min_row_len) st->min_row_len = len;
    if (len == 0) st->empty_rows += (int) rows;
}

static int check_plan(void)
{
    spmv_stats st;
    spmv_plan pl;
    SPMV_METHODS act;
    spmv_options op;
    int rc = 0;
    /* equal rows of 32 (config 2): one pass per row with 8 lanes, nothing handed to the long-row path */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 10000000;
    hist_put(&st, 32, 10000000); st.mean_row_len = 32.0;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    printf("equal32: sched=%d L=%d thr=%d act=%d\n", pl.sched, pl.lanes_per_row, pl.long_thr, act);
    if (pl.sched != SPMV_SCHED_CSR_VECTOR || pl.lanes_per_row != 8 || pl.long_thr != 0 || act != Method_Parallel) rc = 1;
    /* config 4's shape: 90 % rows of ~16, 9 % of ~160, 1 % of ~2500 -> small L, the heavy classes to CSR5 */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 10000000;
    hist_put(&st, 8, 3000000); hist_put(&st, 14, 3000000); hist_put(&st, 22, 3000000);
    hist_put(&st, 100, 450000); hist_put(&st, 220, 450000); hist_put(&st, 2500, 100000);
    st.mean_row_len = (double) st.nnz / st.m;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 4, &op, &pl, &act, 1));
    printf("skewed: L=%d thr=%d\n", pl.lanes_per_row, pl.long_thr);
    if (pl.lanes_per_row < 4 || pl.lanes_per_row > 8 || pl.long_thr != 64) rc = 2;
    /* Balanced: a row longer than a worker's share flips the handle to Balanced2 (parallel_balanced2_spmv.c:72-92) */
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Balanced, &st, 4, &op, &pl, &act, 1));
    if (act != Method_Balanced2 || pl.sched != SPMV_SCHED_NNZ_SPLIT) rc = 3;
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; hist_put(&st, 32, 1000); st.mean_row_len = 32.0;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Balanced2, &st, 8, &op, &pl, &act, 1));
    if (act != Method_Balanced || pl.sched != SPMV_SCHED_ROWBLOCK) rc = 4;
    /* no histogram (m = 0): the mean rule, no crash */
    memset(&st, 0, sizeof st);
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    if (pl.lanes_per_row != 1 || pl.long_thr != 0) rc = 5;
    /* very long equal rows: 64 lanes */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; hist_put(&st, 5000, 1000); st.mean_row_len = 5000.0;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    printf("long5000: L=%d thr=%d\n", pl.lanes_per_row, pl.long_thr);
    if (pl.lanes_per_row != 64) rc = 6;
    /* out-of-range method -> serial (common.c:136); SELL / CSR5 map 1:1 */
    (spmv_options_snapshot(&op), spmv_plan_choose((SPMV_METHODS) 99, &st, 8, &op, &pl, &act, 1));
    if (act != Method_Serial || pl.sched != SPMV_SCHED_CSR_SCALAR) rc = 7;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_SellCSigma, &st, 8, &op, &pl, &act, 1));
    if (pl.sched != SPMV_SCHED_SELL || pl.sell_c != 64 || pl.sell_sigma != 1024) rc = 8;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_CSR5SPMV, &st, 4, &op, &pl, &act, 1));
    if (pl.sched != SPMV_SCHED_CSR5 || act != Method_CSR5SPMV) rc = 9;
    /* auto_method: regular -> CSR-vector, skewed -> CSR5; spmv_plan_choose(…, allow_auto = 0) ignores the option */
    if (spmv_hip_set_option("auto_method", 1) != 0) rc = 10;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Serial, &st, 8, &op, &pl, &act, 1));
    if (act != Method_Parallel) rc = 11;
    st.max_row_len = 100000;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Serial, &st, 8, &op, &pl, &act, 1));
    if (act != Method_CSR5SPMV) rc = 12;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Serial, &st, 8, &op, &pl, &act, 0));
    if (act != Method_Serial) rc = 13;
    spmv_hip_set_option("auto_method", 0);
    /* SELL: which rows leave the slabs for the long-row path -- config 4's shape: 90 % rows of 8..24, 9 % of 64..256, 1 % of 1000..4000 */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 10000000;
    hist_put(&st, 8, 1000000); hist_put(&st, 12, 4000000); hist_put(&st, 20, 4000000);
    hist_put(&st, 64, 5000); hist_put(&st, 100, 300000); hist_put(&st, 200, 595000);
    hist_put(&st, 1500, 40000); hist_put(&st, 3000, 60000);
    st.mean_row_len = (double) st.nnz / (double) st.m;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_SellCSigma, &st, 4, &op, &pl, &act, 1));
    printf("sell skewed: long_thr=%d\n", pl.sell_long_thr);
    if (pl.sell_long_thr != 32) rc = 18;
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 10000000;
    hist_put(&st, 32, 10000000); st.mean_row_len = 32.0;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_SellCSigma, &st, 8, &op, &pl, &act, 1));
    if (pl.sell_long_thr != 0) rc = 19; /* equal rows: everything stays in the slabs */
    /* very short heavy-tailed rows (webbase-1M-style: most rows 1..3 entries, a few thousands): the row-granular schedules hand the multiply
     * to the nnz-split executor below the method; a forced lanes_per_row keeps CSR-vector; regular short rows (5-point stencil) stay */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 1000000;
    hist_put(&st, 0, 200000); hist_put(&st, 1, 400000); hist_put(&st, 3, 300000); hist_put(&st, 7, 90000); hist_put(&st, 40, 9000); hist_put(&st, 600, 900); hist_put(&st, 4000, 100);
    st.mean_row_len = (double) st.nnz / (double) st.m;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    printf("webbase-style: Parallel sched=%d act=%d mean=%.2f\n", pl.sched, act, st.mean_row_len);
    if (pl.sched != SPMV_SCHED_NNZ_SPLIT || act != Method_Parallel) rc = 20;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_SellCSigma, &st, 8, &op, &pl, &act, 1));
    if (pl.sched != SPMV_SCHED_NNZ_SPLIT || act != Method_SellCSigma) rc = 21;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_CSR5SPMV, &st, 8, &op, &pl, &act, 1));
    if (pl.sched != SPMV_SCHED_CSR5) rc = 22;
    spmv_hip_set_thread_option("lanes_per_row", 2);
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    if (pl.sched != SPMV_SCHED_CSR_VECTOR) rc = 23;
    spmv_hip_clear_thread_options();
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 16000000;
    hist_put(&st, 5, 16000000); st.mean_row_len = 5.0;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    printf("stencil5: sched=%d L=%d\n", pl.sched, pl.lanes_per_row);
    if (pl.sched != SPMV_SCHED_CSR_VECTOR) rc = 24;
    /* a thread-local override wins over the process-wide value for creates on this thread, and only there */
    memset(&st, 0, sizeof st); st.min_row_len = 1 << 30; st.n = 10000000;
    hist_put(&st, 32, 10000000); st.mean_row_len = 32.0;
    if (spmv_hip_set_thread_option("lanes_per_row", 16) != 0) rc = 15;
    (spmv_options_snapshot(&op), spmv_plan_choose(Method_Parallel, &st, 8, &op, &pl, &act, 1));
    if (pl.lanes_per_row != 16 || spmv_hip_get_option("lanes_per_row") != 16 || spmv_options_get(&op, "lanes_per_row") != 16) rc = 16;
    spmv_hip_clear_thread_options();
    if (spmv_hip_get_option("lanes_per_row") != 0) rc = 17;
    /* illegal option values are refused */
    if (spmv_hip_set_option("lanes_per_row", 3) == 0 || spmv_hip_set_option("nope", 1) == 0 || spmv_hip_set_option("cache_block", 3) == 0) rc = 14;
    printf("plan rc=%d\n", rc);
    return rc;
}

int main(int argc, char **argv)
{
    int i;
    if (argc >= 2 && strcmp(argv[1], "rcm") == 0) return check_rcm();
    if (argc >= 2 && strcmp(argv[1], "plan") == 0) return check_plan();
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
