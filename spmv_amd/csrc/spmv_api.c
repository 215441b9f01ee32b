/*
 * spmv_api.c -- host side of the drop-in API, plain C11 (north_star: "host code stays C").
 *
 * Replaces the reference's handle + dispatch layer, src/src_spmv/common.c:
 *   gemv_Handle_init / gemv_create_handle / handle_init_common_parameters   common.c:18-29, 63-83
 *   spmv_create_handle_all_in_one                                           common.c:123-190
 *   spmv (indirect call through spmv_functions[])                           common.c:278-304, 85-94
 *   spmv_clear_handle / spmv_destory_handle                                 common.c:31-71
 *   Methods_names / Vectorized_names / funcNames                            common.c:306-339
 *
 * What is different by design: the per-method "get_handle" inspectors and "_Selected" executors
 * of the reference (serial_spmv.c ... csr5_spmv.cpp) are CPU code; here create() asks the
 * planner (spmv_plan.c) for a GPU schedule and hands it to the HIP shim (spmv_shim.hip), and
 * spmv() forwards to the shim.  Without a working HIP device every call reports SPMV_HIP_E_NODEVICE
 * and computes nothing.  The one piece of host arithmetic (host_rows.c: VECTOR_NONE + Method_Serial /
 * Method_Parallel, BASELINE config 1) runs only when option "host_rows" switches it on -- it is a
 * configuration the caller asks for, never a fallback.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv.h"
#include "spmv_hip.h"
#include "spmv_hip_tools.h"
#include "spmv_internal.h"
#include "spmv_shim.h"
#include "reorder/rcm.h"

/* ---------------------------------------------------------------- name tables (ABI data symbols) */
const char *Methods_names[] = {
    "Method_Serial", "Method_Parallel", "Method_Balanced", "Method_Balanced2",
    "Method_BalancedYid", "Method_SellCSigma", "Method_Csr5Spmv",
};
const char *Vectorized_names[] = {
    "VECTOR_NONE", "VECTOR_AVX2", "VECTOR_AVX512", "VECTOR_HIP",
};
#define FN4(m) m "_VECTOR_NONE", m "_VECTOR_AVX2", m "_VECTOR_AVX512", m "_VECTOR_HIP"
const char *funcNames[] = {
    FN4("Method_Serial"), FN4("Method_Parallel"), FN4("Method_Balanced"), FN4("Method_Balanced2"),
    FN4("Method_BalancedYid"), FN4("Method_SellCSigma"), FN4("Method_Csr5Spmv"),
};

/* ---------------------------------------------------------------- error channel (thread-local) */
static _Thread_local int g_err_code = 0;
static _Thread_local char g_err_text[512] = "";

void spmv_set_error(int code, const char *where, const char *what)
{
    g_err_code = code;
    snprintf(g_err_text, sizeof g_err_text, "%s: %s", where, what ? what : "");
    if (!getenv("SPMV_HIP_QUIET")) fprintf(stderr, "[spmv_hip] error %d in %s\n", code, g_err_text);
    /* the API is all-void like the reference's (which checks nothing, not even malloc): a caller that cannot poll
     * spmv_hip_last_error() may ask for the process to stop at the first failure instead of computing on */
    if (getenv("SPMV_HIP_ABORT_ON_ERROR")) abort();
}
int spmv_hip_last_error(void) { return g_err_code; }
const char *spmv_hip_last_error_string(void) { return g_err_text; }
void spmv_hip_clear_error(void) { g_err_code = 0; g_err_text[0] = 0; }
int spmv_hip_device_count(void) { return spmv_shim_device_count(); }
void spmv_hip_trim_pool(void) { spmv_shim_trim_pool(); }

/* ---------------------------------------------------------------- handle life-cycle */
static void handle_reset(spmv_Handle_t h) /* common.c:18-29 */
{
    h->spmvMethod = Method_Serial;
    h->data_size = 0;
    h->nthreads = 0;
    h->vectorizedWay = VECTOR_NONE;
    h->Level_3_opt_used = 0;
    h->RowPtr = NULL;
    h->ColIdx = NULL;
    h->index = NULL;
    h->Matrix_Val = NULL;
    h->Y_temp = NULL;
    h->extraHandle = NULL;
}

static void index_free(spmv_Handle_t h)
{
    if (h->Level_3_opt_used && h->index) free(h->index); /* the permutation is owned by the handle (common.c:44-50) */
    h->index = NULL;
    h->Level_3_opt_used = 0;
}

static void state_free(spmv_Handle_t h)
{
    index_free(h);
    spmv_hip_state *st = (spmv_hip_state *) h->extraHandle;
    if (st) {
        if (st->dev) spmv_shim_matrix_destroy(st->dev);
        if (st->multi) spmv_shim_multi_destroy(st->multi);
        free(st);
        h->extraHandle = NULL;
    }
}

void spmv_clear_handle(spmv_Handle_t h) /* common.c:31-52, 69-71 */
{
    if (!h) return;
    state_free(h);
    handle_reset(h);
}

void spmv_destory_handle(spmv_Handle_t h) /* common.c:54-61 */
{
    if (!h) return;
    state_free(h);
    free(h);
}

/* Option "reorder" (SURVEY 8f f-4): RCM on host copies of the pattern, B = P A P^T uploaded instead
 * of A, the permutation published in handle->index -- the protocol of the reference's OPT_LEVEL 3
 * path (common.c:144-156: permuted copy + index; test_spmv.c:95-101,130-137: the caller gathers
 * XX[i] = X[index[i]] before spmv() and scatters Y[index[i]] = YY[i] after).  Returns 0 when the
 * permuted matrix is resident, non-zero to fall back to the unpermuted upload. */
/* ---------------------------------------------------------------- values changed in place
 * The reference multiplies the arrays passed to THIS call (common.c:286-298); this library multiplies its HBM-resident copy.  Option
 * "check_values" closes the gap: at create a checksum of Matrix_Val is kept, spmv() recomputes it and refreshes the resident copies when
 * it differs.  Mode 1: the full position-weighted sum (shim; host loop or one device reduction).  Mode 2 (default, HOST arrays -- the
 * reference's only mode, where a call already moves x and y over PCIe): the same sum over a SAMPLE -- every 64th 32-bit word (at most
 * 65536 of them) plus the first and last 1024 words -- certain to see an update that touches the whole array (a Newton or time step),
 * blind to most single-entry edits (spmv_hip_update_values or mode 1 are for those). */
static unsigned long long sampled_host_checksum(const void *val, long long words)
{
    const unsigned *w = (const unsigned *) val;
    unsigned long long s = 0;
    long long i;
    const long long edge = words < 2048 ? words : 1024;
    for (i = 0; i < edge; ++i) s += ((unsigned long long) w[i] + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long) i + 1ull);
    if (words >= 2048) {
        /* every 64th word, but never more than 65536 samples: a 2.56 GB value array (config 2) costs 65 k cache lines per call, under a millisecond */
        const long long stride = (words - 2048) / 65536 > 64 ? (words - 2048) / 65536 : 64;
        for (i = words - 1024; i < words; ++i) s += ((unsigned long long) w[i] + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long) i + 1ull);
        for (i = 1024; i < words - 1024; i += stride) s += ((unsigned long long) w[i] + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long) i + 1ull);
    }
    return s;
}

/* take the checksum the handle's options ask for (create, re-inspection, update_values) */
static void watch_values(spmv_Handle_t h, spmv_hip_state *st, const void *Val, long long nnz)
{
    const long mode = st->opts.v[SPMV_OPT_CHECK_VALUES];
    st->val_sum_valid = 0;
    st->val_words = nnz * (long long) ((h->data_size == sizeof(double) ? sizeof(double) : sizeof(float)) / 4);
    if (!Val || st->val_words <= 0 || mode == 0) return;
    if (mode == 1) {
        if (spmv_shim_checksum_words(Val, st->val_words, &st->val_sum) == SPMV_HIP_OK) st->val_sum_valid = 1;
    } else if (!spmv_shim_is_device_ptr(Val)) {
        st->val_sum = sampled_host_checksum(Val, st->val_words);
        st->val_sum_valid = 2;
    }
}

/* host copies of P A P^T and the permutation (all malloc'ed; 0 on success) */
static int reorder_on_host(size_t vs, int m, const int *RowPtr, const int *ColIdx, const void *Val, int **rp2, int **ci2, void **va2, int **perm_out)
{
    int *rp = (int *) malloc(sizeof(int) * ((size_t) m + 1)), *ci = NULL, *perm = NULL;
    void *va = NULL;
    int rc = 1, nnz;
    *rp2 = *ci2 = NULL; *va2 = NULL; *perm_out = NULL;
    if (!rp || spmv_shim_copy_to_host(rp, RowPtr, sizeof(int) * ((size_t) m + 1))) goto out;
    nnz = rp[m];
    if (rp[0] != 0 || nnz < 0) goto out;
    { /* RowPtr must be monotone within [0, nnz] before anything indexes ColIdx/Val with it */
        int i;
        for (i = 0; i < m; ++i)
            if (rp[i] > rp[i + 1] || rp[i + 1] > nnz) goto out;
    }
    ci = (int *) malloc(sizeof(int) * (size_t) (nnz ? nnz : 1));
    va = malloc(vs * (size_t) (nnz ? nnz : 1));
    perm = (int *) malloc(sizeof(int) * (size_t) m);
    if (!ci || !va || !perm) goto out;
    if (spmv_shim_copy_to_host(ci, ColIdx, sizeof(int) * (size_t) nnz) || spmv_shim_copy_to_host(va, Val, vs * (size_t) nnz)) goto out;
    if (spmv_rcm_order(m, rp, ci, perm) || spmv_permute_csr(m, rp, ci, va, vs, perm, rp2, ci2, va2)) goto out;
    *perm_out = perm;
    perm = NULL;
    rc = 0;
out:
    free(rp); free(ci); free(va); free(perm);
    return rc;
}

static int upload_reordered(spmv_Handle_t h, spmv_hip_state *st, int m, int n, const int *RowPtr,
                            const int *ColIdx, const void *Val)
{
    const size_t vs = h->data_size == sizeof(double) ? sizeof(double) : sizeof(float);
    int *rp2 = NULL, *ci2 = NULL, *perm = NULL;
    void *va2 = NULL;
    int rc = 1;
    if (reorder_on_host(vs, m, RowPtr, ColIdx, Val, &rp2, &ci2, &va2, &perm)) goto out;
    if (spmv_shim_matrix_create(&st->dev, m, n, rp2, ci2, va2, vs) != SPMV_HIP_OK) goto out;
    h->index = perm;
    h->Level_3_opt_used = 1;
    perm = NULL;
    rc = 0;
out:
    free(perm); free(rp2); free(ci2); free(va2);
    return rc;
}

/* Option "gpus" > 0: row blocks over the GPUs of this process (shim/multi.hpp; BASELINE config 5, SURVEY 8e, the GPU
 * analogue of numa.c:277-304).  Every shard is planned from ITS row statistics and built on its device. */
/* plan + inspect every shard (and its boundary sub-matrix, if the range exchange split one off) of a multi-GPU state */
static int multi_plan_shards(spmv_Handle_t h, spmv_hip_state *st, SPMV_METHODS *actual)
{
    const int G = spmv_shim_multi_count(st->multi);
    int g, part, rc = SPMV_HIP_OK;
    for (g = 0; g < G && !rc; ++g)
        for (part = 0; part < 2 && !rc; ++part) {
            spmv_dev *dev = part == 0 ? spmv_shim_multi_shard(st->multi, g) : spmv_shim_multi_boundary(st->multi, g);
            spmv_stats stats;
            spmv_plan plan;
            SPMV_METHODS a = st->requested;
            if (!dev) continue;
            rc = spmv_shim_matrix_stats(dev, &stats);
            if (!rc) {
                spmv_plan_choose(st->requested, &stats, (size_t) h->data_size, &st->opts, &plan, &a, 1);
                rc = spmv_shim_build(dev, &plan);
            }
            if (!rc && g == 0 && part == 0) { *actual = a; st->plan = plan; }
        }
    if (rc) {
        spmv_set_error(rc, "create/multi shard", spmv_shim_error_text());
        spmv_shim_multi_destroy(st->multi);
        st->multi = NULL;
    }
    return rc;
}

static int state_build_multi(spmv_Handle_t h, spmv_hip_state *st, int m, int n, const int *RowPtr,
                             const int *ColIdx, const void *Val)
{
    SPMV_METHODS actual = st->requested;
    int rc;
    if (st->multi) { spmv_shim_multi_destroy(st->multi); st->multi = NULL; }
    index_free(h);
    rc = -1;
    if (st->opts.v[SPMV_OPT_REORDER] >= 1 && m == n && m > 1 && RowPtr && ColIdx && Val) {
        /* Option "reorder" on a multi-GPU handle: P A P^T is what gets cut into equal-nnz row blocks -- the reason a partitioner exists in the
         * reference at all (fewer off-block columns: HyperGraphInterface.cpp:60-147 feeding the NUMA row blocks, numa.c:277-304).  The
         * caller-side protocol is the single-GPU one: XX[i] = X[index[i]], Y[index[i]] = YY[i] (test_spmv.c:95-101, 130-137). */
        const size_t vs = h->data_size == sizeof(double) ? sizeof(double) : sizeof(float);
        int *rp2 = NULL, *ci2 = NULL, *perm = NULL;
        void *va2 = NULL;
        spmv_dev *tmp = NULL;
        if (st->opts.v[SPMV_OPT_REORDER] == 1 && (perm = (int *) malloc(sizeof(int) * (size_t) m)) != NULL &&
            spmv_shim_matrix_create(&tmp, m, n, RowPtr, ColIdx, Val, vs) == SPMV_HIP_OK && spmv_shim_reorder_rcm(tmp, perm) == SPMV_HIP_OK) {
            /* reorder = 1: on the device (kernels/rcm.hpp) -- the whole matrix on the current device for a moment, P A P^T handed to the sharding as device arrays */
            const int *drp = NULL, *dci = NULL;
            const void *dva = NULL;
            spmv_shim_matrix_arrays(tmp, &drp, &dci, &dva);
            rc = spmv_shim_multi_create(&st->multi, (int) st->opts.v[SPMV_OPT_GPUS], (int) st->opts.v[SPMV_OPT_X_EXCHANGE], m, n, drp, dci, dva, vs);
            if (rc == SPMV_HIP_OK) { h->index = perm; h->Level_3_opt_used = 1; perm = NULL; }
        } else if ((free(perm), perm = NULL, spmv_hip_clear_error(), 1) && reorder_on_host(vs, m, RowPtr, ColIdx, Val, &rp2, &ci2, &va2, &perm) == 0) {
            rc = spmv_shim_multi_create(&st->multi, (int) st->opts.v[SPMV_OPT_GPUS], (int) st->opts.v[SPMV_OPT_X_EXCHANGE], m, n, rp2, ci2, va2, vs);
            if (rc == SPMV_HIP_OK) { h->index = perm; h->Level_3_opt_used = 1; perm = NULL; }
        } else {
            /* never silently: the caller asked for a permutation and will gather x / scatter y by handle->index -- which stays NULL, i.e. identity */
            spmv_set_error(SPMV_HIP_E_ARG, "create/multi", "option reorder: the matrix could not be reordered (bad RowPtr or out of host memory); created unpermuted, handle->index = NULL");
        }
        if (tmp) spmv_shim_matrix_destroy(tmp);
        free(perm); free(rp2); free(ci2); free(va2);
    }
    if (rc != SPMV_HIP_OK)
        rc = spmv_shim_multi_create(&st->multi, (int) st->opts.v[SPMV_OPT_GPUS], (int) st->opts.v[SPMV_OPT_X_EXCHANGE], m, n, RowPtr, ColIdx, Val,
                                    (size_t) h->data_size);
    if (rc) { spmv_set_error(rc, "create/multi", spmv_shim_error_text()); return rc; }
    rc = multi_plan_shards(h, st, &actual);
    if (rc) return rc;
    st->m = m;
    st->n = n;
    watch_values(h, st, Val, spmv_shim_multi_nnz(st->multi));
    st->from_blocks = 0;
    h->spmvMethod = actual; /* shard 0's: the shards of a skewed matrix may differ (Balanced vs Balanced2) */
    h->RowPtr = (BASIC_INT_TYPE *) RowPtr;
    h->ColIdx = (BASIC_INT_TYPE *) ColIdx;
    h->Matrix_Val = (void *) Val;
    return SPMV_HIP_OK;
}

/* Extension: a multi-GPU handle from row blocks that already exist separately -- block g: rows[g] rows, LOCAL 0-based int32 RowPtr,
 * GLOBAL column indices in [0, n), values; host or device pointers -- the way the reference's NUMA experiment hands every node its
 * block (src/samples/numa.c:277-304, 129-158).  No monolithic CSR exists, so BASELINE config 5 (8 x 1e7 rows x 32 = 2.56e9
 * non-zeros, beyond one int32 RowPtr) is expressible through the C API.  Block g lives on device g; x_exchange as for option "gpus".
 * spmv() on such a handle takes full-length X / Y and IGNORES its CSR arguments (pass NULL). */
void spmv_hip_create_handle_from_blocks(spmv_Handle_t *Handle, int blocks, const BASIC_INT_TYPE *rows, BASIC_INT_TYPE n,
                                        BASIC_INT_TYPE *const *RowPtr, BASIC_INT_TYPE *const *ColIdx, void *const *Matrix_Val,
                                        SPMV_METHODS Function, BASIC_SIZE_TYPE size)
{
    spmv_Handle_t h;
    spmv_hip_state *st;
    SPMV_METHODS actual;
    int rc;
    if (!Handle) { spmv_set_error(SPMV_HIP_E_ARG, "create_from_blocks", "Handle is NULL"); return; }
    h = (spmv_Handle_t) malloc(sizeof(spmv_Handle));
    *Handle = h;
    if (!h) { spmv_set_error(SPMV_HIP_E_ALLOC, "create_from_blocks", "malloc(handle)"); return; }
    handle_reset(h);
    if ((int) Function < (int) Method_Serial || (int) Function >= (int) Method_Total_Size) Function = Method_Serial;
    h->nthreads = 1;
    h->vectorizedWay = VECTOR_HIP;
    h->data_size = size;
    h->spmvMethod = Function;
    st = (spmv_hip_state *) calloc(1, sizeof *st);
    if (!st) { spmv_set_error(SPMV_HIP_E_ALLOC, "create_from_blocks", "malloc(state)"); return; }
    st->requested = actual = Function;
    spmv_options_snapshot(&st->opts);
    rc = spmv_shim_multi_create_blocks(&st->multi, blocks, (int) st->opts.v[SPMV_OPT_X_EXCHANGE], rows, n, (const int *const *) RowPtr,
                                       (const int *const *) ColIdx, (const void *const *) Matrix_Val, (size_t) size);
    if (rc) { spmv_set_error(rc, "create_from_blocks", spmv_shim_error_text()); free(st); return; }
    if (multi_plan_shards(h, st, &actual) != SPMV_HIP_OK) { free(st); return; }
    st->m = spmv_shim_multi_rows(st->multi);
    st->n = n;
    st->from_blocks = 1;
    h->spmvMethod = actual;
    h->extraHandle = st;
}

/* A matrix with locality in PART of its entries (every tenth row random; web graphs: 90 % of a row near the diagonal, 10 % on hub
 * columns) stages no x window -- one stray entry per tile is enough -- and the blocked executor pays for the local entries too.  When
 * the shim's sample says a real part of the entries, but not all, lies near its tile's centre column, the matrix is split once into
 * A_near + A_far (csrc/kernels/split.hpp), each half planned and inspected like any matrix -- near: the method's tile schedule, every
 * tile staged by construction, never the blocked executor; far: always the blocked executor, accumulating into y -- and the pair is
 * timed against the schedule as built; the faster stays (spmv_hip_info.split_ms, far_nnz). */
static void try_split(spmv_Handle_t h, spmv_hip_state *st, SPMV_METHODS actual)
{
    spmv_dev *halves[2] = {NULL, NULL};
    double t_built, t_split = -1.0;
    int k, ok = 1;
    if (st->opts.v[SPMV_OPT_SPLIT] != 1 || !spmv_shim_split_candidate(st->dev)) return;
    t_built = spmv_shim_time_self(st->dev, 5);
    if (t_built <= 0.0 || spmv_shim_split(st->dev, &halves[0], &halves[1]) != SPMV_HIP_OK) return;
    for (k = 0; k < 2 && ok; ++k) {
        spmv_stats stats;
        spmv_plan plan;
        spmv_options o = st->opts;
        SPMV_METHODS a = actual;
        o.v[SPMV_OPT_CACHE_BLOCK] = k == 0 ? 0 : 2;
        /* the near half of a CSR-vector request: rows that lost entries to the far half are no longer regular (every tenth row empty:
         * 57 % of CSR-vector's 8-row steps run masked, 0.89 vs 0.55 ms under CSR5) -- let the row statistics choose, as auto_method = 1 does */
        if (k == 0 && actual == Method_Parallel && o.v[SPMV_OPT_AUTO_METHOD] < 1) o.v[SPMV_OPT_AUTO_METHOD] = 1;
        ok = spmv_shim_matrix_stats(halves[k], &stats) == SPMV_HIP_OK;
        if (ok) {
            spmv_plan_choose(actual == Method_Serial ? Method_Parallel : actual, &stats, (size_t) h->data_size, &o, &plan, &a, k == 0 && actual == Method_Parallel);
            ok = spmv_shim_build(halves[k], &plan) == SPMV_HIP_OK;
        }
    }
    if (ok && getenv("SPMV_HIP_SPLIT_DEBUG")) {
        spmv_hip_info a, b;
        double tn = spmv_shim_time_self(halves[0], 5), tf = spmv_shim_time_self(halves[1], 5);
        (void) spmv_shim_info(halves[0], &a);
        (void) spmv_shim_info(halves[1], &b);
        fprintf(stderr, "[spmv_hip] split: as built %.4f ms; near %lld nnz %s %.4f ms; far %lld nnz %s %.4f ms (inspect %.1f / %.1f ms)\n", t_built, a.nnz, a.kernel_name,
                tn, b.nnz, b.kernel_name, tf, a.inspect_ms, b.inspect_ms);
    }
    if (ok && spmv_shim_attach_split(st->dev, halves[0], halves[1], 0) == SPMV_HIP_OK) {
        t_split = spmv_shim_time_self(st->dev, 5);
        if (t_split > 0.0 && t_split < 0.9 * t_built) (void) spmv_shim_attach_split(st->dev, halves[0], halves[1], 1); /* keep: drop the unsplit schedule */
        else (void) spmv_shim_attach_split(st->dev, NULL, NULL, 0);                                                    /* destroys the halves */
    } else {
        if (halves[0]) spmv_shim_matrix_destroy(halves[0]);
        if (halves[1]) spmv_shim_matrix_destroy(halves[1]);
    }
    spmv_shim_note_split_ms(st->dev, t_built, t_split);
    spmv_hip_clear_error();
}

/* Upload + plan + inspect.  Used by create and by spmv() when it is handed another matrix. */
static int state_build(spmv_Handle_t h, spmv_hip_state *st, int m, int n, const int *RowPtr,
                       const int *ColIdx, const void *Val)
{
    spmv_stats stats;
    SPMV_METHODS actual = st->requested;
    int rc;
    if (st->opts.v[SPMV_OPT_GPUS] > 0) return state_build_multi(h, st, m, n, RowPtr, ColIdx, Val);
    if (st->dev) { spmv_shim_matrix_destroy(st->dev); st->dev = NULL; }
    if (m < 0 || n < 0 || (m > 0 && !RowPtr)) {
        spmv_set_error(SPMV_HIP_E_ARG, "create", "negative size or NULL RowPtr");
        return SPMV_HIP_E_ARG;
    }
    index_free(h);
    rc = -1;
    if (st->opts.v[SPMV_OPT_REORDER] == 2 && m == n && m > 1 && RowPtr && ColIdx && Val)
        rc = upload_reordered(h, st, m, n, RowPtr, ColIdx, Val); /* the host BFS of round 1 (reorder/rcm.c), kept for comparison; 0 = uploaded the permuted matrix */
    if (rc != 0) rc = spmv_shim_matrix_create(&st->dev, m, n, RowPtr, ColIdx, Val, (size_t) h->data_size);
    if (rc) { spmv_set_error(rc, "create/upload", spmv_shim_error_text()); return rc; }
    if (st->opts.v[SPMV_OPT_REORDER] == 1 && m == n && m > 1) {
        /* reverse Cuthill-McKee ON THE DEVICE over the matrix just uploaded (kernels/rcm.hpp): P A P^T replaces it, handle->index = the
         * permutation (test_spmv.c:95-101, 130-137: the caller gathers x and scatters y).  A failure leaves the unpermuted matrix resident
         * and index NULL -- and says so. */
        int *perm = (int *) malloc(sizeof(int) * (size_t) m);
        if (perm && spmv_shim_reorder_rcm(st->dev, perm) == SPMV_HIP_OK) {
            h->index = perm;
            h->Level_3_opt_used = 1;
        } else {
            free(perm);
            spmv_set_error(SPMV_HIP_E_RUNTIME, "create/reorder", perm ? spmv_shim_error_text() : "malloc(perm)");
        }
    }
    rc = spmv_shim_matrix_stats(st->dev, &stats);
    if (rc) { spmv_set_error(rc, "create/stats", spmv_shim_error_text()); return rc; }
    spmv_plan_choose(st->requested, &stats, (size_t) h->data_size, &st->opts, &st->plan, &actual, 1);
    rc = spmv_shim_build(st->dev, &st->plan);
    if (rc) {
        spmv_set_error(rc, "create/inspect", spmv_shim_error_text());
        spmv_shim_matrix_destroy(st->dev);
        st->dev = NULL;
        return rc;
    }
    /* A schedule that could not stage a single x window on a matrix whose x is far larger than an L2 is
     * switched to the row-block x column-slab executor INSIDE spmv_shim_build (option "cache_block", default
     * automatic) -- whatever the method, so Method_Parallel / Method_CSR5SPMV requests on matrices without
     * column locality no longer run the gather-bound tile kernels. */
    /* automatic choice, measured (auto_method = 2): the rules above pick from row statistics; which schedule is
     * fastest also depends on the columns and, by a few percent, on the device.  For matrices
     * large enough to be worth it, every candidate schedule is built and timed on scratch vectors and the
     * fastest is kept (the rule-based choice stays on a tie within 2 %). */
    if (st->opts.v[SPMV_OPT_AUTO_METHOD] == 2 && stats.nnz >= (1ll << 20)) {
        static const SPMV_METHODS cand[] = {Method_Parallel, Method_CSR5SPMV, Method_SellCSigma, Method_Balanced_Yid, Method_Balanced};
        spmv_plan best_plan = st->plan;
        SPMV_METHODS best_method = actual;
        double best_ms = spmv_shim_time_self(st->dev, 5);
        unsigned k, j, nseen = 1;
        spmv_plan seen[1 + sizeof cand / sizeof cand[0]]; /* schedules already built and timed: several methods may plan the same one (short heavy-tailed rows) */
        int current_is_best = best_ms >= 0.0;
        seen[0] = st->plan;
        for (k = 0; k < sizeof cand / sizeof cand[0] && best_ms >= 0.0; ++k) {
            spmv_plan p;
            SPMV_METHODS a = cand[k];
            double ms;
            int dup = 0;
            spmv_plan_choose(cand[k], &stats, (size_t) h->data_size, &st->opts, &p, &a, 0);
            for (j = 0; j < nseen; ++j) /* the same schedule with the same shape is the same multiply whatever the method is called: timing noise must not choose */
                if (seen[j].sched == p.sched && seen[j].lanes_per_row == p.lanes_per_row && seen[j].long_thr == p.long_thr && seen[j].sell_sigma == p.sell_sigma &&
                    seen[j].sell_long_thr == p.sell_long_thr && seen[j].csr5_sigma == p.csr5_sigma && seen[j].rowblock_nnz == p.rowblock_nnz) dup = 1;
            if (dup) continue;
            seen[nseen++] = p;
            if (spmv_shim_build(st->dev, &p) != SPMV_HIP_OK) { current_is_best = 0; continue; }
            current_is_best = 0;
            ms = spmv_shim_time_self(st->dev, 5);
            if (ms >= 0.0 && ms < 0.98 * best_ms) { best_ms = ms; best_plan = p; best_method = a; current_is_best = 1; }
        }
        if (!current_is_best) {
            rc = spmv_shim_build(st->dev, &best_plan);
            if (rc) {
                spmv_set_error(rc, "create/inspect", spmv_shim_error_text());
                spmv_shim_matrix_destroy(st->dev);
                st->dev = NULL;
                return rc;
            }
        }
        st->plan = best_plan;
        actual = best_method;
    }
    try_split(h, st, actual);
    if (st->opts.v[SPMV_OPT_KEEP_COLUMNS] == 0) (void) spmv_shim_release_columns(st->dev); /* the last build of this create is done */
    if (st->stream_set) spmv_shim_set_stream(st->dev, st->stream);
    spmv_shim_set_async(st->dev, st->async);
    st->m = m;
    st->n = n;
    watch_values(h, st, Val, stats.nnz);
    h->spmvMethod = actual;
    h->RowPtr = (BASIC_INT_TYPE *) RowPtr;
    h->ColIdx = (BASIC_INT_TYPE *) ColIdx;
    h->Matrix_Val = (void *) Val;
    return SPMV_HIP_OK;
}

void spmv_create_handle_all_in_one(spmv_Handle_t *Handle, BASIC_INT_TYPE m, BASIC_INT_TYPE n,
                                   BASIC_INT_TYPE *RowPtr, BASIC_INT_TYPE *ColIdx, void *Matrix_Val,
                                   BASIC_SIZE_TYPE nthreads, SPMV_METHODS Function, BASIC_SIZE_TYPE size,
                                   VECTORIZED_WAY vectorizedWay, const char *MtxToken)
{
    spmv_Handle_t h;
    spmv_hip_state *st;
    (void) MtxToken; /* the reference uses it for METIS cache file names only (common.c:152-154) */
    if (!Handle) { spmv_set_error(SPMV_HIP_E_ARG, "create", "Handle is NULL"); return; }
    h = (spmv_Handle_t) malloc(sizeof(spmv_Handle)); /* common.c:63-67 */
    *Handle = h;
    if (!h) { spmv_set_error(SPMV_HIP_E_ALLOC, "create", "malloc(handle)"); return; }
    handle_reset(h);
    if ((int) Function < (int) Method_Serial || (int) Function >= (int) Method_Total_Size)
        Function = Method_Serial; /* common.c:136 */
    /* common.c:74-83 */
    h->nthreads = nthreads;
    h->vectorizedWay = vectorizedWay;
    h->data_size = size;
    h->spmvMethod = Function;

    st = (spmv_hip_state *) calloc(1, sizeof *st);
    if (!st) { spmv_set_error(SPMV_HIP_E_ALLOC, "create", "malloc(state)"); return; }
    st->requested = Function;
    spmv_options_snapshot(&st->opts);
    h->extraHandle = st;
    /* BASELINE config 1 ("reference plumbing, no GPU"): VECTOR_NONE + Method_Serial / Method_Parallel run
     * the plain-C row loop of host_rows.c on the caller's arrays, which are BORROWED like the reference
     * does (common.c:157-159) -- but only when option "host_rows" asks for it. */
    if (st->opts.v[SPMV_OPT_HOST_ROWS] == 1 && vectorizedWay == VECTOR_NONE &&
        (Function == Method_Serial || Function == Method_Parallel)) {
        if (spmv_shim_is_device_ptr(RowPtr) || spmv_shim_is_device_ptr(ColIdx) || spmv_shim_is_device_ptr(Matrix_Val)) {
            spmv_set_error(SPMV_HIP_E_ARG, "create(host_rows)", "the host row loop needs HOST arrays; device pointers were passed");
            free(st);
            h->extraHandle = NULL;
            return;
        }
        if (m < 0 || n < 0 || (m > 0 && (!RowPtr || (RowPtr[m] > 0 && (!ColIdx || !Matrix_Val))))) {
            spmv_set_error(SPMV_HIP_E_ARG, "create(host_rows)", "negative size or NULL CSR array");
            free(st);
            h->extraHandle = NULL;
            return;
        }
        st->host_rows = 1;
        st->m = m;
        st->n = n;
        h->RowPtr = RowPtr;
        h->ColIdx = ColIdx;
        h->Matrix_Val = Matrix_Val;
        return;
    }
    if (state_build(h, st, m, n, RowPtr, ColIdx, Matrix_Val) != SPMV_HIP_OK) {
        /* keep a valid handle whose spmv() is a reported no-op */
        free(st);
        h->extraHandle = NULL;
    }
}

void spmv(const spmv_Handle_t handle, BASIC_INT_TYPE m, const BASIC_INT_TYPE *RowPtr,
          const BASIC_INT_TYPE *ColIdx, const void *Matrix_Val, const void *X, void *Y)
{
    spmv_hip_state *st;
    int rc;
    if (handle == NULL) return; /* common.c:285 */
    st = (spmv_hip_state *) handle->extraHandle;
    if (st && st->host_rows) { /* the arguments of THIS call are what is multiplied (common.c:286-298) */
        if (m > 0 && (!RowPtr || !Y || (RowPtr[m] > 0 && (!ColIdx || !Matrix_Val || !X)))) {
            spmv_set_error(SPMV_HIP_E_ARG, "spmv(host_rows)", "NULL argument");
            return;
        }
        spmv_host_rows(m, RowPtr, ColIdx, Matrix_Val, (size_t) handle->data_size, X, Y,
                       handle->spmvMethod == Method_Parallel ? (int) (handle->nthreads > 0 ? handle->nthreads : 1) : 1);
        return;
    }
    if (!st || (!st->dev && !st->multi)) {
        spmv_set_error(SPMV_HIP_E_NOSTATE, "spmv", "handle has no device state (create failed?)");
        return;
    }
    /* The reference re-reads the CSR arguments on every call (common.c:286-298).  Same pointers
     * and m as at create -> the HBM-resident matrix; anything else -> re-inspect that matrix. */
    if (st->from_blocks) {
        /* created from row blocks: there is no monolithic CSR the arguments could name; they are ignored */
    } else if (m != st->m || RowPtr != handle->RowPtr || ColIdx != handle->ColIdx ||
               Matrix_Val != handle->Matrix_Val) {
        if (!st->warned_rebuild && !getenv("SPMV_HIP_QUIET")) {
            fprintf(stderr, "[spmv_hip] spmv(): CSR arguments differ from create(); re-inspecting "
                            "(slow path, DESIGN.md \"CSR arguments\")\n");
            st->warned_rebuild = 1;
        }
        if (state_build(handle, st, m, st->n, RowPtr, ColIdx, Matrix_Val) != SPMV_HIP_OK) return;
    } else if (st->val_sum_valid) {
        /* option "check_values": the reference re-reads Matrix_Val on every call, so a caller may change the
         * values in place between calls (Newton steps, time stepping).  Detect that by checksum and refresh
         * the resident copies (no re-inspection: the pattern is the same). */
        unsigned long long sum = 0;
        int have = 1;
        if (st->val_sum_valid == 2) sum = sampled_host_checksum(Matrix_Val, st->val_words);
        else have = spmv_shim_checksum_words(Matrix_Val, st->val_words, &sum) == SPMV_HIP_OK;
        if (have && sum != st->val_sum) {
            if (st->multi && !handle->Level_3_opt_used) { /* every shard refreshes its slice of the values in place */
                rc = spmv_shim_multi_update_values(st->multi, Matrix_Val);
                if (rc) { spmv_set_error(rc, "spmv/refresh values", spmv_shim_error_text()); return; }
                st->val_sum = sum;
            } else if (handle->Level_3_opt_used) {
                /* option "reorder": the resident matrix is P A P^T, whose value order is not the caller's -- the values
                 * cannot be refreshed in place; upload, reorder and inspect the caller's matrix again (state_build
                 * takes a new checksum) */
                if (state_build(handle, st, m, st->n, RowPtr, ColIdx, Matrix_Val) != SPMV_HIP_OK) return;
            } else {
                rc = spmv_shim_update_values(st->dev, Matrix_Val);
                if (rc) { spmv_set_error(rc, "spmv/refresh values", spmv_shim_error_text()); return; }
                st->val_sum = sum;
            }
        }
    }
    rc = st->multi ? spmv_shim_multi_run(st->multi, X, Y) : spmv_shim_run(st->dev, X, Y);
    if (rc) spmv_set_error(rc, "spmv", spmv_shim_error_text());
}

/* ---------------------------------------------------------------- extensions */
static spmv_hip_state *state_of(spmv_Handle_t h, const char *where)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    if (st && st->multi && !st->dev) {
        spmv_set_error(SPMV_HIP_E_ARG, where, "not available on a multi-GPU handle (option \"gpus\"): it owns one stream per device");
        return NULL;
    }
    if (!st || !st->dev) {
        spmv_set_error(SPMV_HIP_E_NOSTATE, where, "handle has no device state");
        return NULL;
    }
    return st;
}

static spmv_multi *multi_of(spmv_Handle_t h, const char *where)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    if (!st || !st->multi) {
        spmv_set_error(SPMV_HIP_E_NOSTATE, where, "not a multi-GPU handle (create it with option \"gpus\" > 0)");
        return NULL;
    }
    return st->multi;
}

/* Number of GPUs the handle's row blocks live on (0: not a multi-GPU handle). */
int spmv_hip_multi_gpus(spmv_Handle_t h)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    return st && st->multi ? spmv_shim_multi_count(st->multi) : 0;
}

int spmv_hip_multi_uses_rccl(spmv_Handle_t h)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    return st && st->multi ? spmv_shim_multi_uses_rccl(st->multi) : 0;
}

int spmv_hip_multi_slices(spmv_Handle_t h, int gpu, void **x_slice, long long *x_first, long long *x_count,
                          void **y_block, long long *y_first, long long *y_count, int *device)
{
    spmv_multi *mt = multi_of(h, "multi_slices");
    int rc;
    if (!mt) return SPMV_HIP_E_NOSTATE;
    rc = spmv_shim_multi_slices(mt, gpu, x_slice, x_first, x_count, y_block, y_first, y_count, device);
    if (rc) spmv_set_error(rc, "multi_slices", spmv_shim_error_text());
    return rc;
}

int spmv_hip_multi_step(spmv_Handle_t h)
{
    spmv_multi *mt = multi_of(h, "multi_step");
    int rc;
    if (!mt) return SPMV_HIP_E_NOSTATE;
    rc = spmv_shim_multi_step(mt);
    if (rc) spmv_set_error(rc, "multi_step", spmv_shim_error_text());
    return rc;
}

int spmv_hip_multi_step_async(spmv_Handle_t h)
{
    spmv_multi *mt = multi_of(h, "multi_step_async");
    int rc;
    if (!mt) return SPMV_HIP_E_NOSTATE;
    rc = spmv_shim_multi_step_async(mt);
    if (rc) spmv_set_error(rc, "multi_step_async", spmv_shim_error_text());
    return rc;
}

int spmv_hip_multi_synchronize(spmv_Handle_t h)
{
    spmv_multi *mt = multi_of(h, "multi_synchronize");
    int rc;
    if (!mt) return SPMV_HIP_E_NOSTATE;
    rc = spmv_shim_multi_sync(mt);
    if (rc) spmv_set_error(rc, "multi_synchronize", spmv_shim_error_text());
    return rc;
}

int spmv_hip_set_stream(spmv_Handle_t h, void *stream)
{
    spmv_hip_state *st = state_of(h, "set_stream");
    if (!st) return SPMV_HIP_E_NOSTATE;
    st->stream = stream;
    st->stream_set = 1;
    return spmv_shim_set_stream(st->dev, stream);
}

int spmv_hip_set_async(spmv_Handle_t h, int async)
{
    spmv_hip_state *st = state_of(h, "set_async");
    if (!st) return SPMV_HIP_E_NOSTATE;
    st->async = async != 0;
    return spmv_shim_set_async(st->dev, st->async);
}

int spmv_hip_synchronize(spmv_Handle_t h)
{
    spmv_hip_state *st = state_of(h, "synchronize");
    int rc;
    if (!st) return SPMV_HIP_E_NOSTATE;
    rc = spmv_shim_sync(st->dev);
    if (rc) spmv_set_error(rc, "synchronize", spmv_shim_error_text());
    return rc;
}

int spmv_hip_get_info(spmv_Handle_t h, spmv_hip_info *out)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    if (st && st->host_rows && out) { /* no device state: describe the host loop */
        const long long s = h->data_size == sizeof(double) ? 8 : 4;
        memset(out, 0, sizeof *out);
        out->device = -1;
        out->schedule = SPMV_SCHED_HOST_ROWS;
        out->m = st->m;
        out->n = st->n;
        out->nnz = out->stored_nnz = st->m > 0 ? h->RowPtr[st->m] : 0;
        out->mean_row_len = st->m > 0 ? (double) out->nnz / st->m : 0.0;
        out->alg_bytes = 4ll * ((long long) st->m + 1) + out->nnz * (4 + s) + s * st->n + s * st->m;
        out->stream_bytes = out->alg_bytes;
        out->schedule_name = "host-rows";
        out->kernel_name = "spmv_host_rows";
        return SPMV_HIP_OK;
    }
    if (st && st->multi && out) { /* shard 0 names the schedule; sizes, byte counts and stored entries are summed over the
                                   * shards (inspect_ms: the slowest shard), so that rates derived from the struct are the whole matrix's */
        int rc = spmv_shim_info(spmv_shim_multi_shard(st->multi, 0), out), g;
        const int G = spmv_shim_multi_count(st->multi);
        for (g = 1; g < G && rc == SPMV_HIP_OK; ++g) {
            spmv_hip_info o;
            rc = spmv_shim_info(spmv_shim_multi_shard(st->multi, g), &o);
            if (rc != SPMV_HIP_OK) break;
            out->stream_bytes += o.stream_bytes;
            out->x_bytes += o.x_bytes;
            out->stored_nnz += o.stored_nnz;
            out->device_bytes += o.device_bytes;
            out->empty_rows += o.empty_rows;
            out->x_groups += o.x_groups;
            out->x_groups_staged += o.x_groups_staged;
            out->run_nnz += o.run_nnz;
            if (o.inspect_ms > out->inspect_ms) out->inspect_ms = o.inspect_ms;
            if (o.max_row_len > out->max_row_len) out->max_row_len = o.max_row_len;
            if (o.min_row_len < out->min_row_len) out->min_row_len = o.min_row_len;
        }
        if (rc == SPMV_HIP_OK) {
            const long long s = h->data_size == sizeof(double) ? 8 : 4;
            out->m = st->m; out->n = st->n; out->nnz = spmv_shim_multi_nnz(st->multi);
            out->mean_row_len = st->m > 0 ? (double) out->nnz / st->m : 0.0;
            out->alg_bytes = 4ll * ((long long) st->m + 1) + out->nnz * (4 + s) + s * st->n + s * st->m;
        }
        return rc;
    }
    st = state_of(h, "get_info");
    if (!st || !out) return SPMV_HIP_E_NOSTATE;
    return spmv_shim_info(st->dev, out);
}

/* New values behind the pattern the handle was created with: Val (host or device, nnz entries in CSR order)
 * is copied to HBM and re-permuted into the schedule's private layouts (SELL slabs, CSR5 tiles, long-row
 * sub-matrix, row-block x column-slab streams) by device kernels; RowPtr / ColIdx, descriptors, x windows
 * and the autotune result are kept.  Cost: one pass over the values (config 2: ~1 ms against ~30 ms of
 * clear + create).  handle->Matrix_Val is set to Val, so later spmv() calls may pass either pointer. */
int spmv_hip_update_values(spmv_Handle_t h, const void *Val)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    int rc;
    if (st && st->host_rows) { h->Matrix_Val = (void *) Val; return SPMV_HIP_OK; } /* borrowed arrays: nothing resident */
    if (st && st->multi) {
        if (!Val) { spmv_set_error(SPMV_HIP_E_ARG, "update_values", "Val is NULL"); return SPMV_HIP_E_ARG; }
        if (h->Level_3_opt_used) {
            spmv_set_error(SPMV_HIP_E_ARG, "update_values", "not available on a reordered handle (option \"reorder\")");
            return SPMV_HIP_E_ARG;
        }
        rc = spmv_shim_multi_update_values(st->multi, Val);
        if (rc) { spmv_set_error(rc, "update_values", spmv_shim_error_text()); return rc; }
        h->Matrix_Val = (void *) Val;
        watch_values(h, st, Val, spmv_shim_multi_nnz(st->multi));
        return SPMV_HIP_OK;
    }
    st = state_of(h, "update_values");
    if (!st) return SPMV_HIP_E_NOSTATE;
    if (!Val) { spmv_set_error(SPMV_HIP_E_ARG, "update_values", "Val is NULL"); return SPMV_HIP_E_ARG; }
    if (h->Level_3_opt_used) { /* the resident matrix is P A P^T: its value order is not the caller's */
        spmv_set_error(SPMV_HIP_E_ARG, "update_values", "not available on a reordered handle (option \"reorder\")");
        return SPMV_HIP_E_ARG;
    }
    rc = spmv_shim_update_values(st->dev, Val);
    if (rc) { spmv_set_error(rc, "update_values", spmv_shim_error_text()); return rc; }
    h->Matrix_Val = (void *) Val;
    watch_values(h, st, Val, st->val_words / (long long) ((h->data_size == sizeof(double) ? sizeof(double) : sizeof(float)) / 4));
    return SPMV_HIP_OK;
}

long spmv_hip_get_handle_option(spmv_Handle_t h, const char *key)
{
    spmv_hip_state *st = h ? (spmv_hip_state *) h->extraHandle : NULL;
    return st ? spmv_options_get(&st->opts, key) : -1;
}

double spmv_hip_time_launches(spmv_Handle_t h, const void *x, void *y, int warmup, int iters, float *ms_out)
{
    spmv_hip_state *st = state_of(h, "time_launches");
    double r;
    if (!st) return -1.0;
    r = spmv_shim_time(st->dev, x, y, warmup, iters, ms_out);
    if (r < 0) spmv_set_error(SPMV_HIP_E_RUNTIME, "time_launches", spmv_shim_error_text());
    return r;
}
