/* spmv_internal.h -- private to the host C side (spmv_api.c, spmv_plan.c, host_rows.c). */
#ifndef SPMV_INTERNAL_H
#define SPMV_INTERNAL_H
#include <stddef.h>
#include "spmv_Defines.h"
#include "spmv_shim.h"

/* Option ids (keys and ranges: spmv_plan.c).  A handle carries the values it was created with. */
enum {
    SPMV_OPT_LANES_PER_ROW, SPMV_OPT_SELL_C, SPMV_OPT_SELL_SIGMA, SPMV_OPT_SELL_LDS_X, SPMV_OPT_SELL_LONG_THR,
    SPMV_OPT_CSR5_SIGMA, SPMV_OPT_ROWBLOCK_NNZ, SPMV_OPT_CACHE_BLOCK, SPMV_OPT_SLAB_KIB, SPMV_OPT_BLOCK_ROWS,
    SPMV_OPT_VECTOR_FORM, SPMV_OPT_X_WINDOWS, SPMV_OPT_XCD_ORDER, SPMV_OPT_CSR5_TWO_DEEP, SPMV_OPT_RUN_TILES, SPMV_OPT_ROW_FORWARD, SPMV_OPT_AUTO_METHOD, SPMV_OPT_AUTOTUNE, SPMV_OPT_REORDER, SPMV_OPT_HOST_ROWS,
    SPMV_OPT_CHECK_VALUES, SPMV_OPT_GPUS, SPMV_OPT_X_EXCHANGE, SPMV_OPT_SPLIT,
    SPMV_OPT_KEEP_COLUMNS, SPMV_OPT_BLK_WAVES, SPMV_OPT_BLK_GROUPS, SPMV_OPT_BLK_SUBSORT, SPMV_OPT_DETERMINISTIC,
    SPMV_N_OPTS
};
typedef struct spmv_options { long v[SPMV_N_OPTS]; } spmv_options;
/* process-wide values overlaid by the calling thread's overrides */
void spmv_options_snapshot(spmv_options *out);
long spmv_options_get(const spmv_options *o, const char *key); /* -1: unknown key */

/* What handle->extraHandle points to.  The reference hangs a per-method struct there
 * (balancedEnv, balancedYidEnv, sigmaEnv, anonymouslibHandle: SURVEY 8a a10-a15); here it is
 * one struct whatever the method, and the per-schedule products live in HBM inside `dev`. */
typedef struct spmv_hip_state {
    spmv_dev *dev;          /* one GPU (NULL for host-rows and multi-GPU handles) */
    struct spmv_multi *multi; /* row blocks over several GPUs of this process (option "gpus"), else NULL */
    int host_rows;          /* 1: VECTOR_NONE + option "host_rows": the plain-C row loop (host_rows.c), no device state */
    SPMV_METHODS requested; /* method asked for at create (handle->spmvMethod may be rewritten) */
    spmv_plan plan;
    spmv_options opts;      /* the options this handle was created with */
    int m, n;
    void *stream;
    int stream_set, async, warned_rebuild;
    unsigned long long val_sum; /* option "check_values": checksum of Matrix_Val as last uploaded */
    int val_sum_valid;      /* 0: values are not watched; 1: full checksum (option check_values = 1); 2: SAMPLED checksum of a HOST array (the default) */
    long long val_words;    /* 32-bit words of Matrix_Val (nnz x size / 4) */
    int from_blocks;        /* multi-GPU handle created from separate row blocks (spmv_hip_create_handle_from_blocks): spmv() ignores its CSR arguments */
} spmv_hip_state;

void spmv_set_error(int code, const char *where, const char *what);

/* Policy: reference method id + row statistics + options -> GPU schedule and its parameters.
 * *actual receives the method id the handle will report (the reference rewrites it too:
 * parallel_balanced2_spmv.c:87-92).  allow_auto = 0: ignore option "auto_method". */
void spmv_plan_choose(SPMV_METHODS requested, const spmv_stats *st, size_t value_size, const spmv_options *opt,
                      spmv_plan *plan, SPMV_METHODS *actual, int allow_auto);

/* host_rows.c */
double spmv_host_dot_d(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const double *val, const double *x);
float spmv_host_dot_s(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const float *val, const float *x);
void spmv_host_rows(BASIC_INT_TYPE m, const BASIC_INT_TYPE *rowptr, const BASIC_INT_TYPE *colidx, const void *val,
                    size_t value_size, const void *x, void *y, int threads);

#endif
