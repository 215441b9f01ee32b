/* spmv_internal.h -- private to the host C side (spmv_api.c, spmv_plan.c). */
#ifndef SPMV_INTERNAL_H
#define SPMV_INTERNAL_H
#include <stddef.h>
#include "spmv_Defines.h"
#include "spmv_shim.h"

/* What handle->extraHandle points to.  The reference hangs a per-method struct there
 * (balancedEnv, balancedYidEnv, sigmaEnv, anonymouslibHandle: SURVEY 8a a10-a15); here it is
 * one struct whatever the method, and the per-schedule products live in HBM inside `dev`. */
typedef struct spmv_hip_state {
    spmv_dev *dev;
    SPMV_METHODS requested; /* method asked for at create (handle->spmvMethod may be rewritten) */
    spmv_plan plan;
    int m, n;
    void *stream;
    int stream_set, async, warned_rebuild;
} spmv_hip_state;

void spmv_set_error(int code, const char *where, const char *what);

/* Policy: reference method id + row statistics -> GPU schedule and its parameters.
 * *actual receives the method id the handle will report (the reference rewrites it too:
 * parallel_balanced2_spmv.c:87-92). */
void spmv_plan_choose(SPMV_METHODS requested, const spmv_stats *stats, size_t value_size,
                      spmv_plan *plan, SPMV_METHODS *actual);
/* allow_auto = 0: ignore option "auto_method" (second stage of the automatic choice, spmv_api.c) */
void spmv_plan_choose_ex(SPMV_METHODS requested, const spmv_stats *st, size_t value_size,
                         spmv_plan *plan, SPMV_METHODS *actual, int allow_auto);

#endif
