/*
 * spmv_shim.h -- the ONE place host C meets HIP ("thin shim = the only place
 * host C meets HIP").  spmv_api.c / spmv_plan.c (plain C11, gcc) sit above this interface and
 * hold the reference-shaped logic (handle life-cycle, method -> schedule policy, argument
 * rules); spmv_shim.hip (hipcc, gfx950) sits below it and holds device memory, inspectors and
 * kernels.  Plain pointers and sizes only.
 */
#ifndef SPMV_SHIM_H
#define SPMV_SHIM_H
#include <stddef.h>
#include "spmv_hip.h"

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct spmv_dev spmv_dev; /* opaque: device-resident matrix + inspector products */

enum spmv_sched {
    SPMV_SCHED_CSR_SCALAR = 0,
    SPMV_SCHED_CSR_VECTOR = 1,
    SPMV_SCHED_ROWBLOCK = 2,
    SPMV_SCHED_NNZ_SPLIT = 3,
    SPMV_SCHED_SELL = 4,
    SPMV_SCHED_CSR5 = 5,
    SPMV_SCHED_COUNT,
    SPMV_SCHED_HOST_ROWS = 100 /* no device schedule: the plain-C row loop (host_rows.c), reported by spmv_hip_get_info */
};

#define SPMV_LEN_BUCKETS 11 /* <=4, 8, 16, ..., 2048, longer */

typedef struct spmv_stats {
    int m, n;
    long long nnz;
    int max_row_len, min_row_len, empty_rows;
    double mean_row_len;
    /* row-length histogram: bucket b counts the rows with 4*2^(b-1) < len <= 4*2^b (b = 0: len <= 4;
     * the last bucket: everything longer), and the non-zeros those rows hold */
    long long hist_rows[SPMV_LEN_BUCKETS], hist_nnz[SPMV_LEN_BUCKETS];
} spmv_stats;

typedef struct spmv_plan {
    int sched;          /* enum spmv_sched */
    int lanes_per_row;  /* csr-vector */
    int long_thr;       /* csr-vector: rows longer than this go to the long-row (CSR5 sub-matrix) path; 0 = max(64 L, 256) */
    int sell_c, sell_sigma, sell_lds_x, sell_long_thr;
    int csr5_sigma;
    int slab_kib, block_rows; /* row-block x column-slab executor shape (0 = defaults) */
    int blk_waves, blk_groups, blk_subsort; /* ... waves sharing a block's accumulators (0 auto / 1 / 4 / 8), groups per step (0 = timed), sparse cells sorted by column */
    int deterministic;  /* 1: bit-reproducible results required (default) */
    int cache_block;    /* nnz-split family: 0 never, 1 automatic, 2 always use the row-block x column-slab executor */
    int rowblock_nnz;
    int vector_form, x_windows, xcd_order, csr5_two_deep, run_tiles, row_forward; /* executor-form selectors (spmv_plan.c option table) */
    int forced;         /* any of them differs from its default: create() does not time alternatives */
    int autotune;       /* csr-vector: time the applicable kernel forms at create and keep the fastest */
} spmv_plan;

/* All functions return SPMV_HIP_OK or an SPMV_HIP_E_* code and record a message retrievable
 * with spmv_shim_error_text() (thread-local). */
int spmv_shim_device_count(void);
const char *spmv_shim_error_text(void);

/* Classify + copy the CSR arrays into HBM (host or device sources), compute row statistics. */
int spmv_shim_matrix_create(spmv_dev **out, int m, int n, const int *rowptr, const int *colidx,
                            const void *val, size_t value_size);
int spmv_shim_matrix_stats(const spmv_dev *d, spmv_stats *out);
/* Run the inspector of plan->sched (device side); may be called again with another plan. */
int spmv_shim_build(spmv_dev *d, const spmv_plan *plan);
/* y = A x.  x, y: host or device pointers. */
int spmv_shim_run(spmv_dev *d, const void *x, void *y);
int spmv_shim_set_stream(spmv_dev *d, void *stream);
int spmv_shim_set_async(spmv_dev *d, int async);
int spmv_shim_sync(spmv_dev *d);
int spmv_shim_info(const spmv_dev *d, spmv_hip_info *out);
double spmv_shim_time(spmv_dev *d, const void *x, void *y, int warmup, int iters, float *ms_out);
/* min over `iters` (<= 64) launches on scratch vectors, in ms; < 0 on failure */
double spmv_shim_time_self(spmv_dev *d, int iters);
void spmv_shim_matrix_destroy(spmv_dev *d);
/* memcpy that accepts a host or a device source (the reordering inspector works on host copies) */
int spmv_shim_copy_to_host(void *dst, const void *src, size_t bytes);
/* release the device blocks the library keeps for re-use between handles (shim/state.hpp "device-memory pool") */
void spmv_shim_trim_pool(void);
/* 1 if a kernel can use the pointer as is (device or managed memory), else 0 (also without any device) */
int spmv_shim_is_device_ptr(const void *p);
/* After the last spmv_shim_build of a create: give the resident ColIdx copy back when the built schedule's multiply never reads it */
int spmv_shim_release_columns(spmv_dev *d);
/* New values (host or device, nnz entries in CSR order) behind the same pattern: copied to HBM and
 * re-permuted into the schedule's private value layouts; nothing else is rebuilt. */
int spmv_shim_update_values(spmv_dev *d, const void *val);
/* Order-independent 64-bit checksum (sum of the 32-bit words) of nnz values at `val` (host or device). */
int spmv_shim_checksum(spmv_dev *d, const void *val, unsigned long long *out);
/* the same sum over `words` 32-bit words at `val` (host or device) without a matrix: multi-GPU handles */
int spmv_shim_checksum_words(const void *val, long long words, unsigned long long *out);

/* Reverse Cuthill-McKee of the resident (square) matrix on the device; P A P^T replaces it, perm (m ints, host) receives the permutation
 * (row i of the new matrix = row perm[i] of the old).  Before spmv_shim_build. */
int spmv_shim_reorder_rcm(spmv_dev *d, int *perm_host);

/* the resident CSR arrays (device pointers; ColIdx may be NULL after spmv_shim_release_columns) */
void spmv_shim_matrix_arrays(const spmv_dev *d, const int **rowptr, const int **colidx, const void **val);

/* ---- A = A_near + A_far (shim/split.hpp): a matrix with locality in part of its entries ---- */
int spmv_shim_split_candidate(spmv_dev *d);                                   /* 1: worth building and timing */
int spmv_shim_split(spmv_dev *d, spmv_dev **near_out, spmv_dev **far_out);    /* the two halves, unplanned; far multiplies accumulating */
int spmv_shim_attach_split(spmv_dev *d, spmv_dev *near_dev, spmv_dev *far_dev, int release_parent_schedule); /* NULLs: detach + destroy */
void spmv_shim_note_split_ms(spmv_dev *d, double as_built_ms, double split_ms);

/* ---- row blocks over several GPUs of this process (shim/multi.hpp; option "gpus") ---- */
typedef struct spmv_multi spmv_multi;
/* Split the matrix into min(gpus, visible devices) equal-nnz row blocks, one shard (spmv_dev) per device, each with its
 * x buffer, y block and stream; xchg: 0 allgather, 2 broadcast.  The caller then plans + builds every shard. */
int spmv_shim_multi_create(spmv_multi **out, int gpus, int xchg, int m, int n, const int *rowptr, const int *colidx,
                           const void *val, size_t value_size);
/* ... or from G row blocks handed over separately: rows[g] rows, LOCAL 0-based RowPtr, GLOBAL columns (numa.c:277-304) */
int spmv_shim_multi_create_blocks(spmv_multi **out, int G, int xchg, const int *rows, int n, const int *const *rowptr, const int *const *colidx,
                                  const void *const *val, size_t value_size);
int spmv_shim_multi_count(const spmv_multi *mt);
int spmv_shim_multi_rows(const spmv_multi *mt);
/* the boundary rows' sub-matrix of shard g ("range" exchange with overlap), or NULL: planned and built like a shard */
spmv_dev *spmv_shim_multi_boundary(spmv_multi *mt, int g);
int spmv_shim_multi_uses_rccl(const spmv_multi *mt);
long long spmv_shim_multi_nnz(const spmv_multi *mt);
spmv_dev *spmv_shim_multi_shard(spmv_multi *mt, int g);
/* y = A x with full-length host or device vectors: upload / exchange / multiply / collect */
int spmv_shim_multi_run(spmv_multi *mt, const void *x, void *y);
/* distributed vectors: shard g's slice of x (inside its full-length copy) and its block of y, on device *device */
int spmv_shim_multi_slices(spmv_multi *mt, int g, void **x_slice, long long *x_first, long long *x_count, void **y_block,
                           long long *y_first, long long *y_count, int *device);
/* exchange the x slices between the devices and multiply; y stays in the shards' blocks */
int spmv_shim_multi_step(spmv_multi *mt);
/* the same, enqueued only (ordered behind the work already submitted to each device's default stream); _sync waits */
int spmv_shim_multi_step_async(spmv_multi *mt);
int spmv_shim_multi_sync(spmv_multi *mt);
int spmv_shim_multi_update_values(spmv_multi *mt, const void *val);
void spmv_shim_multi_destroy(spmv_multi *mt);

#if defined(__cplusplus)
}
#endif
#endif
