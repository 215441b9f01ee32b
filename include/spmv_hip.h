/*
 * spmv_hip.h -- extensions of the MI355X build.  NOT part of the reference API: a program that
 * only uses spmv.h never needs this header.  Everything here is optional control around the four
 * drop-in functions; no extension changes what spmv() computes.
 *
 * Why they exist (SURVEY 8b "Errors", "Host-pointer cost", 5 "Config / flags"):
 *   - the reference API is all-void with no error channel      -> spmv_hip_last_error*()
 *   - the reference has no notion of a device or a stream      -> spmv_hip_set_stream / _set_async / _synchronize
 *   - SELL's C and sigma and CSR5's sigma are hard-wired in the reference (common.c:139-140,
 *     csr5_spmv.cpp:30)                                         -> spmv_hip_set_option (also env SPMV_HIP_<KEY>)
 *   - measurement (hipEvent per launch on the launch stream)    -> include/spmv_hip_tools.h (bench.py and tools/ only)
 */
#include "spmv_Defines.h"
#if defined(__cplusplus)
extern "C" {
#endif
#ifndef SPMV_HIP_EXT_H
#define SPMV_HIP_EXT_H

/* ---- error channel ------------------------------------------------------------------------ */
enum {
    SPMV_HIP_OK = 0,
    SPMV_HIP_E_NODEVICE = 1,   /* no usable gfx950 device / HIP runtime failure at init */
    SPMV_HIP_E_ALLOC = 2,      /* hipMalloc failed */
    SPMV_HIP_E_ARG = 3,        /* NULL / negative / inconsistent argument */
    SPMV_HIP_E_RUNTIME = 4,    /* a HIP call or kernel launch failed */
    SPMV_HIP_E_NOSTATE = 5,    /* handle has no device state (create failed or handle was cleared) */
    SPMV_HIP_E_RANGE = 6       /* nnz or padded size does not fit the index type */
};
/* Failures are also printed to stderr (env SPMV_HIP_QUIET silences that); env SPMV_HIP_ABORT_ON_ERROR makes the
 * first failure abort() the process -- for drop-in callers that never look at the error channel.
 * Code of the most recent failure on the calling thread (0 if none since the last clear). */
int spmv_hip_last_error(void);
/* Human-readable text for it ("" if none).  Valid until the next failing call on this thread. */
const char *spmv_hip_last_error_string(void);
void spmv_hip_clear_error(void);

/* ---- device / stream ---------------------------------------------------------------------- */
/* Number of visible HIP devices (0 if none or the runtime cannot initialise). */
int spmv_hip_device_count(void);
/* Launch this handle's kernels on `hip_stream` (a hipStream_t; NULL = the default stream). */
int spmv_hip_set_stream(spmv_Handle_t handle, void *hip_stream);
/* async != 0: spmv() with DEVICE x and y returns after enqueueing (stream-ordered); the caller
 * synchronises.  Default 0: spmv() returns when Y is complete, like the reference.
 * Host x or y always synchronise. */
int spmv_hip_set_async(spmv_Handle_t handle, int async);
int spmv_hip_synchronize(spmv_Handle_t handle);
/* Device blocks freed by destroy / clear / re-inspection are kept (up to SPMV_HIP_POOL_MB MiB, default an eighth of the device's memory; 0 = keep
 * nothing) and handed to the next create: on this runtime a hipMalloc that follows a large hipFree can take seconds.
 * This returns them to the driver now. */
void spmv_hip_trim_pool(void);

/* ---- values changed in place ---------------------------------------------------------------- */
/* The reference re-reads Matrix_Val on every spmv() (common.c:286-298); this library multiplies its
 * HBM-resident copy.  After changing values IN PLACE (same pattern) call this with the array (host or device
 * pointer, RowPtr[m] entries in CSR order): the values are copied to HBM and re-permuted into the schedule's
 * private layouts by device kernels -- no re-inspection, no autotune.  spmv() also watches the array by itself (option
 * "check_values"): by default a HOST Matrix_Val -- the reference's only mode -- is sample-checksummed on every call and a change of the
 * whole array is picked up without any call; SPMV_HIP_CHECK_VALUES=1 makes that a full checksum (device arrays too), at the price of
 * reading the values once more per call.  Not available on handles created with
 * option "reorder".  Returns 0 or an SPMV_HIP_E_* code. */
int spmv_hip_update_values(spmv_Handle_t handle, const void *Matrix_Val);

/* ---- options --------------------------------------------------------------------------------
 * Resolved once per handle, at create: process-wide value (spmv_hip_set_option / env), overridden by the
 * calling thread's value (spmv_hip_set_thread_option) -- so two threads can create differently tuned handles
 * without racing -- and stored in the handle (spmv_hip_get_handle_option). */
/* keys: "lanes_per_row" (CSR-vector, 0 = auto, else 1..64 power of two)
 *       "sell_c" (64)  "sell_sigma" (1024)  "sell_lds_x" (0/1: stage narrow x windows in LDS)
 *       "sell_long_thr" (rows longer than this stay out of the slabs; 0 = from the row-length histogram: length classes with
 *                        less than a chunk's worth of rows per sigma window leave, at most max(64, 8 x mean row length) stays)
 *       "csr5_sigma" (0 = auto)  "rowblock_nnz" (equal-nnz share of one Balanced row block, 0 = auto = 8192)
 *       executor-form selectors (what create() otherwise chooses by rule or by timing; tests force every form through them):
 *       "vector_form" (CSR-vector: 0 timed at create, 4 pipe, 5 / 12 tile two deep, 10 / 11 tile four deep, 6 tile eight deep)
 *       "x_windows" (0/1, default 1: stage the tile groups' x windows in LDS)   "xcd_order" (0/1, default 1)   "csr5_two_deep" (0 auto / 1 never / 2 always)
 *       "run_tiles" (0/1, default 1: RUN / BYTE tiles -- spmv_hip_info.run_nnz, byte_nnz)
 *       "row_forward" (0/1, default 1: nnz-split tiles that gather through L2 finish the rows they start -- one launch, no carry fix-up; spmv_hip_info.launch_kernels)
 *       "autotune" (0/1, default 1: for matrices above 2^24 nnz create() times the applicable CSR-vector
 *                   kernel forms once on the resident matrix and keeps the fastest, ~10 ms)
 *       "reorder" (0/1/2, default 0; 1: reverse Cuthill-McKee on the device (kernels/rcm.hpp), 2: the host BFS of round 1.  For matrices whose band was lost to
                  a bad numbering; NOT for power-law matrices: the blocked executor is indifferent to vertex numbers and RCM scatters the hubs (measured
                  slower, profiles/r04_reorder.txt).  Square matrices are reordered at create, B = P A P^T is what stays
 *                  resident, and handle->index holds the permutation -- the caller gathers
 *                  XX[i] = X[index[i]] and scatters Y[index[i]] = YY[i] exactly as the reference's harness
 *                  does for its OPT_LEVEL 3 path, test_spmv.c:95-101, 130-137)
 *       "auto_method" (0/1/2; 2 = like 1, then every candidate schedule is built and timed on scratch vectors at
 *                      create and the fastest kept, +0.1..0.3 s for 3e8 nnz.  1: create() replaces the requested method by the one its row statistics
 *                      favour -- CSR-vector for regular rows, CSR5 otherwise -- and, if that schedule cannot
 *                      stage a single x window on a matrix whose x is far larger than an L2, by
 *                      Method_Balanced_Yid with the cache-blocked executor; the handle reports it)
 *       "cache_block" (0 never / 1 automatic (default) / 2 always: every schedule but the debug kernel CSR-scalar hands the
 *                      multiply to the row-block x column-slab executor when no x window of the matrix fits LDS,
 *                      nnz >= 2^21 and n * size >= 4 MiB (x as large as an XCD's L2; 12 MiB when rows average fewer than 8 entries): ~3x faster on columns
 *                      without locality.  A row block's products are added in an order fixed by the matrix -- one wavefront per block, or the waves
 *                      of the wide form taking turns --, so results are bit-reproducible unless option "deterministic" = 0 waives that.)
 *       "split" (0/1, default 1: a matrix whose entries are partly local, partly scattered may be multiplied as A_near + A_far when
 *               create() measures that faster -- spmv_hip_info.split_ms, far_nnz)
 *       "slab_kib" (KiB of x per column slab, 0 = as narrow as the cell table allows)
 *       "block_rows" (uniform row blocks of that many rows, at most 16384; 0 = equal-work blocks sized by the executor form: up to 9982 rows with a wave per
                     block, up to 20350 in the wide forms)
       "blk_waves" (0 automatic / 1 / 2 / 4 / 8: wavefronts sharing ONE row block's LDS accumulators; 0: create() builds the one-wave form and, on large
                    matrices, the wide form that fits option "deterministic", times them and keeps the faster)
       "blk_groups" (groups of 128 fp64 / 256 fp32 entries per pipeline step, 0 = timed at create)   "blk_subsort" (0/1, default 1)
       "keep_columns" (0/1, default 0: at the end of create() the HBM-resident int32 ColIdx copy is released when the built schedule's multiply never reads it --
                       every tile / group staged, SELL slabs, CSR5 tiles -- 4 B per non-zero less (spmv_hip_info.device_bytes); 1 keeps it)
       "deterministic" (0/1, default 1: every executor adds a row's products in an order fixed by the matrix, so results are bit-identical run to
                        run; 0 lets the wide blocked form add in arrival order -- 10-16 % faster on matrices without column locality, equal to rounding)
 *       "host_rows" (0/1, default 0: 1 = handles created with VECTOR_NONE and Method_Serial / Method_Parallel run
 *                    a plain-C row loop on the HOST over the caller's arrays (BASELINE config 1: the reference's
 *                    plumbing case); never selected automatically -- without it a missing GPU is an error)
 *       "check_values" (0/1/2, default 2: spmv() watches Matrix_Val for in-place changes and refreshes the resident copies by itself.  2: HOST arrays
                       only, by a SAMPLED checksum (every 64th word, at most 65536, and both ends: sees any update of the whole array, misses most
                       single-entry edits; under a millisecond per call); 1: the full position-weighted checksum on every call, host or device array
                       (reads the values once more per call); 0: never.  Works on multi-GPU handles (option "gpus") too.)
 *       "gpus" (0 = this handle lives on the current device; G > 0: row blocks over min(G, visible devices) GPUs
 *               of this process, see "multi-GPU" below)   "x_exchange" (multi-GPU: 0 allgather, 1 range, 2 broadcast)
 * Each key can also be preset with the environment variable SPMV_HIP_<KEY IN CAPS>.
 * Returns 0, or SPMV_HIP_E_ARG for an unknown key / illegal value. */
int spmv_hip_set_option(const char *key, long value);
long spmv_hip_get_option(const char *key);            /* the value a create() on this thread would use */
int spmv_hip_set_thread_option(const char *key, long value); /* override for handles created by the calling thread */
void spmv_hip_clear_thread_options(void);
long spmv_hip_get_handle_option(spmv_Handle_t handle, const char *key); /* what the handle was created with; -1 unknown */

/* ---- introspection ------------------------------------------------------------------------ */
typedef struct spmv_hip_info {
    int device;                 /* HIP device ordinal the handle lives on */
    int schedule;               /* 0 csr-scalar 1 csr-vector 2 row-block 3 nnz-split 4 sell-c-sigma 5 csr5 */
    int lanes_per_row;          /* csr-vector */
    int sell_c, sell_sigma;     /* sell */
    int tile_nnz;               /* nnz-split / csr5 tile size */
    int m, n;
    long long nnz;
    long long stored_nnz;       /* incl. SELL padding */
    int max_row_len, min_row_len, empty_rows;
    double mean_row_len;
    long long device_bytes;     /* HBM held by the handle */
    long long alg_bytes;        /* B_alg = 4(m+1) + nnz(4+s) + s*n + s*m    */
    double inspect_ms;          /* wall time of the inspector inside create */
    const char *schedule_name;
    const char *kernel_name;    /* symbol of the dominant kernel (as rocprofv3 shows it) */
    int tuned_choice;           /* csr-vector autotune: 0 none, 10/11 tile 4-deep, 5/12 tile 2-deep, 4 pipe (row-block schedule and csr-vector's wide form: 10 / 5 = the rows
                                   kernel four / two steps deep); cache_blocked: 100 / 101 =
                                   the smaller / larger groups-per-pipeline-step form (8 / 12; 6 / 8 with blk_waves = 8) */
    float tune_ms[3];           /* measured ms of {tile/4-deep, tile/2-deep, pipe} at create (0 if not tuned; pipe: 0 when 99 % of the tiles stage their x windows -- not timed); cache_blocked: of the
                                   row-block executor's two groups-per-step forms ([2] unused) */
    int x_groups;               /* tiles / tile groups / sigma windows the inspector analysed for x windows */
    int x_groups_staged;        /* ... of which have their x windows staged in LDS (0: every gather goes to L1/L2) */
    int cache_blocked;          /* 1: the row-block x column-slab executor runs (option "cache_block") */
    long long stream_bytes;     /* HBM bytes ONE spmv() has to move given the schedule's storage format: the value and
                                 * column streams as stored (2 B/nnz LDS slots where x windows are staged -- none for run_nnz --, padding
                                 * included), row pointers / descriptors / window tables, the x elements staged (at most
                                 * 8 n: one pass per XCD; or n once where x is gathered through L2), y written once, carries.  This -- not alg_bytes --
                                 * is what divides by the launch time to give the HBM rate actually sustained. */
    long long x_bytes;          /* the part of stream_bytes charged for reading x */
    float route_ms[2];          /* only when part of the tile groups stage their x windows and part do not: create() builds the tile
                                 * schedule AND the row-block x column-slab executor, times both -- [0] tile schedule, [1] blocked
                                 * executor, ms -- and keeps the faster (0, 0: the choice needed no measurement) */
    float split_ms[2];          /* when a sizeable part of the entries -- not all -- lies near its tile's centre column, create() also builds
                                 * A = A_near + A_far (near: the tile schedule, every tile staged; far: the blocked executor, accumulating) and
                                 * times it: [0] schedule as built, [1] the split pair, ms (0, 0: not tried) */
    long long far_nnz;          /* entries the blocked executor multiplies in a split handle (0: the handle is not split) */
    long long run_nnz;          /* CSR-vector / row-block tile kernels, SELL slabs, CSR5 tile groups: entries in RUN tiles / groups -- staged ones whose rows
                                 * each reference one run of consecutive columns (banded matrices; CSR5: of at least sigma entries); their column stream is
                                 * not read at all (16 bits, SELL: a word, per ROW; CSR5: a word per lane and tile instead) */
    long long byte_nnz;         /* CSR-vector / row-block tile kernels, SELL window groups: entries in BYTE tiles / groups -- staged ones in which every row's LDS slots lie
                                 * within 255 of the row's smallest (SELL: first) slot (banded matrices with holes, block rows): their column stream is one byte per entry
                                 * + 16 bits (SELL: a word) per row */
    long long tmpl_nnz;         /* ... entries in TEMPLATE tiles -- staged tiles whose rows all have the same slot offsets from their first entry (stencil interiors,
                                 * block rows): no column stream either, 16 bits per row + one offset list per tile */
    int blk_waves;              /* cache_blocked: wavefronts that share one row block's accumulators (1: a wave per block, two blocks per CU; 2 / 4 / 8: the
                                 * wide form, one block of up to ~20 k rows per CU); 0 when another executor runs */
    char launch_kernels[160];   /* every kernel ONE spmv() launches, in order, '+'-separated (e.g. "sell_window_kernel+csr5_group_pipe_kernel+csr5_fixup_kernel") */
    int reproducible;           /* 1: the executor adds every row's products in an order fixed by the matrix -- identical bits run to run and handle to
                                 * handle; 0 only for the wide blocked form under option "deterministic" = 0 */
} spmv_hip_info;
int spmv_hip_get_info(spmv_Handle_t handle, spmv_hip_info *out);

/* ---- multi-GPU: row blocks over the GPUs of ONE process (option "gpus"; BASELINE config 5) --------------
 * A handle created while option "gpus" = G > 0 splits the matrix into min(G, visible devices) equal-nnz row blocks, one
 * per device, each with its own schedule, stream, full-length x buffer and y block (the GPU analogue of the reference's
 * NUMA experiment, src/samples/numa.c:277-304).  spmv() keeps its signature and meaning: X and Y are FULL vectors (host
 * or device pointers); per call X is distributed (option "x_exchange": 0 = every device receives its slice and the
 * slices are all-gathered over xGMI with RCCL, 2 = X goes to device 0 and is broadcast -- north_star's literal form,
 * 1 = "range": every device receives only the columns its row block references, x[min ColIdx .. max ColIdx] -- for a
 * banded matrix its own slice and a few values either side; in the distributed step they are pulled from the
 * neighbouring devices by peer copies, no collective),
 * every device multiplies its block, and the y blocks are collected into Y.
 * A solver loop keeps the vectors DISTRIBUTED instead: write this step's x into the devices' slices, call
 * spmv_hip_multi_step (exchange + multiply, nothing crosses PCIe), read y from the devices' blocks.
 * RCCL is dlopen'ed only when G > 1; without it the exchange uses peer-to-peer copies. */
int spmv_hip_multi_gpus(spmv_Handle_t handle);        /* devices the handle spans; 0 for an ordinary handle */
int spmv_hip_multi_uses_rccl(spmv_Handle_t handle);   /* 1: the x exchange runs through RCCL communicators */
/* Device `gpu`'s slice of x (x_count elements from global index x_first, device memory on *device, inside that device's
 * full-length copy) and its block of y (rows y_first ... y_first + y_count - 1).  Any out-pointer may be NULL. */
int spmv_hip_multi_slices(spmv_Handle_t handle, int gpu, void **x_slice, long long *x_first, long long *x_count,
                          void **y_block, long long *y_first, long long *y_count, int *device);
/* Exchange the x slices between the devices and multiply; y stays distributed.  Synchronous: every device is drained first
 * (hipDeviceSynchronize), so the x slices may have been written on any stream; returns when the y blocks are complete.
 * x_exchange = 1 ("range") runs the halo copies beside the multiply and redoes the rows that needed them. */
int spmv_hip_multi_step(spmv_Handle_t handle);
/* The same, enqueued only, ordered behind the work the caller has submitted to each device's DEFAULT stream (the x slices must have
 * been written there, or be complete); spmv_hip_multi_synchronize waits for every device (results in the y blocks).  Steps may be enqueued
 * back to back (a step's halo copies wait for the previous step's multiplies); the caller writes the NEXT x slices on the default
 * streams, which a step is ordered behind -- but nothing orders those writes behind a step still running: synchronize (or wait on
 * an event of your own) before overwriting x slices a running step reads. */
int spmv_hip_multi_step_async(spmv_Handle_t handle);
int spmv_hip_multi_synchronize(spmv_Handle_t handle);
/* A multi-GPU handle from row blocks that exist separately -- the way the reference's NUMA experiment hands each memory node
 * its rows (src/samples/numa.c:277-304).  Block g: rows[g] rows, LOCAL 0-based int32 RowPtr[g] (rows[g] + 1 entries), GLOBAL
 * column indices ColIdx[g] in [0, n), values Matrix_Val[g]; host or device pointers; block g is placed on device g (at most
 * as many blocks as visible devices).  No monolithic array exists, so the int32 limit of RowPtr applies per block: BASELINE
 * config 5 (8 x 1e7 rows x 32 = 2.56e9 non-zeros) is created this way.  Options (x_exchange, ...) are read as at any create.
 * spmv() on the handle takes full-length X (n) / Y (sum of rows) and ignores its CSR arguments; spmv_hip_multi_slices /
 * _step work as for option "gpus"; spmv_hip_update_values is not available (clear and create again). */
void spmv_hip_create_handle_from_blocks(spmv_Handle_t *Handle, int blocks, const BASIC_INT_TYPE *rows, BASIC_INT_TYPE n,
                                        BASIC_INT_TYPE *const *RowPtr, BASIC_INT_TYPE *const *ColIdx, void *const *Matrix_Val,
                                        SPMV_METHODS Function, BASIC_SIZE_TYPE size);

#endif /* SPMV_HIP_EXT_H */

#if defined(__cplusplus)
}
#endif
