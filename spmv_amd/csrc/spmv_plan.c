/*
 * spmv_plan.c -- SPMV_METHODS -> GPU schedule policy, plain C (north_star: "SPMV_METHODS mapping
 * to GPU schedules").  The reference hard-wires its tuning constants at create
 * (C = 4, Times = m/nthreads/C: common.c:139-140; CSR5 sigma = 16: csr5_spmv.cpp:30); here they
 * are options with defaults chosen for a 64-lane wavefront.
 *
 * Method map (VECTOR_HIP column):
 *   Method_Serial        -> CSR-scalar
 *   Method_Parallel      -> CSR-vector, L lanes per row from the mean row length
 *   Method_Balanced(2)   -> same rule as the reference's parallel_balanced2_get_handle
 *                           (parallel_balanced2_spmv.c:72-92): if some row is longer than one
 *                           worker's share the handle becomes Method_Balanced2 = nnz-split with
 *                           carries, otherwise Method_Balanced = equal-nnz row blocks.  On the GPU
 *                           the worker is a lane group (64 steps of 4L elements).
 *   Method_Balanced_Yid  -> nnz-split with carries (parallel_balanced_Yid_spmv.c:16-53 semantics)
 *   Method_SellCSigma    -> SELL-C-sigma, C = 64, sigma = 1024
 *   Method_CSR5SPMV      -> CSR5, omega = 64 (fp32 too: the reference falls back to SELL for
 *                           fp32, common.c:174-181; this build has a native fp32 CSR5)
 */
#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_hip.h"
#include "spmv_internal.h"

/* ---------------------------------------------------------------- options
 * Three layers, resolved ONCE per handle at create (spmv_options_snapshot) and stored in the handle:
 *   process-wide defaults   spmv_hip_set_option / env SPMV_HIP_<KEY>      (mutex-protected)
 *   per-thread overrides    spmv_hip_set_thread_option                    (thread-local: two threads
 *                           creating handles with different tunings do not race)
 *   the handle's snapshot   spmv_hip_get_handle_option; re-used when spmv() is handed another matrix
 * The reference has no options at all: its constants are compiled in (common.c:139-140, csr5_spmv.cpp:30). */
typedef struct { const char *key; long value; long lo, hi; int pow2; int env_read; } opt_t;
static opt_t g_opts[SPMV_N_OPTS] = {
    [SPMV_OPT_LANES_PER_ROW] = {"lanes_per_row", 0, 0, 64, 1, 0},     /* CSR-vector: 0 = from the row-length histogram */
    [SPMV_OPT_SELL_C] = {"sell_c", 64, 64, 64, 1, 0},                 /* one wavefront per chunk: C is the wave width */
    [SPMV_OPT_SELL_SIGMA] = {"sell_sigma", 1024, 64, 1 << 20, 1, 0},
    [SPMV_OPT_SELL_LDS_X] = {"sell_lds_x", 1, 0, 1, 0, 0},            /* SELL: stage the x windows of the sigma windows in LDS */
    [SPMV_OPT_SELL_LONG_THR] = {"sell_long_thr", 0, 0, 1 << 20, 0, 0},/* SELL: rows longer than this leave the slabs for the long-row path (0 = from the histogram, choose_sell_long_threshold) */
    [SPMV_OPT_CSR5_SIGMA] = {"csr5_sigma", 0, 0, 16, 0, 0},           /* CSR5 / nnz-split tiles of 64 x sigma entries: 0 = auto, else 4, 8, 16 */
    [SPMV_OPT_ROWBLOCK_NNZ] = {"rowblock_nnz", 0, 0, 1 << 20, 0, 0},  /* Balanced: equal-nnz share of one row block, 0 = 256 mean-length rows */
    [SPMV_OPT_CACHE_BLOCK] = {"cache_block", 1, 0, 2, 0, 0},          /* row-block x column-slab executor when no x window can be staged:
                                                                       * 1 = automatic (every schedule but CSR-scalar), 2 = always, 0 = never */
    [SPMV_OPT_SLAB_KIB] = {"slab_kib", 0, 0, 1 << 16, 1, 0},          /* ... KiB of x per column slab (0 = as narrow as the cell table allows) */
    [SPMV_OPT_BLOCK_ROWS] = {"block_rows", 0, 0, 16384, 1, 0},        /* ... uniform blocks of that many rows (0 = equal-work blocks, two per CU) */
    [SPMV_OPT_SPLIT] = {"split", 1, 0, 1, 0, 0},                      /* 1: a matrix with locality in PART of its entries may be multiplied as A_near (tile schedule) +
                                                                       * A_far (blocked executor) when create() measures that faster (kernels/split.hpp); 0: never */
    /* executor-form selectors: what create() would otherwise choose by rule or by timing.  Tests force every form through them; setting any of
     * them switches the create-time timing of alternatives off for that handle. */
    [SPMV_OPT_VECTOR_FORM] = {"vector_form", 0, 0, 12, 0, 0},         /* CSR-vector kernel form: 0 = timed at create; 4 pipe, 5 / 12 tile two steps deep (with / without the
                                                                       * pre-issued step), 10 / 11 tile four deep, 6 tile eight deep (shim/vector_forms.hpp) */
    [SPMV_OPT_X_WINDOWS] = {"x_windows", 1, 0, 1, 0, 0},              /* 1: tile schedules stage the x windows of their tile groups in LDS; 0: never (global gathers; the blocked
                                                                       * executor does not take over either) */
    [SPMV_OPT_XCD_ORDER] = {"xcd_order", 1, 0, 1, 0, 0},              /* tile kernels that gather x through L2: 1 = XCD-aware block order, 0 = dispatch order */
    [SPMV_OPT_CSR5_TWO_DEEP] = {"csr5_two_deep", 0, 0, 2, 0, 0},      /* staged CSR5 group kernel two tiles deep: 0 = fp32 only (measured), 1 = never, 2 = also fp64 */
    [SPMV_OPT_RUN_TILES] = {"run_tiles", 1, 0, 1, 0, 0},              /* 1: tiles / groups whose rows are runs of consecutive columns read no column stream (RUN), tiles whose rows span
                                                                       * under 256 slots a byte per entry (BYTE); 0: every staged tile reads its 16-bit slot stream */
    [SPMV_OPT_ROW_FORWARD] = {"row_forward", 1, 0, 1, 0, 0},          /* nnz-split tiles that gather through L2 (short heavy-tailed rows): 1 = a row cut by a tile boundary is finished by the tile
                                                                       * it starts in, rows longer than a tile by a workgroup each -- ONE launch; 0 = carries + fix-up launch (kernels/csr5.hpp) */
    [SPMV_OPT_AUTO_METHOD] = {"auto_method", 0, 0, 2, 0, 0},          /* 1: create() picks the schedule from the matrix by rules (two stages, spmv_api.c);
                                                                       * 2: ... by building the candidate schedules and timing them */
    [SPMV_OPT_AUTOTUNE] = {"autotune", 1, 0, 1, 0, 0},                /* 1: create() times the CSR-vector kernel forms on matrices >= 2^24 nnz */
    [SPMV_OPT_REORDER] = {"reorder", 0, 0, 2, 0, 0},                  /* 1: square matrices are RCM-reordered at create, on the device (kernels/rcm.hpp); handle->index = permutation;
                                                                       * 2: the host BFS of round 1 (reorder/rcm.c; also what multi-GPU handles use) */
    [SPMV_OPT_HOST_ROWS] = {"host_rows", 0, 0, 1, 0, 0},              /* 1: VECTOR_NONE + Method_Serial / Method_Parallel run the plain-C row loop on
                                                                       * the host (host_rows.c; BASELINE config 1).  Never chosen by itself. */
    [SPMV_OPT_CHECK_VALUES] = {"check_values", 2, 0, 2, 0, 0},        /* spmv() watches Matrix_Val for changes IN PLACE behind an unchanged pointer (the reference re-reads it on
                                                                       * every call, common.c:286-298) and refreshes the resident copies: 1 = full position-weighted checksum on
                                                                       * every call (host or device array); 2 (default) = HOST arrays only, a SAMPLED checksum -- every 64th word
                                                                       * and both ends: any whole-array update (Newton step, time step) is seen for 1/64 of the read; 0 = never */
    [SPMV_OPT_GPUS] = {"gpus", 0, 0, 64, 0, 0},                       /* > 0: row blocks over min(gpus, visible devices) GPUs in this one process (multi.hpp) */
    [SPMV_OPT_KEEP_COLUMNS] = {"keep_columns", 0, 0, 1, 0, 0},        /* 0: the resident int32 ColIdx copy is released at the end of create() when the schedule's multiply never reads it
                                                                       * (every tile / group staged: 4 B per non-zero less); 1: always kept */
    [SPMV_OPT_BLK_WAVES] = {"blk_waves", 0, 0, 8, 1, 0},              /* row-block x column-slab executor: wavefronts that share ONE row block's accumulators: 1 = a wave per block, two blocks
                                                                       * per CU (rounds 2-3); 4 / 8 = one block of up to ~20 k rows per CU (kernels/blocked.hpp "wide form"); 0 = automatic */
    [SPMV_OPT_BLK_GROUPS] = {"blk_groups", 0, 0, 12, 0, 0},           /* ... groups per pipeline step, 0 = create() times the forms of the chosen width and keeps the faster */
    [SPMV_OPT_BLK_SUBSORT] = {"blk_subsort", 1, 0, 1, 0, 0},          /* ... sparse (block, slab) cells stored sorted by column (1) or in CSR order (0: round 3's order, A/B) */
    [SPMV_OPT_DETERMINISTIC] = {"deterministic", 1, 0, 1, 0, 0},      /* 1: results are bit-reproducible run to run and handle to handle (every executor); 0: the wide blocked form may let its
                                                                       * waves add into the shared accumulators in arrival order (faster, correct to rounding, not reproducible) */
    [SPMV_OPT_X_EXCHANGE] = {"x_exchange", 0, 0, 2, 0, 0},            /* multi-GPU: 0 = allgather of the x slices, 1 = range (each device gets x[min col .. max col] of its block), 2 = broadcast from device 0 */
};

static pthread_mutex_t g_opt_lock = PTHREAD_MUTEX_INITIALIZER;
static _Thread_local long t_over[SPMV_N_OPTS];
static _Thread_local unsigned char t_over_set[SPMV_N_OPTS];

static int is_pow2_or_zero(long v) { return v == 0 || (v & (v - 1)) == 0; }
static int opt_legal(const opt_t *o, long v) { return v >= o->lo && v <= o->hi && (!o->pow2 || is_pow2_or_zero(v)); }

/* index of `key`, -1 if unknown; SPMV_HIP_<KEY> presets the process-wide value once (call under the lock) */
static int opt_find(const char *key)
{
    int i;
    if (!key) return -1;
    for (i = 0; i < SPMV_N_OPTS; ++i) {
        opt_t *o = &g_opts[i];
        if (strcmp(o->key, key) != 0) continue;
        if (!o->env_read) {
            char name[64] = "SPMV_HIP_";
            size_t k, off = strlen(name);
            const char *e;
            for (k = 0; key[k] && off + k + 1 < sizeof name; ++k) name[off + k] = (char) toupper((unsigned char) key[k]);
            name[off + k] = 0;
            e = getenv(name);
            if (e && *e) {
                long v = strtol(e, NULL, 10);
                if (opt_legal(o, v)) o->value = v;
            }
            o->env_read = 1;
        }
        return i;
    }
    return -1;
}

int spmv_hip_set_option(const char *key, long value)
{
    int i, rc = SPMV_HIP_OK;
    pthread_mutex_lock(&g_opt_lock);
    i = opt_find(key);
    if (i < 0 || !opt_legal(&g_opts[i], value)) rc = SPMV_HIP_E_ARG;
    else g_opts[i].value = value;
    pthread_mutex_unlock(&g_opt_lock);
    if (rc) spmv_set_error(rc, "set_option", key ? key : "(null)");
    return rc;
}

long spmv_hip_get_option(const char *key)
{
    long v = -1;
    int i;
    pthread_mutex_lock(&g_opt_lock);
    i = opt_find(key);
    if (i >= 0) v = t_over_set[i] ? t_over[i] : g_opts[i].value;
    pthread_mutex_unlock(&g_opt_lock);
    return v;
}

int spmv_hip_set_thread_option(const char *key, long value)
{
    int i, rc = SPMV_HIP_OK;
    pthread_mutex_lock(&g_opt_lock);
    i = opt_find(key);
    if (i < 0 || !opt_legal(&g_opts[i], value)) rc = SPMV_HIP_E_ARG;
    pthread_mutex_unlock(&g_opt_lock);
    if (rc) { spmv_set_error(rc, "set_thread_option", key ? key : "(null)"); return rc; }
    t_over[i] = value;
    t_over_set[i] = 1;
    return SPMV_HIP_OK;
}

void spmv_hip_clear_thread_options(void) { memset(t_over_set, 0, sizeof t_over_set); }

void spmv_options_snapshot(spmv_options *out)
{
    int i;
    pthread_mutex_lock(&g_opt_lock);
    for (i = 0; i < SPMV_N_OPTS; ++i) {
        (void) opt_find(g_opts[i].key); /* env preset */
        out->v[i] = t_over_set[i] ? t_over[i] : g_opts[i].value;
    }
    pthread_mutex_unlock(&g_opt_lock);
}

long spmv_options_get(const spmv_options *o, const char *key)
{
    int i;
    for (i = 0; i < SPMV_N_OPTS; ++i)
        if (key && strcmp(g_opts[i].key, key) == 0) return o->v[i];
    return -1;
}

/* ---------------------------------------------------------------- policy */
static int pow2_at_least(double v, int lo, int hi)
{
    int p = lo;
    while (p < hi && (double) p < v) p <<= 1;
    return p;
}

/*
 * CSR-vector shape -- L lanes per row and the long-row threshold -- from the row-length histogram.
 * Cost model, in wave steps (64 lanes x 4 entries = 256 entry slots):
 *   - a step serves R = 64/L rows and lasts as long as its longest row: ceil(len / 4L) iterations,
 *     the first one costing 1, later ones 3 (the tail loop is not software-pipelined: each pass waits out a full memory latency).  With rows
 *     taken as independent draws from the histogram, E[max] = sum_b cost(b) (F_b^R - F_(b-1)^R);
 *   - rows above the threshold run on the CSR5 sub-matrix path at nnz / 256 steps x 1.1.
 * Thresholds are tried at the histogram's bucket bounds (>= 64 and >= 4L) and at the historical
 * rule max(64 L, 256).  A matrix of equal rows gets L = len/4 and no long rows; a skewed one gets a
 * small L for its many short rows and hands the heavy tail to CSR5 (config 4: 1.11 -> 0.7 ms).
 */
static double pow_int(double f, int r) /* r = power of two */
{
    while (r > 1) { f *= f; r >>= 1; }
    return f;
}

static void choose_vector_shape(const spmv_stats *st, int *lanes_out, int *thr_out, double *cost_out)
{
    const int NB = SPMV_LEN_BUCKETS;
    double best = -1.0;
    int L, best_l = pow2_at_least(st->mean_row_len / 4.0, 1, 64), best_thr = 0;
    long long rows_all = 0;
    int b;
    for (b = 0; b < NB; ++b) rows_all += st->hist_rows[b];
    *cost_out = -1.0;
    if (rows_all <= 0) { *lanes_out = best_l; *thr_out = 0; return; }
    for (L = 1; L <= 64; L <<= 1) {
        const int R = 64 / L;
        const int def_thr = L * 64 > 256 ? L * 64 : 256;
        int cand;
        for (cand = -1; cand < NB - 1; ++cand) { /* -1: the default rule; else thr = 4 << cand */
            const int thr = cand < 0 ? def_thr : (4 << cand);
            long long rows_short = 0, cum = 0;
            double nnz_long = 0.0, e_max = 0.0, f_prev = 0.0, cost;
            int last_short = -1;
            if (cand >= 0 && (thr < 64 || thr < 4 * L || thr >= def_thr)) continue;
            for (b = 0; b < NB; ++b) {
                const int upper_ok = b < NB - 1 ? (4 << b) <= thr : st->max_row_len <= thr;
                if (upper_ok) { rows_short += st->hist_rows[b]; last_short = b; }
                else nnz_long += (double) st->hist_nnz[b];
            }
            for (b = 0; b <= last_short && rows_short > 0; ++b) {
                double mean_len, it, f;
                if (st->hist_rows[b] == 0) continue;
                cum += st->hist_rows[b];
                mean_len = (double) st->hist_nnz[b] / (double) st->hist_rows[b];
                it = (double) (long long) ((mean_len + 4.0 * L - 1.0) / (4.0 * L));
                if (it < 1.0) it = 1.0;
                f = pow_int((double) cum / (double) rows_short, R);
                e_max += (1.0 + 3.0 * (it - 1.0)) * (f - f_prev);
                f_prev = f;
            }
            cost = (double) rows_short / R * e_max + nnz_long / 256.0 * 1.1;
            if (best < 0.0 || cost < best * 0.98 || (cost <= best && L > best_l)) { /* 2 % hysteresis towards fewer, simpler pieces */
                best = cost;
                best_l = L;
                best_thr = cand < 0 ? 0 : thr;
            }
        }
    }
    *lanes_out = best_l;
    *thr_out = best_thr;
    *cost_out = best;
}

/*
 * SELL-C-sigma: which rows stay in the slabs?  Rows are sorted by length inside a sigma window, so a chunk of 64 rows pads
 * to its longest row: a length class with N rows per window spreads over N / 64 chunks and pads each row by about half
 * the lengths one chunk spans, (range of the class) x min(1, 64 / N) / 2 -- negligible for the bulk of the rows, ruinous
 * for a class with fewer than a chunk's worth of rows per window (config 4: 9 % of the rows hold 64..256 entries = 92
 * rows per 1024-row window = 1.4 chunks, padded by ~40 %; the rule of round 1, "rows longer than max(64, 8 x mean)
 * leave the slabs", kept them in: padded / stored = 1.30, 0.71 ms; with them on the long-row (CSR5) path: 0.65 ms).
 * The threshold is chosen at a histogram bucket bound to minimise the estimated bytes: padded slab slots for the rows
 * below it, nnz x 1.05 on the long-row path for the rows above it.  Returns 0 for "no row leaves the slabs".
 */
static int choose_sell_long_threshold(const spmv_stats *st, int sigma)
{
    const int NB = SPMV_LEN_BUCKETS;
    double best_cost = -1.0;
    int best_thr = 0, best_cand = SPMV_LEN_BUCKETS, cand, b;
    if (st->m <= 0 || st->nnz <= 0) return 0;
    for (cand = NB; cand >= 1; --cand) { /* buckets 0 .. cand-1 stay in the slabs; start from "all of them" */
        double cost = 0.0;
        for (b = 0; b < NB; ++b) {
            const double rows = (double) st->hist_rows[b], nnz = (double) st->hist_nnz[b];
            if (rows <= 0.0) continue;
            if (b < cand) {
                double upper = b < NB - 1 ? (double) (4 << b) : (double) st->max_row_len;
                if (upper > (double) st->max_row_len) upper = (double) st->max_row_len; /* the top class in use ends at the longest row, not at its bucket bound */
                const double lower = b == 0 ? 0.0 : (double) (4 << (b - 1));
                const double per_window = rows * (double) sigma / (double) st->m;
                double span = (upper - lower) * (per_window >= 64.0 ? 64.0 / per_window : 1.0);
                if (per_window < 64.0) span += upper - nnz / rows; /* shares its chunk with longer rows: padded to the class bound at least */
                cost += nnz + rows * span * 0.5;
            } else {
                cost += nnz * 1.05;
            }
        }
        if (best_cost < 0.0 || cost < best_cost * 0.97) { /* rows leave the slabs only for a clear (3 %) gain */
            best_cost = cost;
            best_cand = cand;
        }
    }
    /* a class right below the cut with less than one chunk's worth of rows per window would pad a chunk of shorter rows: out too */
    while (best_cand < NB && best_cand > 1) {
        const double per_window = (double) st->hist_rows[best_cand - 1] * (double) sigma / (double) st->m;
        if (per_window > 0.0 && per_window < 64.0) --best_cand; else break;
    }
    best_thr = best_cand >= NB ? 0 : (4 << (best_cand - 1));
    return best_thr;
}

void spmv_plan_choose(SPMV_METHODS requested, const spmv_stats *st, size_t value_size, const spmv_options *opt,
                      spmv_plan *plan, SPMV_METHODS *actual, int allow_auto)
{
    long lanes = opt->v[SPMV_OPT_LANES_PER_ROW];
    long rb = opt->v[SPMV_OPT_ROWBLOCK_NNZ];
    double vector_cost = -1.0; /* wave steps of the best CSR-vector shape (choose_vector_shape's model) */
    memset(plan, 0, sizeof *plan);
    plan->vector_form = (int) opt->v[SPMV_OPT_VECTOR_FORM];
    plan->x_windows = (int) opt->v[SPMV_OPT_X_WINDOWS];
    plan->xcd_order = (int) opt->v[SPMV_OPT_XCD_ORDER];
    plan->csr5_two_deep = (int) opt->v[SPMV_OPT_CSR5_TWO_DEEP];
    plan->run_tiles = (int) opt->v[SPMV_OPT_RUN_TILES];
    plan->row_forward = (int) opt->v[SPMV_OPT_ROW_FORWARD];
    plan->forced = plan->vector_form != 0 || plan->x_windows != 1 || plan->xcd_order != 1 || plan->csr5_two_deep != 0 || plan->run_tiles != 1 || plan->row_forward != 1;
    plan->autotune = (int) opt->v[SPMV_OPT_AUTOTUNE];
    plan->sell_c = (int) opt->v[SPMV_OPT_SELL_C];
    plan->sell_sigma = (int) opt->v[SPMV_OPT_SELL_SIGMA];
    plan->sell_lds_x = (int) opt->v[SPMV_OPT_SELL_LDS_X];
    plan->sell_long_thr = (int) opt->v[SPMV_OPT_SELL_LONG_THR];
    plan->cache_block = (int) opt->v[SPMV_OPT_CACHE_BLOCK];
    plan->slab_kib = (int) opt->v[SPMV_OPT_SLAB_KIB];
    plan->block_rows = (int) opt->v[SPMV_OPT_BLOCK_ROWS];
    plan->blk_waves = (int) opt->v[SPMV_OPT_BLK_WAVES];
    plan->blk_groups = (int) opt->v[SPMV_OPT_BLK_GROUPS];
    plan->blk_subsort = (int) opt->v[SPMV_OPT_BLK_SUBSORT];
    plan->deterministic = (int) opt->v[SPMV_OPT_DETERMINISTIC];
    plan->csr5_sigma = (int) opt->v[SPMV_OPT_CSR5_SIGMA];
    /* one workgroup's equal-nnz share (Method_Balanced): the non-zeros of 256 mean-length rows, so that a
     * block is about one 256-row slab of the CSR-vector wave program (8192 for config 2; a share that is
     * not a multiple of the slab leaves three of the four waves idle in the block's last slab: on the
     * 27-point stencil Balanced went from 1.18x to 1.04x of CSR-vector's time); a row longer than the share flips the handle to Method_Balanced2 like
     * the reference */
    if (rb > 0) plan->rowblock_nnz = (int) rb;
    else {
        double share = 256.0 * st->mean_row_len;
        if (share < 2048.0) share = 2048.0;
        if (share > 65536.0) share = 65536.0;
        plan->rowblock_nnz = (int) share;
    }
    /* CSR-vector: a lane group of L lanes takes 4L entries of its row per step (16 B loads); L and
     * the long-row threshold come from the row-length histogram (choose_vector_shape) */
    if (lanes > 0) plan->lanes_per_row = (int) lanes; /* forced: default long-row rule */
    else choose_vector_shape(st, &plan->lanes_per_row, &plan->long_thr, &vector_cost);
    (void) value_size;
    /* the reference's README ends on an empty "Matrix inspect and choose best
     * method to run" heading (README.md:222).  With auto_method = 1 the request is replaced by:
     *   regular rows (longest row <= 4 x mean, mean >= 4, < 1 % empty rows) that fill the lane groups'
     *   4L-entry chunks to >= 90 %                                            -> CSR-vector
     *   everything else (skewed, very short or many empty rows, or rows that leave the chunks
     *   emptier -- 27-entry stencil rows in 32-entry chunks: 84 %)            -> CSR5
     * (measured: CSR-vector leads on regular shapes that fill its chunks, CSR5 elsewhere -- DESIGN.md
     * section 3).  spmv_api.c adds a second stage for matrices without column locality.  The handle reports
     * the method actually used. */
    if (allow_auto && opt->v[SPMV_OPT_AUTO_METHOD] >= 1 && st->m > 0) {
        const int regular = st->mean_row_len >= 4.0 && (double) st->max_row_len <= 4.0 * st->mean_row_len &&
                            (double) st->empty_rows <= 0.01 * (double) st->m;
        /* chunk fill of CSR-vector with the L just chosen: nnz / sum over rows of ceil(len / 4L) * 4L, rows
         * taken at their bucket's mean length (exact when all rows are equal) */
        double slots = 0.0;
        const double chunk = 4.0 * plan->lanes_per_row;
        int b;
        for (b = 0; b < SPMV_LEN_BUCKETS; ++b)
            if (st->hist_rows[b] > 0) {
                const double mean_len = (double) st->hist_nnz[b] / (double) st->hist_rows[b];
                double passes = (double) (long long) ((mean_len + chunk - 1.0) / chunk);
                if (passes < 1.0) passes = 1.0;
                slots += (double) st->hist_rows[b] * passes * chunk;
            }
        requested = regular && (slots <= 0.0 || (double) st->nnz >= 0.9 * slots) ? Method_Parallel : Method_CSR5SPMV;
    }
    *actual = requested;
    switch (requested) {
    case Method_Parallel:
        plan->sched = SPMV_SCHED_CSR_VECTOR;
        break;
    case Method_Balanced:
    case Method_Balanced2: {
        /* the reference flips to Balanced2 when some row is longer than one worker's equal-nnz share
         * (parallel_balanced2_spmv.c:72-92).  Here a worker of the row-granular schedule is one lane
         * group: 64 steps of 4L elements (beyond that the long-row kernels take over), capped by the
         * workgroup share. */
        int worker_share = plan->lanes_per_row * 64 > 256 ? plan->lanes_per_row * 64 : 256;
        if (worker_share > plan->rowblock_nnz) worker_share = plan->rowblock_nnz;
        if (st->max_row_len > worker_share) {
            *actual = Method_Balanced2;
            plan->sched = SPMV_SCHED_NNZ_SPLIT;
        } else {
            *actual = Method_Balanced;
            plan->sched = SPMV_SCHED_ROWBLOCK;
        }
        break;
    }
    case Method_Balanced_Yid:
        plan->sched = SPMV_SCHED_NNZ_SPLIT;
        break;
    case Method_SellCSigma:
        plan->sched = SPMV_SCHED_SELL;
        if (plan->sell_long_thr == 0) { /* automatic: from the row-length histogram; never above the round-1 rule max(64, 8 x mean) */
            const int t = choose_sell_long_threshold(st, plan->sell_sigma);
            double cap = 8.0 * st->mean_row_len;
            if (cap < 64.0) cap = 64.0;
            if (t > 0 && (double) t < cap) plan->sell_long_thr = t;
        }
        break;
    case Method_CSR5SPMV:
        plan->sched = SPMV_SCHED_CSR5;
        break;
    case Method_Serial:
    default:
        *actual = Method_Serial;
        plan->sched = SPMV_SCHED_CSR_SCALAR;
        break;
    }
    /* Row-granular schedules on very short, heavy-tailed rows (webbase-1M-style: mean 2.6, max 4.7 k): a lane group of CSR-vector
     * or a chunk column of SELL serves a handful of entries per step, and the same cost model that shapes CSR-vector prices the
     * equal-nnz tiles (nnz / 256 x 1.1 steps) several times cheaper -- measured 0.046 ms (CSR-vector, SELL) against 0.027 ms
     * (nnz-split) on the 1e6-row stand-in.  When the model says "under half", the multiply is handed to the nnz-split executor
     * below the method, like the blocked executor is for matrices without locality: the handle keeps reporting the method asked for.
     * A forced lanes_per_row (or executor form) keeps the named schedule, and so does an explicit SELL request that sets
     * any of SELL's own options (sigma, long-row threshold, plain slab kernel): whoever tunes the schedule gets the schedule. */
    if ((plan->sched == SPMV_SCHED_CSR_VECTOR || plan->sched == SPMV_SCHED_SELL) && lanes == 0 && !plan->forced && vector_cost > 0.0 &&
        !(plan->sched == SPMV_SCHED_SELL && (opt->v[SPMV_OPT_SELL_SIGMA] != 1024 || opt->v[SPMV_OPT_SELL_LONG_THR] != 0 || opt->v[SPMV_OPT_SELL_LDS_X] != 1)) &&
        st->mean_row_len < 8.0 && (double) st->nnz / 256.0 * 1.1 < 0.5 * vector_cost)
        plan->sched = SPMV_SCHED_NNZ_SPLIT;
}
