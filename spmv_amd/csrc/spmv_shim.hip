// spmv_shim.hip -- the HIP side of the C-ABI shim (see spmv_shim.h): device memory, the
// device-side inspectors and the kernel launches.  gfx950 only.
//
// Division of labour with the reference (all CPU there):
//   matrix storage     the reference BORROWS the caller's CSR arrays (common.c:157-159); here they
//                      are copied into HBM once at create and stay resident (288 GB per GPU).
//   inspectors         parallel_balanced2_get_handle / parallel_balanced_Yid_get_handle /
//                      sell_C_Sigma_get_handle_Selected / csr5 asCSR5 run on the host in the
//                      reference; here every inspector is a device kernel over the resident CSR,
//                      so create() never walks the matrix on the CPU.
//   executors          one launch (two for nnz-split: tiles + carry fix-up; SELL: slabs + long rows).
#include <hip/hip_runtime.h>

#include <chrono>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <vector>

#include "spmv_shim.h"
#include "kernels/common.hpp"
#include "kernels/csr_rows.hpp"
#include "kernels/csr_vector4.hpp"
#include "kernels/rowblock.hpp"
#include "kernels/sell.hpp"
#include "kernels/csr5.hpp"
#include "kernels/long_rows.hpp"
#include "kernels/blocked.hpp"
#include "kernels/split.hpp"
#include "kernels/rcm.hpp"
#include "kernels/csr_vector_tile.hpp"

using namespace spmv;

#include "shim/state.hpp"
#include "shim/vector_forms.hpp"

// row statistics of the resident RowPtr (also validates it): d->nnz, d->stats
static int matrix_row_stats(spmv_dev *d)
{
    const int m = d->m, n = d->n;
    DevStats hs{0, INT_MAX, 0, 0, 0, 0};
    DevStats *ds = nullptr;
    if (pool_malloc((void **) &ds, sizeof(DevStats)) != hipSuccess) return fail(SPMV_HIP_E_ALLOC, "pool_malloc(stats)");
    (void) hipMemcpy(ds, &hs, sizeof hs, hipMemcpyHostToDevice);
    if (m > 0) {
        stats_kernel<<<grid_for(m, kBlock, d->cus * 8), kBlock>>>(m, d->rowptr, ds);
        if (hipGetLastError() != hipSuccess) { (void) pool_free(ds); return fail(SPMV_HIP_E_RUNTIME, "stats kernel launch failed"); }
    }
    hipError_t e = hipMemcpy(&hs, ds, sizeof hs, hipMemcpyDeviceToHost);
    (void) pool_free(ds);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "stats kernel: %s", hipGetErrorString(e));
    if (m > 0 && (hs.bad || hs.first != 0 || hs.last < 0))
        return fail(SPMV_HIP_E_ARG, "RowPtr must start at 0 and be non-decreasing (RowPtr[0]=%d)", hs.first);
    d->nnz = m > 0 ? hs.last : 0;
    if (d->nnz > (long long) INT_MAX - 4096)
        return fail(SPMV_HIP_E_RANGE, "nnz = %lld is within 4096 of INT_MAX: 16 B tail reads would overflow int32 indices", d->nnz);
    d->stats.m = m;
    d->stats.n = n;
    d->stats.nnz = d->nnz;
    d->stats.max_row_len = m > 0 ? hs.max_len : 0;
    d->stats.min_row_len = m > 0 ? hs.min_len : 0;
    d->stats.empty_rows = hs.empty;
    d->stats.mean_row_len = m > 0 ? (double) d->nnz / m : 0.0;
    for (int b = 0; b < SPMV_LEN_BUCKETS; ++b) {
        d->stats.hist_rows[b] = (long long) hs.hist_rows[b];
        d->stats.hist_nnz[b] = (long long) hs.hist_nnz[b];
    }
    return SPMV_HIP_OK;
}

// ------------------------------------------------------------------------------------ create
extern "C" int spmv_shim_matrix_create(spmv_dev **out, int m, int n, const int *rowptr, const int *colidx,
                                       const void *val, size_t value_size)
{
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_E_NODEVICE, "no HIP device visible (this library has no CPU path)");
    }
    spmv_dev *d = new spmv_dev();
    hipDeviceProp_t prop;
    if (hipGetDevice(&d->device) != hipSuccess || hipGetDeviceProperties(&prop, d->device) != hipSuccess) {
        (void) hipGetLastError();
        delete d;
        return fail(SPMV_HIP_E_NODEVICE, "cannot query the current HIP device");
    }
    d->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    d->m = m;
    d->n = n;
    d->vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float); // serial_spmv.c:48-54
    int rc = SPMV_HIP_OK;
    auto bail = [&](int code) { spmv_shim_matrix_destroy(d); return code; };

    if ((rc = dev_alloc(d, (void **) &d->rowptr, sizeof(int) * ((size_t) m + 1), false))) return bail(rc);
    if (m > 0) {
        if (hipMemcpy(d->rowptr, rowptr, sizeof(int) * ((size_t) m + 1), hipMemcpyDefault) != hipSuccess)
            return bail(fail(SPMV_HIP_E_RUNTIME, "copy RowPtr to HBM: %s", hipGetErrorString(hipGetLastError())));
    } else {
        (void) hipMemset(d->rowptr, 0, sizeof(int));
    }
    if ((rc = matrix_row_stats(d))) return bail(rc);
    if (d->nnz > 0 && (!colidx || !val)) return bail(fail(SPMV_HIP_E_ARG, "ColIdx / Matrix_Val is NULL"));

    // padded by kStreamPad elements: the 16 B-per-lane kernels round a row's tail read up
    if ((rc = dev_alloc(d, (void **) &d->colidx, sizeof(int) * ((size_t) d->nnz + kStreamPad), false))) return bail(rc);
    if ((rc = dev_alloc(d, &d->val, d->vsize * ((size_t) d->nnz + kStreamPad), false))) return bail(rc);
    (void) hipMemset(d->colidx + d->nnz, 0, sizeof(int) * kStreamPad);
    (void) hipMemset((char *) d->val + d->vsize * (size_t) d->nnz, 0, d->vsize * kStreamPad);
    if (d->nnz > 0) {
        if (hipMemcpy(d->colidx, colidx, sizeof(int) * (size_t) d->nnz, hipMemcpyDefault) != hipSuccess ||
            hipMemcpy(d->val, val, d->vsize * (size_t) d->nnz, hipMemcpyDefault) != hipSuccess)
            return bail(fail(SPMV_HIP_E_RUNTIME, "copy ColIdx/Val to HBM: %s", hipGetErrorString(hipGetLastError())));
        // every column index must address x: the reference would read out of bounds, a GPU would fault
        int host2[2] = {INT_MAX, INT_MIN};
        int *mnmx = nullptr;
        if (pool_malloc((void **) &mnmx, sizeof host2) != hipSuccess) return bail(fail(SPMV_HIP_E_ALLOC, "pool_malloc(colidx range)"));
        (void) hipMemcpy(mnmx, host2, sizeof host2, hipMemcpyHostToDevice);
        colidx_range_kernel<<<grid_for(d->nnz, kBlock * 16, d->cus * 8), kBlock>>>(d->nnz, d->colidx, mnmx);
        const hipError_t e2 = hipMemcpy(host2, mnmx, sizeof host2, hipMemcpyDeviceToHost);
        (void) pool_free(mnmx);
        if (e2 != hipSuccess) return bail(fail(SPMV_HIP_E_RUNTIME, "colidx range kernel: %s", hipGetErrorString(e2)));
        if (host2[0] < 0 || host2[1] >= n)
            return bail(fail(SPMV_HIP_E_ARG, "ColIdx out of range: min %d, max %d, n = %d", host2[0], host2[1], n));
        d->col_min = host2[0];
        d->col_max = host2[1];
    }
    *out = d;
    return SPMV_HIP_OK;
}

extern "C" void spmv_shim_matrix_arrays(const spmv_dev *d, const int **rowptr, const int **colidx, const void **val)
{
    if (rowptr) *rowptr = d ? d->rowptr : nullptr;
    if (colidx) *colidx = d ? d->colidx : nullptr;
    if (val) *val = d ? d->val : nullptr;
}

extern "C" int spmv_shim_copy_to_host(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return SPMV_HIP_OK;
    if (!dst || !src) return fail(SPMV_HIP_E_ARG, "copy_to_host: NULL");
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_matrix_stats(const spmv_dev *d, spmv_stats *out)
{
    if (!d || !out) return fail(SPMV_HIP_E_ARG, "stats: NULL");
    *out = d->stats;
    return SPMV_HIP_OK;
}

extern "C" void spmv_shim_matrix_destroy(spmv_dev *d)
{
    if (!d) return;
    quiesce(d); // an asynchronous multiply may still read the arrays that go back to the pool below
    if (d->sp_near) { spmv_shim_matrix_destroy(d->sp_near); d->sp_near = nullptr; }
    if (d->sp_far) { spmv_shim_matrix_destroy(d->sp_far); d->sp_far = nullptr; }
    if (d->sp_centre) { (void) pool_free(d->sp_centre); d->sp_centre = nullptr; }
    free_schedule(d);
    if (d->rowptr) (void) pool_free(d->rowptr);
    if (d->colidx) (void) pool_free(d->colidx);
    if (d->val) (void) pool_free(d->val);
    if (d->x_stage) (void) pool_free(d->x_stage);
    if (d->y_stage) (void) pool_free(d->y_stage);
    if (d->scratch8) (void) pool_free(d->scratch8);
    delete d;
}

#include "shim/inspect.hpp"

// The resident int32 ColIdx copy (4 B per non-zero: 1.28 GB on config 2) is read by the inspectors -- and afterwards only by executors that
// gather through global columns.  Once create() has settled on a schedule whose multiply never touches it (every tile / group staged: the 16-bit
// slot streams, RUN / BYTE / TEMPLATE data, SELL slabs and CSR5 tiles are copies of their own), it goes back to the pool.  Kept: CSR-scalar, the pipe
// form, any schedule with unstaged groups, the blocked executor (its values refresh re-derives the cells from the columns), split handles.
extern "C" int spmv_shim_release_columns(spmv_dev *d)
{
    if (!d || !d->built || !d->colidx || d->nnz == 0) return SPMV_HIP_OK;
    if (d->sp_near || d->sp_far || d->accumulate || d->blk_on) return SPMV_HIP_OK;
    bool unused = false;
    switch (d->plan.sched) {
    case SPMV_SCHED_CSR_VECTOR:
        unused = d->vt_tiles > 0 && d->vt_staged == d->vt_tiles && d->plan.vector_form != VEC_PIPE && d->vec_choice != VEC_PIPE;
        break;
    case SPMV_SCHED_ROWBLOCK: unused = d->vt_tiles > 0 && d->vt_staged == d->vt_tiles; break;
    case SPMV_SCHED_NNZ_SPLIT: unused = d->ns.groups > 0 && d->ns.staged == d->ns.groups; break; // natural layout: unstaged groups read the matrix's own columns
    case SPMV_SCHED_SELL:
    case SPMV_SCHED_CSR5: unused = true; break;                                                    // slabs / transposed tiles are copies
    default: break;
    }
    if (!unused) return SPMV_HIP_OK;
    quiesce(d);
    (void) pool_free(d->colidx);
    d->colidx = nullptr;
    d->device_bytes -= (long long) (sizeof(int) * ((size_t) d->nnz + kStreamPad));
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_build(spmv_dev *d, const spmv_plan *plan)
{
    if (!d || !plan) return fail(SPMV_HIP_E_ARG, "build: NULL");
    if (!d->colidx && d->nnz > 0) return fail(SPMV_HIP_E_NOSTATE, "build: the column indices of this matrix were released after create (option keep_columns = 1 keeps them)");
    if (plan->sched < 0 || plan->sched >= SPMV_SCHED_COUNT) return fail(SPMV_HIP_E_ARG, "unknown schedule %d", plan->sched);
    if (d->sp_near || d->sp_far) { // a rebuild starts from the unsplit matrix
        if (d->sp_near) spmv_shim_matrix_destroy(d->sp_near);
        if (d->sp_far) spmv_shim_matrix_destroy(d->sp_far);
        d->sp_near = d->sp_far = nullptr;
    }
    free_schedule(d);
    d->plan = *plan;
    d->route_ms[0] = d->route_ms[1] = 0;
    d->split_ms[0] = d->split_ms[1] = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = SPMV_HIP_OK, staged = -1; // staged: tile groups with x windows in LDS (-1: schedule without windows)
    const bool f64 = d->vsize == sizeof(double);
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    int groups = 0;                    // ... out of this many
    // Decided before anything is built: the blocked executor multiplies whatever the tile schedule would look like (option
    // cache_block = 2), or a sample of the columns shows that no tile group could stage its x windows -- then the tile schedule's
    // inspector is skipped altogether (round 2 built windows / CSR5 transposes / SELL slabs and dropped them: 40-80 ms for 3e8 nnz).
    bool tiles = true;
    if (plan->sched != SPMV_SCHED_CSR_SCALAR && d->nnz > 0 && plan->x_windows &&
        ((plan->cache_block == 2 && blocked_possible(d)) || (plan->cache_block == 1 && blocked_size_ok(d) && !(plan->sched == SPMV_SCHED_SELL && !plan->sell_lds_x) && sample_says_no_locality(d))))
        tiles = false;
    auto build_tiles = [&]() -> int { // the inspector of the method's own tile schedule
        int rc = SPMV_HIP_OK;
        switch (plan->sched) {
        case SPMV_SCHED_CSR_SCALAR: break;
        case SPMV_SCHED_CSR_VECTOR: {
            const int L = plan->lanes_per_row;
            if (L < 1 || L > 64 || (L & (L - 1))) return fail(SPMV_HIP_E_ARG, "lanes_per_row must be a power of two in [1, 64], got %d", L);
            // a lane group takes 4L elements per step; rows longer than the planner's threshold (default: ~64
            // steps) are handed to the long-row path
            const int thr = plan->long_thr > 0 ? plan->long_thr : (L * 64 > 256 ? L * 64 : 256);
            rc = f64 ? build_long_rows<double>(d, thr) : build_long_rows<float>(d, thr);
            if (!rc) rc = f64 ? build_vector_tiles<double>(d) : build_vector_tiles<float>(d);
            staged = d->vt_staged;
            groups = d->vt_tiles;
            if (!rc && plan->autotune && blocked_mode(d, staged, groups) != 1) {
                if (d->vt_wide) rc = f64 ? autotune_rows<double>(d, nullptr) : autotune_rows<float>(d, nullptr); // wide x windows: the rows kernel over uniform blocks
                else rc = f64 ? autotune_vector<double>(d) : autotune_vector<float>(d);
            }
            break;
        }
        case SPMV_SCHED_NNZ_SPLIT:
            // equal-nnz tiles over the matrix's own arrays: CSR5 descriptors + carry fix-up, natural layout
            rc = f64 ? build_csr5<double>(d, d->ns, d->m, d->nnz, d->rowptr, d->colidx, (const double *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr, true)
                     : build_csr5<float>(d, d->ns, d->m, d->nnz, d->rowptr, d->colidx, (const float *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr, true);
            staged = d->ns.staged;
            groups = d->ns.groups;
            break;
        case SPMV_SCHED_ROWBLOCK:
            if (plan->rowblock_nnz < 64) return fail(SPMV_HIP_E_ARG, "rowblock_nnz must be >= 64");
            if (d->stats.max_row_len > plan->rowblock_nnz) return fail(SPMV_HIP_E_ARG, "row-block schedule needs max_row_len <= rowblock_nnz");
            rc = build_rowblock(d);
            if (!rc) rc = f64 ? build_rowblock_tiles<double>(d) : build_rowblock_tiles<float>(d);
            staged = d->vt_staged;
            groups = d->vt_tiles;
            if (!rc && plan->autotune && blocked_mode(d, staged, groups) != 1) rc = f64 ? autotune_rows<double>(d, d->rb_split) : autotune_rows<float>(d, d->rb_split);
            break;
        case SPMV_SCHED_SELL:
            rc = f64 ? build_sell<double>(d) : build_sell<float>(d);
            staged = d->plan.sell_lds_x ? d->sell_staged : -1; // sell_lds_x = 0: the caller asked for the plain slab kernel
            groups = d->sell_nwin;
            break;
        case SPMV_SCHED_CSR5:
            rc = f64 ? build_csr5<double>(d, d->c5, d->m, d->nnz, d->rowptr, d->colidx, (const double *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr)
                     : build_csr5<float>(d, d->c5, d->m, d->nnz, d->rowptr, d->colidx, (const float *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr);
            staged = d->c5.staged;
            groups = d->c5.groups;
            break;
        }
        return rc;
    };
    if (tiles) rc = build_tiles();
    // Columns without locality (tile groups whose x windows do not fit LDS) and x far larger than an L2: every gather of such a
    // group crosses the fabric -> row blocks x column slabs (kernels/blocked.hpp) take over the multiply, whatever the method
    // (CSR-scalar, the debug kernel, excepted).  When only PART of the groups stage, both executors exist for a moment and create()
    // keeps the one that multiplies faster (the share of unstaged groups at which the blocked executor wins depends on what the
    // staged part looks like: 2-15 %).  The loser's products are released.
    auto blk_best = [](const BlkSet &b) { return b.tune_ms[1] > 0 && b.tune_ms[1] < b.tune_ms[0] ? b.tune_ms[1] : b.tune_ms[0]; };
    auto blk_release = [&](const BlkSet &b) {
        quiesce(d);
        for (void *p : {(void *) b.row0, (void *) b.gstart, (void *) b.dir, (void *) b.order, b.val, (void *) b.meta, (void *) b.hdr}) if (p) sched_free(d, p);
    };
    auto blk_build = [&](int rule, int waves, bool ordered) { return f64 ? build_blocked<double>(d, rule, waves, ordered) : build_blocked<float>(d, rule, waves, ordered); };
    auto blk_tune = [&]() { return f64 ? autotune_blocked<double>(d) : autotune_blocked<float>(d); };
    // another layout next to the one that stands: built, timed, kept when it multiplies at least 3 % faster -- otherwise (or when it cannot be
    // built: memory) released again
    auto blk_try = [&](int rule, int waves, bool ordered) {
        const BlkSet first = d->blk;
        const float best0 = blk_best(first);
        d->blk_on = false;
        d->blk = BlkSet();
        const int rc2 = blk_build(rule, waves, ordered);
        if (!rc2 && d->blk_on) {
            (void) blk_tune();
            const float best1 = blk_best(d->blk);
            if (best1 > 0 && best1 < 0.97f * best0) blk_release(first);
            else { blk_release(d->blk); d->blk = first; }
        } else {
            blk_release(d->blk);
            d->blk = first;
            (void) hipGetLastError();
        }
        d->blk_on = true;
    };
    const int mode = !tiles ? 1 : (!rc && staged >= 0 ? blocked_mode(d, staged, groups) : 0);
    if (!rc && mode) {
        const size_t keep_from = d->sched_allocs.size();
        d->x_groups_seen = groups;
        double tile_ms = -1.0;
        if (mode == 2) tile_ms = f64 ? time_schedule<double>(d, 5) : time_schedule<float>(d, 5);
        // The blocked executor streams vsize + 4 bytes per entry and has never moved them faster than 5.0 TB/s next to its gathers: a tile schedule that is already
        // faster than 5.5 TB/s of THOSE bytes cannot lose to it -- neither executor is built nor timed (27-point stencil with periodic boundaries, 1 % of the tile
        // groups unstaged: create 33 -> 17 ms)
        const bool tiles_win_anyway = mode == 2 && tile_ms > 0 && plan->cache_block == 1 && tile_ms * 1e-3 * 5.5e12 <= (double) d->nnz * ((double) d->vsize + 4.0);
        if (tiles_win_anyway) d->route_ms[0] = (float) tile_ms;
        // forced width (option blk_waves) or the one-wave form first; the wide forms are tried against it below.  Where the one-wave form would surely be gather-bound
        // -- under one entry per 128-byte line of x and block if the columns were uniform (config 2-ii: 0.5) -- the wide form is built FIRST and the one-wave form only
        // if the measurement contradicts the estimate (config 2-ii: create 66 -> 40 ms; nothing changes for the others)
        const int forced = plan->blk_waves;
        const bool can_try = plan->autotune && !plan->forced && plan->blk_groups == 0 && plan->block_rows == 0 && forced == 0 && d->nnz >= (1ll << 22);
        const double rows_one = std::min(9982.0, std::max(1.0, (double) d->m / (2.0 * (double) d->cus)));
        const double lambda_one = rows_one * ((double) d->nnz / (double) d->m) * (128.0 / (double) d->vsize) / (double) d->n;
        const bool wide_first = can_try && plan->deterministic != 0 && (long long) d->m >= 2048ll * d->cus && lambda_one < 1.0;
        int rcb = tiles_win_anyway ? SPMV_HIP_OK : blk_build(0, forced > 0 ? forced : (wide_first ? 2 : 1), plan->deterministic != 0);
        if (rcb && wide_first) { // the wide layout could not be built: the one-wave form before anything is given up
            blk_release(d->blk);
            d->blk = BlkSet();
            d->blk_on = false;
            (void) hipGetLastError();
            rcb = blk_build(0, 1, true);
        }
        if (!rcb && d->blk_on) rcb = blk_tune();
        if (rcb) {
            // The blocked executor could not be built (device memory, or padded positions beyond 32 bits).  It is an alternative, not a
            // requirement: what it allocated goes back, the error is cleared and the tile schedule multiplies -- the one already built
            // and timed (mode 2), or, when the sample had skipped its inspector, the one built now.
            blk_release(d->blk);
            d->blk = BlkSet();
            d->blk_on = false;
            (void) hipGetLastError();
            if (plan->cache_block == 2 || rcb == SPMV_HIP_E_ARG) rc = rcb; // asked for explicitly (or a bad option): report
            else if (!tiles) { rc = build_tiles(); tiles = true; }
            if (mode == 2) d->route_ms[0] = (float) tile_ms;
        }
        if (!rc && d->blk_on && mode == 2) {
            const double blk_ms = blk_best(d->blk);
            d->route_ms[0] = (float) tile_ms;
            d->route_ms[1] = (float) blk_ms;
            if (tile_ms > 0 && (blk_ms <= 0 || tile_ms <= blk_ms)) { // the tile schedule stays
                blk_release(d->blk);
                d->blk = BlkSet();
                d->blk_on = false;
            }
        }
        if (!rc && d->blk_on) drop_tile_schedule(d, keep_from);
        const bool may_try = !rc && d->blk_on && can_try;
        auto stream_rate = [&]() { return may_try && blk_best(d->blk) > 0 ? (double) d->nnz * ((double) d->vsize + 4.0) / ((double) blk_best(d->blk) * 1e-3) : 0.0; }; // the streams' bytes per second
        double rate = stream_rate();
        if (may_try && wide_first && d->blk.waves > 1 && rate >= 4.2e12) { // not gather-bound after all: the one-wave form against it
            blk_try(0, 1, true);
            rate = stream_rate();
        }
        const bool one_wave = d->blk.waves <= 1;
        // Stream-bound under rule 0 (the streams alone move >= 3.6 TB/s; web-like 4e6 x 24 in fp32 sits at 4.2)?  Then thinner blocks may do better: build the
        // rule-1 set next to this one, time it, keep the faster (one more inspector pass, only for such matrices).
        if (may_try && one_wave && blocked_differs(d) && rate >= 3.6e12) blk_try(1, 1, true);
        // The wide forms: ONE block of up to ~20 k rows per CU, its accumulators shared by the waves of a workgroup -- fewer cache lines of x per
        // entry (kernels/blocked.hpp).  Reproducible results required (default): two waves taking turns at adding, tried where the one-wave
        // form is gather-bound (config 2-ii: 1.40 -> 1.19 ms; stream-bound shapes lose 3-5 % to the turns and are not tried).  Option
        // deterministic = 0: four waves adding as their products arrive (config 2-ii 1.22, Orkut-style R-MAT 0.55 -> 0.49, uniform
        // 0.66 -> 0.54, web-like 4e6 x 24 0.243 -> 0.216).
        if (may_try && one_wave && (long long) d->m >= 2048ll * d->cus) {
            if (plan->deterministic == 0) blk_try(0, 4, false);
            else if (rate < 4.2e12 && !wide_first) blk_try(0, 2, true);
        }
    }
    if (!rc) rc = account_stream_bytes(d);
    if (rc) { free_schedule(d); return rc; }
    d->inspect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    d->built = true;
    return SPMV_HIP_OK;
}

// ------------------------------------------------------------------------------------ values changed in place
template <typename T>
static int refresh_csr5_values(spmv_dev *d, Csr5Plan &P, const T *src)
{
    if (P.nnz == 0 || P.natural) return SPMV_HIP_OK; // natural layout reads the matrix's own value array
    const int grid = grid_for(P.tiles, kBlock / kWave, INT_MAX);
    switch (P.sigma) {
    case 4: csr5_transpose_kernel<T, 4><<<grid, kBlock, 0, d->stream>>>((int) P.nnz, P.tiles, nullptr, src, nullptr, (T *) P.val); break;
    case 8: csr5_transpose_kernel<T, 8><<<grid, kBlock, 0, d->stream>>>((int) P.nnz, P.tiles, nullptr, src, nullptr, (T *) P.val); break;
    default: csr5_transpose_kernel<T, 16><<<grid, kBlock, 0, d->stream>>>((int) P.nnz, P.tiles, nullptr, src, nullptr, (T *) P.val); break;
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

template <typename T>
static int update_values(spmv_dev *d, const void *val)
{
    if (d->nnz == 0) return SPMV_HIP_OK;
    if (val != d->val) HIP_TRY(hipMemcpyAsync(d->val, val, sizeof(T) * (size_t) d->nnz, hipMemcpyDefault, d->stream)); // the halves of a split handle refresh in place
    const T *v = (const T *) d->val;
    int rc = SPMV_HIP_OK;
    if (d->blk_on) { // the blocked streams are the only copy the executor reads
        rc = blocked_fill<T>(d, true);
    } else {
        if (d->nlong > 0 && d->lsub_val) { // long-row sub-matrix, then its CSR5 tiles
            long_rows_gather_kernel<T><<<d->nlong, kBlock, 0, d->stream>>>(d->long_rows, d->rowptr, nullptr, v, d->lsub_rowptr, nullptr, (T *) d->lsub_val);
            HIP_TRY(hipGetLastError());
            rc = refresh_csr5_values<T>(d, d->c5_long, (const T *) d->lsub_val);
        }
        if (!rc && d->plan.sched == SPMV_SCHED_SELL && d->nchunks > 0) {
            sell_fill_kernel<T><<<grid_for(d->nchunks, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>(
                d->nchunks, d->rowptr, nullptr, v, d->perm, d->chunk_ptr, nullptr, (T *) d->sval);
            HIP_TRY(hipGetLastError());
        }
        if (!rc && d->plan.sched == SPMV_SCHED_CSR5) rc = refresh_csr5_values<T>(d, d->c5, v);
        // CSR-scalar, CSR-vector, Balanced and nnz-split read d->val itself
    }
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream)); // the caller may overwrite `val` again at once
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_update_values(spmv_dev *d, const void *val)
{
    if (!d || !d->built) return fail(SPMV_HIP_E_NOSTATE, "update_values: schedule not built");
    if (!val && d->nnz > 0) return fail(SPMV_HIP_E_ARG, "update_values: NULL");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    if (d->sp_near && d->sp_far) { // split handle: the resident values first, then both halves from them (same positions as at create)
        d->sp_near->stream = d->sp_far->stream = d->stream;
        HIP_TRY(hipMemcpyAsync(d->val, val, d->vsize * (size_t) d->nnz, hipMemcpyDefault, d->stream));
        int rc = d->vsize == sizeof(double) ? split_make<double>(d, nullptr, nullptr, true) : split_make<float>(d, nullptr, nullptr, true);
        if (!rc) rc = spmv_shim_update_values(d->sp_near, d->sp_near->val);
        if (!rc) rc = spmv_shim_update_values(d->sp_far, d->sp_far->val);
        return rc;
    }
    return d->vsize == sizeof(double) ? update_values<double>(d, val) : update_values<float>(d, val);
}

// Checksum of the value array (option "check_values"): the sum over the 32-bit words of word * (odd multiplier of its
// position) modulo 2^64.  Position-weighted, so swapped or permuted values and changes whose plain word sums cancel are
// seen too; still a commutative sum, so host loop and device reduction agree whatever their order.
__host__ __device__ static inline unsigned long long checksum_term(unsigned w, long long i)
{
    return ((unsigned long long) w + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long) i + 1ull);
}

static __global__ __launch_bounds__(kBlock) void checksum_kernel(long long words, const unsigned *__restrict__ w, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < words; i += stride) s += checksum_term(w[i], i);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(out, s);
}

extern "C" int spmv_shim_checksum(spmv_dev *d, const void *val, unsigned long long *out)
{
    if (!d || !out) return fail(SPMV_HIP_E_ARG, "checksum: NULL");
    *out = 0;
    const long long words = d->nnz * (long long) (d->vsize / 4);
    if (words == 0) return SPMV_HIP_OK;
    if (!val) return fail(SPMV_HIP_E_ARG, "checksum: NULL values");
    if (!is_device_ptr(val)) { // host array: summed where it lives
        const unsigned *w = (const unsigned *) val;
        unsigned long long s = 0;
        for (long long i = 0; i < words; ++i) s += checksum_term(w[i], i);
        *out = s;
        return SPMV_HIP_OK;
    }
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    if (!d->scratch8) ALLOC_TRY(d, &d->scratch8, 8, false);
    HIP_TRY(hipMemsetAsync(d->scratch8, 0, 8, d->stream));
    checksum_kernel<<<grid_for(words, kBlock * 8, d->cus * 8), kBlock, 0, d->stream>>>(words, (const unsigned *) val, (unsigned long long *) d->scratch8);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d->scratch8, 8, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_checksum_words(const void *val, long long words, unsigned long long *out)
{
    if (!out) return fail(SPMV_HIP_E_ARG, "checksum: NULL");
    *out = 0;
    if (words <= 0) return SPMV_HIP_OK;
    if (!val) return fail(SPMV_HIP_E_ARG, "checksum: NULL values");
    if (!is_device_ptr(val)) {
        const unsigned *w = (const unsigned *) val;
        unsigned long long s = 0;
        for (long long i = 0; i < words; ++i) s += checksum_term(w[i], i);
        *out = s;
        return SPMV_HIP_OK;
    }
    hipPointerAttribute_t attr;
    int dev = 0;
    if (hipPointerGetAttributes(&attr, val) == hipSuccess) dev = attr.device; else (void) hipGetLastError();
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", dev);
    unsigned long long *acc = nullptr;
    HIP_TRY(pool_malloc((void **) &acc, 8));
    hipError_t e = hipMemset(acc, 0, 8);
    if (e == hipSuccess) {
        checksum_kernel<<<grid_for(words, kBlock * 8, 2048), kBlock>>>(words, (const unsigned *) val, acc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, acc, 8, hipMemcpyDeviceToHost);
    (void) pool_free(acc);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "checksum: %s", hipGetErrorString(e));
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_is_device_ptr(const void *p) { return is_device_ptr(p) ? 1 : 0; }

#include "shim/launch.hpp"
#include "shim/split.hpp"

extern "C" int spmv_shim_run(spmv_dev *d, const void *x, void *y)
{
    if (!d || !d->built) return fail(SPMV_HIP_E_NOSTATE, "run: schedule not built");
    if ((d->n > 0 && d->nnz > 0 && !x) || (d->m > 0 && !y)) return fail(SPMV_HIP_E_ARG, "run: X or Y is NULL");
    DeviceGuard guard(d->device); // the caller's current device is restored on return
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    const bool xdev = is_device_ptr(x), ydev = is_device_ptr(y);
    const void *xd = x;
    void *yd = y;
    if (!xdev && d->n > 0 && x) { // host x: stage through HBM (correct, PCIe-bound)
        if (!d->x_stage) ALLOC_TRY(d, &d->x_stage, d->vsize * (size_t) d->n, false);
        HIP_TRY(hipMemcpyAsync(d->x_stage, x, d->vsize * (size_t) d->n, hipMemcpyHostToDevice, d->stream));
        xd = d->x_stage;
    }
    if (!ydev && d->m > 0) {
        if (!d->y_stage) ALLOC_TRY(d, &d->y_stage, d->vsize * (size_t) d->m, false);
        yd = d->y_stage;
    }
    int rc = d->vsize == sizeof(double) ? launch<double>(d, (const double *) xd, (double *) yd)
                                        : launch<float>(d, (const float *) xd, (float *) yd);
    if (rc) return rc;
    if (!ydev && d->m > 0) HIP_TRY(hipMemcpyAsync(y, d->y_stage, d->vsize * (size_t) d->m, hipMemcpyDeviceToHost, d->stream));
    if (!d->async || !xdev || !ydev) HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_set_stream(spmv_dev *d, void *stream)
{
    if (!d) return fail(SPMV_HIP_E_ARG, "set_stream: NULL");
    d->stream = (hipStream_t) stream;
    // the halves of a split handle launch on the parent's stream; a values refresh between set_stream and the next multiply must already see it
    if (d->sp_near) d->sp_near->stream = d->stream;
    if (d->sp_far) d->sp_far->stream = d->stream;
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_set_async(spmv_dev *d, int async)
{
    if (!d) return fail(SPMV_HIP_E_ARG, "set_async: NULL");
    d->async = async;
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_sync(spmv_dev *d)
{
    if (!d) return fail(SPMV_HIP_E_ARG, "sync: NULL");
    HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}

extern "C" double spmv_shim_time(spmv_dev *d, const void *x, void *y, int warmup, int iters, float *ms_out)
{
    if (!d || !d->built || iters <= 0) { fail(SPMV_HIP_E_ARG, "time: bad arguments"); return -1.0; }
    if (!is_device_ptr(x) || !is_device_ptr(y)) { fail(SPMV_HIP_E_ARG, "time: x and y must be device pointers"); return -1.0; }
    const int keep_async = d->async;
    d->async = 1;
    std::vector<hipEvent_t> ev((size_t) iters + 1);
    for (auto &e : ev) if (hipEventCreate(&e) != hipSuccess) { d->async = keep_async; fail(SPMV_HIP_E_RUNTIME, "hipEventCreate"); return -1.0; }
    int rc = SPMV_HIP_OK;
    for (int i = 0; i < warmup && !rc; ++i) rc = spmv_shim_run(d, x, y);
    for (int i = 0; i < iters && !rc; ++i) {
        (void) hipEventRecord(ev[i], d->stream);
        rc = spmv_shim_run(d, x, y);
    }
    (void) hipEventRecord(ev[iters], d->stream);
    hipError_t e = hipStreamSynchronize(d->stream);
    d->async = keep_async;
    double mean = -1.0;
    if (!rc && e == hipSuccess) {
        double tot = 0;
        for (int i = 0; i < iters; ++i) {
            float ms = 0;
            (void) hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            if (ms_out) ms_out[i] = ms;
            tot += ms;
        }
        mean = tot / iters;
    } else if (e != hipSuccess) {
        fail(SPMV_HIP_E_RUNTIME, "time: %s", hipGetErrorString(e));
    }
    for (auto &v : ev) (void) hipEventDestroy(v);
    return mean;
}

// min over `iters` launches of the built schedule on scratch vectors (x = 1): what the measured automatic
// choice (spmv_api.c, auto_method = 2) compares.  Returns ms, < 0 on failure.
extern "C" double spmv_shim_time_self(spmv_dev *d, int iters)
{
    if (!d || !d->built || iters <= 0 || iters > 64) { fail(SPMV_HIP_E_ARG, "time_self: bad arguments"); return -1.0; }
    void *x = nullptr, *y = nullptr;
    if (pool_malloc(&x, d->vsize * (size_t) (d->n > 0 ? d->n : 1)) != hipSuccess || pool_malloc(&y, d->vsize * (size_t) (d->m > 0 ? d->m : 1)) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) pool_free(x);
        fail(SPMV_HIP_E_ALLOC, "time_self: scratch vectors");
        return -1.0;
    }
    if (d->vsize == sizeof(double)) fill_value_kernel<double><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, (double *) x, 1.0);
    else fill_value_kernel<float><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, (float *) x, 1.0f);
    float ms[64];
    const double mean = spmv_shim_time(d, x, y, 2, iters, ms);
    (void) pool_free(x);
    (void) pool_free(y);
    if (mean < 0) return -1.0;
    float best = ms[0];
    for (int i = 1; i < iters; ++i) best = ms[i] < best ? ms[i] : best;
    return (double) best;
}

// ------------------------------------------------------------------------------------ info
static const char *kSchedNames[] = {"csr-scalar", "csr-vector", "row-block", "nnz-split", "sell-c-sigma", "csr5"};
static const char *kKernelNames[] = {"csr_scalar_kernel", "csr_vector_pipe_kernel", "csr_vector_rows_kernel",
                                     "nat_kernel", "sell_kernel", "csr5_kernel"};

// kernels of one CSR5 / nnz-split plan's multiply, appended to `buf` ('+'-separated, in launch order)
static void append_name(char *buf, size_t cap, const char *name)
{
    const size_t n = strlen(buf);
    if (n + strlen(name) + 2 >= cap) return;
    if (n) strcat(buf, "+");
    strcat(buf, name);
}

static void csr5_kernel_names(const spmv_dev *d, const Csr5Plan &P, char *buf, size_t cap)
{
    if (P.nnz == 0) return;
    if (P.staged > 0) append_name(buf, cap, P.natural ? "nat_group_kernel" : (csr5_two_deep(d, P) ? "csr5_group_pipe_kernel" : "csr5_group_kernel"));
    else append_name(buf, cap, P.natural ? "nat_kernel" : "csr5_kernel");
    if (P.fixup && P.tiles > 1 && !P.forward) append_name(buf, cap, "csr5_fixup_kernel");
}

extern "C" int spmv_shim_info(const spmv_dev *d, spmv_hip_info *o)
{
    if (!d || !o) return fail(SPMV_HIP_E_ARG, "info: NULL");
    if (d->sp_near && d->sp_far) { // split handle: the near half names schedule and kernel; sizes and bytes are the whole matrix's
        const int rc = spmv_shim_info(d->sp_near, o);
        if (rc) return rc;
        spmv_hip_info f;
        (void) spmv_shim_info(d->sp_far, &f);
        const long long s = (long long) d->vsize;
        o->nnz = d->nnz;
        o->stored_nnz += f.stored_nnz;
        o->max_row_len = d->stats.max_row_len; o->min_row_len = d->stats.min_row_len; o->empty_rows = d->stats.empty_rows; o->mean_row_len = d->stats.mean_row_len;
        o->device_bytes = d->device_bytes + d->sp_near->device_bytes + d->sp_far->device_bytes;
        o->alg_bytes = 4ll * ((long long) d->m + 1) + d->nnz * (4 + s) + s * d->n + s * d->m;
        o->stream_bytes += f.stream_bytes;
        o->x_bytes += f.x_bytes;
        o->inspect_ms = d->inspect_ms;
        o->route_ms[0] = d->route_ms[0]; o->route_ms[1] = d->route_ms[1];
        o->split_ms[0] = d->split_ms[0]; o->split_ms[1] = d->split_ms[1];
        o->far_nnz = d->sp_far->nnz;
        o->blk_waves = f.blk_waves;
        o->reproducible = o->reproducible && f.reproducible;
        append_name(o->launch_kernels, sizeof o->launch_kernels, f.launch_kernels);
        return SPMV_HIP_OK;
    }
    memset(o, 0, sizeof *o);
    o->device = d->device;
    o->schedule = d->plan.sched;
    o->lanes_per_row = d->plan.sched == SPMV_SCHED_CSR_VECTOR ? d->plan.lanes_per_row : 0;
    o->sell_c = d->plan.sched == SPMV_SCHED_SELL ? kSellC : 0;
    o->sell_sigma = d->plan.sched == SPMV_SCHED_SELL ? d->plan.sell_sigma : 0;
    o->tile_nnz = d->plan.sched == SPMV_SCHED_NNZ_SPLIT ? kWave * d->ns.sigma : (d->plan.sched == SPMV_SCHED_ROWBLOCK ? d->rb_stride : (d->plan.sched == SPMV_SCHED_CSR5 ? kWave * d->c5.sigma : 0));
    o->m = d->m;
    o->n = d->n;
    o->nnz = d->nnz;
    o->stored_nnz = d->plan.sched == SPMV_SCHED_SELL ? d->sell_cols * kSellC : (d->plan.sched == SPMV_SCHED_CSR5 ? (long long) d->c5.tiles * kWave * d->c5.sigma : d->nnz);
    o->max_row_len = d->stats.max_row_len;
    o->min_row_len = d->stats.min_row_len;
    o->empty_rows = d->stats.empty_rows;
    o->mean_row_len = d->stats.mean_row_len;
    o->device_bytes = d->device_bytes;
    const long long s = (long long) d->vsize;
    o->alg_bytes = 4ll * ((long long) d->m + 1) + d->nnz * (4 + s) + s * d->n + s * d->m;
    o->inspect_ms = d->inspect_ms;
    o->tuned_choice = d->blk_on ? 100 + d->blk.form : d->vec_choice; // cache_blocked: 100 / 101 = the smaller / larger form of blocked_forms()
    if (!d->blk_on && d->rows_depth && (d->plan.sched == SPMV_SCHED_ROWBLOCK || (d->plan.sched == SPMV_SCHED_CSR_VECTOR && d->vt_wide))) o->tuned_choice = d->rows_depth == 2 ? VEC_TILE_D2 : VEC_TILE_D4; // the rows kernel's depth, in the tile forms' codes
    for (int k = 0; k < 3; ++k) o->tune_ms[k] = d->blk_on ? d->blk.tune_ms[k] : d->tune_ms[k];
    o->schedule_name = kSchedNames[d->plan.sched];
    o->kernel_name = kKernelNames[d->plan.sched];
    if (d->plan.sched == SPMV_SCHED_CSR_VECTOR && d->vt_tiles > 0 && d->vec_choice != VEC_PIPE &&
        (d->vt_staged * 2 >= d->vt_tiles || (d->vec_choice != VEC_AUTO && d->vec_choice != VEC_PIPE)))
        o->kernel_name = "csr_vector_tile_kernel";
    if (d->plan.sched == SPMV_SCHED_CSR_VECTOR && d->vt_wide) o->kernel_name = "csr_vector_rows_kernel";
    o->cache_blocked = d->blk_on ? 1 : 0;
    o->stream_bytes = d->stream_bytes;
    o->x_bytes = d->x_bytes;
    o->route_ms[0] = d->route_ms[0];
    o->route_ms[1] = d->route_ms[1];
    o->split_ms[0] = d->split_ms[0];
    o->split_ms[1] = d->split_ms[1];
    o->far_nnz = d->sp_far ? d->sp_far->nnz : 0;
    {
        const bool tiles_run = !d->blk_on && ((d->plan.sched == SPMV_SCHED_CSR_VECTOR && d->vt_tiles > 0 && (d->vt_wide || d->vt_staged * 2 >= d->vt_tiles) && d->vec_choice != VEC_PIPE) || d->plan.sched == SPMV_SCHED_ROWBLOCK);
        o->run_nnz = tiles_run ? d->vt_run_nnz : 0;
        o->byte_nnz = tiles_run ? d->vt_byte_nnz : 0;
        o->tmpl_nnz = tiles_run ? d->vt_tmpl_nnz : 0;
        if (!d->blk_on && d->plan.sched == SPMV_SCHED_SELL && d->sell_staged > 0) { o->run_nnz = d->sell_run_nnz - d->sell_tmpl_nnz; o->tmpl_nnz = d->sell_tmpl_nnz; o->byte_nnz = d->sell_byte_nnz; }
        if (!d->blk_on && d->plan.sched == SPMV_SCHED_CSR5) o->run_nnz = d->c5.run_tiles * kWave * d->c5.sigma;
    }
    if (d->blk_on) { o->stored_nnz = d->blk.groups << d->blk.ge; o->x_groups = d->x_groups_seen; o->x_groups_staged = 0; }
    if (!d->blk_on) switch (d->plan.sched) {
    case SPMV_SCHED_CSR_VECTOR:
    case SPMV_SCHED_ROWBLOCK: o->x_groups = d->vt_tiles; o->x_groups_staged = d->vt_staged; break;
    case SPMV_SCHED_NNZ_SPLIT: o->x_groups = d->ns.groups; o->x_groups_staged = d->ns.staged; break;
    case SPMV_SCHED_SELL: o->x_groups = d->sell_nwin; o->x_groups_staged = d->sell_staged; break;
    case SPMV_SCHED_CSR5: o->x_groups = d->c5.groups; o->x_groups_staged = d->c5.staged; break;
    default: o->x_groups = o->x_groups_staged = 0; break;
    }
    o->blk_waves = d->blk_on ? d->blk.waves : 0;
    o->reproducible = d->blk_on && d->blk.waves > 1 && !d->blk.ordered ? 0 : 1;
    if (d->blk_on) o->kernel_name = d->blk.waves > 1 ? "blk_wide_kernel" : "blk_kernel";
    else if (d->plan.sched == SPMV_SCHED_NNZ_SPLIT)
        o->kernel_name = d->ns.staged > 0 ? "nat_group_kernel" : "nat_kernel";
    if (d->plan.sched == SPMV_SCHED_CSR5 && d->c5.staged > 0) o->kernel_name = csr5_two_deep(d, d->c5) ? "csr5_group_pipe_kernel" : "csr5_group_kernel";
    if (d->plan.sched == SPMV_SCHED_SELL && d->sell_staged > 0) o->kernel_name = "sell_window_kernel";
    { // every kernel of one multiply, in launch order
        char *b = o->launch_kernels;
        const size_t cap = sizeof o->launch_kernels;
        b[0] = 0;
        if (d->nnz == 0) { if (!d->accumulate && d->m > 0) append_name(b, cap, "fill_zero_kernel"); }
        else if (d->blk_on) append_name(b, cap, o->kernel_name);
        else switch (d->plan.sched) {
        case SPMV_SCHED_NNZ_SPLIT: csr5_kernel_names(d, d->ns, b, cap); break;
        case SPMV_SCHED_CSR5: csr5_kernel_names(d, d->c5, b, cap); break;
        case SPMV_SCHED_CSR_SCALAR: append_name(b, cap, o->kernel_name); break;
        default: // CSR-vector, row blocks, SELL: the row-granular kernel, then the long rows' CSR5 sub-matrix
            append_name(b, cap, o->kernel_name);
            if (d->nlong > 0) csr5_kernel_names(d, d->c5_long, b, cap);
            break;
        }
    }
    return SPMV_HIP_OK;
}

#ifdef SPMV_BLK_DEBUG_FORMS
// tools only: per-workgroup timestamps of the last blk_kernel launch (4 x 8192 words)
extern "C" int spmv_shim_debug_blk_times(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(blk_dbg_times), sizeof(unsigned long long) * 4 * 8192) == hipSuccess ? 0 : 1;
}
#endif

#include "shim/reorder.hpp"
#include "shim/multi.hpp"
