// shim/inspect.hpp -- part of the single translation unit spmv_shim.hip: the device-side INSPECTORS, one
// build_* per schedule (x windows, CSR-vector tiles, long rows, row blocks, SELL, CSR5 / natural tiles,
// row blocks x column slabs).  They run once at create (spmv_shim_build) and only launch kernels from
// kernels/*.hpp; nothing here touches host copies of the matrix.
#pragma once

// ------------------------------------------------------------------------------------ inspectors
// x windows over contiguous ranges of a PRIVATE column array (xwindows.hpp).  Two passes: count the
// groups whose columns fit LDS; only if at least half do (or in_place_ok is false and any does...)
// rewrite the array into LDS slots.  Returns staged groups (0 = array untouched) and the LDS need.
static int build_range_windows(spmv_dev *d, int groups, long long total, long long group_len, const long long *bounds, int bstride,
                               int scale, int max_cols, int *cols, TileWindows *wins, int *staged_out, int *maxtotal_out,
                               unsigned short *cols16 = nullptr, int pack16 = 0, long long nbounds = 0)
{
    int *cnt = nullptr;
    int host2[2] = {0, 0};
    *staged_out = *maxtotal_out = 0;
    if (groups <= 0) return SPMV_HIP_OK;
    HIP_TRY(pool_malloc((void **) &cnt, 2 * sizeof(int)));
    hipError_t e = hipMemsetAsync(cnt, 0, 2 * sizeof(int), d->stream);
    range_windows_kernel<<<groups, kBlock, 0, d->stream>>>(total, group_len, bounds, bstride, scale, nbounds, d->n, max_cols, cols, cols16, pack16, wins, cnt, 0);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host2, cnt, sizeof host2, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess && host2[0] * 2 >= groups) { // worth it: rewrite the staged groups into LDS slots
        range_windows_kernel<<<groups, kBlock, 0, d->stream>>>(total, group_len, bounds, bstride, scale, nbounds, d->n, max_cols, cols, cols16, pack16, wins, cnt, 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        *staged_out = host2[0];
        *maxtotal_out = host2[1];
    }
    (void) pool_free(cnt);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "x-window inspector: %s", hipGetErrorString(e));
    return SPMV_HIP_OK;
}

constexpr size_t kVecXTileBytes = 48 * 1024; // LDS budget of one row tile's x span (CSR-vector, Balanced)
template <typename T> static int build_long_rows(spmv_dev *d, int thr);
template <typename T>
static int build_csr5(spmv_dev *d, Csr5Plan &P, int m, long long nnz, const int *rowptr, const int *colidx, const T *val, int empty_rows,
                      double mean_row_len, const int *out_rows, bool natural = false, int sigma_override = 0);
template <typename T> static int autotune_vector(spmv_dev *d);
template <typename T> static int autotune_rows(spmv_dev *d, const int *split);
template <typename T> static int autotune_blocked(spmv_dev *d);
template <typename T> static double time_schedule(spmv_dev *d, int iters);
template <typename T> static int split_make(spmv_dev *d, spmv_dev **near_out, spmv_dev **far_out, bool values_only);
template <typename T> static int build_tile_windows(spmv_dev *d, int tiles, const int *split, int rows_per_tile = kVecTileRows, bool wide = false);
static int wins_sum(spmv_dev *d, const TileWindows *wins, int count, long long *elems, long long *tiles);
constexpr size_t kVecWideXTileBytes = 96 * 1024; // budget of the wide form (slot indices; two workgroups per CU)

static int build_rowblock(spmv_dev *d)
{
    d->rb_stride = d->plan.rowblock_nnz;
    d->nblocks = (int) ((d->nnz + d->rb_stride - 1) / d->rb_stride);
    if (d->nblocks < 1) d->nblocks = 1;
    ALLOC_TRY(d, &d->rb_split, sizeof(int) * ((size_t) d->nblocks + 1), true);
    rowblock_split_kernel<<<grid_for((long long) d->nblocks + 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(
        d->m, (int) d->nnz, d->nblocks, d->rb_stride, d->rowptr, d->rb_split);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}

// Balanced executor = the CSR-vector wave program over the equal-nnz row blocks: long rows + x spans
template <typename T>
static int build_rowblock_tiles(spmv_dev *d)
{
    const int L = d->plan.lanes_per_row;
    int rc = build_long_rows<T>(d, L * 64 > 256 ? L * 64 : 256);
    if (rc) return rc;
    rc = build_tile_windows<T>(d, d->nblocks, d->rb_split);
    if (rc) return rc;
    if (d->vt_staged * 2 < d->vt_tiles && !d->plan.forced) { // wide windows: slot indices, 96 KiB budget (same blocks)
        rc = build_tile_windows<T>(d, d->nblocks, d->rb_split, kVecTileRows, true);
        if (rc) return rc;
        if (d->vt_staged * 2 < d->vt_tiles) rc = build_tile_windows<T>(d, d->nblocks, d->rb_split);
    }
    return rc;
}

// Windows of every row tile + the tile-local ColIdx copy (kernels/csr_vector_tile.hpp).
template <typename T>
static int build_tile_windows(spmv_dev *d, int tiles, const int *split, int rows_per_tile, bool wide)
{
    int *cnt = nullptr;
    int host2[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    d->vt_tiles = tiles;
    ALLOC_TRY(d, &cnt, sizeof host2, true);
    static_assert(kVecXTileBytes <= 65536, "LDS byte offsets must fit 16 bits");
    static_assert(kVecWideXTileBytes / sizeof(float) <= 65536, "LDS slot indices must fit 16 bits");
    if (!d->vt_col) {
        ALLOC_TRY(d, &d->vt_col, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), true);
        HIP_TRY(hipMemsetAsync(d->vt_col, 0, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), d->stream));
    }
    if (!d->vt_rowslot && d->plan.run_tiles) { // option run_tiles = 0 (A/B): every staged tile reads its 16-bit column stream
        ALLOC_TRY(d, &d->vt_rowslot, sizeof(unsigned short) * ((size_t) d->m + kStreamPad), true);
        HIP_TRY(hipMemsetAsync(d->vt_rowslot, 0, sizeof(unsigned short) * ((size_t) d->m + kStreamPad), d->stream));
    }
    if (!d->vt_col8 && d->vt_rowslot && !getenv("SPMV_HIP_NO_BYTE_TILES")) { // A/B switch of tools/: staged tiles that are not RUN tiles read their 16-bit stream
        ALLOC_TRY(d, &d->vt_col8, (size_t) d->nnz + kStreamPad, true);
        HIP_TRY(hipMemsetAsync(d->vt_col8, 0, (size_t) d->nnz + kStreamPad, d->stream));
    }
    if (d->vt_tmpl) { sched_free(d, d->vt_tmpl); d->vt_tmpl = nullptr; } // sized by the tile count, which differs between the narrow and the wide attempt
    if (d->vt_rowslot && !getenv("SPMV_HIP_NO_TEMPLATE_TILES")) {
        ALLOC_TRY(d, &d->vt_tmpl, sizeof(unsigned short) * kTmplCount * kTmplMax * (size_t) tiles, true);
        if (!d->vt_rowtid) {
            ALLOC_TRY(d, &d->vt_rowtid, (size_t) d->m + kStreamPad, true);
            HIP_TRY(hipMemsetAsync(d->vt_rowtid, 0, (size_t) d->m + kStreamPad, d->stream));
        }
    }
    if (d->vt_wins) sched_free(d, d->vt_wins);
    ALLOC_TRY(d, &d->vt_wins, sizeof(TileWindows) * (size_t) tiles, true);
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof host2, d->stream));
    // narrow form: the stream holds LDS byte offsets (budget 48 KiB); wide form: slot indices (budget 96 KiB)
    csr_tile_windows_kernel<<<tiles, kBlock, 0, d->stream>>>(d->m, d->n, rows_per_tile, d->long_thr,
                                                             (int) ((wide ? kVecWideXTileBytes : kVecXTileBytes) / sizeof(T)) - 1, wide ? 1 : (int) sizeof(T),
                                                             split, d->rowptr, d->colidx,
                                                             d->vt_wins, d->vt_col, d->vt_rowslot, d->vt_col8, d->vt_tmpl, d->vt_rowtid, cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host2, cnt, sizeof host2, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->vt_staged = host2[0];
    d->vt_maxspan = host2[1];
    d->vt_run_tiles = host2[2];
    d->vt_run_nnz = host2[3];
    d->vt_run_rows = host2[4];
    d->vt_byte_tiles = host2[5];
    d->vt_byte_nnz = host2[6];
    d->vt_byte_rows = host2[7];
    d->vt_tmpl_tiles = host2[8];
    d->vt_tmpl_nnz = host2[9];
    d->vt_tmpl_rows = host2[10];
    if (d->vt_tmpl && d->vt_tmpl_tiles == 0) { sched_free(d, d->vt_tmpl); d->vt_tmpl = nullptr; sched_free(d, d->vt_rowtid); d->vt_rowtid = nullptr; }
    if (d->vt_col8 && d->vt_byte_tiles == 0) { sched_free(d, d->vt_col8); d->vt_col8 = nullptr; } // no tile qualified: the byte stream is not kept
    d->vt_wide = wide;
    d->vt_rows = rows_per_tile;
    return SPMV_HIP_OK;
}

// LDS budget of one SELL window's x tile: 96 KiB of the CU's 160 KiB (one 512-thread workgroup per
// window; fp32 24576 columns, fp64 12288 columns).
constexpr size_t kSellXTileBytes = 96 * 1024;

// Rows longer than thr -> long_rows[] (row order), gathered into a sub-CSR with its own CSR5 plan
// (d->c5_long; kernels/long_rows.hpp).
template <typename T>
static int build_long_rows(spmv_dev *d, int thr)
{
    d->long_thr = thr;
    d->nlong = 0;
    d->c5_long = Csr5Plan();
    if (d->stats.max_row_len <= thr) return SPMV_HIP_OK;
    // deterministic compaction of the long rows (flags -> scan -> scatter), as csr5 does for non-empty rows
    const int nb = (int) (((long long) d->m + kScanTile - 1) / kScanTile);
    int *flags = nullptr, *sums = nullptr, *total = nullptr, *scratch = nullptr;
    ALLOC_TRY(d, &sums, sizeof(int) * (size_t) nb, true);
    ALLOC_TRY(d, &total, sizeof(int), true);
    HIP_TRY(pool_malloc((void **) &flags, sizeof(int) * (size_t) d->m));
    long_rows_flag_kernel<<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, thr, d->rowptr, flags);
    scan_block_sums_kernel<<<nb, kBlock, 0, d->stream>>>(d->m, flags, sums);
    scan_sums_inplace_kernel<<<1, kBlock, 0, d->stream>>>(nb, sums, total);
    if (hipMemcpyAsync(&d->nlong, total, sizeof(int), hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) {
        (void) pool_free(flags);
        d->nlong = 0;
        return fail(SPMV_HIP_E_RUNTIME, "long-row scan failed");
    }
    if (d->nlong == 0) { (void) pool_free(flags); return SPMV_HIP_OK; }
    int rc = dev_alloc(d, (void **) &d->long_rows, sizeof(int) * (size_t) d->nlong, true);
    if (!rc) rc = dev_alloc(d, (void **) &scratch, sizeof(int) * (size_t) d->nlong, true);
    if (rc) { (void) pool_free(flags); d->nlong = 0; return rc; }
    csr5_compact_kernel<<<nb, kBlock, 0, d->stream>>>(d->m, flags, sums, d->rowptr, scratch, d->long_rows);
    hipError_t e = hipStreamSynchronize(d->stream);
    (void) pool_free(flags);
    if (e != hipSuccess) { d->nlong = 0; return fail(SPMV_HIP_E_RUNTIME, "long-row compaction: %s", hipGetErrorString(e)); }
    ALLOC_TRY(d, &d->lr_seg_start, sizeof(long long) * ((size_t) d->nlong + 1), true);

    { // sub-CSR of the long rows + CSR5 over it
        long long sub_nnz = 0;
        long_rows_len_kernel<<<grid_for(d->nlong, kBlock, INT_MAX), kBlock, 0, d->stream>>>(d->nlong, d->long_rows, d->rowptr, scratch);
        scan_i32_to_i64_kernel<<<1, kBlock, 0, d->stream>>>(d->nlong, scratch, d->lr_seg_start);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&sub_nnz, d->lr_seg_start + d->nlong, sizeof(long long), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        d->lsub_nnz = sub_nnz;
        ALLOC_TRY(d, &d->lsub_rowptr, sizeof(int) * ((size_t) d->nlong + 1), true);
        ALLOC_TRY(d, &d->lsub_colidx, sizeof(int) * (size_t) sub_nnz, true);
        ALLOC_TRY(d, &d->lsub_val, sizeof(T) * (size_t) sub_nnz, true);
        narrow_i64_kernel<<<grid_for((long long) d->nlong + 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(d->nlong + 1, d->lr_seg_start, d->lsub_rowptr);
        long_rows_gather_kernel<T><<<d->nlong, kBlock, 0, d->stream>>>(d->long_rows, d->rowptr, d->colidx, (const T *) d->val, d->lsub_rowptr,
                                                                      d->lsub_colidx, (T *) d->lsub_val);
        HIP_TRY(hipGetLastError());
        return build_csr5<T>(d, d->c5_long, d->nlong, sub_nnz, d->lsub_rowptr, d->lsub_colidx, (const T *) d->lsub_val, 0,
                             (double) sub_nnz / (double) d->nlong, d->long_rows);
    }
}


// x windows of every 256-row tile (kernels/csr_vector_tile.hpp)
template <typename T>
static int build_vector_tiles(spmv_dev *d)
{
    d->vt_staged = d->vt_maxspan = 0;
    const int rows = kVecTileRows;
    d->vt_tiles = (int) (((long long) d->m + rows - 1) / rows);
    if (d->vt_tiles == 0 || d->nnz == 0) return SPMV_HIP_OK;
    int rc = build_tile_windows<T>(d, d->vt_tiles, nullptr, rows);
    if (rc || d->plan.forced) return rc;
    if (d->vt_staged * 2 >= d->vt_tiles) {
        // The narrow tiles stage -- but at what price?  Rows scattered +-4096 columns around the diagonal make every 256-row tile stage 8 448
        // columns for 256 rows' worth of entries: with 16 entries per row the windows are 1.4 x the bytes the tile streams (fp32), read from L2,
        // and the workgroup waits for them.  When the windows cost more than half of the stream, the same test is made for 1024-row blocks (wide
        // form: a quarter of the window traffic per entry) and they are kept if every block stages.  SPMV_HIP_NO_WIDE_BY_COST=1: off (A/B).
        long long welems = 0, wtiles = 0;
        const long long stream = (d->nnz - d->lsub_nnz) * ((long long) sizeof(T) + 2);
        if (getenv("SPMV_HIP_NO_WIDE_BY_COST") || wins_sum(d, d->vt_wins, d->vt_tiles, &welems, &wtiles) != SPMV_HIP_OK || stream <= 0 ||
            (double) welems * sizeof(T) <= 0.5 * (double) stream)
            return SPMV_HIP_OK;
        const int narrow_tiles = d->vt_tiles, w_rows = 4 * kVecTileRows;
        const int w_tiles = (int) (((long long) d->m + w_rows - 1) / w_rows);
        rc = build_tile_windows<T>(d, w_tiles, nullptr, w_rows, true);
        if (rc) return rc;
        long long welems_w = 0;
        if (d->vt_staged == d->vt_tiles && wins_sum(d, d->vt_wins, d->vt_tiles, &welems_w, &wtiles) == SPMV_HIP_OK && welems_w * 2 <= welems) return SPMV_HIP_OK;
        return build_tile_windows<T>(d, narrow_tiles, nullptr, rows); // no: back to the narrow tiles
    }
    // fewer than half of the 256-row tiles fit the 48 KiB budget.  Before settling for global gathers (pipe
    // form), try the WIDE form: 1024-row blocks walked by the rows kernel, 96 KiB budget, slot indices in the
    // stream (windows above 64 KiB: fp64 rows scattered over +-4096 columns, config 4 in fp64: 2.04 -> 1.52 ms)
    const int wide_rows = 4 * kVecTileRows;
    const int wide_tiles = (int) (((long long) d->m + wide_rows - 1) / wide_rows);
    rc = build_tile_windows<T>(d, wide_tiles, nullptr, wide_rows, true);
    if (rc) return rc;
    if (d->vt_staged * 2 >= d->vt_tiles) return SPMV_HIP_OK;
    return build_tile_windows<T>(d, (int) (((long long) d->m + rows - 1) / rows), nullptr, rows); // no: back to the narrow tiles (unstaged -> pipe form)
}

template <typename T> static int launch_csr5(spmv_dev *d, const Csr5Plan &P, const T *x, T *y);

template <typename T>
static void launch_long_rows(spmv_dev *d, const T *x, T *y)
{
    if (d->nlong > 0 && d->c5_long.nnz > 0) (void) launch_csr5<T>(d, d->c5_long, x, y);
}

template <typename T>
static int build_sell(spmv_dev *d)
{
    const int sigma = d->plan.sell_sigma;
    if (d->plan.sell_c != kSellC) return fail(SPMV_HIP_E_ARG, "sell_c must be 64 (one wavefront per chunk)");
    if (sigma < kSellC || sigma > 4096 || (sigma & (sigma - 1)))
        return fail(SPMV_HIP_E_ARG, "sell_sigma must be a power of two in [64, 4096], got %d", sigma);
    if (d->m == 0) return SPMV_HIP_OK;
    const int nwin = (int) (((long long) d->m + sigma - 1) / sigma);
    d->nchunks = nwin * (sigma / kSellC);
    // rows that would pad a whole chunk to their length are kept in CSR (see sell.hpp)
    double thr = 8.0 * d->stats.mean_row_len;
    if (thr < 64.0) thr = 64.0;
    if (d->plan.sell_long_thr > 0) thr = (double) d->plan.sell_long_thr;
    {
        const int rc = build_long_rows<T>(d, thr > (double) INT_MAX ? INT_MAX : (int) thr);
        if (rc) return rc;
    }
    int *width = nullptr;
    ALLOC_TRY(d, &d->perm, sizeof(int) * (size_t) nwin * sigma, true);
    ALLOC_TRY(d, &width, sizeof(int) * (size_t) d->nchunks, true);
    ALLOC_TRY(d, &d->chunk_ptr, sizeof(long long) * ((size_t) d->nchunks + 1), true);
    sell_sort_kernel<<<nwin, kBlock, sizeof(unsigned long long) * (size_t) sigma, d->stream>>>(
        d->m, sigma, d->long_thr, d->rowptr, d->perm, width);
    HIP_TRY(hipGetLastError());
    scan_i32_to_i64_kernel<<<1, kBlock, 0, d->stream>>>(d->nchunks, width, d->chunk_ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&d->sell_cols, d->chunk_ptr + d->nchunks, sizeof(long long), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    const size_t slots = (size_t) d->sell_cols * kSellC;
    ALLOC_TRY(d, &d->scol, sizeof(int) * slots, true);
    ALLOC_TRY(d, &d->sval, sizeof(T) * slots, true);
    sell_fill_kernel<T><<<grid_for(d->nchunks, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>(
        d->nchunks, d->rowptr, d->colidx, (const T *) d->val, d->perm, d->chunk_ptr, d->scol, (T *) d->sval);
    HIP_TRY(hipGetLastError());
    d->sell_nwin = nwin;
    d->sell_staged = 0;
    d->sell_group = 1;
    if (d->plan.sell_lds_x && d->plan.x_windows) { // x windows of every sigma window, in place on scol (xwindows.hpp)
        static_assert(kSellXTileBytes / sizeof(float) <= 65536, "LDS slots must fit 16 bits");
        d->sell_xcap = (int) (kSellXTileBytes / sizeof(T)) - 1; // one slot stays free: the zero slot of padding entries
        ALLOC_TRY(d, &d->sell_wins, sizeof(TileWindows) * (size_t) nwin, true);
        ALLOC_TRY(d, &d->scol16, sizeof(unsigned short) * (slots + 4), true);
        // one sigma window per workgroup, or 2 / 4 / 8 consecutive ones while staging the x windows costs more
        // than 15 % of the bytes the group streams (short rows + scattered columns: config 4)
        auto inspect = [&](int g) -> int {
            d->sell_group = g;
            d->sell_nwin = (nwin + g - 1) / g;
            return build_range_windows(d, d->sell_nwin, (long long) slots, 0, d->chunk_ptr, g * (sigma / kSellC), kSellC, d->sell_xcap, d->scol, d->sell_wins,
                                       &d->sell_staged, &d->sell_maxspan, d->scol16, 0, d->nchunks);
        };
        int rc = inspect(1);
        if (rc) return rc;
        // LDS of a workgroup: the group's x windows (<= 96 KiB) + its row sums (sell_window_kernel collects them for one coalesced store)
        auto lds_fits = [&](int g) { return kSellXTileBytes + sizeof(T) * (size_t) g * (size_t) sigma <= 150 * 1024; };
        while (d->sell_staged > 0 && d->sell_group < 8 && lds_fits(d->sell_group * 2) &&
               (double) d->sell_maxspan * sizeof(T) > 0.15 * (double) slots * (sizeof(T) + 2) / (double) d->sell_nwin) {
            const int prev = d->sell_group;
            rc = inspect(prev * 2);
            if (rc) return rc;
            if (d->sell_staged == 0) { rc = inspect(prev); if (rc) return rc; break; }
        }
        if (d->sell_staged == d->sell_nwin) { sched_free(d, d->scol); d->scol = nullptr; } // no window reads global columns
        else if (d->sell_staged == 0) { sched_free(d, d->scol16); d->scol16 = nullptr; }
        if (d->sell_staged > 0 && d->plan.run_tiles) { // RUN groups: rows that are runs of consecutive columns need no slot slab (sell.hpp)
            unsigned long long *cnt = nullptr, h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            ALLOC_TRY(d, &d->sell_run, sizeof(unsigned) * (size_t) d->nchunks * kSellC, true);
            if (!getenv("SPMV_HIP_NO_TEMPLATE_TILES")) ALLOC_TRY(d, &d->sell_tmpl, sizeof(unsigned short) * kSellTmplCount * kSellTmplMax * (size_t) d->sell_nwin, true);
            if (!getenv("SPMV_HIP_NO_BYTE_TILES")) ALLOC_TRY(d, &d->scol8, (size_t) slots + 16, true);
            HIP_TRY(pool_malloc((void **) &cnt, sizeof h));
            hipError_t e = hipMemsetAsync(cnt, 0, sizeof h, d->stream);
            sell_runs_kernel<<<d->sell_nwin, kBlock, 0, d->stream>>>(d->sell_group * (sigma / kSellC), (long long) d->nchunks, d->chunk_ptr, d->scol16, d->perm, d->rowptr,
                                                                    d->sell_wins, d->sell_run, cnt, d->sell_tmpl, d->scol8);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, d->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
            (void) pool_free(cnt);
            if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "SELL run inspector: %s", hipGetErrorString(e));
            d->sell_run_groups = (int) h[0];
            d->sell_run_nnz = (long long) h[1];
            d->sell_run_stored = (long long) h[2];
            d->sell_run_slots = (long long) h[3];
            d->sell_tmpl_nnz = (long long) h[4];
            d->sell_byte_nnz = (long long) h[5];
            d->sell_byte_stored = (long long) h[6];
            d->sell_byte_slots = (long long) h[7];
            if (d->sell_tmpl && d->sell_tmpl_nnz == 0) { sched_free(d, d->sell_tmpl); d->sell_tmpl = nullptr; }
            if (d->scol8 && d->sell_byte_nnz == 0) { sched_free(d, d->scol8); d->scol8 = nullptr; }
            if (d->sell_run_groups == d->sell_nwin && d->scol16) { sched_free(d, d->scol16); d->scol16 = nullptr; } // every group is RUN / TEMPLATE / BYTE: nobody reads the 16-bit slab
            if (getenv("SPMV_HIP_SELL_DEBUG")) fprintf(stderr, "[spmv_hip] sell: groups %d staged %d RUN %d (entries %lld, stored %lld, row slots %lld)\n", d->sell_nwin, d->sell_staged, d->sell_run_groups, d->sell_run_nnz, d->sell_run_stored, d->sell_run_slots);
        }
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}

constexpr size_t kCsr5XTileBytes = 128 * 1024; // LDS budget of one tile group's x span (a CU has 160 KiB)
constexpr size_t kNatXTileBytes = 96 * 1024;   // same for natural-layout tiles, whose waves also park their tile in LDS (up to 46 KiB)

template <typename T, int SIGMA>
static int build_csr5_sigma(spmv_dev *d, Csr5Plan &P, const int *rp, int m2, const int *colidx, const T *val)
{
    constexpr int TN = kWave * SIGMA;
    const int p = (int) ((P.nnz + TN - 1) / TN);
    P.tiles = p;
    ALLOC_TRY(d, &P.tile_ptr, sizeof(int) * ((size_t) p + 1), true);
    ALLOC_TRY(d, &P.desc, sizeof(unsigned) * (size_t) p * kWave, true);
    ALLOC_TRY(d, &P.run_len, sizeof(int) * (size_t) p, true);
    ALLOC_TRY(d, &P.carry, sizeof(T) * (size_t) p, true);
    if (P.natural) { // the tiles read the matrix's own arrays
        P.col = const_cast<int *>(colidx);
        P.val = const_cast<T *>(val);
    } else {
        ALLOC_TRY(d, &P.col, sizeof(int) * (size_t) p * TN, true);
        ALLOC_TRY(d, &P.val, sizeof(T) * (size_t) p * TN, true);
    }
    int *flag = nullptr; // [0] some tile carries into an earlier tile's row, [1] max rows of a tile
    ALLOC_TRY(d, &flag, 2 * sizeof(int), true);
    HIP_TRY(hipMemsetAsync(flag, 0, 2 * sizeof(int), d->stream));
    csr5_tile_ptr_kernel<<<grid_for((long long) p + 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(m2, (int) P.nnz, p, TN, rp, P.tile_ptr);
    HIP_TRY(hipGetLastError());
    csr5_tile_rows_max_kernel<<<grid_for(p, kBlock, INT_MAX), kBlock, 0, d->stream>>>(p, P.tile_ptr, flag + 1);
    HIP_TRY(hipGetLastError());
    csr5_desc_kernel<SIGMA><<<grid_for(p, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>(m2, (int) P.nnz, p, rp, P.tile_ptr, P.desc, P.run_len, flag);
    HIP_TRY(hipGetLastError());
    if (!P.natural) {
        csr5_transpose_kernel<T, SIGMA><<<grid_for(p, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>((int) P.nnz, p, colidx, val, P.col, (T *) P.val);
        HIP_TRY(hipGetLastError());
    }
    int flags_h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(flags_h, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    P.fixup = flags_h[0];
    P.max_tile_rows = flags_h[1];
    // x windows of every group of consecutive tiles -> the 16-bit slot stream (xwindows.hpp).  Group size:
    // 16 tiles (natural layout: 32) unless staging the windows costs more than 15 % of the bytes the group
    // streams -- wide windows, e.g. columns scattered +-4096 around the diagonal -- then 32 and 64 tiles are
    // tried as long as the groups still fit LDS (config 4: CSR5 0.64 -> 0.58 ms fp32, 1.22 -> 1.05 ms fp64;
    // narrow windows lose 3-7 % with larger groups, so they keep 16).
    static_assert(kCsr5XTileBytes / sizeof(float) <= 65536, "LDS slots must fit 16 bits");
    const int base_gt = P.natural ? 2 * kCsr5GroupTiles : kCsr5GroupTiles;
    const long long total = P.natural ? P.nnz : (long long) p * TN;
    const int max_cols = (int) ((P.natural ? kNatXTileBytes : kCsr5XTileBytes) / sizeof(T)) - 1;
    ALLOC_TRY(d, &P.wins, sizeof(TileWindows) * (size_t) ((p + 7) / 8), true);
    if (P.natural) {
        ALLOC_TRY(d, &P.col16, sizeof(unsigned short) * ((size_t) P.nnz + kStreamPad), true);
        HIP_TRY(hipMemsetAsync(P.col16, 0, sizeof(unsigned short) * ((size_t) P.nnz + kStreamPad), d->stream));
    } else {
        ALLOC_TRY(d, &P.col16, sizeof(unsigned short) * (size_t) p * TN, true);
    }
    auto inspect = [&](int gt) -> int {
        P.group_tiles = gt;
        P.groups = (p + gt - 1) / gt;
        return build_range_windows(d, !d->plan.x_windows ? 0 : P.groups, total, (long long) gt * TN, nullptr, 1, 1, max_cols, P.col, P.wins,
                                   &P.staged, &P.maxspan, P.col16, P.natural ? 0 : SIGMA);
    };
    int rc = inspect(base_gt);
    if (rc) return rc;
    while (P.staged > 0 && P.group_tiles < 64 &&
           (double) P.maxspan * sizeof(T) > 0.15 * (double) P.group_tiles * TN * (sizeof(T) + 2)) {
        const int prev = P.group_tiles;
        rc = inspect(prev * 2);
        if (rc) return rc;
        if (P.staged == 0) { // the larger groups no longer fit: back to the last size that did
            rc = inspect(prev);
            if (rc) return rc;
            break;
        }
    }
    if (getenv("SPMV_HIP_CSR5_DEBUG") && P.groups > 0) { // developer print: the distribution of the groups' staged window sizes
        std::vector<TileWindows> hw((size_t) P.groups);
        if (hipMemcpy(hw.data(), P.wins, sizeof(TileWindows) * (size_t) P.groups, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<int> tot;
            for (const auto &w : hw) if (w.nwin > 0) tot.push_back(w.total);
            std::sort(tot.begin(), tot.end());
            auto q = [&](double f) { return tot.empty() ? 0 : tot[(size_t) (f * (double) (tot.size() - 1))]; };
            fprintf(stderr, "[spmv_hip] csr5 plan: rows %d nnz %lld tiles %d group_tiles %d groups %d staged %d maxspan %d; window elements p50 %d p90 %d p99 %d p99.9 %d max %d\n",
                    m2, P.nnz, p, P.group_tiles, P.groups, P.staged, P.maxspan, q(0.5), q(0.9), q(0.99), q(0.999), q(1.0));
        }
        (void) hipGetLastError();
    }
    P.run_groups = 0;
    P.run_tiles = 0;
    if (P.staged > 0 && !P.natural && d->plan.run_tiles) { // RUN groups: a word per lane and tile instead of SIGMA slots (csr5.hpp; not for the natural layout: no gain); option run_tiles = 0: off
        unsigned long long *cnt = nullptr, h[2] = {0, 0};
        ALLOC_TRY(d, &P.lane_run, sizeof(unsigned) * (size_t) p * kWave, true);
        HIP_TRY(pool_malloc((void **) &cnt, sizeof h));
        hipError_t e = hipMemsetAsync(cnt, 0, sizeof h, d->stream);
        csr5_runs_kernel<SIGMA><<<P.groups, kBlock, 0, d->stream>>>(P.group_tiles, p, P.nnz, P.natural ? 1 : 0, P.desc, P.col16, P.wins, P.lane_run, cnt);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        (void) pool_free(cnt);
        if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "CSR5 run inspector: %s", hipGetErrorString(e));
        P.run_groups = (int) h[0];
        P.run_tiles = (long long) h[1];
    }
    if (P.staged == 0) { sched_free(d, P.col16); P.col16 = nullptr; }
    else if (!P.natural && P.staged == P.groups) { sched_free(d, P.col); P.col = nullptr; } // no group reads global columns
    // Forward completion (nat_kernel, csr5.hpp): short heavy-tailed rows in ONE launch.  Where no group stages (the tiles gather through L2 and a multiply is a
    // few tens of microseconds) the fix-up launch is a fifth of the time: tiles finish the rows they start, the few rows longer than a tile get a workgroup each.
    // Not with many or very long long rows (they would be the launch's tail): those keep carries + fix-up.
    P.forward = 0;
    if (P.natural && P.staged == 0 && P.fixup && p > 1 && d->plan.row_forward) {
        constexpr int kLongCap = 1024, kLongMax = 2 * (kNatLongU) * kBlock; // a long row's workgroup: at most two passes
        int *cnt = nullptr, h[2] = {0, 0};
        ALLOC_TRY(d, &P.fwd, sizeof(int) * (size_t) p, true);
        ALLOC_TRY(d, &P.long_list, sizeof(int4) * kLongCap, true);
        HIP_TRY(pool_malloc((void **) &cnt, sizeof h));
        hipError_t e = hipMemsetAsync(cnt, 0, sizeof h, d->stream);
        nat_forward_kernel<<<grid_for(p, kBlock, INT_MAX), kBlock, 0, d->stream>>>(p, TN, rp, P.tile_ptr, P.row_map, P.fwd, P.long_list, kLongCap, cnt);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        (void) pool_free(cnt);
        if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "forward-completion inspector: %s", hipGetErrorString(e));
        if (getenv("SPMV_HIP_CSR5_DEBUG")) fprintf(stderr, "[spmv_hip] forward completion: %d tiles, %d rows longer than a tile (longest %d)\n", p, h[0], h[1]);
        if (h[0] <= kLongCap && h[1] <= kLongMax) {
            P.forward = 1;
            P.n_long = h[0];
        } else {
            sched_free(d, P.fwd); P.fwd = nullptr;
            sched_free(d, P.long_list); P.long_list = nullptr;
        }
    }
    return SPMV_HIP_OK;
}

// CSR5 over the CSR (m rows, nnz) given by rowptr / colidx / val.  out_rows (nullable, no empty rows
// allowed then) names the y row of each CSR row -- used for the long-row sub-matrix.
template <typename T>
static int build_csr5(spmv_dev *d, Csr5Plan &P, int m, long long nnz, const int *rowptr, const int *colidx, const T *val, int empty_rows,
                      double mean_row_len, const int *out_rows, bool natural, int sigma_override)
{
    const size_t alloc_mark = d->sched_allocs.size();
    int sigma = sigma_override ? sigma_override : d->plan.csr5_sigma;
    // sigma = 16 whatever the row length (the reference's CPU heuristic shrinks sigma for short rows; here larger
    // tiles amortise the per-tile descriptor / tile_ptr / carry work: 5-entry rows ran 0.48 / 0.34 / 0.28 ms at
    // sigma 4 / 8 / 16); smaller tiles only when there would be too few of them to fill the chip
    (void) mean_row_len;
    if (sigma == 0) sigma = nnz >= (1ll << 22) ? 16 : (nnz >= (1ll << 19) ? 8 : 4);
    if (sigma != 4 && sigma != 8 && sigma != 16) return fail(SPMV_HIP_E_ARG, "csr5_sigma must be 4, 8 or 16 (0 = auto), got %d", sigma);
    P = Csr5Plan();
    P.sigma = sigma;
    P.nnz = nnz;
    P.row_map = out_rows;
    P.natural = natural;
    if (nnz == 0) return SPMV_HIP_OK;
    const int *rp = rowptr;
    int m2 = m;
    if (empty_rows > 0) { // build over the compacted (non-empty) row space
        if (out_rows) return fail(SPMV_HIP_E_ARG, "csr5: a row map and empty rows cannot be combined");
        const int nb = (int) (((long long) m + kScanTile - 1) / kScanTile);
        int *flags = nullptr, *sums = nullptr, *total = nullptr, *rp2 = nullptr, *rmap = nullptr, *elist = nullptr;
        HIP_TRY(pool_malloc((void **) &flags, sizeof(int) * (size_t) m));
        auto cleanup = [&]() { (void) pool_free(flags); };
        if (dev_alloc(d, (void **) &sums, sizeof(int) * (size_t) nb, true) || dev_alloc(d, (void **) &total, sizeof(int), true)) { cleanup(); return SPMV_HIP_E_ALLOC; }
        csr5_nonempty_kernel<<<grid_for(m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(m, rowptr, flags);
        scan_block_sums_kernel<<<nb, kBlock, 0, d->stream>>>(m, flags, sums);
        scan_sums_inplace_kernel<<<1, kBlock, 0, d->stream>>>(nb, sums, total);
        if (hipMemcpyAsync(&m2, total, sizeof(int), hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
            hipStreamSynchronize(d->stream) != hipSuccess) { cleanup(); return fail(SPMV_HIP_E_RUNTIME, "csr5 compaction scan failed"); }
        if (dev_alloc(d, (void **) &rp2, sizeof(int) * ((size_t) m2 + 1), true) ||
            dev_alloc(d, (void **) &rmap, sizeof(int) * (size_t) (m2 > 0 ? m2 : 1), true) ||
            dev_alloc(d, (void **) &elist, sizeof(int) * (size_t) (m - m2 > 0 ? m - m2 : 1), true)) { cleanup(); return SPMV_HIP_E_ALLOC; }
        csr5_compact_kernel<<<nb, kBlock, 0, d->stream>>>(m, flags, sums, rowptr, rp2, rmap, elist);
        const int nnz32 = (int) nnz;
        hipError_t e = hipMemcpyAsync(rp2 + m2, &nnz32, sizeof(int), hipMemcpyHostToDevice, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        cleanup();
        if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "csr5 compaction: %s", hipGetErrorString(e));
        rp = rp2;
        P.row_map = rmap;
        P.n_empty = m - m2;
        P.empty_list = elist;
    }
    P.m2 = m2;
    int rc;
    switch (sigma) {
    case 4: rc = build_csr5_sigma<T, 4>(d, P, rp, m2, colidx, val); break;
    case 8: rc = build_csr5_sigma<T, 8>(d, P, rp, m2, colidx, val); break;
    default: rc = build_csr5_sigma<T, 16>(d, P, rp, m2, colidx, val); break;
    }
    // Natural-layout tiles of 512 entries (the automatic size between 2^19 and 2^22 non-zeros) of which no group stages its x windows, finished by forward completion:
    // 256-entry tiles run the launch in one round of twice as many, lighter workgroups (webbase-style R-MAT stand-in 26.1 -> 25.0 us; with the carry fix-up launch
    // the sizes measured equal).  The inspector at this size is a fraction of a millisecond: build again.
    if (!rc && natural && !sigma_override && d->plan.csr5_sigma == 0 && sigma == 8 && P.staged == 0 && P.forward) {
        quiesce(d);
        while (d->sched_allocs.size() > alloc_mark) sched_free(d, d->sched_allocs.back().first);
        return build_csr5<T>(d, P, m, nnz, rowptr, colidx, val, empty_rows, mean_row_len, out_rows, natural, 4);
    }
    return rc;
}

// Row blocks x column slabs (kernels/blocked.hpp): the inspector's three kernels.  values_only re-permutes new values into the
// positions of the existing layout (spmv_hip_update_values): cursors are recomputed inside the fill kernel, nothing is kept.
template <typename T>
static int blocked_fill(spmv_dev *d, bool values_only)
{
    BlkSet &S = d->blk;
    const int B = S.B, K = S.K;
    int *groups = nullptr, *occupied = nullptr;
    unsigned short *rowin = nullptr;
    std::vector<int> occ_h;
    auto cleanup = [&]() { if (groups) (void) pool_free(groups); if (occupied) (void) pool_free(occupied); if (rowin) (void) pool_free(rowin); };
    hipError_t e = hipSuccess;
    if (!values_only) {
        if (pool_malloc((void **) &groups, sizeof(int) * (size_t) B) != hipSuccess || pool_malloc((void **) &occupied, sizeof(int) * (size_t) B) != hipSuccess ||
            pool_malloc((void **) &rowin, sizeof(unsigned short) * ((size_t) d->nnz + 8)) != hipSuccess) {
            (void) hipGetLastError();
            cleanup();
            return fail(SPMV_HIP_E_ALLOC, "pool_malloc(block inspector scratch)");
        }
        ensure_lds<blk_rows_kernel>(d, sizeof(int) * ((size_t) S.R + 1));
        blk_rows_kernel<<<B, kBlkThreads, sizeof(int) * ((size_t) S.R + 1), d->stream>>>(S.row0, d->rowptr, rowin);
        ensure_lds<blk_count_kernel>(d, sizeof(unsigned) * (size_t) K);
        blk_count_kernel<<<B, kBlkThreads, sizeof(unsigned) * (size_t) K, d->stream>>>(S.row0, K, S.wshift, S.ge, d->rowptr, d->colidx, groups, occupied);
        e = hipGetLastError();
        int rc = dev_alloc(d, (void **) &S.gstart, sizeof(long long) * ((size_t) B + 1), true);
        if (!rc) rc = dev_alloc(d, (void **) &S.dir, sizeof(BlkDir) * (size_t) B, true);
        if (rc) { cleanup(); return rc; }
        scan_i32_to_i64_kernel<<<1, kBlock, 0, d->stream>>>(B, groups, S.gstart);
        long long total = 0;
        if (e == hipSuccess) e = hipGetLastError();
        occ_h.resize((size_t) B);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, S.gstart + B, sizeof(long long), hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(occ_h.data(), occupied, sizeof(int) * (size_t) B, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        if (e != hipSuccess) { cleanup(); return fail(SPMV_HIP_E_RUNTIME, "block inspector: %s", hipGetErrorString(e)); }
        S.groups = total;
        if ((total + kBlkPadGroups) >> (31 - S.ge) > 0) { cleanup(); return fail(SPMV_HIP_E_RANGE, "blocked layout: %lld groups exceed 32-bit positions", total); }
        const size_t ng = (size_t) total + kBlkPadGroups, slots = ng << S.ge; // the executor loads up to three steps past a block's end, unguarded: keep that in bounds
        rc = dev_alloc(d, &S.val, sizeof(T) * slots, true);
        if (!rc) rc = dev_alloc(d, (void **) &S.meta, sizeof(unsigned) * slots, true);
        if (!rc) rc = dev_alloc(d, (void **) &S.hdr, sizeof(int) * ng, true);
        if (rc) { cleanup(); return rc; }
        (void) hipMemsetAsync(S.val, 0, sizeof(T) * slots, d->stream);      // padding entries: value 0, column offset 0,
        fill_value_kernel<unsigned><<<grid_for((long long) slots, kBlock * 4, d->cus * 16), kBlock, 0, d->stream>>>((long long) slots, S.meta, (unsigned) S.R << 16); // row = the junk accumulator
        (void) hipMemsetAsync(S.hdr, 0, sizeof(int) * ng, d->stream);
    }
    const size_t lds = 3 * sizeof(unsigned) * (size_t) K;
    unsigned short *kscr = nullptr; // values-only refresh of a column-sorted layout: the cells' sort keys, which the stored words no longer hold in CSR order
    if (values_only) {
        if (S.subsort && pool_malloc((void **) &kscr, sizeof(unsigned short) * (((size_t) S.groups + kBlkPadGroups) << S.ge)) != hipSuccess) {
            (void) hipGetLastError();
            return fail(SPMV_HIP_E_ALLOC, "pool_malloc(block refresh scratch)");
        }
        ensure_lds<blk_fill_kernel<T, true>>(d, lds);
        blk_fill_kernel<T, true><<<B, kBlkThreads, lds, d->stream>>>(S.row0, K, S.wshift, S.ge, d->rowptr, d->colidx, (const T *) d->val, nullptr, S.gstart, (T *) S.val, S.meta, S.hdr, S.dir,
                                                                     S.subsort ? 1 : 0, kscr);
    } else {
        ensure_lds<blk_fill_kernel<T, false>>(d, lds);
        blk_fill_kernel<T, false><<<B, kBlkThreads, lds, d->stream>>>(S.row0, K, S.wshift, S.ge, d->rowptr, d->colidx, (const T *) d->val, rowin, S.gstart, (T *) S.val, S.meta, S.hdr, S.dir,
                                                                      S.subsort ? 1 : 0, nullptr);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    cleanup();
    if (kscr) (void) pool_free(kscr);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "block fill: %s", hipGetErrorString(e));
    if (!values_only) { // launch order (blk_kernel): by how scattered a block's entries are -- classes of occupied cells per entry, the most scattered first --, row order inside a class, empty blocks last
        std::vector<BlkDir> hd((size_t) B);
        HIP_TRY(hipMemcpy(hd.data(), S.dir, sizeof(BlkDir) * (size_t) B, hipMemcpyDeviceToHost));
        auto klass = [&](int b) { // 0 = empty (last); else 1 + log2-ish class of occupied cells per group, capped: homogeneous matrices stay in row order
            if (hd[(size_t) b].ns <= 0) return 0;
            const double per_group = (double) occ_h[(size_t) b] / (double) hd[(size_t) b].ns; // <= 2^ge
            int k = 1;
            for (double t = 0.25; t <= per_group && k < 12; t *= 2.0) ++k;
            return k;
        };
        std::vector<int> order((size_t) B);
        for (int b = 0; b < B; ++b) order[(size_t) b] = b;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return klass(a) > klass(b); });
        int rc = dev_alloc(d, (void **) &S.order, sizeof(int) * (size_t) B, true);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(S.order, order.data(), sizeof(int) * (size_t) B, hipMemcpyHostToDevice));
    }
    return SPMV_HIP_OK;
}

// LDS of one executor workgroup: the block's accumulators + the junk slot
static size_t blocked_lds_bytes(const BlkSet &S) { return sizeof(double) * (size_t) ((S.R + 2) & ~1); }

constexpr int kBlkRowCap = 78 * 1024 / (int) sizeof(double) - 2; // most rows of a block such that TWO blocks fit a CU's 160 KiB of LDS (78 KiB each)
constexpr int kBlkRowCapWide = 159 * 1024 / (int) sizeof(double) - 2; // wide form: ONE block per CU (159 KiB of accumulators)

// How many row blocks, and how many rows at most in one.  rule 0 (first choice): FULL ROUNDS of fat blocks -- two blocks are
// resident per CU (2 x 78 KiB of its 160 KiB of LDS), so the block count is a multiple of 2 * CUs with at most rmax rows each:
// the grid then runs in whole rounds.  With 8192-row blocks config 2-ii has 1221 blocks = 2.4 rounds of 512 and pays for three
// (1.75 ms); with 1024 blocks exactly two (1.54 ms); 1028 blocks: 2.02 ms, the four stragglers cost a round.  Orkut-style uniform:
// 512 blocks 0.79 ms vs 750 blocks of 4096 rows 1.02; R-MAT 0.69 vs 0.70.  rule 1: at least 512 blocks of at most 8192 rows (a
// power of two) -- more, thinner blocks = more waves per CU, what a STREAM-bound matrix wants (web-like 4e6 x 24: 977 blocks
// 0.245 ms vs 512 blocks 0.275); spmv_shim_build tries it when rule 0 turns out stream-bound and keeps the faster set.  The cut
// points themselves follow the work, not the row count (blk_partition_kernel).  Option block_rows: uniform blocks of that many rows.
static void blocked_block_rule(const spmv_dev *d, int rule, int rmax, int *btarget, int *rcap)
{
    if (d->plan.block_rows > 0) {
        *rcap = d->plan.block_rows < 16384 ? d->plan.block_rows : 16384; // 128 KiB of double accumulators; row numbers inside a block are 16-bit
        *btarget = (int) (((long long) d->m + *rcap - 1) / *rcap);
        return;
    }
    if (rule == 0) {
        const long long slots = (rmax > kBlkRowCap ? 1ll : 2ll) * (d->cus > 0 ? d->cus : 256); // wide form: one block per CU
        const long long rounds = ((long long) d->m + slots * rmax - 1) / (slots * rmax);
        long long B = slots * (rounds > 0 ? rounds : 1);
        if ((long long) d->m / B < 1024) B = ((long long) d->m + 1023) / 1024; // small matrices: blocks of about 1024 rows
        *btarget = (int) (B > 0 ? B : 1);
        *rcap = rmax;
    } else {
        int R = (int) (64 * 1024 / sizeof(double));
        while (R > 1024 && (long long) d->m / R < 512) R >>= 1; // small matrices: at least ~512 blocks, down to 1024 rows
        *btarget = (int) (((long long) d->m + R - 1) / R);
        *rcap = R + R / 4; // equal-work cut points may stretch a block of light rows
        if (*rcap > rmax) *rcap = rmax;
    }
}

// would rule 1 cut the rows differently from rule 0?
static bool blocked_differs(const spmv_dev *d)
{
    int b0, r0, b1, r1;
    blocked_block_rule(d, 0, kBlkRowCap, &b0, &r0);
    blocked_block_rule(d, 1, kBlkRowCap, &b1, &r1);
    return b0 != b1;
}

// cut points of the row blocks: uniform for option block_rows, equal work otherwise
static int blocked_partition(spmv_dev *d, int btarget, int rcap)
{
    BlkSet &S = d->blk;
    const size_t cap_blocks = (size_t) btarget + (size_t) (d->m / rcap) + 2;
    int rc = dev_alloc(d, (void **) &S.row0, sizeof(int) * (cap_blocks + 1), true);
    if (rc) return rc;
    int B = btarget, R = rcap;
    if (d->plan.block_rows > 0) {
        std::vector<int> h((size_t) B + 1);
        for (int b = 0; b <= B; ++b) h[(size_t) b] = (int) std::min<long long>((long long) b * rcap, d->m);
        HIP_TRY(hipMemcpyAsync(S.row0, h.data(), sizeof(int) * ((size_t) B + 1), hipMemcpyHostToDevice, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
    } else {
        int *out = nullptr, hout[2] = {0, 0}; // [0..1] results, [2 ..] the kernel's cut-point scratch
        HIP_TRY(pool_malloc((void **) &out, sizeof(int) * ((size_t) btarget + 3)));
        // fixed cost of a row, in entries.  The far half of a split matrix (shim/split.hpp) has rows without any entry by construction --
        // often long stretches of them (a banded matrix whose last tenth is random: 9e6 empty rows, which at cost 1 each took a fifth
        // of the "work" and left the real entries to 120 of 1024 blocks): there a row costs nothing and the row cap alone ends blocks
        const long long c = d->accumulate ? 0 : std::max<long long>(1, (long long) (d->stats.mean_row_len / 8.0));
        blk_partition_kernel<<<1, kBlock, 0, d->stream>>>(d->m, d->rowptr, btarget, rcap, c, out + 2, S.row0, out);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(hout, out, sizeof hout, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
        (void) pool_free(out);
        if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "block partition: %s", hipGetErrorString(e));
        B = hout[0];
        R = hout[1] > 0 ? hout[1] : 1;
        if (B < 1 || (size_t) B > cap_blocks) return fail(SPMV_HIP_E_RUNTIME, "block partition produced %d blocks", B);
    }
    S.B = B;
    S.R = R;
    return SPMV_HIP_OK;
}

template <typename T>
static int build_blocked(spmv_dev *d, int rule, int waves, bool ordered)
{
    BlkSet &S = d->blk;
    S = BlkSet();
    // Slab width: 128 columns, wider only when a block would have more than 12288 cells (the inspector's per-cell state lives in LDS),
    // or by option slab_kib.  Entries that share an executor instruction are 64 consecutive ones of the (slab, CSR) order, so what a
    // narrow slab buys is that the few entries a block has in the same cache line of x end up in the same instruction and merge.
    int wshift = kBlkSlabShift;
    if (d->plan.slab_kib > 0) {
        wshift = 0;
        while ((sizeof(T) << wshift) < ((size_t) d->plan.slab_kib << 10)) ++wshift;
    }
    while (wshift < kBlkSuperShift && ((((long long) d->n - 1) >> wshift) + 1) > kBlkMaxCells) ++wshift;
    if (wshift > kBlkSuperShift) wshift = kBlkSuperShift;
    S.wshift = wshift;
    S.K = (int) ((((long long) d->n - 1) >> wshift) + 1);
    S.ge = sizeof(T) == 8 ? 7 : 8; // 64 lanes x 16 bytes of values
    int btarget = 1, rcap = 1024;
    S.waves = waves;
    if (S.waves != 1 && S.waves != 2 && S.waves != 4 && S.waves != 8) return fail(SPMV_HIP_E_ARG, "blk_waves must be 0, 1, 2, 4 or 8, got %d", waves);
    S.ordered = ordered;
    S.subsort = d->plan.blk_subsort != 0;
    blocked_block_rule(d, S.waves > 1 ? 0 : rule, S.waves > 1 ? kBlkRowCapWide : kBlkRowCap, &btarget, &rcap);
    int rc = blocked_partition(d, btarget, rcap);
    if (!rc) rc = blocked_fill<T>(d, false);
    if (rc) return rc;
    if (getenv("SPMV_HIP_BLK_DEBUG"))
        fprintf(stderr, "[spmv_hip] blocked: m %d nnz %lld -> B %d (target %d) R %d K %d wshift %d groups %lld accumulate %d waves %d ordered %d subsort %d\n", d->m, d->nnz, S.B, btarget, S.R, S.K, S.wshift, S.groups, (int) d->accumulate, S.waves, (int) S.ordered, (int) S.subsort);
    d->blk_on = true;
    return SPMV_HIP_OK;
}

// Is the matrix large enough for the row-block x column-slab executor to pay?  From which x size on?  Uniformly random columns,
// 16 nnz/row (tools/ab_threshold.py): the blocked executor wins as soon as x reaches one XCD's L2 -- fp64, x = 4 / 8 / 16 / 32 MB:
// 49 / 94 / 180 / 352 us against 55-62 / 157-168 / 434-450 / 1030-1050 us on the tile kernels (all four of them); at 2 MB the tile
// kernels lead (45 vs 50 us).  With very short rows the per-row work of a block (zeroing and writing its y) weighs more: the
// 1e6-row power-law stand-in (2.6 nnz/row) at x = 8 / 16 / 32 MB (tools/ab_short_rows.py): tile 26.8 / 50.4 / 98.4 us against
// blocked 28.0 / 46.9 / 77.4 (R-MAT columns), 21.9 / 41.3 / 76.1 against 25.0 / 36.2 / 58.2 (web-like), 38.2 / 90.9 / 201 against
// 37.2 / 64.5 / 121 (uniform) -> 12 MiB there.
// the inspector keeps a row block's cells in LDS: at most kBlkMaxCells slabs of at most 65536 columns
static bool blocked_possible(const spmv_dev *d) { return (long long) d->n <= (long long) kBlkMaxCells << kBlkSuperShift; }

static bool blocked_size_ok(const spmv_dev *d)
{
    if (!blocked_possible(d)) return false;
    const long long min_x = d->stats.mean_row_len >= 8.0 ? (4ll << 20) : (12ll << 20);
    return d->nnz >= (1ll << 21) && (long long) d->n * (long long) d->vsize >= min_x;
}

// The tile schedule just built stages `staged` of its `groups` tile groups' x windows in LDS; the others gather x through
// L1/L2 -- across the fabric when x is far larger than an L2 (8 % of roofline).  Who multiplies?  0: the tile schedule,
// 1: row blocks x column slabs, 2: build both and let create() time them (option "cache_block": 1 automatic, 2 always, 0 never).
// Round 2's rule was staged == 0, which left a random-column matrix with ONE stageable group among 39 063 on the 6 ms kernels.
static int blocked_mode(const spmv_dev *d, int staged, int groups)
{
    if (d->plan.cache_block == 2) return d->nnz > 0 && blocked_possible(d) ? 1 : 0;
    if (d->plan.cache_block != 1 || !d->plan.x_windows) return 0; // x_windows = 0: the tile executors with global gathers (A/B)
    if (!blocked_size_ok(d) || groups <= 0 || staged >= groups) return 0;
    if ((long long) staged * 2 < groups) return 1;                   // most groups gather globally: no contest (the tile inspectors do not even stage the rest then)
    if ((long long) (groups - staged) * 200 < groups) return 0;      // under 0.5 % of the groups: at most a few percent of the time
    return d->plan.autotune ? 2 : ((long long) (groups - staged) * 10 >= groups ? 1 : 0); // without timing: from 10 % on
}

// Sample before building anything: 64 windows of 4096 consecutive non-zeros, evenly spaced.  A window "has no locality" when
// its entries touch so many distinct 64-column segments of x that not even the largest LDS budget (128 KiB) could hold them,
// and more than a third of its entries sit in a segment of their own.  If (nearly) every window says so, no tile schedule will
// stage anything and its inspector products (windows, CSR5 transposes, SELL slabs) would be built only to be dropped.
static __global__ __launch_bounds__(kBlock) void locality_sample_kernel(long long nnz, int windows, int wlen, const int *__restrict__ colidx, int seg_limit, int *__restrict__ hopeless)
{
    constexpr int kSlots = 8192; // open-addressing set of segment ids, 2 x the window length
    __shared__ int set[kSlots];
    __shared__ int distinct;
    for (int i = threadIdx.x; i < kSlots; i += kBlock) set[i] = -1;
    if (threadIdx.x == 0) distinct = 0;
    __syncthreads();
    const long long start = windows > 1 ? (nnz - wlen) / (windows - 1) * blockIdx.x : 0;
    int mine = 0;
    for (int i = threadIdx.x; i < wlen; i += kBlock) {
        const int seg = colidx[start + i] >> 6;
        unsigned h = ((unsigned) seg * 2654435761u) >> 19; // 13 bits
        for (;;) {
            const int prev = atomicCAS(&set[h], -1, seg);
            if (prev == -1) { ++mine; break; }
            if (prev == seg) break;
            h = (h + 1) & (kSlots - 1);
        }
    }
    if (mine) atomicAdd(&distinct, mine);
    __syncthreads();
    if (threadIdx.x == 0 && distinct > seg_limit && distinct * 3 > wlen) atomicAdd(hopeless, 1);
}

static bool sample_says_no_locality(spmv_dev *d)
{
    constexpr int kWindows = 64, kLen = 4096;
    if (d->nnz < (long long) kWindows * kLen * 4) return false;
    int *cnt = nullptr, h = 0;
    if (pool_malloc((void **) &cnt, sizeof(int)) != hipSuccess) { (void) hipGetLastError(); return false; }
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int), d->stream);
    const int seg_limit = (int) (kCsr5XTileBytes / d->vsize / 64); // segments the largest x-window budget holds
    locality_sample_kernel<<<kWindows, kBlock, 0, d->stream>>>(d->nnz, kWindows, kLen, d->colidx, seg_limit, cnt);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h, cnt, sizeof(int), hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void) pool_free(cnt);
    if (e != hipSuccess) { (void) hipGetLastError(); return false; }
    return h * 16 >= kWindows * 15; // 15 of 16 windows
}

// ------------------------------------------------------------------------------------ traffic model
// spmv_hip_info.stream_bytes: the HBM bytes ONE launch of the built schedule has to move, from the sizes of
// the arrays its executor streams (as stored: 16-bit slot streams where x windows are staged, padding
// included), the x elements it stages (sum of the window sizes; n once where it gathers through L2), y written
// once and the carries.  bench.py divides this by the measured launch time (roofline.achieved); the rocprofv3
// FETCH_SIZE / WRITE_SIZE passes under profiles/ check the model (config 2: model 3.43 GB, counters 3.43 GB).
struct Traffic {
    long long bytes = 0, x_elems = 0;
    bool gathers_global = false; // some tile reads x through L1/L2: charge the whole vector once
};

static int wins_sum(spmv_dev *d, const TileWindows *wins, int count, long long *elems, long long *tiles)
{
    *elems = *tiles = 0;
    if (!wins || count <= 0) return SPMV_HIP_OK;
    unsigned long long *dv = nullptr, hv[2] = {0, 0};
    HIP_TRY(pool_malloc((void **) &dv, sizeof hv));
    hipError_t e = hipMemsetAsync(dv, 0, sizeof hv, d->stream);
    wins_total_kernel<<<grid_for(count, kBlock, d->cus * 4), kBlock, 0, d->stream>>>(count, wins, dv, dv + 1);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hv, dv, sizeof hv, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void) pool_free(dv);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "window totals: %s", hipGetErrorString(e));
    *elems = (long long) hv[0];
    *tiles = (long long) hv[1];
    return SPMV_HIP_OK;
}

// value + column streams of `stored` entries of which the fraction fs sits in staged groups
static long long stream_part(long long stored, long long s, double fs)
{
    return stored * s + (long long) ((double) stored * (2.0 * fs + 4.0 * (1.0 - fs)));
}

static int csr5_traffic(spmv_dev *d, const Csr5Plan &P, Traffic &t)
{
    if (P.nnz == 0) return SPMV_HIP_OK;
    const long long s = (long long) d->vsize, TN = (long long) kWave * P.sigma, p = P.tiles;
    long long welems = 0, wtiles = 0;
    if (P.staged > 0) { const int rc = wins_sum(d, P.wins, P.groups, &welems, &wtiles); if (rc) return rc; }
    const double fs = P.groups > 0 ? (double) wtiles / (double) P.groups : 0.0;
    t.bytes += stream_part(P.natural ? P.nnz : p * TN, s, fs);
    t.bytes += 4 * (p + 1) + 4 * kWave * p + s * p;            // tile_ptr, descriptors, carries written
    if (P.forward) t.bytes += 4ll * p;                          // forward completion: a count per tile, no fix-up launch
    else if (P.fixup && p > 1) t.bytes += (s + 8) * p;          // fix-up launch: carries, tile_ptr, run_len
    if (P.staged > 0) t.bytes += (long long) sizeof(TileWindows) * P.groups;
    t.bytes += P.run_tiles * (4ll * kWave - 2ll * TN); // RUN groups: a word per lane and tile instead of 16 bits per entry
    if (P.row_map) t.bytes += 4ll * P.m2;
    t.bytes += s * P.m2 + (s + 4) * (long long) P.n_empty;      // y of the plan's rows; zero fill of the empty rows
    t.x_elems += welems;
    if (wtiles < P.groups) t.gathers_global = true;
    return SPMV_HIP_OK;
}

static int account_stream_bytes(spmv_dev *d)
{
    const long long s = (long long) d->vsize, m = d->m, n = d->n;
    Traffic t;
    int rc = SPMV_HIP_OK;
    if (d->blk_on) { // value + (column | row) word per stored entry, the group headers, the block directory, y once
        t.bytes = (d->blk.groups << d->blk.ge) * (s + 4) + 4ll * d->blk.groups + (long long) (sizeof(BlkDir) + 8) * d->blk.B + s * m;
        t.gathers_global = true;
    } else {
        switch (d->plan.sched) {
        case SPMV_SCHED_CSR_VECTOR:
        case SPMV_SCHED_ROWBLOCK: {
            long long welems = 0, wtiles = 0;
            rc = wins_sum(d, d->vt_wins, d->vt_tiles, &welems, &wtiles);
            if (rc) return rc;
            const bool tile_form = d->vt_tiles > 0 && (d->plan.sched == SPMV_SCHED_ROWBLOCK || d->vt_wide || d->vt_staged * 2 >= d->vt_tiles);
            const double fs = tile_form && d->vt_tiles > 0 ? (double) wtiles / (double) d->vt_tiles : 0.0; // pipe form: int32 columns
            t.bytes = 4 * (m + 1) + stream_part(d->nnz - d->lsub_nnz, s, fs) + s * (m - d->nlong);
            if (tile_form) { t.bytes += (long long) sizeof(TileWindows) * d->vt_tiles; t.x_elems = welems; }
            if (tile_form) t.bytes += 2ll * d->vt_run_rows - 2ll * d->vt_run_nnz; // RUN tiles: a 16-bit slot per row instead of one per entry
            if (tile_form) t.bytes += 3ll * d->vt_tmpl_rows + 2ll * kTmplCount * kTmplMax * d->vt_tmpl_tiles - 2ll * d->vt_tmpl_nnz; // TEMPLATE tiles: a 16-bit slot and a list number per row, the lists once per tile
            if (tile_form) t.bytes += 2ll * d->vt_byte_rows - 1ll * d->vt_byte_nnz; // BYTE tiles: a byte per entry and a 16-bit slot per row instead of 16 bits per entry
            if (d->plan.sched == SPMV_SCHED_ROWBLOCK) t.bytes += 4ll * (d->nblocks + 1);
            if (!tile_form || wtiles < d->vt_tiles) t.gathers_global = true;
            rc = csr5_traffic(d, d->c5_long, t);
            break;
        }
        case SPMV_SCHED_NNZ_SPLIT: rc = csr5_traffic(d, d->ns, t); break;
        case SPMV_SCHED_CSR5: rc = csr5_traffic(d, d->c5, t); break;
        case SPMV_SCHED_SELL: {
            long long welems = 0, wtiles = 0;
            if (d->sell_staged > 0) { rc = wins_sum(d, d->sell_wins, d->sell_nwin, &welems, &wtiles); if (rc) return rc; }
            const double fs = d->sell_nwin > 0 ? (double) wtiles / (double) d->sell_nwin : 0.0;
            t.bytes = stream_part(d->sell_cols * kSellC, s, fs) + 8ll * (d->nchunks + 1) + 4ll * d->nchunks * kSellC + s * (m - d->nlong);
            t.bytes += 4ll * d->sell_run_slots - 2ll * d->sell_run_stored; // RUN groups: a word per row slot instead of a 16-bit slot per stored entry
            t.bytes += 4ll * d->sell_byte_slots - d->sell_byte_stored;    // BYTE groups: the word + 1 B per stored entry instead of 2
            if (d->sell_staged > 0) t.bytes += (long long) sizeof(TileWindows) * d->sell_nwin;
            t.x_elems = welems;
            if (wtiles < d->sell_nwin) t.gathers_global = true;
            rc = csr5_traffic(d, d->c5_long, t);
            break;
        }
        default: // CSR-scalar: the plain arrays
            t.bytes = 4 * (m + 1) + d->nnz * (4 + s) + s * m;
            t.gathers_global = d->nnz > 0;
            break;
        }
    }
    if (rc) return rc;
    // x: the windows staged, but no more than one pass over x per XCD -- windows of neighbouring tiles that overlap almost entirely (rows scattered
    // +-4096 columns around the diagonal: 256-row tiles stage 8 448 columns each) are re-read from that XCD's L2, not from memory (without the cap
    // the model had CSR-vector on 1e7 x 16 such rows move 10 TB/s)
    const long long x_cap = 8ll * n;
    d->x_bytes = s * ((t.x_elems < x_cap ? t.x_elems : x_cap) + (t.gathers_global ? n : 0));
    d->stream_bytes = t.bytes + d->x_bytes;
    return SPMV_HIP_OK;
}
