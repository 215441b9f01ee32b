// shim/split.hpp -- part of the single translation unit spmv_shim.hip: a handle whose matrix is multiplied as A_near + A_far
// (kernels/split.hpp).  The host C side (spmv_api.c) asks whether a split is worth trying, has the two halves made, plans and
// builds each like any matrix -- near: never the blocked executor; far: always, accumulating -- attaches them, times the pair against
// the unsplit schedule and keeps the faster (the same protocol as the shards of a multi-GPU handle).
#pragma once

// most columns either side of a tile's centre that still fit the smallest x-window budget (CSR-vector's 48 KiB) with room for
// the 64-column segment rounding and the drift of the centre over nnz-based tile groups
static int split_half_width(const spmv_dev *d) { return (int) (kVecXTileBytes / d->vsize) / 2 - 256; }

// Share of the entries that lie near their tile's centre column, from a sample (64 windows x 4096 entries); -1 if not sampled.
static float sample_near_share(spmv_dev *d)
{
    constexpr int kWindows = 64, kLen = 4096;
    if (d->nnz < (long long) kWindows * kLen * 4) return -1.f;
    unsigned long long *cnt = nullptr, h[2] = {0, 0};
    if (pool_malloc((void **) &cnt, sizeof h) != hipSuccess) { (void) hipGetLastError(); return -1.f; }
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof h, d->stream);
    split_sample_kernel<<<kWindows, kBlock, 0, d->stream>>>(d->nnz, kWindows, kLen, d->colidx, split_half_width(d), cnt);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, cnt, sizeof h, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void) pool_free(cnt);
    if (e != hipSuccess || h[1] == 0) { (void) hipGetLastError(); return -1.f; }
    return (float) ((double) h[0] / (double) h[1]);
}

// Is a near / far split worth building and timing?  Only where the blocked executor would be considered at all (size, option
// cache_block = 1, autotune on, no A/B variant), the schedule as built leaves tile groups unstaged (or the blocked executor took the
// whole matrix), and the sampled near share says that part of the entries -- at least 3 %, not nearly all -- has locality.
extern "C" int spmv_shim_split_candidate(spmv_dev *d)
{
    if (!d || !d->built || d->sp_near || d->accumulate) return 0;
    if (d->plan.sched == SPMV_SCHED_CSR_SCALAR || d->plan.cache_block != 1 || d->plan.forced || !d->plan.autotune || d->plan.block_rows != 0) return 0;
    if (!blocked_size_ok(d) || d->nnz < (1ll << 22)) return 0;
    if (!d->blk_on && d->route_ms[0] == 0.f) return 0; // every group stages (or under 0.5 % do not): nothing to gain
    if (!d->blk_on && d->route_ms[0] > 0.f && (double) d->stream_bytes >= 5.6e12 * (double) d->route_ms[0] * 1e-3) return 0; // the tile schedule moves its own bytes at >= 0.70 of the HBM peak: no split can be worth its create time
    DeviceGuard guard(d->device);
    if (!guard.ok) return 0;
    if (d->near_share < 0.f) d->near_share = sample_near_share(d);
    return d->near_share >= 0.03f && d->near_share <= 0.995f; // create() times the pair: a hopeless split costs create time only
}

template <typename T>
static int split_make(spmv_dev *d, spmv_dev **near_out, spmv_dev **far_out, bool values_only)
{
    const int m = d->m, half = split_half_width(d);
    const int tiles = (int) (((long long) m + kSplitTileRows - 1) / kSplitTileRows);
    const int nb = (int) (((long long) m + kScanTile - 1) / kScanTile);
    int *cnt = nullptr, *sums = nullptr, *total = nullptr;
    spmv_dev *dn = values_only ? d->sp_near : nullptr, *df = values_only ? d->sp_far : nullptr;
    auto cleanup = [&]() { for (void *p : {(void *) cnt, (void *) sums, (void *) total}) if (p) (void) pool_free(p); };
    auto bail = [&](int code) {
        cleanup();
        if (!values_only) { if (dn) spmv_shim_matrix_destroy(dn); if (df) spmv_shim_matrix_destroy(df); }
        return code;
    };
    if (!values_only) {
        if (!d->sp_centre && pool_malloc((void **) &d->sp_centre, sizeof(int) * (size_t) (tiles > 0 ? tiles : 1)) != hipSuccess) { (void) hipGetLastError(); return fail(SPMV_HIP_E_ALLOC, "split: tile centres"); }
        split_center_kernel<<<grid_for(tiles, kBlock, INT_MAX), kBlock, 0, d->stream>>>(m, d->rowptr, d->colidx, tiles, d->sp_centre);
        d->sp_half = half;
        if (pool_malloc((void **) &cnt, sizeof(int) * (size_t) m) != hipSuccess || pool_malloc((void **) &sums, sizeof(int) * (size_t) nb) != hipSuccess ||
            pool_malloc((void **) &total, sizeof(int)) != hipSuccess) { (void) hipGetLastError(); return bail(fail(SPMV_HIP_E_ALLOC, "split: scratch")); }
        split_count_kernel<<<grid_for(m, kBlock / 16, d->cus * 16), kBlock, 0, d->stream>>>(m, d->rowptr, d->colidx, d->sp_centre, half, cnt);
        scan_block_sums_kernel<<<nb, kBlock, 0, d->stream>>>(m, cnt, sums);
        scan_sums_inplace_kernel<<<1, kBlock, 0, d->stream>>>(nb, sums, total);
        int nnz_near = 0;
        if (hipMemcpyAsync(&nnz_near, total, sizeof(int), hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) {
            (void) hipGetLastError();
            return bail(fail(SPMV_HIP_E_RUNTIME, "split: count failed"));
        }
        const long long nnz_far = d->nnz - nnz_near;
        for (int which = 0; which < 2; ++which) { // the two halves: all m rows each, arrays padded like any resident CSR
            spmv_dev *c = new spmv_dev();
            (which == 0 ? dn : df) = c;
            c->device = d->device; c->cus = d->cus; c->m = m; c->n = d->n; c->vsize = d->vsize; c->stream = d->stream; c->async = 1;
            c->col_min = d->col_min; c->col_max = d->col_max;
            const size_t nz = (size_t) (which == 0 ? nnz_near : nnz_far);
            int rc = dev_alloc(c, (void **) &c->rowptr, sizeof(int) * ((size_t) m + 1), false);
            if (!rc) rc = dev_alloc(c, (void **) &c->colidx, sizeof(int) * (nz + kStreamPad), false);
            if (!rc) rc = dev_alloc(c, &c->val, d->vsize * (nz + kStreamPad), false);
            if (rc) return bail(rc);
            (void) hipMemsetAsync(c->colidx + nz, 0, sizeof(int) * kStreamPad, d->stream);
            (void) hipMemsetAsync((char *) c->val + d->vsize * nz, 0, d->vsize * kStreamPad, d->stream);
        }
        scan_apply_kernel<<<nb, kBlock, 0, d->stream>>>(m, cnt, sums, dn->rowptr, d->rowptr, df->rowptr);
        const int last[2] = {nnz_near, (int) nnz_far};
        if (hipMemcpyAsync(dn->rowptr + m, &last[0], sizeof(int), hipMemcpyHostToDevice, d->stream) != hipSuccess ||
            hipMemcpyAsync(df->rowptr + m, &last[1], sizeof(int), hipMemcpyHostToDevice, d->stream) != hipSuccess) { (void) hipGetLastError(); return bail(fail(SPMV_HIP_E_RUNTIME, "split: row pointers")); }
    }
    split_scatter_kernel<T><<<grid_for(m, kBlock / 16, d->cus * 16), kBlock, 0, d->stream>>>(m, d->rowptr, d->colidx, (const T *) d->val, d->sp_centre, d->sp_half, dn->rowptr, df->rowptr,
                                                                                         values_only ? nullptr : dn->colidx, (T *) dn->val, values_only ? nullptr : df->colidx, (T *) df->val);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e != hipSuccess) return bail(fail(SPMV_HIP_E_RUNTIME, "split: scatter: %s", hipGetErrorString(e)));
    cleanup();
    if (!values_only) {
        int rc = matrix_row_stats(dn);
        if (!rc) rc = matrix_row_stats(df);
        if (rc) { spmv_shim_matrix_destroy(dn); spmv_shim_matrix_destroy(df); return rc; }
        df->accumulate = true;
        *near_out = dn;
        *far_out = df;
    }
    return SPMV_HIP_OK;
}

// Make the two halves (matrices on the parent's device and stream, unplanned).  The caller plans + builds them and either attaches
// them (spmv_shim_attach_split) or destroys them.
extern "C" int spmv_shim_split(spmv_dev *d, spmv_dev **near_out, spmv_dev **far_out)
{
    if (!d || !near_out || !far_out) return fail(SPMV_HIP_E_ARG, "split: NULL");
    *near_out = *far_out = nullptr;
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    return d->vsize == sizeof(double) ? split_make<double>(d, near_out, far_out, false) : split_make<float>(d, near_out, far_out, false);
}

// Hand the halves to the parent (it owns them from now on; its own schedule's products are released, the resident CSR stays: values
// are refreshed from it) -- or, with NULLs, take them away again and destroy them.
extern "C" int spmv_shim_attach_split(spmv_dev *d, spmv_dev *near_dev, spmv_dev *far_dev, int release_parent_schedule)
{
    if (!d) return fail(SPMV_HIP_E_ARG, "attach_split: NULL");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    quiesce(d);
    if (d->sp_near && d->sp_near != near_dev) spmv_shim_matrix_destroy(d->sp_near);
    if (d->sp_far && d->sp_far != far_dev) spmv_shim_matrix_destroy(d->sp_far);
    d->sp_near = near_dev;
    d->sp_far = far_dev;
    if (near_dev && far_dev && release_parent_schedule) { // the parent's own executor products are dead weight now
        for (auto &a : d->sched_allocs) { (void) pool_free(a.first); d->device_bytes -= (long long) a.second; }
        d->sched_allocs.clear();
        reset_tile_fields(d);
        d->blk_on = false;
        d->blk = BlkSet();
        d->stream_bytes = near_dev->stream_bytes + far_dev->stream_bytes;
        d->x_bytes = near_dev->x_bytes + far_dev->x_bytes;
    }
    return SPMV_HIP_OK;
}

extern "C" void spmv_shim_note_split_ms(spmv_dev *d, double as_built_ms, double split_ms)
{
    if (!d) return;
    d->split_ms[0] = (float) as_built_ms;
    d->split_ms[1] = (float) split_ms;
}
