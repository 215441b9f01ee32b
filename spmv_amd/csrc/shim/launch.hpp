// shim/launch.hpp -- part of the single translation unit spmv_shim.hip: the EXECUTORS' launchers (which
// kernel form, which template arguments, grid and LDS size per schedule), the create-time autotune of the
// CSR-vector forms, and launch<T>() = one y = A x on the handle's stream.
#pragma once

// ------------------------------------------------------------------------------------ executors
// The CSR-vector family's launchers (tile / pipe / rows kernels for every lanes-per-row value: the bulk of the library's device
// code) live in their own translation unit, spmv_vector.hip (shim/launch_vector.hpp), compiled beside this one.
#include "vector_forms.hpp"
template <typename T> void launch_vector_any(spmv_dev *d, const T *x, T *y);
template <typename T> void launch_rows_any(spmv_dev *d, const T *x, T *y, const int *split);
extern template void launch_vector_any<double>(spmv_dev *, const double *, double *);
extern template void launch_vector_any<float>(spmv_dev *, const float *, float *);
extern template void launch_rows_any<double>(spmv_dev *, const double *, double *, const int *);
extern template void launch_rows_any<float>(spmv_dev *, const float *, float *, const int *);

// Time the applicable CSR-vector forms on the resident matrix (x = 1) and keep the fastest.
template <typename T>
static int autotune_vector(spmv_dev *d)
{
    d->vec_choice = VEC_AUTO;
    if (d->nnz < (1ll << 24) || d->plan.forced || d->vt_tiles <= 0) return SPMV_HIP_OK;
    if (d->vt_wide || d->vt_staged * 2 < d->vt_tiles) return SPMV_HIP_OK; // wide form: one kernel form; x windows not staged: the pipe form runs, nothing to choose (and 45 gather-bound launches would cost ~0.3 s)
    T *x = nullptr, *y = nullptr;
    if (pool_malloc((void **) &x, sizeof(T) * (size_t) d->n) != hipSuccess || pool_malloc((void **) &y, sizeof(T) * (size_t) d->m) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) pool_free(x);
        return SPMV_HIP_OK; // no room to tune: keep the default
    }
    fill_value_kernel<T><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, x, T(1));
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    constexpr int kCand = 5;
    const int cand[kCand] = {VEC_TILE_D4, VEC_TILE_D4_NOPRE, VEC_TILE_D2, VEC_TILE_D2_NOPRE, VEC_PIPE};
    // (nearly) every tile staged: the pipe form (4-byte columns, gathers through L1 / L2) has nothing to win and is not timed -- on config 2 its nine
    // launches were a quarter of the inspector's 26 ms
    const int ncand = (long long) d->vt_staged * 100 >= (long long) d->vt_tiles * 99 ? kCand - 1 : kCand;
    float tmin[kCand];
    for (int k = 0; k < kCand; ++k) tmin[k] = 1e30f;
    for (int k = 0; k < ncand; ++k) { d->vec_choice = cand[k]; launch_vector_any<T>(d, x, y); } // warm every form once
    for (int round = 0; round < 3; ++round) // interleaved rounds (one process, same clocks): min per form
        for (int k = 0; k < ncand; ++k) {
            d->vec_choice = cand[k];
            (void) hipEventRecord(e0, d->stream);
            launch_vector_any<T>(d, x, y); // one launch per sample (round 3: two): the forms differ by whole percents where they differ, events resolve a microsecond
            (void) hipEventRecord(e1, d->stream);
            (void) hipEventSynchronize(e1);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, e0, e1);
            if (ms < tmin[k]) tmin[k] = ms;
        }
    float best = 1e30f;
    int best_c = VEC_AUTO;
    for (int k = 0; k < kCand - 1; ++k) {
        if (tmin[k] < best) { best = tmin[k]; best_c = cand[k]; }
    }
    // the pipe form (int32 columns, global gathers) only on a clear win: a noisy sample -- e.g. another
    // process on the device during create -- must not cost 30 % on every later launch
    if (ncand == kCand && tmin[kCand - 1] < 0.95f * best) { best = tmin[kCand - 1]; best_c = VEC_PIPE; }
    d->tune_ms[0] = tmin[0] < tmin[1] ? tmin[0] : tmin[1]; // tile, 4 steps in flight (best of the two issue orders)
    d->tune_ms[1] = tmin[2] < tmin[3] ? tmin[2] : tmin[3]; // tile, 2 steps in flight
    d->tune_ms[2] = ncand == kCand ? tmin[4] : 0.f;        // pipe (0: not timed)
    d->vec_choice = best_c;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) pool_free(x);
    (void) pool_free(y);
    if (hipGetLastError() != hipSuccess) d->vec_choice = VEC_AUTO;
    return SPMV_HIP_OK;
}

// Row blocks x column slabs (kernels/blocked.hpp).  One-wave form: a single-wave workgroup per row block, two per CU; wide form
// (S.waves = 4 / 8): one workgroup of that many waves per block, one block per CU, the waves taking turns at adding (S.ordered) or
// not.  Forms = groups per pipeline step (S.form 0 / 1, chosen by autotune_blocked; option blk_groups forces one): 8 / 12 with one
// or four waves, 6 / 8 with eight (256 registers per wave).  All forms of one layout add the same products; the one-wave forms and
// the ordered wide forms each in a fixed order (tests compare their bits).
static void blocked_forms(const BlkSet &S, int un[2])
{
    un[0] = S.waves == 8 ? 6 : 8;
    un[1] = S.waves == 8 ? 8 : 12;
}

template <typename T>
static void launch_blocked(spmv_dev *d, const T *x, T *y)
{
    const BlkSet &S = d->blk;
    const size_t lds = blocked_lds_bytes(S);
#define SPMV_BLK_LAUNCH(UN, DBG)                                                                                                  \
    do {                                                                                                                          \
        ensure_lds<blk_kernel<T, UN, DBG>>(d, lds);                                                                               \
        blk_kernel<T, UN, DBG><<<S.B, kWave, lds, d->stream>>>(S.row0, S.R, S.dir, (const T *) S.val, S.meta, S.hdr, S.order, x, y, d->accumulate ? 1 : 0); \
    } while (0)
#define SPMV_BLK_WLAUNCH(UN, W, ORD)                                                                                              \
    do {                                                                                                                          \
        ensure_lds<blk_wide_kernel<T, UN, W, ORD>>(d, lds);                                                                       \
        blk_wide_kernel<T, UN, W, ORD><<<S.B, kWave * W, lds, d->stream>>>(S.row0, S.R, S.dir, (const T *) S.val, S.meta, S.hdr, S.order, x, y, d->accumulate ? 1 : 0); \
    } while (0)
#define SPMV_BLK_WFORM(UN, W) do { if (S.ordered) SPMV_BLK_WLAUNCH(UN, W, true); else SPMV_BLK_WLAUNCH(UN, W, false); } while (0)
#ifdef SPMV_BLK_DEBUG_FORMS // A/B builds of tools/ only (wrong results): SPMV_HIP_BLK_DEBUG_FORM = 1 / 2 / 3 = no gathers / no LDS adds / neither, 8 groups per step
    static const int dbg_form = getenv("SPMV_HIP_BLK_DEBUG_FORM") ? atoi(getenv("SPMV_HIP_BLK_DEBUG_FORM")) : 0;
    if (dbg_form == 1) { SPMV_BLK_LAUNCH(8, 1); return; }
    if (dbg_form == 2) { SPMV_BLK_LAUNCH(8, 2); return; }
    if (dbg_form == 3) { SPMV_BLK_LAUNCH(8, 3); return; }
#endif
    int un[2];
    blocked_forms(S, un);
    const int form = S.form;
    const int g = d->plan.blk_groups > 0 ? d->plan.blk_groups : un[form];
    if (S.waves == 2) {
        if (g == 12) SPMV_BLK_WFORM(12, 2); else SPMV_BLK_WFORM(8, 2);
    } else if (S.waves == 4) {
        if (g == 12) SPMV_BLK_WFORM(12, 4); else if (g == 6) SPMV_BLK_WFORM(6, 4); else if (g == 4) SPMV_BLK_WFORM(4, 4); else SPMV_BLK_WFORM(8, 4);
    } else if (S.waves == 8) {
        if (g == 8) SPMV_BLK_WFORM(8, 8); else if (g == 4) SPMV_BLK_WFORM(4, 8); else SPMV_BLK_WFORM(6, 8);
    } else {
        if (g == 12) SPMV_BLK_LAUNCH(12, 0);
        else SPMV_BLK_LAUNCH(8, 0);
    }
#undef SPMV_BLK_WFORM
#undef SPMV_BLK_LAUNCH
#undef SPMV_BLK_WLAUNCH
}

// Time the executor forms on the resident streams (x = 1: the gather pattern does not depend on the values) and keep the faster.
template <typename T>
static int autotune_blocked(spmv_dev *d)
{
    d->blk.form = 0;
    d->blk.tune_ms[0] = d->blk.tune_ms[1] = d->blk.tune_ms[2] = 0;
    if (!d->blk_on || !d->plan.autotune || d->plan.blk_groups != 0) return SPMV_HIP_OK;
    T *x = nullptr, *y = nullptr;
    if (pool_malloc((void **) &x, sizeof(T) * (size_t) d->n) != hipSuccess || pool_malloc((void **) &y, sizeof(T) * (size_t) d->m) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) pool_free(x);
        return SPMV_HIP_OK; // no room to tune: keep the default
    }
    fill_value_kernel<T><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, x, T(1));
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    constexpr int kForms = 2;
    float tmin[kForms] = {1e30f, 1e30f};
    for (int f = 0; f < kForms; ++f) { d->blk.form = f; launch_blocked<T>(d, x, y); } // warm every form once
    for (int round = 0; round < 4; ++round) // interleaved rounds: min per form
        for (int f = 0; f < kForms; ++f) {
            d->blk.form = f;
            (void) hipEventRecord(e0, d->stream);
            launch_blocked<T>(d, x, y);
            (void) hipEventRecord(e1, d->stream);
            (void) hipEventSynchronize(e1);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, e0, e1);
            if (ms < tmin[f]) tmin[f] = ms;
        }
    int best = 0;
    for (int f = 0; f < kForms; ++f) { d->blk.tune_ms[f] = tmin[f]; if (tmin[f] < tmin[best]) best = f; }
    d->blk.form = best;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) pool_free(x);
    (void) pool_free(y);
    if (hipGetLastError() != hipSuccess) d->blk.form = 0;
    return SPMV_HIP_OK;
}

// The rows kernel (Balanced's row blocks with `split`, CSR-vector's wide form without): two or four steps in flight?  Timed like the tile forms, on matrices of
// >= 2^24 entries whose blocks mostly stage (27-point stencil under Method_Balanced: 0.55 ms four deep, the fp64 default, 0.47 two deep).
template <typename T>
static int autotune_rows(spmv_dev *d, const int *split)
{
    d->rows_depth = 0;
    if (d->nnz < (1ll << 24) || d->plan.forced || d->vt_tiles <= 0 || d->vt_staged * 2 < d->vt_tiles) return SPMV_HIP_OK;
    T *x = nullptr, *y = nullptr;
    if (pool_malloc((void **) &x, sizeof(T) * (size_t) d->n) != hipSuccess || pool_malloc((void **) &y, sizeof(T) * (size_t) d->m) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) pool_free(x);
        return SPMV_HIP_OK;
    }
    fill_value_kernel<T><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, x, T(1));
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    const int depth[2] = {4, 2};
    float tmin[2] = {1e30f, 1e30f};
    for (int k = 0; k < 2; ++k) { d->rows_depth = depth[k]; launch_rows_any<T>(d, x, y, split); }
    for (int round = 0; round < 2; ++round)
        for (int k = 0; k < 2; ++k) {
            d->rows_depth = depth[k];
            (void) hipEventRecord(e0, d->stream);
            launch_rows_any<T>(d, x, y, split);
            (void) hipEventRecord(e1, d->stream);
            (void) hipEventSynchronize(e1);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, e0, e1);
            if (ms < tmin[k]) tmin[k] = ms;
        }
    d->tune_ms[0] = tmin[0];
    d->tune_ms[1] = tmin[1];
    d->rows_depth = tmin[1] < tmin[0] ? 2 : 4;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) pool_free(x);
    (void) pool_free(y);
    if (hipGetLastError() != hipSuccess) d->rows_depth = 0;
    return SPMV_HIP_OK;
}

// The staged CSR5 group kernel two tiles deep (csr5_group_pipe_kernel)?  When EVERY group is staged (the pipelined kernel has no
// global-column path) and the values are fp32: in fp64 the second register set costs a wave per SIMD and measured no gain (config 2
// under CSR5: 0.511 vs 0.515 ms).  Option csr5_two_deep 1 / 2: never / also for fp64 (A/B).
static bool csr5_two_deep(const spmv_dev *d, const Csr5Plan &P)
{
    if (P.natural || P.staged <= 0 || P.staged != P.groups || P.group_tiles > kCsr5PipeMaxGroupTiles || d->plan.csr5_two_deep == 1) return false;
    return d->vsize == sizeof(float) || d->plan.csr5_two_deep == 2;
}

template <typename T, int SIGMA, bool MAPPED>
static void launch_csr5_form(spmv_dev *d, const Csr5Plan &P, const T *x, T *y)
{
    // MAPPED (matrices with empty rows, long-row sub-matrices): every wave stages its tile's row map in LDS, behind the x
    // windows in the dynamic segment.  A tile of long rows holds few row starts: kWave ints per wave then (csr5.hpp)
    const int rm_stride = MAPPED ? (P.max_tile_rows < kWave ? kWave : (SIGMA + 1) * kWave) : 0;
    const size_t rmb = (size_t) (kBlock / kWave) * rm_stride * sizeof(int);
    if (P.staged > 0) { // the inspector staged (at least half of) the groups: their column stream is the 16-bit slot array
        const size_t lds = ((((size_t) P.maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
        if (P.natural) {
            // the waves' tile buffers are static LDS next to the x windows: if the full-size buffers would leave
            // one workgroup per CU, hand the tiles over in two halves (half the buffers)
            constexpr size_t full = (size_t) (kBlock / kWave) * NatLds<T, SIGMA, false>::kBytes;
            const bool half = lds + rmb + full > 76 * 1024; // two workgroups no longer fit a CU's 160 KiB
            if (half) {
                ensure_lds<nat_group_kernel<T, SIGMA, MAPPED, true>>(d, lds + rmb, (size_t) (kBlock / kWave) * NatLds<T, SIGMA, true>::kBytes);
                nat_group_kernel<T, SIGMA, MAPPED, true><<<P.groups, kBlock, lds + rmb, d->stream>>>(P.group_tiles, P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, P.col16,
                                                                                                    (const T *) P.val, P.row_map, P.wins, x, y, (T *) P.carry, P.n_empty, P.empty_list,
                                                                                                    (int) lds, rm_stride);
            } else {
                ensure_lds<nat_group_kernel<T, SIGMA, MAPPED, false>>(d, lds + rmb, full);
                nat_group_kernel<T, SIGMA, MAPPED, false><<<P.groups, kBlock, lds + rmb, d->stream>>>(P.group_tiles, P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, P.col16,
                                                                                                     (const T *) P.val, P.row_map, P.wins, x, y, (T *) P.carry, P.n_empty, P.empty_list,
                                                                                                     (int) lds, rm_stride);
            }
            return;
        }
        if (csr5_two_deep(d, P)) {
            ensure_lds<csr5_group_pipe_kernel<T, SIGMA, MAPPED>>(d, lds + rmb);
            csr5_group_pipe_kernel<T, SIGMA, MAPPED><<<P.groups, kBlock, lds + rmb, d->stream>>>(P.group_tiles, P.tiles, P.tile_ptr, P.desc, P.col16, (const T *) P.val, P.row_map, P.wins, P.lane_run,
                                                                                                x, y, (T *) P.carry, P.n_empty, P.empty_list, (int) lds, rm_stride);
            return;
        }
        ensure_lds<csr5_group_kernel<T, SIGMA, MAPPED>>(d, lds + rmb);
        csr5_group_kernel<T, SIGMA, MAPPED><<<P.groups, kBlock, lds + rmb, d->stream>>>(P.group_tiles, P.tiles, P.tile_ptr, P.desc, P.col, P.col16, (const T *) P.val, P.row_map, P.wins, P.lane_run,
                                                                                       x, y, (T *) P.carry, P.n_empty, P.empty_list, (int) lds, rm_stride);
        return;
    }
    const int grid = grid_for(P.tiles, kBlock / kWave, INT_MAX);
    const int xcd = d->plan.xcd_order ? 1 : 0; // XCD-aware block order (common.hpp: xcd_block); option xcd_order = 0: dispatch order, for A/B
    if (P.natural && P.forward) {
        const int long_blocks = (P.n_long + 7) & ~7;
        nat_kernel<T, SIGMA, MAPPED, true><<<grid + long_blocks, kBlock, rmb, d->stream>>>(P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, (const T *) P.val, P.row_map, x, y, (T *) P.carry,
                                                                                          P.n_empty, P.empty_list, rm_stride, xcd, P.fwd, P.long_list, P.n_long, long_blocks);
    } else if (P.natural)
        nat_kernel<T, SIGMA, MAPPED><<<grid, kBlock, rmb, d->stream>>>(P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, (const T *) P.val, P.row_map, x, y, (T *) P.carry, P.n_empty,
                                                                      P.empty_list, rm_stride, xcd);
    else
        csr5_kernel<T, SIGMA, MAPPED><<<grid, kBlock, rmb, d->stream>>>(P.tiles, P.tile_ptr, P.desc, P.col, (const T *) P.val, P.row_map, x, y, (T *) P.carry, P.n_empty, P.empty_list,
                                                                       rm_stride, xcd);
}

template <typename T, int SIGMA>
static void launch_csr5_sigma(spmv_dev *d, const Csr5Plan &P, const T *x, T *y)
{
    if (P.row_map) launch_csr5_form<T, SIGMA, true>(d, P, x, y);
    else launch_csr5_form<T, SIGMA, false>(d, P, x, y);
}

// One CSR5 multiply: [y = 0 for the rows outside the plan] + tiles + carry fix-up.
template <typename T>
static int launch_csr5(spmv_dev *d, const Csr5Plan &P, const T *x, T *y)
{
    if (P.nnz == 0) return SPMV_HIP_OK;
    switch (P.sigma) {
    case 4: launch_csr5_sigma<T, 4>(d, P, x, y); break;
    case 8: launch_csr5_sigma<T, 8>(d, P, x, y); break;
    default: launch_csr5_sigma<T, 16>(d, P, x, y); break;
    }
    if (P.fixup && P.tiles > 1 && !P.forward) {
        const int g = grid_for(P.tiles - 1, kBlock, INT_MAX);
        if (P.row_map) csr5_fixup_kernel<T, true><<<g, kBlock, 0, d->stream>>>(P.tiles, P.tile_ptr, P.run_len, P.row_map, (const T *) P.carry, y);
        else csr5_fixup_kernel<T, false><<<g, kBlock, 0, d->stream>>>(P.tiles, P.tile_ptr, P.run_len, nullptr, (const T *) P.carry, y);
    }
    return SPMV_HIP_OK;
}

template <typename T>
static int launch(spmv_dev *d, const T *x, T *y)
{
    if (d->m == 0) return SPMV_HIP_OK;
    if (d->sp_near && d->sp_far) { // A = A_near + A_far (shim/split.hpp): the tile schedule writes every row, the blocked executor adds its part
        d->sp_near->stream = d->sp_far->stream = d->stream;
        const int rc = launch<T>(d->sp_near, x, y);
        return rc ? rc : launch<T>(d->sp_far, x, y);
    }
    if (d->nnz == 0) { // nothing to multiply: y = 0 (the far half of a split: y += 0)
        if (d->accumulate) return SPMV_HIP_OK;
        fill_zero_kernel<T><<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, y);
        HIP_TRY(hipGetLastError());
        return SPMV_HIP_OK;
    }
    // y += A x exists in the blocked executor only: a far half that ever ended on another executor would overwrite A_near x
    if (d->accumulate && !d->blk_on) return fail(SPMV_HIP_E_RUNTIME, "internal: the accumulating half of a split matrix has no blocked executor");
    const T *val = (const T *) d->val;
    switch (d->plan.sched) {
    case SPMV_SCHED_CSR_SCALAR:
        csr_scalar_kernel<T><<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, d->rowptr, d->colidx, val, x, y);
        break;
    case SPMV_SCHED_CSR_VECTOR:
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; } // no x window could be staged (option "cache_block")
        if (d->vt_wide) launch_rows_any<T>(d, x, y, nullptr); // wide x windows: uniform 1024-row blocks, slot-index stream
        else launch_vector_any<T>(d, x, y);
        launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_NNZ_SPLIT: {
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        const int rc = launch_csr5<T>(d, d->ns, x, y);
        if (rc) return rc;
        break;
    }
    case SPMV_SCHED_ROWBLOCK:
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        launch_rows_any<T>(d, x, y, d->rb_split);
        launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_SELL:
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        // staged path when at least half of the windows fit their x span in LDS; the LDS request is
        // sized by the largest staged span actually present (rounded to 16 KiB) to keep occupancy
        if (d->sell_staged > 0) {
            const int cpw = d->sell_group * (d->plan.sell_sigma / kSellC);
            const size_t xbytes = ((((size_t) d->sell_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // x windows + the zero slot
            const size_t lds = xbytes + sizeof(T) * (size_t) cpw * kSellC;                                  // + the group's row sums
            ensure_lds<sell_window_kernel<T>>(d, lds);
            sell_window_kernel<T><<<d->sell_nwin, kSellWinThreads, lds, d->stream>>>(cpw, (long long) d->nchunks, d->m, d->chunk_ptr, d->scol, d->scol16, (const T *) d->sval,
                                                                                     d->perm, d->sell_wins, d->sell_run, d->sell_tmpl, d->scol8, x, y, (int) xbytes);
        }
        else
            sell_kernel<T><<<grid_for(d->nchunks, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>(
                d->nchunks, d->chunk_ptr, d->scol, (const T *) d->sval, d->perm, x, y);
        launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_CSR5: {
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        const int rc = launch_csr5<T>(d, d->c5, x, y);
        if (rc) return rc;
        break;
    }
    default: return fail(SPMV_HIP_E_ARG, "schedule %d has no executor", d->plan.sched);
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

// min over `iters` launches of the schedule as built, on scratch vectors (x = 1), in ms; < 0 if it cannot be timed
template <typename T>
static double time_schedule(spmv_dev *d, int iters)
{
    T *x = nullptr, *y = nullptr;
    if (pool_malloc((void **) &x, sizeof(T) * (size_t) (d->n > 0 ? d->n : 1)) != hipSuccess || pool_malloc((void **) &y, sizeof(T) * (size_t) (d->m > 0 ? d->m : 1)) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) pool_free(x);
        return -1.0;
    }
    fill_value_kernel<T><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, x, T(1));
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    float best = 1e30f;
    int rc = launch<T>(d, x, y); // warm
    for (int i = 0; i < iters && !rc; ++i) {
        (void) hipEventRecord(e0, d->stream);
        rc = launch<T>(d, x, y);
        (void) hipEventRecord(e1, d->stream);
        (void) hipEventSynchronize(e1);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) pool_free(x);
    (void) pool_free(y);
    if (rc || hipGetLastError() != hipSuccess) return -1.0;
    return (double) best;
}
