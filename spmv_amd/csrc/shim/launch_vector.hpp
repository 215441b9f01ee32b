// shim/launch_vector.hpp -- the launchers of the CSR-vector family (CSR-vector's tile and pipe forms, the rows kernel of
// Balanced and of CSR-vector's wide form), compiled in their own translation unit (spmv_vector.hip): seven lanes-per-row
// values x five tile forms x two value types are most of the library's device code, and the two units build side by side.
#pragma once

// One workgroup per kVecNB * (256/L) consecutive rows, dispatched in row order: measured on the
// config-2 shape a plain in-order grid beats a persistent grid-stride loop by ~10 %.
constexpr int kVecNB = 4;
template <typename T, int L, int DEPTH, bool PRE = true>
static void launch_vector_tile(spmv_dev *d, const T *x, T *y, int long_thr)
{
    const size_t lds = ((((size_t) d->vt_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
    ensure_lds<csr_vector_tile_kernel<T, L, DEPTH, PRE>>(d, lds);
    csr_vector_tile_kernel<T, L, DEPTH, PRE><<<d->vt_tiles, kVecTileThreads, lds, d->stream>>>(d->m, long_thr, d->rowptr, d->colidx, d->vt_col, (const T *) d->val,
                                                                                           d->vt_wins, d->vt_rowslot, d->vt_col8, d->vt_tmpl, d->vt_rowtid, x, y);
}

template <typename T, int L>
static void launch_vector(spmv_dev *d, const T *x, T *y)
{
    const int v = d->plan.vector_form ? d->plan.vector_form : d->vec_choice;
    const int long_thr = d->long_thr;
    const bool tile_default = d->vt_staged * 2 >= d->vt_tiles; // most x tiles fit LDS
    const bool tile_forced = v == VEC_TILE_D2 || v == VEC_TILE_D4 || v == VEC_TILE_D8 || v == VEC_TILE_D4_NOPRE || v == VEC_TILE_D2_NOPRE;
    if (d->vt_tiles > 0 && v != VEC_PIPE && (tile_default || tile_forced)) { // tile kernel (unstaged tiles gather from L1/L2)
        if (v == VEC_TILE_D2) launch_vector_tile<T, L, 2>(d, x, y, long_thr);
        else if (v == VEC_TILE_D8) launch_vector_tile<T, L, 8>(d, x, y, long_thr);
        else if (v == VEC_TILE_D4) launch_vector_tile<T, L, 4>(d, x, y, long_thr);
        else if (v == VEC_TILE_D4_NOPRE) launch_vector_tile<T, L, 4, false>(d, x, y, long_thr);
        else if (v == VEC_TILE_D2_NOPRE) launch_vector_tile<T, L, 2, false>(d, x, y, long_thr);
        else launch_vector_tile<T, L, (sizeof(T) == 8 ? 4 : 2)>(d, x, y, long_thr); // measured default
        return;
    }
    constexpr int rows = kBlock / L * kVecNB;
    const int grid = grid_for(d->m, rows, INT_MAX);
    csr_vector_pipe_kernel<T, L, kVecNB><<<grid, kBlock, 0, d->stream>>>(d->m, long_thr, d->rowptr, d->colidx, (const T *) d->val, x, y);
}

template <typename T>
void launch_vector_any(spmv_dev *d, const T *x, T *y)
{
    switch (d->plan.lanes_per_row) {
    case 1: launch_vector<T, 1>(d, x, y); break;
    case 2: launch_vector<T, 2>(d, x, y); break;
    case 4: launch_vector<T, 4>(d, x, y); break;
    case 8: launch_vector<T, 8>(d, x, y); break;
    case 16: launch_vector<T, 16>(d, x, y); break;
    case 32: launch_vector<T, 32>(d, x, y); break;
    default: launch_vector<T, 64>(d, x, y); break;
    }
}

// The rows kernel: Balanced's equal-nnz row blocks (split), or CSR-vector's wide form (uniform blocks).
template <typename T, int L, int DEPTH>
static void launch_rows_depth(spmv_dev *d, const T *x, T *y, const int *split)
{
    const size_t lds = ((((size_t) d->vt_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
    if (d->vt_wide) {
        ensure_lds<csr_vector_rows_kernel<T, L, DEPTH, true>>(d, lds);
        csr_vector_rows_kernel<T, L, DEPTH, true><<<d->vt_tiles, kVecTileThreads, lds, d->stream>>>(
            d->long_thr, split, d->vt_rows, d->m, d->rowptr, d->colidx, d->vt_col, (const T *) d->val, d->vt_wins, d->vt_rowslot, d->vt_col8, d->vt_tmpl, d->vt_rowtid, x, y);
        return;
    }
    ensure_lds<csr_vector_rows_kernel<T, L, DEPTH, false>>(d, lds);
    csr_vector_rows_kernel<T, L, DEPTH><<<d->vt_tiles, kVecTileThreads, lds, d->stream>>>(
        d->long_thr, split, d->vt_rows, d->m, d->rowptr, d->colidx, d->vt_col, (const T *) d->val, d->vt_wins, d->vt_rowslot, d->vt_col8, d->vt_tmpl, d->vt_rowtid, x, y);
}

// Steps in flight: what create() measured (rows_depth, autotune_rows), what option vector_form names (5 / 12: two, 10 / 11: four), else the dtype's default.
template <typename T, int L>
static void launch_rows(spmv_dev *d, const T *x, T *y, const int *split)
{
    const int v = d->plan.vector_form;
    int depth = d->rows_depth ? d->rows_depth : (sizeof(T) == 8 ? 4 : 2);
    if (v == VEC_TILE_D2 || v == VEC_TILE_D2_NOPRE) depth = 2;
    if (v == VEC_TILE_D4 || v == VEC_TILE_D4_NOPRE) depth = 4;
    if (depth == 2) launch_rows_depth<T, L, 2>(d, x, y, split);
    else launch_rows_depth<T, L, 4>(d, x, y, split);
}

template <typename T>
void launch_rows_any(spmv_dev *d, const T *x, T *y, const int *split)
{
    switch (d->plan.lanes_per_row) {
    case 1: launch_rows<T, 1>(d, x, y, split); break;
    case 2: launch_rows<T, 2>(d, x, y, split); break;
    case 4: launch_rows<T, 4>(d, x, y, split); break;
    case 8: launch_rows<T, 8>(d, x, y, split); break;
    case 16: launch_rows<T, 16>(d, x, y, split); break;
    case 32: launch_rows<T, 32>(d, x, y, split); break;
    default: launch_rows<T, 64>(d, x, y, split); break;
    }
}


// explicit instantiations, dealt over four objects (build.py compiles spmv_vector.hip once per SPMV_VEC_PART, side by side)
#ifndef SPMV_VEC_PART
#error "spmv_vector.hip is compiled with -DSPMV_VEC_PART=0..3"
#endif
#if SPMV_VEC_PART == 0
template void launch_vector_any<double>(spmv_dev *, const double *, double *);
#elif SPMV_VEC_PART == 1
template void launch_vector_any<float>(spmv_dev *, const float *, float *);
#elif SPMV_VEC_PART == 2
template void launch_rows_any<double>(spmv_dev *, const double *, double *, const int *);
#else
template void launch_rows_any<float>(spmv_dev *, const float *, float *, const int *);
#endif
