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
#include <vector>

#include "spmv_shim.h"
#include "kernels/common.hpp"
#include "kernels/csr_rows.hpp"
#include "kernels/csr_vector4.hpp"
#include "kernels/nnz_split.hpp"
#include "kernels/rowblock.hpp"
#include "kernels/sell.hpp"
#include "kernels/csr5.hpp"
#include "kernels/long_rows.hpp"
#include "kernels/blocked.hpp"
#include "kernels/csr_vector_tile.hpp"

using namespace spmv;

// ------------------------------------------------------------------------------------ errors
static thread_local char t_err[400] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof t_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *spmv_shim_error_text(void) { return t_err; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (void) hipGetLastError();                                                              \
            return fail(e__ == hipErrorOutOfMemory ? SPMV_HIP_E_ALLOC : SPMV_HIP_E_RUNTIME,        \
                        "%s -> %s", #expr, hipGetErrorString(e__));                                \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------------------------ state
struct DevStats {
    int max_len, min_len, empty, bad, first, last;
    unsigned long long hist_rows[SPMV_LEN_BUCKETS], hist_nnz[SPMV_LEN_BUCKETS];
};

// One CSR5 instance (kernels/csr5.hpp): the whole matrix for Method_CSR5SPMV, or the sub-matrix of
// the long rows that CSR-vector / Balanced / SELL hand over (see build_long_rows).
struct Csr5Plan {
    int sigma = 0, tiles = 0, m2 = 0, fixup = 0, groups = 0, staged = 0, maxspan = 0, group_tiles = kCsr5GroupTiles;
    long long nnz = 0;
    int n_empty = 0;              // empty rows (outside row_map): the tile kernel zeroes y for them
    const int *empty_list = nullptr;
    bool natural = false;         // nnz-split: no transposed copies, col/val are the matrix's own arrays (kernels/csr5.hpp, nat_tile)
    TileWindows *wins = nullptr;
    int *tile_ptr = nullptr, *run_len = nullptr, *col = nullptr; // col: transposed global columns (freed when every group is staged)
    unsigned short *col16 = nullptr; // 16-bit LDS slots of the staged groups
    const int *row_map = nullptr; // CSR5 row -> y row (NULL: identity)
    unsigned *desc = nullptr;
    void *val = nullptr, *carry = nullptr;
};

struct spmv_dev {
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    int async = 0;
    int m = 0, n = 0;
    long long nnz = 0;
    size_t vsize = 8;
    // resident CSR
    int *rowptr = nullptr, *colidx = nullptr;
    void *val = nullptr;
    spmv_stats stats{};
    spmv_plan plan{};
    bool built = false;
    // nnz-split
    int ntiles = 0, need_fixup = 0;
    int *tile_first = nullptr;
    void *carry = nullptr;
    int ns_groups = 0, ns_staged = 0, ns_maxspan = 0;
    unsigned short *ns_col = nullptr; // 16-bit LDS slots of the staged groups' entries
    TileWindows *ns_wins = nullptr;
    // row blocks
    int nblocks = 0, rb_stride = 0;
    int *rb_split = nullptr;
    // csr-vector x tiles
    int vt_tiles = 0, vt_staged = 0, vt_maxspan = 0, vec_choice = 0;
    float tune_ms[3] = {0, 0, 0}; // tile D4, tile D2, pipe (autotune_vector)
    unsigned short *vt_col = nullptr; // tile-local column stream: 16-bit LDS slots (staged tiles only)
    TileWindows *vt_wins = nullptr; // x windows of every tile
    // long rows (csr-vector, sell)
    int nlong = 0, long_thr = INT_MAX, lr_segs = 0, lr_maxspan = 0;
    int *long_rows = nullptr, *lr_seg_lr = nullptr, *lr_seg_lo = nullptr, *lr_seg_span = nullptr;
    long long *lr_seg_start = nullptr;
    void *lr_part = nullptr;
    // sell
    int nchunks = 0;
    long long sell_cols = 0; // sum of chunk widths
    int *perm = nullptr, *scol = nullptr;
    TileWindows *sell_wins = nullptr;
    unsigned short *scol16 = nullptr; // 16-bit LDS slots of the staged sigma windows
    int sell_nwin = 0, sell_staged = 0, sell_xcap = 0, sell_maxspan = 0, sell_group = 1; // windows, windows with x staged in LDS, LDS capacity in elements
    long long *chunk_ptr = nullptr;
    void *sval = nullptr;
    // csr5
    Csr5Plan c5, c5_long, ns; // ns: the natural-layout plan of the nnz-split schedule
    // row blocks x column slabs (kernels/blocked.hpp): the nnz-split executor for columns without locality
    bool blk_on = false;
    int blk_R = 0, blk_K = 0, blk_B = 0;
    long long *blk_start = nullptr, *blk_end = nullptr;
    void *blk_val = nullptr;
    int *blk_col = nullptr;
    unsigned short *blk_row = nullptr;
    // long-row sub-matrix (rows longer than long_thr, in row order), the input of c5_long
    int *lsub_rowptr = nullptr, *lsub_colidx = nullptr;
    void *lsub_val = nullptr;
    long long lsub_nnz = 0;
    // staging for host x / y
    void *x_stage = nullptr, *y_stage = nullptr;
    long long device_bytes = 0;
    double inspect_ms = 0;
    std::vector<void *> sched_allocs; // freed when the schedule is rebuilt
};

static int dev_alloc(spmv_dev *d, void **p, size_t bytes, bool sched)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc(p, bytes));
    d->device_bytes += (long long) bytes;
    if (sched) d->sched_allocs.push_back(*p);
    return SPMV_HIP_OK;
}
#define ALLOC_TRY(d, p, bytes, sched)                                        \
    do {                                                                     \
        int rc__ = dev_alloc((d), (void **) (p), (bytes), (sched));          \
        if (rc__) return rc__;                                               \
    } while (0)

// release one schedule-owned allocation early
static void sched_free(spmv_dev *d, void *p)
{
    for (size_t i = 0; i < d->sched_allocs.size(); ++i)
        if (d->sched_allocs[i] == p) {
            d->sched_allocs.erase(d->sched_allocs.begin() + (long) i);
            (void) hipFree(p);
            return;
        }
}

static void free_schedule(spmv_dev *d)
{
    for (void *p : d->sched_allocs) (void) hipFree(p);
    d->sched_allocs.clear();
    d->tile_first = nullptr; d->carry = nullptr; d->rb_split = nullptr; d->ns_col = nullptr; d->ns_wins = nullptr; d->ns_groups = d->ns_staged = 0;
    d->perm = d->scol = d->long_rows = d->lr_seg_lr = d->lr_seg_lo = d->lr_seg_span = nullptr; d->sell_wins = nullptr; d->scol16 = nullptr; d->sell_staged = d->sell_nwin = 0; d->chunk_ptr = d->lr_seg_start = nullptr;
    d->sval = d->lr_part = nullptr;
    d->ntiles = d->nblocks = d->nchunks = d->nlong = d->lr_segs = 0;
    d->long_thr = INT_MAX;
    d->vt_col = nullptr; d->vt_wins = nullptr; d->vt_tiles = d->vt_staged = d->vt_maxspan = 0;
    d->c5 = Csr5Plan();
    d->c5_long = Csr5Plan();
    d->ns = Csr5Plan();
    d->blk_on = false; d->blk_start = d->blk_end = nullptr; d->blk_val = nullptr; d->blk_col = nullptr; d->blk_row = nullptr;
    d->lsub_rowptr = d->lsub_colidx = nullptr; d->lsub_val = nullptr; d->lsub_nnz = 0;
    d->built = false;
}

// true if the pointer is usable by a kernel as is (device or managed memory)
static bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void) hipGetLastError(); // plain malloc memory: "invalid value", not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

extern "C" int spmv_shim_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return n;
}

// ------------------------------------------------------------------------------------ stats
__global__ __launch_bounds__(kBlock) void stats_kernel(int m, const int *__restrict__ rowptr, DevStats *s)
{
    __shared__ unsigned h_rows[SPMV_LEN_BUCKETS];
    __shared__ unsigned long long h_nnz[SPMV_LEN_BUCKETS];
    if (threadIdx.x < SPMV_LEN_BUCKETS) { h_rows[threadIdx.x] = 0; h_nnz[threadIdx.x] = 0; }
    __syncthreads();
    int mx = 0, mn = INT_MAX, em = 0, bad = 0;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride) {
        const int len = rowptr[r + 1] - rowptr[r];
        mx = max(mx, len);
        mn = min(mn, len);
        em += len == 0;
        bad |= len < 0;
        int b = len <= 4 ? 0 : 32 - __clz((len - 1) >> 2); // smallest b with len <= 4 * 2^b
        if (b > SPMV_LEN_BUCKETS - 1) b = SPMV_LEN_BUCKETS - 1;
        atomicAdd(&h_rows[b], 1u);
        atomicAdd(&h_nnz[b], (unsigned long long) (len > 0 ? len : 0));
    }
    __syncthreads();
    if (threadIdx.x < SPMV_LEN_BUCKETS && h_rows[threadIdx.x]) {
        atomicAdd(&s->hist_rows[threadIdx.x], (unsigned long long) h_rows[threadIdx.x]);
        atomicAdd(&s->hist_nnz[threadIdx.x], h_nnz[threadIdx.x]);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mx = max(mx, __shfl_xor(mx, o, kWave));
        mn = min(mn, __shfl_xor(mn, o, kWave));
        em += __shfl_xor(em, o, kWave);
        bad |= __shfl_xor(bad, o, kWave);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&s->max_len, mx);
        atomicMin(&s->min_len, mn);
        atomicAdd(&s->empty, em);
        if (bad) atomicOr(&s->bad, 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { s->first = rowptr[0]; s->last = rowptr[m]; }
}

__global__ __launch_bounds__(kBlock) void count_longer_kernel(int m, int thr, const int *__restrict__ rowptr, int *count)
{
    int c = 0;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride)
        c += (rowptr[r + 1] - rowptr[r]) > thr;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) c += __shfl_xor(c, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && c) atomicAdd(count, c);
}

// min / max of ColIdx (create-time validation: an index outside [0, n) would make a gather fault)
__global__ __launch_bounds__(kBlock) void colidx_range_kernel(long long nnz, const int *__restrict__ colidx, int *__restrict__ mnmx)
{
    int mn = INT_MAX, mx = INT_MIN;
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < nnz; i += stride) {
        const int c = ld_stream(colidx + i);
        mn = min(mn, c);
        mx = max(mx, c);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, kWave));
        mx = max(mx, __shfl_xor(mx, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicMin(mnmx, mn); atomicMax(mnmx + 1, mx); }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void fill_value_kernel(long long n, T *y, T v)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = v;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void fill_zero_kernel(long long n, T *y)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = T(0);
}

static int grid_for(long long work_items, int per_block, int cap)
{
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int) g;
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
    // row statistics (also validates RowPtr)
    DevStats hs{0, INT_MAX, 0, 0, 0, 0};
    DevStats *ds = nullptr;
    if (hipMalloc((void **) &ds, sizeof(DevStats)) != hipSuccess) return bail(fail(SPMV_HIP_E_ALLOC, "hipMalloc(stats)"));
    (void) hipMemcpy(ds, &hs, sizeof hs, hipMemcpyHostToDevice);
    if (m > 0) {
        stats_kernel<<<grid_for(m, kBlock, d->cus * 8), kBlock>>>(m, d->rowptr, ds);
        if (hipGetLastError() != hipSuccess) { (void) hipFree(ds); return bail(fail(SPMV_HIP_E_RUNTIME, "stats kernel launch failed")); }
    }
    hipError_t e = hipMemcpy(&hs, ds, sizeof hs, hipMemcpyDeviceToHost);
    (void) hipFree(ds);
    if (e != hipSuccess) return bail(fail(SPMV_HIP_E_RUNTIME, "stats kernel: %s", hipGetErrorString(e)));
    if (m > 0 && (hs.bad || hs.first != 0 || hs.last < 0))
        return bail(fail(SPMV_HIP_E_ARG, "RowPtr must start at 0 and be non-decreasing (RowPtr[0]=%d)", hs.first));
    d->nnz = m > 0 ? hs.last : 0;
    if (d->nnz > (long long) INT_MAX - 4096)
        return bail(fail(SPMV_HIP_E_RANGE, "nnz = %lld is within 4096 of INT_MAX: 16 B tail reads would overflow int32 indices", d->nnz));
    if (d->nnz > 0 && (!colidx || !val)) return bail(fail(SPMV_HIP_E_ARG, "ColIdx / Matrix_Val is NULL"));
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
        if (hipMalloc((void **) &mnmx, sizeof host2) != hipSuccess) return bail(fail(SPMV_HIP_E_ALLOC, "hipMalloc(colidx range)"));
        (void) hipMemcpy(mnmx, host2, sizeof host2, hipMemcpyHostToDevice);
        colidx_range_kernel<<<grid_for(d->nnz, kBlock * 16, d->cus * 8), kBlock>>>(d->nnz, d->colidx, mnmx);
        const hipError_t e2 = hipMemcpy(host2, mnmx, sizeof host2, hipMemcpyDeviceToHost);
        (void) hipFree(mnmx);
        if (e2 != hipSuccess) return bail(fail(SPMV_HIP_E_RUNTIME, "colidx range kernel: %s", hipGetErrorString(e2)));
        if (host2[0] < 0 || host2[1] >= n)
            return bail(fail(SPMV_HIP_E_ARG, "ColIdx out of range: min %d, max %d, n = %d", host2[0], host2[1], n));
    }
    *out = d;
    return SPMV_HIP_OK;
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
    free_schedule(d);
    if (d->rowptr) (void) hipFree(d->rowptr);
    if (d->colidx) (void) hipFree(d->colidx);
    if (d->val) (void) hipFree(d->val);
    if (d->x_stage) (void) hipFree(d->x_stage);
    if (d->y_stage) (void) hipFree(d->y_stage);
    delete d;
}

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
    HIP_TRY(hipMalloc((void **) &cnt, 2 * sizeof(int)));
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
    (void) hipFree(cnt);
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "x-window inspector: %s", hipGetErrorString(e));
    return SPMV_HIP_OK;
}

constexpr size_t kVecXTileBytes = 48 * 1024; // LDS budget of one row tile's x span (CSR-vector, Balanced)
template <typename T> static int build_long_rows(spmv_dev *d, int thr);
template <typename T>
static int build_csr5(spmv_dev *d, Csr5Plan &P, int m, long long nnz, const int *rowptr, const int *colidx, const T *val, int empty_rows,
                      double mean_row_len, const int *out_rows, bool natural = false);
template <typename T> static int autotune_vector(spmv_dev *d);
template <typename T> static int build_tile_windows(spmv_dev *d, int tiles, const int *split, int rows_per_tile = kVecTileRows);

constexpr size_t kSplitXTileBytes = 48 * 1024; // LDS budget of one nnz-split tile group's x span

template <typename T>
static int build_nnz_split(spmv_dev *d)
{
    constexpr int tile = SplitCfg<T>::Tile;
    d->ntiles = (int) ((d->nnz + tile - 1) / tile);
    if (d->ntiles == 0) return SPMV_HIP_OK;
    ALLOC_TRY(d, &d->tile_first, sizeof(int) * ((size_t) d->ntiles + 1), true);
    ALLOC_TRY(d, &d->carry, sizeof(T) * (size_t) d->ntiles, true);
    int *flag = nullptr;
    ALLOC_TRY(d, &flag, sizeof(int), true);
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), d->stream));
    nnz_tile_first_kernel<<<grid_for((long long) d->ntiles + 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(
        d->m, d->ntiles, tile, d->rowptr, d->tile_first, flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&d->need_fixup, flag, sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    // x windows of every group of kSplitGroupTiles tiles (xwindows.hpp) on a private ColIdx copy
    d->ns_groups = (d->ntiles + kSplitGroupTiles - 1) / kSplitGroupTiles;
    ALLOC_TRY(d, &d->ns_col, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), true);
    ALLOC_TRY(d, &d->ns_wins, sizeof(TileWindows) * (size_t) d->ns_groups, true);
    HIP_TRY(hipMemsetAsync(d->ns_col, 0, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), d->stream));
    {
        const int rc = build_range_windows(d, d->plan.variant == 3 ? 0 : d->ns_groups, d->nnz, (long long) kSplitGroupTiles * tile, nullptr, 1, 1,
                                           (int) (kSplitXTileBytes / sizeof(T)) - 1, d->colidx, d->ns_wins, &d->ns_staged, &d->ns_maxspan, d->ns_col, 0);
        if (rc) return rc;
    }
    return SPMV_HIP_OK;
}

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
    return SPMV_HIP_OK;
}

// Windows of every row tile + the tile-local ColIdx copy (kernels/csr_vector_tile.hpp).
template <typename T>
static int build_tile_windows(spmv_dev *d, int tiles, const int *split, int rows_per_tile)
{
    int *cnt = nullptr;
    int host2[2] = {0, 0};
    d->vt_tiles = tiles;
    ALLOC_TRY(d, &cnt, 2 * sizeof(int), true);
    static_assert(kVecXTileBytes <= 65536, "LDS byte offsets must fit 16 bits");
    ALLOC_TRY(d, &d->vt_col, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), true);
    ALLOC_TRY(d, &d->vt_wins, sizeof(TileWindows) * (size_t) tiles, true);
    HIP_TRY(hipMemsetAsync(d->vt_col, 0, sizeof(unsigned short) * ((size_t) d->nnz + kStreamPad), d->stream));
    HIP_TRY(hipMemsetAsync(cnt, 0, 2 * sizeof(int), d->stream));
    csr_tile_windows_kernel<<<tiles, kBlock, 0, d->stream>>>(d->m, d->n, rows_per_tile, d->long_thr, (int) (kVecXTileBytes / sizeof(T)) - 1, (int) sizeof(T), split, d->rowptr, d->colidx,
                                                             d->vt_wins, d->vt_col, cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host2, cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->vt_staged = host2[0];
    d->vt_maxspan = host2[1];
    return SPMV_HIP_OK;
}

// LDS budget of one SELL window's x tile: 96 KiB of the CU's 160 KiB (one 512-thread workgroup per
// window; fp32 24576 columns, fp64 12288 columns).
constexpr size_t kSellXTileBytes = 96 * 1024;

constexpr size_t kLongXTileBytes = 48 * 1024; // LDS budget of one long-row segment's x span

// Rows longer than thr -> long_rows[] (row order).  Default: gathered into a sub-CSR with its own CSR5
// plan (d->c5_long); variant 13: cut into kLongSeg segments for long_rows_kernel (kernels/long_rows.hpp).
template <typename T>
static int build_long_rows(spmv_dev *d, int thr)
{
    d->long_thr = thr;
    d->nlong = 0;
    d->lr_segs = 0;
    d->c5_long = Csr5Plan();
    if (d->stats.max_row_len <= thr) return SPMV_HIP_OK;
    // deterministic compaction of the long rows (flags -> scan -> scatter), as csr5 does for non-empty rows
    const int nb = (int) (((long long) d->m + kScanTile - 1) / kScanTile);
    int *flags = nullptr, *sums = nullptr, *total = nullptr, *scratch = nullptr, *seg_cnt = nullptr;
    ALLOC_TRY(d, &sums, sizeof(int) * (size_t) nb, true);
    ALLOC_TRY(d, &total, sizeof(int), true);
    HIP_TRY(hipMalloc((void **) &flags, sizeof(int) * (size_t) d->m));
    long_rows_flag_kernel<<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, thr, d->rowptr, flags);
    scan_block_sums_kernel<<<nb, kBlock, 0, d->stream>>>(d->m, flags, sums);
    scan_sums_inplace_kernel<<<1, kBlock, 0, d->stream>>>(nb, sums, total);
    if (hipMemcpyAsync(&d->nlong, total, sizeof(int), hipMemcpyDeviceToHost, d->stream) != hipSuccess || hipStreamSynchronize(d->stream) != hipSuccess) {
        (void) hipFree(flags);
        d->nlong = 0;
        return fail(SPMV_HIP_E_RUNTIME, "long-row scan failed");
    }
    if (d->nlong == 0) { (void) hipFree(flags); return SPMV_HIP_OK; }
    int rc = dev_alloc(d, (void **) &d->long_rows, sizeof(int) * (size_t) d->nlong, true);
    if (!rc) rc = dev_alloc(d, (void **) &scratch, sizeof(int) * (size_t) d->nlong, true);
    if (rc) { (void) hipFree(flags); d->nlong = 0; return rc; }
    csr5_compact_kernel<<<nb, kBlock, 0, d->stream>>>(d->m, flags, sums, d->rowptr, scratch, d->long_rows);
    hipError_t e = hipStreamSynchronize(d->stream);
    (void) hipFree(flags);
    if (e != hipSuccess) { d->nlong = 0; return fail(SPMV_HIP_E_RUNTIME, "long-row compaction: %s", hipGetErrorString(e)); }
    ALLOC_TRY(d, &d->lr_seg_start, sizeof(long long) * ((size_t) d->nlong + 1), true);

    if (d->plan.variant != 13) { // sub-CSR of the long rows + CSR5 over it
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

    int *cnt = total;
    ALLOC_TRY(d, &seg_cnt, sizeof(int) * (size_t) d->nlong, true);
    long_rows_segcount_kernel<<<grid_for(d->nlong, kBlock, INT_MAX), kBlock, 0, d->stream>>>(d->nlong, d->long_rows, d->rowptr, seg_cnt);
    scan_i32_to_i64_kernel<<<1, kBlock, 0, d->stream>>>(d->nlong, seg_cnt, d->lr_seg_start);
    HIP_TRY(hipGetLastError());
    long long nsegs = 0;
    HIP_TRY(hipMemcpyAsync(&nsegs, d->lr_seg_start + d->nlong, sizeof(long long), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->lr_segs = (int) nsegs;
    ALLOC_TRY(d, &d->lr_seg_lr, sizeof(int) * (size_t) nsegs, true);
    ALLOC_TRY(d, &d->lr_part, sizeof(T) * (size_t) nsegs, true);
    ALLOC_TRY(d, &d->lr_seg_lo, sizeof(int) * (size_t) nsegs, true);
    ALLOC_TRY(d, &d->lr_seg_span, sizeof(int) * (size_t) nsegs, true);
    long_rows_segfill_kernel<<<grid_for(d->nlong, kBlock, INT_MAX), kBlock, 0, d->stream>>>(d->nlong, d->lr_seg_start, d->lr_seg_lr);
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), d->stream));
    long_rows_segspan_kernel<<<(int) nsegs, kBlock, 0, d->stream>>>(d->lr_seg_lr, d->lr_seg_start, d->long_rows, d->rowptr, d->colidx,
                                                                    d->lr_seg_lo, d->lr_seg_span, (int) (kLongXTileBytes / sizeof(T)), cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&d->lr_maxspan, cnt, sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return SPMV_HIP_OK;
}


// x windows of every 256-row tile (kernels/csr_vector_tile.hpp)
template <typename T>
static int build_vector_tiles(spmv_dev *d)
{
    d->vt_staged = d->vt_maxspan = 0;
    const int rows = kVecTileRows;
    d->vt_tiles = (int) (((long long) d->m + rows - 1) / rows);
    if (d->vt_tiles == 0 || d->nnz == 0) return SPMV_HIP_OK;
    return build_tile_windows<T>(d, d->vt_tiles, nullptr, rows);
}

template <typename T> static int launch_csr5(spmv_dev *d, const Csr5Plan &P, const T *x, T *y);

template <typename T>
static void launch_long_rows(spmv_dev *d, const T *x, T *y)
{
    if (d->nlong <= 0) return;
    if (d->c5_long.nnz > 0) { (void) launch_csr5<T>(d, d->c5_long, x, y); return; }
    // LDS request = the largest span that is actually staged (keeps several workgroups per CU)
    const size_t xbytes = (((size_t) d->lr_maxspan * sizeof(T)) + 1023) & ~(size_t) 1023;
    long_rows_kernel<T><<<d->lr_segs, kBlock, xbytes, d->stream>>>(
        d->lr_segs, (int) (kLongXTileBytes / sizeof(T)), d->lr_seg_lr, d->lr_seg_start, d->long_rows, d->lr_seg_lo, d->lr_seg_span, d->rowptr, d->colidx, (const T *) d->val, x, y, (T *) d->lr_part);
    if (d->lr_segs > d->nlong)
        long_rows_combine_kernel<T><<<grid_for(d->nlong, kBlock, INT_MAX), kBlock, 0, d->stream>>>(d->nlong, d->lr_seg_start, d->long_rows,
                                                                                                  (const T *) d->lr_part, y);
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
    if (d->plan.sell_lds_x && d->plan.variant != 3) { // x windows of every sigma window, in place on scol (xwindows.hpp)
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
        while (d->sell_staged > 0 && d->sell_group < 8 &&
               (double) d->sell_maxspan * sizeof(T) > 0.15 * (double) slots * (sizeof(T) + 2) / (double) d->sell_nwin) {
            const int prev = d->sell_group;
            rc = inspect(prev * 2);
            if (rc) return rc;
            if (d->sell_staged == 0) { rc = inspect(prev); if (rc) return rc; break; }
        }
        if (d->sell_staged == d->sell_nwin) { sched_free(d, d->scol); d->scol = nullptr; } // no window reads global columns
        else if (d->sell_staged == 0) { sched_free(d, d->scol16); d->scol16 = nullptr; }
        HIP_TRY(hipFuncSetAttribute((const void *) sell_window_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) kSellXTileBytes));
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
    int *flag = nullptr;
    ALLOC_TRY(d, &flag, sizeof(int), true);
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), d->stream));
    csr5_tile_ptr_kernel<<<grid_for((long long) p + 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(m2, (int) P.nnz, p, TN, rp, P.tile_ptr);
    HIP_TRY(hipGetLastError());
    csr5_desc_kernel<SIGMA><<<grid_for(p, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>(m2, (int) P.nnz, p, rp, P.tile_ptr, P.desc, P.run_len, flag);
    HIP_TRY(hipGetLastError());
    if (!P.natural) {
        csr5_transpose_kernel<T, SIGMA><<<grid_for(p, kBlock / kWave, INT_MAX), kBlock, 0, d->stream>>>((int) P.nnz, p, colidx, val, P.col, (T *) P.val);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(&P.fixup, flag, sizeof(int), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    // x windows of every group of consecutive tiles -> the 16-bit slot stream (xwindows.hpp).  Group size:
    // 16 tiles (natural layout: 32) unless staging the windows costs more than 15 % of the bytes the group
    // streams -- wide windows, e.g. columns scattered +-4096 around the diagonal -- then 32 and 64 tiles are
    // tried as long as the groups still fit LDS (config 4: CSR5 0.64 -> 0.58 ms fp32, 1.22 -> 1.05 ms fp64;
    // narrow windows lose 3-7 % with larger groups, so they keep 16).
    static_assert(kCsr5XTileBytes / sizeof(float) <= 65536, "LDS slots must fit 16 bits");
    const int base_gt = P.natural ? 2 * kCsr5GroupTiles : kCsr5GroupTiles;
    const int forced_gt = d->plan.variant == 40 ? 32 : (d->plan.variant == 41 ? 64 : (d->plan.variant == 42 ? 8 : 0)); // A/B
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
        return build_range_windows(d, d->plan.variant == 3 ? 0 : P.groups, total, (long long) gt * TN, nullptr, 1, 1, max_cols, P.col, P.wins,
                                   &P.staged, &P.maxspan, P.col16, P.natural ? 0 : SIGMA);
    };
    int rc = inspect(forced_gt ? forced_gt : base_gt);
    if (rc) return rc;
    while (!forced_gt && P.staged > 0 && P.group_tiles < 64 &&
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
    if (P.staged == 0) { sched_free(d, P.col16); P.col16 = nullptr; }
    else if (!P.natural && P.staged == P.groups) { sched_free(d, P.col); P.col = nullptr; } // no group reads global columns
    return SPMV_HIP_OK;
}

// CSR5 over the CSR (m rows, nnz) given by rowptr / colidx / val.  out_rows (nullable, no empty rows
// allowed then) names the y row of each CSR row -- used for the long-row sub-matrix.
template <typename T>
static int build_csr5(spmv_dev *d, Csr5Plan &P, int m, long long nnz, const int *rowptr, const int *colidx, const T *val, int empty_rows,
                      double mean_row_len, const int *out_rows, bool natural)
{
    int sigma = d->plan.csr5_sigma;
    if (sigma == 0) sigma = mean_row_len <= 4.0 ? 4 : (mean_row_len <= 12.0 ? 8 : 16);
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
        HIP_TRY(hipMalloc((void **) &flags, sizeof(int) * (size_t) m));
        auto cleanup = [&]() { (void) hipFree(flags); };
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
    switch (sigma) {
    case 4: return build_csr5_sigma<T, 4>(d, P, rp, m2, colidx, val);
    case 8: return build_csr5_sigma<T, 8>(d, P, rp, m2, colidx, val);
    default: return build_csr5_sigma<T, 16>(d, P, rp, m2, colidx, val);
    }
}

// Row blocks x column slabs (kernels/blocked.hpp).  R rows per block: y of a block = 64 KiB of LDS;
// W columns per slab: 256 KiB of x.
template <typename T>
static int build_blocked(spmv_dev *d)
{
    const int R = d->plan.block_rows > 0 ? d->plan.block_rows : (int) (64 * 1024 / sizeof(T));
    const size_t slab_bytes = (size_t) (d->plan.slab_kib > 0 ? d->plan.slab_kib : 256) << 10;
    int wshift = 0;
    while ((sizeof(T) << wshift) < slab_bytes) ++wshift;
    const int K = (int) ((((long long) d->n - 1) >> wshift) + 1);
    const int B = (int) (((long long) d->m + R - 1) / R);
    if ((long long) B * K > (1ll << 26)) return SPMV_HIP_OK; // cell table too large: keep the tile executor
    int *cnt = nullptr, *tot = nullptr;
    long long *cursor = nullptr;
    const size_t cells = (size_t) B * K;
    HIP_TRY(hipMalloc((void **) &cnt, sizeof(int) * cells));
    auto cleanup = [&]() { (void) hipFree(cnt); if (tot) (void) hipFree(tot); if (cursor) (void) hipFree(cursor); };
    if (hipMalloc((void **) &tot, sizeof(int) * (size_t) B) != hipSuccess || hipMalloc((void **) &cursor, sizeof(long long) * cells) != hipSuccess) {
        cleanup();
        return fail(SPMV_HIP_E_ALLOC, "hipMalloc(block cells)");
    }
    int rc = dev_alloc(d, (void **) &d->blk_start, sizeof(long long) * ((size_t) B + 1), true);
    if (!rc) rc = dev_alloc(d, (void **) &d->blk_end, sizeof(long long) * (size_t) B, true);
    if (rc) { cleanup(); return rc; }
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int) * cells, d->stream);
    blk_count_kernel<<<grid_for(d->m, kBlock / 16, d->cus * 16), kBlock, 0, d->stream>>>(d->m, R, K, wshift, d->rowptr, d->colidx, cnt);
    blk_totals_kernel<<<grid_for(B, kBlock, INT_MAX), kBlock, 0, d->stream>>>(B, K, cnt, tot);
    scan_i32_to_i64_kernel<<<1, kBlock, 0, d->stream>>>(B, tot, d->blk_start);
    blk_cells_kernel<<<grid_for(B, kBlock, INT_MAX), kBlock, 0, d->stream>>>(B, K, cnt, d->blk_start, cursor, d->blk_end);
    long long total = 0;
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d->blk_start + B, sizeof(long long), hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e != hipSuccess) { cleanup(); return fail(SPMV_HIP_E_RUNTIME, "block inspector: %s", hipGetErrorString(e)); }
    const size_t slots = (size_t) total + 4096; // the last groups of a block read up to 3 load groups past its end
    rc = dev_alloc(d, &d->blk_val, sizeof(T) * slots, true);
    if (!rc) rc = dev_alloc(d, (void **) &d->blk_col, sizeof(int) * slots, true);
    if (!rc) rc = dev_alloc(d, (void **) &d->blk_row, sizeof(unsigned short) * slots, true);
    if (rc) { cleanup(); return rc; }
    blk_fill_kernel<T><<<grid_for(d->m, kBlock / 16, d->cus * 16), kBlock, 0, d->stream>>>(d->m, R, K, wshift, d->rowptr, d->colidx, (const T *) d->val,
                                                                                       (unsigned long long *) cursor, (T *) d->blk_val, d->blk_col, d->blk_row);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    cleanup();
    if (e != hipSuccess) return fail(SPMV_HIP_E_RUNTIME, "block fill: %s", hipGetErrorString(e));
    d->blk_R = R; d->blk_K = K; d->blk_B = B;
    d->blk_on = true;
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_build(spmv_dev *d, const spmv_plan *plan)
{
    if (!d || !plan) return fail(SPMV_HIP_E_ARG, "build: NULL");
    if (plan->sched < 0 || plan->sched >= SPMV_SCHED_COUNT) return fail(SPMV_HIP_E_ARG, "unknown schedule %d", plan->sched);
    free_schedule(d);
    d->plan = *plan;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = SPMV_HIP_OK;
    const bool f64 = d->vsize == sizeof(double);
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
        if (!rc && plan->autotune) rc = f64 ? autotune_vector<double>(d) : autotune_vector<float>(d);
        break;
    }
    case SPMV_SCHED_NNZ_SPLIT:
        if (plan->variant == 8) { // A/B: the first-round 256-nnz tiles with LDS row marks (kernels/nnz_split.hpp)
            rc = f64 ? build_nnz_split<double>(d) : build_nnz_split<float>(d);
            break;
        }
        // equal-nnz tiles over the matrix's own arrays: CSR5 descriptors + carry fix-up, natural layout
        rc = f64 ? build_csr5<double>(d, d->ns, d->m, d->nnz, d->rowptr, d->colidx, (const double *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr, true)
                 : build_csr5<float>(d, d->ns, d->m, d->nnz, d->rowptr, d->colidx, (const float *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr, true);
        // columns without locality (no tile group's x windows fit LDS) and x far larger than an L2: gathers are
        // fabric-bound -> row blocks x column slabs (kernels/blocked.hpp)
        if (!rc && (plan->cache_block == 2 ||
                    (plan->cache_block == 1 && d->ns.staged == 0 && d->nnz >= (1ll << 22) && (long long) d->n * (long long) d->vsize >= (16ll << 20))))
            rc = f64 ? build_blocked<double>(d) : build_blocked<float>(d);
        break;
    case SPMV_SCHED_ROWBLOCK:
        if (plan->rowblock_nnz < 64) return fail(SPMV_HIP_E_ARG, "rowblock_nnz must be >= 64");
        if (d->stats.max_row_len > plan->rowblock_nnz) return fail(SPMV_HIP_E_ARG, "row-block schedule needs max_row_len <= rowblock_nnz");
        rc = build_rowblock(d);
        if (!rc) rc = f64 ? build_rowblock_tiles<double>(d) : build_rowblock_tiles<float>(d);
        // same fall-back as nnz-split for columns without locality (Method_Balanced and Method_Balanced2 are one family)
        if (!rc && (plan->cache_block == 2 ||
                    (plan->cache_block == 1 && d->vt_staged == 0 && d->nnz >= (1ll << 22) && (long long) d->n * (long long) d->vsize >= (16ll << 20))))
            rc = f64 ? build_blocked<double>(d) : build_blocked<float>(d);
        break;
    case SPMV_SCHED_SELL: rc = f64 ? build_sell<double>(d) : build_sell<float>(d); break;
    case SPMV_SCHED_CSR5:
        rc = f64 ? build_csr5<double>(d, d->c5, d->m, d->nnz, d->rowptr, d->colidx, (const double *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr)
                 : build_csr5<float>(d, d->c5, d->m, d->nnz, d->rowptr, d->colidx, (const float *) d->val, d->stats.empty_rows, d->stats.mean_row_len, nullptr);
        break;
    }
    if (rc) { free_schedule(d); return rc; }
    d->inspect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    d->built = true;
    return SPMV_HIP_OK;
}

// ------------------------------------------------------------------------------------ executors
// One workgroup per kVecNB * (256/L) consecutive rows, dispatched in row order: measured on the
// config-2 shape a plain in-order grid beats a persistent grid-stride loop by ~10 %.
constexpr int kVecNB = 4;
// Kernel forms of the CSR-vector schedule.  Which one is fastest differs between MI355X boxes by a
// few percent, so create() times the applicable ones once on the resident matrix
// (autotune_vector) and keeps the winner in d->vec_choice; plan.variant overrides for A/B runs.
enum { VEC_AUTO = 0, VEC_STRIDED = 1, VEC_NO_LONG = 2, VEC_PIPE = 4, VEC_TILE_D2 = 5, VEC_TILE_D8 = 6, VEC_TILE_D4 = 10, VEC_TILE_D4_NOPRE = 11,
       VEC_TILE_D2_NOPRE = 12, VEC_LONG_SEGMENTS = 13 /* long rows through long_rows_kernel instead of the CSR5 sub-matrix */ };

template <typename T, int L, int DEPTH, bool PRE = true>
static void launch_vector_tile(spmv_dev *d, const T *x, T *y, int long_thr)
{
    const size_t lds = ((((size_t) d->vt_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
    if (lds > 64 * 1024)
        (void) hipFuncSetAttribute((const void *) csr_vector_tile_kernel<T, L, DEPTH, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    csr_vector_tile_kernel<T, L, DEPTH, PRE><<<d->vt_tiles, kVecTileThreads, lds, d->stream>>>(d->m, long_thr, d->rowptr, d->colidx, d->vt_col, (const T *) d->val,
                                                                                           d->vt_wins, x, y);
}

template <typename T, int L>
static void launch_vector(spmv_dev *d, const T *x, T *y)
{
    const int v = d->plan.variant ? d->plan.variant : d->vec_choice;
    const int long_thr = v == VEC_NO_LONG ? INT_MAX : d->long_thr;
    if (v == VEC_STRIDED) { // A/B: the first-round strided kernel
        csr_vector_kernel<T, (L < 2 ? 2 : L)><<<grid_for(d->m, kBlock / (L < 2 ? 2 : L), d->cus * 32), kBlock, 0, d->stream>>>(
            d->m, d->rowptr, d->colidx, (const T *) d->val, x, y);
        return;
    }
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
static void launch_vector_any(spmv_dev *d, const T *x, T *y)
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

// Time the applicable CSR-vector forms on the resident matrix (x = 1) and keep the fastest.
template <typename T>
static int autotune_vector(spmv_dev *d)
{
    d->vec_choice = VEC_AUTO;
    if (d->nnz < (1ll << 24) || d->plan.variant != 0 || d->vt_tiles <= 0) return SPMV_HIP_OK;
    T *x = nullptr, *y = nullptr;
    if (hipMalloc((void **) &x, sizeof(T) * (size_t) d->n) != hipSuccess || hipMalloc((void **) &y, sizeof(T) * (size_t) d->m) != hipSuccess) {
        (void) hipGetLastError();
        if (x) (void) hipFree(x);
        return SPMV_HIP_OK; // no room to tune: keep the default
    }
    fill_value_kernel<T><<<grid_for(d->n, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->n, x, T(1));
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    constexpr int kCand = 5;
    const int cand[kCand] = {VEC_TILE_D4, VEC_TILE_D4_NOPRE, VEC_TILE_D2, VEC_TILE_D2_NOPRE, VEC_PIPE};
    float tmin[kCand];
    for (int k = 0; k < kCand; ++k) tmin[k] = 1e30f;
    for (int c : cand) { d->vec_choice = c; launch_vector_any<T>(d, x, y); } // warm every form once
    for (int round = 0; round < 4; ++round) // interleaved rounds (one process, same clocks): min per form
        for (int k = 0; k < kCand; ++k) {
            d->vec_choice = cand[k];
            (void) hipEventRecord(e0, d->stream);
            launch_vector_any<T>(d, x, y);
            launch_vector_any<T>(d, x, y);
            (void) hipEventRecord(e1, d->stream);
            (void) hipEventSynchronize(e1);
            float ms = 0;
            (void) hipEventElapsedTime(&ms, e0, e1);
            if (ms * 0.5f < tmin[k]) tmin[k] = ms * 0.5f;
        }
    float best = 1e30f;
    int best_c = VEC_AUTO;
    for (int k = 0; k < kCand - 1; ++k) {
        if (tmin[k] < best) { best = tmin[k]; best_c = cand[k]; }
    }
    // the pipe form (int32 columns, global gathers) only on a clear win: a noisy sample -- e.g. another
    // process on the device during create -- must not cost 30 % on every later launch
    if (tmin[kCand - 1] < 0.95f * best) { best = tmin[kCand - 1]; best_c = VEC_PIPE; }
    d->tune_ms[0] = tmin[0] < tmin[1] ? tmin[0] : tmin[1]; // tile, 4 steps in flight (best of the two issue orders)
    d->tune_ms[1] = tmin[2] < tmin[3] ? tmin[2] : tmin[3]; // tile, 2 steps in flight
    d->tune_ms[2] = tmin[4];                               // pipe
    d->vec_choice = best_c;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    (void) hipFree(x);
    (void) hipFree(y);
    if (hipGetLastError() != hipSuccess) d->vec_choice = VEC_AUTO;
    return SPMV_HIP_OK;
}

template <typename T>
static void launch_blocked(spmv_dev *d, const T *x, T *y)
{
    const size_t lds = (size_t) d->blk_R * sizeof(T);
    if (lds > 64 * 1024) (void) hipFuncSetAttribute((const void *) blk_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    blk_kernel<T><<<d->blk_B, kBlkThreads, lds, d->stream>>>(d->m, d->blk_R, d->blk_start, d->blk_end, (const T *) d->blk_val, d->blk_col, d->blk_row, x, y);
}

template <typename T, int SIGMA, bool MAPPED>
static void launch_csr5_form(spmv_dev *d, const Csr5Plan &P, const T *x, T *y)
{
    if (P.staged > 0) { // the inspector staged (at least half of) the groups: their column stream is the 16-bit slot array
        const size_t lds = ((((size_t) P.maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
        if (P.natural) {
            if (lds > 16 * 1024) // static tile buffers (up to 46 KiB) + this may pass the default 64 KiB limit
                (void) hipFuncSetAttribute((const void *) nat_group_kernel<T, SIGMA, MAPPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            nat_group_kernel<T, SIGMA, MAPPED><<<P.groups, kBlock, lds, d->stream>>>(P.group_tiles, P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, P.col16, (const T *) P.val,
                                                                                    P.row_map, P.wins, x, y, (T *) P.carry, P.n_empty, P.empty_list);
            return;
        }
        if (lds > 64 * 1024) // above the default dynamic-LDS limit: raise it for this instantiation (idempotent, cheap)
            (void) hipFuncSetAttribute((const void *) csr5_group_kernel<T, SIGMA, MAPPED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        csr5_group_kernel<T, SIGMA, MAPPED><<<P.groups, kBlock, lds, d->stream>>>(P.group_tiles, P.tiles, P.tile_ptr, P.desc, P.col, P.col16, (const T *) P.val, P.row_map, P.wins,
                                                                                 x, y, (T *) P.carry, P.n_empty, P.empty_list);
        return;
    }
    const int grid = grid_for(P.tiles, kBlock / kWave, INT_MAX);
    if (P.natural)
        nat_kernel<T, SIGMA, MAPPED><<<grid, kBlock, 0, d->stream>>>(P.tiles, (int) P.nnz, P.tile_ptr, P.desc, P.col, (const T *) P.val, P.row_map, x, y, (T *) P.carry, P.n_empty, P.empty_list);
    else
        csr5_kernel<T, SIGMA, MAPPED><<<grid, kBlock, 0, d->stream>>>(P.tiles, P.tile_ptr, P.desc, P.col, (const T *) P.val, P.row_map, x, y, (T *) P.carry, P.n_empty, P.empty_list);
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
    if (P.fixup && P.tiles > 1) {
        const int g = grid_for(P.tiles - 1, kBlock, INT_MAX);
        if (P.row_map) csr5_fixup_kernel<T, true><<<g, kBlock, 0, d->stream>>>(P.tiles, P.tile_ptr, P.run_len, P.row_map, (const T *) P.carry, y);
        else csr5_fixup_kernel<T, false><<<g, kBlock, 0, d->stream>>>(P.tiles, P.tile_ptr, P.run_len, nullptr, (const T *) P.carry, y);
    }
    return SPMV_HIP_OK;
}

template <typename T, int L>
static void launch_rows(spmv_dev *d, const T *x, T *y)
{
    const size_t lds = ((((size_t) d->vt_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023; // + the zero slot
    if (lds > 64 * 1024)
        (void) hipFuncSetAttribute((const void *) csr_vector_rows_kernel<T, L, (sizeof(T) == 8 ? 4 : 2)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    csr_vector_rows_kernel<T, L, (sizeof(T) == 8 ? 4 : 2)><<<d->nblocks, kVecTileThreads, lds, d->stream>>>(
        d->long_thr, d->rb_split, d->rowptr, d->colidx, d->vt_col, (const T *) d->val, d->vt_wins, x, y);
}

template <typename T>
static int launch(spmv_dev *d, const T *x, T *y)
{
    if (d->m == 0) return SPMV_HIP_OK;
    if (d->nnz == 0) { // nothing to multiply: y = 0
        fill_zero_kernel<T><<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, y);
        HIP_TRY(hipGetLastError());
        return SPMV_HIP_OK;
    }
    const T *val = (const T *) d->val;
    switch (d->plan.sched) {
    case SPMV_SCHED_CSR_SCALAR:
        csr_scalar_kernel<T><<<grid_for(d->m, kBlock, d->cus * 8), kBlock, 0, d->stream>>>(d->m, d->rowptr, d->colidx, val, x, y);
        break;
    case SPMV_SCHED_CSR_VECTOR:
        launch_vector_any<T>(d, x, y);
        if (d->plan.variant != 2) launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_NNZ_SPLIT: {
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        if (d->plan.variant != 8) {
            const int rc = launch_csr5<T>(d, d->ns, x, y);
            if (rc) return rc;
            break;
        }
        if (d->ns_staged > 0) {
            const size_t lds = ((((size_t) d->ns_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023;
            nnz_group_kernel<T><<<d->ns_groups, kBlock, lds, d->stream>>>((int) d->nnz, d->ntiles, d->rowptr, d->colidx, d->ns_col, val, d->ns_wins,
                                                                         x, y, d->tile_first, (T *) d->carry);
        } else {
            const int grid = grid_for(d->ntiles, kBlock / kWave, d->plan.variant == 1 ? d->cus * 8 : INT_MAX);
            nnz_split_kernel<T><<<grid, kBlock, 0, d->stream>>>(d->m, (int) d->nnz, d->ntiles, d->rowptr, d->colidx, val, x, y,
                                                                 d->tile_first, (T *) d->carry);
        }
        if (d->need_fixup && d->ntiles > 1)
            nnz_fixup_kernel<T><<<grid_for(d->ntiles - 1, kBlock, INT_MAX), kBlock, 0, d->stream>>>(
                d->ntiles, d->rowptr, d->tile_first, (const T *) d->carry, y);
        break;
    }
    case SPMV_SCHED_ROWBLOCK:
        if (d->blk_on) { launch_blocked<T>(d, x, y); break; }
        if (d->plan.variant == 7 && d->rb_stride <= 4096) { // A/B: the first-round LDS-products kernel
            rowblock_kernel<T><<<d->nblocks, kBlock, 2 * (size_t) d->rb_stride * sizeof(T), d->stream>>>(d->rb_split, d->rowptr, d->colidx, val, x, y);
            break;
        }
        switch (d->plan.lanes_per_row) {
        case 1: launch_rows<T, 1>(d, x, y); break;
        case 2: launch_rows<T, 2>(d, x, y); break;
        case 4: launch_rows<T, 4>(d, x, y); break;
        case 8: launch_rows<T, 8>(d, x, y); break;
        case 16: launch_rows<T, 16>(d, x, y); break;
        case 32: launch_rows<T, 32>(d, x, y); break;
        default: launch_rows<T, 64>(d, x, y); break;
        }
        launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_SELL:
        // staged path when at least half of the windows fit their x span in LDS; the LDS request is
        // sized by the largest staged span actually present (rounded to 16 KiB) to keep occupancy
        if (d->sell_staged > 0)
            sell_window_kernel<T><<<d->sell_nwin, kSellWinThreads, ((((size_t) d->sell_maxspan + 1) * sizeof(T)) + 1023) & ~(size_t) 1023, d->stream>>>(
                d->sell_group * (d->plan.sell_sigma / kSellC), (long long) d->nchunks, d->chunk_ptr, d->scol, d->scol16, (const T *) d->sval, d->perm, d->sell_wins, x, y);
        else
            sell_kernel<T><<<grid_for(d->nchunks, kBlock / kWave, d->plan.variant == 1 ? d->cus * 8 : INT_MAX), kBlock, 0, d->stream>>>(
                d->nchunks, d->chunk_ptr, d->scol, (const T *) d->sval, d->perm, x, y);
        launch_long_rows<T>(d, x, y);
        break;
    case SPMV_SCHED_CSR5: {
        const int rc = launch_csr5<T>(d, d->c5, x, y);
        if (rc) return rc;
        break;
    }
    default: return fail(SPMV_HIP_E_ARG, "schedule %d has no executor", d->plan.sched);
    }
    HIP_TRY(hipGetLastError());
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_run(spmv_dev *d, const void *x, void *y)
{
    if (!d || !d->built) return fail(SPMV_HIP_E_NOSTATE, "run: schedule not built");
    if ((d->n > 0 && d->nnz > 0 && !x) || (d->m > 0 && !y)) return fail(SPMV_HIP_E_ARG, "run: X or Y is NULL");
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != d->device) HIP_TRY(hipSetDevice(d->device));
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

// ------------------------------------------------------------------------------------ info
static const char *kSchedNames[] = {"csr-scalar", "csr-vector", "row-block", "nnz-split", "sell-c-sigma", "csr5"};
static const char *kKernelNames[] = {"csr_scalar_kernel", "csr_vector_pipe_kernel", "csr_vector_rows_kernel",
                                     "nnz_split_kernel", "sell_kernel", "csr5_kernel"};

extern "C" int spmv_shim_info(const spmv_dev *d, spmv_hip_info *o)
{
    if (!d || !o) return fail(SPMV_HIP_E_ARG, "info: NULL");
    memset(o, 0, sizeof *o);
    o->device = d->device;
    o->schedule = d->plan.sched;
    o->lanes_per_row = d->plan.sched == SPMV_SCHED_CSR_VECTOR ? d->plan.lanes_per_row : 0;
    o->sell_c = d->plan.sched == SPMV_SCHED_SELL ? kSellC : 0;
    o->sell_sigma = d->plan.sched == SPMV_SCHED_SELL ? d->plan.sell_sigma : 0;
    o->tile_nnz = d->plan.sched == SPMV_SCHED_NNZ_SPLIT ? (d->plan.variant == 8 ? (d->vsize == 8 ? SplitCfg<double>::Tile : SplitCfg<float>::Tile) : kWave * d->ns.sigma) : (d->plan.sched == SPMV_SCHED_ROWBLOCK ? d->rb_stride : (d->plan.sched == SPMV_SCHED_CSR5 ? kWave * d->c5.sigma : 0));
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
    o->tuned_choice = d->vec_choice;
    for (int k = 0; k < 3; ++k) o->tune_ms[k] = d->tune_ms[k];
    o->schedule_name = kSchedNames[d->plan.sched];
    o->kernel_name = kKernelNames[d->plan.sched];
    if (d->plan.sched == SPMV_SCHED_CSR_VECTOR && d->vt_tiles > 0 && d->vec_choice != VEC_PIPE &&
        (d->vt_staged * 2 >= d->vt_tiles || (d->vec_choice != VEC_AUTO && d->vec_choice != VEC_PIPE)))
        o->kernel_name = "csr_vector_tile_kernel";
    o->cache_blocked = d->blk_on ? 1 : 0;
    switch (d->plan.sched) {
    case SPMV_SCHED_CSR_VECTOR:
    case SPMV_SCHED_ROWBLOCK: o->x_groups = d->vt_tiles; o->x_groups_staged = d->vt_staged; break;
    case SPMV_SCHED_NNZ_SPLIT: o->x_groups = d->plan.variant == 8 ? d->ns_groups : d->ns.groups; o->x_groups_staged = d->plan.variant == 8 ? d->ns_staged : d->ns.staged; break;
    case SPMV_SCHED_SELL: o->x_groups = d->sell_nwin; o->x_groups_staged = d->sell_staged; break;
    case SPMV_SCHED_CSR5: o->x_groups = d->c5.groups; o->x_groups_staged = d->c5.staged; break;
    default: o->x_groups = o->x_groups_staged = 0; break;
    }
    if (d->blk_on) o->kernel_name = "blk_kernel";
    else if (d->plan.sched == SPMV_SCHED_NNZ_SPLIT)
        o->kernel_name = d->plan.variant == 8 ? (d->ns_staged > 0 ? "nnz_group_kernel" : "nnz_split_kernel") : (d->ns.staged > 0 ? "nat_group_kernel" : "nat_kernel");
    if (d->plan.sched == SPMV_SCHED_CSR5 && d->c5.staged > 0) o->kernel_name = "csr5_group_kernel";
    if (d->plan.sched == SPMV_SCHED_SELL && d->sell_staged > 0) o->kernel_name = "sell_window_kernel";
    return SPMV_HIP_OK;
}
