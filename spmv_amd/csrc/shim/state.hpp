// shim/state.hpp -- part of the single translation unit spmv_shim.hip (included there, in order):
// error channel, the device-side state of a handle (spmv_dev), allocation bookkeeping and the small
// utility kernels (row statistics, ColIdx validation, fills).
#pragma once
#include <atomic>

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

#ifndef SPMV_TU_SECONDARY
extern "C" const char *spmv_shim_error_text(void) { return t_err; }
#endif

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (void) hipGetLastError();                                                              \
            return fail(e__ == hipErrorOutOfMemory ? SPMV_HIP_E_ALLOC : SPMV_HIP_E_RUNTIME,        \
                        "%s -> %s", #expr, hipGetErrorString(e__));                                \
        }                                                                                          \
    } while (0)

// ------------------------------------------------------------------------------------ device-memory pool
// hipFree is lazy on this runtime: it returns at once and the release happens later -- inside some LATER hipMalloc,
// which then takes 0.4-3 s for a few GB (tools/malloc_probe.py: "malloc 3 x 2 GiB again: 480 ms"; config 4: the schedule
// created right after another handle was destroyed took 3.2 s of which 16 ms were inspector kernels).  create() frees
// and allocates in quick succession (re-inspection, auto_method = 2 building five schedules, the blocked executor
// replacing a tile schedule), so freed blocks are kept in a small per-process pool and handed out again: same
// device, at least the size asked for and at most 25 % more.  Cap: SPMV_HIP_POOL_MB (default: an eighth of the
// device's memory; 0 = no pool); spmv_hip_trim_pool() releases everything; an allocation that fails trims the pool and retries.
struct PoolBlock { void *p; size_t bytes; int device; };
static std::mutex g_pool_lock;
static std::vector<PoolBlock> g_pool;              // free blocks
static std::vector<PoolBlock> g_pool_live;         // blocks handed out (to learn their size at free time)
static size_t g_pool_bytes = 0;

static size_t pool_cap()
{
    static long long cap = -1;
    if (cap < 0) {
        const char *e = getenv("SPMV_HIP_POOL_MB");
        if (e && *e) cap = atoll(e) << 20;
        else { // default: an eighth of the device's memory (36 GB of 288): the blocks of one large handle
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); total_b = (size_t) 64 << 30; }
            cap = (long long) (total_b / 8);
        }
        if (cap < 0) cap = 0;
    }
    return (size_t) cap;
}

static void pool_trim_locked()
{
    for (auto &b : g_pool) (void) hipFree(b.p);
    g_pool.clear();
    g_pool_bytes = 0;
}

static hipError_t pool_malloc(void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void) hipGetLastError(); dev = 0; }
    const size_t want = bytes >= (1u << 20) ? (bytes + ((1u << 21) - 1)) & ~(size_t) ((1u << 21) - 1) : bytes; // large blocks in 2 MiB steps
    std::lock_guard<std::mutex> g(g_pool_lock);
    if (want >= (1u << 20)) {
        size_t best = (size_t) -1;
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == dev && g_pool[i].bytes >= want && g_pool[i].bytes <= want + want / 4 &&
                (best == (size_t) -1 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
        if (best != (size_t) -1) {
            PoolBlock b = g_pool[best];
            g_pool.erase(g_pool.begin() + (long) best);
            g_pool_bytes -= b.bytes;
            g_pool_live.push_back(b);
            *p = b.p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess && !g_pool.empty()) { // make room and try once more
        (void) hipGetLastError();
        pool_trim_locked();
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess && want >= (1u << 20)) g_pool_live.push_back({*p, want, dev});
    return e;
}
template <typename T> static hipError_t pool_malloc(T **p, size_t bytes) { return pool_malloc((void **) p, bytes); }

static hipError_t pool_free(void *p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> g(g_pool_lock);
    for (size_t i = 0; i < g_pool_live.size(); ++i)
        if (g_pool_live[i].p == p) {
            PoolBlock b = g_pool_live[i];
            g_pool_live.erase(g_pool_live.begin() + (long) i);
            if (g_pool_bytes + b.bytes <= pool_cap()) {
                g_pool.push_back(b);
                g_pool_bytes += b.bytes;
                return hipSuccess;
            }
            break;
        }
    return hipFree(p);
}

#ifndef SPMV_TU_SECONDARY
extern "C" void spmv_shim_trim_pool(void)
{
    std::lock_guard<std::mutex> g(g_pool_lock);
    pool_trim_locked();
}
#endif

// Makes `device` current for the life of the guard and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void) hipGetLastError(); prev = -1; }
        if (prev != device && hipSetDevice(device) != hipSuccess) { (void) hipGetLastError(); ok = false; }
        if (prev == device) prev = -1; // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

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
    int max_tile_rows = 0;        // max over the tiles of tile_ptr[t + 1] - tile_ptr[t]: below kWave the waves' row maps need kWave ints, not (sigma + 1) * kWave
    int n_empty = 0;              // empty rows (outside row_map): the tile kernel zeroes y for them
    const int *empty_list = nullptr;
    bool natural = false;         // nnz-split: no transposed copies, col/val are the matrix's own arrays (kernels/csr5.hpp, nat_tile)
    TileWindows *wins = nullptr;
    int *tile_ptr = nullptr, *run_len = nullptr, *col = nullptr; // col: transposed global columns (freed when every group is staged)
    unsigned short *col16 = nullptr; // 16-bit LDS slots of the staged groups
    unsigned *lane_run = nullptr;    // RUN groups (csr5.hpp): per lane and tile, first slot | slot at the lane's one row start << 16 -- read instead of col16
    int run_groups = 0;
    long long run_tiles = 0;
    const int *row_map = nullptr; // CSR5 row -> y row (NULL: identity)
    unsigned *desc = nullptr;
    void *val = nullptr, *carry = nullptr;
    // forward completion (csr5.hpp, nat_kernel<.., FWD>): no fix-up launch -- a row cut by a tile boundary is finished by the tile it starts in
    int forward = 0, n_long = 0;  // on / rows longer than a tile (a workgroup each, in front of the tiles)
    int *fwd = nullptr;           // per tile: entries behind the tile that finish its last row (-1 / -2: long-row cases, csr5.hpp)
    int4 *long_list = nullptr;    // per long row: y row, first entry, end
};

// One row-block x column-slab layout (kernels/blocked.hpp; built by build_blocked).
struct BlkSet {
    int R = 0, K = 0, B = 0, wshift = kBlkSlabShift; // most rows of a block (= the junk accumulator's slot), slabs, blocks, log2(columns per slab)
    int ge = 7;                   // log2(entries per group)
    int form = 0;                 // executor form: 0 / 1 = the fewer / more groups per step of the width's two forms (launch_blocked; chosen by autotune_blocked)
    int waves = 1;                // wavefronts sharing one block's accumulators: 1 = a wave per block, two blocks per CU; 4 / 8 = the wide form, one block per CU
    bool ordered = true;          // wide form: the waves take turns at adding (bit-reproducible); false: arrival order (option deterministic = 0)
    bool subsort = true;          // sparse cells stored sorted by column
    float tune_ms[3] = {0, 0, 0};
    long long groups = 0;         // groups stored (without the padding behind the last block)
    int *row0 = nullptr;          // [B + 1] first row of every block (equal-work cut points, blk_partition_kernel)
    long long *gstart = nullptr;  // [B + 1] first group of every block
    BlkDir *dir = nullptr;        // [B] what the executor reads
    int *order = nullptr;         // [B] launch order: blocks with entries first (row order), empty blocks last
    void *val = nullptr;
    unsigned *meta = nullptr;     // 16-bit column offset | 16-bit row in the block
    int *hdr = nullptr;           // per group: first column of its super-slab
};

struct spmv_dev {
    int device = 0;
    int cus = 256;
    hipStream_t stream = nullptr;
    int async = 0;
    int m = 0, n = 0;
    long long nnz = 0;
    size_t vsize = 8;
    int col_min = 0, col_max = -1; // range of ColIdx (create-time validation; the multi-GPU "range" exchange moves only x[col_min .. col_max])
    // resident CSR
    int *rowptr = nullptr, *colidx = nullptr;
    void *val = nullptr;
    spmv_stats stats{};
    spmv_plan plan{};
    bool built = false;
    // row blocks
    int nblocks = 0, rb_stride = 0;
    int *rb_split = nullptr;
    // csr-vector x tiles
    int vt_tiles = 0, vt_staged = 0, vt_maxspan = 0, vec_choice = 0;
    int rows_depth = 0;           // rows kernel (Balanced's row blocks, CSR-vector's wide form): steps in flight, 2 / 4; 0 = the dtype's default (autotune_rows)
    bool vt_wide = false;           // windows above 64 KiB: slot-index stream, blocks of vt_rows rows, rows kernel
    int vt_rows = 256;
    float tune_ms[3] = {0, 0, 0}; // tile D4, tile D2, pipe (autotune_vector)
    unsigned short *vt_col = nullptr; // tile-local column stream: 16-bit LDS slots (staged tiles only)
    unsigned short *vt_rowslot = nullptr; // RUN tiles (every row one run of consecutive columns): slot of each row's first column; no column stream read
    int vt_run_tiles = 0, vt_run_nnz = 0, vt_run_rows = 0;
    unsigned char *vt_col8 = nullptr;     // BYTE tiles (every row's slots within 255 of its smallest): one byte per entry, read INSTEAD of vt_col; vt_rowslot holds the smallest slot
    int vt_byte_tiles = 0, vt_byte_nnz = 0, vt_byte_rows = 0;
    unsigned short *vt_tmpl = nullptr;    // TEMPLATE tiles (rows share a few lists of slot offsets from their first entry): kTmplCount lists of kTmplMax offsets per tile, read INSTEAD of any column stream
    unsigned char *vt_rowtid = nullptr;   // ... and the list number of every row
    int vt_tmpl_tiles = 0, vt_tmpl_nnz = 0, vt_tmpl_rows = 0;
    TileWindows *vt_wins = nullptr; // x windows of every tile
    // long rows (csr-vector, sell)
    int nlong = 0, long_thr = INT_MAX;
    int *long_rows = nullptr;
    long long *lr_seg_start = nullptr; // int64 prefix sum of the long rows' lengths
    // sell
    int nchunks = 0;
    long long sell_cols = 0; // sum of chunk widths
    int *perm = nullptr, *scol = nullptr;
    TileWindows *sell_wins = nullptr;
    unsigned *sell_run = nullptr;      // RUN groups (sell.hpp): per row slot, first LDS slot | row length << 16
    unsigned short *sell_tmpl = nullptr; // TEMPLATE groups: kSellTmplCount lists of kSellTmplMax slot offsets per window group
    long long sell_tmpl_nnz = 0;
    unsigned char *scol8 = nullptr;   // BYTE window groups (sell.hpp): 8-bit offsets from the row's first slot, at the 16-bit slab's positions
    long long sell_byte_nnz = 0, sell_byte_stored = 0, sell_byte_slots = 0;
    int sell_run_groups = 0;
    long long sell_run_nnz = 0, sell_run_stored = 0, sell_run_slots = 0;
    unsigned short *scol16 = nullptr; // 16-bit LDS slots of the staged sigma windows
    int sell_nwin = 0, sell_staged = 0, sell_xcap = 0, sell_maxspan = 0, sell_group = 1; // windows, windows with x staged in LDS, LDS capacity in elements
    long long *chunk_ptr = nullptr;
    void *sval = nullptr;
    // csr5
    Csr5Plan c5, c5_long, ns; // ns: the natural-layout plan of the nnz-split schedule
    // row blocks x column slabs (kernels/blocked.hpp): the executor for columns without locality
    bool blk_on = false;
    BlkSet blk;
    // long-row sub-matrix (rows longer than long_thr, in row order), the input of c5_long
    int *lsub_rowptr = nullptr, *lsub_colidx = nullptr;
    void *lsub_val = nullptr;
    long long lsub_nnz = 0;
    // staging for host x / y
    void *x_stage = nullptr, *y_stage = nullptr, *scratch8 = nullptr;
    long long stream_bytes = 0, x_bytes = 0; // traffic model of one launch (account_stream_bytes)
    int x_groups_seen = 0;                   // tile groups analysed before the blocked executor took over
    float route_ms[2] = {0, 0};              // measured tile schedule vs blocked executor (spmv_shim_build, mode 2)
    // A = A_near + A_far (shim/split.hpp): the two halves, planned and built like any matrix; the parent keeps the resident CSR
    spmv_dev *sp_near = nullptr, *sp_far = nullptr;
    int *sp_centre = nullptr;                // centre column of every 256-row tile
    int sp_half = 0;                         // near = within +-sp_half of the tile's centre
    float near_share = -1.f;                 // sampled share of near entries (-1: not sampled)
    float split_ms[2] = {0, 0};              // measured: schedule as built vs the split pair
    bool accumulate = false;                 // this matrix is the far half of a split: y += A x (blocked executor only)
    long long device_bytes = 0;
    double inspect_ms = 0;
    std::vector<std::pair<void *, size_t>> sched_allocs; // (pointer, bytes): freed when the schedule is rebuilt
};

static int dev_alloc(spmv_dev *d, void **p, size_t bytes, bool sched)
{
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(pool_malloc(p, bytes));
    d->device_bytes += (long long) bytes;
    if (sched) d->sched_allocs.push_back({*p, bytes});
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
        if (d->sched_allocs[i].first == p) {
            d->device_bytes -= (long long) d->sched_allocs[i].second;
            d->sched_allocs.erase(d->sched_allocs.begin() + (long) i);
            (void) pool_free(p);
            return;
        }
}

static void reset_tile_fields(spmv_dev *d)
{
    d->rb_split = nullptr;
    d->perm = d->scol = d->long_rows = nullptr; d->sell_wins = nullptr; d->scol16 = nullptr; d->sell_run = nullptr; d->sell_tmpl = nullptr; d->sell_tmpl_nnz = 0; d->scol8 = nullptr; d->sell_byte_nnz = d->sell_byte_stored = d->sell_byte_slots = 0; d->sell_run_groups = 0; d->sell_run_nnz = d->sell_run_stored = d->sell_run_slots = 0; d->sell_staged = d->sell_nwin = 0; d->chunk_ptr = d->lr_seg_start = nullptr;
    d->sval = nullptr;
    d->nblocks = d->nchunks = d->nlong = 0;
    d->long_thr = INT_MAX;
    d->vt_col = nullptr; d->vt_wins = nullptr; d->vt_tiles = d->vt_staged = d->vt_maxspan = 0; d->vt_wide = false; d->vt_rows = 256;
    d->vt_rowslot = nullptr; d->vt_run_tiles = d->vt_run_nnz = d->vt_run_rows = 0; d->rows_depth = 0;
    d->vt_col8 = nullptr; d->vt_byte_tiles = d->vt_byte_nnz = d->vt_byte_rows = 0;
    d->vt_tmpl = nullptr; d->vt_rowtid = nullptr; d->vt_tmpl_tiles = d->vt_tmpl_nnz = d->vt_tmpl_rows = 0;
    d->c5 = Csr5Plan();
    d->c5_long = Csr5Plan();
    d->ns = Csr5Plan();
    d->lsub_rowptr = d->lsub_colidx = nullptr; d->lsub_val = nullptr; d->lsub_nnz = 0;
}

// Blocks returned to the pool may be handed to another handle at once (hipFree would have waited for the device): no
// launch of this handle may still be reading them.  In async mode (spmv_hip_set_async) a multiply can be in flight when
// the caller destroys or rebuilds the handle.
static void quiesce(spmv_dev *d)
{
    DeviceGuard guard(d->device);
    if (hipStreamSynchronize(d->stream) != hipSuccess) (void) hipGetLastError();
}

static void free_schedule(spmv_dev *d)
{
    if (!d->sched_allocs.empty()) quiesce(d);
    for (auto &a : d->sched_allocs) { (void) pool_free(a.first); d->device_bytes -= (long long) a.second; }
    d->sched_allocs.clear();
    reset_tile_fields(d);
    d->blk_on = false;
    d->blk = BlkSet();
    d->built = false;
}

// The row-block x column-slab executor has taken over: release what the tile schedule built (the first
// `count` schedule allocations), keep the blocked streams.
static void drop_tile_schedule(spmv_dev *d, size_t count)
{
    quiesce(d);
    for (size_t i = 0; i < count && i < d->sched_allocs.size(); ++i) { (void) pool_free(d->sched_allocs[i].first); d->device_bytes -= (long long) d->sched_allocs[i].second; }
    d->sched_allocs.erase(d->sched_allocs.begin(), d->sched_allocs.begin() + (long) (count < d->sched_allocs.size() ? count : d->sched_allocs.size()));
    reset_tile_fields(d);
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

#ifndef SPMV_TU_SECONDARY
extern "C" int spmv_shim_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return n;
}
#endif

// ------------------------------------------------------------------------------------ stats
static __global__ __launch_bounds__(kBlock) void stats_kernel(int m, const int *__restrict__ rowptr, DevStats *s)
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

static __global__ __launch_bounds__(kBlock) void count_longer_kernel(int m, int thr, const int *__restrict__ rowptr, int *count)
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
static __global__ __launch_bounds__(kBlock) void colidx_range_kernel(long long nnz, const int *__restrict__ colidx, int *__restrict__ mnmx)
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

// Dynamic LDS above the 64 KiB default needs hipFuncAttributeMaxDynamicSharedMemorySize raised once per kernel
// instantiation (and per device): remembered here, so that launches do not pay the call every time.
static std::mutex g_lds_attr_lock;

template <auto Kernel>
static void ensure_lds(const spmv_dev *d, size_t bytes, size_t static_bytes = 0)
{
    static std::atomic<size_t> granted[64]; // per device ordinal; zero-initialised = the 64 KiB default
    const int dev = d->device >= 0 && d->device < 64 ? d->device : 0;
    if (bytes + static_bytes <= 64 * 1024 || bytes <= granted[dev].load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> g(g_lds_attr_lock); // the attribute only ever grows: a second thread must not set a smaller value after a larger one
    if (bytes <= granted[dev].load(std::memory_order_relaxed)) return;
    if (hipFuncSetAttribute((const void *) Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes) == hipSuccess) granted[dev].store(bytes, std::memory_order_release);
    else (void) hipGetLastError();
}

static int grid_for(long long work_items, int per_block, int cap)
{
    long long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int) g;
}
