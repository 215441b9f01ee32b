// shim/multi.hpp -- part of the single translation unit spmv_shim.hip: ROW BLOCKS OVER THE GPUS OF ONE PROCESS
// (BASELINE config 5; option "gpus").
//
// The reference's only distribution idea is its NUMA experiment: row blocks per memory node, x cut into
// contiguous slices, owner = col / slice, all inside one C program (src/samples/numa.c:277-304, 149-152).  This
// is its GPU analogue behind the UNCHANGED C signature: one host process, G devices, and per device
//   - a shard handle (spmv_dev) holding an equal-nnz block of rows (the splitter of parallel_balanced2_spmv.c:41-53
//     applied to RowPtr) with local int32 RowPtr and GLOBAL columns -- never a monolithic device array,
//   - a full-length copy of x (G * slice elements, slice = ceil(n / G)), a y block, a stream.
// One spmv() = bring x to every device, multiply everywhere at once, collect y:
//   x_exchange 0 "allgather"  every device receives ITS slice of the caller's X (host X: G partial H2D copies, one
//                             per PCIe link; device X: G device-to-device copies), then the slices are all-gathered
//                             over xGMI (ncclAllGather in place);
//   x_exchange 2 "bcast"      north_star's literal form: X goes to device 0 once and is broadcast (ncclBroadcast);
//   x_exchange 1 "range"      every device receives only x[col_min .. col_max] of ITS shard (known from create-time
//                             validation): for a banded matrix its own slice plus 16 values either side instead of
//                             the whole vector -- from the caller's X directly in spmv(), from the neighbours'
//                             slices by peer copies over xGMI in the distributed step.  No collective at all, and the
//                             copies run BESIDE the multiply: the rows that need them are redone afterwards
//                             (multi_plan_boundary).
// A solver-style caller keeps x and y DISTRIBUTED instead (spmv_hip_multi_x_slice / _y_slice / _step): the step is
// then all-gather + multiply with nothing crossing PCIe -- the path the 6x-at-8-GPUs target is about.
//
// RCCL is loaded with dlopen (librccl.so): the library has no link-time dependency on it, a one-GPU box never
// loads it, and when it is missing -- or when several shards share a device (SPMV_HIP_GPUS_VIRTUAL=1, how this
// file is tested on a one-GPU box) -- the same exchanges are done with hipMemcpyPeerAsync between the shards'
// buffers, ordered by events.  G > 1 on real devices has not been measured: no multi-GPU box was available.
// The matrix may also arrive as G row blocks of its own (spmv_shim_multi_create_blocks): config 5's 2.56e9 non-zeros do not
// fit one int32 RowPtr.
#pragma once
#include <dlfcn.h>

typedef struct ncclComm *ncclComm_t_;
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t_ *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t_) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

static RcclApi rccl_load()
{
    RcclApi a;
    if (getenv("SPMV_HIP_NO_RCCL")) return a;
    a.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) return a;
    a.CommInitAll = (decltype(a.CommInitAll)) dlsym(a.lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy)) dlsym(a.lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather)) dlsym(a.lib, "ncclAllGather");
    a.Broadcast = (decltype(a.Broadcast)) dlsym(a.lib, "ncclBroadcast");
    a.GroupStart = (decltype(a.GroupStart)) dlsym(a.lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd)) dlsym(a.lib, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString)) dlsym(a.lib, "ncclGetErrorString");
    a.ok = a.CommInitAll && a.CommDestroy && a.AllGather && a.Broadcast && a.GroupStart && a.GroupEnd;
    return a;
}

static RcclApi &rccl_api()
{
    static RcclApi a = rccl_load(); // initialised once, also when several threads create multi-GPU handles at the same time
    return a;
}

struct MultiShard {
    spmv_dev *dev = nullptr;
    spmv_dev *bnd = nullptr;       // "range" exchange with overlap: the rows that reference columns outside this shard's own slice of x, as a sub-matrix
    int device = 0;
    int row0 = 0, rows = 0;        // this shard's rows [row0, row0 + rows)
    long long p0 = 0, nnz = 0;     // ... and its range of the caller's ColIdx / Val (p0 < 0: the shard was handed over as a block of its own)
    void *x = nullptr, *y = nullptr; // x: G * slice elements (the whole vector); y: rows elements
    void *y_bnd = nullptr;         // results of the boundary rows, scattered into y
    int *bnd_rows = nullptr;       // their row numbers inside the shard
    int nbnd = 0;
    hipStream_t stream = nullptr;  // multiplies (and collectives)
    hipStream_t copy = nullptr;    // halo copies of the "range" exchange: run beside the interior multiply
    hipEvent_t ready = nullptr;    // "my slice of x (or, on device 0, all of x) is in place"
    hipEvent_t halo = nullptr;     // "the columns I pull from other devices' slices have arrived"
    hipEvent_t done = nullptr;     // "my multiplies of the last step have read x": the next step's halo copies wait for it
    ncclComm_t_ comm = nullptr;
};

struct spmv_multi {
    int G = 0, xchg = 0, m = 0, n = 0;
    size_t vsize = 8;
    long long slice = 0;           // elements of x per device: ceil(n / G)
    bool rccl = false;             // RCCL communicators are up (G distinct devices, librccl loaded)
    bool from_blocks = false;      // created from per-device CSR blocks: there is no monolithic value array to refresh from
    std::vector<MultiShard> sh;
};

static void multi_free(spmv_multi *mt)
{
    if (!mt) return;
    for (auto &s : mt->sh) {
        (void) hipSetDevice(s.device);
        if (s.comm && rccl_api().ok) (void) rccl_api().CommDestroy(s.comm);
        if (s.bnd) spmv_shim_matrix_destroy(s.bnd);
        if (s.dev) spmv_shim_matrix_destroy(s.dev);
        if (s.x) (void) pool_free(s.x);
        if (s.y) (void) pool_free(s.y);
        if (s.y_bnd) (void) pool_free(s.y_bnd);
        if (s.bnd_rows) (void) pool_free(s.bnd_rows);
        if (s.ready) (void) hipEventDestroy(s.ready);
        if (s.halo) (void) hipEventDestroy(s.halo);
        if (s.done) (void) hipEventDestroy(s.done);
        if (s.stream) (void) hipStreamDestroy(s.stream);
        if (s.copy) (void) hipStreamDestroy(s.copy);
    }
    delete mt;
}

extern "C" void spmv_shim_multi_destroy(spmv_multi *mt)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = -1; }
    multi_free(mt);
    if (cur >= 0) (void) hipSetDevice(cur);
}

// flags[r] = 1 if row r references a column outside [lo, hi); 16 lanes sweep a row
static __global__ __launch_bounds__(kBlock) void rows_outside_kernel(int m, const int *__restrict__ rowptr, const int *__restrict__ colidx, int lo, int hi, int *__restrict__ flags)
{
    const int sub = threadIdx.x / 16, l = threadIdx.x % 16;
    const long long stride = (long long) gridDim.x * (kBlock / 16);
    for (long long r = (long long) blockIdx.x * (kBlock / 16) + sub; r < m; r += stride) {
        int out = 0;
        for (int p = rowptr[r] + l; p < rowptr[r + 1]; p += 16) {
            const int c = colidx[p];
            out |= c < lo || c >= hi;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) out |= __shfl_xor(out, o, 16);
        if (l == 0) flags[r] = out;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(int count, const int *__restrict__ rows, const T *__restrict__ src, T *__restrict__ dst)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < count) dst[rows[i]] = src[i];
}

// "range" exchange with overlap (the halo of this host; spmv_amd/dist.py has the same idea per process): the rows of shard g that
// reference columns outside the shard's own slice of x become a sub-matrix of their own.  A step then multiplies the WHOLE shard
// while the halo copies run on a second stream -- the boundary rows come out wrong (they read columns still in flight) -- and, once
// the halo has landed, multiplies the boundary sub-matrix and overwrites exactly those rows.  No second copy of the interior, no row
// emptied.  Only when the boundary is under a tenth of the rows; a shard of a banded matrix has 2 x 16 of them.
template <typename T>
static int multi_plan_boundary(spmv_multi *mt, int g)
{
    MultiShard &s = mt->sh[(size_t) g];
    spmv_dev *d = s.dev;
    const long long lo = mt->slice * g, hi = std::min((long long) mt->n, mt->slice * (g + 1));
    if (d->m <= 0 || d->nnz <= 0 || (d->col_min >= lo && d->col_max < hi)) return SPMV_HIP_OK; // purely diagonal block: nothing to wait for
    const int nb = (int) (((long long) d->m + kScanTile - 1) / kScanTile);
    int *flags = nullptr, *sums = nullptr, *total = nullptr, *lens = nullptr, *rp_b = nullptr, *ci_b = nullptr, *scratch = nullptr;
    long long *rp64 = nullptr;
    T *va_b = nullptr;
    auto cleanup = [&]() { for (void *p : {(void *) flags, (void *) sums, (void *) total, (void *) lens, (void *) rp_b, (void *) ci_b, (void *) scratch, (void *) rp64, (void *) va_b}) if (p) (void) pool_free(p); };
    if (pool_malloc((void **) &flags, sizeof(int) * (size_t) d->m) != hipSuccess || pool_malloc((void **) &sums, sizeof(int) * (size_t) nb) != hipSuccess ||
        pool_malloc((void **) &total, sizeof(int)) != hipSuccess) { (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_ALLOC, "multi: boundary scratch"); }
    rows_outside_kernel<<<grid_for(d->m, kBlock / 16, d->cus * 16), kBlock, 0, s.stream>>>(d->m, d->rowptr, d->colidx, (int) lo, (int) hi, flags);
    scan_block_sums_kernel<<<nb, kBlock, 0, s.stream>>>(d->m, flags, sums);
    scan_sums_inplace_kernel<<<1, kBlock, 0, s.stream>>>(nb, sums, total);
    int nbnd = 0;
    if (hipMemcpyAsync(&nbnd, total, sizeof(int), hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) {
        (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_RUNTIME, "multi: boundary scan failed");
    }
    if (nbnd == 0 || (long long) nbnd * 10 > d->m) { cleanup(); return SPMV_HIP_OK; } // no halo at all (cannot happen here) / too many boundary rows: wait for the halo first
    long long sub_nnz = 0;
    if (pool_malloc((void **) &s.bnd_rows, sizeof(int) * (size_t) nbnd) != hipSuccess || pool_malloc((void **) &scratch, sizeof(int) * (size_t) nbnd) != hipSuccess ||
        pool_malloc((void **) &lens, sizeof(int) * (size_t) nbnd) != hipSuccess || pool_malloc((void **) &rp64, sizeof(long long) * ((size_t) nbnd + 1)) != hipSuccess ||
        pool_malloc((void **) &rp_b, sizeof(int) * ((size_t) nbnd + 1)) != hipSuccess) { (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_ALLOC, "multi: boundary rows"); }
    csr5_compact_kernel<<<nb, kBlock, 0, s.stream>>>(d->m, flags, sums, d->rowptr, scratch, s.bnd_rows);
    long_rows_len_kernel<<<grid_for(nbnd, kBlock, INT_MAX), kBlock, 0, s.stream>>>(nbnd, s.bnd_rows, d->rowptr, lens);
    scan_i32_to_i64_kernel<<<1, kBlock, 0, s.stream>>>(nbnd, lens, rp64);
    if (hipMemcpyAsync(&sub_nnz, rp64 + nbnd, sizeof(long long), hipMemcpyDeviceToHost, s.stream) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) {
        (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_RUNTIME, "multi: boundary lengths failed");
    }
    if (pool_malloc((void **) &ci_b, sizeof(int) * (size_t) (sub_nnz > 0 ? sub_nnz : 1)) != hipSuccess || pool_malloc((void **) &va_b, sizeof(T) * (size_t) (sub_nnz > 0 ? sub_nnz : 1)) != hipSuccess ||
        pool_malloc(&s.y_bnd, sizeof(T) * (size_t) nbnd) != hipSuccess) { (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_ALLOC, "multi: boundary sub-matrix"); }
    narrow_i64_kernel<<<grid_for((long long) nbnd + 1, kBlock, INT_MAX), kBlock, 0, s.stream>>>(nbnd + 1, rp64, rp_b);
    long_rows_gather_kernel<T><<<nbnd, kBlock, 0, s.stream>>>(s.bnd_rows, d->rowptr, d->colidx, (const T *) d->val, rp_b, ci_b, va_b);
    if (hipStreamSynchronize(s.stream) != hipSuccess) { (void) hipGetLastError(); cleanup(); return fail(SPMV_HIP_E_RUNTIME, "multi: boundary gather failed"); }
    const int rc = spmv_shim_matrix_create(&s.bnd, nbnd, mt->n, rp_b, ci_b, va_b, mt->vsize); // copies the three arrays
    cleanup();
    if (rc) return rc;
    (void) spmv_shim_set_stream(s.bnd, s.stream);
    (void) spmv_shim_set_async(s.bnd, 1);
    s.nbnd = nbnd;
    return SPMV_HIP_OK;
}

// Common part of the two create functions: shard g = rows[g] rows with LOCAL RowPtr (host copy lrp[g]), columns / values at
// colidx[g] / val[g] (host or device pointers).
static int multi_build(spmv_multi *mt, int ndev, const std::vector<std::vector<int>> &lrp, const int *const *colidx, const void *const *val, int cur)
{
    const int G = mt->G, n = mt->n;
    auto bail = [&](int code) { multi_free(mt); (void) hipSetDevice(cur); return code; };
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        s.device = g % ndev;
        if (hipSetDevice(s.device) != hipSuccess) { (void) hipGetLastError(); return bail(fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d)", s.device)); }
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s.copy, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.halo, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            (void) hipGetLastError();
            return bail(fail(SPMV_HIP_E_RUNTIME, "multi: stream / event creation failed on device %d", s.device));
        }
        const int rc = spmv_shim_matrix_create(&s.dev, s.rows, n, lrp[(size_t) g].data(), colidx[g], val[g], mt->vsize);
        if (rc) return bail(rc);
        (void) spmv_shim_set_stream(s.dev, s.stream);
        (void) spmv_shim_set_async(s.dev, 1);
        const size_t xb = mt->vsize * (size_t) (mt->slice * G > 0 ? mt->slice * G : 1), yb = mt->vsize * (size_t) (s.rows > 0 ? s.rows : 1);
        if (pool_malloc(&s.x, xb) != hipSuccess || pool_malloc(&s.y, yb) != hipSuccess) {
            (void) hipGetLastError();
            return bail(fail(SPMV_HIP_E_ALLOC, "multi: x / y buffers on device %d", s.device));
        }
        (void) hipMemsetAsync(s.x, 0, xb, s.stream);
        if (mt->xchg == 1 && G > 1 && !getenv("SPMV_HIP_MULTI_NO_OVERLAP")) {
            const int rb = mt->vsize == sizeof(double) ? multi_plan_boundary<double>(mt, g) : multi_plan_boundary<float>(mt, g);
            if (rb) return bail(rb);
        }
    }
    // RCCL only between distinct devices (one rank per device); peer copies otherwise
    bool distinct = G <= ndev;
    if (G > 1 && distinct && mt->xchg != 1 && rccl_api().ok) {
        std::vector<int> devs((size_t) G);
        std::vector<ncclComm_t_> comms((size_t) G, nullptr);
        for (int g = 0; g < G; ++g) devs[(size_t) g] = mt->sh[(size_t) g].device;
        if (rccl_api().CommInitAll(comms.data(), G, devs.data()) == 0) {
            for (int g = 0; g < G; ++g) mt->sh[(size_t) g].comm = comms[(size_t) g];
            mt->rccl = true;
        }
    } else if (G == 1 && getenv("SPMV_HIP_RCCL_SINGLE") && rccl_api().ok) { // exercise the RCCL calls with one rank (test hook)
        int dv = mt->sh[0].device;
        ncclComm_t_ c = nullptr;
        if (rccl_api().CommInitAll(&c, 1, &dv) == 0) { mt->sh[0].comm = c; mt->rccl = true; }
    }
    if (!mt->rccl && G > 1)
        for (int g = 0; g < G; ++g) { // peer copies: let every device reach the others (no-op between shards of one device)
            (void) hipSetDevice(mt->sh[(size_t) g].device);
            for (int h = 0; h < G; ++h) {
                int can = 0;
                const int dg = mt->sh[(size_t) g].device, dh = mt->sh[(size_t) h].device;
                if (dg != dh && hipDeviceCanAccessPeer(&can, dg, dh) == hipSuccess && can) (void) hipDeviceEnablePeerAccess(dh, 0);
            }
            (void) hipGetLastError();
        }
    (void) hipSetDevice(cur);
    return SPMV_HIP_OK;
}

static int multi_device_count(int *ndev)
{
    *ndev = 0;
    if (hipGetDeviceCount(ndev) != hipSuccess || *ndev <= 0) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_E_NODEVICE, "no HIP device visible (this library has no CPU path)");
    }
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_multi_create(spmv_multi **out, int gpus, int xchg, int m, int n, const int *rowptr, const int *colidx,
                                      const void *val, size_t value_size)
{
    *out = nullptr;
    int ndev = 0;
    if (multi_device_count(&ndev)) return SPMV_HIP_E_NODEVICE;
    if (m < 0 || n < 0 || (m > 0 && !rowptr)) return fail(SPMV_HIP_E_ARG, "multi: negative size or NULL RowPtr");
    const bool virt = getenv("SPMV_HIP_GPUS_VIRTUAL") != nullptr; // testing: several shards may share a device
    int G = gpus < 1 ? 1 : gpus;
    if (!virt && G > ndev) G = ndev;
    if (G > 64) G = 64;
    if (G > m) G = m; // never more shards than rows; an empty matrix is one (empty) shard
    if (G < 1) G = 1;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    // RowPtr on the host (the splitter is a few binary searches; reference: parallel_balanced2_spmv.c:41-53)
    std::vector<int> rp((size_t) m + 1, 0);
    if (m > 0 && hipMemcpy(rp.data(), rowptr, sizeof(int) * ((size_t) m + 1), hipMemcpyDefault) != hipSuccess) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_E_RUNTIME, "multi: cannot read RowPtr");
    }
    const long long nnz = m > 0 ? rp[m] : 0;
    if (m > 0 && rp[0] != 0) return fail(SPMV_HIP_E_ARG, "RowPtr must start at 0 (RowPtr[0]=%d)", rp[0]);
    for (int i = 0; i < m; ++i)
        if (rp[i] > rp[i + 1]) return fail(SPMV_HIP_E_ARG, "RowPtr must be non-decreasing (row %d)", i);
    spmv_multi *mt = new spmv_multi();
    mt->G = G; mt->xchg = xchg == 2 ? 2 : (xchg == 1 ? 1 : 0); mt->m = m; mt->n = n;
    mt->vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float);
    mt->slice = ((long long) n + G - 1) / G;
    mt->sh.resize((size_t) G);
    std::vector<int> cut((size_t) G + 1, 0);
    cut[G] = m;
    for (int g = 1; g < G; ++g) { // first row whose start is at or past g * nnz / G, kept monotone; never an empty shard while rows remain
        const long long key = nnz * g / G;
        int r = (int) (std::upper_bound(rp.begin(), rp.end(), (int) key) - rp.begin()) - 1;
        if (r < cut[g - 1] + 1) r = cut[g - 1] + 1;
        if (r > m - (G - g)) r = m - (G - g);
        cut[g] = r;
    }
    std::vector<std::vector<int>> lrp((size_t) G);
    std::vector<const int *> cols((size_t) G, nullptr);
    std::vector<const void *> vals((size_t) G, nullptr);
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        s.row0 = cut[g];
        s.rows = cut[g + 1] - cut[g];
        s.p0 = m > 0 ? rp[s.row0] : 0;
        s.nnz = m > 0 ? rp[s.row0 + s.rows] - s.p0 : 0;
        lrp[(size_t) g].assign((size_t) s.rows + 1, 0);
        for (int i = 0; i <= s.rows; ++i) lrp[(size_t) g][(size_t) i] = rp[(size_t) s.row0 + i] - (int) s.p0;
        cols[(size_t) g] = colidx ? colidx + s.p0 : nullptr;
        vals[(size_t) g] = val ? (const char *) val + mt->vsize * (size_t) s.p0 : nullptr;
    }
    const int rc = multi_build(mt, ndev, lrp, cols.data(), vals.data(), cur);
    if (rc) return rc;
    *out = mt;
    return SPMV_HIP_OK;
}

// The matrix arrives as G row blocks of its own -- block g: rows[g] rows, LOCAL 0-based int32 RowPtr, GLOBAL column indices --
// the way the reference's NUMA experiment hands every node its block (numa.c:277-304).  No monolithic array exists, so the
// int32 limit applies per block: BASELINE config 5 (8 x 1e7 rows x 32 = 2.56e9 non-zeros) is expressible.  Block g goes to
// device g mod (visible devices); pointers may be host or device memory (of any device).
extern "C" int spmv_shim_multi_create_blocks(spmv_multi **out, int G, int xchg, const int *rows, int n, const int *const *rowptr, const int *const *colidx,
                                             const void *const *val, size_t value_size)
{
    *out = nullptr;
    int ndev = 0;
    if (multi_device_count(&ndev)) return SPMV_HIP_E_NODEVICE;
    if (G < 1 || G > 64 || n < 0 || !rows || !rowptr || !colidx || !val) return fail(SPMV_HIP_E_ARG, "multi: 1..64 blocks with non-NULL array lists expected");
    const bool virt = getenv("SPMV_HIP_GPUS_VIRTUAL") != nullptr;
    if (!virt && G > ndev) return fail(SPMV_HIP_E_ARG, "multi: %d row blocks but %d device(s) (SPMV_HIP_GPUS_VIRTUAL=1 lets blocks share a device)", G, ndev);
    long long m = 0;
    for (int g = 0; g < G; ++g) {
        if (rows[g] < 0 || (rows[g] > 0 && !rowptr[g])) return fail(SPMV_HIP_E_ARG, "multi: block %d has a negative row count or NULL RowPtr", g);
        m += rows[g];
    }
    if (m > INT_MAX) return fail(SPMV_HIP_E_RANGE, "multi: %lld rows in all exceed the int row index of spmv()", m);
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    spmv_multi *mt = new spmv_multi();
    mt->G = G; mt->xchg = xchg == 2 ? 2 : (xchg == 1 ? 1 : 0); mt->m = (int) m; mt->n = n;
    mt->vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float);
    mt->slice = ((long long) n + G - 1) / G;
    mt->from_blocks = true;
    mt->sh.resize((size_t) G);
    std::vector<std::vector<int>> lrp((size_t) G);
    int row0 = 0;
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        s.row0 = row0;
        s.rows = rows[g];
        row0 += rows[g];
        s.p0 = -1;
        lrp[(size_t) g].assign((size_t) s.rows + 1, 0);
        if (s.rows > 0 && hipMemcpy(lrp[(size_t) g].data(), rowptr[g], sizeof(int) * ((size_t) s.rows + 1), hipMemcpyDefault) != hipSuccess) {
            (void) hipGetLastError();
            delete mt;
            return fail(SPMV_HIP_E_RUNTIME, "multi: cannot read RowPtr of block %d", g);
        }
        s.nnz = s.rows > 0 ? lrp[(size_t) g][(size_t) s.rows] : 0;
    }
    const int rc = multi_build(mt, ndev, lrp, colidx, val, cur);
    if (rc) return rc;
    *out = mt;
    return SPMV_HIP_OK;
}

extern "C" int spmv_shim_multi_count(const spmv_multi *mt) { return mt ? mt->G : 0; }
extern "C" long long spmv_shim_multi_nnz(const spmv_multi *mt)
{
    long long t = 0;
    if (mt) for (const auto &s : mt->sh) t += s.nnz;
    return t;
}
extern "C" int spmv_shim_multi_rows(const spmv_multi *mt) { return mt ? mt->m : 0; }
extern "C" int spmv_shim_multi_uses_rccl(const spmv_multi *mt) { return mt && mt->rccl ? 1 : 0; }
extern "C" spmv_dev *spmv_shim_multi_shard(spmv_multi *mt, int g) { return mt && g >= 0 && g < mt->G ? mt->sh[(size_t) g].dev : nullptr; }
// the boundary sub-matrix of shard g ("range" exchange with overlap), NULL if the shard has none: planned and built like a shard
extern "C" spmv_dev *spmv_shim_multi_boundary(spmv_multi *mt, int g) { return mt && g >= 0 && g < mt->G ? mt->sh[(size_t) g].bnd : nullptr; }

extern "C" int spmv_shim_multi_slices(spmv_multi *mt, int g, void **x_slice, long long *x_first, long long *x_count, void **y_block,
                                      long long *y_first, long long *y_count, int *device)
{
    if (!mt || g < 0 || g >= mt->G) return fail(SPMV_HIP_E_ARG, "multi: no shard %d", g);
    const MultiShard &s = mt->sh[(size_t) g];
    const long long first = mt->slice * g, cnt = std::max(0ll, std::min((long long) mt->n, first + mt->slice) - first);
    if (x_slice) *x_slice = (char *) s.x + mt->vsize * (size_t) first;
    if (x_first) *x_first = first;
    if (x_count) *x_count = cnt;
    if (y_block) *y_block = s.y;
    if (y_first) *y_first = s.row0;
    if (y_count) *y_count = s.rows;
    if (device) *device = s.device;
    return SPMV_HIP_OK;
}

// The x exchange between the shards' buffers, every device's own slice (allgather) or device 0's whole vector
// (bcast) being in place once the shard's `ready` event has fired.  "range": the copies go to the shard's COPY stream and
// s.halo fires when they have landed; the collectives / full peer copies are ordered on the multiply stream itself.
static int multi_exchange(spmv_multi *mt)
{
    const int G = mt->G;
    if (G == 1 && !mt->rccl) return SPMV_HIP_OK;
    const size_t sb = mt->vsize * (size_t) mt->slice;
    if (mt->xchg == 1) { // "range": pull the referenced columns that live in other devices' slices (peer copies, no collective)
        for (int g = 0; g < G; ++g) {
            MultiShard &s = mt->sh[(size_t) g];
            const long long lo = s.dev->col_min, hi = (long long) s.dev->col_max + 1; // [lo, hi)
            (void) hipSetDevice(s.device);
            HIP_TRY(hipStreamWaitEvent(s.copy, s.done, 0)); // two steps enqueued back to back: the last step's boundary rows still read the halo
            for (int h = 0; h < G && hi > lo; ++h) {
                if (h == g) continue;
                const MultiShard &o = mt->sh[(size_t) h];
                const long long a = std::max(lo, mt->slice * h), b = std::min(hi, std::min((long long) mt->n, mt->slice * (h + 1)));
                if (b <= a) continue;
                HIP_TRY(hipStreamWaitEvent(s.copy, o.ready, 0));
                HIP_TRY(hipMemcpyPeerAsync((char *) s.x + mt->vsize * (size_t) a, s.device, (const char *) o.x + mt->vsize * (size_t) a, o.device,
                                           mt->vsize * (size_t) (b - a), s.copy));
            }
            HIP_TRY(hipEventRecord(s.halo, s.copy));
        }
        return SPMV_HIP_OK;
    }
    if (mt->rccl) {
        RcclApi &R = rccl_api();
        int rc = R.GroupStart();
        for (int g = 0; g < G && !rc; ++g) {
            MultiShard &s = mt->sh[(size_t) g];
            (void) hipSetDevice(s.device);
            if (mt->xchg == 2) rc = R.Broadcast(s.x, s.x, sb * (size_t) G, 0 /* ncclInt8 */, 0, s.comm, s.stream);
            else rc = R.AllGather((const char *) s.x + sb * (size_t) g, s.x, sb, 0 /* ncclInt8 */, s.comm, s.stream);
        }
        const int rc2 = R.GroupEnd();
        if (rc || rc2) return fail(SPMV_HIP_E_RUNTIME, "RCCL x exchange failed: %s", R.GetErrorString ? R.GetErrorString(rc ? rc : rc2) : "?");
        return SPMV_HIP_OK;
    }
    for (int g = 0; g < G; ++g) { // peer copies, pulled by the receiver on its own stream
        MultiShard &s = mt->sh[(size_t) g];
        (void) hipSetDevice(s.device);
        if (mt->xchg == 2) {
            if (g == 0) continue;
            HIP_TRY(hipStreamWaitEvent(s.stream, mt->sh[0].ready, 0));
            HIP_TRY(hipMemcpyPeerAsync(s.x, s.device, mt->sh[0].x, mt->sh[0].device, sb * (size_t) G, s.stream));
        } else {
            for (int h = 0; h < G; ++h) {
                if (h == g) continue;
                const MultiShard &o = mt->sh[(size_t) h];
                HIP_TRY(hipStreamWaitEvent(s.stream, o.ready, 0));
                HIP_TRY(hipMemcpyPeerAsync((char *) s.x + sb * (size_t) h, s.device, (const char *) o.x + sb * (size_t) h, o.device, sb, s.stream));
            }
        }
    }
    return SPMV_HIP_OK;
}

// multiply of shard g on its stream; range exchange: beside / behind the halo (see multi_plan_boundary)
static int multi_multiply(spmv_multi *mt, int g, bool halo_pending)
{
    MultiShard &s = mt->sh[(size_t) g];
    (void) hipSetDevice(s.device);
    if (halo_pending && !s.bnd) HIP_TRY(hipStreamWaitEvent(s.stream, s.halo, 0)); // no split: the halo first
    int rc = spmv_shim_run(s.dev, s.x, s.y);
    if (rc) return rc;
    if (halo_pending && s.bnd) {
        HIP_TRY(hipStreamWaitEvent(s.stream, s.halo, 0));
        rc = spmv_shim_run(s.bnd, s.x, s.y_bnd);
        if (rc) return rc;
        if (mt->vsize == sizeof(double)) scatter_rows_kernel<double><<<grid_for(s.nbnd, kBlock, INT_MAX), kBlock, 0, s.stream>>>(s.nbnd, s.bnd_rows, (const double *) s.y_bnd, (double *) s.y);
        else scatter_rows_kernel<float><<<grid_for(s.nbnd, kBlock, INT_MAX), kBlock, 0, s.stream>>>(s.nbnd, s.bnd_rows, (const float *) s.y_bnd, (float *) s.y);
        HIP_TRY(hipGetLastError());
    }
    if (halo_pending) HIP_TRY(hipEventRecord(s.done, s.stream));
    return SPMV_HIP_OK;
}

// exchange (the x slices / device 0's x are already in the shard buffers) + multiply on every device; y stays in the
// shards' blocks.  Everything is ENQUEUED: the shards' streams are ordered behind whatever the caller has submitted to each
// device's default stream (an event recorded there -- no host-side synchronisation, round 2 called hipDeviceSynchronize per
// device and step), and spmv_shim_multi_sync waits for the results.
extern "C" int spmv_shim_multi_step_async(spmv_multi *mt)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    int rc = SPMV_HIP_OK;
    for (auto &s : mt->sh) { // "my slice is in place" = everything submitted to the device's default stream so far
        (void) hipSetDevice(s.device);
        if (hipEventRecord(s.ready, nullptr) != hipSuccess || hipStreamWaitEvent(s.stream, s.ready, 0) != hipSuccess) {
            (void) hipGetLastError();
            rc = fail(SPMV_HIP_E_RUNTIME, "multi: event record on device %d", s.device);
        }
    }
    if (!rc) rc = multi_exchange(mt);
    const bool halo = mt->xchg == 1 && mt->G > 1;
    for (int g = 0; g < mt->G && !rc; ++g) rc = multi_multiply(mt, g, halo);
    (void) hipSetDevice(cur);
    return rc;
}

extern "C" int spmv_shim_multi_sync(spmv_multi *mt)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    int cur = -1, rc = SPMV_HIP_OK;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    for (auto &s : mt->sh) {
        (void) hipSetDevice(s.device);
        if ((hipStreamSynchronize(s.stream) != hipSuccess || hipStreamSynchronize(s.copy) != hipSuccess) && !rc) {
            (void) hipGetLastError();
            rc = fail(SPMV_HIP_E_RUNTIME, "multi: stream of device %d failed", s.device);
        }
    }
    (void) hipSetDevice(cur);
    return rc;
}

// synchronous form (like spmv()): returns when every y block is complete
extern "C" int spmv_shim_multi_step(spmv_multi *mt)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    // The synchronous entry keeps its round-2 contract: the x slices may have been written on ANY stream of their device (side streams,
    // non-blocking streams), so every device is drained first.  Only the _async entry relies on the default-stream ordering.
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
        for (auto &s : mt->sh) {
            (void) hipSetDevice(s.device);
            if (hipDeviceSynchronize() != hipSuccess) { (void) hipGetLastError(); (void) hipSetDevice(cur); return fail(SPMV_HIP_E_RUNTIME, "multi: device %d failed before the step", s.device); }
        }
        (void) hipSetDevice(cur);
    }
    const int rc = spmv_shim_multi_step_async(mt);
    const int rc2 = mt ? spmv_shim_multi_sync(mt) : SPMV_HIP_OK;
    return rc ? rc : rc2;
}

// y = A x with FULL vectors x (n) and y (m), host or device pointers -- the drop-in spmv() of a multi-GPU handle.
extern "C" int spmv_shim_multi_run(spmv_multi *mt, const void *x, void *y)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    if ((mt->n > 0 && !x) || (mt->m > 0 && !y)) return fail(SPMV_HIP_E_ARG, "run: X or Y is NULL");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    const int G = mt->G;
    const size_t vs = mt->vsize;
    int rc = SPMV_HIP_OK;
    auto done = [&](int code) { (void) hipSetDevice(cur); return code; };
    // 1. x to the devices
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        if (hipSetDevice(s.device) != hipSuccess) { (void) hipGetLastError(); return done(fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d)", s.device)); }
        hipError_t e = hipSuccess;
        if (mt->xchg == 1) { // range: exactly the columns this shard references, straight from the caller's vector
            const long long lo = s.dev->col_min, cnt = (long long) s.dev->col_max + 1 - lo;
            if (cnt > 0) e = hipMemcpyAsync((char *) s.x + vs * (size_t) lo, (const char *) x + vs * (size_t) lo, vs * (size_t) cnt, hipMemcpyDefault, s.stream);
        } else if (mt->xchg == 2 || G == 1) { // bcast: the whole vector to device 0 only (G == 1: that is everything)
            if (g == 0 && mt->n > 0) e = hipMemcpyAsync(s.x, x, vs * (size_t) mt->n, hipMemcpyDefault, s.stream);
        } else {
            const long long first = mt->slice * g, cnt = std::max(0ll, std::min((long long) mt->n, first + mt->slice) - first);
            if (cnt > 0) e = hipMemcpyAsync((char *) s.x + vs * (size_t) first, (const char *) x + vs * (size_t) first, vs * (size_t) cnt, hipMemcpyDefault, s.stream);
        }
        if (e == hipSuccess) e = hipEventRecord(s.ready, s.stream);
        if (e != hipSuccess) { (void) hipGetLastError(); return done(fail(SPMV_HIP_E_RUNTIME, "multi: x upload to device %d: %s", s.device, hipGetErrorString(e))); }
    }
    // 2. exchange over xGMI (range mode: every device already took what it needs from X), 3. multiply everywhere,
    // 4. y blocks back to the caller's vector
    if (mt->xchg != 1) rc = multi_exchange(mt);
    for (int g = 0; g < G && !rc; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        rc = multi_multiply(mt, g, false);
        if (!rc && s.rows > 0 && hipMemcpyAsync((char *) y + vs * (size_t) s.row0, s.y, vs * (size_t) s.rows, hipMemcpyDefault, s.stream) != hipSuccess) {
            (void) hipGetLastError();
            rc = fail(SPMV_HIP_E_RUNTIME, "multi: y download from device %d", s.device);
        }
    }
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        (void) hipSetDevice(s.device);
        if (hipStreamSynchronize(s.stream) != hipSuccess && !rc) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_RUNTIME, "multi: stream of device %d failed", s.device); }
    }
    return done(rc);
}

extern "C" int spmv_shim_multi_update_values(spmv_multi *mt, const void *val)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    if (mt->from_blocks) return fail(SPMV_HIP_E_ARG, "update_values: the handle was created from row blocks; there is no monolithic value array (clear and create again)");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    int rc = SPMV_HIP_OK;
    for (auto &s : mt->sh) {
        rc = spmv_shim_update_values(s.dev, (const char *) val + mt->vsize * (size_t) s.p0);
        if (!rc && s.bnd && s.bnd->nnz > 0) { // the boundary rows' copy: gathered from the shard's refreshed values
            (void) hipSetDevice(s.device);
            void *tmp = nullptr;
            if (pool_malloc(&tmp, mt->vsize * (size_t) s.bnd->nnz) != hipSuccess) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_ALLOC, "update_values: boundary scratch"); break; }
            if (mt->vsize == sizeof(double)) long_rows_gather_kernel<double><<<s.nbnd, kBlock, 0, s.stream>>>(s.bnd_rows, s.dev->rowptr, nullptr, (const double *) s.dev->val, s.bnd->rowptr, nullptr, (double *) tmp);
            else long_rows_gather_kernel<float><<<s.nbnd, kBlock, 0, s.stream>>>(s.bnd_rows, s.dev->rowptr, nullptr, (const float *) s.dev->val, s.bnd->rowptr, nullptr, (float *) tmp);
            if (hipStreamSynchronize(s.stream) != hipSuccess) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_RUNTIME, "update_values: boundary gather"); }
            if (!rc) rc = spmv_shim_update_values(s.bnd, tmp);
            (void) pool_free(tmp);
        }
        if (rc) break;
    }
    (void) hipSetDevice(cur);
    return rc;
}
