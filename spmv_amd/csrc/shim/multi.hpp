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
//                             slices by peer copies over xGMI in the distributed step.  No collective at all.
// A solver-style caller keeps x and y DISTRIBUTED instead (spmv_hip_multi_x_slice / _y_slice / _step): the step is
// then all-gather + multiply with nothing crossing PCIe -- the path the 6x-at-8-GPUs target is about.
//
// RCCL is loaded with dlopen (librccl.so): the library has no link-time dependency on it, a one-GPU box never
// loads it, and when it is missing -- or when several shards share a device (SPMV_HIP_GPUS_VIRTUAL=1, how this
// file is tested on a one-GPU box) -- the same exchanges are done with hipMemcpyPeerAsync between the shards'
// buffers, ordered by events.  G > 1 on real devices has not been measured: no multi-GPU box was available.
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
    int device = 0;
    int row0 = 0, rows = 0;        // this shard's rows [row0, row0 + rows)
    long long p0 = 0, nnz = 0;     // ... and its range of the caller's ColIdx / Val
    void *x = nullptr, *y = nullptr; // x: G * slice elements (the whole vector); y: rows elements
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;    // "my slice of x (or, on device 0, all of x) is in place"
    ncclComm_t_ comm = nullptr;
};

struct spmv_multi {
    int G = 0, xchg = 0, m = 0, n = 0;
    size_t vsize = 8;
    long long slice = 0;           // elements of x per device: ceil(n / G)
    bool rccl = false;             // RCCL communicators are up (G distinct devices, librccl loaded)
    std::vector<MultiShard> sh;
};

static void multi_free(spmv_multi *mt)
{
    if (!mt) return;
    for (auto &s : mt->sh) {
        (void) hipSetDevice(s.device);
        if (s.comm && rccl_api().ok) (void) rccl_api().CommDestroy(s.comm);
        if (s.dev) spmv_shim_matrix_destroy(s.dev);
        if (s.x) (void) pool_free(s.x);
        if (s.y) (void) pool_free(s.y);
        if (s.ready) (void) hipEventDestroy(s.ready);
        if (s.stream) (void) hipStreamDestroy(s.stream);
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

extern "C" int spmv_shim_multi_create(spmv_multi **out, int gpus, int xchg, int m, int n, const int *rowptr, const int *colidx,
                                      const void *val, size_t value_size)
{
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void) hipGetLastError();
        return fail(SPMV_HIP_E_NODEVICE, "no HIP device visible (this library has no CPU path)");
    }
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
    auto bail = [&](int code) { multi_free(mt); (void) hipSetDevice(cur); return code; };
    std::vector<int> cut((size_t) G + 1, 0);
    cut[G] = m;
    for (int g = 1; g < G; ++g) { // first row whose start is at or past g * nnz / G, kept monotone; never an empty shard while rows remain
        const long long key = nnz * g / G;
        int r = (int) (std::upper_bound(rp.begin(), rp.end(), (int) key) - rp.begin()) - 1;
        if (r < cut[g - 1] + 1) r = cut[g - 1] + 1;
        if (r > m - (G - g)) r = m - (G - g);
        cut[g] = r;
    }
    std::vector<int> local;
    for (int g = 0; g < G; ++g) {
        MultiShard &s = mt->sh[(size_t) g];
        s.device = g % ndev;
        s.row0 = cut[g];
        s.rows = cut[g + 1] - cut[g];
        s.p0 = m > 0 ? rp[s.row0] : 0;
        s.nnz = m > 0 ? rp[s.row0 + s.rows] - s.p0 : 0;
        if (hipSetDevice(s.device) != hipSuccess) { (void) hipGetLastError(); return bail(fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d)", s.device)); }
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess) {
            (void) hipGetLastError();
            return bail(fail(SPMV_HIP_E_RUNTIME, "multi: stream / event creation failed on device %d", s.device));
        }
        local.assign((size_t) s.rows + 1, 0);
        for (int i = 0; i <= s.rows; ++i) local[(size_t) i] = rp[(size_t) s.row0 + i] - (int) s.p0;
        const int rc = spmv_shim_matrix_create(&s.dev, s.rows, n, local.data(), colidx ? colidx + s.p0 : nullptr,
                                               val ? (const char *) val + mt->vsize * (size_t) s.p0 : nullptr, mt->vsize);
        if (rc) return bail(rc);
        (void) spmv_shim_set_stream(s.dev, s.stream);
        (void) spmv_shim_set_async(s.dev, 1);
        const size_t xb = mt->vsize * (size_t) (mt->slice * G > 0 ? mt->slice * G : 1), yb = mt->vsize * (size_t) (s.rows > 0 ? s.rows : 1);
        if (pool_malloc(&s.x, xb) != hipSuccess || pool_malloc(&s.y, yb) != hipSuccess) {
            (void) hipGetLastError();
            return bail(fail(SPMV_HIP_E_ALLOC, "multi: x / y buffers on device %d", s.device));
        }
        (void) hipMemsetAsync(s.x, 0, xb, s.stream);
    }
    // RCCL only between distinct devices (one rank per device); peer copies otherwise
    bool distinct = G <= ndev;
    if (G > 1 && distinct && rccl_api().ok) {
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
extern "C" int spmv_shim_multi_uses_rccl(const spmv_multi *mt) { return mt && mt->rccl ? 1 : 0; }
extern "C" spmv_dev *spmv_shim_multi_shard(spmv_multi *mt, int g) { return mt && g >= 0 && g < mt->G ? mt->sh[(size_t) g].dev : nullptr; }

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
// (bcast) being in place once the shard's `ready` event has fired.
static int multi_exchange(spmv_multi *mt)
{
    const int G = mt->G;
    if (G == 1 && !mt->rccl) return SPMV_HIP_OK;
    const size_t sb = mt->vsize * (size_t) mt->slice;
    if (mt->xchg == 1) { // "range": pull the referenced columns that live in other devices' slices (peer copies, no collective)
        for (int g = 0; g < G; ++g) {
            MultiShard &s = mt->sh[(size_t) g];
            const long long lo = s.dev->col_min, hi = (long long) s.dev->col_max + 1; // [lo, hi)
            if (hi <= lo) continue;
            (void) hipSetDevice(s.device);
            for (int h = 0; h < G; ++h) {
                if (h == g) continue;
                const MultiShard &o = mt->sh[(size_t) h];
                const long long a = std::max(lo, mt->slice * h), b = std::min(hi, std::min((long long) mt->n, mt->slice * (h + 1)));
                if (b <= a) continue;
                HIP_TRY(hipStreamWaitEvent(s.stream, o.ready, 0));
                HIP_TRY(hipMemcpyPeerAsync((char *) s.x + mt->vsize * (size_t) a, s.device, (const char *) o.x + mt->vsize * (size_t) a, o.device,
                                           mt->vsize * (size_t) (b - a), s.stream));
            }
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

// exchange (the x slices / device 0's x are already in the shard buffers) + multiply on every device; y stays in the
// shards' blocks.  Returns after everything has been enqueued and completed (synchronous like spmv()).
extern "C" int spmv_shim_multi_step(spmv_multi *mt)
{
    if (!mt) return fail(SPMV_HIP_E_ARG, "multi: NULL");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); cur = 0; }
    int rc = SPMV_HIP_OK;
    for (auto &s : mt->sh) { // the caller wrote the slices on whatever stream: order our streams behind the device's work
        (void) hipSetDevice(s.device);
        if (hipDeviceSynchronize() != hipSuccess) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_RUNTIME, "multi: device %d failed to synchronise", s.device); }
        if (!rc && hipEventRecord(s.ready, s.stream) != hipSuccess) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_RUNTIME, "multi: event record"); }
    }
    if (!rc) rc = multi_exchange(mt);
    for (size_t g = 0; g < mt->sh.size() && !rc; ++g) {
        MultiShard &s = mt->sh[g];
        (void) hipSetDevice(s.device);
        rc = spmv_shim_run(s.dev, s.x, s.y);
    }
    for (auto &s : mt->sh) {
        (void) hipSetDevice(s.device);
        if (hipStreamSynchronize(s.stream) != hipSuccess && !rc) { (void) hipGetLastError(); rc = fail(SPMV_HIP_E_RUNTIME, "multi: stream of device %d failed", s.device); }
    }
    (void) hipSetDevice(cur);
    return rc;
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
        (void) hipSetDevice(s.device);
        rc = spmv_shim_run(s.dev, s.x, s.y);
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
    for (auto &s : mt->sh) {
        const int rc = spmv_shim_update_values(s.dev, (const char *) val + mt->vsize * (size_t) s.p0);
        if (rc) return rc;
    }
    return SPMV_HIP_OK;
}
