// gbench.hip -- microbenchmark behind DESIGN.md 3.7: how many scattered 8-byte (or 4-byte) reads per second
// can the chip serve from a region that is L2-resident / Infinity-Cache-resident, per load flavour?
//   build: hipcc -O3 --offload-arch=gfx950 tools/gbench.hip -o build/gbench      run: build/gbench
// Every lane reads G random elements per step from a table of `bytes` (indices from a hash, no index stream),
// 16 loads in flight per lane; all waves use the same table, so after the first touch an XCD's L2 holds it
// (tables <= 2 MiB) or the Infinity Cache does (tables of 64 MiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE> __device__ __forceinline__ void ld8(double &v, const double *p)
{
    if constexpr (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
}

// sorted != 0: the 64 lanes of an instruction read from `sorted` consecutive 128-byte lines... (lane-adjacent elements share lines)
// cluster > 0: `cluster` adjacent lanes read consecutive elements of one aligned run; cluster < 0: -cluster adjacent lanes read
// the same 128-byte line at RANDOM offsets inside it (what a line-sorted gather stream looks like)
template <int MODE, int U>
__global__ __launch_bounds__(64) void gather8(const double *__restrict__ tab, unsigned mask, int steps, int cluster, double *__restrict__ out)
{
    if (cluster < 0) {
        const unsigned gid = blockIdx.x * 64 + threadIdx.x, cl = (unsigned) -cluster;
        double acc = 0;
        for (int s = 0; s < steps; ++s) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned h = mix((gid / cl) * 0x9e3779b9u + (unsigned) (s * U + u) * 0x85ebca6bu);
                const unsigned off = mix(gid * 0x2545f491u + (unsigned) (s * U + u)) & 15u;
                ld8<MODE>(v[u], tab + (((h & mask) & ~15u) | off));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
        }
        if (acc == 1.2345) out[gid] = acc;
        return;
    }
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    double acc = 0;
    for (int s = 0; s < steps; ++s) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned h = mix((gid / cluster) * 0x9e3779b9u + (unsigned) (s * U + u) * 0x85ebca6bu);
            const unsigned idx = ((h & mask) & ~(unsigned) (cluster - 1)) | (gid & (cluster - 1)); // `cluster` adjacent lanes share one aligned run
            ld8<MODE>(v[u], tab + idx);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
    }
    if (acc == 1.2345) out[gid] = acc;
}

template <int U>
__global__ __launch_bounds__(64) void gather4(const float *__restrict__ tab, unsigned mask, int steps, float *__restrict__ out)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    float acc = 0;
    for (int s = 0; s < steps; ++s) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned idx = mix(gid * 0x9e3779b9u + (unsigned) (s * U + u) * 0x85ebca6bu) & mask;
            asm volatile("global_load_dword %0, %1, off" : "=v"(v[u]) : "v"(tab + idx) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) { asm volatile("" : "+v"(v[u])); acc += v[u]; }
    }
    if (acc == 1.2345f) out[gid] = acc;
}

// the blocked executor's mix: per gather 14 bytes of nt stream (8 + 4 + 2) next to the scattered read
template <int U>
__global__ __launch_bounds__(64) void stream_gather(const double *__restrict__ tab, unsigned mask, int steps, const double *__restrict__ sv,
                                                    const int *__restrict__ sc, const unsigned short *__restrict__ sr, int do_gather, int do_stream,
                                                    double *__restrict__ out)
{
    const unsigned gid = blockIdx.x * 64 + threadIdx.x;
    const size_t base = (size_t) blockIdx.x * steps * U * 128 + threadIdx.x * 2; // two entries per lane and group, like the executor
    double acc = 0;
    for (int s = 0; s < steps; ++s) {
        double v[U];
        double a[U][2]; int c[U][2]; int r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (do_stream) {
                const size_t p = base + ((size_t) s * U + u) * 128;
                typedef double d2_t __attribute__((ext_vector_type(2)));
                typedef int i2_t __attribute__((ext_vector_type(2)));
                const d2_t q = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(sv + p));
                const i2_t cc = __builtin_nontemporal_load(reinterpret_cast<const i2_t *>(sc + p));
                r[u] = __builtin_nontemporal_load(reinterpret_cast<const int *>(sr + p));
                a[u][0] = q.x; a[u][1] = q.y; c[u][0] = cc.x; c[u][1] = cc.y;
            } else { a[u][0] = a[u][1] = 1; c[u][0] = c[u][1] = r[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned idx = mix(gid * 0x9e3779b9u + (unsigned) (s * U + u) * 0x85ebca6bu) & mask;
            if (do_gather) v[u] = tab[idx + (c[u][0] & 1)]; else v[u] = 1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] * (a[u][0] + a[u][1]) + (double) (c[u][1] + r[u]);
    }
    if (acc == 1.2345) out[gid] = acc;
}

// pure streams: what do 16-byte-per-lane reads, writes and a read+write copy sustain?
__global__ __launch_bounds__(256) void stream_rw(const double *__restrict__ src, double *__restrict__ dst, size_t n2, int mode, int nt)
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const size_t i = (size_t) blockIdx.x * 256 * 4 + threadIdx.x;
    d2_t acc = {0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t p = i + (size_t) u * 256;
        if (p >= n2) break;
        if (mode != 1) { // read
            const d2_t v = nt ? __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(src) + p) : reinterpret_cast<const d2_t *>(src)[p];
            acc += v;
            if (mode == 2) { if (nt) __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(dst) + p); else reinterpret_cast<d2_t *>(dst)[p] = v; }
        } else {
            const d2_t v = {(double) p, 1.0};
            if (nt) __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(dst) + p); else reinterpret_cast<d2_t *>(dst)[p] = v;
        }
    }
    if (mode == 0 && acc.x == 1.2345) dst[i] = acc.y;
}

// FETCH_SIZE calibration (tools/profile_configs.sh): a read of a KNOWN byte count with the 16-byte-per-lane nt loads the
// library's streams use.  MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half the bytes of such reads.
__global__ __launch_bounds__(256) void calib_read(const double *__restrict__ src, size_t n2, double *__restrict__ out)
{
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const size_t i = (size_t) blockIdx.x * 256 * 4 + threadIdx.x;
    d2_t acc = {0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t p = i + (size_t) u * 256;
        if (p < n2) acc += __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(src) + p);
    }
    if (acc.x == 1.2345) out[i] = acc.y;
}


// LDS float atomics from ONE wavefront per workgroup (the blocked executor's accumulation): U adds per step per lane into a
// region of `words` elements.  pattern 0: lane-consecutive (no bank conflict), 1: hashed (random banks), 2: ascending with a
// random stride (what a cell sorted by row looks like).
template <typename T, int U>
__global__ __launch_bounds__(64) void lds_atomics(int words, int steps, int pattern, T *__restrict__ out)
{
    extern __shared__ unsigned char raw[];
    T *ys = reinterpret_cast<T *>(raw);
    for (int i = threadIdx.x; i < words; i += 64) ys[i] = 0;
    __syncthreads();
    unsigned h = blockIdx.x * 64u + threadIdx.x;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned idx;
            if (pattern == 0) idx = (threadIdx.x + (s * U + u) * 64u);
            else if (pattern == 1) { h = h * 1664525u + 1013904223u; idx = h >> 8; }
            else if (pattern == 2) { h = h * 1664525u + 1013904223u; idx = (s * U + u) * 37u + threadIdx.x * 6u + ((h >> 20) & 3u); }
            else if (pattern == 3) idx = (s * U + u) * 37u;                       // all 64 lanes on one address
            else if (pattern == 4) idx = (s * U + u) * 37u + (threadIdx.x >> 3);  // runs of 8 lanes per address
            else idx = (threadIdx.x + (s * U + u) * 64u);                         // 5, 6: lane-consecutive, a quarter / one lane active
            if (pattern == 5 && (threadIdx.x & 3)) continue;
            if (pattern == 6 && threadIdx.x) continue;
            unsafeAtomicAdd(&ys[idx & (unsigned) (words - 1)], (T) 1);
        }
    }
    __syncthreads();
    T acc = 0;
    for (int i = threadIdx.x; i < words; i += 64) acc += ys[i];
    if (acc == (T) 1.2345) out[blockIdx.x] = acc;
}

template <typename F> static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    if (argc > 2 && strcmp(argv[1], "read") == 0) { // same-box ceiling for bench.py: a pure 16-byte-per-lane in-order read of `bytes` (one line of JSON)
        size_t bytes = (size_t) strtoull(argv[2], nullptr, 10);
        if (bytes < (1u << 20)) bytes = 1u << 20;
        const size_t n2 = bytes / 16;
        const int reps = argc > 3 ? atoi(argv[3]) : 10;
        double *a, *o;
        CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&o, 1 << 20));
        CK(hipMemset(a, 0, n2 * 16));
        const float t = time_ms([&] { calib_read<<<(int) ((n2 + 1023) / 1024), 256>>>(a, n2, o); }, reps);
        printf("{\"kernel\": \"calib_read\", \"bytes\": %zu, \"launches\": %d, \"ms_min\": %.5f, \"read_gbps\": %.1f}\n", n2 * 16, reps, t, n2 * 16 / t / 1e6);
        return 0;
    }
    if (argc > 1 && strcmp(argv[1], "calib") == 0) { // known-bytes read for the FETCH_SIZE calibration
        const size_t n2 = (size_t) 240 << 20;       // 16-byte elements: 3.75 GiB, far beyond the 256 MiB Infinity Cache
        double *a, *o;
        CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&o, 1 << 20));
        CK(hipMemset(a, 0, n2 * 16));
        for (int r = 0; r < 3; ++r) calib_read<<<(int) ((n2 + 1023) / 1024), 256>>>(a, n2, o);
        CK(hipDeviceSynchronize());
        printf("CALIB kernel=calib_read bytes=%zu launches=3\n", n2 * 16);
        return 0;
    }
    if (argc > 1 && strcmp(argv[1], "lds") == 0) { // LDS atomic-add rate, one wavefront per workgroup
        double *o; CK(hipMalloc(&o, 1 << 20));
        const int steps = 256; constexpr int U = 16;
        for (int grid : {1024, 4096})
            for (int pattern = 0; pattern < 7; ++pattern) {
                const double adds = (double) grid * 64 * steps * U;
                float t8 = time_ms([&] { lds_atomics<double, U><<<grid, 64, 4096 * 8>>>(4096, steps, pattern, o); }, 3);
                float t4 = time_ms([&] { lds_atomics<float, U><<<grid, 64, 4096 * 4>>>(4096, steps, pattern, (float *) o); }, 3);
                float t4b = time_ms([&] { lds_atomics<float, U><<<grid, 64, 8192 * 4>>>(8192, steps, pattern, (float *) o); }, 3);
                printf("lds_atomics grid %d pattern %d: f64 %.3g/s  f32 %.3g/s  f32(32 KiB) %.3g/s\n", grid, pattern, adds / t8 * 1e3, adds / t4 * 1e3, adds / t4b * 1e3);
            }
        return 0;
    }
    const size_t maxb = 256u << 20;
    double *tab, *out;
    CK(hipMalloc(&tab, maxb));
    CK(hipMemset(tab, 0, maxb));
    CK(hipMalloc(&out, 64u << 20));
    const int steps = 64;
    constexpr int U = 16;
    for (int wpc : {8}) {              // waves per CU
        const int grid = 256 * wpc;
        const double lanes = (double) grid * 64 * steps * U;
        for (size_t bytes : {(size_t) 1 << 20, (size_t) 2 << 20, (size_t) 64 << 20}) {
            const unsigned mask = (unsigned) (bytes / 8 - 1);
            printf("waves/CU %2d table %3zu MiB:", wpc, bytes >> 20);
            float t;
            t = time_ms([&] { gather8<0, U><<<grid, 64>>>(tab, mask, steps, 1, out); }, 3); printf("  plain %.3g/s", lanes / t * 1e3);
            t = time_ms([&] { gather8<1, U><<<grid, 64>>>(tab, mask, steps, 1, out); }, 3); printf("  nt %.3g/s", lanes / t * 1e3);
            t = time_ms([&] { gather8<2, U><<<grid, 64>>>(tab, mask, steps, 1, out); }, 3); printf("  sc0 %.3g/s", lanes / t * 1e3);
            t = time_ms([&] { gather8<3, U><<<grid, 64>>>(tab, mask, steps, 1, out); }, 3); printf("  sc1 %.3g/s", lanes / t * 1e3);
            t = time_ms([&] { gather8<4, U><<<grid, 64>>>(tab, mask, steps, 1, out); }, 3); printf("  sc0sc1 %.3g/s", lanes / t * 1e3);
            t = time_ms([&] { gather4<U><<<grid, 64>>>((const float *) tab, (unsigned) (bytes / 4 - 1), steps, (float *) out); }, 3); printf("  f32 %.3g/s", lanes / t * 1e3);
            for (int cl : {2, 4, 8, 16}) {
                t = time_ms([&] { gather8<0, U><<<grid, 64>>>(tab, mask, steps, cl, out); }, 3); printf("  cl%d %.3g/s", cl, lanes / t * 1e3);
            }
            for (int cl : {2, 4, 8}) {
                t = time_ms([&] { gather8<0, U><<<grid, 64>>>(tab, mask, steps, -cl, out); }, 3); printf("  sameline%d %.3g/s", cl, lanes / t * 1e3);
            }
            printf("\n");
        }
    }
    { // pure 16-byte-per-lane streams over 2.56 GB
        const size_t n2 = (size_t) 160 << 20; // 16-byte elements
        double *a, *b;
        CK(hipMalloc(&a, n2 * 16)); CK(hipMalloc(&b, n2 * 16));
        CK(hipMemset(a, 0, n2 * 16)); CK(hipMemset(b, 0, n2 * 16));
        const int grid = (int) ((n2 + 1023) / 1024);
        const char *names[3] = {"read", "write", "copy"};
        for (int mode = 0; mode < 3; ++mode)
            for (int nt = 0; nt < 2; ++nt) {
                float t = time_ms([&] { stream_rw<<<grid, 256>>>(a, b, n2, mode, nt); }, 5);
                printf("stream %-5s %s: %.3f ms for %.2f GB -> %.0f GB/s\n", names[mode], nt ? "nt   " : "plain", t, n2 * 16 * (mode == 2 ? 2 : 1) / 1e9,
                       n2 * 16 * (mode == 2 ? 2 : 1) / t / 1e6);
            }
        CK(hipFree(a)); CK(hipFree(b));
    }
    { // stream + gather: is the sum of the two what the chip does, or the max?
        const int wpc = 2, grid = 256 * wpc, st = 256;
        const size_t entries = (size_t) grid * st * 8 * 128;
        double *sv; int *sc; unsigned short *sr;
        CK(hipMalloc(&sv, entries * 8 + 4096)); CK(hipMalloc(&sc, entries * 4 + 4096)); CK(hipMalloc(&sr, entries * 2 + 4096));
        CK(hipMemset(sv, 0, entries * 8)); CK(hipMemset(sc, 0, entries * 4)); CK(hipMemset(sr, 0, entries * 2));
        const unsigned mask = (unsigned) ((2u << 20) / 8 - 1) & ~1u;
        const double gathers = (double) grid * 64 * st * 8;
        for (int mode = 0; mode < 3; ++mode) {
            const int g = mode != 1, s = mode != 0;
            float t = time_ms([&] { stream_gather<8><<<grid, 64>>>(tab, mask, st, sv, sc, sr, g, s, out); }, 3);
            printf("stream_gather %s%s: %.3f ms for %.3g gathers (%.3g/s) and %.2f GB of stream (%.0f GB/s)\n", g ? "gather " : "", s ? "stream" : "", t, gathers,
                   g ? gathers / t * 1e3 : 0.0, s ? entries * 14 / 1e9 : 0.0, s ? entries * 14 / t / 1e6 : 0.0);
        }
    }
    return 0;
}
