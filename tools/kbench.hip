// kbench.hip -- developer harness: A/B kernel variants on the config-2 shape in ONE process,
// interleaved rounds (cdna_hip_programming.md 5.4 rule 24).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -I spmv_amd/csrc -I include tools/kbench.hip -o gpurun_out/kbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "kernels/common.hpp"
#include "kernels/csr_rows.hpp"
#include "kernels/nnz_split.hpp"
#include "kernels/csr_vector4.hpp"

using namespace spmv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void gen_banded(int m, int n, int k, int *rowptr, int *col, double *val)
{
    const long long total = (long long) m * k;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x) {
        const long long r = i / k; const int j = (int) (i % k);
        long long c = r - k / 2 + j; c = (c % n + n) % n;
        col[i] = (int) c;
        unsigned h = (unsigned) (i * 2654435761u) ^ (unsigned) (i >> 13);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        val[i] = (double) (h & 0xFFFFF) / 524288.0 - 1.0;
    }
    for (long long r = (long long) blockIdx.x * blockDim.x + threadIdx.x; r <= m; r += (long long) gridDim.x * blockDim.x) rowptr[r] = (int) (r * k);
}
__global__ void gen_x(int n, double *x) {
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) {
        unsigned h = (unsigned) i * 747796405u + 2891336453u; h ^= h >> 16; x[i] = (double) (h & 0xFFFF) / 32768.0 - 1.0;
    }
}
// pure stream: read col+val with 16B loads, fold into one value per lane (ceiling for the matrix stream)
template <bool NT>
__global__ __launch_bounds__(256) void stream_read(long long nnz, const int *col, const double *val, double *out)
{
    double acc = 0; int iacc = 0;
    const long long nq = nnz / 4;
    for (long long q = (long long) blockIdx.x * 256 + threadIdx.x; q < nq; q += (long long) gridDim.x * 256) {
        int c[4]; double v[4];
        if (NT) { ld_stream4(col + q * 4, c); ld_stream4(val + q * 4, v); }
        else { const i32x4 cc = *(const i32x4 *) (col + q * 4); const f64x2 a = *(const f64x2 *) (val + q * 4), b = *((const f64x2 *) (val + q * 4) + 1);
               c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; }
        acc += v[0] + v[1] + v[2] + v[3]; iacc += c[0] ^ c[1] ^ c[2] ^ c[3];
    }
    if (acc == 1.2345 && iacc == 77) out[0] = acc;
}

// the tile kernel's stream after index compression: 16-bit column slots by 4 B/lane loads + doubles by
// 16 B/lane loads (10 B per non-zero) -- the FETCH_SIZE calibration for that mix of load widths
__global__ __launch_bounds__(256) void stream_read16(long long nnz, const unsigned short *col16, const double *val, double *out)
{
    double acc = 0; int iacc = 0;
    const long long q = (long long) blockIdx.x * 256 + threadIdx.x; // one pair of entries per lane and load
    if (q * 2 + 1 < nnz) {
        const int c = __builtin_nontemporal_load((const int *) (col16 + q * 2));
        const f64x2 a = __builtin_nontemporal_load((const f64x2 *) (val + q * 2));
        acc += a.x + a.y; iacc ^= c;
    }
    if (acc == 1.2345 && iacc == 77) out[0] = acc;
}

// stream + store experiments: MODE 0 none, 1 = 8 lanes x 8 B per wave at the end, 2 = same at the START,
// 3 = 64 lanes x 8 B from every 8th wave (same bytes, full lines), 4 = 8 lanes x 8 B nontemporal at end
template <int MODE>
__global__ __launch_bounds__(256) void stream_store(long long nnz, const int *col, const double *val, double *out)
{
    const long long q = (long long) blockIdx.x * 256 + threadIdx.x;
    const long long w = q >> 6; const int lane = threadIdx.x & 63;
    if (MODE == 2 && (lane & 7) == 0) out[w * 8 + (lane >> 3)] = (double) q;
    int c[4]; double v[4];
    ld_stream4(col + q * 4, c); ld_stream4(val + q * 4, v);
    double acc = v[0] + v[1] + v[2] + v[3] + (double) (c[0] ^ c[1] ^ c[2] ^ c[3]);
    asm volatile("" :: "v"(acc));
    if (MODE == 1 && (lane & 7) == 0) out[w * 8 + (lane >> 3)] = acc;
    if (MODE == 4 && (lane & 7) == 0) __builtin_nontemporal_store(acc, out + w * 8 + (lane >> 3));
    if (MODE == 3 && (w & 7) == 0) out[w * 8 + lane] = acc;
    if (MODE == 5 && (w & 63) == 0) out[w * 8 + lane] = acc;             // 1/8 of the bytes
    if (MODE == 6 && (lane & 7) == 0) __hip_atomic_store(out + w * 8 + (lane >> 3), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 7 && (lane & 7) == 0) __hip_atomic_store(out + w * 8 + (lane >> 3), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 8 && (w & 7) == 0) out[(w >> 3) * 64 + lane] = acc;      // same bytes, dense target (y region 1/8 size... no: same size)
    if (MODE == 9 && (w & 255) == 0) { for (int i = 0; i < 32; ++i) out[w * 8 + i * 64 + lane] = acc; }   // 16 KB bursts, same bytes
    if (MODE == 10 && (w & 2047) == 0) { for (int i = 0; i < 256; ++i) out[w * 8 + i * 64 + lane] = acc; } // 128 KB bursts
    if (MODE == 11 && (lane & 7) == 0) out[(w * 8 + (lane >> 3)) & 131071] = acc;          // same stores into a 1 MB window (L2-resident)
    if (MODE == 12 && (lane & 7) == 0) out[(w * 8 + (lane >> 3)) & 4194303] = acc;         // 32 MB window (MALL-resident)
    if (MODE == 13 && (lane & 7) == 0) atomicAdd(out + w * 8 + (lane >> 3), acc);          // memory-side atomics instead of stores
    if (MODE == 0 && acc == 1.2345) out[0] = acc;
}
// write-only and copy references
__global__ __launch_bounds__(256) void fill_k(long long n4, f64x2 *out) { const long long q = (long long) blockIdx.x * 256 + threadIdx.x; if (q < n4) { f64x2 v = {1.0, 2.0}; out[q] = v; } }
__global__ __launch_bounds__(256) void copy_k(long long n4, const f64x2 *in, f64x2 *out) { const long long q = (long long) blockIdx.x * 256 + threadIdx.x; if (q < n4) out[q] = __builtin_nontemporal_load(in + q); }

struct Variant { std::string name; std::function<void()> run; };

int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 10000000, k = 32, n = m;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const long long nnz = (long long) m * k;
    int *rowptr, *col; double *val, *x, *y, *yref;
    CK(hipMalloc(&rowptr, sizeof(int) * (m + 1))); CK(hipMalloc(&col, sizeof(int) * nnz + 8192)); CK(hipMalloc(&val, sizeof(double) * nnz + 8192));
    CK(hipMalloc(&x, sizeof(double) * n)); CK(hipMalloc(&y, sizeof(double) * m)); CK(hipMalloc(&yref, sizeof(double) * m));
    gen_banded<<<4096, 256>>>(m, n, k, rowptr, col, val); gen_x<<<2048, 256>>>(n, x); CK(hipDeviceSynchronize());
    const int cus = 256;
    // nnz-split inspector
    const int ntiles = (int) ((nnz + kSplitTile - 1) / kSplitTile);
    int *tile_first, *flag; double *carry;
    CK(hipMalloc(&tile_first, sizeof(int) * (ntiles + 1))); CK(hipMalloc(&carry, sizeof(double) * ntiles)); CK(hipMalloc(&flag, 4));
    nnz_tile_first_kernel<<<(ntiles + 256) / 256, 256>>>(m, ntiles, kSplitTile, rowptr, tile_first, flag); CK(hipDeviceSynchronize());

    csr_scalar_kernel<double><<<cus * 8, 256>>>(m, rowptr, col, val, x, yref); CK(hipDeviceSynchronize());

    std::vector<Variant> vs;
    auto grid_rows = [&](int rows_per_block, int cap) { long long g = ((long long) m + rows_per_block - 1) / rows_per_block; return (int) std::min<long long>(g, cap); };
    vs.push_back({"stream_read plain", [&] { stream_read<false><<<cus * 8, 256>>>(nnz, col, val, y); }});
    vs.push_back({"stream_read nt", [&] { stream_read<true><<<cus * 8, 256>>>(nnz, col, val, y); }});
    vs.push_back({"csr_vector L16 (lib)", [&] { csr_vector_kernel<double, 16><<<grid_rows(16, cus * 32), 256>>>(m, rowptr, col, val, x, y); }});
    vs.push_back({"csr_vector L8  (lib)", [&] { csr_vector_kernel<double, 8><<<grid_rows(32, cus * 32), 256>>>(m, rowptr, col, val, x, y); }});
    vs.push_back({"nnz_split (lib)", [&] { nnz_split_kernel<double><<<cus * 8, 256>>>(m, (int) nnz, ntiles, rowptr, col, val, x, y, tile_first, carry); }});
#define V4(L, RED, U, CAP) vs.push_back({"vec4 L" #L " red" #RED " U" #U " cap" #CAP, [&] { csr_vector4_kernel<double, L, RED, U><<<grid_rows(256 / L * U, cus * CAP), 256>>>(m, rowptr, col, val, x, y); }})
    V4(8, 0, 1, 8); V4(8, 1, 1, 8); V4(8, 1, 2, 8); V4(8, 1, 1, 16); V4(8, 1, 1, 4096);
    V4(8, 1, 2, 4096);
#define VP(L, NB, NT) vs.push_back({"pipe L" #L " NB" #NB " nt" #NT, [&] { constexpr int rpb = 256 / L * NB; csr_vector_pipe_kernel<double, L, NB, NT><<<(int) (((long long) m + rpb - 1) / rpb), 256>>>(m, 1 << 30, rowptr, col, val, x, y); }})
#define VPA(L, NB, ABL) vs.push_back({"pipe L" #L " NB" #NB " ABL" #ABL, [&] { constexpr int rpb = 256 / L * NB; csr_vector_pipe_kernel<double, L, NB, false, ABL><<<(int) (((long long) m + rpb - 1) / rpb), 256>>>(m, 1 << 30, rowptr, col, val, x, y); }})
    VPA(8, 4, 1); VPA(8, 4, 2); VPA(8, 4, 8); VPA(8, 4, 4); VPA(8, 4, 16); VPA(8, 4, 31); VPA(8, 4, 9); VPA(8, 4, 17);
    VP(8, 1, false); VP(8, 2, false); VP(8, 4, false); VP(8, 8, false); VP(8, 16, false); VP(8, 4, true); VP(8, 1, true); VP(16, 4, false); VP(4, 4, false);
#define V4A(L, U, CAP, ABL) vs.push_back({"vec4 L" #L " U" #U " cap" #CAP " ABL" #ABL, [&] { csr_vector4_kernel<double, L, 1, U, ABL><<<grid_rows(256 / L * U, cus * CAP), 256>>>(m, rowptr, col, val, x, y); }})
    V4A(8, 1, 4096, 1); V4A(8, 1, 4096, 9); V4A(8, 1, 4096, 17); V4A(8, 1, 4096, 25); V4A(8, 1, 4096, 29); V4A(8, 1, 4096, 8); V4A(8, 1, 4096, 16); V4A(8, 2, 4096, 25);
    vs.push_back({"stream_read nt x2blocks", [&] { stream_read<true><<<cus * 16, 256>>>(nnz, col, val, y); }});
    vs.push_back({"stream_read16 (10 B/nnz)", [&] { stream_read16<<<(int) ((nnz / 2 + 255) / 256), 256>>>(nnz, (const unsigned short *) col, val, y); }});
    vs.push_back({"stream_read nt nonpersist", [&] { stream_read<true><<<(int) (nnz / 4 / 256), 256>>>(nnz, col, val, y); }});
#define SS(M) vs.push_back({"stream_store mode" #M, [&] { stream_store<M><<<(int) (nnz / 4 / 256), 256>>>(nnz, col, val, y); }})
    SS(0); SS(1);
    double *y_unc = nullptr, *y_fine = nullptr;
    if (hipExtMallocWithFlags((void **) &y_unc, sizeof(double) * m, hipDeviceMallocUncached) != hipSuccess) { y_unc = nullptr; (void) hipGetLastError(); }
    if (hipExtMallocWithFlags((void **) &y_fine, sizeof(double) * m, hipDeviceMallocFinegrained) != hipSuccess) { y_fine = nullptr; (void) hipGetLastError(); }
    if (y_unc) vs.push_back({"stream_store mode1 y=uncached", [&] { stream_store<1><<<(int) (nnz / 4 / 256), 256>>>(nnz, col, val, y_unc); }});
    if (y_fine) vs.push_back({"stream_store mode1 y=finegrained", [&] { stream_store<1><<<(int) (nnz / 4 / 256), 256>>>(nnz, col, val, y_fine); }});
    if (y_unc) vs.push_back({"stream_store mode3 y=uncached", [&] { stream_store<3><<<(int) (nnz / 4 / 256), 256>>>(nnz, col, val, y_unc); }});
    if (0) vs.push_back({"fill 2.56GB (as stream bytes)", [&] { fill_k<<<(int) (nnz / 2 / 256), 256>>>(nnz / 2, (f64x2 *) val); }});
    if (0) vs.push_back({"copy 1.28GB->1.28GB", [&] { copy_k<<<(int) (nnz / 4 / 256), 256>>>(nnz / 4, (const f64x2 *) val, (f64x2 *) val + nnz / 4); }});
    vs.push_back({"nnz_split nonpersist", [&] { nnz_split_kernel<double><<<(ntiles + 3) / 4, 256>>>(m, (int) nnz, ntiles, rowptr, col, val, x, y, tile_first, carry); }});

    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t i = 0; i < vs.size(); ++i) { // correctness first
        CK(hipMemset(y, 0xFF, sizeof(double) * m));
        vs[i].run(); CK(hipDeviceSynchronize());
        if (vs[i].name.rfind("stream", 0) == 0) continue;
        std::vector<double> a(4096), b(4096);
        double maxerr = 0;
        for (long long off : {0ll, (long long) m / 2, (long long) m - 4096}) {
            CK(hipMemcpy(a.data(), y + off, 4096 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref + off, 4096 * 8, hipMemcpyDeviceToHost));
            for (int j = 0; j < 4096; ++j) { double e = std::fabs(a[j] - b[j]); if (!(e <= 1e300)) e = 1e300; maxerr = std::max(maxerr, e); }
        }
        if (vs[i].name.find("ABL") != std::string::npos) continue;
        if (maxerr > 1e-9) printf("!! %s WRONG maxerr %g\n", vs[i].name.c_str(), maxerr);
    }
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            vs[i].run(); // warm
            CK(hipEventRecord(e0)); for (int q = 0; q < 5; ++q) vs[i].run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[i].push_back(ms / 5);
        }
    const double alg = 4.0 * (m + 1) + nnz * 12.0 + 8.0 * n + 8.0 * m;
    printf("%-34s %9s %9s %9s %7s\n", "variant", "min_ms", "med_ms", "GB/s(min)", "frac");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double mn = t[i][0], med = t[i][t[i].size() / 2];
        const double bytes = vs[i].name.rfind("stream", 0) == 0 ? nnz * 12.0 : alg;
        printf("%-34s %9.4f %9.4f %9.1f %7.3f\n", vs[i].name.c_str(), mn, med, bytes / mn / 1e6, bytes / mn / 1e6 / 8000.0);
    }
    return 0;
}
