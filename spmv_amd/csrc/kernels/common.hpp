// common.hpp -- device helpers shared by every gfx950 kernel of this library.
// Written for CDNA4 only: 64-lane wavefronts, no other target is considered.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spmv {

constexpr int kWave = 64;   // wavefront width (cdna_hip_programming.md 1: hard-code 64)
constexpr int kBlock = 256; // 4 waves per workgroup: one per SIMD

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Matrix streams (RowPtr/ColIdx/Val) are read exactly once per SpMV: load them non-temporal so
// they do not push x out of L2 / Infinity Cache (MI355X_MICROARCH.md "nt-weights": once-read
// streams; x is the re-read operand).
template <typename T>
__device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }

// 16-byte streaming loads of 4 consecutive elements starting at a 16-byte aligned element index.
__device__ __forceinline__ void ld_stream4(const int *p, int (&o)[4])
{
    const i32x4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
// 4 consecutive 16-bit LDS slots (one 8-byte load), left PACKED in o[0], o[1]: the compressed
// column stream of staged tiles (slot k = half k&1 of word k>>1, see lds_slot)
__device__ __forceinline__ void ld_stream4(const unsigned short *p, int (&o)[4])
{
    const i32x2 v = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(p));
    o[0] = v.x; o[1] = v.y;
}
template <int K>
__device__ __forceinline__ unsigned lds_slot(const int (&o)[4])
{
    return (K & 1) ? (unsigned) o[K >> 1] >> 16 : (unsigned) o[K >> 1] & 0xffffu;
}
__device__ __forceinline__ void ld_stream4(const float *p, float (&o)[4])
{
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld_stream4(const double *p, double (&o)[4])
{
    const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
    const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p) + 1);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}

__device__ __forceinline__ float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmadd(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- DPP butterfly pieces (all lanes of the group end up with the group total) ----------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int) (b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int) (b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}

// Sum over groups of W consecutive lanes, W in {1,2,4,8,16,32,64}; result valid in every lane of the
// group for W <= 16, and at least in the group's first lane for W = 32 / 64.
template <int W, typename T>
__device__ __forceinline__ T group_sum_dpp(T v)
{
    if (W >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if (W >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if (W >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of each 8
    if (W >= 16) v += dpp_mov<0x140>(v); // row_mirror: the other half of each 16
    if (W >= 32) v += __shfl_xor(v, 16, kWave);
    if (W >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// Lane 0's value in every lane (v_readlane: a scalar register, no LDS permute).
__device__ __forceinline__ float lane0_value(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)); }
__device__ __forceinline__ double lane0_value(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int) (b & 0xFFFFFFFFll), 0), hi = __builtin_amdgcn_readlane((int) (b >> 32), 0);
    return __builtin_bit_cast(double, ((long long) hi << 32) | (unsigned) lo);
}

// Sum over groups of W consecutive lanes (W = power of two <= 64); every lane of the group gets
// the total (xor butterfly), so any lane may store it.
template <int W, typename T>
__device__ __forceinline__ T group_sum(T v)
{
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Same with a run-time width (row-block kernel).
template <typename T>
__device__ __forceinline__ T group_sum_rt(T v, int w)
{
    for (int o = w >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// LDS hand-off between lanes of ONE wave (no s_barrier: the waves of a workgroup run different
// trip counts).  DS operations of a wave execute in issue order; the fences keep the compiler
// from moving LDS accesses across the point and make it wait for outstanding DS results.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// First index in [0, n) with a[idx] >= key (n if none).  std::lower_bound; the reference's
// lower_bound shim is csr5_spmv.cpp:54-56.
__device__ __forceinline__ int lower_bound_dev(const int *a, int n, long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((long long) a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Workgroups are dealt to the eight XCDs round-robin (workgroup b runs on XCD b % 8, each with its own 4 MiB L2).
// Logical block of workgroup b such that every XCD owns a CONTIGUOUS eighth of the nb blocks: neighbouring tiles
// (neighbouring rows, hence overlapping parts of x when the columns have locality) then share one L2 instead of
// pulling every x line across the fabric into all eight.  Bijective for any nb.  Used by the tile kernels that gather x
// through L2 (nat_kernel / csr5_kernel: webbase-style 1e6 rows with web-like columns 25.4 -> 21.9 us, 4e6 rows 80 -> 75 us,
// R-MAT / uniform columns unchanged); NOT by the kernels that stage x windows in LDS and stream at HBM rate -- those lose 2-6 %
// when their blocks are not dispatched in address order (config 2: 0.516 -> 0.534 ms, config 4 CSR-vector 0.745 -> 0.792).
__device__ __forceinline__ int xcd_block(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7, k = b & 7, i = b >> 3;
    return k * q + (k < r ? k : r) + i;
}

// First index in [0, n) with a[idx] > key (n if none).  The reference's
// binary_search_right_boundary_kernel (parallel_balanced_spmv.c:17-37) computes the same thing.
__device__ __forceinline__ int upper_bound_dev(const int *a, int n, long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if ((long long) a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

} // namespace spmv
