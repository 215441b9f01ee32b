// csr5.hpp -- CSR5 with omega = 64 (one wavefront per tile).
//
// GPU schedule of Method_CSR5SPMV.  Reference: the vendored CSR5 (csr5_spmv.cpp:16-52 glue;
// csr5_avx2/anonymouslib_avx2.h:112-242 asCSR5; avx2/format_avx2.h:7-425 tile pointer, tile
// descriptor, empty-row offsets, in-place transpose; avx2/csr5_spmv_avx2.h:7-410 tile kernel,
// calibrator, tail) with omega = 4 AVX2 lanes and sigma = 16.  The idea is kept -- equal-nnz
// tiles of omega x sigma elements, lane x owns sigma consecutive nnz, row starts kept as bit flags
// in a per-lane descriptor, tile storage transposed so that element i of all lanes is one
// contiguous line -- and re-designed for a 64-lane wavefront (SURVEY A.4 "GPU re-design"):
//
//   tile          64 lanes x SIGMA elements (SIGMA in {4, 8, 16}); p = ceil(nnz / (64 SIGMA))
//   tile_ptr[t]   int32, row that holds the tile's first nnz (format_avx2.h:14-24)
//   desc[t][x]    uint32 per lane: bits 0..SIGMA-1 = "a row starts at my element i",
//                 bits 16..25 = y_offset = number of row starts in lanes < x  (the reference packs
//                 y_offset | scansum_offset | flags into one word too, format_avx2.h:94-110,
//                 196-213; scansum_offset is not stored here: the cross-lane combine uses a
//                 ballot mask of "lane has a start" instead)
//   col/val       COPIES in tile-transposed order, element i of lane x at t*64*SIGMA + i*64 + x
//                 (the reference transposes the CALLER's arrays in place, format_avx2.h:347-400;
//                 here the caller's arrays are never touched).  The last tile is padded with
//                 col = -1 / val = 0, so there is no separate "tail" kernel
//                 (csr5_spmv_avx2.h:337-366 in the reference).
//   run_len[t]    int32: if the row that is open at the start of tile t began in tile t-1,
//                 the number of tiles it continues through; else 0.  Drives the deterministic
//                 carry fix-up that replaces the reference's per-thread calibrator
//                 (csr5_spmv_avx2.h:320-335).
//   empty rows    the reference keeps per-tile offset tables for tiles whose row span contains
//                 empty rows (format_avx2.h:256-326) and never writes y for empty rows
//                 (SURVEY 4.3).  Here, if the matrix has empty rows, CSR5 is built over the
//                 COMPACTED row space (row_map[k] = k-th non-empty row) and the tile kernel's
//                 workgroups zero y for the listed empty rows on the side (empty_list); without
//                 empty rows neither array is allocated and no indirection runs.
// fp32 is native (the reference silently falls back to SELL for fp32, common.c:174-181).
// Extra HBM traffic on top of B_alg per tile of 64*SIGMA nnz: 256 B desc + 4 B tile_ptr + s carry.
#pragma once
#include <climits>
#include "common.hpp"
#include "xwindows.hpp"

namespace spmv {

constexpr unsigned kCsr5FlagMask = 0xFFFFu;
constexpr int kCsr5YoffShift = 16;

// ---------------------------------------------------------------------------- inspector kernels
// flags[r] = row r is non-empty
static __global__ __launch_bounds__(kBlock) void csr5_nonempty_kernel(int m, const int *__restrict__ rowptr, int *__restrict__ flags)
{
    const long long stride = (long long) gridDim.x * kBlock;
    for (long long r = (long long) blockIdx.x * kBlock + threadIdx.x; r < m; r += stride)
        flags[r] = rowptr[r + 1] > rowptr[r];
}

// Three-pass exclusive scan of int32 (block sums, scan of the sums by one workgroup, apply).
constexpr int kScanTile = kBlock * 4;
static __global__ __launch_bounds__(kBlock) void scan_block_sums_kernel(long long n, const int *__restrict__ in, int *__restrict__ sums)
{
    __shared__ int wsum[kBlock / kWave];
    const long long base = (long long) blockIdx.x * kScanTile;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = base + threadIdx.x + k * kBlock;
        s += i < n ? in[i] : 0;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
static __global__ __launch_bounds__(kBlock) void scan_sums_inplace_kernel(int nb, int *__restrict__ sums, int *__restrict__ total)
{
    __shared__ int wave_tot[kBlock / kWave];
    __shared__ int carry_s;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += kBlock) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (i < nb) sums[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
// Compaction: for every non-empty row r (flag 1) at compacted position k: rp2[k] = rowptr[r],
// row_map[k] = r.  One thread handles 4 strided-by-block elements in order, so positions are
// block_offset + (prefix inside the block), computed with a workgroup scan per 256-element slab.
static __global__ __launch_bounds__(kBlock) void csr5_compact_kernel(long long m, const int *__restrict__ flags,
                                                              const int *__restrict__ block_off,
                                                              const int *__restrict__ rowptr,
                                                              int *__restrict__ rp2, int *__restrict__ row_map,
                                                              int *__restrict__ empty_list = nullptr /* nullable: the other rows, in order */)
{
    __shared__ int wave_tot[kBlock / kWave];
    __shared__ int slab_base;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (threadIdx.x == 0) slab_base = block_off[blockIdx.x];
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const long long r = (long long) blockIdx.x * kScanTile + k * kBlock + threadIdx.x;
        const int f = r < m ? flags[r] : 0;
        int inc = f;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int off = slab_base;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (f) {
            const int pos = off + inc - 1;
            rp2[pos] = rowptr[r];
            row_map[pos] = (int) r;
        } else if (empty_list && r < m) {
            empty_list[r - (off + inc)] = (int) r; // off + inc flagged rows lie in front of r
        }
        __syncthreads();
        if (threadIdx.x == kBlock - 1) slab_base = off + inc;
        __syncthreads();
    }
}

// tile_ptr[t] = last row r (of the m2-row space) with rp[r] <= min(t*T, nnz), t = 0..p
static __global__ __launch_bounds__(kBlock) void csr5_tile_ptr_kernel(int m2, int nnz, int p, int tile_nnz,
                                                               const int *__restrict__ rp, int *__restrict__ tile_ptr)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t > p) return;
    long long key = (long long) t * tile_nnz;
    if (key > nnz) key = nnz;
    int r = upper_bound_dev(rp, m2 + 1, key) - 1;
    if (r > m2 - 1) r = m2 - 1; // key == nnz: clamp to the last row
    tile_ptr[t] = r;
}

// out[0] = max over the tiles of tile_ptr[t + 1] - tile_ptr[t] (out zeroed by the caller): sizes the waves' row-map buffers.
static __global__ __launch_bounds__(kBlock) void csr5_tile_rows_max_kernel(int p, const int *__restrict__ tile_ptr, int *__restrict__ out)
{
    const int t = blockIdx.x * kBlock + threadIdx.x;
    int v = t < p ? tile_ptr[t + 1] - tile_ptr[t] : 0;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_xor(v, d, kWave);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && v > 0) atomicMax(out, v);
}

// One wavefront per tile: bit flags of the row starts inside each lane's SIGMA elements, y_offset =
// exclusive count of starts over the lanes.  Also the run length of the row open at the tile start.
template <int SIGMA>
__global__ __launch_bounds__(kBlock) void csr5_desc_kernel(int m2, int nnz, int p, const int *__restrict__ rp,
                                                           const int *__restrict__ tile_ptr,
                                                           unsigned *__restrict__ desc, int *__restrict__ run_len,
                                                           int *__restrict__ any_head)
{
    constexpr int T = kWave * SIGMA;
    __shared__ unsigned fl[kBlock / kWave][kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t = blockIdx.x * (kBlock / kWave) + wave;
    if (t >= p) return;
    const long long base = (long long) t * T;
    fl[wave][lane] = 0;
    wave_lds_sync();
    const int r_lo = tile_ptr[t], r_hi = tile_ptr[t + 1];
    for (int r = r_lo + lane; r <= r_hi && r < m2; r += kWave) {
        const long long ptr = rp[r];
        if (ptr >= base && ptr < base + T && ptr < nnz) {
            const int o = (int) (ptr - base);
            atomicOr(&fl[wave][o / SIGMA], 1u << (o % SIGMA));
        }
    }
    wave_lds_sync();
    const unsigned f = fl[wave][lane];
    const int cnt = __popc(f);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += o;
    }
    desc[(long long) t * kWave + lane] = f | ((unsigned) (inc - cnt) << kCsr5YoffShift);
    if (lane == 0) {
        int rl = 0;
        if (t > 0 && !(f & 1u)) { // the tile opens inside row r_lo
            *any_head = 1;
            if ((long long) rp[r_lo] >= base - T) { // ... which began in the previous tile: run start
                const long long hend = rp[r_lo + 1];
                rl = (int) ((hend - 1) / T) - t + 1;
            }
        }
        run_len[t] = rl;
    }
}

// Where entry i of lane x sits inside the tile's transposed VALUE copy: 16 bytes per lane and load instruction (E = 4 floats / 2 doubles:
// entries i = gE .. gE + E - 1 of a lane side by side), 1 KiB per wave-instruction instead of the 256 / 512 B of one value per lane
// (config 4, fp32: CSR5 0.571 -> 0.52 ms with the two-deep loop below).  The column copies: tcol [i * 64 + x], tcol16 packed alike (pack16).
template <typename T, int SIGMA>
__device__ __forceinline__ int csr5_val_pos(int i, int lane)
{
    constexpr int E = 16 / (int) sizeof(T);
    static_assert(SIGMA % E == 0, "sigma is a multiple of the values per 16-byte load");
    return (i / E) * (E * kWave) + lane * E + (i % E);
}

// v[i] = the lane's SIGMA values of tile t (SIGMA / E loads of 16 bytes)
template <typename T, int SIGMA>
__device__ __forceinline__ void csr5_load_vals(const T *__restrict__ tval, int t, int lane, T (&v)[SIGMA])
{
    constexpr int E = 16 / (int) sizeof(T), TN = kWave * SIGMA;
    const T *src = tval + (long long) t * TN + lane * E;
#pragma unroll
    for (int g = 0; g < SIGMA / E; ++g) {
        if constexpr (E == 4) {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + g * (E * kWave)));
            v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
        } else {
            const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(src + g * (E * kWave)));
            v[2 * g] = q.x; v[2 * g + 1] = q.y;
        }
    }
}

// Tile-transposed copies: tcol[t*T + i*64 + x] = colidx[t*T + x*SIGMA + i], tval[t*T + csr5_val_pos(i, x)] = val[...]; padding col = -1, val = 0.
template <typename T, int SIGMA>
__global__ __launch_bounds__(kBlock) void csr5_transpose_kernel(int nnz, int p, const int *__restrict__ colidx,
                                                                const T *__restrict__ val,
                                                                int *__restrict__ tcol, T *__restrict__ tval)
{
    constexpr int TN = kWave * SIGMA;
    const int lane = threadIdx.x & (kWave - 1);
    const int t = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= p) return;
    const long long base = (long long) t * TN;
#pragma unroll
    for (int i = 0; i < SIGMA; ++i) {
        const long long src = base + (long long) lane * SIGMA + i;
        const bool in = src < nnz;
        if (tcol) tcol[base + i * kWave + lane] = in ? colidx[src] : -1; // NULL: values only (spmv_hip_update_values)
        tval[base + csr5_val_pos<T, SIGMA>(i, lane)] = in ? val[src] : T(0);
    }
}

// RUN groups (as csr_vector_tile.hpp's RUN tiles, for the ENTRY-granular tiles of CSR5; the natural-layout nnz-split kernel measured no gain -- it is
// bound by its LDS hand-over, not by the 2 B/nnz it would save -- and keeps its slot stream): when every lane of every tile of a staged
// group holds runs of consecutive slots with AT MOST ONE row start past its first entry -- rows of at least SIGMA entries that are runs of
// consecutive columns: banded matrices --, a word per lane and tile (lane_run: slot of the lane's first entry | slot at that row start << 16)
// replaces the SIGMA 16-bit slots: 4 instead of 2 * SIGMA bytes per lane.  The row start's position is in the descriptor's flag bits.
template <int SIGMA>
__device__ __forceinline__ void csr5_run_slots(unsigned run, unsigned d, int (&c)[SIGMA])
{
    const unsigned s_first = run & 0xffffu, s_after = run >> 16;
    const unsigned fl = (d & kCsr5FlagMask) >> 1;             // row starts at entries 1 .. SIGMA - 1
    const int pos = fl ? __ffs((int) fl) : SIGMA;             // the entry the (one) row start sits at
#pragma unroll
    for (int i = 0; i < SIGMA; ++i) c[i] = (int) (i < pos ? s_first + (unsigned) i : s_after + (unsigned) (i - pos));
}

// slot of entry i of lane x of tile t in the 16-bit stream: transposed + packed (CSR5, pack16 = SIGMA) or the matrix's own order (nnz-split)
template <int SIGMA>
__device__ __forceinline__ long long csr5_slot_pos(int t, int lane, int i, bool natural)
{
    constexpr long long TN = (long long) kWave * SIGMA;
    return natural ? t * TN + (long long) lane * SIGMA + i : t * TN + (i / 4) * (4 * kWave) + lane * 4 + (i % 4);
}

// Inspector: which staged groups are RUN groups?  One workgroup per group, a wave per tile.  counters[0] += groups, [1] += their tiles.
template <int SIGMA>
__global__ __launch_bounds__(kBlock) void csr5_runs_kernel(int group_tiles, int p, long long nnz, int natural, const unsigned *__restrict__ desc,
                                                           const unsigned short *__restrict__ col16, TileWindows *__restrict__ wins,
                                                           unsigned *__restrict__ lane_run, unsigned long long *__restrict__ counters)
{
    constexpr long long TN = (long long) kWave * SIGMA;
    const int g = blockIdx.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t0 = g * group_tiles;
    int ok = wins[g].nwin > 0;
    int tiles = 0;
    if (ok)
        for (int k = wave; k < group_tiles; k += kBlock / kWave) {
            const int t = t0 + k;
            if (t >= p) break;
            if ((long long) (t + 1) * TN > nnz) { ok = 0; break; } // the matrix's last, partly filled tile: padding slots
            const unsigned fl = (desc[(long long) t * kWave + lane] & kCsr5FlagMask) >> 1;
            ok &= __popc(fl) <= 1;
            const int pos = fl ? __ffs((int) fl) : SIGMA;
            unsigned prev = col16[csr5_slot_pos<SIGMA>(t, lane, 0, natural != 0)];
#pragma unroll
            for (int i = 1; i < SIGMA; ++i) {
                const unsigned cur = col16[csr5_slot_pos<SIGMA>(t, lane, i, natural != 0)];
                if (i != pos) ok &= cur == prev + 1u;
                prev = cur;
            }
            if (lane == 0) ++tiles;
        }
    ok = __syncthreads_and(ok);
    if (!ok) return;
    for (int k = wave; k < group_tiles; k += kBlock / kWave) {
        const int t = t0 + k;
        if (t >= p) break;
        const unsigned fl = (desc[(long long) t * kWave + lane] & kCsr5FlagMask) >> 1;
        const int pos = fl ? __ffs((int) fl) : SIGMA;
        const unsigned s_first = col16[csr5_slot_pos<SIGMA>(t, lane, 0, natural != 0)];
        const unsigned s_after = pos < SIGMA ? col16[csr5_slot_pos<SIGMA>(t, lane, pos, natural != 0)] : 0u;
        lane_run[(long long) t * kWave + lane] = s_first | (s_after << 16);
    }
    if (lane == 0 && tiles) atomicAdd(counters + 1, (unsigned long long) tiles);
    if (threadIdx.x == 0) { wins[g].runs = 1; atomicAdd(counters, 1ull); }
}

// ---------------------------------------------------------------------------- executor
// The arithmetic of one tile once the lane holds its SIGMA entries (c: LDS slots when STAGED, else
// global columns with -1 = padding; v: values): gathers, per-lane segmented sums cut at the row-start
// flags, cross-lane combine, y stores and the tile's carry.  Shared by the transposed (CSR5) and
// the natural-layout (nnz-split) tile loaders.
// Row map of a tile in LDS (matrices with empty rows: the tiles run over the compacted row space and y[row_map[r]] is
// where a row's sum goes).  A tile writes rows r0 .. r1 = tile_ptr[t] .. tile_ptr[t + 1], at most 64 * SIGMA + 1 of
// them; all of their row_map entries are fetched up front by unconditional, coalesced loads and parked in `rm`
// ((SIGMA + 1) * 64 ints, private to the wavefront).  Before: every row start did "load row_map[seg_row] -> wait ->
// store y" inside its branch, and since stores count in vmcnt on gfx9 each wait also drained the previous store --
// up to SIGMA + 1 memory latencies per tile, the whole run time of the 1e6-row power-law stand-in (21 us).
// The calling wave's row-map buffer inside the workgroup's dynamic LDS.  rm_stride (ints per wave) is kWave when no tile of
// the plan holds kWave row starts or more -- the long-row sub-matrix of SELL / CSR-vector -- and (SIGMA + 1) * kWave otherwise
// (launch_csr5_form): 1 KiB instead of 17 KiB per workgroup next to the x windows.
__device__ __forceinline__ int *wave_row_map(unsigned char *dyn_lds, int rm_off, int rm_stride)
{
    return reinterpret_cast<int *>(dyn_lds + rm_off) + (threadIdx.x / kWave) * rm_stride;
}

template <int SIGMA>
__device__ __forceinline__ void csr5_stage_row_map(int lane, int r0, int r1, const int *__restrict__ row_map, int *__restrict__ rm)
{
    // Tiles of long rows hold fewer than 64 row starts (the long-row sub-matrix of SELL: at most 32 per tile of 1024):
    // one load covers them; the other SIGMA sit behind ONE wave-uniform branch (config 4 through SELL: 0.63 ms, 0.72-0.75
    // with all SIGMA + 1 loads issued for every tile).
    const int span = __builtin_amdgcn_readfirstlane(r1 - r0);
    const int first = row_map[r0 + (lane < span ? lane : span)];
    if (span >= kWave) {
        int tmp[SIGMA];
#pragma unroll
        for (int j = 1; j <= SIGMA; ++j) {
            const int k = j * kWave + lane;
            tmp[j - 1] = row_map[r0 + (k < span ? k : span)];
        }
#pragma unroll
        for (int j = 1; j <= SIGMA; ++j) rm[j * kWave + lane] = tmp[j - 1];
    }
    rm[lane] = first;
    wave_lds_sync();
}

// The forward part of a tile (nat_kernel<.., FWD>, "forward completion"): the row that is open at the tile's END is finished by the tile it
// STARTS in -- count entries of the next tile (count < the tile size: longer rows are on the long-row list, count = -1, and a workgroup of their own
// writes them).  The first 64 of them arrive in v / c, one per lane (0 / column 0 beyond count), loaded and gathered with the tile's own entries;
// the rest are read in place from base (rare with short rows).  Variants measured on the webbase-style stand-in and not kept: the first FOUR entries
// only, by wave-uniform loads (the compiler makes them s_loads, whose waits drain the tile's LDS hand-over) or by four lanes + DPP (more tiles take
// the in-place loop: two more round trips for those waves, and the launch is as long as its slowest wave).
template <typename T>
struct TileForward {
    int count = 0;
    T v = 0;
    int c = 0;
    const int *__restrict__ colidx = nullptr;
    const T *__restrict__ val = nullptr;
    long long base = 0; // position of the next tile's first entry
};

template <typename T, int SIGMA, bool MAPPED, bool STAGED, bool FWD = false>
__device__ __forceinline__ void csr5_tile_compute(int t, int lane, const int (&c)[SIGMA], const T (&v)[SIGMA],
                                                  unsigned d, int r0, const int *__restrict__ rm /* MAPPED: the tile's row map in LDS */,
                                                  const T *__restrict__ x, const T *__restrict__ xs,
                                                  T *__restrict__ y, T *__restrict__ carry, const TileForward<T> &fw = TileForward<T>())
{
    const unsigned flags = d & kCsr5FlagMask;
    // row of the tile's first row start: r0 itself if the tile begins on a row boundary
    const unsigned f0 = (unsigned) __builtin_amdgcn_readfirstlane((int) flags);
    int seg_row = r0 + ((f0 & 1u) ? 0 : 1) + (int) (d >> kCsr5YoffShift);
    T xv[SIGMA];
#pragma unroll
    for (int i = 0; i < SIGMA; ++i) {
        if (STAGED) xv[i] = xs[c[i]];
        else xv[i] = x[c[i] >= 0 ? c[i] : 0];
    }
    T fx = 0;
    if constexpr (FWD) fx = x[fw.c]; // with the tile's own gathers: the forward part adds no round trip

    T head = 0, acc = 0;
    bool started = false;
#pragma unroll
    for (int i = 0; i < SIGMA; ++i) {
        if (flags & (1u << i)) {
            if (started) {
                y[MAPPED ? rm[seg_row - r0] : seg_row] = acc; // began and ended inside this lane
                ++seg_row;
            } else {
                head = acc;
                started = true;
            }
            acc = 0;
        }
        if (STAGED || c[i] >= 0) acc = fmadd(v[i], xv[i], acc); // STAGED: padding is 0 * xs[zero slot]
    }
    if (!started) { head = acc; acc = 0; }

    const unsigned long long starts = __ballot(started);
    T B = head;
#pragma unroll
    for (int dd = 1; dd < kWave; dd <<= 1) {
        const T nb = __shfl_down(B, dd, kWave);
        const bool cut = ((starts >> lane) & ((1ull << dd) - 1ull)) != 0ull;
        if (!cut && lane + dd < kWave) B += nb;
    }
    T next = __shfl_down(B, 1, kWave);
    if (lane == kWave - 1) next = 0;
    T out = acc + next;
    bool write = started;
    if constexpr (FWD) {
        if (fw.count != 0) { // wave-uniform: the tile's last row goes on behind the tile
            T fs = fw.v * fx;
            for (int k = kWave + lane; k < fw.count; k += kWave) fs = fmadd(fw.val[fw.base + k], x[fw.colidx[fw.base + k]], fs); // more than 64 entries behind the cut: rare
            fs = lane0_value(group_sum_dpp<kWave>(fs));
            if (lane == 63 - __clzll((long long) starts)) { // the lane that holds the row's start
                out += fs;
                write = fw.count > 0; // a long row: its own workgroup writes it (nat_long_row)
            }
        }
    }
    if (write) y[MAPPED ? rm[seg_row - r0] : seg_row] = out;
    if (lane == 0) carry[t] = B;
}


// One tile by one wavefront.  STAGED: the x windows of the workgroup's tiles are in LDS (xs) and the
// column stream is tcol16: 16-bit LDS slots, four per lane and 8-byte load (range_windows_kernel,
// pack16 = SIGMA); padding entries point at the zero slot behind the windows, so no entry needs a test.
template <typename T, int SIGMA, bool MAPPED, bool STAGED, bool RUNS = false>
__device__ __forceinline__ void csr5_tile(int t, int lane, const int *__restrict__ tile_ptr,
                                          const unsigned *__restrict__ desc, const int *__restrict__ tcol,
                                          const unsigned short *__restrict__ tcol16,
                                          const T *__restrict__ tval, const int *__restrict__ row_map, int *__restrict__ rm,
                                          const T *__restrict__ x, const T *__restrict__ xs,
                                          T *__restrict__ y, T *__restrict__ carry, const unsigned *__restrict__ lane_run = nullptr)
{
    constexpr int TN = kWave * SIGMA;
    const long long base = (long long) t * TN + lane;
    const unsigned d = desc[(long long) t * kWave + lane]; // descriptor and row range first: nothing below waits for them
    const int r0 = tile_ptr[t], r1 = MAPPED ? tile_ptr[t + 1] : 0;
    int c[SIGMA];
    T v[SIGMA];
    if constexpr (STAGED && RUNS) { // RUN group: a word per lane instead of SIGMA slots
        csr5_run_slots<SIGMA>(lane_run[(long long) t * kWave + lane], d, c);
    } else if (STAGED) {
#pragma unroll
        for (int q = 0; q < SIGMA / 4; ++q) {
            int w[4];
            ld_stream4(tcol16 + (long long) t * TN + q * (4 * kWave) + lane * 4, w);
            c[4 * q + 0] = (int) lds_slot<0>(w);
            c[4 * q + 1] = (int) lds_slot<1>(w);
            c[4 * q + 2] = (int) lds_slot<2>(w);
            c[4 * q + 3] = (int) lds_slot<3>(w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < SIGMA; ++i) c[i] = ld_stream(tcol + base + i * kWave);
    }
    csr5_load_vals<T, SIGMA>(tval, t, lane, v);
    if constexpr (MAPPED) csr5_stage_row_map<SIGMA>(lane, r0, r1, row_map, rm);
    csr5_tile_compute<T, SIGMA, MAPPED, STAGED>(t, lane, c, v, d, r0, rm, x, xs, y, carry);
}

// y = 0 for the matrix's empty rows (they are outside the compacted row space the tiles write), spread
// over the workgroups of the tile kernel itself: no separate zero-fill launch, no 8·m-byte memset.
template <typename T>
__device__ __forceinline__ void zero_empty_rows(int n_empty, const int *__restrict__ empty_list, T *__restrict__ y)
{
    const long long stride = (long long) gridDim.x * blockDim.x;
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n_empty; i += 4 * stride) { // four list entries in flight: the loop sits in front of the workgroup's first tile
        const int a = empty_list[i], b = empty_list[i + stride], c = empty_list[i + 2 * stride], d = empty_list[i + 3 * stride];
        y[a] = T(0); y[b] = T(0); y[c] = T(0); y[d] = T(0);
    }
    for (; i < n_empty; i += stride) y[empty_list[i]] = T(0);
}

template <typename T, int SIGMA, bool MAPPED>
__global__ __launch_bounds__(kBlock) void csr5_kernel(int p, const int *__restrict__ tile_ptr,
                                                      const unsigned *__restrict__ desc,
                                                      const int *__restrict__ tcol, const T *__restrict__ tval,
                                                      const int *__restrict__ row_map,
                                                      const T *__restrict__ x, T *__restrict__ y,
                                                      T *__restrict__ carry, int n_empty, const int *__restrict__ empty_list, int rm_stride, int xcd)
{
    if (MAPPED) zero_empty_rows(n_empty, empty_list, y);
    extern __shared__ __attribute__((aligned(16))) unsigned char csr5_x_lds[]; // MAPPED: the waves' row maps (rm_stride ints each)
    const int lane = threadIdx.x & (kWave - 1);
    const int t = (xcd ? xcd_block(blockIdx.x, gridDim.x) : (int) blockIdx.x) * (kBlock / kWave) + threadIdx.x / kWave;
    if (t >= p) return;
    csr5_tile<T, SIGMA, MAPPED, false>(t, lane, tile_ptr, desc, tcol, nullptr, tval, row_map, wave_row_map(csr5_x_lds, 0, rm_stride), x, nullptr, y, carry);
}

// ---- LDS-staged x windows (xwindows.hpp) --------------------------------------------------------
// A workgroup owns kCsr5GroupTiles consecutive tiles.  range_windows_kernel covers the columns of
// those tiles with up to 16 windows and writes the group's LDS slots to the 16-bit stream tcol16
// (2 B/nnz instead of tcol's 4); the executor stages the windows once and all gathers of its tiles
// are LDS reads.  Groups whose columns do not fit read the global columns in tcol and gather from L1/L2.
constexpr int kCsr5GroupTiles = 16;

template <typename T, int SIGMA, bool MAPPED>
__global__ __launch_bounds__(kBlock) void csr5_group_kernel(int group_tiles, int p, const int *__restrict__ tile_ptr,
                                                            const unsigned *__restrict__ desc,
                                                            const int *__restrict__ tcol, const unsigned short *__restrict__ tcol16,
                                                            const T *__restrict__ tval,
                                                            const int *__restrict__ row_map,
                                                            const TileWindows *__restrict__ wins, const unsigned *__restrict__ lane_run,
                                                            const T *__restrict__ x, T *__restrict__ y,
                                                            T *__restrict__ carry, int n_empty, const int *__restrict__ empty_list, int rm_off, int rm_stride)
{
    if (MAPPED) zero_empty_rows(n_empty, empty_list, y);
    extern __shared__ __attribute__((aligned(16))) unsigned char csr5_x_lds[]; // x windows, then (MAPPED) the waves' row maps at byte rm_off
    T *xs = reinterpret_cast<T *>(csr5_x_lds);
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0, runs = tw.runs != 0;
    stage_windows<kBlock, T>(tw, x, xs);
    if (staged) {
        if (threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of padding entries
        __syncthreads();
    }
    int *rm = wave_row_map(csr5_x_lds, rm_off, rm_stride);
    const int lane = threadIdx.x & (kWave - 1);
    const int t0 = blockIdx.x * group_tiles;
    for (int k = threadIdx.x / kWave; k < group_tiles; k += kBlock / kWave) {
        const int t = t0 + k;
        if (t >= p) break;
        if (runs) csr5_tile<T, SIGMA, MAPPED, true, true>(t, lane, tile_ptr, desc, tcol, tcol16, tval, row_map, rm, x, xs, y, carry, lane_run);
        else if (staged) csr5_tile<T, SIGMA, MAPPED, true>(t, lane, tile_ptr, desc, tcol, tcol16, tval, row_map, rm, x, xs, y, carry);
        else csr5_tile<T, SIGMA, MAPPED, false>(t, lane, tile_ptr, desc, tcol, tcol16, tval, row_map, rm, x, xs, y, carry);
    }
}

// The same group kernel, two tiles deep (plans whose groups are all staged).  A wave of
// csr5_group_kernel takes its tiles strictly one after the other: descriptor + row range + streams -> wait -> [row map -> wait] ->
// compute -> next tile, and the x windows (up to 128 KiB) leave a CU 1-4 workgroups to hide those round trips behind.  Here a
// wave reads the row ranges of ALL its tiles first (one vector load, lane j = tile j, handed out by v_readlane), issues tile j + 1's
// streams -- and the first 64 entries of its row map, no longer behind the row range -- before it computes tile j, and keeps the two
// tiles in two register sets (the loop is unrolled, so the sets are never copied).  The column slots stay packed (two per
// register) until they are used.  Config 4 (fp32): long rows of the SELL / CSR-vector schedules (MAPPED) and CSR5 itself.
template <typename T, int SIGMA>
struct Csr5TileRegs {
    int w[SIGMA / 4][2]; // 16-bit slots, packed (RUN group: w[0][0] = the lane's run word)
    T v[SIGMA];
    unsigned d;
    int first;           // MAPPED: row_map[r0 + min(lane, span)]
};

template <typename T, int SIGMA, bool MAPPED, bool RUNS>
__device__ __forceinline__ void csr5_tile_issue(int t, int lane, int r0, int r1, const unsigned *__restrict__ desc, const unsigned short *__restrict__ tcol16,
                                                const T *__restrict__ tval, const int *__restrict__ row_map, Csr5TileRegs<T, SIGMA> &R,
                                                const unsigned *__restrict__ lane_run)
{
    constexpr int TN = kWave * SIGMA;
    R.d = desc[(long long) t * kWave + lane];
    if constexpr (RUNS) {
        R.w[0][0] = (int) lane_run[(long long) t * kWave + lane];
    } else {
#pragma unroll
        for (int q = 0; q < SIGMA / 4; ++q) {
            const i32x2 u = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(tcol16 + (long long) t * TN + q * (4 * kWave) + lane * 4));
            R.w[q][0] = u.x; R.w[q][1] = u.y;
        }
    }
    csr5_load_vals<T, SIGMA>(tval, t, lane, R.v);
    R.first = 0;
    if constexpr (MAPPED) {
        const int span = r1 - r0;
        R.first = row_map[r0 + (lane < span ? lane : span)];
    }
}

template <typename T, int SIGMA, bool MAPPED, bool RUNS>
__device__ __forceinline__ void csr5_tile_finish(int t, int lane, int r0, int r1, const Csr5TileRegs<T, SIGMA> &R, const int *__restrict__ row_map, int *__restrict__ rm,
                                                 const T *__restrict__ x, const T *__restrict__ xs, T *__restrict__ y, T *__restrict__ carry)
{
    int c[SIGMA];
    if constexpr (RUNS) {
        csr5_run_slots<SIGMA>((unsigned) R.w[0][0], R.d, c);
    } else {
#pragma unroll
        for (int q = 0; q < SIGMA / 4; ++q) {
            c[4 * q + 0] = (int) ((unsigned) R.w[q][0] & 0xffffu);
            c[4 * q + 1] = (int) ((unsigned) R.w[q][0] >> 16);
            c[4 * q + 2] = (int) ((unsigned) R.w[q][1] & 0xffffu);
            c[4 * q + 3] = (int) ((unsigned) R.w[q][1] >> 16);
        }
    }
    if constexpr (MAPPED) {
        const int span = r1 - r0; // wave-uniform
        if (span >= kWave) {      // many row starts in this tile: the rest of its row map (csr5_stage_row_map)
            int tmp[SIGMA];
#pragma unroll
            for (int j = 1; j <= SIGMA; ++j) {
                const int k = j * kWave + lane;
                tmp[j - 1] = row_map[r0 + (k < span ? k : span)];
            }
#pragma unroll
            for (int j = 1; j <= SIGMA; ++j) rm[j * kWave + lane] = tmp[j - 1];
        }
        rm[lane] = R.first;
        wave_lds_sync();
    }
    csr5_tile_compute<T, SIGMA, MAPPED, true>(t, lane, c, R.v, R.d, r0, rm, x, xs, y, carry);
}

constexpr int kCsr5PipeMaxGroupTiles = kWave * (kBlock / kWave); // lane j of a wave holds the row range of the wave's j-th tile

// body of the two-deep group kernel; RUNS: the group is a RUN group (the whole workgroup takes the same instantiation)
template <typename T, int SIGMA, bool MAPPED, bool RUNS>
__device__ __forceinline__ void csr5_group_pipe_body(int group_tiles, int p, const int *__restrict__ tile_ptr,
                                                     const unsigned *__restrict__ desc, const unsigned short *__restrict__ tcol16,
                                                     const T *__restrict__ tval, const int *__restrict__ row_map,
                                                     const TileWindows &tw, const unsigned *__restrict__ lane_run,
                                                     const T *__restrict__ x, T *__restrict__ y, T *__restrict__ carry,
                                                     unsigned char *__restrict__ lds, int rm_off, int rm_stride)
{
    T *xs = reinterpret_cast<T *>(lds);
    constexpr int NW = kBlock / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave));
    const int t0 = blockIdx.x * group_tiles;
    // the wave's tiles: t0 + wave + j * NW, j < nt
    int last = group_tiles < p - t0 ? group_tiles : p - t0; // tiles of this group that exist
    const int nt = last > wave ? (last - wave + NW - 1) / NW : 0;
    int lp0 = 0, lp1 = 0; // lane j: row range of the wave's j-th tile (one vector load for all of them, in flight during the x staging)
    if (lane < nt) {
        lp0 = tile_ptr[t0 + wave + lane * NW];
        lp1 = tile_ptr[t0 + wave + lane * NW + 1];
    }
    Csr5TileRegs<T, SIGMA> R0, R1;
    // tile 0's streams are issued before the x staging and the barrier (its row map follows the row range: after the barrier)
    auto issue = [&](int j, Csr5TileRegs<T, SIGMA> &R) {
        csr5_tile_issue<T, SIGMA, MAPPED, RUNS>(t0 + wave + j * NW, lane, __builtin_amdgcn_readlane(lp0, j), __builtin_amdgcn_readlane(lp1, j), desc, tcol16, tval, row_map, R, lane_run);
    };
    int *rm = wave_row_map(lds, rm_off, rm_stride);
    auto finish = [&](int j, const Csr5TileRegs<T, SIGMA> &R) {
        csr5_tile_finish<T, SIGMA, MAPPED, RUNS>(t0 + wave + j * NW, lane, __builtin_amdgcn_readlane(lp0, j), __builtin_amdgcn_readlane(lp1, j), R, row_map, rm, x, xs, y, carry);
    };
    if (nt > 0) issue(0, R0);
    stage_windows<kBlock, T>(tw, x, xs);
    if (threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of padding entries
    __syncthreads();
    for (int j = 0; j < nt; j += 2) {
        if (j + 1 < nt) issue(j + 1, R1);
        finish(j, R0);
        if (j + 1 >= nt) break;
        if (j + 2 < nt) issue(j + 2, R0);
        finish(j + 1, R1);
    }
}

template <typename T, int SIGMA, bool MAPPED>
__global__ __launch_bounds__(kBlock) void csr5_group_pipe_kernel(int group_tiles, int p, const int *__restrict__ tile_ptr,
                                                                 const unsigned *__restrict__ desc, const unsigned short *__restrict__ tcol16,
                                                                 const T *__restrict__ tval, const int *__restrict__ row_map,
                                                                 const TileWindows *__restrict__ wins, const unsigned *__restrict__ lane_run,
                                                                 const T *__restrict__ x, T *__restrict__ y,
                                                                 T *__restrict__ carry, int n_empty, const int *__restrict__ empty_list, int rm_off, int rm_stride)
{
    if (MAPPED) zero_empty_rows(n_empty, empty_list, y);
    extern __shared__ __attribute__((aligned(16))) unsigned char csr5_x_lds[]; // x windows, then (MAPPED) the waves' row maps at byte rm_off
    const TileWindows &tw = wins[blockIdx.x]; // every group of the plan is staged (launch_csr5_form)
    if (tw.runs) csr5_group_pipe_body<T, SIGMA, MAPPED, true>(group_tiles, p, tile_ptr, desc, tcol16, tval, row_map, tw, lane_run, x, y, carry, csr5_x_lds, rm_off, rm_stride);
    else csr5_group_pipe_body<T, SIGMA, MAPPED, false>(group_tiles, p, tile_ptr, desc, tcol16, tval, row_map, tw, lane_run, x, y, carry, csr5_x_lds, rm_off, rm_stride);
}

// ---- natural-layout tiles (the nnz-split schedule: Method_Balanced2 / Method_Balanced_Yid) ------------
// Same tiles, descriptors and carry fix-up, but NO transposed copies: the tile's 64*SIGMA entries are
// read from the caller-ordered ColIdx / Val arrays (lane x owns entries x*SIGMA .. x*SIGMA+SIGMA-1, as
// the reference's equal-nnz workers own consecutive nnz, parallel_balanced2_spmv.c:41-53).  A lane
// reading its own SIGMA consecutive values directly would make every load instruction touch 64
// different cache lines (measured: tiles wider than 4 entries per lane lost more than they won),
// so the wave fetches the tile with fully coalesced 16-byte loads, parks it in LDS and each lane
// reads its row back (row stride padded by 16 B / 4 B: conflict-free for ds_read_b128 / b32) -- in two
// halves of 32 lane rows, which halves the buffer (LDS is what limits the resident workgroups here).
// Extra HBM: none for values; a 2 B/nnz slot stream for staged groups.
template <typename T, int SIGMA, bool HALF>
struct NatLds {
    static constexpr int kRows = HALF ? kWave / 2 : kWave;       // HALF: the tile is handed over in two halves of 32 lane rows
    static constexpr int kValRow = SIGMA * (int) sizeof(T) + 16; // bytes per lane row, padded
    static constexpr int kColRow = SIGMA * 2 + 4;
    static constexpr int kValBytes = kRows * kValRow;
    static constexpr int kColBytes = (kRows * kColRow + 15) & ~15;
    static constexpr int kBytes = kValBytes + kColBytes;         // per wavefront: 11.5 KiB (5.8 KiB HALF) for fp64, sigma = 16
};

template <typename T, int SIGMA, bool MAPPED, bool STAGED, bool HALF, bool FWD = false>
__device__ __forceinline__ void nat_tile(int t, int lane, int nnz, unsigned zslot, unsigned char *__restrict__ wl,
                                         const int *__restrict__ tile_ptr, const unsigned *__restrict__ desc,
                                         const int *__restrict__ colidx, const unsigned short *__restrict__ col16,
                                         const T *__restrict__ val, const int *__restrict__ row_map, int *__restrict__ rm,
                                         const T *__restrict__ x, const T *__restrict__ xs,
                                         T *__restrict__ y, T *__restrict__ carry, const int *__restrict__ fwd = nullptr)
{
    using NL = NatLds<T, SIGMA, HALF>;
    const unsigned d = desc[(long long) t * kWave + lane]; // descriptor and row range first: in flight with the tile itself
    const int r0 = tile_ptr[t], r1 = MAPPED ? tile_ptr[t + 1] : 0;
    constexpr int TN = kWave * SIGMA;
    TileForward<T> fw;
    if constexpr (FWD) { // the first entries behind the tile, loaded whether or not they are needed (fwd[t] decides when it arrives): no round trip of their own
        fw.base = (long long) (t + 1) * TN;
        fw.count = fwd[t];
        const long long q = fw.base + lane < nnz ? fw.base + lane : nnz - 1;
        fw.v = val[q];
        fw.c = colidx[q];
        fw.colidx = colidx;
        fw.val = val;
    }
    constexpr int EPL = 16 / (int) sizeof(T);              // values per 16-byte load
    constexpr int VL = SIGMA / EPL;                        // value loads per lane
    constexpr int CB = SIGMA * 2 < 16 ? SIGMA * 2 : 16;    // bytes of slot stream per lane and load
    constexpr int CS = CB / 2, CL = SIGMA * 2 / CB;        // slots per load, loads per lane
    const long long tb = (long long) t * TN;
    const int left = nnz - tb < TN ? (int) (nnz - tb) : TN; // entries of this tile that exist
    unsigned char *lv = wl, *lc = wl + NL::kValBytes;
    // 1. coalesced fetch into registers: load group j of a lane holds entries (j*64 + lane) * EPL ... of the tile;
    //    every load of a whole tile is issued before the first LDS write, so they are in flight together
    int c[SIGMA];
    T tv[VL][EPL];
    unsigned tw[CL][CS / 2]; // slot pairs
    if (left == TN) {
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int p = (j * kWave + lane) * EPL;
            if constexpr (EPL == 4) {
                const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(val + tb + p));
                tv[j][0] = q.x; tv[j][1] = q.y; tv[j][2] = q.z; tv[j][3] = q.w;
            } else {
                const f64x2 q = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + tb + p));
                tv[j][0] = q.x; tv[j][1] = q.y;
            }
        }
        if (STAGED) {
#pragma unroll
            for (int j = 0; j < CL; ++j) {
                const int p = (j * kWave + lane) * CS;
                if constexpr (CB == 16) {
                    const i32x4 q = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(col16 + tb + p));
                    tw[j][0] = (unsigned) q.x; tw[j][1] = (unsigned) q.y; tw[j][2] = (unsigned) q.z; tw[j][3] = (unsigned) q.w;
                } else {
                    const i32x2 q = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(col16 + tb + p));
                    tw[j][0] = (unsigned) q.x; tw[j][1] = (unsigned) q.y;
                }
            }
        } else { // global columns: straight from ColIdx (this path is gather-bound anyway)
#pragma unroll
            for (int q = 0; q < SIGMA; q += 4) ld_stream4(colidx + tb + lane * SIGMA + q, *reinterpret_cast<int(*)[4]>(&c[q]));
        }
    } else { // the matrix's last tile: entry by entry, missing entries = 0 * zero slot
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int p = (j * kWave + lane) * EPL;
#pragma unroll
            for (int e = 0; e < EPL; ++e) tv[j][e] = p + e < left ? val[tb + p + e] : T(0);
        }
        if (STAGED) {
#pragma unroll
            for (int j = 0; j < CL; ++j) {
                const int p = (j * kWave + lane) * CS;
#pragma unroll
                for (int e = 0; e < CS; e += 2) {
                    const unsigned lo = p + e < left ? col16[tb + p + e] : zslot, hi = p + e + 1 < left ? col16[tb + p + e + 1] : zslot;
                    tw[j][e / 2] = lo | (hi << 16);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; ++i) c[i] = lane * SIGMA + i < left ? colidx[tb + lane * SIGMA + i] : -1;
        }
    }
    if constexpr (FWD) { // checked behind the tile's own loads, so that waiting for fwd[t] costs no round trip
        if (__builtin_amdgcn_readfirstlane(fw.count) == -2) return; // the tile lies inside a long row (nat_long_row writes it): no hand-over, no gathers
    }
    // 2. hand-over through LDS in two halves (lane rows 0..31, then 32..63: half the buffer, twice the syncs):
    //    entry p of the tile belongs to lane row p / SIGMA, position p % SIGMA
    T v[SIGMA];
#pragma unroll
    for (int half = 0; half < (HALF ? 2 : 1); ++half) {
        const int r0 = half * NL::kRows;
#pragma unroll
        for (int j = 0; j < VL; ++j) {
            const int p = (j * kWave + lane) * EPL, row = p / SIGMA - r0;
            if (row >= 0 && row < NL::kRows) {
                T *dst = reinterpret_cast<T *>(lv + row * NL::kValRow + (p % SIGMA) * (int) sizeof(T));
#pragma unroll
                for (int e = 0; e < EPL; ++e) dst[e] = tv[j][e];
            }
        }
        if (STAGED) {
#pragma unroll
            for (int j = 0; j < CL; ++j) {
                const int p = (j * kWave + lane) * CS, row = p / SIGMA - r0;
                if (row >= 0 && row < NL::kRows) {
                    unsigned *dst = reinterpret_cast<unsigned *>(lc + row * NL::kColRow + (p % SIGMA) * 2);
#pragma unroll
                    for (int e = 0; e < CS / 2; ++e) dst[e] = tw[j][e];
                }
            }
        }
        wave_lds_sync();
        const int mine = lane - r0;
        if (mine >= 0 && mine < NL::kRows) {
            const T *src = reinterpret_cast<const T *>(lv + mine * NL::kValRow);
#pragma unroll
            for (int i = 0; i < SIGMA; ++i) v[i] = src[i];
            if (STAGED) {
                const unsigned *cs = reinterpret_cast<const unsigned *>(lc + mine * NL::kColRow);
#pragma unroll
                for (int i = 0; i < SIGMA; i += 2) {
                    const unsigned w = cs[i / 2];
                    c[i] = (int) (w & 0xffffu);
                    c[i + 1] = (int) (w >> 16);
                }
            }
        }
        wave_lds_sync(); // the buffer is free for the other half / the wave's next tile
    }
    if constexpr (MAPPED) csr5_stage_row_map<SIGMA>(lane, r0, r1, row_map, rm);
    if constexpr (FWD) {
        fw.count = __builtin_amdgcn_readfirstlane(fw.count);
        if (lane >= fw.count) { fw.v = T(0); fw.c = 0; }
    }
    csr5_tile_compute<T, SIGMA, MAPPED, STAGED, FWD>(t, lane, c, v, d, r0, rm, x, xs, y, carry, fw);
}

// Forward completion: which tiles finish their last row, and which rows are too long for that (inspector, one thread per tile).
// fwd[t] = entries of tile t + 1 (.. further tiles: never, such rows are long) that belong to the row open at the end of tile t, if that row STARTS
// in tile t and is at most one tile long; -1 if it starts in tile t and is longer (row appended to long_list, counters[0] / [1] = number / longest);
// -2 if the whole tile lies inside a long row (the wave returns before it gathers anything); else 0.
static __global__ __launch_bounds__(kBlock) void nat_forward_kernel(int p, int tile_nnz, const int *__restrict__ rp, const int *__restrict__ tile_ptr, const int *__restrict__ row_map,
                                                                    int *__restrict__ fwd, int4 *__restrict__ long_list, int cap, int *__restrict__ counters)
{
    auto list = [&](int r, long long b, long long e) { // (y row, first entry, end): everything the row's workgroup needs, in one load
        const int k = atomicAdd(&counters[0], 1);
        if (k < cap) long_list[k] = make_int4(row_map ? row_map[r] : r, (int) b, (int) e, 0);
        atomicMax(&counters[1], (int) (e - b < INT_MAX ? e - b : INT_MAX));
    };
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= p) return;
    int f = 0;
    {
        const int r0 = tile_ptr[t];
        const long long b0 = rp[r0], e0 = rp[r0 + 1], lo = (long long) t * tile_nnz;
        if (b0 <= lo && e0 >= lo + tile_nnz && e0 - b0 > tile_nnz) { // the whole tile lies inside a long row: nothing for the tile to do
            if (b0 == lo) list(r0, b0, e0); // ... which starts with the tile: list it here
            fwd[t] = -2;
            return;
        }
    }
    if (t + 1 < p) {
        const long long cut = (long long) (t + 1) * tile_nnz;
        const int r = tile_ptr[t + 1];
        const long long b = rp[r], e = rp[r + 1];
        if (b < cut && e > cut && b >= cut - tile_nnz) {
            if (e - b <= tile_nnz) f = (int) (e - cut);
            else {
                f = -1;
                list(r, b, e);
            }
        }
    }
    fwd[t] = f;
}

// One long row by one workgroup: thread i takes entries i, i + 256, ... -- kNatLongU in flight, so that a row of up to 256 kNatLongU entries is two memory round
// trips behind the list entry, like a tile (the launch is one round of workgroups, all in flight together: a round trip takes microseconds, and a row that
// needed five would be the launch's tail) --, fixed-shape combine: the same bits on every run.
constexpr int kNatLongU = 20;    // 5120 entries per pass (webbase-1M's longest row: 4700).  fp64: 98 registers instead of 70, five workgroups per CU instead of six -- measured against
                                 // 14 per thread (74 registers, two passes for the longest rows): 25.9 vs 28.2 us on the stand-in; the row with two passes is the launch's tail
template <typename T>
__device__ __forceinline__ void nat_long_row(int4 row, const int *__restrict__ colidx, const T *__restrict__ val, const T *__restrict__ x, T *__restrict__ y)
{
    __shared__ T part[kBlock / kWave];
    const int e = row.z;
    constexpr int U = kNatLongU;
    T acc = 0;
    for (int k = row.y + (int) threadIdx.x; k < e; k += U * kBlock) {
        int c[U];
        T v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int q = k + j * kBlock;
            c[j] = q < e ? colidx[q] : 0;
            v[j] = q < e ? val[q] : T(0);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc = fmadd(v[j], x[c[j]], acc);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) y[row.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

template <typename T, int SIGMA, bool MAPPED, bool FWD = false>
__global__ __launch_bounds__(kBlock) void nat_kernel(int p, int nnz, const int *__restrict__ tile_ptr,
                                                     const unsigned *__restrict__ desc,
                                                     const int *__restrict__ colidx, const T *__restrict__ val,
                                                     const int *__restrict__ row_map,
                                                     const T *__restrict__ x, T *__restrict__ y, T *__restrict__ carry, int n_empty, const int *__restrict__ empty_list,
                                                     int rm_stride, int xcd,
                                                     // FWD ("forward completion": no carry fix-up launch): a row cut by a tile boundary is finished by the tile it starts in
                                                     // (fwd[t] entries of the next tile); rows longer than a tile by the first long_blocks workgroups (a multiple of 8:
                                                     // the tiles' XCD order stays), one row each
                                                     const int *__restrict__ fwd = nullptr, const int4 *__restrict__ long_list = nullptr, int n_long = 0, int long_blocks = 0)
{
    if (MAPPED) zero_empty_rows(n_empty, empty_list, y);
    extern __shared__ __attribute__((aligned(16))) unsigned char csr5_x_lds[]; // MAPPED: the waves' row maps (rm_stride ints each)
    __shared__ __attribute__((aligned(16))) unsigned char nat_lds[kBlock / kWave][NatLds<T, SIGMA, false>::kBytes];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    int b = (int) blockIdx.x, nb = (int) gridDim.x;
    if constexpr (FWD) {
        if (b < long_blocks) {
            if (b < n_long) nat_long_row<T>(long_list[b], colidx, val, x, y);
            return;
        }
        b -= long_blocks;
        nb -= long_blocks;
    }
    const int t = (xcd ? xcd_block(b, nb) : b) * (kBlock / kWave) + wave;
    if (t >= p) return;
    nat_tile<T, SIGMA, MAPPED, false, false, FWD>(t, lane, nnz, 0u, nat_lds[wave], tile_ptr, desc, colidx, nullptr, val, row_map, wave_row_map(csr5_x_lds, 0, rm_stride), x, nullptr, y, carry, fwd);
}

// HALF: two-half hand-over (half the tile buffers, twice the wave syncs: ~8 % slower per tile) -- chosen by
// the launcher when the x windows are so large that the full buffers would leave one workgroup per CU.
template <typename T, int SIGMA, bool MAPPED, bool HALF>
__global__ __launch_bounds__(kBlock) void nat_group_kernel(int group_tiles, int p, int nnz, const int *__restrict__ tile_ptr,
                                                           const unsigned *__restrict__ desc,
                                                           const int *__restrict__ colidx, const unsigned short *__restrict__ col16,
                                                           const T *__restrict__ val,
                                                           const int *__restrict__ row_map,
                                                           const TileWindows *__restrict__ wins,
                                                           const T *__restrict__ x, T *__restrict__ y,
                                                           T *__restrict__ carry, int n_empty, const int *__restrict__ empty_list, int rm_off, int rm_stride)
{
    if (MAPPED) zero_empty_rows(n_empty, empty_list, y);
    extern __shared__ __attribute__((aligned(16))) unsigned char csr5_x_lds[]; // x windows, then (MAPPED) the waves' row maps at byte rm_off
    __shared__ __attribute__((aligned(16))) unsigned char nat_lds[kBlock / kWave][NatLds<T, SIGMA, HALF>::kBytes];
    int *rm = wave_row_map(csr5_x_lds, rm_off, rm_stride);
    T *xs = reinterpret_cast<T *>(csr5_x_lds);
    const TileWindows &tw = wins[blockIdx.x];
    const bool staged = tw.nwin > 0;
    stage_windows<kBlock, T>(tw, x, xs);
    if (staged) {
        if (threadIdx.x == 0) xs[tw.total] = T(0); // the zero slot of the last tile's missing entries
        __syncthreads();
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int t0 = blockIdx.x * group_tiles;
    for (int k = wave; k < group_tiles; k += kBlock / kWave) {
        const int t = t0 + k;
        if (t >= p) break;
        if (staged) nat_tile<T, SIGMA, MAPPED, true, HALF>(t, lane, nnz, (unsigned) tw.total, nat_lds[wave], tile_ptr, desc, colidx, col16, val, row_map, rm, x, xs, y, carry);
        else nat_tile<T, SIGMA, MAPPED, false, HALF>(t, lane, nnz, 0u, nat_lds[wave], tile_ptr, desc, colidx, col16, val, row_map, rm, x, xs, y, carry);
    }
}

template <typename T, bool MAPPED>
__global__ __launch_bounds__(kBlock) void csr5_fixup_kernel(int p, const int *__restrict__ tile_ptr,
                                                            const int *__restrict__ run_len,
                                                            const int *__restrict__ row_map,
                                                            const T *__restrict__ carry, T *__restrict__ y)
{
    const int t = blockIdx.x * kBlock + threadIdx.x + 1;
    if (t >= p) return;
    const int n = run_len[t];
    if (n <= 0) return;
    T sum = 0;
    const int r = tile_ptr[t];
    for (int u = t; u < t + n; u += 8) { // a row that spans many tiles: eight carries in flight, added in tile order as before
        T c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = carry[u + k < t + n ? u + k : t + n - 1];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (u + k < t + n) sum += c[k];
    }
    y[MAPPED ? row_map[r] : r] += sum;
}

} // namespace spmv
