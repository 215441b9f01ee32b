// xwindows.hpp -- LDS-staged x windows, shared by every tile schedule (north_star: "LDS staging of
// x-vector tiles").
//
// A tile (256 rows of CSR-vector, an equal-nnz row block of Balanced, 16 tiles of CSR5 / nnz-split, a
// sigma window of SELL) references some set of columns.  The inspector covers that set with up to
// 16 WINDOWS -- one when the plain span [min, max] fits the LDS budget (banded matrices), several
// when the columns sit in a few far-apart bands (3-D stencils) -- and rewrites the tile's private
// copy of the column indices so that each entry already IS the LDS slot of its column.  The
// executor stages the windows once per tile (coalesced reads of x) and every gather becomes
// xs[slot]: no index arithmetic, no L1/TA traffic.  Tiles whose columns do not fit keep global
// indices and gather from L1/L2 as before.
//
// Why it matters on this chip (measured): a scattered gather instruction occupies the
// L1/TA path for up to 64 cache-line lookups, an LDS gather for a few cycles -- CSR5 on the banded
// config 2 went 0.72 -> 0.60 ms, on config 4 1.72 -> 0.70 ms, CSR-vector on a 27-point stencil
// 0.78 -> 0.60 ms.
#pragma once
#include <climits>
#include <type_traits>
#include "common.hpp"

namespace spmv {

constexpr int kWinMax = 16;           // windows per tile
constexpr int kWinSegShift = 6;       // windows are built from 64-column segments
constexpr int kWinBitmapWords = 1024; // 32768 segments: spans up to 2M columns are analysed

struct TileWindows {
    int nwin;            // 0: tile not staged (its column copy holds global columns)
    int total;           // staged elements = sum of len
    int start[kWinMax];  // first column of each window (ascending)
    int len[kWinMax];
    int base[kWinMax];   // LDS slot of the window's first column
    int runs;            // CSR-vector tiles only: 1 = every row's columns are ONE run of consecutive columns (csr_vector_tile.hpp: no column stream)
};

// sum of TileWindows::total over `count` tiles: the x elements one launch stages (traffic model, spmv_hip_info.stream_bytes)
static __global__ __launch_bounds__(kBlock) void wins_total_kernel(int count, const TileWindows *__restrict__ wins, unsigned long long *__restrict__ sum,
                                                            unsigned long long *__restrict__ staged_tiles)
{
    unsigned long long t = 0, c = 0;
    for (long long i = (long long) blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long) gridDim.x * kBlock)
        if (wins[i].nwin > 0) { t += (unsigned long long) wins[i].total; c += 1; }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { t += __shfl_xor(t, o, kWave); c += __shfl_xor(c, o, kWave); }
    if ((threadIdx.x & (kWave - 1)) == 0 && c) { atomicAdd(sum, t); atomicAdd(staged_tiles, c); }
}

// Stage the tile's windows: xs[base_w + i] = x[start_w + i].  NT = threads of the workgroup.  16-byte loads, four in flight per
// thread, from the first 16-byte boundary of the window on (x is the caller's pointer: any element alignment): a wide window --
// 10 000 columns of a CSR5 / SELL group whose rows scatter +-4096 around the diagonal -- is three round trips to L2 instead of eleven,
// and the workgroup does nothing else while it stages.
template <int NT, typename T>
__device__ __forceinline__ void stage_windows(const TileWindows &tw, const T *__restrict__ x, T *__restrict__ xs)
{
    constexpr int E = 16 / (int) sizeof(T);
    using V = typename std::conditional<E == 4, f32x4, f64x2>::type;
    const int nwin = tw.nwin;
    for (int w = 0; w < nwin; ++w) {
        const int st = tw.start[w], ln = tw.len[w], bs = tw.base[w];
        const T *src = x + st;
        if (ln <= 16 * NT) { // narrow windows (banded matrices: a few hundred to a few thousand columns): element loads, four in flight per thread --
            int i = threadIdx.x; // the set-up of the 16-byte path costs such a group more than it saves (nnz-split on config 2: +1.3 %)
            for (; i + 3 * NT < ln; i += 4 * NT) {
                const T a = src[i], b = src[i + NT], c = src[i + 2 * NT], d = src[i + 3 * NT];
                xs[bs + i] = a; xs[bs + i + NT] = b; xs[bs + i + 2 * NT] = c; xs[bs + i + 3 * NT] = d;
            }
            for (; i < ln; i += NT) xs[bs + i] = src[i];
            continue;
        }
        int head = (int) (((16u - (unsigned) (reinterpret_cast<size_t>(src) & 15u)) & 15u) / (unsigned) sizeof(T)); // elements in front of the boundary
        if (head > ln) head = ln;
        if ((int) threadIdx.x < head) xs[bs + threadIdx.x] = src[threadIdx.x];
        const int nv = (ln - head) / E;
        const V *vs = reinterpret_cast<const V *>(src + head);
        T *dst = xs + bs + head;
        auto put = [&](int i, const V &q) {
            if constexpr (E == 4) { dst[4 * i] = q.x; dst[4 * i + 1] = q.y; dst[4 * i + 2] = q.z; dst[4 * i + 3] = q.w; }
            else { dst[2 * i] = q.x; dst[2 * i + 1] = q.y; }
        };
        int i = threadIdx.x;
        for (; i + 3 * NT < nv; i += 4 * NT) {
            const V a = vs[i], b = vs[i + NT], c = vs[i + 2 * NT], d = vs[i + 3 * NT];
            put(i, a); put(i + NT, b); put(i + 2 * NT, c); put(i + 3 * NT, d);
        }
        for (; i < nv; i += NT) put(i, vs[i]);
        const int done = head + nv * E; // fewer than E elements left
        if ((int) threadIdx.x < ln - done) xs[bs + done + threadIdx.x] = src[done + threadIdx.x];
    }
}

// Inspector core, run by one 256-thread workgroup per tile.  `loop(body)` must call body(c, pos) for
// every (column, position) of the tile, distributing the entries over the workgroup's threads
// (entries with c < 0 are padding and do not count).  On return `out` describes the windows and,
// when the tile is staged and `rewrite` is set, store(pos, slot, total) has been called for every
// entry with the LDS slot of its column (slot = -1 for padding; total = the tile's staged
// elements, i.e. the first free slot -- executors keep a zero there for padding / masked entries).
template <typename Loop, typename Store>
__device__ __forceinline__ void build_windows(int n, int max_cols, Loop loop, Store store,
                                              TileWindows &out, int *__restrict__ staged /* [0] count, [1] max total */,
                                              bool rewrite = true)
{
    __shared__ unsigned bitmap[kWinBitmapWords];
    __shared__ int smin[kBlock / kWave], smax[kBlock / kWave], wave_cnt[kBlock / kWave];
    __shared__ int s_start[kWinMax], s_end[kWinMax], s_base[kWinMax];
    __shared__ int s_nwin, s_total, s_bits;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;

    int mn = INT_MAX, mx = -1;
    loop([&](int c, long long) { if (c >= 0) { mn = min(mn, c); mx = max(mx, c); } });
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, kWave));
        mx = max(mx, __shfl_xor(mx, o, kWave));
    }
    if (lane == 0) { smin[wave] = mn; smax[wave] = mx; }
    if (threadIdx.x == 0) { s_nwin = 0; s_total = 0; s_bits = 0; }
    __syncthreads();
    mn = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
    mx = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    const long long span = mx >= mn ? (long long) mx - mn + 1 : 0;
    const int seg_lo = mn >> kWinSegShift;
    const int nseg = span > 0 ? (mx >> kWinSegShift) - seg_lo + 1 : 0;
    const int nwords = (nseg + 31) / 32;

    if (span > 0 && span <= max_cols) { // one window: the plain span
        if (threadIdx.x == 0) { s_nwin = 1; s_total = (int) span; s_start[0] = mn; s_end[0] = mx + 1; s_base[0] = 0; }
    } else if (span > 0 && nwords <= kWinBitmapWords) { // several windows: runs of touched 64-column segments
        for (int w = threadIdx.x; w < nwords; w += kBlock) bitmap[w] = 0u;
        __syncthreads();
        loop([&](int c, long long) {
            if (c >= 0) {
                const int s = (c >> kWinSegShift) - seg_lo;
                atomicOr(&bitmap[s >> 5], 1u << (s & 31));
            }
        });
        __syncthreads();
        int starts = 0, bits = 0; // each thread owns 4 consecutive words
        for (int k = 0; k < 4; ++k) {
            const int w = threadIdx.x * 4 + k;
            if (w < nwords) {
                const unsigned cur = bitmap[w];
                const unsigned prev = w > 0 ? bitmap[w - 1] >> 31 : 0u;
                starts += __popc(cur & ~((cur << 1) | prev));
                bits += __popc(cur);
            }
        }
        int inc = starts;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) bits += __shfl_xor(bits, o, kWave);
        if (lane == kWave - 1) wave_cnt[wave] = inc;
        if (lane == 0) atomicAdd(&s_bits, bits);
        __syncthreads();
        int idx = inc - starts; // run starts in front of this thread's words
        for (int w = 0; w < wave; ++w) idx += wave_cnt[w];
        const int runs = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (runs <= kWinMax && ((long long) s_bits << kWinSegShift) <= max_cols) {
            for (int k = 0; k < 4; ++k) {
                const int w = threadIdx.x * 4 + k;
                if (w >= nwords) break;
                const unsigned cur = bitmap[w];
                const unsigned prev = w > 0 ? bitmap[w - 1] >> 31 : 0u;
                unsigned st = cur & ~((cur << 1) | prev);
                while (st) { // run starts of this word, ascending
                    const int b = __ffs((int) st) - 1;
                    st &= st - 1;
                    s_start[idx++] = (seg_lo + w * 32 + b) << kWinSegShift;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) { // run ends: walk each run (<= max_cols/64 set segments in total)
                int base = 0;
                for (int k = 0; k < runs; ++k) {
                    int s = (s_start[k] >> kWinSegShift) - seg_lo;
                    while (s < nseg && ((bitmap[s >> 5] >> (s & 31)) & 1u)) ++s;
                    int end = (seg_lo + s) << kWinSegShift;
                    if (end > n) end = n;
                    s_end[k] = end;
                    s_base[k] = base;
                    base += end - s_start[k];
                }
                s_total = base;
                s_nwin = runs;
            }
        }
    }
    __syncthreads();
    const int nwin = s_nwin;
    if (threadIdx.x == 0) {
        out.nwin = nwin;
        out.total = nwin ? s_total : 0;
        out.runs = 0;
        for (int k = 0; k < kWinMax; ++k) {
            out.start[k] = k < nwin ? s_start[k] : 0;
            out.len[k] = k < nwin ? s_end[k] - s_start[k] : 0;
            out.base[k] = k < nwin ? s_base[k] : 0;
        }
        if (nwin) { atomicAdd(staged, 1); atomicMax(staged + 1, s_total); }
    }
    if (nwin == 0 || !rewrite) return; // rewrite = false: count only (the caller decides, then runs again)
    const int total = s_total;
    loop([&](int c, long long pos) { // the tile's private column copy now holds LDS slots
        int slot = -1;
        if (c >= 0) {
            int w = 0;
            for (int k = 1; k < nwin; ++k) w = c >= s_start[k] ? k : w; // windows are sorted by start
            slot = s_base[w] + (c - s_start[w]);
        }
        store(pos, slot, total);
    });
}

// Inspector for tiles that are CONTIGUOUS RANGES of a private column array (CSR5 tile groups,
// nnz-split tile groups, SELL sigma windows): group g covers cols[b, e) with
// b = bounds ? bounds[g * bstride] * scale : g * group_len,  e likewise (clipped to total).
// cols16 == NULL: in place (cols then holds int32 slots for staged groups).  Else cols is left
// alone (unstaged groups keep reading global columns from it) and the staged groups' slots go to the
// 16-bit stream cols16, padding entries to the zero slot; pack16 = 0: same positions, pack16 =
// sigma (CSR5): position t*64*sigma + i*64 + lane -> t*64*sigma + (i/4)*256 + lane*4 + i%4, so a
// lane fetches four slots with one 8-byte load.
static __global__ __launch_bounds__(kBlock) void range_windows_kernel(long long total, long long group_len,
                                                               const long long *__restrict__ bounds, int bstride, int scale,
                                                               long long nbounds /* bounds has nbounds + 1 entries */,
                                                               int n, int max_cols, int *__restrict__ cols,
                                                               unsigned short *__restrict__ cols16, int pack16,
                                                               TileWindows *__restrict__ wins, int *__restrict__ staged, int rewrite)
{
    long long b, e;
    if (bounds) {
        long long i0 = (long long) blockIdx.x * bstride, i1 = i0 + bstride;
        if (i0 > nbounds) i0 = nbounds;
        if (i1 > nbounds) i1 = nbounds;
        b = bounds[i0] * scale;
        e = bounds[i1] * scale;
    }
    else { b = (long long) blockIdx.x * group_len; e = b + group_len; }
    if (e > total) e = total;
    auto loop = [&](auto body) {
        for (long long i = b + threadIdx.x; i < e; i += kBlock) body(cols[i], i);
    };
    auto store = [&](long long pos, int slot, int tile_total) {
        if (!cols16) { if (slot >= 0) cols[pos] = slot; return; }
        long long q = pos;
        if (pack16) {
            const long long tn = (long long) kWave * pack16, t = pos / tn;
            const int o = (int) (pos - t * tn), i = o / kWave, lane = o % kWave;
            q = t * tn + (i / 4) * (4 * kWave) + lane * 4 + (i % 4);
        }
        cols16[q] = (unsigned short) (slot >= 0 ? slot : tile_total);
    };
    build_windows(n, max_cols, loop, store, wins[blockIdx.x], staged, rewrite != 0);
}

} // namespace spmv
