/*
 * rcm.c -- reverse Cuthill-McKee reordering.
 *
 * The reference's idea of reordering is OPT_LEVEL 3: METIS k-way partitioning of the rows
 * (HyperGraphInterface.cpp:60-147), a permuted COPY of the matrix kept by the handle
 * (common.c:144-156) and the permutation published in handle->index so that the caller gathers x
 * and scatters y (test_spmv.c:95-101, 130-137).  METIS is not available (and is compiled out of
 * the reference by default); what the GPU schedules need from a reordering is a NARROW COLUMN SPAN
 * per row tile -- so that the x window of a tile fits LDS -- which is exactly what
 * bandwidth reduction gives.  Hence RCM: BFS from a pseudo-peripheral vertex, neighbours by
 * increasing degree, order reversed.  Same protocol towards the caller as the reference's.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "rcm.h"

typedef struct { int deg, v; } dv_t;

static int cmp_dv(const void *a, const void *b)
{
    const dv_t *x = (const dv_t *) a, *y = (const dv_t *) b;
    if (x->deg != y->deg) return x->deg < y->deg ? -1 : 1;
    return x->v < y->v ? -1 : (x->v > y->v);
}

/* adjacency of A + A^T without self loops (duplicates allowed: BFS ignores visited vertices) */
static int build_adjacency(int m, const int *rowptr, const int *colidx, long long **adj_ptr, int **adj)
{
    long long *ap = (long long *) calloc((size_t) m + 1, sizeof(long long));
    long long *fill;
    int *aj;
    int r;
    if (!ap) return -1;
    for (r = 0; r < m; ++r) {
        int p;
        for (p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const int c = colidx[p];
            if (c == r || c < 0 || c >= m) continue;
            ap[r + 1]++;
            ap[c + 1]++;
        }
    }
    for (r = 0; r < m; ++r) ap[r + 1] += ap[r];
    aj = (int *) malloc(sizeof(int) * (size_t) (ap[m] ? ap[m] : 1));
    fill = (long long *) malloc(sizeof(long long) * ((size_t) m + 1));
    if (!aj || !fill) { free(ap); free(aj); free(fill); return -1; }
    memcpy(fill, ap, sizeof(long long) * ((size_t) m + 1));
    for (r = 0; r < m; ++r) {
        int p;
        for (p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const int c = colidx[p];
            if (c == r || c < 0 || c >= m) continue;
            aj[fill[r]++] = c;
            aj[fill[c]++] = r;
        }
    }
    free(fill);
    *adj_ptr = ap;
    *adj = aj;
    return 0;
}

/* BFS from `start` over unvisited-in-`mark` vertices; writes the level structure into order[lo..)
 * and returns the number of vertices reached; *last = a vertex of the last level with least degree */
static int bfs_levels(int start, const long long *ap, const int *aj, int *mark, int stamp, int *queue, int *last, int *depth)
{
    int head = 0, tail = 0, level_end = 1, levels = 0, best = start;
    queue[tail++] = start;
    mark[start] = stamp;
    while (head < tail) {
        const int v = queue[head++];
        long long p;
        for (p = ap[v]; p < ap[v + 1]; ++p) {
            const int w = aj[p];
            if (mark[w] != stamp) { mark[w] = stamp; queue[tail++] = w; }
        }
        if (head == level_end && head < tail) { /* a new level starts at head */
            int i, bd = INT_MAX;
            ++levels;
            for (i = head; i < tail; ++i) {
                const int dg = (int) (ap[queue[i] + 1] - ap[queue[i]]);
                if (dg < bd) { bd = dg; best = queue[i]; }
            }
            level_end = tail;
        }
    }
    *last = best;
    *depth = levels;
    return tail;
}

int spmv_rcm_order(int m, const int *rowptr, const int *colidx, int *perm)
{
    long long *ap = NULL;
    int *aj = NULL, *mark = NULL, *queue = NULL, *done = NULL;
    dv_t *nb = NULL;
    int out = 0, v, stamp = 0, maxdeg = 0;
    if (m <= 0) return 0;
    if (build_adjacency(m, rowptr, colidx, &ap, &aj)) return -1;
    mark = (int *) calloc((size_t) m, sizeof(int));
    queue = (int *) malloc(sizeof(int) * (size_t) m);
    done = (int *) calloc((size_t) m, sizeof(int));
    for (v = 0; v < m; ++v) if (ap[v + 1] - ap[v] > maxdeg) maxdeg = (int) (ap[v + 1] - ap[v]);
    nb = (dv_t *) malloc(sizeof(dv_t) * (size_t) (maxdeg ? maxdeg : 1));
    if (!mark || !queue || !done || !nb) { free(ap); free(aj); free(mark); free(queue); free(done); free(nb); return -1; }

    for (v = 0; v < m; ++v) {
        int start, far, next, depth, depth2, tries, head, tail;
        if (done[v]) continue;
        /* pseudo-peripheral start vertex of v's component (George-Liu): BFS from v, then from a
         * least-degree vertex of the last level, for as long as the level structure gets deeper */
        (void) bfs_levels(v, ap, aj, mark, ++stamp, queue, &far, &depth);
        for (tries = 0;; ++tries) {
            (void) bfs_levels(far, ap, aj, mark, ++stamp, queue, &next, &depth2);
            if (depth2 > depth && tries < 4) { far = next; depth = depth2; }
            else break;
        }
        start = far;
        /* Cuthill-McKee BFS: neighbours appended by increasing degree */
        head = tail = out;
        perm[tail++] = start;
        done[start] = 1;
        while (head < tail) {
            const int u = perm[head++];
            long long p;
            int k = 0, i;
            for (p = ap[u]; p < ap[u + 1]; ++p) {
                const int w = aj[p];
                if (!done[w]) { done[w] = 1; nb[k].v = w; nb[k].deg = (int) (ap[w + 1] - ap[w]); ++k; }
            }
            if (k > 1) qsort(nb, (size_t) k, sizeof(dv_t), cmp_dv);
            for (i = 0; i < k; ++i) perm[tail++] = nb[i].v;
        }
        out = tail;
    }
    /* reverse */
    for (v = 0; v < m / 2; ++v) { const int t = perm[v]; perm[v] = perm[m - 1 - v]; perm[m - 1 - v] = t; }
    free(ap); free(aj); free(mark); free(queue); free(done); free(nb);
    return 0;
}

int spmv_permute_csr(int m, const int *rowptr, const int *colidx, const void *val, size_t value_size,
                     const int *perm, int **rowptr_out, int **colidx_out, void **val_out)
{
    const size_t vs = value_size == sizeof(double) ? sizeof(double) : sizeof(float);
    const long long nnz = m > 0 ? rowptr[m] : 0;
    int *inv = (int *) malloc(sizeof(int) * (size_t) (m ? m : 1));
    int *rp = (int *) malloc(sizeof(int) * ((size_t) m + 1));
    int *ci = (int *) malloc(sizeof(int) * (size_t) (nnz ? nnz : 1));
    char *vv = (char *) malloc(vs * (size_t) (nnz ? nnz : 1));
    int i;
    if (!inv || !rp || !ci || !vv) { free(inv); free(rp); free(ci); free(vv); return -1; }
    for (i = 0; i < m; ++i) inv[perm[i]] = i;
    rp[0] = 0;
    for (i = 0; i < m; ++i) rp[i + 1] = rp[i] + (rowptr[perm[i] + 1] - rowptr[perm[i]]);
    for (i = 0; i < m; ++i) {
        const int src = rowptr[perm[i]], len = rowptr[perm[i] + 1] - src, dst = rp[i];
        int k;
        for (k = 0; k < len; ++k) {
            const int c = colidx[src + k];
            ci[dst + k] = (c >= 0 && c < m) ? inv[c] : c;
        }
        memcpy(vv + vs * (size_t) dst, (const char *) val + vs * (size_t) src, vs * (size_t) len);
    }
    free(inv);
    *rowptr_out = rp;
    *colidx_out = ci;
    *val_out = vv;
    return 0;
}

long long spmv_csr_bandwidth(int m, const int *rowptr, const int *colidx)
{
    long long bw = 0;
    int r, p;
    for (r = 0; r < m; ++r)
        for (p = rowptr[r]; p < rowptr[r + 1]; ++p) {
            const long long d = colidx[p] > r ? (long long) colidx[p] - r : (long long) r - colidx[p];
            if (d > bw) bw = d;
        }
    return bw;
}
