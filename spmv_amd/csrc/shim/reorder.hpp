// shim/reorder.hpp -- part of spmv_shim.hip: reverse Cuthill-McKee of the RESIDENT matrix on the device (kernels/rcm.hpp) and the permuted
// matrix P A P^T in its place.  Called by spmv_api.c for option "reorder" on square matrices; perm (host, m ints) is what handle->index publishes.
#pragma once

// returns SPMV_HIP_OK and leaves d holding P A P^T (statistics recomputed, nothing built yet), or an error code with d untouched
extern "C" int spmv_shim_reorder_rcm(spmv_dev *d, int *perm_host)
{
    if (!d || !perm_host) return fail(SPMV_HIP_E_ARG, "reorder: NULL");
    if (d->m != d->n || d->m < 2) return fail(SPMV_HIP_E_ARG, "reorder: needs a square matrix");
    DeviceGuard guard(d->device);
    if (!guard.ok) return fail(SPMV_HIP_E_RUNTIME, "hipSetDevice(%d) failed", d->device);
    const int m = d->m;
    const long long nnz = d->nnz;
    const bool f64 = d->vsize == sizeof(double);
    hipStream_t st = d->stream;
    struct Scratch { // returned to the pool on every path out of this function
        std::vector<void *> blocks;
        ~Scratch() { for (void *p : blocks) (void) pool_free(p); }
    } tmp;
    auto scratch = [&](void **p, size_t bytes) { const hipError_t e = pool_malloc(p, bytes ? bytes : 16); if (e == hipSuccess) tmp.blocks.push_back(*p); return e; };
    auto done = [&](int rc) { return rc; };
    auto scan = [&](const int *in, int *out, long long n, int *sums, int *total) { // exclusive scan, out[n] = total (csr5.hpp / split.hpp kernels)
        const int nb = (int) ((n + kScanTile - 1) / kScanTile);
        scan_block_sums_kernel<<<nb, kBlock, 0, st>>>(n, in, sums);
        scan_sums_inplace_kernel<<<1, kBlock, 0, st>>>(nb, sums, total);
        scan_apply_kernel<<<nb, kBlock, 0, st>>>(n, in, sums, out, nullptr, nullptr);
        (void) hipMemcpyAsync(out + n, total, sizeof(int), hipMemcpyDeviceToDevice, st);
    };
    const int nb = (int) (((long long) m + kScanTile - 1) / kScanTile);
    int *cnt = nullptr, *tptr = nullptr, *cursor = nullptr, *trow = nullptr, *level = nullptr, *q0 = nullptr, *q1 = nullptr, *sums = nullptr, *total = nullptr, *wide_cnt = nullptr;
    unsigned long long *best = nullptr;
    RcmState *state = nullptr;
    if (scratch((void **) &cnt, sizeof(int) * ((size_t) m + 1)) != hipSuccess || scratch((void **) &tptr, sizeof(int) * ((size_t) m + 1)) != hipSuccess ||
        scratch((void **) &cursor, sizeof(int) * ((size_t) m + 1)) != hipSuccess || scratch((void **) &trow, sizeof(int) * (size_t) (nnz ? nnz : 1)) != hipSuccess ||
        scratch((void **) &level, sizeof(int) * (size_t) m) != hipSuccess || scratch((void **) &q0, sizeof(int) * (size_t) m) != hipSuccess ||
        scratch((void **) &q1, sizeof(int) * (size_t) m) != hipSuccess || scratch((void **) &sums, sizeof(int) * ((size_t) nb + 1)) != hipSuccess ||
        scratch((void **) &total, sizeof(int)) != hipSuccess || scratch((void **) &wide_cnt, sizeof(int)) != hipSuccess ||
        scratch((void **) &best, sizeof(unsigned long long)) != hipSuccess || scratch((void **) &state, sizeof(RcmState)) != hipSuccess) {
        (void) hipGetLastError();
        return done(fail(SPMV_HIP_E_ALLOC, "reorder: scratch"));
    }
    const int g_m = grid_for(m, kBlock, d->cus * 16), g_nnz = grid_for(nnz, kBlock * 4, d->cus * 16), g_rows = grid_for(m, kBlock / kWave, d->cus * 32);
    // A^T's pattern
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t) m + 1), st));
    rcm_col_count_kernel<<<g_nnz, kBlock, 0, st>>>(nnz, m, d->colidx, cnt);
    scan(cnt, tptr, m, sums, total);
    HIP_TRY(hipMemcpyAsync(cursor, tptr, sizeof(int) * ((size_t) m + 1), hipMemcpyDeviceToDevice, st));
    rcm_transpose_fill_kernel<<<g_rows, kBlock, 0, st>>>(m, d->rowptr, d->colidx, cursor, trow);
    HIP_TRY(hipMemsetAsync(level, 0xff, sizeof(int) * (size_t) m, st)); // -1: unvisited
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void) hipGetLastError(); return done(fail(SPMV_HIP_E_RUNTIME, "reorder: transpose failed")); }

    auto pick = [&](bool last_level, int last, unsigned long long *out) -> int { // least (degree, id) among the unvisited / the last level
        const unsigned long long none = ~0ull;
        HIP_TRY(hipMemcpyAsync(best, &none, sizeof none, hipMemcpyHostToDevice, st));
        if (last_level) rcm_last_level_kernel<<<g_m, kBlock, 0, st>>>(m, last, d->rowptr, tptr, level, best);
        else rcm_min_degree_kernel<<<g_m, kBlock, 0, st>>>(m, d->rowptr, tptr, level, best);
        HIP_TRY(hipMemcpyAsync(out, best, sizeof none, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return SPMV_HIP_OK;
    };
    // BFS from `root` at level `lvl0`; returns the state after the component is exhausted (h.count == 0)
    auto bfs = [&](int root, int lvl0, int visited0, RcmState *h) -> int {
        *h = RcmState{1, lvl0, 0, visited0 + 1};
        HIP_TRY(hipMemcpyAsync(level + root, &lvl0, sizeof(int), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(q0, &root, sizeof(int), hipMemcpyHostToDevice, st));
        while (h->count > 0) {
            if (h->count <= kRcmSmall) { // one workgroup walks the levels until the frontier is empty or outgrows it
                HIP_TRY(hipMemcpyAsync(state, h, sizeof *h, hipMemcpyHostToDevice, st));
                rcm_bfs_small_kernel<<<1, kRcmThreads, 0, st>>>(m, d->rowptr, d->colidx, tptr, trow, level, q0, q1, state);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(h, state, sizeof *h, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            } else { // the whole grid, one level
                int next = 0;
                HIP_TRY(hipMemsetAsync(wide_cnt, 0, sizeof(int), st));
                rcm_bfs_wide_kernel<<<grid_for(h->count, kBlock / kWave, d->cus * 32), kBlock, 0, st>>>(m, h->count, h->lvl, d->rowptr, d->colidx, tptr, trow, level,
                                                                                                      h->cur ? q1 : q0, h->cur ? q0 : q1, wide_cnt);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(&next, wide_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                h->count = next;
                h->visited += next;
                h->lvl += 1;
                h->cur ^= 1;
            }
        }
        return SPMV_HIP_OK;
    };

    RcmState h{0, 0, 0, 0};
    int visited = 0, lvl = 0, roots = 0, rc = SPMV_HIP_OK;

    constexpr int kMaxRoots = 1024; // components started one by one; what is left after that many is ordered by (degree, id) only
    while (visited < m && !rc) {
        unsigned long long b = 0;
        if ((rc = pick(false, 0, &b))) break;
        if (b == ~0ull) break;
        const int root = (int) (b & 0xffffffffull);
        const bool isolated = (b >> 32) == 0;
        if (isolated || roots >= kMaxRoots) { // every isolated vertex at once; or: the component budget is spent
            int got = 0;
            HIP_TRY(hipMemsetAsync(wide_cnt, 0, sizeof(int), st));
            rcm_claim_rest_kernel<<<g_m, kBlock, 0, st>>>(m, level, lvl, isolated && roots < kMaxRoots ? 1 : 0, d->rowptr, tptr, wide_cnt);
            HIP_TRY(hipMemcpyAsync(&got, wide_cnt, sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            visited += got;
            lvl += 1;
            if (got == 0) break;
            continue;
        }
        if ((rc = bfs(root, lvl, visited, &h))) break;
        if (roots == 0 && h.lvl - 1 > lvl) { // George-Liu, one round, for the first (usually the only) component: restart from the far end
            unsigned long long far = 0;
            if ((rc = pick(true, h.lvl - 1, &far))) break;
            if (far != ~0ull && (int) (far & 0xffffffffull) != root) {
                rcm_fill_level_kernel<<<g_m, kBlock, 0, st>>>(m, level, lvl, -1); // forget THIS component's first sweep (earlier components keep their levels)
                if ((rc = bfs((int) (far & 0xffffffffull), lvl, visited, &h))) break;
            }
        }
        visited = h.visited;
        lvl = h.lvl; // the next component's first level
        ++roots;
    }
    if (rc) return done(rc);
    if (visited != m) return done(fail(SPMV_HIP_E_RUNTIME, "reorder: BFS reached %d of %d vertices", visited, m));

    // sort by (level, degree, id); reverse; permute
    long long n2 = 1;
    while (n2 < m) n2 <<= 1;
    unsigned long long *key = nullptr;
    unsigned *id = nullptr;
    int *perm = nullptr, *inv = nullptr, *newlen = nullptr;
    if (scratch((void **) &key, sizeof(unsigned long long) * (size_t) n2) != hipSuccess || scratch((void **) &id, sizeof(unsigned) * (size_t) n2) != hipSuccess ||
        scratch((void **) &perm, sizeof(int) * (size_t) m) != hipSuccess || scratch((void **) &inv, sizeof(int) * (size_t) m) != hipSuccess ||
        scratch((void **) &newlen, sizeof(int) * ((size_t) m + 1)) != hipSuccess) {
        (void) hipGetLastError();
        return done(fail(SPMV_HIP_E_ALLOC, "reorder: sort scratch"));
    }
    const int g_n2 = grid_for(n2, kBlock * 2, d->cus * 16);
    rcm_keys_kernel<<<g_n2, kBlock, 0, st>>>(n2, m, d->rowptr, tptr, level, key, id);
    for (long long kk = 2; kk <= n2; kk <<= 1)
        for (long long jj = kk >> 1; jj > 0; jj >>= 1) rcm_bitonic_kernel<<<g_n2, kBlock, 0, st>>>(n2, kk, jj, key, id);
    rcm_perm_kernel<<<g_m, kBlock, 0, st>>>(m, id, d->rowptr, perm, inv, newlen);
    HIP_TRY(hipGetLastError());
    int *rp2 = nullptr, *ci2 = nullptr;
    void *va2 = nullptr;
    // the permuted matrix replaces the resident one (same sizes, same padding)
    if ((rc = dev_alloc(d, (void **) &rp2, sizeof(int) * ((size_t) m + 1), false)) || (rc = dev_alloc(d, (void **) &ci2, sizeof(int) * ((size_t) nnz + kStreamPad), false)) ||
        (rc = dev_alloc(d, &va2, d->vsize * ((size_t) nnz + kStreamPad), false))) {
        for (void *p : {(void *) rp2, (void *) ci2, va2}) if (p) (void) pool_free(p);
        return done(rc);
    }
    scan(newlen, rp2, m, sums, total);
    (void) hipMemsetAsync(ci2 + nnz, 0, sizeof(int) * kStreamPad, st);
    (void) hipMemsetAsync((char *) va2 + d->vsize * (size_t) nnz, 0, d->vsize * kStreamPad, st);
    if (f64) rcm_permute_kernel<double><<<g_rows, kBlock, 0, st>>>(m, perm, inv, d->rowptr, d->colidx, (const double *) d->val, rp2, ci2, (double *) va2);
    else rcm_permute_kernel<float><<<g_rows, kBlock, 0, st>>>(m, perm, inv, d->rowptr, d->colidx, (const float *) d->val, rp2, ci2, (float *) va2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(perm_host, perm, sizeof(int) * (size_t) m, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        for (void *p : {(void *) rp2, (void *) ci2, va2}) (void) pool_free(p);
        return done(fail(SPMV_HIP_E_RUNTIME, "reorder: permute: %s", hipGetErrorString(e)));
    }
    const long long old_bytes = (long long) (sizeof(int) * ((size_t) m + 1) + sizeof(int) * ((size_t) nnz + kStreamPad) + d->vsize * ((size_t) nnz + kStreamPad));
    (void) pool_free(d->rowptr); (void) pool_free(d->colidx); (void) pool_free(d->val);
    d->device_bytes -= old_bytes;
    d->rowptr = rp2; d->colidx = ci2; d->val = va2;
    return done(SPMV_HIP_OK);
}
