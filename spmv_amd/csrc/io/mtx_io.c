/*
 * mtx_io.c -- Matrix Market coordinate reader + the reference's binary CSR cache (include/spmv_io.h).
 * Plain C11, host only.  Behaviour follows src/samples/mmio_highlevel.h:325-584 of the reference
 * (see the header for the conventions kept); the implementation is new: the file is read in one
 * piece and tokenised with strtol/strtod instead of one fscanf per entry.
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <string.h>

#include "spmv_io.h"

static void *alloc64(size_t bytes)
{
    void *p = NULL;
    if (bytes == 0) bytes = 64;
    if (posix_memalign(&p, 64, (bytes + 63) & ~(size_t) 63)) return NULL;
    return p;
}

void spmv_io_free(void *p) { free(p); }

static int word_is(const char *w, const char *lit)
{
    while (*w && *lit) {
        if (tolower((unsigned char) *w) != *lit) return 0;
        ++w; ++lit;
    }
    return *lit == 0 && (*w == 0 || isspace((unsigned char) *w));
}

static const char *next_word(const char *p)
{
    while (*p && !isspace((unsigned char) *p)) ++p;
    while (*p == ' ' || *p == '\t') ++p;
    return p;
}

static void store_val(void *val, size_t vsize, long long i, double v)
{
    if (vsize == sizeof(double)) ((double *) val)[i] = v;
    else ((float *) val)[i] = (float) v;
}

int spmv_io_read_mtx(const char *path, size_t value_size, int *m, int *n, int *nnz, int *is_symmetric,
                     int **rowptr, int **colidx, void **val)
{
    FILE *f = fopen(path, "rb");
    char *buf = NULL, *p, *end;
    long fsize;
    int rc = SPMV_IO_OK, field_pattern = 0, field_complex = 0, field_integer = 0, symmetric = 0;
    long long rows, cols, entries, i, total;
    int *ri = NULL, *ci = NULL, *rp = NULL, *fill = NULL, *out_ci = NULL;
    double *vv = NULL;
    void *out_v = NULL;
    const size_t vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float);

    if (!f) return SPMV_IO_E_OPEN;
    if (fseek(f, 0, SEEK_END) || (fsize = ftell(f)) < 0 || fseek(f, 0, SEEK_SET)) { fclose(f); return SPMV_IO_E_OPEN; }
    buf = (char *) malloc((size_t) fsize + 1);
    if (!buf) { fclose(f); return SPMV_IO_E_NOMEM; }
    if (fread(buf, 1, (size_t) fsize, f) != (size_t) fsize) { fclose(f); free(buf); return SPMV_IO_E_OPEN; }
    fclose(f);
    buf[fsize] = 0;
    end = buf + fsize;

    /* banner: %%MatrixMarket matrix coordinate <field> <symmetry> */
    p = buf;
    if (strncmp(p, "%%MatrixMarket", 14) != 0) { rc = SPMV_IO_E_BANNER; goto done; }
    {
        const char *w = next_word(p);                 /* object */
        if (!word_is(w, "matrix")) { rc = SPMV_IO_E_BANNER; goto done; }
        w = next_word(w);                             /* format */
        if (!word_is(w, "coordinate")) { rc = SPMV_IO_E_BANNER; goto done; }
        w = next_word(w);                             /* field */
        if (word_is(w, "real")) { }
        else if (word_is(w, "double")) { }
        else if (word_is(w, "integer")) field_integer = 1;
        else if (word_is(w, "pattern")) field_pattern = 1;
        else if (word_is(w, "complex")) field_complex = 1;
        else { rc = SPMV_IO_E_BANNER; goto done; }
        w = next_word(w);                             /* symmetry */
        if (word_is(w, "general")) symmetric = 0;
        else if (word_is(w, "symmetric") || word_is(w, "hermitian")) symmetric = 1;
        else if (word_is(w, "skew-symmetric")) symmetric = 0; /* read as stored, like the reference */
        else { rc = SPMV_IO_E_BANNER; goto done; }
    }
    (void) field_integer;
    /* skip the banner line and every comment / blank line */
    while (p < end) {
        char *line = p;
        while (p < end && *p != '\n') ++p;
        if (p < end) ++p;
        while (*line == ' ' || *line == '\t') ++line;
        if (*line == '%' || *line == '\n' || *line == '\r' || line >= p) continue;
        p = line;
        break;
    }
    {
        char *q;
        rows = strtoll(p, &q, 10); if (q == p) { rc = SPMV_IO_E_SIZE; goto done; } p = q;
        cols = strtoll(p, &q, 10); if (q == p) { rc = SPMV_IO_E_SIZE; goto done; } p = q;
        entries = strtoll(p, &q, 10); if (q == p) { rc = SPMV_IO_E_SIZE; goto done; } p = q;
    }
    if (rows < 0 || cols < 0 || entries < 0 || rows > INT_MAX - 1 || cols > INT_MAX || entries > INT_MAX) { rc = SPMV_IO_E_SIZE; goto done; }

    ri = (int *) malloc(sizeof(int) * (size_t) (entries ? entries : 1));
    ci = (int *) malloc(sizeof(int) * (size_t) (entries ? entries : 1));
    vv = (double *) malloc(sizeof(double) * (size_t) (entries ? entries : 1));
    rp = (int *) alloc64(sizeof(int) * ((size_t) rows + 1));
    fill = (int *) calloc((size_t) rows + 1, sizeof(int));
    if (!ri || !ci || !vv || !rp || !fill) { rc = SPMV_IO_E_NOMEM; goto done; }

    {
        long long *cnt = (long long *) calloc((size_t) rows + 1, sizeof(long long));
        if (!cnt) { rc = SPMV_IO_E_NOMEM; goto done; }
        for (i = 0; i < entries; ++i) {
            char *q;
            long long a, b;
            double v = 1.0;
            a = strtoll(p, &q, 10); if (q == p) { rc = SPMV_IO_E_DATA; free(cnt); goto done; } p = q;
            b = strtoll(p, &q, 10); if (q == p) { rc = SPMV_IO_E_DATA; free(cnt); goto done; } p = q;
            if (!field_pattern) {
                v = strtod(p, &q); if (q == p) { rc = SPMV_IO_E_DATA; free(cnt); goto done; } p = q;
                if (field_complex) { (void) strtod(p, &q); if (q == p) { rc = SPMV_IO_E_DATA; free(cnt); goto done; } p = q; }
            }
            if (a < 1 || a > rows || b < 1 || b > cols) { rc = SPMV_IO_E_DATA; free(cnt); goto done; }
            ri[i] = (int) (a - 1);
            ci[i] = (int) (b - 1);
            vv[i] = v;
            cnt[a - 1]++;
            if (symmetric && a != b) {
                if (b > rows) { rc = SPMV_IO_E_DATA; free(cnt); goto done; }
                cnt[b - 1]++;
            }
        }
        total = 0;
        for (i = 0; i < rows; ++i) {
            if (total > INT_MAX) break;
            rp[i] = (int) total;
            total += cnt[i];
        }
        free(cnt);
        if (total > INT_MAX) { rc = SPMV_IO_E_RANGE; goto done; }
        rp[rows] = (int) total;
    }
    out_ci = (int *) alloc64(sizeof(int) * (size_t) total);
    out_v = alloc64(vsize * (size_t) total);
    if (!out_ci || !out_v) { rc = SPMV_IO_E_NOMEM; goto done; }
    /* scatter in file order; a symmetric off-diagonal entry is followed at once by its mirror
     * (same interleaving as mmio_highlevel.h:441-470) */
    for (i = 0; i < entries; ++i) {
        long long o = (long long) rp[ri[i]] + fill[ri[i]]++;
        out_ci[o] = ci[i];
        store_val(out_v, vsize, o, vv[i]);
        if (symmetric && ri[i] != ci[i]) {
            o = (long long) rp[ci[i]] + fill[ci[i]]++;
            out_ci[o] = ri[i];
            store_val(out_v, vsize, o, vv[i]);
        }
    }
    *m = (int) rows;
    *n = (int) cols;
    *nnz = (int) total;
    if (is_symmetric) *is_symmetric = symmetric;
    *rowptr = rp; rp = NULL;
    *colidx = out_ci; out_ci = NULL;
    *val = out_v; out_v = NULL;
done:
    free(buf); free(ri); free(ci); free(vv); free(fill); free(rp); free(out_ci); free(out_v);
    return rc;
}

int spmv_io_cache_path(const char *mtx_path, char *out, size_t cap)
{
    static const char pre[] = "mtx_cache/", suf[] = ".bin";
    const size_t len = strlen(mtx_path);
    size_t i;
    if (sizeof pre - 1 + len + sizeof suf > cap) return -1;
    memcpy(out, pre, sizeof pre - 1);
    for (i = 0; i < len; ++i) {
        const char ch = mtx_path[i];
        out[sizeof pre - 1 + i] = (ch == '/' || ch == '\\' || ch == ' ') ? '_' : ch;
    }
    memcpy(out + sizeof pre - 1 + len, suf, sizeof suf);
    return 0;
}

int spmv_io_write_bin(const char *bin_path, int m, int n, int nnz, const int *rowptr, const int *colidx,
                      const void *val, size_t value_size)
{
    const size_t vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float);
    FILE *f = fopen(bin_path, "wb");
    int ok;
    int32_t hdr[3];
    if (!f) return SPMV_IO_E_OPEN;
    hdr[0] = m; hdr[1] = n; hdr[2] = nnz;
    ok = fwrite(hdr, sizeof(int32_t), 3, f) == 3 &&
         fwrite(rowptr, sizeof(int32_t), (size_t) m + 1, f) == (size_t) m + 1 &&
         fwrite(colidx, sizeof(int32_t), (size_t) nnz, f) == (size_t) nnz &&
         fwrite(val, vsize, (size_t) nnz, f) == (size_t) nnz;
    if (fclose(f)) ok = 0;
    return ok ? SPMV_IO_OK : SPMV_IO_E_DATA;
}

int spmv_io_read_bin(const char *bin_path, size_t value_size, int *m, int *n, int *nnz,
                     int **rowptr, int **colidx, void **val)
{
    const size_t vsize = value_size == sizeof(double) ? sizeof(double) : sizeof(float);
    FILE *f = fopen(bin_path, "rb");
    int32_t hdr[3];
    int *rp = NULL, *ci = NULL;
    void *v = NULL;
    int rc = SPMV_IO_OK;
    if (!f) return SPMV_IO_E_OPEN;
    if (fread(hdr, sizeof(int32_t), 3, f) != 3 || hdr[0] < 0 || hdr[1] < 0 || hdr[2] < 0) { fclose(f); return SPMV_IO_E_SIZE; }
    rp = (int *) alloc64(sizeof(int) * ((size_t) hdr[0] + 1));
    ci = (int *) alloc64(sizeof(int) * (size_t) hdr[2]);
    v = alloc64(vsize * (size_t) hdr[2]);
    if (!rp || !ci || !v) rc = SPMV_IO_E_NOMEM;
    else if (fread(rp, sizeof(int32_t), (size_t) hdr[0] + 1, f) != (size_t) hdr[0] + 1 ||
             fread(ci, sizeof(int32_t), (size_t) hdr[2], f) != (size_t) hdr[2] ||
             fread(v, vsize, (size_t) hdr[2], f) != (size_t) hdr[2] || rp[hdr[0]] != hdr[2])
        rc = SPMV_IO_E_DATA;
    /* The format does not record the value width (nor does the reference's, mmio_highlevel.h:531-584), so a
     * cache written as fp64 would read back "successfully" as fp32 -- the first half of the doubles taken
     * as floats.  The file must end exactly after nnz values of the width asked for; otherwise the caller
     * (spmv_io_load) re-parses the .mtx file and rewrites the cache in its own width. */
    else if (hdr[2] > 0 && fgetc(f) != EOF)
        rc = SPMV_IO_E_DATA;
    fclose(f);
    if (rc) { free(rp); free(ci); free(v); return rc; }
    *m = hdr[0]; *n = hdr[1]; *nnz = hdr[2];
    *rowptr = rp; *colidx = ci; *val = v;
    return SPMV_IO_OK;
}

int spmv_io_load(const char *mtx_path, size_t value_size, int *m, int *n, int *nnz, int *is_symmetric,
                 int **rowptr, int **colidx, void **val, int *from_cache)
{
    char bin[2048];
    int rc;
    if (from_cache) *from_cache = 0;
    if (spmv_io_cache_path(mtx_path, bin, sizeof bin) == 0 &&
        spmv_io_read_bin(bin, value_size, m, n, nnz, rowptr, colidx, val) == SPMV_IO_OK) {
        if (from_cache) *from_cache = 1;
        if (is_symmetric) *is_symmetric = 0; /* the cache does not record it (nor does the reference's) */
        return SPMV_IO_OK;
    }
    rc = spmv_io_read_mtx(mtx_path, value_size, m, n, nnz, is_symmetric, rowptr, colidx, val);
    if (rc == SPMV_IO_OK && spmv_io_cache_path(mtx_path, bin, sizeof bin) == 0) {
        (void) mkdir("mtx_cache", 0777); /* the reference expects the directory to exist; create it if we can */
        (void) spmv_io_write_bin(bin, *m, *n, *nnz, *rowptr, *colidx, *val, value_size); /* best effort */
    }
    return rc;
}
