/*
 * spmv_io.h -- Matrix Market loader and the reference's binary CSR cache.
 * Host-side only; nothing here touches the GPU.
 *
 * Replaces, with the same observable behaviour, the sample-side helpers of the reference:
 *   mmio_allinone        src/samples/mmio_highlevel.h:325-491   .mtx (coordinate) -> CSR
 *   mmio_save_as_bin     src/samples/mmio_highlevel.h:531-553   CSR -> mtx_cache/<name>.bin
 *   mmio_read_from_bin   src/samples/mmio_highlevel.h:554-584   mtx_cache/<name>.bin -> CSR
 * (the NIST parser they sit on, src/samples/mmio.h, is re-implemented here, not copied).
 *
 * Conventions kept from the reference:
 *   - "coordinate" files only; field real / integer / pattern / complex (real part kept; pattern
 *     entries get value 1); symmetry general / symmetric / hermitian / skew-symmetric, where
 *     symmetric and hermitian files are expanded (each off-diagonal entry also stored mirrored,
 *     value copied unchanged) and skew-symmetric is read as stored (mmio_highlevel.h:362-366);
 *   - entries keep file order inside a row (no sorting, duplicates kept);
 *   - cache file name: "mtx_cache/" + path with every '/', '\\' and ' ' replaced by '_' + ".bin";
 *     layout: int32 m, n, nnz; int32 rowptr[m+1]; int32 colidx[nnz]; value[nnz] (float or double
 *     as compiled into the reference, chosen here by value_size); the directory must already
 *     exist (the reference fails silently otherwise; here the return code says so).
 */
#include <stddef.h>
#if defined(__cplusplus)
extern "C" {
#endif
#ifndef SPMV_HIP_IO_H
#define SPMV_HIP_IO_H

enum {
    SPMV_IO_OK = 0,
    SPMV_IO_E_OPEN = -1,      /* cannot open the file */
    SPMV_IO_E_BANNER = -2,    /* not a MatrixMarket coordinate file this loader supports */
    SPMV_IO_E_SIZE = -4,      /* bad size line (the reference returns -4 too) */
    SPMV_IO_E_DATA = -5,      /* malformed or out-of-range entry */
    SPMV_IO_E_NOMEM = -6,
    SPMV_IO_E_RANGE = -7      /* expanded nnz does not fit int32 */
};

/* .mtx -> CSR.  value_size selects double (8) or float (anything else).  The three arrays are
 * malloc'ed (64-byte aligned) and owned by the caller: release them with spmv_io_free(). */
int spmv_io_read_mtx(const char *path, size_t value_size, int *m, int *n, int *nnz, int *is_symmetric,
                     int **rowptr, int **colidx, void **val);

/* Name of the cache file the reference would use for `mtx_path`; returns 0, or -1 if cap is too small. */
int spmv_io_cache_path(const char *mtx_path, char *out, size_t cap);

/* Write / read the binary cache at an explicit path (spmv_io_cache_path gives the reference's). */
int spmv_io_write_bin(const char *bin_path, int m, int n, int nnz, const int *rowptr, const int *colidx,
                      const void *val, size_t value_size);
int spmv_io_read_bin(const char *bin_path, size_t value_size, int *m, int *n, int *nnz,
                     int **rowptr, int **colidx, void **val);

/* The harness' loading rule (test_spmv.c:166-185): try the cache, else parse the .mtx and write
 * the cache (best effort).  *from_cache tells which happened. */
int spmv_io_load(const char *mtx_path, size_t value_size, int *m, int *n, int *nnz, int *is_symmetric,
                 int **rowptr, int **colidx, void **val, int *from_cache);

void spmv_io_free(void *p);

#endif
#if defined(__cplusplus)
}
#endif
