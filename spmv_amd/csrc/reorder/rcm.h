/* rcm.h -- host-side symmetric reordering for x locality, plain C. */
#ifndef SPMV_RCM_H
#define SPMV_RCM_H
#include <stddef.h>

/* Reverse Cuthill-McKee order of the graph of A + A^T (m x m, CSR pattern).  perm[new] = old row.
 * Returns 0, or -1 when memory runs out. */
int spmv_rcm_order(int m, const int *rowptr, const int *colidx, int *perm);

/* B = P A P^T: row `new` of B is row perm[new] of A with every column c renamed to inv[c].
 * Entries keep their order inside a row.  Arrays are malloc'ed; free() them.  Returns 0 / -1. */
int spmv_permute_csr(int m, const int *rowptr, const int *colidx, const void *val, size_t value_size,
                     const int *perm, int **rowptr_out, int **colidx_out, void **val_out);

/* Half bandwidth max |i - j| over the entries (what RCM tries to shrink). */
long long spmv_csr_bandwidth(int m, const int *rowptr, const int *colidx);
#endif
