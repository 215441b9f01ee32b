/*
 * spmv.h -- the drop-in C API (replaces the reference's include/spmv.h:19-71).
 *
 * Four functions, same names (including the "destory" spelling, which is ABI), same argument
 * lists, same void returns.  A program written against the reference recompiles against this
 * header unchanged and links libspmv_hip.so in place of libmv_l2.so.
 *
 * Differences a caller can observe:
 *   - <omp.h> and <immintrin.h> are pulled in when the compiler has them, as the reference's header does
 *     (spmv.h:9-10): its own sample calls omp_set_num_threads() and uses intrinsics with only <spmv.h>
 *     included (test_spmv.c:88), so it compiles against this header without extra flags.  Define
 *     SPMV_HIP_NO_COMPAT_INCLUDES to keep them out;
 *   - X and Y (and the CSR arrays) may be HOST or DEVICE pointers; the library classifies each
 *     pointer (hipPointerGetAttributes).  Host vectors are staged through HBM (correct, PCIe
 *     bound); device vectors are used in place.  See spmv_hip.h for stream control.
 *   - Y is always fully overwritten, empty rows included (Method_Serial semantics,
 *     serial_spmv.c:16-21); the reference's other methods leave some rows unwritten
 *      -- those defects are not reproduced.
 *   - failures (no GPU, out of memory, bad arguments) never change a signature: they are
 *     reported on stderr and through spmv_hip_last_error() (spmv_hip.h).
 */
#include "spmv_Defines.h"
#if defined(__cplusplus)
extern "C" {
#endif
#ifndef SPMV_HIP_SPMV_H
#define SPMV_HIP_SPMV_H

#if !defined(SPMV_HIP_NO_COMPAT_INCLUDES) && defined(__has_include)
#if __has_include(<omp.h>)
#include <omp.h>
#endif
#if (defined(__x86_64__) || defined(__i386__)) && __has_include(<immintrin.h>)
#include <immintrin.h>
#endif
#endif

#define ALIGENED_SIZE 64 /* reference spmv.h:12; callers use it for aligned_alloc */

/* Free everything the handle owns (device buffers, inspector products) and the handle itself.
 * NULL is a no-op.  Replaces common.c:54-61. */
void spmv_destory_handle(spmv_Handle_t this_handle);

/* Release what the handle owns and return it to the freshly-initialised state
 * (Method_Serial, no matrix).  The handle stays allocated.  Replaces common.c:69-71. */
void spmv_clear_handle(spmv_Handle_t this_handle);

/*
 * Inspector: allocate *Handle, copy the CSR matrix into HBM and build what the schedule needs.
 * Replaces common.c:123-190.
 *
 *   m, n          rows / columns
 *   RowPtr        m+1 entries, 0-based, RowPtr[0] == 0      (host or device)
 *   ColIdx        RowPtr[m] entries in [0, n)                (host or device)
 *   Matrix_Val    RowPtr[m] values, double if size == 8 else float (host or device)
 *   nthreads      stored; not used by the GPU schedules
 *   Function      schedule; out-of-range values select Method_Serial (common.c:136)
 *   size          sizeof(double) or sizeof(float)
 *   vectorizedWay stored; every value runs the HIP back end
 *   MtxToken      optional label, may be NULL (the reference uses it for METIS cache files only)
 *
 * The caller's arrays are not modified and are not needed after the call returns.
 * On failure *Handle is still a valid handle whose spmv() is a reported no-op.
 */
void spmv_create_handle_all_in_one(spmv_Handle_t *Handle,
                                   BASIC_INT_TYPE m,
                                   BASIC_INT_TYPE n,
                                   BASIC_INT_TYPE *RowPtr,
                                   BASIC_INT_TYPE *ColIdx,
                                   void *Matrix_Val,
                                   BASIC_SIZE_TYPE nthreads,
                                   SPMV_METHODS Function,
                                   BASIC_SIZE_TYPE size,
                                   VECTORIZED_WAY vectorizedWay,
                                   const char *MtxToken);

/*
 * Executor: Y = A * X.  Replaces common.c:278-304.
 *
 *   handle        from spmv_create_handle_all_in_one; NULL is a no-op (common.c:285)
 *   m, RowPtr, ColIdx, Matrix_Val
 *                 the reference re-reads these on every call.  Here: if they are the pointers
 *                 (and m) seen at create, the HBM-resident copy is used; if they differ the
 *                 matrix is re-inspected first (correct, slow -- see DESIGN.md "CSR arguments").
 *                 VALUES CHANGED IN PLACE behind an unchanged Matrix_Val pointer are not seen: call
 *                 spmv_hip_update_values(handle, Matrix_Val) (spmv_hip.h) after changing them, or run with
 *                 SPMV_HIP_CHECK_VALUES=1, which checksums Matrix_Val on every call and refreshes by itself.
 *   Vector_Val_X  n values  (host or device)
 *   Vector_Val_Y  m values, all overwritten (host or device)
 *
 * Synchronous unless a stream was attached with spmv_hip_set_stream().
 */
void spmv(const spmv_Handle_t handle,
          BASIC_INT_TYPE m,
          const BASIC_INT_TYPE *RowPtr,
          const BASIC_INT_TYPE *ColIdx,
          const void *Matrix_Val,
          const void *Vector_Val_X,
          void *Vector_Val_Y);

#endif /* SPMV_HIP_SPMV_H */

#if defined(__cplusplus)
}
#endif
