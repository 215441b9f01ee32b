/*
 * spmv_Defines.h -- types of the drop-in SpMV C API, MI355X (gfx950) build.
 *
 * This header replaces the reference's include/spmv_Defines.h:10-70.  Everything a
 * caller of the reference can observe is kept bit-compatible:
 *
 *   - BASIC_INT_TYPE / BASIC_SIZE_TYPE stay overridable macros (reference :10-16).  The
 *     library itself is built with the defaults (int / unsigned long); a caller that
 *     overrides them must rebuild the library with the same values, as with the reference.
 *   - VECTORIZED_WAY keeps NONE=0, AVX2=1, AVX512=2 (reference :18-23) and gains
 *     VECTOR_HIP=3 in front of VECTOR_TOTAL_SIZE (north_star: "VECTORIZED_WAY gaining a
 *     VECTOR_HIP entry").  The reference stores the field and never reads it again
 *     (common.c:80), so on this build every value selects the HIP schedules.
 *   - SPMV_METHODS keeps its numeric values (reference :26-36).
 *   - struct spmv_Handle keeps its field order and types (reference :44-68); callers read
 *     handle->index and handle->spmvMethod directly (test_spmv.c:95,130).  All device state
 *     lives behind extraHandle.
 */
#if defined(__cplusplus)
extern "C" {
#endif
#ifndef SPMV_HIP_DEFINES_H
#define SPMV_HIP_DEFINES_H

#ifndef BASIC_INT_TYPE
#define BASIC_INT_TYPE int
#endif

#ifndef BASIC_SIZE_TYPE
#define BASIC_SIZE_TYPE unsigned long
#endif

/* Which arithmetic back end a handle asks for.  Numeric values 0..2 are the reference's. */
typedef enum VECTORIZED_WAY {
    VECTOR_NONE = 0,
    VECTOR_AVX2 = 1,
    VECTOR_AVX512 = 2,
    VECTOR_HIP = 3,          /* new: hand-written gfx950 kernels (the only back end of this build) */
    VECTOR_TOTAL_SIZE        /* number of entries in Vectorized_names[] */
} VECTORIZED_WAY;

/* Indexed by VECTORIZED_WAY; has VECTOR_TOTAL_SIZE entries. */
extern const char *Vectorized_names[];

/*
 * Schedules.  On this build each one maps to a GPU schedule ("Method map"):
 *   Method_Serial        CSR-scalar  (one lane per row; debug / plumbing kernel)
 *   Method_Parallel      CSR-vector  (L lanes per row, wavefront shuffle reduction)
 *   Method_Balanced      equal-nnz row blocks, LDS-staged products (falls to nnz-split on long rows)
 *   Method_Balanced2     nnz-split with carry fix-up
 *   Method_Balanced_Yid  nnz-split with carry fix-up
 *   Method_SellCSigma    SELL-C-sigma, C = 64, sigma = 1024
 *   Method_CSR5SPMV      CSR5 tiles, omega = 64
 */
typedef enum SPMV_METHODS {
    Method_Serial = 0,
    Method_Parallel = 1,
    Method_Balanced = 2,
    Method_Balanced2 = 3,
    Method_Balanced_Yid = 4,
    Method_SellCSigma = 5,
    Method_CSR5SPMV = 6,
    Method_Total_Size = 7,   /* number of entries in Methods_names[] */
    Method_Numa = 8          /* reference: compiled out (#ifdef NUMA); here: treated as out of range */
} SPMV_METHODS;

/* Indexed by SPMV_METHODS; has Method_Total_Size entries (spelling as in the reference, common.c:322-331). */
extern const char *Methods_names[];

/* "<method>_<vectorized>" labels, method-major; Method_Total_Size * VECTOR_TOTAL_SIZE entries. */
extern const char *funcNames[];

/*
 * The handle.  Field order, names and types are the reference's (spmv_Defines.h:44-68).
 *   spmvMethod       schedule actually in use; create() may rewrite the requested one
 *                    (reference does the same: parallel_balanced2_spmv.c:87-92, common.c:177-180)
 *   data_size        sizeof(double) or sizeof(float); anything != 8 is float (serial_spmv.c:48-54)
 *   nthreads         stored as given; the GPU schedules do not use host threads
 *   Level_3_opt_used 1 when option "reorder" produced a row/column permutation (then `index` is set), else 0
 *   RowPtr/ColIdx/Matrix_Val  the CALLER's pointers as passed to create (borrowed, never freed,
 *                    never written) -- used only to recognise the same matrix in spmv()
 *   index            NULL, or with option "reorder" the permutation (m ints, owned by the handle): the
 *                    caller gathers XX[i] = X[index[i]] and scatters Y[index[i]] = YY[i] like the
 *                    reference's harness does (test_spmv.c:95-101, 130-137)
 *   Y_temp           always NULL
 *   extraHandle      opaque device-side state (struct spmv_hip_state, private)
 */
typedef struct spmv_Handle {
    SPMV_METHODS spmvMethod;
    BASIC_SIZE_TYPE data_size;
    BASIC_SIZE_TYPE nthreads;
    VECTORIZED_WAY vectorizedWay;
    int Level_3_opt_used;
    BASIC_INT_TYPE *RowPtr;
    BASIC_INT_TYPE *ColIdx;
    BASIC_INT_TYPE *index;
    void *Matrix_Val;
    void *Y_temp;
    void *extraHandle;
} spmv_Handle;

typedef spmv_Handle *spmv_Handle_t;

/*
 * Type-punning helpers the reference publishes for code that is generic over data_size
 * (spmv_Defines.h:73-81); kept so that callers using them keep compiling.  `size` is the handle's
 * data_size: sizeof(double) selects double, anything else float (serial_spmv.c:48-54).
 */
#define CONVERT_FLOAT_T(pointer) ((float *) (pointer))
#define CONVERT_DOUBLE_T(pointer) ((double *) (pointer))
#define CONVERT_FLOAT(pointer) (*CONVERT_FLOAT_T(pointer))
#define CONVERT_DOUBLE(pointer) (*CONVERT_DOUBLE_T(pointer))
#define CONVERT_EQU(pointer, size, other) \
    (((size) == sizeof(double)) ? (CONVERT_DOUBLE(pointer) = (other)) : (CONVERT_FLOAT(pointer) = (other)))
#define CONVERT_ADDEQU(pointer1, size, pointer2)                                          \
    (((size) == sizeof(double)) ? (CONVERT_DOUBLE(pointer1) += CONVERT_DOUBLE(pointer2)) \
                                : (CONVERT_FLOAT(pointer1) += CONVERT_FLOAT(pointer2)))

/*
 * Sparse dot product of one CSR row with x, indexed by VECTORIZED_WAY (reference spmv_Defines.h:84-91,
 * tables in inner_spmv.h).  VECTOR_TOTAL_SIZE entries each.  These are HOST functions (a caller hands
 * them host pointers); on this build every entry is the same plain-C loop (csrc/host_rows.c) -- the GPU
 * schedules do not go through them.
 */
extern float (*const Dot_s_Products[])(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const float *Val, const float *X);
extern double (*const Dot_d_Products[])(BASIC_INT_TYPE len, const BASIC_INT_TYPE *indx, const double *Val, const double *X);

#endif /* SPMV_HIP_DEFINES_H */

#if defined(__cplusplus)
}
#endif
