/*
 * spmv_hip_tools.h -- measurement entry points used by bench.py, tests/ and tools/ only.  NOT part of the drop-in surface (spmv.h) and not
 * part of the extension API an application needs (spmv_hip.h): a program that multiplies never includes this header.  The reference times
 * its multiplies from the outside, with gettimeofday around 100 calls (test_spmv.c:103-127); on a GPU the launch stream has to be
 * bracketed by events, which only the library can place between its own launches.
 */
#include "spmv_Defines.h"
#if defined(__cplusplus)
extern "C" {
#endif
#ifndef SPMV_HIP_TOOLS_H
#define SPMV_HIP_TOOLS_H

/* `warmup` untimed + `iters` timed spmv() launches back to back on the handle's stream, each
 * timed launch bracketed by hipEvents recorded on that stream; ms_out[i] (may be NULL) receives
 * launch i's duration.  x and y must be DEVICE pointers.  Returns the mean in ms, < 0 on error. */
double spmv_hip_time_launches(spmv_Handle_t handle, const void *x, void *y,
                              int warmup, int iters, float *ms_out);

#endif /* SPMV_HIP_TOOLS_H */
#if defined(__cplusplus)
}
#endif
