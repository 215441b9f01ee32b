#!/usr/bin/env python3
"""Probe: what does hipMalloc cost right after a large hipFree?  (create-time outliers)"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
def malloc(n):
    p = C.c_void_p(); t = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); return p, (time.perf_counter() - t) * 1e3, rc
def free(p):
    t = time.perf_counter(); hip.hipFree(p); return (time.perf_counter() - t) * 1e3
hip.hipDeviceSynchronize()
for size in (256 << 20, 2 << 30, 4 << 30):
    for rep in range(3):
        p, tm, rc = malloc(size)
        t = time.perf_counter(); hip.hipMemset(p, 0, size); hip.hipDeviceSynchronize(); ts = (time.perf_counter() - t) * 1e3
        tf = free(p)
        print(f"size {size >> 20:5d} MiB rep {rep}: malloc {tm:8.2f} ms  first memset {ts:8.2f} ms  free {tf:8.2f} ms")
# several buffers alive, then freed, then allocated again (a handle destroyed, the next created)
bufs = [malloc(2 << 30)[0] for _ in range(3)]
for b in bufs: hip.hipMemset(b, 0, 2 << 30)
hip.hipDeviceSynchronize()
t = time.perf_counter()
for b in bufs: hip.hipFree(b)
print("free 3 x 2 GiB: %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter()
bufs = [malloc(2 << 30)[0] for _ in range(3)]
print("malloc 3 x 2 GiB again: %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter()
for b in bufs: hip.hipMemset(b, 0, 2 << 30)
hip.hipDeviceSynchronize()
print("memset them: %.2f ms" % ((time.perf_counter() - t) * 1e3))
