"""spmv_amd -- MI355X-native SpMV engine behind the C API of DevilInChina/spmv.

    spmv_amd.api     ctypes mirror of the C ABI (spmv_create_handle_all_in_one / spmv / ...)
    spmv_amd.build   hipcc build of spmv_amd/lib/libspmv_hip.so (gfx950)
    spmv_amd.synth   synthetic CSR generators (host + device)
    spmv_amd.dist    row-block partitioning across GPUs, x exchange over RCCL
    spmv_amd.csrc    host C (handle, planner) + HIP shim + kernels
"""
__version__ = "0.1.0"
