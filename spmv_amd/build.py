"""Build libspmv_hip.so in-tree (spmv_amd/lib/) for gfx950.

    python -m spmv_amd.build            # rebuild what is out of date
    python -m spmv_amd.build --force

Host C (spmv_api.c, spmv_plan.c) is compiled by gcc as C11; the HIP shim by hipcc for gfx950
only (no other offload arch, no CUDA path).  The shared object carries no torch dependency: its
entry points are the plain C ABI of include/*.h.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libspmv_hip.so")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CC = os.environ.get("CC") or "gcc"
ARCH = "gfx950"

C_SOURCES = ["spmv_api.c", "spmv_plan.c", "host_rows.c", os.path.join("io", "mtx_io.c"), os.path.join("reorder", "rcm.c")]
TOOL_SOURCES = {"test_spmv": os.path.join("tools", "test_spmv_csv.c")}   # -> spmv_amd/bin/<name>
HIP_TOOLS = {"gbench": os.path.join(ROOT, "tools", "gbench.hip")}          # measurement programs (bench.py's same-box read ceiling); not part of the library
BINDIR = os.path.join(PKG, "bin")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
# (source, extra defines, object name): compiled side by side.  The CSR-vector family's executors (seven lanes-per-row values x five
# forms x two value types) are most of the device code: spmv_vector.hip is compiled four times, a quarter of the instantiations each
HIP_SOURCES = [("spmv_shim.hip", [], "spmv_shim.hip.o")] + [("spmv_vector.hip", [f"SPMV_VEC_PART={k}"], f"spmv_vector{k}.hip.o") for k in range(4)]
HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
             "-ffp-contract=fast"]
C_FLAGS = ["-O2", "-std=c11", "-fPIC", "-fopenmp", "-Wall", "-Wextra", "-D_POSIX_C_SOURCE=200809L"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _deps():
    out = []
    for base, _, files in os.walk(CSRC):
        out += [os.path.join(base, f) for f in files]
    out += [os.path.join(INC, f) for f in os.listdir(INC)]
    return out


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def _run_all(cmds):
    """Independent compiles, side by side."""
    procs = [(c, subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for c in cmds]
    errs = []
    for c, p in procs:
        out, err = p.communicate()
        if p.returncode != 0:
            errs.append("build failed: " + " ".join(c) + "\n" + out + err)
        elif err.strip():
            sys.stderr.write(err)
    if errs:
        raise RuntimeError("\n".join(errs))


def build_debug(defines, name="libspmv_hip_dbg.so", verbose=False):
    """A/B build for tools/ (never loaded by the package): the same sources with extra -D defines -> spmv_amd/lib/<name>."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    for src in C_SOURCES:
        obj = os.path.join(OBJDIR, "dbg_" + src.replace(os.sep, "_") + ".o")
        _run([CC, *C_FLAGS, f"-I{INC}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    cmds = []
    for src, defs, oname in HIP_SOURCES:
        obj = os.path.join(OBJDIR, "dbg_" + oname)
        cmds.append([HIPCC, *HIP_FLAGS, *[f"-D{d}" for d in list(defines) + defs], f"-I{INC}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    _run_all(cmds)
    gomp = subprocess.run([CC, "-print-file-name=libgomp.so"], capture_output=True, text=True).stdout.strip()
    out = os.path.join(LIBDIR, name)
    _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out, *objs, gomp, "-lpthread", "-lm"])
    return out


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    tools = [os.path.join(BINDIR, t) for t in list(TOOL_SOURCES) + list(HIP_TOOLS)]
    if not force and all(os.path.exists(p) for p in [LIB] + tools) and min(os.path.getmtime(p) for p in [LIB] + tools) >= _newest(_deps() + list(HIP_TOOLS.values())):
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found ({HIPCC}); libspmv_hip.so cannot be built")
    objs = []
    for src in C_SOURCES:
        obj = os.path.join(OBJDIR, src.replace(os.sep, "_") + ".o")
        cmd = [CC, *C_FLAGS, f"-I{INC}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
        objs.append(obj)
    cmds = []
    for src, defs, oname in HIP_SOURCES:
        obj = os.path.join(OBJDIR, oname)
        cmd = [HIPCC, *HIP_FLAGS, *[f"-D{d}" for d in defs], f"-I{INC}", f"-I{CSRC}", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        cmds.append(cmd)
        objs.append(obj)
    for name, src in HIP_TOOLS.items():
        cmds.append([HIPCC, "-O3", "-std=c++17", f"--offload-arch={ARCH}", src, "-o", os.path.join(BINDIR, name)])
    _run_all(cmds)
    # host_rows.c is compiled with -fopenmp by gcc: link GNU libgomp by path (hipcc's own -fopenmp would pull LLVM's runtime)
    gomp = subprocess.run([CC, "-print-file-name=libgomp.so"], capture_output=True, text=True).stdout.strip()
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, gomp, "-lpthread", "-lm"]
    if verbose:
        print(" ".join(cmd))
    _run(cmd)
    # host-side C tools that link the library (the reference builds its harness the same way)
    for name, src in TOOL_SOURCES.items():
        cmd = [CC, *C_FLAGS, "-D__HIP_PLATFORM_AMD__", f"-I{INC}", f"-I{ROCM}/include", os.path.join(CSRC, src),
               f"-L{LIBDIR}", "-lspmv_hip", f"-L{ROCM}/lib", "-lamdhip64", "-lm",
               "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{ROCM}/lib", "-o", os.path.join(BINDIR, name)]
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
    return LIB


if __name__ == "__main__":
    if "--debug" in sys.argv:   # python -m spmv_amd.build --debug SPMV_BLK_DEBUG_FORMS [...]
        print(build_debug(sys.argv[sys.argv.index("--debug") + 1:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
