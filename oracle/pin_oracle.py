#!/usr/bin/env python3
"""Pin the oracle against the REAL reference and (re)write the golden fixtures.

Run in the build container (needs /root/reference for `make -C oracle ref`):

    python oracle/pin_oracle.py            # verify + write tests/golden/*.npz + manifest.json
    python oracle/pin_oracle.py --check    # verify only (fixtures on disk must match)

For every case in oracle/cases.py:
  1. y_ref  = real reference: spmv_create_handle_all_in_one(Method_Serial, nthreads=1,
              VECTOR_NONE, MtxToken=NULL) + spmv()   [common.c:123-190, 278-304], y pre-filled
              with NaN so an unwritten row would show;
  2. y_orc  = oracle/oracle_spmv.c (oracle_spmv_serial);
  3. require y_orc == y_ref BIT FOR BIT (fp64 and fp32, uniform and eighths fills);
  4. for the eighths fill additionally require y_ref == the double-accumulated exact sum
     (the reference harness' own golden, test_spmv.c:204-207) -- exact arithmetic;
  5. store inputs + y_ref as tests/golden/<case>.npz.

The fixtures are data (inputs and the reference's outputs); no reference source is stored.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import cases  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()

    oracle.build()
    if not oracle.have_ref():
        sys.exit("oracle/_ref/libmv_l2.so missing: /root/reference not present?")
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {"reference": "DevilInChina/spmv @ /root/reference, Method_Serial, nthreads=1, VECTOR_NONE",
                "build": "oracle/Makefile (gcc -O3 -mavx -mavx2 -mfma -fopenmp)", "cases": {}}
    bad = 0
    for name in cases.case_names():
        csr, x = cases.build_case(name)
        y_ref, actual = oracle.ref_spmv(csr, x, method=0, nthreads=1)
        assert actual == 0
        y_orc = oracle.spmv_serial(csr, x)
        unwritten = int(np.isnan(y_ref).sum())
        same = np.array_equal(y_ref.view(np.uint8), y_orc.view(np.uint8))
        exact_ok = True
        if name.endswith("eighths"):
            exact_ok = np.array_equal(y_ref.astype(np.float64), oracle.spmv_exact(csr, x))
        status = "ok" if (same and exact_ok and unwritten == 0) else "MISMATCH"
        if status != "ok":
            bad += 1
        print(f"{name:28s} m={csr.m:5d} n={csr.n:5d} nnz={csr.nnz:7d} bitwise={same} exact={exact_ok} "
              f"unwritten={unwritten} {status}")
        path = os.path.join(GOLDEN, name + ".npz")
        payload = dict(m=np.int64(csr.m), n=np.int64(csr.n), rowptr=csr.rowptr, colidx=csr.colidx,
                       val=csr.val, x=x, y_ref=y_ref)
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(payload[k]).tobytes()
                                         for k in ("rowptr", "colidx", "val", "x", "y_ref"))).hexdigest()
        manifest["cases"][name] = {"m": csr.m, "n": csr.n, "nnz": csr.nnz, "sha256": digest}
        if args.check:
            with np.load(path) as z:
                for k in ("rowptr", "colidx", "val", "x", "y_ref"):
                    if not np.array_equal(z[k].view(np.uint8), np.ascontiguousarray(payload[k]).view(np.uint8)):
                        print(f"  fixture {name}:{k} differs from regenerated data")
                        bad += 1
        else:
            np.savez_compressed(path, **payload)
    if not args.check:
        with open(os.path.join(GOLDEN, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
    print("PINNED" if bad == 0 else f"{bad} problem(s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
