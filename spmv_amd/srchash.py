"""Hash of the library's sources (spmv_amd/csrc/** + include/*): what a committed rocprofv3 measurement is valid for.
bench.py reports counter traffic from profiles/traffic_rNN.json only when the file's `csrc_sha` equals the tree's hash."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    h = hashlib.sha256()
    files = []
    for base in (os.path.join(ROOT, "spmv_amd", "csrc"), os.path.join(ROOT, "include")):
        for dirpath, _, names in os.walk(base):
            files += [os.path.join(dirpath, n) for n in names if n.endswith((".h", ".hpp", ".hip", ".c"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
