"""Identity of the MRLA HIP library: sha256 of mrla_amd/libmrla_hip.so and of the sources it is built from.

Counter passes (scripts/pmc_bench.sh, scripts/pmc_mfma.sh) write this record into their summaries (`_meta`); bench.py
compares it with the library it has loaded and refuses to quote counters of another build (`roofline.traffic_stale`).

    python3 scripts/lib_identity.py            # prints the record as JSON
"""
import glob
import hashlib
import json
import os
import sys


def identity(root):
    lib = os.path.join(root, "mrla_amd", "libmrla_hip.so")
    h = hashlib.sha256()
    with open(lib, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    src = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "mrla_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mrla_amd", "csrc", "*.h"))
                   + [os.path.join(root, "mrla_amd", "csrc", "Makefile"), os.path.join(root, "include", "mrla_hip.h")])
    for path in files:
        src.update(os.path.relpath(path, root).encode() + b"\0")
        with open(path, "rb") as f:
            src.update(f.read())
        src.update(b"\0")
    return {"lib_sha256": h.hexdigest(), "src_sha256": src.hexdigest(), "sources": len(files)}


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    print(json.dumps(identity(os.path.abspath(root))))
