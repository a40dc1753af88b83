#!/usr/bin/env python3
"""Hash of the sources a measurement of the step depends on: the kernels, the C ABI and the Python engine (vit-gan_amd/**,
include/*.h; no build products).  The PMC summaries under profiles/ carry it (tools/profile_round.sh), and bench.py quotes their
figures only when the tree it runs from hashes the same - the GPU box has no .git, so a commit id cannot make that link there."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".py", ".hip", ".h", "Makefile")


def tree_hash(root=ROOT):
    files = []
    for base in ("vit-gan_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(root, base)):
            dn[:] = sorted(d for d in dn if d not in ("build", "build_var", "__pycache__", ".pytest_cache"))
            for f in sorted(fn):
                if f.endswith(EXT):
                    files.append(os.path.join(dp, f))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(tree_hash(sys.argv[1] if len(sys.argv) > 1 else ROOT))
