#!/usr/bin/env python3
"""Identity of everything the GPU test suite and the bench execute: sources, tests, fixtures (not docs, not profiles,
not built artefacts).  tools/final_check.sh records it with its results; ``tools/tree_sha.py --verify`` (run in the
build container before the round is closed) fails when the tree has changed since the last full GPU run."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".py", ".hip", ".h", ".c", ".npz", ".sh")
SKIP_DIRS = {".git", "gpurun_out", "__pycache__", ".pytest_cache", "profiles", "build", "lib", "_build", "_ref", ".hypothesis"}


def tree_sha():
    h = hashlib.sha256()
    for base, dirs, files in os.walk(ROOT):
        dirs[:] = sorted(d for d in dirs if d not in SKIP_DIRS)
        for f in sorted(files):
            if f.endswith(EXT) or f == "Makefile":
                p = os.path.join(base, f)
                h.update(os.path.relpath(p, ROOT).encode())
                with open(p, "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sha = tree_sha()
    if "--verify" in sys.argv:
        rec = os.path.join(ROOT, "gpurun_out", "final_check.json")
        try:
            d = json.load(open(rec))
        except (OSError, ValueError):
            sys.exit(f"no {rec}: the full GPU suite has not been run (tools/final_check.sh)")
        if d.get("tree_sha") != sha:
            sys.exit(f"tree changed since the last full GPU run ({d.get('tree_sha')} -> {sha}): run tools/final_check.sh again")
        if not d.get("gpu_tests_ok"):
            sys.exit(f"last full GPU run was RED: {d}")
        print("final check is current:", d)
    else:
        print(sha)
