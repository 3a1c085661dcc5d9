#!/usr/bin/env python3
"""Folds gpurun_out/grad_parity_observed.jsonl (written by tests/conftest.py::check_grads during `pytest -m gpu`) into
profiles/r03_grad_parity_observed.json: per test label the largest max|g - g_ref| / max|g_ref| per parameter group, and
the overall maximum the tolerance in tests/conftest.py (GRAD_TOL) is set against."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "grad_parity_observed.jsonl")
out = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r03_grad_parity_observed.json")
labels = {}
for line in open(src):
    r = json.loads(line)
    d = labels.setdefault(r["label"], {"max_err_over_tensor_scale": {}, "worst": r["worst"]})
    for g, v in r["max_err_over_tensor_scale"].items():
        d["max_err_over_tensor_scale"][g] = max(d["max_err_over_tensor_scale"].get(g, 0.0), v)
    if r["worst"]["err"] > d["worst"]["err"]:
        d["worst"] = r["worst"]
overall = max(d["worst"]["err"] for d in labels.values())
json.dump({"metric": "max|g - g_ref| / max|g_ref| per parameter tensor, maximum over the tensors of a parameter group",
           "reference": "gradients recorded from the reference's own modules (tests/golden/surrogate*_golden.npz); the "
                        "'fused vs plain' labels compare the two CUDA paths on the same device",
           "overall_max": overall, "labels": labels}, open(out, "w"), indent=1)
print("overall max", overall, "->", out)
