#!/usr/bin/env python3
"""DESIGN.md section 2's 'Measured' column, regenerated from the parity report of a GPU test run (tests/_report.py -> gpurun_out/parity_report.json, committed as
profiles/rNN_parity_report.json): python tools/parity_table.py profiles/r05_parity_report.json > profiles/r05_parity_table.md"""
import json
import sys

rep = json.load(open(sys.argv[1]))
meta = rep.pop("_meta", {})
print(f"Measured parity figures of the GPU test run on the shipped build (libagenda_hip.so sha256[:16] = {meta.get('libagenda_hip_sha16', '?')}); one row per `report()` call of the tests.\n")
print("| test (report id) | measured |")
print("|---|---|")
for k in sorted(rep):
    v = rep[k]
    cells = ", ".join(f"{m} {v[m]:.5g}" if isinstance(v[m], float) else f"{m} {v[m]}" for m in sorted(v))
    print(f"| `{k}` | {cells} |")
