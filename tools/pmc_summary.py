#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + --pmc passes) into one small per-kernel table.

    python3 tools/pmc_summary.py <dir with pass sub-directories> [out.json] > summary.txt
Per kernel name: calls, mean duration, and the per-launch mean of every counter collected; the same
table as JSON when a second argument is given (bench.py reads profiles/round1_pmc.json).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    """sha256 over the kernel sources (velox_amd/csrc/*, sorted by name): bench.py quotes counters from a
    committed profile only while this still matches the tree it runs from."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "velox_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:48]


def main(root, json_out=None):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        if os.sep + "trace" + os.sep not in f:
            continue
        with open(f) as fh:
            for r in csv.DictReader(fh):
                dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                key = (r["Dispatch_Id"], r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (did, c), v in per_dispatch.items():
            ctr[names[did]][c].append(v)
    kernels = sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0])))
    table = {}
    for k in kernels:
        if not k.startswith("k_"):
            continue
        d = dur.get(k, [])
        row = {"calls": len(d), "avg_us": round(sum(d) / len(d) / 1e3, 1) if d else 0.0}
        line = f"{k:50s} calls={len(d):3d} avg_us={row['avg_us']:10.1f}"
        print(line)
        for c in sorted(ctr.get(k, {})):
            v = ctr[k][c]
            row[c] = round(sum(v) / len(v), 1)
            print(f"    {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")
        # derived, as bench.py computes them: the fraction of SIMD issue time spent on vector ALU instructions
        # (4 cycles per wave64 instruction, 1024 SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs), and HBM bytes per
        # launch (FETCH_SIZE doubled for gfx950's 128-byte read requests, both in KiB)
        if row.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in row:
            row["VALUBusy"] = round(row["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (row["GRBM_GUI_ACTIVE"] / 8.0), 3)
            print(f"    {'VALUBusy (derived)':28s} {row['VALUBusy']:18.3f}")
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["hbm_bytes"] = int((2.0 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024)
            print(f"    {'HBM bytes (derived)':28s} {row['hbm_bytes']:18d}")
        table[k] = row
    table["_meta"] = {"csrc_sha256": csrc_hash(), "git_head": os.environ.get("VLX_GIT_HEAD", ""), "program": os.environ.get("PROF_PROG", "bench.py")}
    print(f"# csrc_sha256={table['_meta']['csrc_sha256']} git_head={table['_meta']['git_head']} program={table['_meta']['program']}")
    if json_out:
        with open(json_out, "w") as fh:
            json.dump(table, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
