"""Summarise rocprofv3 --pmc counter CSVs (one pass per counter) into the small JSON bench.py reads for `roofline.traffic`:
  python tools/pmc_summary.py OUT.json FETCH_SIZE=dir1 WRITE_SIZE=dir2 [SQ_WAVES=dir3 ...]
Per counter and kernel family (vertex*, edge*, ...): launches and mean raw counter value per launch (KB for FETCH/WRITE_SIZE)."""
import csv
import glob
import json
import os
import sys


def summarise(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    out = {}
    for r in rows:
        name = r.get("Kernel_Name") or r.get("kernel_name") or ""
        val = float(r.get("Counter_Value") or r.get("counter_value") or 0.0)
        fam = "vertex_wg_kernel" if "vertex_wg_kernel" in name else ("vertex_kernel" if "vertex_kernel" in name else
              ("edge_kernel" if "edge_kernel" in name else ("finalize_control" if "finalize" in name else None)))
        if fam is None:
            continue
        e = out.setdefault(fam, {"launches": 0, "sum": 0.0})
        e["launches"] += 1; e["sum"] += val
    return {k: {"launches": v["launches"], "mean_KB_per_launch": v["sum"] / max(v["launches"], 1)} for k, v in out.items()}


if __name__ == "__main__":
    res = {}
    for arg in sys.argv[2:]:
        ctr, d = arg.split("=", 1)
        res[ctr] = summarise(d)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res))
