#!/usr/bin/env python
"""HBM bytes per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--kernel-trace --pmc X --output-format csv`)
of tools/probe_model.py: per kernel template, launches and mean / total bytes per forward.
Correction per MI355X_MICROARCH.md §HBM (re-calibrated in round 1, tools/pmc_calib.hip): FETCH_SIZE counts 1/2 of the bytes
read, WRITE_SIZE is exact; both are in KB.   usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <forwards> [out.json]"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("fv::", "")


def collect(d, counter):
    path = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[(short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r.get("Workgroup_Size", 256) or 256), 1))].append(float(r["Counter_Value"]))
    return out


if __name__ == "__main__":
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    fw = int(sys.argv[3])
    rows = []
    for k, v in fetch.items():
        w = write.get(k, [0.0])
        rd = sum(v) / len(v) * 1024 * 2.0
        wr = sum(w) / len(w) * 1024
        rows.append({"kernel": k[0], "workgroups": k[1], "launches_per_forward": len(v) / fw, "read_bytes_per_launch": rd,
                     "write_bytes_per_launch": wr, "hbm_bytes_per_forward": (rd + wr) * len(v) / fw})
    rows.sort(key=lambda r: -r["hbm_bytes_per_forward"])
    res = {"forwards_sampled": fw, "total_hbm_bytes_per_forward": sum(r["hbm_bytes_per_forward"] for r in rows), "kernels": rows,
           "note": "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KB -> bytes; rows = (kernel template, workgroup count)"}
    if len(sys.argv) > 4:
        json.dump(res, open(sys.argv[4], "w"), indent=1)
    print(f"total {res['total_hbm_bytes_per_forward'] / 1e9:.2f} GB per forward")
    for r in rows[:12]:
        print(f"  {r['hbm_bytes_per_forward'] / 1e6:9.1f} MB  x{r['launches_per_forward']:5.1f}  {r['read_bytes_per_launch'] / 1e6:8.1f} R {r['write_bytes_per_launch'] / 1e6:8.1f} W  {r['kernel'][:90]} grid={r['workgroups']}")
