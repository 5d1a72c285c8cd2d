#!/usr/bin/env python
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--output-format csv --kernel-trace --pmc X`) of
tools/probe_forward.py into profiles/traffic.json: HBM bytes per launch for every engine kernel, keyed like bench.py's
roofline key.  Correction per MI355X_MICROARCH.md §HBM, re-calibrated here with tools/pmc_calib.hip (2 GiB streams, 4 B and
16 B per lane): FETCH_SIZE counts exactly 1/2 of the bytes read, WRITE_SIZE is exact; both are in KB.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json] [--merge] [--build NAME]"""
import collections, csv, glob, json, os, re, sys


def kernel_source_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources (vocoder_amd/csrc/*.hip, *.h, sorted by name): recorded next to a PMC collection, and
    compared by bench.py with the tree it runs from — a traffic figure looked up for a kernel whose source changed since is flagged stale."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vocoder_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def key_of(kernel_name: str, grid_threads: int):
    blocks = grid_threads // 256
    m = re.search(r"conv_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>", kernel_name)
    if m:
        ks, dil, wm, wn, mt, nt = map(int, m.groups())
        return f"conv_mfma k={ks} d={dil} tile={wm * mt * 32}x{wn * nt * 32} grid={blocks}"
    m = re.search(r"conv_wino_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", kernel_name)   # tile = rows x output PAIRS
    if m:
        ks, dil, wm, wn, nt = map(int, m.groups())
        return f"conv_wino k={ks} d={dil} tile={wm * 32}x{wn * nt * 32}p grid={blocks}"
    # (KS, DIL, VAR: 1 = the 64-channel layers, MT [, PRE, QR: round 5 — the activation in front and the row-split epilogue do not change the key]): 64 MT rows x 32 quad columns
    # (round 6: any number of trailing flags — QR, FLAT, PERS: with two of them the old pattern stopped matching and the dominant key kept a stale entry)
    m = re.search(r"conv_wino44_kernel<(\d+), (\d+), (\d+), (\d+)(?:, \d+)?(?:, (?:true|false))*>", kernel_name)
    if m:
        return f"conv_wino44 k={m.group(1)} d={m.group(2)} tile={64 * int(m.group(4))}x32q{' c64' if m.group(3) == '1' else ''} grid={blocks}"
    m = re.search(r"conv_wino4_kernel<(\d+), (\d+), (\d+), (true|false)>", kernel_name)   # (KS, DIL, WM, C64): tile = rows x QUAD columns, 128 WM threads
    if m:
        return f"conv_wino4 k={m.group(1)} d={m.group(2)} tile={32 * int(m.group(3))}x32q{' c64' if m.group(4) == 'true' else ''} grid={grid_threads // (128 * int(m.group(3)))}"
    m = re.search(r"conv_mfma_splitk_kernel<(\d+), (\d+), (\d+), (\d+)(?:, (?:true|false))?>", kernel_name)
    if m:
        return f"conv_mfma k={m.group(1)} d={m.group(2)} tile=splitK32x{32 * int(m.group(3))} grid={blocks}"
    m = re.search(r"conv_f16x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", kernel_name)
    if m:
        ks, dil, wm, wn, nt = map(int, m.groups())
        return f"conv_f16x3 k={ks} d={dil} tile={wm * 32}x{wn * nt * 32} grid={blocks}"
    m = re.search(r"pair_f16x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", kernel_name)
    if m:
        wg = 64 * int(m.group(3)) * int(m.group(4))   # C = 256 runs eight waves
        return f"pair_f16x3 k={m.group(1)} d={m.group(2)} C={32 * int(m.group(3))} grid={grid_threads // wg}"
    m = re.search(r"pair16_f16x3_kernel<(\d+), (\d+)>", kernel_name)
    if m:
        return f"pair_f16x3 k={m.group(1)} d={m.group(2)} C=16 grid={blocks}"
    m = re.search(r"pair_wino44_kernel<(\d+), (\d+), (\d+)>", kernel_name)   # (KS, DIL, C)
    if m:
        return f"pair_wino44 k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={blocks}"
    m = re.search(r"pair_wino(?:16|32)_kernel<(\d+), (\d+), (\d+),", kernel_name)   # (KS, DIL, C, chunk, ...)
    if m:
        return f"pair_wino k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={blocks}"
    m = re.search(r"conv_wino_lat_kernel<(\d+), (\d+), (\d+), (\d+)>", kernel_name)      # (KS, DIL, NT, MT)
    if m:
        return f"conv_wino_lat k={m.group(1)} d={m.group(2)} tile={16 * int(m.group(4))}x{16 * int(m.group(3))}p grid={blocks}"
    m = re.search(r"conv_wino_lat44_kernel<(\d+), (\d+), (\d+)>", kernel_name)      # (KS, DIL, MT)
    if m:
        return f"conv_wino_lat44 k={m.group(1)} d={m.group(2)} tile={16 * int(m.group(3))}x16q grid={blocks}"
    m = re.search(r"resblock_pair16_kernel<(\d+), (\d+)>", kernel_name)
    if m:
        return f"resblock_pair k={m.group(1)} d={m.group(2)} C=16 grid={blocks}"
    m = re.search(r"resblock_pair32_kernel<(\d+), (\d+), (\d+)(?:, \d+, (?:true|false))?>", kernel_name)   # (+ wave layout, residual source)
    if m:
        return f"resblock_pair k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={blocks}"
    return None


def bench_key(label: str):
    """Same key from a bench.py / fv_profile label."""
    m = re.search(r"conv_mfma<\w+ k=(\d+) d=(\d+) tile=(\w+)>.*grid=(\d+)", label)
    if m:
        return f"conv_mfma k={m.group(1)} d={m.group(2)} tile={m.group(3)} grid={m.group(4)}"
    m = re.search(r"conv_wino<k=(\d+) d=(\d+) tile=(\w+)>.*grid=(\d+)", label)
    if m:
        return f"conv_wino k={m.group(1)} d={m.group(2)} tile={m.group(3)} grid={m.group(4)}"
    m = re.search(r"conv_f16x3<k=(\d+) d=(\d+) tile=(\w+)>.*grid=(\d+)", label)
    if m:
        return f"conv_f16x3 k={m.group(1)} d={m.group(2)} tile={m.group(3)} grid={m.group(4)}"
    m = re.search(r"pair_f16x3<k=(\d+) d=(\d+) C=(\d+)> grid=(\d+)", label)
    if m:
        return f"pair_f16x3 k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={m.group(4)}"
    m = re.search(r"resblock_pair<k=(\d+) d=(\d+) C=(\d+)> grid=(\d+)", label)
    if m:
        return f"resblock_pair k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={m.group(4)}"
    m = re.search(r"pair_wino44<k=(\d+) d=(\d+) C=(\d+)> grid=(\d+)", label)
    if m:
        return f"pair_wino44 k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={(int(m.group(4)) + 7) // 8 * 8}"
    m = re.search(r"pair_wino<k=(\d+) d=(\d+) C=(\d+)> grid=(\d+)", label)
    if m:   # (the launch's grid is rounded up to a multiple of the 8 XCDs: key_of sees the rounded count)
        return f"pair_wino k={m.group(1)} d={m.group(2)} C={m.group(3)} grid={(int(m.group(4)) + 7) // 8 * 8}"
    m = re.search(r"conv_wino44<k=(\d+) d=(\d+) tile=(\w+)> cin=(\d+) cout=(\d+)(?: flat)? grid=(\d+)", label)
    if m:
        c64 = " c64" if (m.group(4), m.group(5)) == ("64", "64") else ""
        return f"conv_wino44 k={m.group(1)} d={m.group(2)} tile={m.group(3)}{c64} grid={(int(m.group(6)) + 7) // 8 * 8}"
    m = re.search(r"conv_wino4<k=(\d+) d=(\d+) tile=(\w+)> cin=(\d+) cout=(\d+) grid=(\d+)", label)
    if m:
        c64 = " c64" if (m.group(4), m.group(5)) == ("64", "64") else ""
        return f"conv_wino4 k={m.group(1)} d={m.group(2)} tile={m.group(3)}{c64} grid={(int(m.group(6)) + 7) // 8 * 8}"
    m = re.search(r"conv_wino_lat44<k=(\d+) d=(\d+) tile=(\w+)>.*grid=(\d+)", label)
    if m:
        return f"conv_wino_lat44 k={m.group(1)} d={m.group(2)} tile={m.group(3)} grid={m.group(4)}"
    m = re.search(r"conv_wino_lat<k=(\d+) d=(\d+) tile=(\w+)>.*grid=(\d+)", label)
    if m:
        return f"conv_wino_lat k={m.group(1)} d={m.group(2)} tile={m.group(3)} grid={m.group(4)}"
    return None


def collect(d, counter):
    # newest pass in the directory (rocprofv3 nests its output one level down, under the host name)
    path = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = key_of(r["Kernel_Name"], int(r["Grid_Size"]))
        if k:
            out[k].append(float(r["Counter_Value"]))
    return out


if __name__ == "__main__":
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k, v in fetch.items():
        f = sum(v) / len(v) * 1024.0 * 2.0          # KB -> B, x2 gfx950 correction
        w = sum(write.get(k, [0.0])) / max(len(write.get(k, [])), 1) * 1024.0
        res[k] = {"hbm_bytes_per_launch": f + w, "read_bytes": f, "write_bytes": w, "launches_sampled": len(v)}
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
    if os.path.exists(out) and "--merge" in sys.argv:   # add to an existing table (e.g. the f16x3 kernels to the fp32 ones)
        old = json.load(open(out))
        old.update(res)
        res = old
    for i, a in enumerate(sys.argv):   # --build NAME: recorded as "_build" (bench.py quotes it in roofline.traffic_source)
        if a == "--build" and i + 1 < len(sys.argv):
            res["_build"] = sys.argv[i + 1]
            res["_src_sha16"] = kernel_source_sha16()
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(f"{len(res)} kernels -> {out}")
