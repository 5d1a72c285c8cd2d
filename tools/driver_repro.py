#!/usr/bin/env python
"""Writes profiles/<tag>_driver_repro.md: how `roofline.traffic` of the bench line was measured, with the exact commands, the counter correction and the
by-kernel tables of the three BASELINE configs — so that a reader can recompute the figure for the dominant kernel without trusting the lookup in
profiles/traffic.json.   python tools/driver_repro.py <collection dir> <tag> [bench.json]"""
import json, os, sys

d, tag = sys.argv[1], sys.argv[2]
bench = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else None
out = []
w = out.append
w(f"# {tag}: reproducing `roofline.traffic` (HBM bytes per launch of the dominant kernel)\n")
w("`bench.py` measures the dominant kernel's duration live (hipEvents around every launch, `fv_profile_begin/end`) but LOOKS UP its HBM traffic in "
  "`profiles/traffic.json` by kernel key — PMC counters cannot be read from inside the timed process.  The lookup is guarded: the collection stores a hash of "
  "`vocoder_amd/csrc/*.hip|*.h` (`_src_sha16`) and the bench line's `traffic_source` says whether the kernel sources changed since.  This file is the recipe "
  "behind that number.\n")
w("## Commands (one MI355X, from the repo root; FETCH_SIZE and WRITE_SIZE need separate passes)\n")
w("```bash\ncd /tmp && export TMPDIR=/tmp\nR=/root/repo\nfor m in hifigan bigvgan vocos; do\n"
  "  B=$([ $m = hifigan ] && echo 32 || ([ $m = bigvgan ] && echo 64 || echo 128))\n"
  "  for c in FETCH_SIZE WRITE_SIZE; do\n"
  "    rocprofv3 --kernel-trace --pmc $c --output-format csv -d out/pmc_${m}_$c -- python $R/tools/probe_model.py $m $B 2\n"
  "  done\n"
  "  python $R/tools/pmc_summary.py out/pmc_${m}_FETCH_SIZE out/pmc_${m}_WRITE_SIZE 2 out/${m}_hbm_traffic.json\ndone\n"
  "python $R/tools/pmc_traffic.py out/pmc_hifigan_FETCH_SIZE out/pmc_hifigan_WRITE_SIZE $R/profiles/traffic.json --build " + tag + "   # written fresh, not merged (LOG R6.15)\n```\n")
w("`tools/probe_model.py <model> <batch> 2` runs two forwards of the BASELINE configuration on ONE stream without graphs, in the TREE form of the branch mean (`FV_SINGLE_STREAM=2`: the kernel instances of the shipped three-stream step, every launch its own row). "
  "`--pmc` is combined with `--kernel-trace` only (no `--sys-trace` / hip / hsa / memory-copy domains).\n")
w("## Counter correction (`/opt/skills/guides/MI355X_MICROARCH.md`, HBM / rocprofv3 section; re-calibrated with `tools/pmc_calib.hip`)\n")
w("* `FETCH_SIZE` and `WRITE_SIZE` are reported in **KB**;\n* on gfx950 `FETCH_SIZE` counts **half** of the bytes read: bytes read = `FETCH_SIZE x 1024 x 2`;\n"
  "* `WRITE_SIZE` is exact: bytes written = `WRITE_SIZE x 1024`;\n* per launch = the mean over the launches of one (kernel template, workgroup count) row.\n")
for m, title in (("hifigan", "HiFiGAN-V1 44.1 kHz, B = 32 x 1 s (BASELINE config[1], the headline)"), ("bigvgan", "BigVGAN-24k, B = 64 x 1 s (config[2])"),
                 ("vocos", "Vocos-24k, B = 128 x 1 s (config[3])")):
    p = os.path.join(d, f"{m}_hbm_traffic.json")
    if not os.path.exists(p):
        w(f"## {title}\n\n(not collected)\n")
        continue
    j = json.load(open(p))
    w(f"## {title}: {j['total_hbm_bytes_per_forward'] / 1e9:.2f} GB per forward\n")
    w("| MB per forward | launches per forward | read MB / launch | written MB / launch | kernel (template instance) | workgroups |\n|---:|---:|---:|---:|---|---:|")
    for r in j["kernels"][:24]:
        w(f"| {r['hbm_bytes_per_forward'] / 1e6:.1f} | {r['launches_per_forward']:.1f} | {r['read_bytes_per_launch'] / 1e6:.1f} | {r['write_bytes_per_launch'] / 1e6:.1f} | "
          f"`{r['kernel'][:110]}` | {r['workgroups']} |")
    w("")
if bench:
    r = bench.get("roofline", {})
    w("## The bench line's dominant kernel\n")
    w(f"`{r.get('kernel')}`: `traffic` = {r.get('traffic')} bytes per launch ({r.get('traffic_source')}); algorithmic bytes per launch {r.get('bytes_per_launch')}; "
      f"duration {r.get('avg_ms')} ms by hipEvents.  Recompute: find the row with the same template instance and workgroup count in the HiFiGAN table above; "
      "read + written MB per launch = `traffic`.\n")
print("\n".join(out))
