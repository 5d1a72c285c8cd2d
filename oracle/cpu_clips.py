"""Worker of bench.py's ``cpu_baseline`` leg: one host PROCESS that synthesises whole clips with the CPU oracle.

TEST / BENCH INFRASTRUCTURE ONLY (see the header of ``fv_oracle.c``); the product package never imports this.

The reference synthesises clip by clip on one process (``/root/reference/fish_vocoder/test.py:73-99``: one file per model
call); its only parallelism is inside torch's convs.  The honest many-core CPU figure for a *batch* of independent clips is
clip-parallel: every worker owns whole clips and runs the oracle's convs serially (OMP_NUM_THREADS=1).  Workers are separate
processes — the oracle's forward is a Python walk over ~300 C calls per clip, and 256 threads of ONE interpreter would
serialise on its lock for about as long as the convs take.

Protocol (stdin / stdout, one line each way): the parent writes ``run <n_clips> <frames> <seed> <start_at_epoch>``; the
worker sleeps until ``start_at``, synthesises its clips and answers with a JSON line ``{"t0", "t1", "clips", "samples",
"checksum"}`` (wall-clock stamps of its first and last conv).  ``quit`` ends it.
"""
from __future__ import annotations

import json
import os
import sys
import time

os.environ["OMP_NUM_THREADS"] = "1"
_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
sys.path[:] = [REPO] + [p for p in sys.path if os.path.abspath(p or ".") not in (_HERE, REPO)]   # `oracle` = the package, not oracle.py


def main() -> None:
    import numpy as np
    from oracle import oracle as orc
    from vocoder_amd import synthetic as syn
    cfg = dict(syn.HIFIGAN_V1_44K)
    sd = syn.hifigan_state_dict(cfg, seed=0)
    orc.set_num_threads(1)
    orc.hifigan_forward(sd, cfg, syn.synthetic_mel(1, cfg["num_mels"], 4, seed=1))   # page the library and the weights in
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    for line in sys.stdin:
        parts = line.split()
        if not parts or parts[0] == "quit":
            break
        n, frames, seed, start_at = int(parts[1]), int(parts[2]), int(parts[3]), float(parts[4])
        mel = syn.synthetic_mel(n, cfg["num_mels"], frames, seed=seed)
        delay = start_at - time.time()
        if delay > 0:
            time.sleep(delay)
        t0 = time.time()
        samples, checksum = 0, 0.0
        for i in range(n):
            y = orc.hifigan_forward(sd, cfg, mel[i:i + 1])
            samples += y.shape[-1]
            checksum += float(np.abs(y).sum())
        t1 = time.time()
        sys.stdout.write(json.dumps({"t0": t0, "t1": t1, "clips": n, "samples": samples, "checksum": checksum}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
