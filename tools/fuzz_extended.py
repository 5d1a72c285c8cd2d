"""Extended differential fuzz run against the CPU oracle with seeds the regular set (tools/fuzz_all.py) does not use:
    python tools/fuzz_extended.py [seed_base]      (~6 min on one MI355X; bar 1e-4)"""
import importlib, os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(HERE, ".."))
base = int(sys.argv[1]) if len(sys.argv) > 1 else 9101
jobs = [("fuzz_conv", dict(n_cases=12000, seed=base)), ("fuzz_hifigan", dict(n_cases=1500, seed=base + 1)),
        ("fuzz_hifigan", dict(n_cases=200, seed=base + 2, large=True)), ("fuzz_hifigan", dict(n_cases=800, seed=base + 3, model="bigvgan")),
        ("fuzz_hifigan", dict(n_cases=100, seed=base + 4, large=True, model="bigvgan")), ("fuzz_vocos", dict(n_cases=400, seed=base + 5)),
        ("fuzz_sequence", dict(n_calls=2400, seed=base + 6)), ("fuzz_firefly", dict(n_cases=150, seed=base + 7)),
        ("fuzz_refinegan", dict(n_cases=150, seed=base + 8))]
n = sum(kw.get("n_cases", kw.get("n_calls", 0)) for _, kw in jobs)
print(f"extended differential fuzz run against the CPU oracle, seeds {base} .. {base + len(jobs) - 1} ({n} cases; bar 1e-4):", flush=True)
for name, kw in jobs:
    t = time.time()
    try:
        w = importlib.import_module(name).run(verbose=False, **kw)
        print(f"{name} {kw}: worst {w:.3e}  ({time.time() - t:.0f} s)", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"{name} {kw}: FAILED {type(e).__name__}: {str(e)[:600]}", flush=True)
