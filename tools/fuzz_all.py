import sys, time
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import importlib
jobs = [("fuzz_hifigan", dict(n_cases=400, seed=311, verbose=False)), ("fuzz_hifigan", dict(n_cases=60, seed=312, verbose=False, large=True)),
        ("fuzz_hifigan", dict(n_cases=400, seed=313, verbose=False, model="bigvgan")), ("fuzz_hifigan", dict(n_cases=60, seed=314, verbose=False, large=True, model="bigvgan")),
        ("fuzz_vocos", dict(n_cases=200, seed=315, verbose=False)), ("fuzz_conv", dict(n_cases=2000, seed=316, verbose=False)),
        ("fuzz_sequence", dict(n_calls=600, seed=317, verbose=False)), ("fuzz_firefly", dict(n_cases=60, seed=318, verbose=False)),
        ("fuzz_refinegan", dict(n_cases=60, seed=319, verbose=False))]
for name, kw in jobs:
    t = time.time()
    try:
        w = importlib.import_module(name).run(**kw)
        print(f"{name} {kw}: worst {w:.3e}  ({time.time() - t:.0f} s)", flush=True)
    except Exception as e:
        print(f"{name} {kw}: FAILED {type(e).__name__}: {str(e)[:600]}", flush=True)
