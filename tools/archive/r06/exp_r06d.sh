set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
run() { # name lib extra-env
  ( [ "$2" = base ] || export FV_LIB_PATH=$R/vocoder_amd/csrc/libfishvoc_$2.so; export $3; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-alt-precision --no-collectives --profile-json $O/$1.json 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$1: ms/step %.3f eager %.3f  dominant %.1f us' % (j['ms_per_step'], j['ms_per_step_eager'], j['roofline']['avg_ms']*1e3))" )
}
for r in 1 2; do
run prev x_prev FV_X=0; run new base FV_X=0
run prev_ss x_prev FV_SINGLE_STREAM=1; run new_ss base FV_SINGLE_STREAM=1
done
python - <<PY
import json
a=json.load(open("$O/prev.json")); b=json.load(open("$O/new.json"))
A={r["kernel"]:r for r in a}; B={r["kernel"]:r for r in b}
rows=[]
for k in sorted(set(A)|set(B)):
    ta=A.get(k,{}).get("total_ms",0)/3; tb=B.get(k,{}).get("total_ms",0)/3
    rows.append((tb-ta,k,ta,tb,A.get(k,{}).get("launches",0)//3))
rows.sort()
for d,k,ta,tb,n in rows:
    if abs(d)>0.004: print("%+.3f ms  %7.3f -> %7.3f  x%d  %s"%(d,ta,tb,n,k))
print("sum prev %.3f new %.3f"%(sum(r["total_ms"] for r in a)/3, sum(r["total_ms"] for r in b)/3))
PY
