mkdir -p gpurun_out/r06h; O=$PWD/gpurun_out/r06h
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^fused grid" | tail -6 > $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err
cat $O/tests.log; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06h/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","roofline")})
print(json.dumps(d["detail"].get("fused"), indent=1))
PY
