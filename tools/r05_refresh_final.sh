#!/bin/bash
# the bench lines of tools/refresh_profiles.sh once more at the round's final code
R=r05; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/refresh; mkdir -p $OUT; cd $ROOT
python3 bench.py 2>/dev/null | tail -1 > $OUT/${R}_bench.json
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${R}_bench_driver_flags.json
{ for C in c3 c5 c4; do python3 bench.py --config $C --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1; done
  python3 tools/bench_configs.py "C1 " 2>/dev/null | grep "^{"; python3 tools/bench_configs.py "64-ch" 2>/dev/null | grep "^{"; } > $OUT/${R}_other_configs.txt
python3 - <<'PY'
import json
for f in ("r05_bench.json", "r05_bench_driver_flags.json"):
    d = json.load(open("gpurun_out/refresh/" + f)); r = d["roofline"]
    print(f, round(d["value"], 4), round(d["ms_per_step"], 4), r["frac"], r["traffic"], round(d["chains2"]["value"], 4), round(d["chains2"]["ms_per_step_per_sample"], 4), d["gpu_over_cpu"])
for l in open("gpurun_out/refresh/r05_other_configs.txt"):
    d = json.loads(l)
    if "metric" in d: print(d["config"]["name"], round(d["ms_per_step"], 4), round(d["value"], 3), d["roofline"]["frac"], d["roofline"]["whole_step_mfma_frac"])
    else: print(str(d.get("config"))[:40], d.get("ms_per_step"))
PY
