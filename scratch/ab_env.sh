#!/bin/bash
# A/B of one environment switch on bench.py's training leg and on the batch-16 fit() leg: ab_env.sh VAR [bf16|fp32]
set -e
VAR=$1; DT=${2:-bf16}
O=gpurun_out/ab_$VAR_$DT; mkdir -p $O
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --dtype $DT --no-cpu-baseline --no-legs > $O/on$rep.log 2>&1
  env $VAR=1 timeout -k 10 300 python bench.py --dtype $DT --no-cpu-baseline --no-legs > $O/off$rep.log 2>&1
done
for rep in 1 2; do
  timeout -k 10 120 python scratch/profile_fit.py $DT 2>/dev/null | tail -1 > $O/fit_on$rep.log
  env $VAR=1 timeout -k 10 120 python scratch/profile_fit.py $DT 2>/dev/null | tail -1 > $O/fit_off$rep.log
done
python - <<P
import json
for f in ("on1","off1","on2","off2","on3","off3"):
    l=[x for x in open("$O/%s.log" % f) if x.startswith('{"metric"')][-1]
    print("$DT $VAR", f, json.loads(l)["train_step"]["ms_per_step"])
for f in ("fit_on1","fit_off1","fit_on2","fit_off2"):
    print("$DT $VAR", f, json.loads(open("$O/%s.log" % f).read())["ms_per_step"])
P
