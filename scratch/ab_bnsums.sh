#!/bin/bash
# A/B of the BatchNorm-sums epilogue (DN_TUNE_NO_BN_SUMS=1 = the two-pass path): bench.py train legs, per precision
set -e
DT=${1:-bf16}
O=gpurun_out/ab_bnsums_$DT
mkdir -p $O
for rep in 1 2; do
  timeout -k 10 300 python bench.py --dtype $DT --no-cpu-baseline --no-legs > $O/on$rep.log 2>&1
  DN_TUNE_NO_BN_SUMS=1 timeout -k 10 300 python bench.py --dtype $DT --no-cpu-baseline --no-legs > $O/off$rep.log 2>&1
done
python - <<P
import json
for f in ("on1","off1","on2","off2"):
    l=[x for x in open("$O/%s.log" % f) if x.startswith('{"metric"')][-1]
    j=json.loads(l)["train_step"]; print("$DT", f, j["ms_per_step"], {k: v["ms_per_step"] for k, v in j["roofline"]["groups"].items()})
P
