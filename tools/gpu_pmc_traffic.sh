#!/bin/bash
# HBM-side traffic of the render kernel for bench.py's roofline.traffic: separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# L2 hit/miss), kernel trace only.  usage (GPU box): tools/gpu_pmc_traffic.sh <tag>  -> gpurun_out/pmc_<tag>.json
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/b_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if 'render_fwd_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in acc.items()}
print(m)
# FETCH_SIZE / WRITE_SIZE are in 64-byte units... rocprofv3 reports them already scaled to KiB on gfx9: keep raw and
# the same corrections as profiles/pmc_render_fwd.json (r1): bytes = raw * 1024 for KiB-reported counters
json.dump(m, open("$R/gpurun_out/pmc_$TAG.json", "w"), indent=1)
PY
