#!/bin/bash
# Utilisation counters of the render kernel (separate PMC passes, kernel trace only).
# usage (GPU box): tools/gpu_pmc_util.sh <tag>  -> gpurun_out/pmc_util_<tag>.json
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_util_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
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
for k, v in sorted(m.items()): print('%-32s %.4g' % (k, v))
json.dump(m, open("$R/gpurun_out/pmc_util_$TAG.json", "w"), indent=1)
PY
