#!/bin/bash
# extra PMC passes for bottleneck hunting: usage tools/gpu_pmc_extra.sh <tag>
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_CVT" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/b_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_fwd' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        print('%-40s %.4g' % (c, sum(v)/len(v)))
PY
