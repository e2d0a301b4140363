# What does the field backward (or NFI_DIAG_KERNEL=bin_reduce_kernel) WAIT for?  The counters of pmc_diag.sh on the replayed training-step backward.
# --pmc passes (kernel trace only).  bash tools/probes/pmc_diag.sh [extra bench flags]      (GPU box)
R=$PWD; O=$R/gpurun_out/pmc_diag_bwd; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/bench_train_backward.py 6"
i=0
for P in "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PERF_SEL_TOTAL_READ_sum TCP_PERF_SEL_TOTAL_HIT_LRU_READ_sum" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o x -- $CMD > $O/p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
KERNEL = os.environ.get("NFI_DIAG_KERNEL", "field_query_bwd_kernel")
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_diag_bwd/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('%-44s mean %.4g  (n=%d, min %.4g, max %.4g)' % (k, sum(v) / len(v), len(v), min(v), max(v)))
PY
rm -rf gpurun_out/pmc_diag_bwd/p*/
