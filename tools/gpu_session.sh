#!/bin/bash
# One GPU-box session: tests, parity report, bench (render + train), rocprofv3 stats + PMC of the render kernel.
# usage (via gpurun): bash tools/gpu_session.sh <tag> [what...]   what: tests report refreport e2e e2e_prof bench train pmc pmc_bwd bwd handoff prof_inv prof_train prof_bench regulariser train_bwd stress   (default: tests report bench train pmc)
TAG=${1:-s}
shift
WHAT=${@:-tests report bench train pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "reference checkout on the GPU box: $(ls -d /root/reference 2>&1); staged copy: $(python oracle/make_ref.py --check 2>&1)" > $O/env.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> $O/env.txt
nproc >> $O/env.txt
for w in $WHAT; do
  case $w in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/env.txt; tail -5 $O/pytest.log;;
    sweep)
      timeout 1500 python tools/parity_sweep.py 10 $SWEEP_ONLY $SWEEP_CPU_ALL > $O/parity_sweep.json 2> $O/parity_sweep.err; echo "sweep rc=$?" >> $O/env.txt; tail -3 $O/parity_sweep.err; tail -c 1500 $O/parity_sweep.json;;
    gsphere)
      timeout 900 python tools/g_sphere.py 300 8 > $O/g_sphere.json 2> $O/g_sphere.err; echo "gsphere rc=$?" >> $O/env.txt; tail -3 $O/g_sphere.err; tail -c 2500 $O/g_sphere.json;;
    gradspread)
      timeout 900 python tools/gradient_spread.py 3 > $O/gradient_spread.json 2> $O/gradient_spread.err; echo "gradspread rc=$?" >> $O/env.txt; tail -3 $O/gradient_spread.err;;
    reftests)
      timeout 1500 python -m pytest tests/test_reference_gpu.py -m gpu -q --durations=10 > $O/pytest_reference.log 2>&1; echo "reftests rc=$?" >> $O/env.txt; tail -8 $O/pytest_reference.log;;
    report)
      timeout 600 python tools/parity_report.py > $O/parity_report.json 2> $O/parity_report.err; echo "report rc=$?" >> $O/env.txt;;
    refreport)
      timeout 1500 python tools/reference_report.py $REFREPORT_SECTIONS > $O/reference_parity.json 2> $O/reference_parity.err; echo "refreport rc=$?" >> $O/env.txt;;
    bench)
      t0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? seconds=$(( $(date +%s) - t0 ))" >> $O/env.txt; tail -c 600 $O/bench.json;;
    train)
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 \
        bench.py --gpus 1 --mode train --steps 30 --warmup 5 --force-dist > $O/train.json 2> $O/train.err; echo "train rc=$?" >> $O/env.txt; tail -c 1500 $O/train.json;;
    pmc)
      timeout 1200 python tools/pmc_collect.py --workdir /tmp/pmc_fwd --kernel render_fwd_kernel --out $O/pmc_render_fwd.json --marched-from-bench -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-variants > $O/pmc.log 2>&1; echo "pmc rc=$?" >> $O/env.txt; tail -30 $O/pmc.log;;
    pmc_bf16)
      timeout 1200 python tools/pmc_collect.py --workdir /tmp/pmc_bf16 --kernel render_fwd_kernel --out $O/pmc_render_fwd_bf16.json --marched-from-bench -- \
        python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-variants --texels bf16 > $O/pmc_bf16.log 2>&1; echo "pmc_bf16 rc=$?" >> $O/env.txt; tail -12 $O/pmc_bf16.log;;
    pmc_bwd)
      # the field backward kernel inside the training step (4 images x 128x128 rays x (64 + 64) samples: ONE launch since the one-node render)
      timeout 1200 python tools/pmc_collect.py --workdir /tmp/pmc_bwd --kernel field_query_bwd_kernel --out $O/pmc_backward.json --units 8388608 -- \
        python $R/bench.py --mode train --steps 4 --warmup 2 > $O/pmc_bwd.log 2>&1; echo "pmc_bwd rc=$?" >> $O/env.txt; tail -30 $O/pmc_bwd.log;;
    pmc_reduce)
      # the scatter's reduce kernel inside the training step: 4 scenes x 8.4 M points x 3 planes = 25.2 M entries per launch
      timeout 1200 python tools/pmc_collect.py --workdir /tmp/pmc_red --kernel bin_reduce_kernel --out $O/pmc_bin_reduce.json --units 25165824 -- \
        python $R/bench.py --mode train --steps 4 --warmup 2 > $O/pmc_reduce.log 2>&1; echo "pmc_reduce rc=$?" >> $O/env.txt; tail -25 $O/pmc_reduce.log;;
    train_fp16)
      timeout 600 python bench.py --mode train --steps 30 --warmup 5 --texels fp16 > $O/train_fp16.json 2> $O/train_fp16.err; tail -c 700 $O/train_fp16.json
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_fp16 -o tr -- python $R/bench.py --mode train --steps 20 --warmup 5 --texels fp16 > $O/prof_train_fp16.log 2>&1)
      python tools/kstats.py $O/prof_train_fp16 8;;
    bwd)
      timeout 600 python tools/bench_backward.py > $O/bench_backward.log 2>&1; tail -20 $O/bench_backward.log;;
    handoff)
      timeout 600 python tools/bench_handoff.py > $O/bench_handoff.log 2>&1; tail -6 $O/bench_handoff.log;;
    prof_inv)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_inv -o inv -- python $R/tools/inversion_synthetic.py > $O/prof_inv.log 2>&1); tail -3 $O/prof_inv.log
      python - <<PY
import csv, glob
for f in glob.glob("$O/prof_inv/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel ms', tot / 1e6, 'launches', sum(int(r['Calls']) for r in rows))
    for r in rows[:22]:
        print('%8.3f ms %6s calls %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
      ;;
    prof_train)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o tr -- python $R/bench.py --mode train --steps 20 --warmup 5 > $O/prof_train.log 2>&1); tail -c 400 $O/prof_train.log
      python - <<PY
import csv, glob
for f in glob.glob("$O/prof_train/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print('total kernel ms', tot / 1e6, 'launches', sum(int(r['Calls']) for r in rows))
    for r in rows[:22]:
        print('%8.3f ms %6s calls %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:90]))
PY
      ;;
    prof_bench)
      (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-variants > $O/prof_bench.log 2>&1); tail -c 300 $O/prof_bench.log
      python tools/kstats.py $O/prof_bench 8;;
    e2e)
      # end to end on the REAL reference classes (plane producer included), both implementations: tools/end_to_end.py
      timeout 1500 python tools/end_to_end.py > $O/end_to_end.json 2> $O/end_to_end.err; echo "e2e rc=$?" >> $O/env.txt; tail -c 1500 $O/end_to_end.json;;
    e2e_prof)
      # rocprofv3 kernel traces of the end-to-end legs with phase markers, cut into renderer / producer / other
      mkdir -p $O/e2e
      run_leg() {   # name, end_to_end.py arguments...
        n=$1; shift
        (cd /tmp && export TMPDIR=/tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e2e_$n -o t -- \
           python $R/tools/end_to_end.py --markers --sidecar /tmp/e2e_$n.json "$@" > $O/e2e/$n.log 2>&1)
        python tools/phase_split.py /tmp/e2e_$n /tmp/e2e_$n.json > $O/e2e/split_$n.json 2>> $O/e2e/$n.log
        f=$(find /tmp/e2e_$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/e2e/kernel_stats_$n.csv
        python - <<PY
import json
try:
    d = json.load(open("$O/e2e/split_$n.json"))
    print("$n", 'step %.2f ms' % d['step_ms_events_median'], {k: round(v, 3) for k, v in d['group_ms_per_step'].items()})
except Exception as e:
    print("$n", 'FAILED', e)
PY
      }
      run_leg render_b8_hip --leg render --impl hip --batch 8 --iters 20
      run_leg render_b8_hip_bf16 --leg render --impl hip --batch 8 --iters 20 --texels bf16
      run_leg render_b8_reference --leg render --impl reference --batch 8 --iters 10
      run_leg render_b1_hip --leg render --impl hip --batch 1 --iters 20
      run_leg render_b8_hip_channels_last --leg render --impl hip --batch 8 --iters 20 --channels-last
      run_leg render_b8_reference_channels_last --leg render --impl reference --batch 8 --iters 10 --channels-last
      run_leg inversion_hip --leg inversion --impl hip
      run_leg inversion_reference --leg inversion --impl reference
      run_leg gstep_hip --leg gstep --impl hip
      run_leg gstep_hip_fused_handoff --leg gstep --impl hip --fused-handoff
      run_leg gstep_hip_path_length --leg gstep --impl hip --path-length
      run_leg gstep_reference --leg gstep --impl reference
      run_leg gstep_reference_path_length --leg gstep --impl reference --path-length
      ;;
    regulariser)
      timeout 300 python tools/bench_regulariser.py > $O/bench_regulariser.log 2>&1; tail -2 $O/bench_regulariser.log;;
    train_bwd)
      timeout 300 python tools/bench_train_backward.py 30 > $O/bench_train_backward.log 2>&1; tail -4 $O/bench_train_backward.log;;
    scatter_ab)
      # fp32 rows (scatter_mode 1) against fp16 rows (2): the real inputs of a training step, two alternating rounds; then
      # the per-kernel durations of both from one rocprofv3 kernel trace
      NFI_SCATTER_MODES=1,2 timeout 300 python tools/bench_train_backward.py 30 > $O/scatter_ab.log 2>&1; tail -5 $O/scatter_ab.log
      (cd /tmp && export TMPDIR=/tmp && NFI_SCATTER_MODES=1,2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/scatter_ab -o t -- python $R/tools/bench_train_backward.py 20 > $O/scatter_ab_prof.log 2>&1)
      f=$(find /tmp/scatter_ab -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_scatter_ab.csv && head -12 "$f" | cut -c1-150;;
    mall)
      timeout 300 python tools/probes/mall_probe.py > $O/mall_probe.log 2>&1; cat $O/mall_probe.log;;
    stress)
      timeout 900 python tools/repro_stress.py > $O/repro_stress.log 2>&1; tail -12 $O/repro_stress.log;;
  esac
done
cat $O/env.txt
