cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
leg() { python tools/end_to_end.py --leg gstep --impl hip --iters 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['ms_median'],2))"; }
ls -la ~/.config/miopen ~/.cache/miopen 2>&1 | head -5
leg fresh_box >> $O/miopen_probe.log
ls ~/.config/miopen 2>&1 | head -5 >> $O/miopen_probe.log
python -m pytest tests/test_reference_gpu.py -m gpu -q -k "training_step" > $O/probe_pytest.log 2>&1; tail -1 $O/probe_pytest.log >> $O/miopen_probe.log
ls -la ~/.config/miopen 2>&1 | head -8 >> $O/miopen_probe.log
leg after_deterministic_tests >> $O/miopen_probe.log
MIOPEN_USER_DB_PATH=/tmp/miopen_fresh_db leg private_user_db >> $O/miopen_probe.log
cat $O/miopen_probe.log
