mkdir -p gpurun_out/r4o
timeout 600 python -m pytest -q -m gpu tests/test_hip_parity.py tests/test_hip_full_size.py -x > gpurun_out/r4o/pytest.log 2>&1; tail -2 gpurun_out/r4o/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r4o/bench.json 2>gpurun_out/r4o/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4o/bench.json").read().strip().splitlines()[-1])
print('value %.1f pipelined %.1f exact %.1f allhit %.1f frac %.3f kernel %s' % (j["value"]/1e6, j["value_pipelined"]/1e6, j["value_mlp_exact_fp32"]/1e6, j["value_all_rays_hit"]/1e6, j["roofline"]["frac"], j["kernel_ms_stats"]))
PY
NFI_PHASES=1 timeout 200 python tools/quick_bench.py 2>&1 | grep -A8 "phase profile" | grep "total\|phase"
