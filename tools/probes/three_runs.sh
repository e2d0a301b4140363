cd $GRAFT_REPO_ROOT
for i in 12 13 14; do
  mkdir -p gpurun_out/r7g
  timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r7g/pytest_run$i.log 2>&1; echo "run $i rc=$?"; tail -1 gpurun_out/r7g/pytest_run$i.log
done
