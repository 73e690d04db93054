timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu6.log
