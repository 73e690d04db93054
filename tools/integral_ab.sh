R=$(pwd)
timeout 300 python -m pytest tests/test_templmatch_gpu.py -m gpu -q -k "integral" --timeout 250 2>&1 | tail -4
for nt in 0 2 3; do echo "== MI355CV_INTEGRAL_NT=$nt"; MI355CV_INTEGRAL_NT=$nt timeout 120 python tools/integral_probe.py 2>&1 | grep -v amdgpu; done
