python tools/bench_configs.py > gpurun_out/cfgs1.log 2>&1; cat gpurun_out/cfgs1.log
