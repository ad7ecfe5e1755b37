#!/bin/bash
# final check of a build: the whole GPU suite, __graft_entry__.smoke(), default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/final_pytest_gpu.log 2>&1; grep -a "passed\|failed\|error" gpurun_out/final_pytest_gpu.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/final_smoke.log
timeout 600 python bench.py 2>/dev/null | grep '^{"metric' > gpurun_out/final_bench.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['cpu_baseline'])"
