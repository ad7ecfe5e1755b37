#!/bin/bash
# Round-4 evidence of the final build, second take (one gpurun call): PMC traffic of the step FIRST (bench.py reads the
# newest profiles/r*_pmc_step_traffic.json for roofline.traffic), then the bench line, the kernel trace of the
# graph-mode step (summary / sequence / fill), schedule marks, secondary lines, per-layer table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev4b
mkdir -p $O
cd $R
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -1 $O/pmc_step.log > $O/pmc_step_traffic.json; cat $O/pmc_step_traffic.json
python -c "import json; json.load(open('$O/pmc_step_traffic.json'))" && cp $O/pmc_step_traffic.json $R/profiles/r4_pmc_step_traffic.json
timeout 900 python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{"metric' $O/bench_n1.out > $O/bench_n1.json; head -c 700 $O/bench_n1.json; echo
tools/trace_step.sh gpurun_out/ev4b/step
SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks.txt; cat $O/schedule_marks.txt
for st in coco vg; do
  timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --dtype bf16 --style $st 2>/dev/null | grep '^{"metric' > $O/bench_bf16_$st.json
  python -c "import json; d=json.load(open('$O/bench_bf16_$st.json')); print('bf16 $st', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['crn_only'])"
done
timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --style vg 2>/dev/null | grep '^{"metric' > $O/bench_f32_vg.json
python -c "import json; d=json.load(open('$O/bench_f32_vg.json')); print('f32 vg', d['ms_per_step'], d['value'])"
timeout 300 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline --force_dist 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('force_dist f32', d['ms_per_step'], d['value'])"
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers.log; tail -2 $O/conv_layers.log
