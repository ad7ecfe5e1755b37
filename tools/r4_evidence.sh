#!/bin/bash
# Round-4 evidence on the final build (one gpurun call): bench line, per-layer table, kernel trace of the graph-mode
# step (summary / sequence / fill), PMC (memory-side traffic of the step and of m4.conv0, matrix-pipe busy of m4.conv0),
# bf16 secondary lines.  Outputs under gpurun_out/ev4/ - copied into profiles/r4_* afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev4
mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{"metric' $O/bench_n1.out > $O/bench_n1.json; head -c 600 $O/bench_n1.json; echo
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers.log; tail -2 $O/conv_layers.log
tools/trace_step.sh gpurun_out/ev4/step
SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks.txt; cat $O/schedule_marks.txt
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -2 $O/pmc_step.log; cp $R/gpurun_out/pmc_step.json $O/pmc_step_traffic.json 2>/dev/null
bash tools/pmc_traffic.sh m4.conv0 > $O/pmc_traffic_m4conv0.txt 2>&1; tail -8 $O/pmc_traffic_m4conv0.txt
bash tools/pmc_one.sh m4.conv0 > $O/pmc_m4conv0.txt 2>&1; head -40 $O/pmc_m4conv0.txt
for st in coco vg; do
  timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --dtype bf16 --style $st 2>/dev/null | grep '^{"metric' > $O/bench_bf16_$st.json
  python -c "import json; d=json.load(open('$O/bench_bf16_$st.json')); print('bf16 $st', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['crn_only'])"
done
timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --style vg 2>/dev/null | grep '^{"metric' > $O/bench_f32_vg.json
python -c "import json; d=json.load(open('$O/bench_f32_vg.json')); print('f32 vg', d['ms_per_step'], d['value'])"
timeout 300 python tools/gcn_stack_probe.py 2>&1 | grep -v amdgpu.ids > $O/gcn_stack_probe.log; head -3 $O/gcn_stack_probe.log
cd $R && timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 300 python tools/bench_layout.py 2>&1 | grep -v amdgpu.ids > $O/layout_kernels.log; cat $O/layout_kernels.log
