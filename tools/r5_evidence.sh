#!/bin/bash
# Round-5 evidence of the final build (one gpurun call): PMC traffic of the step FIRST (bench.py reads the newest
# profiles/r*_pmc_step_traffic.json for roofline.traffic), the bench line, the kernel trace of the graph-mode step
# (summary / FULL kernel-by-kernel sequence / fill / idle), schedule marks, secondary lines (bf16 COCO / VG, fp32 VG,
# --force_dist fp32 and bf16), per-layer table, layout kernels, persistent-kernel probe, smoke.
# Outputs under gpurun_out/ev5/ - copied into profiles/r5_* afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev5
mkdir -p $O
cd $R
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -1 $O/pmc_step.log > $O/pmc_step_traffic.json; cat $O/pmc_step_traffic.json
python -c "import json; json.load(open('$O/pmc_step_traffic.json'))" && cp $O/pmc_step_traffic.json $R/profiles/r5_pmc_step_traffic.json
timeout 900 python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{"metric' $O/bench_n1.out > $O/bench_n1.json; head -c 700 $O/bench_n1.json; echo
tools/trace_step.sh gpurun_out/ev5/step
SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks.txt; cat $O/schedule_marks.txt
for st in coco vg; do
  timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --dtype bf16 --style $st 2>/dev/null | grep '^{"metric' > $O/bench_bf16_$st.json
  python -c "import json; d=json.load(open('$O/bench_bf16_$st.json')); print('bf16 $st', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['crn_only'])"
done
timeout 300 python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --style vg 2>/dev/null | grep '^{"metric' > $O/bench_f32_vg.json
python -c "import json; d=json.load(open('$O/bench_f32_vg.json')); print('f32 vg', d['ms_per_step'], d['value'])"
{
for v in "" "--dtype bf16"; do
  for fd in "" "--force_dist" "--force_dist --dp_schedule 0"; do
    timeout 300 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline $v $fd 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v $fd]', d['ms_per_step'], 'ms/step', d['value'], 'img/s', (d.get('gradient_exchange') or {}).get('payload_dtype'))"
  done
done
timeout 300 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline --eval_generator 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [--eval_generator]', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --no_graphs 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [--no_graphs]', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
} > $O/bench_variants.log 2>&1; cat $O/bench_variants.log
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers.log; tail -2 $O/conv_layers.log
timeout 300 python tools/bench_conv.py --bf16 2>&1 | grep -v amdgpu.ids > $O/conv_layers_bf16.log; tail -1 $O/conv_layers_bf16.log
for cfg in "bf16 coco" "f32 vg" "bf16 vg"; do set -- $cfg
  SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline --dtype $1 --style $2 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks_$1_$2.txt
done
tools/trace_step.sh gpurun_out/ev5/step_bf16_vg --dtype bf16 --style vg > /dev/null 2>&1
timeout 300 python tools/bench_layout.py 2>&1 | grep -v amdgpu.ids > $O/layout_kernels.log; cat $O/layout_kernels.log
timeout 300 python tools/gcn_stack_probe.py 2>&1 | grep -v amdgpu.ids > $O/gcn_stack_probe.log; head -4 $O/gcn_stack_probe.log
cd $R && timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
