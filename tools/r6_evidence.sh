#!/bin/bash
# Round-6 evidence of the final build (one gpurun call).  Outputs under gpurun_out/ev6/ - copied into profiles/r6_* afterwards.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev6
mkdir -p $O
cd $R
bash tools/pmc_step.sh > $O/pmc_step.log 2>&1; tail -1 $O/pmc_step.log > $O/pmc_step_traffic.json; cat $O/pmc_step_traffic.json
python -c "import json; json.load(open('$O/pmc_step_traffic.json'))" && cp $O/pmc_step_traffic.json $R/profiles/r6_pmc_step_traffic.json
timeout 900 python bench.py > $O/bench_n1.out 2> $O/bench_n1.err; grep '^{"metric' $O/bench_n1.out > $O/bench_n1.json; head -c 600 $O/bench_n1.json; echo
tools/trace_step.sh gpurun_out/ev6/step
SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks.txt; cat $O/schedule_marks.txt
D6=1024,512,256,128,64,64
run() { n=$1; shift
  timeout 900 python bench.py --cpu_baseline_steps 0 "$@" > $O/$n.out 2> $O/$n.err
  grep '^{"metric' $O/$n.out > $O/$n.json
  python -c "
import json
d = json.load(open('$O/$n.json')); r = d['roofline'] or {}
print('$n', d['ms_per_step'], 'ms/step', d['value'], 'img/s', 'family', r.get('achieved'), r.get('frac'), 'crn', (r.get('crn_only') or {}).get('tflops'), (r.get('crn_only') or {}).get('frac'))"
}
for dt in f32 bf16; do
  run bench_coco64_$dt --dtype $dt --steps 50 --warmup 10
  run bench_vg64_$dt --style vg --dtype $dt --steps 50 --warmup 10
  run bench_vg128_$dt --style vg --dtype $dt --image_size 128 --refinement_dims $D6 --steps 20 --warmup 5 --n_batches 8
  run bench_vg128_5mod_$dt --style vg --dtype $dt --image_size 128 --steps 20 --warmup 5 --n_batches 8
  run bench_s256_$dt --style vg --dtype $dt --image_size 256 --refinement_dims $D6 --min_objs 10 --max_objs 29 --extra_rels 60 --steps 10 --warmup 3 --n_batches 4
done 2>&1 | tee $O/bench_configs.log
{
for v in "" "--dtype bf16"; do
  for fd in "--force_dist" "--force_dist --dp_schedule 1" "--force_dist --dp_schedule 0"; do
    timeout 300 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline $v $fd 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v $fd]', d['ms_per_step'], 'ms/step', d['value'], 'img/s', (d.get('gradient_exchange') or {}).get('payload_dtype'), 'schedule', (d.get('gradient_exchange') or {}).get('dp_schedule'))"
  done
done
timeout 300 python bench.py --steps 40 --warmup 10 --cpu_baseline_steps 0 --no_roofline --eval_generator 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [--eval_generator]', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu_baseline_steps 0 --no_roofline --no_graphs 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [--no_graphs]', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
} > $O/bench_variants.log 2>&1; cat $O/bench_variants.log
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers.log; tail -1 $O/conv_layers.log
timeout 300 python tools/bench_conv.py --bf16 2>&1 | grep -v amdgpu.ids > $O/conv_layers_bf16.log; tail -1 $O/conv_layers_bf16.log
SG2IM_MARKS=1 python bench.py --steps 60 --warmup 20 --cpu_baseline_steps 0 --no_roofline --dtype bf16 2>&1 >/dev/null | grep '\[mark\]' > $O/schedule_marks_bf16_coco.txt
tools/trace_step.sh gpurun_out/ev6/step_bf16 --dtype bf16 > /dev/null 2>&1
timeout 300 python tools/bench_layout.py 2>&1 | grep -v amdgpu.ids > $O/layout_kernels.log; cat $O/layout_kernels.log
cd $R && timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
