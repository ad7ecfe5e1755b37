#!/bin/bash
# kernel traces (graph mode) of the fp32 headline step and of the bf16 secondary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for dt in f32 bf16; do
  rm -rf /tmp/kt_$dt
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$dt -o kt -- python $R/bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline --dtype $dt > /tmp/kt_$dt.log 2>&1
  tail -1 /tmp/kt_$dt.log > $R/gpurun_out/prof_bench_$dt.json
  DB=$(find /tmp/kt_$dt -name "*.db" | head -1)
  echo "== $dt (steps incl. one capture pass per bucket: 16 first-pass + 4 warm-up + 20 timed = 40 step-equivalents of kernels)" > $R/gpurun_out/prof_kt_$dt.txt
  python $R/tools/prof_summary.py $DB 40 >> $R/gpurun_out/prof_kt_$dt.txt 2>&1
  head -60 $R/gpurun_out/prof_kt_$dt.txt
  cat $R/gpurun_out/prof_bench_$dt.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
