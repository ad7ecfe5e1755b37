#!/bin/bash
# round 2 evidence on the final build: bench line, kernel traces (graph mode, fp32 + bf16), idle analysis,
# per-layer table, PMC (matrix-pipe busy, memory-side traffic).  Outputs under gpurun_out/ev_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py > $O/ev_bench_n1.out 2> $O/ev_bench_n1.err; grep '^{"metric' $O/ev_bench_n1.out > $O/ev_bench_n1.json; head -c 1500 $O/ev_bench_n1.json; echo
timeout 300 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > $O/ev_conv_layers.log; tail -3 $O/ev_conv_layers.log
cd /tmp
for dt in f32 bf16; do
  rm -rf /tmp/kt_$dt
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$dt -o kt -- python $R/bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline --dtype $dt > /tmp/kt_$dt.log 2>&1
  grep '^{"metric' /tmp/kt_$dt.log > $O/ev_trace_bench_$dt.json
  DB=$(find /tmp/kt_$dt -name "*.db" | head -1)
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline --dtype $dt  (final round-2 build, hipGraph replay over a 16-batch stream: 16 first-pass + 4 warm-up + 20 timed = 40 iterations of kernels)" > $O/ev_kt_$dt.txt
  python $R/tools/prof_summary.py $DB 40 >> $O/ev_kt_$dt.txt 2>&1
  python $R/tools/prof_idle.py $DB > $O/ev_idle_$dt.txt 2>&1
  head -8 $O/ev_kt_$dt.txt; cat $O/ev_idle_$dt.txt | head -4
done
bash $R/tools/pmc_step.sh > $O/ev_pmc_step.log 2>&1; tail -2 $O/ev_pmc_step.log; cp $O/pmc_step.json $O/ev_pmc_step_traffic.json 2>/dev/null
bash $R/tools/pmc_one.sh m4.conv0 > $O/ev_pmc_m4conv0.txt 2>&1; head -30 $O/ev_pmc_m4conv0.txt
bash $R/tools/pmc_traffic.sh m4.conv0 > $O/ev_pmc_traffic_m4conv0.txt 2>&1; tail -5 $O/ev_pmc_traffic_m4conv0.txt
