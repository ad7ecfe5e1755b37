#!/bin/bash
# rocprofv3 kernel trace of the graph-mode training step: per-kernel summary, kernel-by-kernel sequence of one
# steady-state step, fill timeline.   tools/trace_step.sh OUT_PREFIX [bench.py flags / ENV=val ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$1; shift
envs=(); flags=()
for a in "$@"; do if [[ "$a" == *=* && "$a" != --* ]]; then envs+=("$a"); else flags+=("$a"); fi; done
rm -rf /tmp/kt_run
env "${envs[@]}" rocprofv3 --kernel-trace --stats -d /tmp/kt_run -o kt -- python $R/bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline "${flags[@]}" > /tmp/kt_run.log 2>&1
grep '^{"metric' /tmp/kt_run.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under the profiler: %.3f ms/step' % d['ms_per_step'])"
DB=$(find /tmp/kt_run -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- ${envs[*]} python bench.py --steps 20 --warmup 4 --cpu_baseline_steps 0 --no_roofline ${flags[*]}   (hipGraph replay over a 16-batch stream: 16 first-pass + 4 warm-up + 20 timed = 40 iterations of kernels)" > $R/${out}_kernel_trace.txt
python $R/tools/prof_summary.py $DB 40 >> $R/${out}_kernel_trace.txt 2>&1
python $R/tools/prof_step_dump.py $DB > $R/${out}_kernel_sequence.txt 2>&1
python $R/tools/prof_timeline.py $DB > $R/${out}_fill_timeline.txt 2>&1
python $R/tools/prof_idle.py $DB > $R/${out}_idle.txt 2>&1
head -3 $R/${out}_kernel_trace.txt; head -6 $R/${out}_fill_timeline.txt
