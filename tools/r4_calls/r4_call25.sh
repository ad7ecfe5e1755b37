#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call25.log
: > $L
KS="shared_pass or in_graph_exchange or bit_reproducible"
KL="shared_pass or conv_forward_dgrad or trainer_two_steps or graph_replay or golden or bit_reproducible or in_graph_exchange or without_a_discriminator or build_cnn_arch"
echo "=== short" >> $L
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "$KS" > gpurun_out/r4_call25_short.txt 2>&1; rc=$?
echo "short rc=$rc" >> $L
if [ $rc -eq 139 ]; then K="$KS"; else K="$KL"; fi
echo "=== rocgdb on: $K" >> $L
timeout 1200 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop nopass" -ex run -ex "bt 40" -ex "info sharedlibrary" --args python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > gpurun_out/r4_call25_gdb.txt 2>&1
grep -n "SIGSEGV" -A45 gpurun_out/r4_call25_gdb.txt | head -80 >> $L
tail -5 gpurun_out/r4_call25_gdb.txt >> $L
cat $L | cut -c1-220
