#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r4_call26.log
: > $L
K="shared_pass or conv_forward_dgrad or trainer_two_steps or graph_replay or golden or bit_reproducible or in_graph_exchange or without_a_discriminator or build_cnn_arch"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > gpurun_out/r4_call26_plain.txt 2>&1; rc=$?
echo "plain rc=$rc" >> $L; grep -v "^  File\|^$" gpurun_out/r4_call26_plain.txt | tail -4 >> $L
if [ $rc -eq 139 ]; then
  echo "=== rocgdb + MALLOC_CHECK_" >> $L
  MALLOC_CHECK_=3 MALLOC_PERTURB_=165 timeout 1200 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop nopass" -ex "handle SIGABRT stop nopass" -ex run -ex "bt 30" --args python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > gpurun_out/r4_call26_gdb.txt 2>&1
  grep -n "received signal" -A32 gpurun_out/r4_call26_gdb.txt | head -60 >> $L
fi
cat $L | cut -c1-220
