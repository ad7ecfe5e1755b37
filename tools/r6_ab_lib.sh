#!/bin/bash
# step A/B of two builds of the library in ONE call: tools/r6_ab_lib.sh TAG   (lib/libsg2im_hip_TAG.so vs the shipped one)
cd $GRAFT_REPO_ROOT
TAG=$1
for rep in 1 2 3; do for lib in "" "_$TAG"; do
  for dt in f32 bf16; do
    SG2IM_LIB=$GRAFT_REPO_ROOT/sg2im_amd/lib/libsg2im_hip$lib.so python bench.py --steps 50 --warmup 10 --cpu_baseline_steps 0 --no_roofline --dtype $dt 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[lib$lib] $dt coco', d['ms_per_step'])"
  done
done; done
