for kv in "SG2IM_HALO=1" "SG2IM_HALO=1 SG2IM_FUSE_BN=0" "SG2IM_HALO=1 SG2IM_AUX=0" "SG2IM_HALO=1 SG2IM_PAR_DIMG=0" "SG2IM_HALO=1 SG2IM_STAGE=0" "SG2IM_HALO=0 SG2IM_FUSE_BN=0 SG2IM_AUX=0 SG2IM_PAR_DIMG=0 SG2IM_STAGE=0" "SG2IM_HALO=1"; do
  env $kv python bench.py --steps 100 --warmup 20 --cpu_baseline_steps 0 --no_roofline --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['ms_per_step'], d['value'])"
done
