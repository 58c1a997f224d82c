#!/bin/bash
# A/B runs of bench.py under environment switches (tuning build: make -C ray-tracer_amd/csrc TUNE=1).  usage: tools/ab_bench.sh "A=1" "RT_NO_CONE=1" ...
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python -u bench.py --steps 20 --no-cpu-baseline --eighth-share 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('Mrays/s', d['value'], 'ms/frame', d['ms_per_step'], 'kernels', d['config']['kernels_ms_per_frame'], d['roofline']['per_ray'], 'trace ms', d['roofline']['trace_ms_per_frame'], 'frac', d['roofline']['frac'], 'share8', d['config'].get('one_gpu_of_eight_share'))" || exit 1
done
