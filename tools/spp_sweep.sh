#!/bin/bash
# the config-2 frame at 4 and 16 samples per pixel (one GPU), and the 1/8 tile share of the 1-spp and the 16-spp frame
for spp in 4 16; do
  timeout -k 10 300 python3 -u bench.py --scaling strong --spp $spp --steps 5 --warmup 2 --no-cpu-baseline --no-frames-in-flight --eighth-share 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('spp', d['config']['spp'], 'Mrays/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'share8', d['config']['one_gpu_of_eight_share'])" || exit 1
done
timeout -k 10 300 python3 -u bench.py --steps 10 --no-cpu-baseline --no-frames-in-flight --eighth-share 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('spp', d['config']['spp'], 'Mrays/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'share8', d['config']['one_gpu_of_eight_share'])"
