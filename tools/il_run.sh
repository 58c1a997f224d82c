# sweep of the interleave threshold (tuning build) at three launch sizes
set -e
P='import json,sys; d=json.loads(sys.stdin.read()); print("Mrays/s", d["value"], "ms/frame", d["ms_per_step"], "trace ms", d["roofline"]["trace_ms_per_frame"], "frac", d["roofline"]["frac"])'
for v in "A=1" "RT_INTERLEAVE_MAX_SHARE=512" "RT_INTERLEAVE_MAX_SHARE=1024" "RT_INTERLEAVE_MAX_SHARE=2048" "RT_INTERLEAVE_MAX_SHARE=4096"; do
  for spp in 1 4 16; do
    echo "== $v  spp $spp"
    env RT_INTERLEAVE_SHIFT=0 $v timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline --spp $spp --scaling strong 2>/dev/null | tail -1 | python3 -c "$P"
  done
done
