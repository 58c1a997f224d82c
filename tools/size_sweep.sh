# time per node step against scene size (cache footprint): config-2 camera, stand-in mesh of 62k ... 4M triangles
set -e
P='import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; n=d["config"]["rays_per_frame"]; print("tris", sys.argv[1], "Mrays/s", d["value"], "trace ms", r["trace_ms_per_frame"], "nodes/ray", r["per_ray"]["nodes"], "tris/ray", r["per_ray"]["tris"], "ns per ray-step %.4f" % (r["trace_ms_per_frame"]*1e6/(n*(r["per_ray"]["nodes"]))), "bvh nodes", d["config"]["bvh_nodes"])'
for t in 62500 250000 1000000 4000000; do
  timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --tris $t 2>/dev/null | tail -1 | python3 -c "$P" $t
done
