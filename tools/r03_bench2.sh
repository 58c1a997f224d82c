#!/bin/bash
# default bench line + a two-rank gloo rehearsal of the N > 1 code path of bench.py on the one GPU of the box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -20 gpurun_out/r03_bench_default.err; exit 1; }
cat gpurun_out/r03_bench_default.json
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo > gpurun_out/r03_bench_rehearsal2.json 2> gpurun_out/r03_bench_rehearsal2.err || { tail -30 gpurun_out/r03_bench_rehearsal2.err; exit 1; }
cat gpurun_out/r03_bench_rehearsal2.json
