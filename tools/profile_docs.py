"""Writes profiles/<round>_bench_kernel_stats.md, profiles/<round>_pmc_trace_kernel.md and profiles/traffic_latest.json from the
outputs of tools/profile_round.sh <tag> (merged back under gpurun_out/).  Optional: the directory tag of tools/pmc_extra.sh and a file with a
later default bench line of the same library.  usage: python tools/profile_docs.py <tag> <round> [<extra_pmc_tag> [<bench_json>]]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
ks = [l for l in open(os.path.join(G, tag + "_kernel_stats.md")).read().splitlines() if l.startswith("|")][:24]
bench = open(os.path.join(G, tag + "_bench.json")).read().strip().splitlines()[-1]
trace = [l for l in ks if "k_trace_generic<false, false, RenderIO>" in l][0].split("|")
avg_us = float(trace[4])
open(os.path.join(ROOT, "profiles", rnd + "_bench_kernel_stats.md"), "w").write("""# Round %s — rocprofv3 kernel trace of bench.py, final library of the round (MI355X, 1 GPU)

Command (on the GPU box, tools/profile_round.sh %s): `cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d gpurun_out/%s_kt -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-frames-in-flight`

tools/rocpd_summary.py over the `kernels` view of the rocpd database (5 timed + 1 warm-up + 1 counting frame = 7 frames; the
`<true, ...>` instantiation is the counting frame; the k_bin / k_reduce_boxes / k_split / k_collapse rows are the one-off BVH build
of the 1M-triangle scene).  The traversal is the exact one (normal-cone pruning rule) with interleaved small launches; the frame makes
no host round trip between ray generation and resolve.  Average `k_trace_generic<false,false,RenderIO>` launch: **%.1f us**
(bench.py's own HIP-event figure, `roofline.avg_launch_ms`, in the un-profiled run below).

%s

bench.py line of the same library (un-profiled run, `python3 bench.py`, defaults; `traffic` is null in this line because the counter
passes that produce profiles/traffic_latest.json ran after it):

```
%s
```
""" % (rnd[1:], tag, tag, avg_us, "\n".join(ks), bench))

summ = open(os.path.join(G, tag + "_pmc_summary.txt")).read()
blk = summ[summ.index("== k_trace\n"):]
blk = blk[:blk.index("\n==", 3)] if "\n==" in blk[3:] else blk
c = {}
for l in blk.splitlines():
    p = l.split()
    if len(p) >= 2 and p[1].isdigit():
        c[p[0]] = int(p[1])
frames = 2
rays = 9313387
cyc = c["GRBM_GUI_ACTIVE"] / 8 / frames
d = dict(cyc=cyc / 1e6, valu=c["SQ_INSTS_VALU"] / frames / 1e9, lanes=c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
         wait=100.0 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], waiti=100.0 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], act=100.0 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
         l2=100.0 * c["TCC_HIT"] / (c["TCC_HIT"] + c["TCC_MISS"]), ta=100.0 * c["TA_TA_BUSY"] / 256 / (cyc * frames), tas=100.0 * c["TA_ADDR_STALLED_BY_TC_CYCLES"] / 256 / (cyc * frames),
         acc=c["TCP_TOTAL_CACHE_ACCESSES"] / frames / 1e9, accr=c["TCP_TOTAL_CACHE_ACCESSES"] / frames / rays, vb=100.0 * c["SQ_INSTS_VALU"] * 4 / (1024 * cyc * frames),
         fetch=c["FETCH_SIZE"] * 1024 / frames / 1e9, write=c["WRITE_SIZE"] * 1024 / frames / 1e9)
d["per_launch"] = (2 * d["fetch"] + d["write"]) / 8
open(os.path.join(ROOT, "profiles", rnd + "_pmc_trace_kernel.md"), "w").write("""# Round %s — hardware counters of the trace kernel (rocprofv3 --pmc, one pass per counter group)

Command per pass (tools/pmc_passes.sh, run by tools/profile_round.sh %s): `timeout -s KILL 150 rocprofv3 --pmc <counters> --output-format csv -d <dir> -o p -- python3 tools/trace_driver.py 2`
(2 frames of config 2 = 16 `k_trace_generic` launches; sums over all launches below, tools/pmc_summary.py).  Library: the release build
whose SHA-256 is recorded in profiles/traffic_latest.json (exact traversal with normal cones, interleaved small launches).

```
%s
```

Derived, per frame (tools/profile_docs.py): %.2f M GPU cycles busy; %.2f G wave-level VALU instructions = %.0f %% of the issue slots of 1024 SIMDs
at 4 cycles per wave64 instruction (the figure a probe build confirms: 32 extra v_nop per loop iteration, 6.7 %% more instructions, cost 4.5 %% of the trace time); active lanes per VALU instruction SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.1f of 64; of the
waves' cycles %.0f %% execute instructions, %.0f %% wait for memory counters (SQ_WAIT_ANY), %.0f %% wait for issue (SQ_WAIT_INST_ANY); L2 hit rate
TCC_HIT / (HIT + MISS) = %.1f %%; vector-cache accesses %.2f G = %.0f per ray (8 per node step + 3 per primitive test);
texture addresser busy %.0f %% of the launch, stalled by the cache %.0f %%; FETCH_SIZE %.2f GB raw = %.2f GB after the gfx950 correction
(x2: 128-byte requests are tallied at 64 B) + WRITE_SIZE %.2f GB per frame -> %.2f GB per launch (`roofline.traffic`).

Against the collection of round 2 (profiles/r02_pmc_trace_kernel.md, same exact traversal, 32.9 node steps per ray): 4.43 G VALU instructions
per frame then, i.e. 644 per loop iteration of a wave against about 475 now (the loop lost its second back edge, loads go through buffer
descriptors whose range check replaces the zero-initialised merge registers, the unit-matrix object ray is prepared at refill).

What limits the kernel: VALU instruction issue at about half the lanes active.  A wave iteration is one node step for the lanes at a node
(45 of 64 on average) plus, in 73 %% of the iterations, one primitive test for the lanes that hold a pending leaf (14 of 64); the last fifth
of all iterations runs after a launch's queue is empty, at 33 of 64 lanes.  The memory side is loaded but not at its limit: HBM traffic %.2f GB per launch is about 40 %% of the 8 TB/s peak over the launch's duration,
the texture addresser is busy 64 %% of the time, and the time per node step does not depend on the scene's footprint (tools/size_sweep.sh).
""" % (rnd[1:], tag, blk.strip(), d["cyc"], d["valu"], d["vb"], d["lanes"], d["act"], d["wait"], d["waiti"], d["l2"], d["acc"], d["accr"], d["ta"], d["tas"],
       d["fetch"], 2 * d["fetch"], d["write"], d["per_launch"], d["per_launch"]))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "traffic_json.py"), os.path.join(G, tag + "_pmc"), "2"])
if len(sys.argv) > 3:
    xs = open(os.path.join(G, sys.argv[3] + "_summary.txt")).read()
    xb = xs[xs.index("== k_trace\n"):]
    xb = xb[:xb.index("\n==", 3)] if "\n==" in xb[3:] else xb
    x = {}
    for l in xb.splitlines():
        q = l.split()
        if len(q) >= 2 and q[1].isdigit():
            x[q[0]] = int(q[1])
    open(os.path.join(ROOT, "profiles", rnd + "_pmc_trace_kernel.md"), "a").write("""
## Two more passes: instruction fetch, instruction cache, LDS, scalar unit (tools/pmc_extra.sh, same driver and library)

```
%s
```

Per frame: %.2f G instruction fetches, instruction-cache misses %.0f k (the 18 KB loop stays resident), LDS bank conflicts %.1f M cycles against
%.0f M cycles of LDS activity, waves waiting for the LDS %.1f %% of their cycles, scalar instructions %.2f G (%.2f per VALU instruction).
VALU issue %.0f %% of the SIMD cycles, texture addresser %.0f %%, scalar and LDS far below: the vector ALU's issue slots are what the loop runs out of.
""" % (xb.strip(), x["SQ_IFETCH"] / frames / 1e9, x["SQC_ICACHE_MISSES"] / frames / 1e3, x["SQ_LDS_BANK_CONFLICT"] / frames / 1e6, x["SQ_LDS_IDX_ACTIVE"] / frames / 1e6,
       100.0 * x["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], x["SQ_INST_CYCLES_SALU"] / frames / 1e9, x["SQ_INST_CYCLES_SALU"] / c["SQ_INSTS_VALU"], d["vb"], d["ta"]))
if len(sys.argv) > 4:
    later = open(sys.argv[4]).read().strip().splitlines()[-1]
    open(os.path.join(ROOT, "profiles", rnd + "_bench_kernel_stats.md"), "a").write("""
A later default run of the same library (`roofline.traffic` now filled from profiles/traffic_latest.json, whose SHA-256 matches the library):

```
%s
```
""" % later)
print("avg trace launch %.1f us; derived:" % avg_us, {k: round(v, 3) for k, v in d.items()})
