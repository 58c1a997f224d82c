#!/bin/bash
# tuning build: knobs that shape the end of a trace launch (grid size, donation thresholds)
bash tools/ab_bench.sh "RT_TRACE_BLOCK_CAP=1024" "RT_TRACE_BLOCK_CAP=768" "RT_TRACE_BLOCK_CAP=512" "RT_DONATE_MIN_IDLE=16" "RT_DONATE_MIN_IDLE=8" "RT_DONATE_MIN_IDLE=48" "RT_TAIL_WINDOW=0" "RT_TAIL_WINDOW=4" "RT_TRI_MIN_LANES=6 RT_DONATE_MIN_IDLE=16"
