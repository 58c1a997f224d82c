#!/bin/bash
# tuning build: sweep of the lane threshold of the primitive path
bash tools/ab_bench.sh "RT_TRI_MIN_LANES=10" "RT_TRI_MIN_LANES=6" "RT_TRI_MIN_LANES=14" "RT_TRI_MIN_LANES=18" "RT_TRI_MIN_LANES=24" "RT_TRI_MIN_LANES=32" "RT_TRI_MIN_LANES=10"
