#!/bin/bash
# lane-occupancy / fallback counters of the traversal kernel for one counted frame of config 2
RT_TRACE_DEBUG=1 timeout -k 10 300 python3 -u tools/trace_driver.py 2 2>&1 | grep -E "^\[rt\] trace lanes|^size|^counted" | head -8
