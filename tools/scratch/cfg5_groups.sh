#!/bin/bash
# the launch groups of the config-5 backward (HBK_BWD_TRACE) and the order of the groups
cd ${GRAFT_REPO_ROOT:-/root/repo}
HBK_BWD_TRACE=1 timeout 300 python tools/sweep.py --big --cases h 2>&1 | grep "hbk bwd" | sort | uniq -c | sort -k4n | head -20
source tools/gpu_r5.sh "none" > /dev/null 2>&1
ab "bwd_large_first:0,1,0,1" h
