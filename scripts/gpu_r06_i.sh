#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python scripts/time_ba_default.py 2>&1 | tail -12
nproc; cat /proc/loadavg
