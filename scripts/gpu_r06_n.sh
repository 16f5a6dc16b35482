#!/bin/bash
# round 6, call N: where the camera-0 pass of the wide variant spends its time (timing-only builds: results invalid)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/gpu_wide_ab.sh 12500 "$@" 2>&1 | grep "^==" | cut -c1-140
