#!/bin/bash
# The whole GPU suite + smoke on one box; the summary lands in gpurun_out/suite/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/suite; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
