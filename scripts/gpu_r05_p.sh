#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/gpu_r05_t.sh _prev
bash scripts/gpu_r05_o.sh 2>&1 | grep -v "^$" | cut -c1-200
