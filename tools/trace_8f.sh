#!/bin/bash
# kernel trace by shape of one eager 8-frame sample (BASELINE config 2): bash tools/trace_8f.sh <tag> -> gpurun_out/<tag>_kernel_trace_by_shape_8f.csv
tag=${1:-tmp}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --eager > /dev/null 2>&1
python tools/kernel_trace_by_shape.py gpurun_out/p_kt gpurun_out/${tag}_kernel_trace_by_shape_8f.csv | head -${2:-30}
rm -rf gpurun_out/p_kt
