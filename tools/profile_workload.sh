#!/bin/bash
# Kernel trace by shape of any bench.py workload on the GPU box.  usage: bash tools/profile_workload.sh <tag> <workload> [extra bench args]
tag=$1; wl=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_wl -- python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-extras "$@" > gpurun_out/${tag}_${wl}_bench_line.json 2> gpurun_out/${tag}_${wl}_bench.err
python tools/kernel_trace_by_shape.py gpurun_out/p_wl gpurun_out/${tag}_${wl}_kernel_trace_by_shape.csv > gpurun_out/${tag}_${wl}_trace_summary.txt
rm -rf gpurun_out/p_wl
tail -3 gpurun_out/${tag}_${wl}_trace_summary.txt
