#!/bin/bash
# Kernel trace by shape of the RE10K training step on the GPU box.  usage: bash tools/profile_train.sh <tag> [workload] [batch]
tag=${1:-r02}; wl=${2:-train_re10k}; bs=${3:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_tr -- python bench.py --workload $wl --batch $bs --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_${wl}_bench_line.json 2> gpurun_out/${tag}_${wl}_bench.err
python tools/kernel_trace_by_shape.py gpurun_out/p_tr gpurun_out/${tag}_${wl}_kernel_trace_by_shape.csv > gpurun_out/${tag}_${wl}_trace_summary.txt
rm -rf gpurun_out/p_tr
tail -4 gpurun_out/${tag}_${wl}_trace_summary.txt
