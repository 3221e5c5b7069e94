#!/bin/bash
# same-box A/B of DFOT_* tuning flags on the 8-frame headline workload: every argument is one env setting ("NAME=VALUE[,NAME=VALUE...]" or "base"),
# run in the given order, twice (interleaved rounds), bench lines appended to gpurun_out/ab_8f.log
mkdir -p gpurun_out
for round in 1 2; do
  for cfg in "$@"; do
    envs=""
    if [ "$cfg" != "base" ]; then envs=$(echo "$cfg" | tr ',' ' '); fi
    line=$(env $envs python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 ${BENCH_ARGS} | tail -1)
    echo "$cfg round $round: $(echo "$line" | python -c 'import json,sys; j=json.loads(sys.stdin.read()); print("%.3f frames/s, %.2f ms/step, attn %.1f us, mode %s" % (j["value"], j["ms_per_step"], 1e3*j["roofline"]["avg_launch_ms"], j["sampler_mode"]))')" | tee -a gpurun_out/ab_8f.log
  done
done
