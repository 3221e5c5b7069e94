#!/bin/bash
# Round profile of the default workload (BASELINE config 2) on the GPU box: kernel trace by shape, MFMA utilisation (SQ + GRBM passes)
# and HBM traffic (FETCH / WRITE passes, collected separately).  usage: bash tools/profile_8f.sh <tag>   -> gpurun_out/<tag>_*
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --eager"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p_kt -- $B > /dev/null 2>&1
python tools/kernel_trace_by_shape.py gpurun_out/p_kt gpurun_out/${tag}_kernel_trace_by_shape_8f.csv | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_st -- $B > /dev/null 2>&1
find gpurun_out/p_st -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats_default_command.csv \;
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/p_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_gr -- $B > /dev/null 2>&1
python tools/pmc_counters.py "8f default command, $tag" gpurun_out/p_sq gpurun_out/p_gr > gpurun_out/${tag}_pmc_mfma_util_8f.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_f -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_w -- $B > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/p_f gpurun_out/p_w "8f default command, $tag" > gpurun_out/${tag}_pmc_hbm_traffic_8f.json
rm -rf gpurun_out/p_kt gpurun_out/p_st gpurun_out/p_sq gpurun_out/p_gr gpurun_out/p_f gpurun_out/p_w
python - <<PY
import json
u=json.load(open("gpurun_out/${tag}_pmc_mfma_util_8f.json"))["kernels"]
t=json.load(open("gpurun_out/${tag}_pmc_hbm_traffic_8f.json"))["kernels"]
for n,k in u.items():
    if k.get("mfma_util",0)>0.05: print("%-90s util %.3f coexec %.2f" % (n[:90], k["mfma_util"], k.get("mfma_valu_coexec_frac_of_mfma_busy",0)))
for n,k in t.items():
    if "attn" in n: print(n[:90], k)
PY
