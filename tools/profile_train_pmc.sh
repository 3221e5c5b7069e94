#!/bin/bash
# MFMA utilisation (SQ + GRBM passes) of the RE10K training step on the GPU box.  usage: bash tools/profile_train_pmc.sh <tag> [batch]
tag=${1:-r02}; bs=${2:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --workload train_re10k --batch $bs --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/p_sq -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_gr -- $B > /dev/null 2>&1
python tools/pmc_counters.py "RE10K training step, batch $bs, $tag" gpurun_out/p_sq gpurun_out/p_gr > gpurun_out/${tag}_pmc_mfma_util_train_re10k.json
rm -rf gpurun_out/p_sq gpurun_out/p_gr
python - <<PY
import json
u=json.load(open("gpurun_out/${tag}_pmc_mfma_util_train_re10k.json"))["kernels"]
for n,k in sorted(u.items(), key=lambda kv:-kv[1].get("mfma_util",0)):
    if k.get("mfma_util",0)>0.05: print("%-90s util %.3f launches %d" % (n[:90], k["mfma_util"], k.get("launches",0)))
PY
