tag=r04g
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --eager"
echo sq; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/p_sq -- $B > /dev/null 2>&1
echo grbm; rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p_gr -- $B > /dev/null 2>&1
python tools/pmc_counters.py "8f default command, $tag" gpurun_out/p_sq gpurun_out/p_gr > gpurun_out/${tag}_pmc_mfma_util_8f.json
rm -rf gpurun_out/p_sq gpurun_out/p_gr
echo fetch; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_f -- $B > /dev/null 2>&1
echo write; rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_w -- $B > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/p_f gpurun_out/p_w "8f default command, $tag" > gpurun_out/${tag}_pmc_hbm_traffic_8f.json
rm -rf gpurun_out/p_f gpurun_out/p_w
echo done
