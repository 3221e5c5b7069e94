cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONLY=L2 ATTN_VARIANTS=2,6,8
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/p1 -- python tools/bench_ops.py attn > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/p2 -- python tools/bench_ops.py attn > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/p3 -- python tools/bench_ops.py attn > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/p4 -- python tools/bench_ops.py attn > /dev/null 2>&1
python tools/pmc_counters.py "attention L2 Bm2: v2 / v3 nomax / ping-pong" gpurun_out/p1 gpurun_out/p2 gpurun_out/p3 gpurun_out/p4 > gpurun_out/r02_f_pmc_attn_variants.json
rm -rf gpurun_out/p1 gpurun_out/p2 gpurun_out/p3 gpurun_out/p4
cat gpurun_out/r02_f_pmc_attn_variants.json
