cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTEST_TIMEOUT=600
tools/gpu_session.sh r05_b "pytest:liop or test_stage or features" "proftool:liop_fused_perf.py" 
PMC_FILTER=liop PMC_HEAD=12 tools/gpu_session.sh r05_b "pmctool:SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS+SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_SALU+SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_LDS,SQ_INSTS_VMEM_RD:liop_fused_perf.py:120000"
