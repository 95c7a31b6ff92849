# Bare matrix-instruction ceiling (tools/micro/mfma_ceiling) next to the real bf16 searches, timed and under one PMC pass each
# (matrix-pipe busy share and clock), in one GPU call.  Output under gpurun_out/$1.
set -e
out=gpurun_out/${1:-ceil}
mkdir -p $out
timeout -k 10 120 tools/micro/mfma_ceiling -1 5 > $out/ceiling_timing.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc_ceiling -- tools/micro/mfma_ceiling -1 3 > $out/pmc_ceiling.log 2>&1
python tools/mfma_pmc_summary.py $(ls $out/pmc_ceiling/*/*counter_collection.csv | head -n 1) ceiling_kernel > $out/pmc_ceiling.txt
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc_search -- python3 tools/mfma_pmc.py > $out/pmc_search.log 2>&1
python tools/mfma_pmc_summary.py $(ls $out/pmc_search/*/*counter_collection.csv | head -n 1) > $out/pmc_search.txt
rm -rf $out/pmc_ceiling $out/pmc_search
cat $out/ceiling_timing.txt $out/pmc_ceiling.txt; grep vq_arg $out/pmc_search.txt
