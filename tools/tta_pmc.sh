# Two counter passes over tools/tta_pmc.py (vector-issue and wait counters, then clock), summarised per kernel.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tta_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/tta_pmc/p1 -- python3 tools/tta_pmc.py > gpurun_out/tta_pmc/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/tta_pmc/p2 -- python3 tools/tta_pmc.py > gpurun_out/tta_pmc/p2.log 2>&1
python tools/pmc_sum.py $(ls gpurun_out/tta_pmc/p1/*/*counter_collection.csv | head -n 1) > gpurun_out/tta_pmc/summary.txt
python tools/pmc_sum.py $(ls gpurun_out/tta_pmc/p2/*/*counter_collection.csv | head -n 1) >> gpurun_out/tta_pmc/summary.txt
rm -rf gpurun_out/tta_pmc/p1 gpurun_out/tta_pmc/p2
cat gpurun_out/tta_pmc/summary.txt
