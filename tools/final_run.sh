# One GPU call that produces a profile set under gpurun_out/r2n: the GPU tests, smoke(), the bench line, the bench under
# rocprofv3 --kernel-trace --stats (kernel statistics, per-launch rows of the object encoder) and the launch timeline of one step.
set -e
mkdir -p gpurun_out/r2n
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r2n/tests.log 2>&1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2n/smoke.log 2>&1
timeout -k 10 240 python bench.py > gpurun_out/r2n/bench.json 2> gpurun_out/r2n/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2n/prof -- python3 bench.py --no-extras > gpurun_out/r2n/bench_prof.json 2> gpurun_out/r2n/bench_prof.err
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2n/steps -- python3 tools/run_steps.py > gpurun_out/r2n/steps.log 2>&1
python tools/step_timeline.py gpurun_out/r2n/steps > gpurun_out/r2n/step_timeline.txt
f=$(ls gpurun_out/r2n/prof/*/*kernel_trace.csv | head -n 1)
python tools/trace_rows.py $f "pointnet_mlp_max_bf16_kernel<4>" --last 20 > gpurun_out/r2n/pointnet_launches.csv
cp $(ls gpurun_out/r2n/prof/*/*kernel_stats.csv | head -n 1) gpurun_out/r2n/kernel_stats.csv
rm -rf gpurun_out/r2n/prof gpurun_out/r2n/steps
tail -n 2 gpurun_out/r2n/tests.log; tail -n 1 gpurun_out/r2n/smoke.log; cut -c1-200 gpurun_out/r2n/bench.json; cut -c1-200 gpurun_out/r2n/bench_prof.json; head -n 1 gpurun_out/r2n/step_timeline.txt; tail -n 1 gpurun_out/r2n/pointnet_launches.csv
