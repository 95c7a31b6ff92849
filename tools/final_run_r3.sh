# One GPU call that produces a profile set under gpurun_out/$1 (default r3c): GPU tests, smoke(), the bench line with the
# driver's flags, the bench under rocprofv3 --kernel-trace --stats (kernel statistics; per-launch rows of the object encoder in both
# precisions and of the hand layer), the launch timeline of one step, FETCH_SIZE / WRITE_SIZE of the HBM-facing kernels
# (one --pmc pass each) and the two-rank gloo rehearsal of `bench.py --gpus 2` on the one GPU.
set -e
out=gpurun_out/${1:-r3c}
mkdir -p $out
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests.log 2>&1 || { tail -n 30 $out/tests.log; exit 1; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 20 --warmup 5 --no-extras > $out/bench_prof.json 2> $out/bench_prof.err
f=$(ls $out/prof/*/*kernel_trace.csv | head -n 1)
python tools/trace_rows.py $f "pointnet_mlp_max_bf16_kernel<4>" --last 80 > $out/launches_bf16.csv
python tools/trace_rows.py $f "mano_lbs_kernel" --last 40 > $out/launches_mano.csv
cp $(ls $out/prof/*/*kernel_stats.csv | head -n 1) $out/kernel_stats.csv
rm -rf $out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof32 -- python3 bench.py --steps 20 --warmup 5 --no-extras --dtype fp32 > $out/bench_prof_fp32.json 2> $out/bench_prof_fp32.err
f=$(ls $out/prof32/*/*kernel_trace.csv | head -n 1)
python tools/trace_rows.py $f "pointnet_mlp_max_kernel<0, 4>" --last 80 > $out/launches_fp32.csv
rm -rf $out/prof32
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/steps -- python3 tools/run_steps.py > $out/steps.log 2>&1
python tools/step_timeline.py $out/steps > $out/step_timeline.txt
rm -rf $out/steps
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -- python3 tools/pn_pmc.py > $out/pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python3 tools/pn_pmc.py > $out/pmc_w.log 2>&1
python tools/pmc_summary.py $(ls $out/pmc_f/*/*counter_collection.csv | head -n 1) $(ls $out/pmc_w/*/*counter_collection.csv | head -n 1) > $out/pmc_fetch_write.txt
rm -rf $out/pmc_f $out/pmc_w
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --batch 1024 > $out/gloo2_bench.json 2> $out/gloo2_bench.err
tail -n 2 $out/tests.log; tail -n 1 $out/smoke.log; cut -c1-300 $out/bench.json; cut -c1-200 $out/bench_prof.json; cut -c1-200 $out/bench_prof_fp32.json; head -n 1 $out/step_timeline.txt; tail -n 1 $out/launches_bf16.csv; tail -n 1 $out/launches_mano.csv; tail -n 2 $out/launches_fp32.csv; grep pointnet $out/pmc_fetch_write.txt; cut -c1-300 $out/gloo2_bench.json
