# GPU tests, then the launch timeline of one generation step (rocprofv3 kernel trace of tools/run_steps.py).  Output under gpurun_out/$1.
set -e
out=gpurun_out/${1:-tl}
mkdir -p $out
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests.log 2>&1 || { tail -n 30 $out/tests.log; exit 1; }
tail -n 2 $out/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $out/steps -- python3 tools/run_steps.py > $out/steps.log 2>&1
python tools/step_timeline.py $out/steps > $out/step_timeline.txt
rm -rf $out/steps
head -n 1 $out/step_timeline.txt; grep -v "prior_\|dvq::prior" $out/step_timeline.txt | head -n 60
