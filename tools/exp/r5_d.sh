cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_tl -o t -- python3 tools/exp/step_prof.py joint 1024 12 > gpurun_out/r5_tl.log 2>&1
f=$(find gpurun_out/r5_tl -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f 3 > gpurun_out/r5_timeline.txt 2>&1
head -3 gpurun_out/r5_timeline.txt
rm -rf gpurun_out/r5_tl
for v in 1 2; do timeout -k 10 120 python3 tools/exp/step_prof.py eva 1024 200; done
timeout -k 10 120 python3 tools/exp/step_prof.py cgan 1024 200
