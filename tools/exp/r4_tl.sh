cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "not bps" 2>&1 | tail -2
for v in 1 2 3; do python3 tools/exp/step_prof.py joint 1024 200; done
python3 tools/exp/step_prof.py cgan 1024 200; python3 tools/exp/step_prof.py eva 1024 200; python3 tools/exp/step_prof.py cgan 256 200; python3 tools/exp/step_prof.py joint 512 200
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_ep3 -o t -- python3 tools/exp/step_prof.py cgan 1024 12 > gpurun_out/r4_ep3.log 2>&1
python3 tools/exp/ktrace_groups.py gpurun_out/r4_ep3 | cut -c1-150 | sort -k10 -n -r | head -40
