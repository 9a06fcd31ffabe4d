cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm" 2>&1 | tail -2
DGG_LIB=tools/exp/libdgg_x5stamp.so timeout -k 10 100 python3 tools/exp/x5_stamps.py 0 1024 768 4120 4 | head -3
DGG_LIB=tools/exp/libdgg_x5stamp.so timeout -k 10 100 python3 tools/exp/x5_stamps.py 2 768 4120 1024 0 | head -3
for v in 1 2; do timeout -k 10 120 python3 tools/exp/step_prof.py joint 1024 200; done
