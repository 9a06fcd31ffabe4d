#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the bench step, and separate --pmc passes (never combined with traces) for the dominant GEMM
#   launches, the BPS encode and Adam.  Summaries land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns
#   them into the committed files under profiles/.
set -e
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# ONLY="bps bps_valu": just those microbench passes, no bench trace
if [ -z "$ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --no-cpu-baseline --no-sub --no-other-mode --repeats 3 > $OUT/bench.log 2>&1
fi
# name:[X4=1:]microbench arguments   (x4_*: the loader / consumer kernel of the >= 1024-workgroup launches, forced by DGG_X4=1 set in the shell — the program
# behind `--` stays python3 itself)
for what in "gemm_nt:gemm --k 4120 --splits 4 --iters 10" "gemm_tn:gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 10" \
            "x4_nt:X4=1:gemm --k 4120 --splits 4 --iters 10" "x4_tn:X4=1:gemm --mode 2 --m 768 --n 4120 --k 1024 --splits 0 --iters 10" \
            "bps:bps --iters 5" "bps_valu:VALU=1:bps --iters 5"; do
  name=${what%%:*}; args=${what#*:}
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then continue; fi
  export DGG_X4=0
  unset DGG_BPS
  if [ "${args%%:*}" = "X4=1" ]; then export DGG_X4=1; args=${args#*:}; fi
  if [ "${args%%:*}" = "VALU=1" ]; then export DGG_BPS=valu; args=${args#*:}; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o t -- python3 $R/tools/microbench.py $args > $OUT/${name}_trace.log 2>&1
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
    tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $pmc --output-format csv -d $OUT/${name}_pmc_$tag -o p -- python3 $R/tools/microbench.py $args > $OUT/${name}_pmc_$tag.log 2>&1
  done
done
echo done > $OUT/done.txt
find $OUT -name "*.csv" | wc -l
