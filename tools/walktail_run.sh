#!/bin/bash
# (GPU box) every variant of tools/walktail_variants.py: N calls of the 125 x 50 kb batch against the first call's
# bits (truth = the host-driven levels, WC_TEST_WALK=0, which never ran the tail); prints the number of calls with a
# difference per variant.   usage: tools/walktail_run.sh [calls] [variant ...]
N=${1:-150}; shift
V=${@:-control typed global waitzero noinline}
mkdir -p gpurun_out
for v in $V; do
  TRUTH_ENV=WC_TEST_WALK=0 WC_LIB_PATH=wisecondor_amd/ab/lib_wt_$v.so timeout 600 python tools/gpu_repeatability.py 125 50000 $N > gpurun_out/wt_$v.log 2>&1
  bad=$(grep -c "calls [1-9][0-9]* differ" gpurun_out/wt_$v.log)
  tot=$(grep -c "^call [0-9]* vs call 0" gpurun_out/wt_$v.log)
  echo "variant $v: $bad of $tot calls differ from the host-driven levels"
done | tee gpurun_out/walktail_summary.txt
