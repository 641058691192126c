#!/bin/bash
# TA / TCP / TLB counters of the test-path kernels (k_zscore first): gpurun -- 'bash tools/pmc_zscore.sh r05 125 50000'
# (few counters of one block per pass: a set the hardware cannot collect makes rocprofv3 abort -- and hang: every pass
# runs under `timeout`, tools/pmc_run.sh)
TAG=${1:-r05}; NS=${2:-125}; BS=${3:-50000}
i=0
for C in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
         "TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
         "TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
         "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
         "TCP_TCP_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum"; do
  i=$((i+1))
  bash tools/pmc_run.sh ${TAG}z${i} "$C" tools/gpu_test_scale.py $NS $BS 3 2>&1 | grep "k_zscore" | grep -v pairs | head -1
done
