#!/bin/bash
# Busy / cache / wait counters of the `test` path kernels (run on the GPU box through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/pmc_test_path.sh r04'
# three counter sets x two batch shapes (128 x 250 kb, 125 x 50 kb), one rocprofv3 --pmc pass each
# (tools/pmc_run.sh), summarised into profiles/<tag>_pmc_busy_test.{md,json} by tools/busy_summary.py.
set -u
TAG=${1:-r04}
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
T="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
W="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
bash tools/pmc_run.sh ${TAG}A_t250 "$A" tools/gpu_test_scale.py 128 250000 6
bash tools/pmc_run.sh ${TAG}T_t250 "$T" tools/gpu_test_scale.py 128 250000 6
bash tools/pmc_run.sh ${TAG}W_t250 "$W" tools/gpu_test_scale.py 128 250000 6
bash tools/pmc_run.sh ${TAG}A_t50 "$A" tools/gpu_test_scale.py 125 50000 4
bash tools/pmc_run.sh ${TAG}T_t50 "$T" tools/gpu_test_scale.py 125 50000 4
bash tools/pmc_run.sh ${TAG}W_t50 "$W" tools/gpu_test_scale.py 125 50000 4
python3 tools/busy_summary.py ${TAG}_test gpurun_out/${TAG}A_t250 gpurun_out/${TAG}T_t250 gpurun_out/${TAG}W_t250 \
    gpurun_out/${TAG}A_t50 gpurun_out/${TAG}T_t50 gpurun_out/${TAG}W_t50 > gpurun_out/${TAG}_busy_test.log 2>&1
mkdir -p gpurun_out/profiles_${TAG}
cp profiles/${TAG}_test_pmc_busy.md profiles/${TAG}_test_pmc_busy.json gpurun_out/profiles_${TAG}/
tail -5 gpurun_out/${TAG}_busy_test.log
