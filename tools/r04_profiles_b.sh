#!/bin/bash
# round-4 evidence, part B: model kernel splits, PMC traffic at the bench's batch sizes, matrix-pipe busy fractions
export R=r04
bash tools/prof_models.sh > gpurun_out/r04_prof_models.log 2>&1; tail -20 gpurun_out/r04_prof_models.log
rm -rf gpurun_out/prof_convnextv2 gpurun_out/prof_dinov2 gpurun_out/prof_medsiglip
bash tools/pmc_busy.sh > gpurun_out/r04_pmc_busy.log 2>&1; tail -12 gpurun_out/r04_mfma_busy.txt
rm -rf gpurun_out/busy_gemm gpurun_out/busy_embed
bash tools/pmc_traffic.sh > gpurun_out/r04_pmc_traffic.log 2>&1; tail -5 gpurun_out/r04_pmc_traffic.json
rm -rf gpurun_out/pmct_*
