#!/bin/bash
# copies what tools/refresh_all.sh left under gpurun_out/r<ROUND>/ (scratch) into profiles/round<ROUND>_* (tracked): run here, after the gpurun call
R=${ROUND:-5}; S=gpurun_out/r$R; P=profiles
for f in $S/bench_*.json $S/kernels_*.txt $S/timeline_*.txt $S/pmc_traffic_*.json $S/outer_*.txt; do [ -f "$f" ] && cp "$f" $P/round${R}_$(basename $f); done
for f in $S/mfma_util_*.json; do [ -f "$f" ] && cp "$f" $P/round${R}_pmc_$(basename $f); done
[ -f $S/rocprofv3_kernel_stats_default.csv ] && cp $S/rocprofv3_kernel_stats_default.csv $P/round${R}_rocprofv3_kernel_stats_default.csv
[ -f $S/gather_pmc_raw.txt ] && cp $S/gather_pmc_raw.txt $P/round${R}_gather_pmc_raw.txt
ls $P | grep round$R
