export ROUND=${ROUND:-5}
R=$GRAFT_REPO_ROOT
bash $R/tools/refresh_profiles.sh > /dev/null 2>&1
bash $R/tools/traffic_pmc.sh > /dev/null 2>&1
bash $R/tools/mfma_pmc.sh > /dev/null 2>&1
bash $R/tools/gather_pmc.sh > /dev/null 2>&1
bash $R/tools/trace_model.sh gru4rec > /dev/null 2>&1
bash $R/tools/trace_model.sh metamodel --steps 120 --warmup 30 > /dev/null 2>&1
bash $R/tools/trace_model.sh fmlp > /dev/null 2>&1
bash $R/tools/trace_model.sh cl4srec > /dev/null 2>&1
bash $R/tools/trace_one.sh sasrec_d128_B256 --steps 200 --warmup 20 --embed-dim 128 > /dev/null 2>&1
bash $R/tools/trace_one.sh sasrec_d128_B8192 --steps 60 --warmup 10 --embed-dim 128 --batch 8192 > /dev/null 2>&1
(cd $R && timeout 300 python bench.py --embed-dim 128 --batch 8192 --steps 60 --warmup 10 --no-cpu-baseline --no-strong --no-throughput-mode 2>/dev/null | tail -1 > $R/gpurun_out/r$ROUND/bench_sasrec_d128_B8192.json)
ls -la $R/gpurun_out/r$ROUND
