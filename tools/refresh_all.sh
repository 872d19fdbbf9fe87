export ROUND=${ROUND:-4}
R=$GRAFT_REPO_ROOT
bash $R/tools/refresh_profiles.sh > /dev/null 2>&1
bash $R/tools/traffic_pmc.sh > /dev/null 2>&1
bash $R/tools/mfma_pmc.sh > /dev/null 2>&1
bash $R/tools/gather_pmc.sh > /dev/null 2>&1
bash $R/tools/trace_model.sh gru4rec > /dev/null 2>&1
bash $R/tools/trace_model.sh metamodel --steps 120 --warmup 30 > /dev/null 2>&1
bash $R/tools/trace_model.sh fmlp > /dev/null 2>&1
bash $R/tools/trace_model.sh cl4srec > /dev/null 2>&1
ls -la $R/gpurun_out/r2
