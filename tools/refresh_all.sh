export ROUND=2
R=$GRAFT_REPO_ROOT
bash $R/tools/refresh_profiles.sh > /dev/null 2>&1
bash $R/tools/traffic_pmc.sh > /dev/null 2>&1
bash $R/tools/mfma_pmc.sh > /dev/null 2>&1
bash $R/tools/gather_pmc.sh > /dev/null 2>&1
ls -la $R/gpurun_out/r2
