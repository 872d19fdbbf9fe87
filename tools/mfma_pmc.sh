#!/bin/bash
# MFMA utilisation of the training-step kernels from hardware counters -> gpurun_out/r<ROUND>/mfma_util_<tag>.json (tools/mfma_util.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-3}; mkdir -p $O
for cfg in "B8192_dense:--batch 8192 --dense" "B8192_toys:--batch 8192" "B256_toys:"; do
  tag=${cfg%%:*}; fl=${cfg#*:}; rm -rf /tmp/p_$tag
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_$tag -o t -- \
    python $R/bench.py --steps 6 --warmup 2 --no-graph --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg $fl > /tmp/p_$tag.log 2>&1
  python $R/tools/mfma_util.py $(find /tmp/p_$tag -name "*.db" | head -1) k_attn2 k_post k_qkv k_wgrad k_embqkv k_wt_ k_attn_ > $O/mfma_util_$tag.json
done
