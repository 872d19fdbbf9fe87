#!/bin/bash
# Where do the waves of the at-scale step kernels spend their cycles?  Several --pmc passes (4 counters each) over the B = 8192 toys bench
# -> gpurun_out/r<ROUND>/sq_pmc_B8192_toys.txt  (per kernel: mean per dispatch of each counter, summed over SEs / XCCs)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r${ROUND:-3}; mkdir -p $O
: > $O/sq_pmc_B8192_toys.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/sq_$i -o t -- \
    python $R/bench.py --steps 6 --warmup 2 --no-graph --no-cpu-baseline --no-throughput-mode --no-strong --no-deterministic-leg --batch 8192 $SQ_EXTRA > /tmp/sq_$i.log 2>&1
  db=$(find /tmp/sq_$i -name "*.db" | head -1)
  [ -n "$db" ] && python - "$db" >> $O/sq_pmc_B8192_toys.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select k.name, p.dispatch_id, p.counter_name, sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by p.dispatch_id, p.counter_name").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(list))
for name, did, cn, s in rows:
    per[name][cn].append(s)
for name, cs in sorted(per.items()):
    if not any(f in name for f in ("k_post", "k_wgrad", "k_attn", "k_embqkv", "k_qkv_embed", "k_wt_")):
        continue
    print(name[:72].ljust(72), " ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(cs.items())))
PY
done
cat $O/sq_pmc_B8192_toys.txt
