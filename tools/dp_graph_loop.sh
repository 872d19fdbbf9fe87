#!/bin/bash
# VERDICT r5 Next #1(a): loop the RCCL-in-graph data-parallel check (tools/dp_graph_check.py: rendezvous, communicator, capture of a K-step DP
# graph with its all-reduce, REPLAYS replays, parity, teardown) N times in FRESH processes with full stderr kept; prints the pass count.
#   tools/dp_graph_loop.sh [N=200] [outfile=gpurun_out/dp_graph_loop.txt]
N=${1:-200}
OUT=${2:-gpurun_out/dp_graph_loop.txt}
cd "$(dirname "$0")/.."
mkdir -p "$(dirname "$OUT")"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TORCH_SHOW_CPP_STACKTRACES=1 MASTER_ADDR=127.0.0.1
export DP_GRAPH_B=${DP_GRAPH_B:-256} DP_GRAPH_K=${DP_GRAPH_K:-4} DP_GRAPH_REPLAYS=${DP_GRAPH_REPLAYS:-30}
: > "$OUT"
ok=0; bad=0; t0=$(date +%s)
for i in $(seq 1 "$N"); do
    port=$(python3 -c 'import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); print(s.getsockname()[1])')
    if timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port "$port" \
            tools/dp_graph_check.py > /tmp/dp_loop_one.out 2> /tmp/dp_loop_one.err && grep -q DP_GRAPH_OK /tmp/dp_loop_one.out; then
        ok=$((ok + 1))
    else
        bad=$((bad + 1))
        { echo "==== run $i FAILED (rc $?)"; tail -50 /tmp/dp_loop_one.out; echo "---- stderr"; tail -200 /tmp/dp_loop_one.err; } >> "$OUT"
    fi
done
t1=$(date +%s)
{ echo "dp_graph_loop: $N fresh-process runs of tools/dp_graph_check.py (B=$DP_GRAPH_B, K=$DP_GRAPH_K steps per graph, $DP_GRAPH_REPLAYS replays each,"
  echo "  data plane = libdr4sr_hip.so's own RCCL communicator, control plane = gloo): $ok passed, $bad failed, $((t1 - t0)) s"
  grep "^DP_GRAPH " /tmp/dp_loop_one.out | tail -1; } | tee -a "$OUT"
[ "$bad" -eq 0 ]
