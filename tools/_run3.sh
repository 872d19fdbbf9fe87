for v in "DR4SR_WGRAD_GW_CAP=96" "DR4SR_WGRAD_GW_CAP=128" "DR4SR_WGRAD_GW_CAP=192" "DR4SR_WGRAD_GW_CAP=256" "DR4SR_WGRAD_GW_CAP=320"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong --no-throughput-mode --batch 8192 > gpurun_out/b3.json 2> gpurun_out/b3.err
  python - <<'PY'
import json
d=json.load(open("gpurun_out/b3.json"))
print("toys B8192 %.4f ms" % d["ms_per_step"], d["kernel_us_per_step"]["wgrad_fused"])
PY
  env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong --no-throughput-mode --batch 8192 --dense > gpurun_out/b3d.json 2> gpurun_out/b3d.err
  python - <<'PY'
import json
d=json.load(open("gpurun_out/b3d.json"))
print("dense B8192 %.4f ms" % d["ms_per_step"], d["kernel_us_per_step"]["wgrad_fused"])
PY
done
