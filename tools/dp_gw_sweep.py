"""sweeps of launch parameters for the one-launch step (A: dr4sr_sasrec_train_steps) and the two-phase step (C / D: phase 1 | phase 2 with
split 1 / 2, no collective); see tools/dp_cost_probe.py.  SWEEP_ENV=NAME, SWEEP_VALUES=comma list ('-' = unset)"""
import os, sys
os.environ["PROBE_B"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "tools", "dp_cost_probe.py")).read().split("\nfor B in [int(x)")[0]
exec(compile(src, "dp_cost_probe_head", "exec"))
name = os.environ.get("SWEEP_ENV", "DR4SR_WGRAD_GW_CAP")
forms = os.environ.get("SWEEP_FORMS", "A,C,D").split(",")
for B in [int(x) for x in os.environ.get("SWEEP_B", "4096,16384").split(",")]:
    for v in os.environ.get("SWEEP_VALUES", "-,96,128,160,192,224,256,320").split(","):
        setenv(**{name: None if v == "-" else v})
        res = []
        for f in forms:
            res.append("%s %.4f" % (f, run(B, f, {"C": "1", "D": "2"}.get(f))))
        print("DP_SWEEP B=%d %s=%s  %s ms" % (B, name, v, "  ".join(res)), flush=True)
parallel.shutdown()
