"""A/B timing of library variants on ONE device (interleaved rounds; cdna guide rule 24).
usage: ab_bench.py [--workload W] libA.so libB.so ..."""
import os, subprocess, sys, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
wl = []
if args and args[0] == "--workload":
    wl = ["--workload", args[1]]; args = args[2:]
libs = args
res = {l: [] for l in libs}
step = {}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, PNR_LIB=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--cpu-rays", "0", "--secondary-steps", "0", "--precision", os.environ.get("PNR_AB_PRECISION", "bf16")] + wl,
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append(d["roofline"]["kernel_ms"])
        step.setdefault(l, []).append(d["ms_per_step"])
for l in libs:
    print(f"{l:40s} kernel_ms median {statistics.median(res[l]):.3f}  min {min(res[l]):.3f}  all {[round(v,3) for v in res[l]]}"
          f"   ms_per_step median {statistics.median(step[l]):.3f}")
