"""A/B timing of library variants in ONE process group on ONE device (interleaved rounds; cdna guide rule 24).
usage: ab_bench.py libA.so libB.so ...   (each variant runs in a subprocess per round; same GPU)"""
import os, subprocess, sys, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, PNR_LIB=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--cpu-rays", "0"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(json.loads(out)["roofline"]["kernel_ms"])
for l in libs:
    print(f"{l:40s} kernel_ms median {statistics.median(res[l]):.3f}  min {min(res[l]):.3f}  all {[round(v,3) for v in res[l]]}")
