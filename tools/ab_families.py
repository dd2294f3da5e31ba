"""Developer tool: same-box A/B of build / run-time variants, per kernel family (in-situ launch timer of bench.py).
usage: python tools/ab_families.py name=ENV1=v,ENV2=v ...   (name 'base' = no overrides)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for spec in sys.argv[1:]:
    name, _, envs = spec.partition("=")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v.replace("$ROOT", ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(name, "FAILED", out.stderr[-2000:])
        continue
    rows[name] = d
    print(f"{name:10s} ms/step {d['ms_per_step']:8.2f}  unet {d['unet_forward']['ms']:7.3f}  single {d['single_frame']['ms_per_frame']:7.2f}", flush=True)
fams = []
for d in rows.values():
    for f in d["kernel_families"]:
        if f["kernel"] not in fams:
            fams.append(f["kernel"])
print("%-52s" % "family (ms per step, in situ)" + "".join(f"{n:>10s}" for n in rows))
for fam in fams[:16]:
    line = "%-52s" % fam[:50]
    for d in rows.values():
        v = [f["ms_per_step"] for f in d["kernel_families"] if f["kernel"] == fam]
        line += f"{v[0]:10.2f}" if v else f"{'-':>10s}"
    print(line)
