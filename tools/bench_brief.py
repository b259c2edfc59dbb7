"""Run bench.py with the given arguments and print the headline fields only (the full JSON line is long)."""
import json, subprocess, sys
p = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in p.stdout.splitlines() if l.startswith("{")]
if not line:
    print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
j = json.loads(line[-1])
out = {k: j.get(k) for k in ("value", "ms_per_step", "ms_per_step_median")}
out["frac"] = j.get("roofline", {}).get("frac")
out["losses"] = j.get("config", {}).get("losses_last_step")
dk = j.get("roofline", {}).get("dominant_kernel")
if dk: out["dominant"] = {k: dk.get(k) for k in ("name", "launches_per_step", "avg_us", "frac")}
out["variants"] = [(v["workload"][:40], v.get("ms_per_step"), v.get("roofline_frac")) for v in j.get("variants", [])]
print(json.dumps(out))
