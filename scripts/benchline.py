import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    c = d.get("cpu_baseline") or {}
    print("cols/s %.1f  ms/step %.0f  kernel %s  alg GB/s %.0f (frac %.3f)  B/col %.3e  cpu %.1f cols/s (%s cores) maxdW %s"
          % (d["value"], d["ms_per_step"], d["config"]["kernel"], r["achieved"], r["frac"], r["alg_bytes_per_column"],
             c.get("value", 0), c.get("cores"), c.get("max_abs_dW_vs_gpu")))
