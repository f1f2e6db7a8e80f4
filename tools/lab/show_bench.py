"""prints the headline numbers of a bench.py JSON line: python tools/lab/show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["ms_per_step"], "ms/step  loss", d.get("loss"))
a = d.get("attention", {})
for k in ("fwd", "bwd"):
    if k in a: print(" attention", k, a[k]["avg_launch_ms"], "ms/launch", a[k]["ms_per_step"], "ms/step", a[k]["frac_of_peak"])
r = d.get("roofline", {})
print(" gemm frac", r.get("frac"), "all_gemm ms/step", r.get("all_gemm_ms_per_step"), "swiglu-bwd launch", (r.get("swiglu_backward_launch") or {}).get("avg_launch_ms"))
for k, v in (r.get("hbm_bound_families") or {}).items():
    if isinstance(v, dict): print("  ", k, v["ms_per_step"], "ms/step", v["GB_per_s"], "GB/s", "(side stream)" if v.get("side_stream") else "")
    else: print("  ", k, v)
