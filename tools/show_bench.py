import json, sys
d = json.load(open(sys.argv[1]))
print("FPS", d["value"], "ms/step", d["ms_per_step"], "| dominant:", d["roofline"]["kernel"], d["roofline"]["frac"])
for k, v in d["kernels"].items():
    print(f"  {k:14s} avg {v['avg_us']:8.1f} us  x{v['launches']:5d}  total {v['total_ms']:8.1f} ms  {v['achieved']:10.3f} {v['unit']}  frac {v['frac']:.4f}")
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"])
