import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["kernels"]
print(tag, "FPS", d["value"], "| " + " ".join(f"{n}={v['avg_us']:.1f}" for n, v in k.items()))
