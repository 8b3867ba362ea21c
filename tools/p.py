import sys,json
d=json.loads(sys.stdin.read()); r=d.get("roofline", {})
print(sys.argv[1], d.get("ms_per_step"), r.get("sweep_ms"), r.get("launch_ms"))
