import sys,json
d=json.loads(sys.stdin.read()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["sweep_ms"] if "roofline" in d else "")
