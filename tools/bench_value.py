import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); print(round(d["value"]), round(d["ms_per_step"]*1e3,1), "us/step")
