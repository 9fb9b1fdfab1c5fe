import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"])
for g in d["in_step"]: print(round(g["ms"],3), g["launches"])
print(d["ntt"])
for k,v in d.get("secondary",{}).items():
    if k!="hoisted_rotations": print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items() if a!="workload"})
print({k:(round(v["forward_GBps"]),round(v["inverse_GBps"])) for k,v in d.get("ntt_by_degree",{}).items()})
