import json,sys
d=json.load(open(sys.argv[1]))
out=[("value",round(d["value"]))]
if "case_batch" in d: out.append(("cb",round(d["case_batch"]["ms_per_step"]*1e3,1)))
for k,l in d.get("legs",{}).items(): out.append((k, round(l["ms_per_step"]*1e3,1)))
print(sys.argv[1], out)
