import json,sys
d=json.load(open(sys.argv[1]))
print(round(d["value"]), round(d["ms_per_step"],4), {k:round(v,4) for k,v in d["kernel_ms_solo"].items()})
