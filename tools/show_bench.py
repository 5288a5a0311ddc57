import json,sys
d=json.load(open(sys.argv[1]))
r=d.pop("roofline"); print(json.dumps(d)[:1800]); pk=r.pop("per_kernel"); r.pop("note"); r.pop("traffic_source",None); print(json.dumps(r)); print(pk)
