import json,sys
d=json.load(open(sys.argv[1]))
tag=sys.argv[2]
def show(d,pre=""):
    for k,v in d.items():
        if isinstance(v,dict): show(v,pre+k+".")
        elif isinstance(v,(int,float)) and pre.startswith("leg.") and ("ms" in k): print(tag, pre+k, round(v,4))
show(d)
