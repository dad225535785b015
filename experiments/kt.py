import sys,json
d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], {k.replace("fql_",""): v["avg_us"] for k,v in d["kernels"].items() if "conv" in k})
