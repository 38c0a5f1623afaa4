import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); s=d["config"]["inner_stats"]
    print(f.split('/')[-1], d["ms_per_step"], s["gmres_iters"], [tuple(o) for o in d["config"]["omega"]])
