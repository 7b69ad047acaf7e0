import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    det=d.get("detail",{})
    keep={k:det[k] for k in det if k in ("grouped_grad_steps_per_s","us_per_lockstep","loop_grad_steps_per_s")}
    for k in det:
        if k.startswith("grouped_K"): keep[k]=det[k].get("aggregate_grad_steps_per_s") if isinstance(det[k],dict) else det[k]
    print(d.get("metric","")[:50], d.get("value"), json.dumps(keep))
