"""Condensed view of bench_aux.py's JSON lines (stdin): value, the leg's headline details, the roofline kernel."""
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
    except Exception:
        continue
    det = d.get("detail", {})
    keep = {k: det[k] for k in det if k in ("grouped_grad_steps_per_s", "us_per_lockstep", "loop_grad_steps_per_s", "us_per_iteration")}
    for k in det:
        if k.startswith("grouped_K"):
            keep[k] = det[k].get("aggregate_grad_steps_per_s") if isinstance(det[k], dict) else det[k]
        if k.startswith("mb"):
            keep[k] = {kk: det[k][kk] for kk in ("train_step_s", "sample_updates_per_s") if kk in det[k]}
    r = d.get("roofline") or {}
    print(d.get("metric", "")[:50], d.get("value"), json.dumps(keep), r.get("kernel"), r.get("frac"), json.dumps(r.get("kernel_ms"))[:300])
