#!/usr/bin/env python3
"""Table of tools/sweep_envelope.py records (JSON lines) -> text."""
import json
import sys

for path in sys.argv[1:]:
    recs = [json.loads(l) for l in open(path) if l.startswith("{")]
    if not recs:
        continue
    r0 = recs[0]
    print("== %s  (%s DOF, numbering %s, jitter %s)" % (r0["workload"], r0["dofs"], r0["numbering"], r0["jitter"]))
    print("%-16s %-8s %-4s %7s %5s %6s %7s %6s %-9s %s" % ("target", "leg", "ok", "s", "fpi", "ops", "solves", "its/s", "rungs", "omega[0] / error"))
    for r in recs:
        t = r["target"]
        tgt = "%g%+gj" % (t[0], t[1]) if t[1] else "%g" % t[0]
        inner = r.get("inner") or {}
        om = r.get("omega")
        tail = ("%.3f%+.3fj" % tuple(om[0])) if om else r.get("error", "")[:110]
        print("%-16s %-8s %-4s %7.2f %5s %6s %7s %6s %-9s %s" % (
            tgt, r["leg"], "ok" if r["ok"] else "FAIL", r.get("seconds", 0.0), r.get("fpi_iterations", ""), r.get("krylov_ops", ""),
            inner.get("solves", ""), inner.get("iters_per_solve", ""), inner.get("ladder", ""), tail))
    print()
