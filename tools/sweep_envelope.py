#!/usr/bin/env python3
"""Envelope of the iterative shift-invert solve: for one workload, the passive quadratic eigen-solve and the
whole active fixed-point call at the targets the reference ran (AnnularCombustor/Micca/fullAnnulus/runAll.sh:14-19,
active_modes.py:41-48: 1000 ... 11000 rad/s) and beyond, until the inner solver gives up.  One JSON line per (target, leg):
iterations per inner solve, solves, time, rungs of the escalation ladder used, failure text.

  python tools/sweep_envelope.py annulus-1M [--numbering rcm --jitter 0.2] [--targets 1000,2000,...] [--out file.jsonl]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import torch  # noqa: E402

import bench  # noqa: E402

REFERENCE_TARGETS = [1000.0, 2000.0, 3225.12 + 481j, 5000.0, 9000.0, 10000.0, 11000.0]
BEYOND = [13000.0, 15000.0, 17500.0, 20000.0, 22000.0, 25000.0, 30000.0]


def inner_of(caches):
    return next((v for v in caches[0].values() if hasattr(v, "stats")), None)


def pick(stats):
    keep = ("levels", "solves", "iters_per_solve", "gmres_iters", "setup_s", "solve_s", "recycled_pairs", "krylov", "cocg_fallbacks",
            "attainable_stops", "harmonic_ritz", "ladder", "rung", "direct", "factor_device_s", "analyse_s", "nnz_L", "fill", "factor_gib",
            "first_residual", "refine", "replaced_pivots", "ordering")
    return {k: stats[k] for k in keep if k in stats}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--numbering", default="native")
    ap.add_argument("--jitter", type=float, default=0.0)
    ap.add_argument("--targets", default=None)
    ap.add_argument("--legs", default="passive,active")
    ap.add_argument("--out", default=None)
    ap.add_argument("--st-type", default=None, help="inner solver: lu | amg | auto (default: the library's choice)")
    ap.add_argument("--stop-after-failures", type=int, default=2, help="stop the sweep after this many consecutive failing targets")
    args = ap.parse_args()
    from helmholtz_x.acoustic_matrices import AcousticMatrices
    from helmholtz_x.eigensolvers import fixed_point_iteration, pep_solver
    from helmholtz_x.flame_matrices import PointwiseFlameMatrix
    from helmholtz_x.parameters_utils import Q_multiple
    targets = ([complex(t) for t in args.targets.split(",")] if args.targets else REFERENCE_TARGETS + BEYOND)
    case = bench.build_case(bench.WORKLOADS[args.workload], args.numbering, args.jitter)
    P, mesh = case["P"], case["mesh"]
    M = AcousticMatrices(mesh, case["facet_tags"], {11: {'Robin': P.R_outlet}}, case["c"])
    h = Q_multiple(mesh, case["subdomains"], P.N_sector).resident()
    D = PointwiseFlameMatrix(mesh, case["subdomains"], P.x_r(), h, P.rho_amb, P.q_0, P.u_b, case["ftf"])
    D.assemble_submatrices('direct')
    out = open(args.out, "a") if args.out else None

    def emit(rec):
        rec.update(workload=args.workload, numbering=case["numbering"], jitter=args.jitter, dofs=mesh.num_vertices)
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    failures = 0
    for target in targets:
        ok = True
        tt = [target.real, target.imag]
        if "passive" in args.legs:
            caches = ({}, {}, {})
            t0 = time.perf_counter()
            try:
                E = pep_solver(M.A, M.B, M.C, target, nev=2, st_type=args.st_type, _caches=caches)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                emit(dict(target=tt, leg="passive", ok=True, seconds=round(dt, 3),
                          omega=[[round(E.getEigenvalue(i).real, 4), round(E.getEigenvalue(i).imag, 4)] for i in range(2)],
                          krylov_ops=caches[2].get("krylov_ops"), inner=pick(inner_of(caches).stats())))
                E._owns_caches = True
                E.destroy()
            except Exception as exc:                           # report, keep sweeping
                ok = False
                inner = inner_of(caches)
                emit(dict(target=tt, leg="passive", ok=False, seconds=round(time.perf_counter() - t0, 3),
                          error="%s: %s" % (type(exc).__name__, str(exc)[:300]),
                          inner=pick(inner.stats()) if inner is not None else None))
                for v in caches[0].values():
                    getattr(v, "destroy", lambda: None)()
        if "active" in args.legs:
            hist = []
            t0 = time.perf_counter()
            E = None
            try:
                E = fixed_point_iteration(M, D, target, i=0, nev=4, tol=1e-3, st_type=args.st_type, history=hist)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                inner = next((v for v in (E.inner_cache or {}).values() if hasattr(v, "stats")), None)
                emit(dict(target=tt, leg="active", ok=True, seconds=round(dt, 3), fpi_iterations=len(hist) - 1,
                          omega=[[round(E.getEigenvalue(i).real, 4), round(E.getEigenvalue(i).imag, 4)]
                                 for i in range(min(E.getConverged(), 4))], converged=E.getConverged(),
                          krylov_ops=(E.state or {}).get("krylov_ops"), ops_per_solve=(E.state or {}).get("ops_per_solve"),
                          inner=pick(inner.stats()) if inner is not None else None))
                E.destroy()
            except Exception as exc:
                ok = False
                emit(dict(target=tt, leg="active", ok=False, seconds=round(time.perf_counter() - t0, 3),
                          fpi_iterations=len(hist) - 1, error="%s: %s" % (type(exc).__name__, str(exc)[:300])))
            import gc
            E = None
            gc.collect()
        failures = 0 if ok else failures + 1
        if failures >= args.stop_after_failures:
            emit(dict(target=tt, leg="sweep", ok=False, error="stopped after %d consecutive failing targets" % failures))
            break


if __name__ == "__main__":
    main()
