#!/usr/bin/env python3
"""cProfile of one WARM hot-path pass (host-side view: where the wall time outside the kernels goes)."""
import cProfile
import contextlib
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "helmholtz-x_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HX_QUIET", "1")
import bench  # noqa: E402

case = bench.build_case(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "annulus-5M"])
case["mesh"].assembler()
pr = cProfile.Profile()
with contextlib.redirect_stdout(sys.stderr):
    E, _ = bench.hot_path(case)            # warm-up pass: code objects, allocator pools
    E.destroy()
    del E
    import gc
    gc.collect()
    pr.enable()
    bench.hot_path(case)
    pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
for pat in sys.argv[3:]:                   # who calls these (regular expressions on the function name)
    st.print_callers(pat)
