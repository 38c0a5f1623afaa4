"""CPU oracle for the helmholtz-x hot path (TEST INFRASTRUCTURE, not product code).

This package is a NumPy/SciPy restatement of the reference algorithm for operator assembly
and the fixed-point shift-invert eigen-iteration.  It exists only so that ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can check / time the HIP
path against an independent CPU computation.  Nothing under ``helmholtz-x_amd/`` imports it.

Parity status: PINNED.  The restatement is checked against the reference's own committed
results (``tests/golden/reference_logs.json``, ``tests/golden/eigenvectors``,
``tests/test_oracle_goldens.py``): Rijke 1-D/2-D/3-D passive / active / adjoint, PRF Rijke
1-D/2-D/3-D direct / adjoint, the analytic 2-D dispersion relation, FlamedDuct (choked
boundaries, variable gamma), RijkeFFD on the original and the optimised shape (16-digit
eigenvalues), the MICCA full annulus on the reference's own XDMF/HDF5 mesh (logged trajectory,
16-digit eigenvalues), the MICCA Bloch sector (passive + active), and the normalised
eigenvectors the reference committed as HDF5 (direct and adjoint, up to the free sign).

The arithmetic the reference delegates to DOLFINx 0.9.0 / PETSc / SLEPc (not vendored under
/root/reference and not installable here) is restated from the published algorithms:
P1 Lagrange element integrals (closed-form simplex monomial integrals, equal to the FFCx
quadrature of sufficient degree to rounding), shift-and-invert Krylov eigen-solves
(ARPACK on the explicit operator), and the first-companion linearisation of the quadratic
eigenproblem.
"""
