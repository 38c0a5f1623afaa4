"""Eigen-solvers with the reference's signatures, running on the GPU.

Drop-in for ``helmholtz_x.eigensolvers`` (reference: eigensolvers.py:8-276): ``results``, ``eps_solver``,
``pep_solver``, ``fixed_point_iteration_eps``, ``fixed_point_iteration_pep``, ``fixed_point_iteration``,
``newtonSolver``.
Same update rule (relaxed secant fixed point, alpha_0 = 0.5), same printed log lines, same returned
solver object ``E``.

Because the shift-invert target never changes during a fixed-point iteration (:53-54, :105), the sparse
part K of the shifted operator is omega-independent: it is factorised ONCE per call and reused, while
the omega-dependent rank-r flame term enters through a Woodbury update (slepc.py).  The reference instead
re-assembles and re-factorises A - D(omega_k), with D explicit, every iteration (:165-178, :232-242).

Deliberate fixes of reference quirks (last paragraph): ``maxiter`` is enforced
(the reference only sizes arrays with it and raises IndexError beyond); the adjoint/direct ValueError is
actually raised.
"""
import numpy as np

from .slepc import DECIDE, EPS, PEP
from .solver_utils import info


def results(E):
    print()
    print("******************************")
    print("*** SLEPc Solution Results ***")
    print("******************************")
    print()
    print("Number of iterations of the method: %d" % E.getIterationNumber())
    print("Solution method: %s" % E.getType())
    nev, ncv, mpd = E.getDimensions()
    print("Number of requested eigenvalues: %d" % nev)
    tol, maxit = E.getTolerances()
    print("Stopping condition: tol=%.4g, maxit=%d" % (tol, maxit))
    nconv = E.getConverged()
    print("Number of converged eigenpairs %d" % nconv)
    if nconv > 0:
        print()
    for i in range(nconv):
        k = E.getEigenvalue(i)
        print("%15f, %15f" % (k.real, k.imag))
    print()


def eps_solver(A, C, target, nev, two_sided=False, print_results=False, st_type=None, _caches=None, shard=None):
    E = EPS().create()
    C = -C
    E.setOperators(A, C)
    st = E.getST()
    st.setType('sinvert')
    E.setTarget(target ** 2)
    E.setWhichEigenpairs(EPS.Which.TARGET_MAGNITUDE)
    E.setTwoSided(two_sided)
    E.setDimensions(nev, DECIDE)
    E.setTolerances(1e-15)
    E.st_type = st_type
    E.shard = shard
    E.setFromOptions()          # -eps_* / -st_* overrides from the options database (slepc.py)
    if _caches is not None:
        E.inner_cache, E.W_cache, E.state = _caches
    info("- EPS solver started.")
    E.solve()
    info("- EPS solver converged. Eigenvalue computed.")
    if print_results:
        results(E)
    return E


def pep_solver(A, B, C, target, nev, print_results=False, st_type=None, _caches=None, shard=None):
    Q = PEP().create()
    Q.setOperators([A, B, C])
    st = Q.getST()
    st.setType('sinvert')
    Q.setTarget(target)
    Q.setWhichEigenpairs(PEP.Which.TARGET_MAGNITUDE)
    Q.setDimensions(nev, DECIDE)
    Q.setTolerances(1e-15)
    Q.st_type = st_type
    Q.shard = shard
    Q.setFromOptions()          # -pep_* / -st_* overrides from the options database (slepc.py)
    if _caches is not None:
        Q.inner_cache, Q.W_cache, Q.state = _caches
    info("- PEP solver started.")
    Q.solve()
    info("- PEP solver converged. Eigenvalue computed.")
    if print_results:
        results(Q)
    return Q


def _fmt(tol):
    digits = int("{:.0e}".format(tol)[-2:])
    return "{{:+.{}f}}".format(digits)


def _inner_rtol(tol, target):
    """Relative tolerance the iterative inner solves need so that the eigenvalue noise stays an order of magnitude below the
    stopping tolerance ``tol`` of the fixed point (an ABSOLUTE tolerance on omega, eigensolvers.py:155,223).  The reference's
    direct solve (PCLU) leaves ~1e-15; the default 1e-11 is enough down to tol/|omega| = 1e-10 and is tightened below that, to at
    most 1e-13 -- FlamedDuct's default call (tol = 1e-8 at |omega| = 1650, i.e. 6e-12 relative) then stops at the same
    iteration as the reference's log instead of one later on 1e-8 worth of solver noise."""
    rel = tol / max(abs(target), 1e-300)
    return min(1e-11, max(1e-13, 0.1 * rel))


def _flame_matrix(D, omega, problem_type):
    D.assemble_matrix(omega, problem_type)
    if problem_type == 'direct':
        return D.matrix
    if problem_type == 'adjoint':
        return D.adjoint_matrix
    raise ValueError("The problem type should be specified as 'direct' or 'adjoint'.")


def _iterate(solve_passive, solve_active, nev, i, tol, maxiter, s, history, state=None):
    """Relaxed secant fixed-point loop shared by the EPS and PEP variants (eigensolvers.py:143-193, 213-257).

    Only eigenpair ``i`` drives the iteration, so the intermediate solves ask the Krylov-Schur solver for i+1
    pairs; once |domega| <= tol the solve of that last iteration (same omega_k, same operator) is repeated for
    the full ``nev`` pairs and returned -- the object the reference returns from its last iteration."""
    omega = np.zeros(maxiter + 1, dtype=complex)
    f = np.zeros(maxiter, dtype=complex)
    alpha = np.zeros(maxiter, dtype=complex)
    lean = min(nev, i + 1)
    if state is not None:
        # lean solves also hand over their best unconverged Ritz directions (HX_WARM_EXTRA more than the missing pairs: the partner of
        # a doublet the nev boundary cuts)
        state["warm_pairs"] = nev - lean + int(__import__("os").environ.get("HX_WARM_EXTRA", "0"))
        state["lean_nev"] = lean
        state["nev_max"] = nev
    if state is not None:
        state["_omega_k"] = None                 # (the passive problem is not a member of the omega-family of the loop)
    E, omega[0] = solve_passive(lean)
    alpha[0] = 0.5
    if history is not None:
        history.append(omega[0])
    domega = 2 * tol
    k = -1
    while abs(domega) > tol:
        k += 1
        if k >= maxiter:
            raise RuntimeError("fixed-point iteration did not converge in maxiter=%d iterations" % maxiter)
        E.destroy()
        print("* iter = {:2d}".format(k + 1))
        if state is not None:
            state["_omega_k"] = complex(omega[k])    # lets the solver extrapolate its start vector along the trajectory
        E, f[k] = solve_active(omega[k], lean)
        if k != 0:
            alpha[k] = 1 / (1 - ((f[k] - f[k - 1]) / (omega[k] - omega[k - 1])))
        omega[k + 1] = alpha[k] * f[k] + (1 - alpha[k]) * omega[k]
        domega = omega[k + 1] - omega[k]
        if history is not None:
            history.append(omega[k + 1])
        print('+ omega = {}  {}j,  |domega| = {:.2e}\n'.format(
            s.format(omega[k + 1].real), s.format(omega[k + 1].imag), abs(domega)))
    if lean < nev:
        E.destroy()
        E, _ = solve_active(omega[k], nev) if k >= 0 else solve_passive(nev)
    E._owns_caches = True          # the returned solver owns the call's caches: its destroy() releases them (see slepc.py)
    return E


def fixed_point_iteration_eps(operators, D, target, nev=2, i=0, tol=1e-8, maxiter=50, print_results=False,
                              problem_type='direct', two_sided=False, st_type=None, history=None, shard=None):
    A, C = operators.A, operators.C
    B = operators.B_adj if problem_type == 'adjoint' else operators.B
    caches = ({}, {}, {"inner_rtol": _inner_rtol(tol, target)})       # inner solvers, Woodbury W, warm-start vector + counters
    s = _fmt(tol)
    info("--> Fixed point iteration started.\n")

    def passive(nev_req):
        E = eps_solver(A, C, target, nev_req, print_results=print_results, st_type=st_type, _caches=caches,
                       shard=shard)
        w0 = np.sqrt(E.getEigenvalue(i))
        print("+ Starting eigenvalue is found: {}  {}j. ".format(s.format(w0.real), s.format(w0.imag)))
        info("-> Iterations are starting.\n ")
        return E, w0

    def active(omega_k, nev_req):
        D_Mat = _flame_matrix(D, omega_k, problem_type)
        D_Mat = (A - D_Mat) if not B else (A + (omega_k * B) - D_Mat)
        E = eps_solver(D_Mat, C, target, nev_req, two_sided=two_sided, print_results=print_results, st_type=st_type,
                       _caches=caches, shard=shard)
        return E, np.sqrt(E.getEigenvalue(i))

    return _iterate(passive, active, nev, i, tol, maxiter, s, history, caches[2])


def fixed_point_iteration_pep(operators, D, target, nev=2, i=0, tol=1e-8, maxiter=50, print_results=False,
                              problem_type='direct', st_type=None, history=None, shard=None):
    A, C = operators.A, operators.C
    B = operators.B_adj if problem_type == 'adjoint' else operators.B
    caches = ({}, {}, {"inner_rtol": _inner_rtol(tol, target)})       # inner solvers, Woodbury W, warm-start vector + counters
    s = _fmt(tol)

    def passive(nev_req):
        E = pep_solver(A, B, C, target, nev_req, print_results=print_results, st_type=st_type, _caches=caches,
                       shard=shard)
        info("-> Fixed point iteration started.\n")
        return E, E.getEigenvalue(i)

    def active(omega_k, nev_req):
        D_Mat = A - _flame_matrix(D, omega_k, problem_type)
        E = pep_solver(D_Mat, B, C, target, nev_req, print_results=print_results, st_type=st_type, _caches=caches,
                       shard=shard)
        return E, E.getEigenvalue(i)

    return _iterate(passive, active, nev, i, tol, maxiter, s, history, caches[2])


def fixed_point_iteration(operators, D, target, nev=2, i=0, tol=1e-8, maxiter=50, print_results=False,
                          problem_type='direct', st_type=None, history=None, shard=None):
    """``shard`` (sharded.ShardContext, beyond the reference's signature): run the solve row-sharded over the ranks of
    a torch.distributed group -- what the reference gets from ``mpirun -np N`` through PETSc's distributed objects."""
    solver = fixed_point_iteration_pep if operators.B else fixed_point_iteration_eps
    return solver(operators, D, target, nev=nev, i=i, tol=tol, maxiter=maxiter, print_results=print_results,
                  problem_type=problem_type, st_type=st_type, history=history, shard=shard)


def newtonSolver(operators, D, init, nev=2, i=0, tol=1e-3, maxiter=100, print_results=False, st_type=None, shard=None):
    """Two-sided Newton iteration on lambda(omega) of L(omega) p = lambda C p (reference: eigensolvers.py:278-348).

    Same update as the reference, including its geometrically decaying relaxation (relaxation *= 0.8 every
    iteration, :336-337) -- which is why the reference's own Newton logs stop short of the fixed-point eigenvalue;
    ``maxiter`` is enforced.  Returns (omega, p) like the reference.  ``shard``: collective run on a distributed mesh (operators
    and flame factors assembled on the rank's part; the brackets p_adj^H L' p are summed over the ranks)."""
    from .eigenvectors import normalize_eigenvector
    from .petsc4py_utils import vector_matrix_vector
    A, B, C = operators.A, operators.B, operators.C
    omega = np.zeros(maxiter + 1, dtype=complex)
    omega[0] = init
    domega = 2 * tol
    k = 0
    s = _fmt(tol)
    relaxation = 1.0
    info("-> Newton solver started.\n")
    p = None
    while abs(domega) > tol and k < maxiter:
        D.assemble_matrix(omega[k])
        if not B:
            L = A + omega[k] ** 2 * C - D.matrix
            dL_domega = 2 * omega[k] * C - D.get_derivative(omega[k])
        else:
            L = A + omega[k] * B + omega[k] ** 2 * C - D.matrix
            dL_domega = B + (2 * omega[k] * C) - D.get_derivative(omega[k])
        # L(omega) p = lambda C p, eigenvalue nearest zero (shift-and-invert at 0), with left eigenvectors
        E = eps_solver(L, -C, 0, nev, two_sided=True, print_results=print_results, st_type=st_type, shard=shard)
        eig = E.getEigenvalue(i)
        omega_dir, p = normalize_eigenvector(operators.mesh, E, i, degree=1, which='right', print_eigs=False)
        omega_adj, p_adj = normalize_eigenvector(operators.mesh, E, i, degree=1, which='left', print_eigs=False)
        p_vec, p_adj_vec = p.x.petsc_vec, p_adj.x.petsc_vec
        num = vector_matrix_vector(p_adj_vec, dL_domega, p_vec)
        den = vector_matrix_vector(p_adj_vec, C, p_vec)
        deig = num / den
        domega = -relaxation * eig / deig
        relaxation *= 0.8
        omega[k + 1] = omega[k] + domega
        print('iter = {:2d},  omega = {}  {}j,  |domega| = {:.2e}'.format(
            k, s.format(omega[k + 1].real), s.format(omega[k + 1].imag), abs(domega)))
        k += 1
        E.destroy()
    return omega[k], p
