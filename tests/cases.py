"""Numeric inputs of the reference's example cases (the constants of their params.py files),
shared by the oracle tests and the HIP parity tests.  Sources are relative to
/root/reference/numerical_examples."""
import numpy as np


class Rijke:
    """Longitudinal/NetworkCode/RijkeTube{1D,2D,3D}/params.py."""
    r_gas = 287.0
    gamma = 1.4
    p_amb = 1e5
    rho_amb = 1.22
    rho_u = rho_amb
    rho_d = 0.85
    c_u = np.sqrt(gamma * p_amb / rho_u)
    c_d = np.sqrt(gamma * p_amb / rho_d)
    T_u = c_u ** 2 / (gamma * r_gas)
    T_d = c_d ** 2 / (gamma * r_gas)
    q_0 = -27.008910380099735
    u_b = 0.10066660027273297
    tau = 0.0015
    d_tube = 0.047
    a_f = 0.025
    a_r = 0.025
    target = 200 * 2 * np.pi

    @staticmethod
    def n(dim):
        if dim == 1:
            return 0.1 / (np.pi * Rijke.d_tube ** 2 / 4)
        if dim == 2:
            return 0.1 / (np.pi * Rijke.d_tube / 4)
        return 0.1

    @staticmethod
    def x_f(dim):
        return np.array([[0.0, 0.0, 0.25]]) if dim == 3 else np.array([[0.25, 0.0, 0.0]])

    @staticmethod
    def x_r(dim):
        return np.array([[0.0, 0.0, 0.20]]) if dim == 3 else np.array([[0.20, 0.0, 0.0]])


class PRF:
    """Longitudinal/PRF/RijkeTube1D/params.py (non-dimensional)."""
    r_gas = 287.0
    gamma = 1.4
    p_amb = 1e5
    rho_amb = 1.22
    c_amb = 339
    rho_in_dim = rho_amb
    rho_out_dim = 0.85
    c_in_dim = np.sqrt(gamma * p_amb / rho_in_dim)
    c_out_dim = np.sqrt(gamma * p_amb / rho_out_dim)
    T_in_dim = p_amb / (r_gas * rho_in_dim)
    T_out_dim = p_amb / (r_gas * rho_out_dim)
    R_in = -0.975 - 0.05j
    R_out = -0.975 - 0.05j
    q_0 = 200.0
    u_b = 0.1
    U_ref = c_amb
    p_ref = p_amb
    rho_u = rho_in_dim * U_ref ** 2 / p_ref
    rho_d = rho_out_dim * U_ref ** 2 / p_ref
    c_u = c_in_dim / U_ref
    c_d = c_out_dim / U_ref
    T_u = T_in_dim * r_gas / U_ref ** 2
    T_d = T_out_dim * r_gas / U_ref ** 2
    n = 0.014 / (np.pi * 0.047 ** 2 / 4) / p_ref
    tau = 0.0015 * U_ref
    x_f = np.array([[0.25, 0.0, 0.0]])
    x_r = np.array([[0.20, 0.0, 0.0]])
    a_f = 0.025
    a_r = 0.025
    target = np.pi

    # PRF/RijkeTube2D and RijkeTube3D/params.py: same non-dimensionalisation; the interaction index is divided by the
    # tube cross-section in 1-D, by (pi/4) d in 2-D and not at all in 3-D; flame / reference points on the tube axis
    @staticmethod
    def n_dim(dim):
        return {1: PRF.n, 2: 0.014 / ((np.pi / 4) * 0.047) / PRF.p_ref, 3: 0.014 / PRF.p_ref}[dim]

    @staticmethod
    def point(dim, s):
        return np.array([[s, 0.0, 0.0]]) if dim < 3 else np.array([[0.0, 0.0, s]])

    robin_tags = {2: (1, 4), 3: (3, 2)}          # (inlet, outlet) physical tags of the gmsh meshes


class Micca:
    """AnnularCombustor/Micca/fullAnnulus/params.py."""
    l_cc = 0.2
    r_p = 0.14
    d_2 = 0.035
    r_f = r_p + d_2
    theta = np.deg2rad(22.5)
    z_f = 0.0
    r_r = r_f
    z_r = -0.02
    r = 287.0
    gamma = 1.4
    p_amb = 101325.0
    T_amb = 300.0
    rho_amb = p_amb / (r * T_amb)
    T_a = 1521.0
    T_b = 1200.0
    q_0 = 2080.0
    u_b = 0.66
    N_sector = 16
    R_outlet = -0.875 - 0.2j
    target = 3225.120 + 481.0j

    @staticmethod
    def x_f():
        return np.array([[Micca.r_f * np.cos(i * Micca.theta), Micca.r_f * np.sin(i * Micca.theta), Micca.z_f]
                         for i in range(Micca.N_sector)])

    @staticmethod
    def x_r():
        return np.array([[Micca.r_r * np.cos(i * Micca.theta), Micca.r_r * np.sin(i * Micca.theta), Micca.z_r]
                         for i in range(Micca.N_sector)])

    @staticmethod
    def c_of_z(z):
        """params.py:53-70 (cell-midpoint rule; note the strict inequalities)."""
        g, r = Micca.gamma, Micca.r
        mid = np.sqrt(g * r * ((Micca.T_b - Micca.T_a) * (z / Micca.l_cc) ** 2 + Micca.T_a))
        return np.where(z < 0, np.sqrt(g * r * Micca.T_amb),
                        np.where((z > 0) & (z < Micca.l_cc), mid, np.sqrt(g * r * Micca.T_b)))


def dispersion_root(k0, Z, L=0.4, h=0.1):
    """Root of manufacturedSolution/matlab_data/dispersion2d.m:1-3 by Newton from k0."""
    def f(k):
        ky = np.sqrt(k ** 2 - (np.pi / L) ** 2 + 0j)
        return np.exp(2j * ky * h) * (ky - k / Z) - (ky + k / Z)
    k = complex(k0)
    for _ in range(60):
        d = (f(k + 1e-7) - f(k - 1e-7)) / 2e-7
        k = k - f(k) / d
    return k


class RijkeFFD:
    """ShapeSensitivities/RijkeFFD/params.py + main.py (degree 1)."""
    r_gas = 287.0
    gamma = 1.4
    p_amb = 1e5
    rho_u = 1.22
    rho_d = 0.85
    T_in = p_amb / (r_gas * rho_u)
    T_out = p_amb / (r_gas * rho_d)
    R_in = -0.975 - 0.05j
    R_out = -0.975 - 0.05j
    q_0 = 200.0
    u_b = 0.1
    n = 0.014
    tau = 0.0015
    x_f = np.array([[0.0, 0.0, 0.25]])
    x_r = np.array([[0.0, 0.0, 0.20]])
    a_f = 0.025
    a_r = 0.025
    target = 180 * 2 * np.pi
    bcs = {1: {'Neumann'}, 2: {'Robin': R_out}, 3: {'Robin': R_in}}


class FlamedDuct:
    """Longitudinal/NetworkCode/FlamedDuct/params.py."""
    p_gas = 100000.0
    r_gas = 287.1
    T_passive = 1000.0
    T_flame = 1500.0
    x_flame = np.array([[0.0, 0.0, 0.50]])
    a_flame = 0.025
    x_ref = np.array([[0.0, 0.0, 0.35]])
    a_ref = 0.025
    n = 1
    tau = 0.2e-3
    q_0 = -57015.232012607579
    u_b = 11.485465769828917
    M0 = 9.2224960671405849e-3
    M1 = 1.1408306741423997e-2
    target_passive = 262 * 2 * np.pi
    target_active = 250 * 2 * np.pi


def rijkeffd_shape_case(golden_dir):
    """The mesh of the reference's degree-2 shape-derivative run (ShapeSensitivities/RijkeFFD/main_shape.py), recovered from
    its result files (tests/golden/make_rijkeffd_shape_fixture.py), with the facet tags gmsh's ``occ.addCylinder`` gives its
    three faces rebuilt by geometry: 1 = lateral surface (Neumann), 2 / 3 = the two end discs (both Robin with the SAME
    reflection coefficient, params.py:27-28, so which disc is which does not matter).
    Returns (x, cells, boundary facets (nf, 3), their cells, facet tags (nf,), goldens dict, npz with p_dir / p_adj)."""
    import json
    import os
    from oracle import meshes
    d = np.load(os.path.join(golden_dir, "rijkeffd_shape", "mesh_and_modes.npz"))
    with open(os.path.join(golden_dir, "rijkeffd_shape", "goldens.json")) as fh:
        g = json.load(fh)
    x, cells = d["x"], d["cells"].astype(np.int32)
    facets, fcell = meshes.boundary_facets(cells)
    assert cells.shape[0] == g["cells"] and facets.shape[0] == g["facets"]          # results.log:1-2
    z = x[facets][:, :, 2]
    tags = np.full(facets.shape[0], 1, dtype=np.int32)
    tags[(z == 1.0).all(axis=1)] = 2
    tags[(z == 0.0).all(axis=1)] = 3
    return x, cells, facets, fcell, tags, g, d
