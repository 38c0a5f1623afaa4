"""Coefficient fields of the example cases (reference: helmholtz_x/parameters_utils.py:8-247).

All are closed-form nodal (P1) or cellwise (DG0) arrays -- host input preparation, not GPU work."""
import numpy as np

from .dolfinx_utils import normalize
from .fem import Function, FunctionSpace

R_GAS_VARIABLE_GAMMA = 287.1     # the constant the reference uses in gamma_function / sound_speed_variable_gamma


def gaussian(x, x_ref, sigma, n):
    x_ref = np.asarray(x_ref, dtype=float)
    if x_ref.ndim == 2:
        x_ref = x_ref[0]
    r2 = sum((x[i] - x_ref[i]) ** 2 for i in range(n))
    return np.exp(-r2 / (2 * sigma ** 2)) / (sigma ** n * (2 * np.pi) ** (n / 2))


def gaussianFunction(mesh, x_r, a_r, degree=1):
    w = Function(FunctionSpace(mesh, "CG", degree))
    w.interpolate(lambda x: gaussian(x, x_r, a_r, mesh.geometry.dim))
    return normalize(w)


def halfGaussianFunction(mesh, x_flame, a_flame, degree=1):
    h = gaussianFunction(mesh, x_flame, a_flame, degree=degree)
    zf = np.asarray(x_flame, dtype=float).reshape(-1, 3)[0][2]
    h.x.array[mesh.geometry.x[:, 2] < zf] = 0.0
    return normalize(h)


def gamma_function(temperature):
    if isinstance(temperature, Function):
        g = Function(temperature.function_space, "gamma")
        cp = 973.60091 + 0.1333 * temperature.x.array
        g.x.array[:] = cp / (cp - R_GAS_VARIABLE_GAMMA)
        return g
    cp = 973.60091 + 0.1333 * temperature
    return cp / (cp - R_GAS_VARIABLE_GAMMA)


def sound_speed_variable_gamma(mesh, temperature, degree=1):
    c = Function(temperature.function_space, "soundspeed")
    c.x.array[:] = np.sqrt(gamma_function(temperature).x.array * R_GAS_VARIABLE_GAMMA * temperature.x.array)
    return c


def sound_speed(temperature):
    c = Function(temperature.function_space, "soundspeed")
    c.x.array[:] = 20.05 * np.sqrt(temperature.x.array)
    return c


def density_step(x, x_f, sigma, rho_d, rho_u):
    return rho_u + (rho_d - rho_u) / 2 * (1 + np.tanh((x - x_f) / sigma))


def _axis_and_position(mesh, x_f):
    axis = 2 if mesh.geometry.dim == 3 else 0
    return axis, np.asarray(x_f, dtype=float).reshape(-1, 3)[0][axis]


def rho_step(mesh, x_f, a_f, rho_d, rho_u, degree=1):
    axis, xf = _axis_and_position(mesh, x_f)
    rho = Function(FunctionSpace(mesh, "CG", degree), "rho")
    rho.x.array[:] = density_step(mesh.geometry.x[:, axis], xf, a_f, rho_d, rho_u)
    return rho


def rho_ideal(temperature, p_0, r_gas):
    rho = Function(temperature.function_space, "rho")
    rho.x.array[:] = p_0 / (r_gas * temperature.x.array)
    return rho


def _step(mesh, x_f, up, down, name):
    axis, xf = _axis_and_position(mesh, x_f)
    f = Function(FunctionSpace(mesh, "CG", 1), name)
    f.x.array[:] = np.where(mesh.geometry.x[:, axis] < xf, up, down)
    return f


def c_step(mesh, x_f, c_u, c_d):
    return _step(mesh, x_f, c_u, c_d, "soundspeed")


def temperature_step(mesh, x_f, T_u, T_d, degree=1):
    return _step(mesh, x_f, T_u, T_d, "temperature")


def c_uniform(mesh, sos, degree=1):
    c = Function(FunctionSpace(mesh, "CG", degree), "soundspeed")
    c.x.array[:] = sos
    return c


def temperature_uniform(mesh, temp):
    T = Function(FunctionSpace(mesh, "CG", 1), "temperature")
    T.x.array[:] = temp
    return T


def temperature(mesh, soundSpeed):
    T = Function(FunctionSpace(mesh, "CG", 1), "temperature")
    cs = soundSpeed.x.array if isinstance(soundSpeed, Function) else soundSpeed
    T.x.array[:] = np.square(cs) / (R_GAS_VARIABLE_GAMMA * 1.4)
    return T


def Q_volumetric(mesh, subdomains, Q_total, flame_tag=0, degree=0):
    """DG0 heat-release density: Q_total / V_flame on the cells tagged ``flame_tag``."""
    q = Function(FunctionSpace(mesh, "DG", degree), "q")
    cells = subdomains.find(flame_tag)
    q.x.array[cells] = Q_total / _volume(mesh, cells)
    return q


def _volume(mesh, cells):
    """Volume of a set of cells; on a distributed local mesh every cell counts on the one rank that owns it and the ranks'
    sums are added (the reference: MPI allreduce, parameters_utils.py:216,236)."""
    vol = mesh.assembler().cell_volumes()[cells]
    part = getattr(mesh, "part", None)
    if part is None:
        return vol.sum()
    return float(mesh.comm.allreduce(float(vol[part["cell_owned"][cells]].sum())))


def Q_multiple(mesh, subdomains, N_sector, degree=0):
    """DG0 field equal to 1/V_f on the cells of flame f = 0..N_sector-1."""
    q = Function(FunctionSpace(mesh, "DG", degree), "q")
    for flame in range(N_sector):
        cells = subdomains.find(flame)
        v = _volume(mesh, cells)          # (collective on a distributed mesh: every rank takes part for every flame)
        if cells.size:
            q.x.array[cells] = 1.0 / v
    return q
