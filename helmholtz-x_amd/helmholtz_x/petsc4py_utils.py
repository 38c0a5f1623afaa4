"""Small Vec/Mat helpers with the reference's names (helmholtz_x/petsc4py_utils.py:4-111)."""
import numpy as np


def multiply(x0, z):
    x1 = x0.copy()
    x1.scale(z)
    return x1


def conjugate(y0):
    y1 = y0.copy()
    y1.conjugate()
    return y1


def conjugate_function(p):
    p.x.array[:] = np.conjugate(p.x.array)
    return p


def vector_vector(y0, x0):
    """y0.dot(x0) with PETSc semantics: VecDot(y0, x0) = x0^H y0."""
    return y0.dot(x0)


def vector_matrix_vector(y0, A, x0):
    x1 = x0.copy()
    A.mult(x0, x1)
    return vector_vector(y0, x1)


def matrix_vector(Mat, x):
    _, vector = Mat.createVecs()
    Mat.mult(x, vector)
    return vector


def FixSign(x):
    """Divide by the phase of entry 0 so that x[0] is real and positive (petsc4py_utils.py:100-111)."""
    x0 = x[0]
    x.scale(1.0 / (x0 / abs(x0)))
