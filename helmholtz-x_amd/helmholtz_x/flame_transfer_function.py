"""Flame transfer functions (scalar host math; reference: helmholtz_x/flame_transfer_function.py:5-42)."""
from math import factorial

import numpy as np


class nTau:
    """FTF(omega) = n exp(+i omega tau)."""

    def __init__(self, n, tau):
        self.n = n
        self.tau = tau

    def __call__(self, omega):
        return self.n * np.exp(1j * omega * self.tau)

    def derivative(self, omega):
        return 1j * self.tau * self(omega)


class stateSpace:
    """FTF(omega) = conj( c (i conj(omega) I - A)^-1 b + d ), state-space fit of a measured FTF."""

    def __init__(self, S1, s2, s3, s4):
        self.A = np.asarray(S1)
        self.b = np.asarray(s2)
        self.c = np.asarray(s3)
        self.d = np.asarray(s4)
        self.Id = np.eye(*self.A.shape)

    def _resolvent_power(self, omega, k):
        R = np.linalg.matrix_power(1j * np.conj(omega) * self.Id - self.A, -(k + 1))
        return ((-1j) ** k * factorial(k)) * (self.c @ R @ self.b)

    def __call__(self, omega):
        return np.conj((self._resolvent_power(omega, 0) + self.d)[0][0])

    def derivative(self, omega):
        return np.conj(self._resolvent_power(omega, 1)[0][0])
