"""Passive acoustic operators A, B, B_adj, C on the GPU.

Drop-in for ``helmholtz_x.acoustic_matrices.AcousticMatrices`` (reference: acoustic_matrices.py:10-138):
same constructor signature, same properties / attributes (``A, B, B_adj, C, V, mesh, c, gamma,
bcs_Dirichlet``), same boundary-condition dictionary semantics (membership tests work for dict, set and
plain-string values; :57-97).  The UFL -> FFCx -> MatSetValues pipeline is replaced by the libhx
vertex-centric P1 assembly kernels (include/hx.h, hx_assemble_p1_*).
"""
import numpy as np

from .fem import Function, FunctionSpace
from .parameters_utils import gamma_function, sound_speed_variable_gamma
from .solver_utils import info


class AcousticMatrices:

    def __init__(self, mesh, facet_tags, boundary_conditions, parameter, degree=1):
        self.mesh = mesh
        self.facet_tags = facet_tags
        self.boundary_conditions = boundary_conditions
        self.parameter = parameter
        self.degree = degree
        self.dimension = mesh.topology.dim
        self.fdim = mesh.topology.dim - 1
        self.V = FunctionSpace(mesh, "Lagrange", degree)
        self.bcs_Dirichlet = []
        self._A = self._B = self._B_adj = self._C = None
        # degree 1: the mesh's P1 assembler; degree 2: quadratic space with P1 coefficient fields (hx_assemble_p2_*)
        asm = self.V.assembler()

        print("Degree of basis functions: ", self.degree, "\n")

        if parameter.name == "temperature":
            self.c = sound_speed_variable_gamma(mesh, parameter, degree=degree)
            self.T = parameter
            self._gamma = gamma_function(self.T)
            info("/\\ Temperature function is used for passive flame matrices.")
        else:
            self.c = parameter
            self._gamma = None                 # gamma = 1.4 everywhere: built on first use (a 29 M-cell copy at 5 M DOF)
            info("\\/ Speed of sound function is used for passive flame matrices.")

        robin_facets, robin_coefs = [], []
        for boundary, spec in boundary_conditions.items():
            facets = np.asarray(facet_tags.indices[facet_tags.values == boundary])
            if 'Neumann' in spec:
                info("- Neumann boundaries on boundary " + str(boundary))
            if 'Dirichlet' in spec:
                if degree == 1:
                    self.bcs_Dirichlet.append(np.unique(mesh.facets[facets]))
                else:                      # quadratic space: the vertices and the edge midpoints of the boundary facets
                    self.bcs_Dirichlet.append(np.unique(asm.facet_dofs(mesh.facets[facets])))
                info("- Dirichlet boundary on boundary " + str(boundary))
            R = None
            if 'Robin' in spec:
                R = spec['Robin']
                info("- Robin boundary on boundary " + str(boundary))
            if 'ChokedInlet' in spec:
                g = self._facet_average(self.gamma, facets)
                M = spec['ChokedInlet']
                t = g * M / (1 + (g - 1) * M ** 2)
                R = (1 - t) / (1 + t)
                info("- Choked inlet boundary on boundary " + str(boundary))
            if 'ChokedOutlet' in spec:
                g = self._facet_average(self.gamma, facets)
                M = spec['ChokedOutlet']
                R = (1 - 0.5 * (g - 1) * M) / (1 + 0.5 * (g - 1) * M)
                info("- Choked outlet boundary on boundary " + str(boundary))
            if R is not None:
                Z = (1 + R) / (1 - R)
                robin_facets.append(facets)
                robin_coefs.append(np.full(facets.size, 1j / Z, dtype=np.complex128))

        info("- Passive matrices are assembling..")
        bc_dofs = np.unique(np.concatenate(self.bcs_Dirichlet)) if self.bcs_Dirichlet else None
        # (distributed mesh: every rank sees the same boundary-condition dictionary, so bc_dofs is a list -- possibly empty, for a
        # part that does not touch the boundary -- on all of them or on none: the flag exchange of apply_dirichlet is collective)

        c_dev = asm._field(self.c) if hasattr(asm, "_field") and degree == 1 else self.c    # one upload for A, C and B
        self._A, self._C = asm.stiffness_mass(c_dev)
        if bc_dofs is not None:
            asm.apply_dirichlet(self._A, bc_dofs, 1.0)
        info("- Matrix A is assembled.")

        if robin_facets:
            B = asm.robin(np.concatenate(robin_facets), np.concatenate(robin_coefs), c_dev)
            B_adj = B.copy()
            B_adj.transpose()
            B_adj.conjugate()
            info("- Matrix B is assembled.")
            self._B, self._B_adj = B, B_adj

        if bc_dofs is not None:
            asm.apply_dirichlet(self._C, bc_dofs, 1.0)
        info("- Matrix C is assembled.\n")

    def _facet_average(self, f, facets):
        """(1/|Gamma|) int_Gamma f ds on the tagged facets (tiny boundary-only reduction, host)."""
        x = self.mesh.geometry.x
        fv = self.mesh.facets[facets]
        if fv.shape[1] == 1:
            meas = np.ones(fv.shape[0])
        elif fv.shape[1] == 2:
            meas = np.linalg.norm(x[fv[:, 1]] - x[fv[:, 0]], axis=1)
        else:
            meas = 0.5 * np.linalg.norm(np.cross(x[fv[:, 1]] - x[fv[:, 0]], x[fv[:, 2]] - x[fv[:, 0]]), axis=1)
        if isinstance(f, Function) and f.function_space.kind == "DG0":
            vals = f.x.array[self.mesh.facet_cells[facets]]
        else:
            vals = f.x.array[fv].mean(axis=1)
        part = getattr(self.mesh, "part", None)
        if part is not None:            # distributed mesh: every facet counts on the rank that owns it (acoustic_matrices.py:76-90)
            own = part["facet_owned"][facets]
            num, den = self.mesh.comm.allreduce(np.array([(vals * meas)[own].sum(), meas[own].sum() + 0j]))
            return num / den.real
        return (vals * meas).sum() / meas.sum()

    @property
    def gamma(self):
        """Ratio of specific heats as a Function (acoustic_matrices.py:49-55): gamma(T), or 1.4 when the speed of sound
        itself was given."""
        if self._gamma is None:
            self._gamma = self.c.copy()
            self._gamma.x.array[:] = 1.4
        return self._gamma

    @gamma.setter
    def gamma(self, value):
        self._gamma = value

    @property
    def A(self):
        return self._A

    @property
    def B(self):
        return self._B

    @property
    def B_adj(self):
        return self._B_adj

    @property
    def C(self):
        return self._C
