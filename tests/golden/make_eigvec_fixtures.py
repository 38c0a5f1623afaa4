#!/usr/bin/env python3
"""Extract the normalised eigenvectors the reference commits next to its logs (DOLFINx XDMF/HDF5 output of
``normalize_eigenvector``, eigenvectors.py:11-64) into small .npz fixtures: node coordinates + real/imag nodal values.

Run once in the build container (needs /root/reference; nothing else):  python tests/golden/make_eigvec_fixtures.py
The .h5 files are read with the package's own minimal HDF5 reader; only DATA is copied."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "helmholtz-x_amd"))
from helmholtz_x._h5lite import H5File  # noqa: E402

REF = "/root/reference/numerical_examples"
CASES = {
    "prf1d_dir": "Longitudinal/PRF/RijkeTube1D/Results/Active/p_dir.h5",
    "prf1d_adj": "Longitudinal/PRF/RijkeTube1D/Results/Active/p_adj.h5",
    "prf2d_dir": "Longitudinal/PRF/RijkeTube2D/Results/Active/p.h5",
    "prf2d_adj": "Longitudinal/PRF/RijkeTube2D/Results/Active/p_adj.h5",
    "prf3d_dir": "Longitudinal/PRF/RijkeTube3D/Results/Active/p.h5",
    "prf3d_adj": "Longitudinal/PRF/RijkeTube3D/Results/Active/p_adj.h5",
    "rijke1d_dir": "Longitudinal/NetworkCode/RijkeTube1D/Results/Active/p.h5",
    "rijke1d_adj": "Longitudinal/NetworkCode/RijkeTube1D/Results/Active/p_adj.h5",
    "rijke2d_dir": "Longitudinal/NetworkCode/RijkeTube2D/Results/Active/p.h5",
    "rijke3d_dir": "Longitudinal/NetworkCode/RijkeTube3D/Results/Active/p.h5",
    "rijkeffd_dir": "ShapeSensitivities/RijkeFFD/Results/Original/p_dir.h5",
    "rijkeffd_adj": "ShapeSensitivities/RijkeFFD/Results/Original/p_adj.h5",
    "rijkeffd_opt_dir": "ShapeSensitivities/RijkeFFD/Results/Optimized/p_dir.h5",
    # Bloch sector: besides the modes, the node ORDER of these files is data the Bloch goldens depend on -- the reference
    # pairs master/slave DOFs by DOLFINx index (bloch_operator.py:30-40), and DOLFINx writes nodes in that order
    "micca_bloch_p1": "AnnularCombustor/Micca/bloch/Results/Passive/p_1.h5",
    "micca_bloch_p2": "AnnularCombustor/Micca/bloch/Results/Passive/p_2.h5",
    "micca_bloch_p3": "AnnularCombustor/Micca/bloch/Results/Passive/p_3.h5",
    "micca_bloch_active_p1": "AnnularCombustor/Micca/bloch/Results/Active/p_1_dir.h5",
}

if __name__ == "__main__":
    out = {}
    for name, rel in CASES.items():
        h = H5File(os.path.join(REF, rel))
        geo = next(k for k in h.datasets if k.endswith("geometry"))
        re = next(k for k in h.datasets if "real_" in k)
        im = next(k for k in h.datasets if "imag_" in k)
        x = h.read(geo)
        out[name + "_x"] = x
        out[name + "_p"] = h.read(re).ravel() + 1j * h.read(im).ravel()
        print("%-14s %6d nodes  dim %d  %s" % (name, x.shape[0], x.shape[1], rel))
    np.savez_compressed(os.path.join(HERE, "eigenvectors", "reference_eigenvectors.npz"), **out)
