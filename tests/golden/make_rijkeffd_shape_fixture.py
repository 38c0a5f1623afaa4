#!/usr/bin/env python3
"""Fixture of the reference's degree-2 shape-derivative run (ShapeSensitivities/RijkeFFD/main_shape.py).

``main_shape.py`` re-generates its mesh with gmsh at run time, so the mesh of the run behind
``Results/ShapeDerivatives/eigenvalues.txt`` (25,246 cells, results.log:2) is NOT the committed
``MeshDir/ShapeDerivatives/mesh.*`` (that one is the 25,253-cell 'Original' mesh).  But DOLFINx wrote the mesh of the run
into every result file: ``Results/ShapeDerivatives/p_dir.h5`` holds ``Mesh/Grid/geometry`` (6078 x 3) and
``Mesh/Grid/topology`` (25246 x 4) next to the vertex samples of the P2 modes (``xdmf_writer`` interpolates degree-2
functions to P1 before writing, io_utils.py:48-53).  This script copies that DATA -- mesh arrays, vertex samples of the direct
and adjoint modes, the two 16-digit eigenvalues and the 36 normalised shape derivatives -- into one .npz + one .json.

Run once in the build container (needs /root/reference; nothing else):  python tests/golden/make_rijkeffd_shape_fixture.py
The .h5 files are read with the package's own minimal HDF5 reader."""
import ast
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "helmholtz-x_amd"))
from helmholtz_x._h5lite import H5File  # noqa: E402

CASE = "/root/reference/numerical_examples/ShapeSensitivities/RijkeFFD"


def _trajectories(path):
    """the '+ omega = ...' lines of a results.log, split at every 'Fixed point iteration started' (direct run, adjoint run)"""
    import re
    runs = []
    with open(path) as f:
        for line in f:
            if "Fixed point iteration started" in line:
                runs.append([])
            m = re.match(r"\+ omega = ([+-][0-9.]+)\s+([+-][0-9.]+)j", line)
            if m:
                runs[-1].append([float(m.group(1)), float(m.group(2))])
    return runs


def _dict_file(path):
    """the reference's dict_writer output (io_utils.py:19-23): json.dumps(str(dict))"""
    with open(path) as f:
        return ast.literal_eval(json.loads(f.read()))


if __name__ == "__main__":
    out = {}
    for which in ("dir", "adj"):
        h = H5File(os.path.join(CASE, "Results/ShapeDerivatives/p_%s.h5" % which))
        x = h.read("Mesh/Grid/geometry")
        t = h.read("Mesh/Grid/topology")
        p = h.read("Function/real_f/0").ravel() + 1j * h.read("Function/imag_f/0").ravel()
        if which == "dir":
            out["x"], out["cells"] = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(t, dtype=np.int32)
        else:
            assert np.array_equal(out["x"], x) and np.array_equal(out["cells"], t)
        out["p_" + which] = p
        print("p_%s: %d vertices, %d cells" % (which, x.shape[0], t.shape[0]))
    np.savez_compressed(os.path.join(HERE, "rijkeffd_shape", "mesh_and_modes.npz"), **out)
    eig = _dict_file(os.path.join(CASE, "Results/ShapeDerivatives/eigenvalues.txt"))
    der = _dict_file(os.path.join(CASE, "ShapeDerivatives/normalized.txt"))
    traj = _trajectories(os.path.join(CASE, "Results/ShapeDerivatives/results.log"))
    gold = {"source": "ShapeSensitivities/RijkeFFD: Results/ShapeDerivatives/eigenvalues.txt, ShapeDerivatives/normalized.txt, "
                      "Results/ShapeDerivatives/results.log",
            "cells": 25246, "facets": 7446,
            "direct": [eig["direct"].real, eig["direct"].imag],
            "adjoint": [eig["adjoint"].real, eig["adjoint"].imag],
            "trajectory_direct": traj[0], "trajectory_adjoint": traj[1],      # printed to 8 decimals
            "lattice_lmn": [2, 4, 9],
            # normalized[zeta][phi], zeta = 0..8 (axial control plane), phi = 0..3 (azimuthal control point)
            "normalized": [[[der[z][p].real, der[z][p].imag] for p in range(4)] for z in range(9)]}
    with open(os.path.join(HERE, "rijkeffd_shape", "goldens.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("direct", eig["direct"], "adjoint", eig["adjoint"])
