"""Instance loaders (no GPU): same (W, h) as the reference's loaders on the bundled data files."""
import os

import numpy as np

from conftest import GOLDEN, load_product

INST = os.path.join(GOLDEN, "instances")


def test_droplet_loader_matches_golden_instance():
    P = load_product()
    W, h = P.instances.txt_to_A_droplet(os.path.join(INST, "chimera128__001.txt"))
    from conftest import golden
    g = golden("mcmc_fixed_chimera128_s0")          # stores J = -W / max|W| in CSR, h = -h_file / max|W|
    assert W.shape == (128, 128) and h.shape == (128, 1)
    nf = np.max(np.abs(W.data))
    assert np.array_equal(W.indptr, g["indptr"]) and np.array_equal(W.indices, g["indices"])
    assert np.allclose(-W.data / nf, g["data"], rtol=0, atol=0)
    assert np.allclose(-h.reshape(-1) / nf, g["h"], rtol=0, atol=0)
    assert (abs(W - W.T)).nnz == 0
    assert W.max(axis=1).shape[0] == 128 and np.max(np.diff(W.indptr)) <= 6      # Chimera: degree <= 6


def test_wishart_and_dcl_loaders():
    P = load_product()
    W, h = P.instances.txt_to_A_wishart(os.path.join(INST, "wishart_N10_a0.50__wishart_planting_N_10_alpha_0.50_inst_1.txt"))
    assert W.shape == (10, 10) and np.all(h == 0) and (abs(W - W.T)).nnz == 0 and W.diagonal().sum() == 0
    W, h = P.instances.txt_to_A_DCL(os.path.join(INST, "DCL_C8__00.txt"))
    assert W.shape[0] == W.shape[1] and np.all(h == 0) and (abs(W - W.T)).nnz == 0
    sol = dict(line.split() for line in open(os.path.join(INST, "DCL_C8__00_sol.txt")) if len(line.split()) == 2)
    assert float(sol["min_energy"]) < 0 and int(sol["nq"]) <= W.shape[0]
