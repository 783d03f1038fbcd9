#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the read-only Python reference in this container.

Container-only tool (the reference never travels to the GPU box; only the small fixtures do).
Rules followed: PYTHONDONTWRITEBYTECODE=1, MPLBACKEND=Agg, cwd under /tmp,
`cachetools` provided by tools/_standins (never exercised: use_hash_table stays False).

Usage:   python tools/make_golden.py            (re-execs itself with the right environment)

Every fixture stores INPUTS (instance, start state, hyper-parameters, both RNG seeds) and the
reference's OUTPUTS.  The RNG stream itself is not stored: the legacy `np.random` MT19937 stream
and stdlib `random` are frozen by NumPy/CPython policy, so re-seeding reproduces it anywhere.
"""
import io
import os
import re
import sys
import contextlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

if os.environ.get("NLMC_GOLDEN_CHILD") != "1":
    import subprocess
    import tempfile
    env = dict(os.environ)
    env.update(PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg", NLMC_GOLDEN_CHILD="1",
               PYTHONPATH=os.pathsep.join([os.path.join(REPO, "tools", "_standins"), REF]))
    work = tempfile.mkdtemp(prefix="nlmc_golden_", dir="/tmp")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], cwd=work, env=env))

import random  # noqa: E402
from concurrent.futures import Future  # noqa: E402

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

from NMC import nmc as ref_nmc  # noqa: E402
from NPT import npt as ref_npt  # noqa: E402
from NPT import apt_ICM as ref_icm  # noqa: E402
from NPT import apt_preprocessor as ref_pre  # noqa: E402


class InlineExecutor:
    """Deterministic stand-in for ProcessPoolExecutor: one global RNG stream in program order."""

    def __init__(self, max_workers=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def submit(self, fn, *args, **kw):
        f = Future()
        try:
            f.set_result(fn(*args, **kw))
        except BaseException as e:  # noqa: BLE001
            f.set_exception(e)
        return f


ref_npt.ProcessPoolExecutor = InlineExecutor
ref_pre.ProcessPoolExecutor = InlineExecutor
ref_pre.as_completed = lambda futs: list(futs)


# ----------------------------------------------------------------------------------------------
# instances
# ----------------------------------------------------------------------------------------------
def inst_gauss_dense(N, seed):
    r = np.random.default_rng(seed)
    h = r.standard_normal(N)
    J = np.zeros((N, N))
    iu = np.triu_indices(N, 1)
    J[iu] = r.standard_normal(len(iu[0]))
    J += J.T
    return J, h


def inst_pmj_sparse(N, seed, with_h=False):
    """SURVEY.md §8d make_instance: exactly 3N distinct undirected edges, J=+-1."""
    r = np.random.default_rng(seed)
    edges = set()
    while len(edges) < 3 * N:
        i, j = (int(v) for v in r.integers(0, N, 2))
        if i != j:
            edges.add((min(i, j), max(i, j)))
    edges = sorted(edges)
    J = np.zeros((N, N))
    for (i, j) in edges:
        J[i, j] = J[j, i] = r.choice([-1.0, 1.0])
    h = r.standard_normal(N) * 0.3 if with_h else np.zeros(N)
    return J, h


def inst_gauss_sparse(N, seed):
    J, _ = inst_pmj_sparse(N, seed)
    r = np.random.default_rng(seed + 1)
    W = np.triu(r.standard_normal((N, N)), 1)
    W = W + W.T
    return J * np.abs(W), r.standard_normal(N) * 0.5


def load_droplet(path):
    """Same convention as NMC/examples/chimera_example.py:8-40 (1-based, diagonal -> h), then J=-W, h=-h."""
    W, h = {}, {}
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        a, b, v = line.split()
        a, b, v = int(a) - 1, int(b) - 1, float(v)
        if a == b:
            h[a] = v
        else:
            W[(a, b)] = v
            W[(b, a)] = v
    N = max(max(k) for k in W) + 1
    J = np.zeros((N, N))
    for (i, j), v in W.items():
        J[i, j] = v
    hv = np.array([h.get(i, 0.0) for i in range(N)])
    return -J, -hv


def csr_parts(J):
    A = sp.csr_matrix(J)
    A.sort_indices()
    return dict(N=np.int64(A.shape[0]), indptr=A.indptr.astype(np.int32), indices=A.indices.astype(np.int32),
                data=A.data.astype(np.float64))


def energies_of(M, J, h):
    return np.array([-(M[:, i] @ J @ M[:, i] / 2 + M[:, i] @ h) for i in range(M.shape[1])])


def save(name, **kw):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"[golden] {name}: {os.path.getsize(path) / 1024:.1f} KB", file=sys.__stdout__)


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


# ----------------------------------------------------------------------------------------------
# G1/G2: MCMC fixed beta and anneal   (NMC/nmc.py:28-91)
# ----------------------------------------------------------------------------------------------
def gen_mcmc():
    chim = os.path.join(REF, "NMC/examples/Chimera_droplet_instances/chimera128_spinglass_power/001.txt")
    cases = [
        ("gauss16", inst_gauss_dense(16, 7), 1.3, 30),
        ("pmj100", inst_pmj_sparse(100, 20250225), 3.0, 25),
        ("pmj100h", inst_pmj_sparse(100, 11, with_h=True), 0.8, 20),
        ("gsparse60", inst_gauss_sparse(60, 5), 2.0, 25),
        ("chimera128", load_droplet(chim), 3.0, 20),
    ]
    for name, (J, h), beta, sweeps in cases:
        nf = np.max(np.abs(J))
        Jn, hn = J / nf, h / nf
        obj = ref_nmc.NMC(Jn, hn)
        for seed in (0, 1, 12345):
            np.random.seed(seed)
            m0 = np.sign(2 * np.random.rand(Jn.shape[0]) - 1)
            M = obj.MCMC(sweeps, m0.copy(), beta, Jn, hn)
            save(f"mcmc_fixed_{name}_s{seed}", **csr_parts(Jn), h=hn, m_start=m0.astype(np.int8), beta=beta,
                 seed=seed, num_sweeps=sweeps, M=M.T.astype(np.int8), energies=energies_of(M, Jn, hn))
        for spb in (1, 3):
            seed = 77 + spb
            np.random.seed(seed)
            m0 = np.sign(2 * np.random.rand(Jn.shape[0]) - 1)
            ns = 24
            M = obj.MCMC(ns, m0.copy(), beta, Jn, hn, anneal=True, sweeps_per_beta=spb, initial_beta=0.1)
            save(f"mcmc_anneal_{name}_spb{spb}", **csr_parts(Jn), h=hn, m_start=m0.astype(np.int8), beta=beta,
                 seed=seed, num_sweeps=ns, sweeps_per_beta=spb, initial_beta=0.1, M=M.T.astype(np.int8),
                 energies=energies_of(M, Jn, hn))
    # apt_ICM.MCMC variant (NPT/apt_ICM.py:52-93): J,h from self, h is [N,1], dense J.dot
    J, h = inst_gauss_sparse(40, 9)
    nf = np.max(np.abs(J))
    Jn, hn = J / nf, h / nf
    icm = ref_icm.APT_ICM(Jn.copy(), hn.copy())
    np.random.seed(5)
    m0 = np.sign(2 * np.random.rand(40) - 1)
    M = icm.MCMC(15, m0.copy(), 1.7)
    save("mcmc_icmvariant_gsparse40_s5", **csr_parts(Jn), h=hn, m_start=m0.astype(np.int8), beta=1.7, seed=5,
         num_sweeps=15, M=M.T.astype(np.int8), energies=energies_of(M, Jn, hn))


def gen_mcmc_large():
    """The same call at sizes where a sweep has many multi-wave levels (N = 1000 +-J, N = 2500 Gaussian with fields):
    a few sweeps each, so that the fixtures stay small."""
    for name, (J, h), beta, sweeps, seed in (("pmj1000", inst_pmj_sparse(1000, 31), 1.2, 6, 3),
                                              ("gsparse2500", inst_gauss_sparse(2500, 41), 1.6, 3, 4)):
        nf = np.max(np.abs(J))
        Jn, hn = J / nf, h / nf
        obj = ref_nmc.NMC(Jn, hn)
        np.random.seed(seed)
        m0 = np.sign(2 * np.random.rand(Jn.shape[0]) - 1)
        M = obj.MCMC(sweeps, m0.copy(), beta, Jn, hn)
        save(f"mcmc_fixed_{name}_s{seed}", **csr_parts(Jn), h=hn, m_start=m0.astype(np.int8), beta=beta,
             seed=seed, num_sweeps=sweeps, M=M.T.astype(np.int8), energies=energies_of(M, Jn, hn))


# ----------------------------------------------------------------------------------------------
# G3: NMC_subroutine with clusters PROVIDED (bypasses LBP)   (NMC/nmc.py:320-440, NPT/npt.py:357-477)
# ----------------------------------------------------------------------------------------------
def gen_nmc_subroutine():
    J, h = inst_gauss_sparse(30, 3)
    nf = np.max(np.abs(J))
    Jn, hn = J / nf, h / nf
    clusters = np.array([2, 3, 5, 11, 17, 18, 29])
    for variant, mod, cls in (("nmc", ref_nmc, "NMC"), ("npt", ref_npt, "NPT")):
        for (cycles, S, fuf, skip, seed) in ((2, 6, 1, 1, 4), (3, 8, 2, 2, 9)):
            obj = getattr(mod, cls)(Jn.copy(), hn.copy())
            np.random.seed(seed)
            m_star = np.sign(2 * np.random.rand(30) - 1)
            with quiet():
                Mo, Eo, Emin, cl = obj.NMC_subroutine(m_star.copy(), cycles, S, fuf, skip, 3.0, 20, 3, 0.01, 0.9,
                                                      0.9999999, 0.999999, 10, np.finfo(float).eps,
                                                      all_clusters=clusters.copy())
            save(f"nmc_subroutine_{variant}_c{cycles}_s{seed}", **csr_parts(Jn), h=hn, m_star=m_star.astype(np.int8),
                 clusters=clusters, num_cycles=cycles, num_sweeps_per_NMC_phase=S, full_update_frequency=fuf,
                 M_skip=skip, global_beta=3.0, temp_x=20.0, seed=seed, M_overall=Mo.T.astype(np.int8),
                 energy_overall=Eo, min_energy=Emin)


# ----------------------------------------------------------------------------------------------
# G7: LBP + full NMC.run   (NMC/nmc.py:93-318, 442-520)
# ----------------------------------------------------------------------------------------------
def gen_lbp_and_run():
    chim = os.path.join(REF, "NMC/examples/Chimera_droplet_instances/chimera128_spinglass_power/001.txt")
    for name, (J, h) in (("gsparse60", inst_gauss_sparse(60, 5)), ("chimera128", load_droplet(chim)),
                         ("gauss10", inst_gauss_dense(10, 21))):
        nf = np.max(np.abs(J))
        Jn, hn = J / nf, h / nf
        N = Jn.shape[0]
        obj = ref_nmc.NMC(Jn.copy(), hn.copy())
        np.random.seed(3)
        m_star = np.sign(2 * np.random.rand(N) - 1)
        # a decent m_star: short anneal
        M = obj.MCMC(40, m_star.copy(), 3.0, Jn, hn, anneal=True)
        E = energies_of(M, Jn, hn)
        m_star = M[:, int(np.argmin(E))].copy()
        eps = np.abs(hn) + np.sum(np.abs(Jn), axis=1)
        with quiet():
            clusters, marg, _, _, _ = obj.LBP_convexified(3, 0.01, 0.9, m_star.copy(), eps, np.finfo(float).eps, 100,
                                                          0.9999999, 0.999999, 3.0)
        lam = np.array(sorted(marg.keys(), reverse=True))
        save(f"lbp_{name}", **csr_parts(Jn), h=hn, m_star=m_star.astype(np.int8), global_beta=3.0,
             lambda_start=3.0, lambda_end=0.01, lambda_reduction_factor=0.9, max_iterations=100,
             threshold_initial=0.9999999, threshold_cutoff=0.999999,
             lambdas=lam, marginals=np.array([marg[l] for l in lam]),
             cluster_sizes=np.array([len(c) for c in clusters], dtype=np.int64),
             clusters_concat=(np.concatenate(clusters).astype(np.int64) if clusters else np.zeros(0, np.int64)))
        # full run (reference mutates self.J/self.h: give it the UN-normalised instance)
        obj = ref_nmc.NMC(J.copy(), h.copy())
        np.random.seed(2024)
        with quiet():
            Mo, Eo, Emin = obj.run(30, 8, 2, 1, 1, 20, 3, 3, 0.01, 0.9, 0.9999999, 0.999999, 100,
                                   np.finfo(float).eps, use_hash_table=False)
        save(f"nmc_run_{name}", **csr_parts(J), h=h, seed=2024, num_sweeps_initial=30, num_sweeps_per_NMC_phase=8,
             num_NMC_cycles=2, full_update_frequency=1, M_skip=1, temp_x=20.0, global_beta=3.0, lambda_start=3.0,
             lambda_end=0.01, lambda_reduction_factor=0.9, threshold_initial=0.9999999, threshold_cutoff=0.999999,
             max_iterations=100, M_overall=Mo.T.astype(np.int8), energy_overall=Eo, min_energy=Emin)


# ----------------------------------------------------------------------------------------------
# G12: where the reference's lambda continuation stops (NMC/nmc.py:139-150) at the C3 shape.  The stopping lambda is decided by
# rounding noise (tolerance = machine epsilon), so it is a STATISTIC of the arithmetic: 64 seeds, lambdas processed per seed.
# Seeds: chain states of the device engine after 10^3 sweeps at beta = 3 (scripts/lbp_c3_probe.py DUMP=...), stored in the fixture.
# ----------------------------------------------------------------------------------------------
def _lbpstat_one(args):
    Jd, m_star, eps = args
    obj = ref_nmc.NMC(Jd.copy(), np.zeros(Jd.shape[0]))
    with quiet():
        try:
            _, marg, _, _, _ = obj.LBP_convexified(3, 0.01, 0.9, m_star.copy(), eps, np.finfo(float).eps, 100, 0.9999999, 0.999999, 3.0)
        except ValueError:
            return 0
    return len(marg)


def gen_lbp_lambda_stats():
    from multiprocessing import Pool
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import make_instance
    dump = os.environ.get("NLMC_LBP_DUMP", os.path.join(REPO, "gpurun_out", "r3", "dump_rx", "lbp_probe_sw1000.npz"))
    ms = np.load(dump)["ms"].astype(np.float64)
    J, h = make_instance(1000)
    Jd = J.toarray()
    eps = np.abs(h) + np.sum(np.abs(Jd), axis=1)
    with Pool(8) as pool:
        nl = pool.map(_lbpstat_one, [(Jd, ms[p], eps) for p in range(ms.shape[0])])
    save("stats_lbp_lambdas_c3", instance="tests/helpers.make_instance(1000)", m_star=ms.astype(np.int8), global_beta=3.0, lambda_start=3.0,
         lambda_end=0.01, lambda_reduction_factor=0.9, max_iterations=100, n_lambdas_reference=np.array(nl, dtype=np.int32))


# ----------------------------------------------------------------------------------------------
# G4: NPT.run (inline executor)   (NPT/npt.py:535-700)
# ----------------------------------------------------------------------------------------------
PAIR_RE = re.compile(r"Selected pair indices: (\d+), (\d+)")


def parse_swap_log(text):
    pairs, acc = [], []
    for line in text.splitlines():
        m = PAIR_RE.search(line)
        if m:
            pairs.append((int(m.group(1)), int(m.group(2))))
            acc.append(0)
        elif line.startswith("Swapping") and acc:
            acc[-1] = 1
    return np.array(pairs, dtype=np.int32).reshape(-1, 2), np.array(acc, dtype=np.int8)


def gen_npt():
    cases = [
        ("pmj40_plain", inst_pmj_sparse(40, 8, with_h=True), [False] * 6, 6, 60, 30, 10, 2, 31),
        ("gsparse30_mixed", inst_gauss_sparse(30, 3), [False, False, True, True], 4, 40, 20, 5, 1, 32),
        ("gauss10_unit", inst_gauss_dense(10, 1), [False, False, True, True], 4, 100, 100, 10, 1, 33),
    ]
    for name, (J, h), doNMC, R, nsw, nread, nswap, npairs, seed in cases:
        beta_list = np.linspace(0.5, 2.0, R)
        obj = ref_npt.NPT(J.copy(), h.copy())
        np.random.seed(seed)
        random.seed(seed)
        gb = 1 / 0.366838 * 5
        with quiet() as buf:
            M, Energy = obj.run(beta_list=beta_list, num_replicas=R, doNMC=list(doNMC), num_sweeps_MCMC=nsw,
                                num_sweeps_read=nread, num_swap_attempts=nswap, num_swapping_pairs=npairs,
                                num_cycles=2, full_update_frequency=1, M_skip=1, temp_x=20, global_beta=gb,
                                lambda_start=3, lambda_end=0.01, lambda_reduction_factor=0.9,
                                threshold_initial=0.9999999, threshold_cutoff=0.999999, max_iterations=10,
                                tolerance=np.finfo(float).eps, use_hash_table=False, num_cores=1)
        pairs, acc = parse_swap_log(buf.getvalue())
        save(f"npt_run_{name}", **csr_parts(J), h=np.asarray(h).reshape(-1), beta_list=beta_list, num_replicas=R,
             doNMC=np.array(doNMC, dtype=np.int8), num_sweeps_MCMC=nsw, num_sweeps_read=nread,
             num_swap_attempts=nswap, num_swapping_pairs=npairs, num_cycles=2, global_beta=gb, seed=seed,
             max_iterations=10, M=M.astype(np.int8), Energy=Energy, swap_pairs=pairs, swap_accepted=acc)


# ----------------------------------------------------------------------------------------------
# G5: APT_ICM.run + find_disagreement_clusters   (NPT/apt_ICM.py:116-305)
# ----------------------------------------------------------------------------------------------
def gen_icm():
    J, h = inst_pmj_sparse(24, 13, with_h=True)
    obj = ref_icm.APT_ICM(J.copy(), h.copy())
    r = np.random.default_rng(99)
    s1s, s2s, sizes, concat = [], [], [], []
    for t in range(6):
        s1 = r.choice([-1.0, 1.0], 24)
        s2 = s1.copy()
        flip = r.random(24) < (0.15 + 0.12 * t)
        s2[flip] *= -1
        cl = obj.find_disagreement_clusters(s1, s2, J)
        s1s.append(s1)
        s2s.append(s2)
        sizes.append(np.array([len(c) for c in cl], dtype=np.int64))
        concat.append(np.concatenate([np.sort(np.array(c, dtype=np.int64)) for c in cl])
                      if cl else np.zeros(0, np.int64))
    save("icm_clusters_pmj24", **csr_parts(J), s1=np.array(s1s, dtype=np.int8), s2=np.array(s2s, dtype=np.int8),
         n_clusters=np.array([len(s) for s in sizes], dtype=np.int64), sizes=np.concatenate(sizes),
         members=np.concatenate(concat))

    for name, (J, h), R, nsw, nread, nswap, npairs, seed in (
            ("pmj12_S1", inst_pmj_sparse(12, 2, with_h=True), 3, 6, 6, 6, 1, 41),
            ("pmj12_S5", inst_pmj_sparse(12, 2, with_h=True), 3, 20, 8, 4, 1, 42),
            ("gauss10_unit", inst_gauss_dense(10, 1), 4, 100, 100, 10, 1, 43)):
        nf = np.max(np.abs(J))
        Jn, hn = J / nf, h / nf
        beta_list = np.linspace(0.4, 1.6, R)
        obj = ref_icm.APT_ICM(Jn.copy(), hn.copy())
        np.random.seed(seed)
        random.seed(seed)
        with quiet() as buf:
            M, Energy = obj.run(beta_list, num_replicas=R, num_sweeps_MCMC=nsw, num_sweeps_read=nread,
                                num_swap_attempts=nswap, num_swapping_pairs=npairs, use_hash_table=0, num_cores=1)
        pairs, acc = parse_swap_log(re.sub(r"swapping", "Swapping", buf.getvalue()))
        save(f"apt_icm_run_{name}", **csr_parts(Jn), h=hn, beta_list=beta_list, num_replicas=R, num_sweeps_MCMC=nsw,
             num_sweeps_read=nread, num_swap_attempts=nswap, num_swapping_pairs=npairs, seed=seed,
             M=M.astype(np.int8), Energy=Energy, swap_pairs=pairs, swap_accepted=acc)


# ----------------------------------------------------------------------------------------------
# f-2: APT_preprocessor.run   (NPT/apt_preprocessor.py:115-204)
# ----------------------------------------------------------------------------------------------
def gen_preprocessor():
    J, h = inst_pmj_sparse(16, 6, with_h=True)
    obj = ref_pre.APT_preprocessor(sp.csr_matrix(J), h.reshape(-1, 1).copy())
    np.random.seed(51)
    with quiet():
        beta, sigma = obj.run(num_sweeps_MCMC=40, num_sweeps_read=20, num_rng=6, beta_start=0.5, alpha=1.25,
                              sigma_E_val=1000, beta_max=4, use_hash_table=0, num_cores=1)
    save("apt_preprocessor_pmj16", **csr_parts(J), h=h, seed=51, num_sweeps_MCMC=40, num_sweeps_read=20, num_rng=6,
         beta_start=0.5, alpha=1.25, sigma_E_val=1000.0, beta_max=4.0, beta=np.array(beta, dtype=np.float64),
         sigma=np.array(sigma, dtype=np.float64))


# ----------------------------------------------------------------------------------------------
# G6: known answers that ship with the reference (data files only)
# ----------------------------------------------------------------------------------------------
def gen_known_answers():
    import shutil
    dst = os.path.join(OUT, "instances")
    os.makedirs(dst, exist_ok=True)
    wdir = os.path.join(REF, "NMC/examples/wishart_small/wishart_planting_N_10_alpha_0.50")
    names = [f"wishart_planting_N_10_alpha_0.50_inst_{i}.txt" for i in (1, 2, 3)]
    for f in names + ["gs_energies.txt"]:
        shutil.copyfile(os.path.join(wdir, f), os.path.join(dst, "wishart_N10_a0.50__" + f))
    cdir = os.path.join(REF, "NMC/examples/Chimera_droplet_instances/chimera128_spinglass_power")
    shutil.copyfile(os.path.join(cdir, "001.txt"), os.path.join(dst, "chimera128__001.txt"))
    with open(os.path.join(cdir, "groundstates_otn2d.txt")) as f, \
            open(os.path.join(dst, "chimera128__groundstate_001.txt"), "w") as g:
        g.write(f.readline())
    c2dir = os.path.join(REF, "NMC/examples/Chimera_droplet_instances/chimera2048_spinglass_power")
    shutil.copyfile(os.path.join(c2dir, "001.txt"), os.path.join(dst, "chimera2048__001.txt"))
    with open(os.path.join(c2dir, "groundstates_otn2d.txt")) as f, \
            open(os.path.join(dst, "chimera2048__groundstate_001.txt"), "w") as g:
        g.write(f.readline())
    ddir = os.path.join(REF, "NMC/examples/DCL_instances/C8")
    for f in sorted(os.listdir(ddir))[:2]:
        shutil.copyfile(os.path.join(ddir, f), os.path.join(dst, "DCL_C8__" + f))
    print("[golden] copied known-answer data files:", sorted(os.listdir(dst)), file=sys.__stdout__)


# ----------------------------------------------------------------------------------------------
# G8: STATISTICS of the reference's Markov kernel, for the device-RNG mode whose stream is not matched
#     ("mean energy vs beta, swap acceptance rate")
# ----------------------------------------------------------------------------------------------
def gen_stats():
    """Equilibrium-ish time averages of the energy under the reference's MCMC (NMC/nmc.py:28-91) at four inverse
    temperatures, several independent chains each (the spread over chains is the error bar), on a +-J and on a
    Gaussian-coupling instance (normalised by max|J| like run() does, NMC/nmc.py:474-476: non-integer couplings);
    and per-rung swap acceptance counts of two small NPT.run calls (NPT/npt.py:649-680)."""
    S, BURN, CHAINS = 1200, 200, 12
    for name, (J, h) in (("pmj48", inst_pmj_sparse(48, 4242)), ("gsparse48", inst_gauss_sparse(48, 77))):
        nf = np.max(np.abs(J))
        Jn, hn = J / nf, h / nf
        obj = ref_nmc.NMC(Jn, hn)
        betas = np.array([0.05, 0.8, 2.0, 4.0])
        chain_means = np.zeros((len(betas), CHAINS))
        chain_min = np.zeros((len(betas), CHAINS))
        for bi, beta in enumerate(betas):
            for c in range(CHAINS):
                np.random.seed(100000 + 1000 * bi + c)
                m0 = np.sign(2 * np.random.rand(Jn.shape[0]) - 1)
                M = obj.MCMC(S, m0.copy(), float(beta), Jn, hn)
                E = energies_of(M[:, BURN:], Jn, hn)
                chain_means[bi, c] = E.mean()
                chain_min[bi, c] = E.min()
        save(f"stats_energy_{name}", **csr_parts(Jn), h=hn, betas=betas, num_sweeps=S, burn_in=BURN,
             chain_means=chain_means, chain_min=chain_min, mean=chain_means.mean(axis=1),
             stderr=chain_means.std(axis=1, ddof=1) / np.sqrt(CHAINS))
    J, h = inst_pmj_sparse(32, 21, with_h=True)
    R, nsw, nswap, npairs = 6, 2400, 240, 2
    beta_list = np.linspace(0.3, 2.0, R)
    att = np.zeros(R - 1, dtype=np.int64)
    acc = np.zeros(R - 1, dtype=np.int64)
    for seed in (901, 902):
        obj = ref_npt.NPT(J.copy(), h.copy())
        np.random.seed(seed)
        random.seed(seed)
        with quiet() as buf:
            obj.run(beta_list=beta_list, num_replicas=R, doNMC=[False] * R, num_sweeps_MCMC=nsw, num_sweeps_read=nsw,
                    num_swap_attempts=nswap, num_swapping_pairs=npairs, num_cycles=2, use_hash_table=False, num_cores=1)
        pairs, a = parse_swap_log(buf.getvalue())
        for (i, _), ok in zip(pairs, a):
            att[i - 1] += 1
            acc[i - 1] += int(ok)
    save("stats_swaps_pmj32", **csr_parts(J), h=np.asarray(h).reshape(-1), beta_list=beta_list, num_replicas=R,
         num_sweeps_MCMC=nsw, num_swap_attempts=nswap, num_swapping_pairs=npairs, attempted=att, accepted=acc)


def gen_stats_min():
    """Minimum energies found by the reference's NPT.run (NPT/npt.py:535-700) on a +-J instance under a fixed budget,
    40 independent runs: the `min-energy vs reference` half of BASELINE.json's metric as a distribution."""
    J, h = inst_pmj_sparse(96, 555)
    R, nsw, nswap, npairs, runs = 6, 48, 6, 2, 40
    beta_list = np.linspace(0.3, 2.5, R)
    E_all = np.zeros((runs, R))
    for s in range(runs):
        obj = ref_npt.NPT(J.copy(), h.copy())
        np.random.seed(7000 + s)
        random.seed(7000 + s)
        import matplotlib.pyplot as _plt
        _plt.close("all")
        with quiet():
            _, E = obj.run(beta_list=beta_list, num_replicas=R, doNMC=[False] * R, num_sweeps_MCMC=nsw, num_sweeps_read=nsw,
                           num_swap_attempts=nswap, num_swapping_pairs=npairs, num_cycles=2, use_hash_table=False, num_cores=1)
        E_all[s] = np.asarray(E).reshape(-1)
    save("stats_minenergy_pmj96", **csr_parts(J), h=np.asarray(h).reshape(-1), beta_list=beta_list, num_replicas=R,
         num_sweeps_MCMC=nsw, num_swap_attempts=nswap, num_swapping_pairs=npairs, energies=E_all, min_energy=E_all.min(axis=1))


def gen_stats_min_mixed():
    """The same distribution with NMC replicas in the ladder (NPT/npt.py:622-647 submits NMC_task :479-512 for the doNMC
    slots -- the house configuration runs the coldest replicas as NMC, NPT/examples/general_example.py:64-84): the two
    coldest of six slots do backbone inference + the three NMC phases every swap round, 40 independent runs.  Also kept:
    the backbone (cluster) sizes the reference printed, as a yardstick for the device-side inference."""
    J, h = inst_pmj_sparse(96, 555)
    R, nsw, nswap, npairs, runs = 6, 48, 6, 2, 40
    doNMC = [False, False, False, False, True, True]
    beta_list = np.linspace(0.3, 2.5, R)
    kw = dict(num_cycles=1, full_update_frequency=1, M_skip=1, temp_x=20, global_beta=2.5, lambda_start=3.0, lambda_end=0.05,
              lambda_reduction_factor=0.8, threshold_initial=0.9999, threshold_cutoff=0.97, max_iterations=100)
    E_all = np.zeros((runs, R))
    sizes = []
    for s in range(runs):
        obj = ref_npt.NPT(J.copy(), h.copy())
        np.random.seed(8000 + s)
        random.seed(8000 + s)
        import matplotlib.pyplot as _plt
        _plt.close("all")
        with quiet() as buf:
            _, E = obj.run(beta_list=beta_list, num_replicas=R, doNMC=list(doNMC), num_sweeps_MCMC=nsw, num_sweeps_read=nsw,
                           num_swap_attempts=nswap, num_swapping_pairs=npairs, tolerance=np.finfo(float).eps,
                           use_hash_table=False, num_cores=1, **kw)
        E_all[s] = np.asarray(E).reshape(-1)
        sizes += [int(m) for m in re.findall(r"cluster size = (\d+)", buf.getvalue())]
        print(f"[golden] mixed run {s}: min {E_all[s].min():.1f}, clusters so far {len(sizes)}", file=sys.__stdout__, flush=True)
    save("stats_minenergy_pmj96_mixed", **csr_parts(J), h=np.asarray(h).reshape(-1), beta_list=beta_list, num_replicas=R,
         doNMC=np.array(doNMC, dtype=np.int8), num_sweeps_MCMC=nsw, num_swap_attempts=nswap, num_swapping_pairs=npairs,
         energies=E_all, min_energy=E_all.min(axis=1), cluster_sizes=np.array(sizes, dtype=np.int64),
         **{k: np.float64(v) for k, v in kw.items()})


def gen_stats_nmc():
    """Minimum energies NMC.run (NMC/nmc.py:442-520: anneal, then cycles of backbone inference + three phases with argmin
    hand-offs) reaches on a GAUSSIAN-coupling instance of 256 spins under a short budget, 40 independent runs: the yardstick
    for the single-chain device-RNG path, whose dynamics run on 24-bit fixed-point couplings (no reference-derived
    fixture covered NMC phases under that arithmetic on non-integer couplings)."""
    J, h = inst_gauss_sparse(256, 9)
    kw = dict(num_sweeps_initial=60, num_sweeps_per_NMC_phase=24, num_NMC_cycles=2, full_update_frequency=1, M_skip=1, temp_x=20,
              global_beta=3.0, lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999,
              threshold_cutoff=0.97, max_iterations=100)
    runs = 40
    mins, lasts, sizes = np.zeros(runs), np.zeros(runs), []
    for s in range(runs):
        obj = ref_nmc.NMC(J.copy(), h.copy())
        np.random.seed(9100 + s)
        import matplotlib.pyplot as _plt
        _plt.close("all")
        with quiet() as buf:
            M, E, mn = obj.run(tolerance=np.finfo(float).eps, use_hash_table=False, **kw)
        mins[s], lasts[s] = mn, E[-1]
        sizes += [int(m) for m in re.findall(r"cluster size = (\d+)", buf.getvalue())]
        print(f"[golden] nmc run {s}: min {mn:.4f}", file=sys.__stdout__, flush=True)
    nf = np.max(np.abs(J))
    save("stats_nmc_run_gsparse256", **csr_parts(J), h=np.asarray(h).reshape(-1), norm_factor=nf, min_energy=mins, last_energy=lasts,
         cluster_sizes=np.array(sizes, dtype=np.int64), **{k: np.float64(v) for k, v in kw.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["mcmc", "large", "nmcsub", "lbp", "npt", "icm", "pre", "known", "stats", "statsmin", "statsmix", "statsnmc"]     # ("lbpstat" needs a device dump: on request)
    table = dict(lbpstat=gen_lbp_lambda_stats, statsmix=gen_stats_min_mixed, statsnmc=gen_stats_nmc, mcmc=gen_mcmc, large=gen_mcmc_large, statsmin=gen_stats_min, nmcsub=gen_nmc_subroutine, lbp=gen_lbp_and_run, npt=gen_npt, icm=gen_icm,
                 pre=gen_preprocessor, known=gen_known_answers, stats=gen_stats)
    for w in which:
        table[w]()
