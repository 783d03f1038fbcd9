"""Drop-in for the reference's `NPT/apt_preprocessor.py`: class APT_preprocessor(J, h) -- adaptive inverse-temperature
ladder beta_{i+1} = beta_i + alpha / <std_E> from `num_rng` independent chains per rung (NPT/apt_preprocessor.py:12-204).

All `num_rng` chains of a rung advance in ONE batched launch of the HIP engine (the reference submits one pool task
per chain); `num_cores` and the hash table are accepted and ignored.  Like the reference, run() writes
Results/data/Energy_iter_k.npy, sigma_iter_k.npy, beta_list_python.npy and sigma_list_python.npy into the working
directory (its on-disk output format, read by NPT/npt.py:722-725); the PNG is opt-in (plot=True).

rng="numpy" (default) consumes the legacy NumPy stream in the reference's in-order program order (chain j: start state,
then its sweeps), reproducing the reference's beta / sigma lists under `np.random.seed`.
"""
import os

import numpy as np

from . import hostlogic
from .base import SweepMixin
from .engine import Engine


class APT_preprocessor(SweepMixin):
    def __init__(self, J, h, rng=None, seed=None, device=0):
        self.J = J
        if isinstance(h, list):
            h = np.array(h)
        if len(h.shape) == 1:
            h = h[:, np.newaxis]
        self.h = h
        self.N = J.shape[0]
        self._init_backend(rng, seed, device)

    def _hflat(self):
        return np.asarray(self.h, dtype=np.float64).reshape(-1)

    def MCMC(self, num_sweeps, m_start, beta, hash_table=None, use_hash_table=False):
        """NPT/apt_preprocessor.py:33-74."""
        M = np.zeros((self.N, num_sweeps))
        if num_sweeps == 0:
            return M
        self._check_hash_table(hash_table, use_hash_table)
        eng = self._cache.engine(self.J, self._hflat(), 1)
        o = self._mcmc_on(eng, num_sweeps, m_start, np.full(num_sweeps, float(beta)))
        M[:, :] = o["spins"][0].T
        return M

    def MCMC_task(self, m_start, beta, num_sweeps_MCMC, num_sweeps_read, use_hash_table=0):
        """NPT/apt_preprocessor.py:76-113: returns (Energy over the last num_sweeps_read sweeps, final state [1,N])."""
        if num_sweeps_MCMC < 0:
            raise ValueError("negative dimensions are not allowed")
        eng = self._cache.engine(self.J, self._hflat(), 1)
        o = self._mcmc_on(eng, num_sweeps_MCMC, m_start, np.full(num_sweeps_MCMC, float(beta)))
        E = o["energy"][0][-num_sweeps_read:] if num_sweeps_read else np.zeros(0)
        return E, o["spins"][0][-1].astype(np.float64).reshape(1, -1)

    def run(self, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_rng=100, beta_start=0.5, alpha=1.25, sigma_E_val=1000,
            beta_max=30, use_hash_table=1, num_cores=8, plot=False):
        """NPT/apt_preprocessor.py:115-204.  Returns (beta list, sigma list)."""
        from copy import deepcopy
        foldername = 'data'
        os.makedirs(os.path.join('Results', foldername), exist_ok=True)
        import scipy.sparse as sp
        norm_factor = np.max(np.abs(self.J)) if not sp.issparse(self.J) else abs(self.J).max()
        self.J = self.J / norm_factor
        self.h = self.h / norm_factor
        if self.h.shape[0] == 1:
            self.h = self.h.T
        if num_sweeps_MCMC < 0:
            raise ValueError("negative dimensions are not allowed")      # np.zeros((N, -k)) inside the reference's task
        inst = self._cache.instance(self.J, self._hflat())
        N, R, S = inst.n, int(num_rng), int(num_sweeps_MCMC)
        numpy_mode = self.rng == "numpy"
        host_rng = None if numpy_mode else np.random.default_rng(self.seed)
        beta = [deepcopy(beta_start)]
        it = 1
        sigma_E = sigma_E_val
        sigma_E_min = 0.5 * np.min(np.abs(inst.data))                     # 0.5 * min |J_ij| over the non-zeros (:139)
        sigma = []
        saved_state = np.zeros((R, N))
        eng = Engine(inst, None, R, device=self._cache.device)
        try:
            while sigma_E > sigma_E_min:
                if it != 1:
                    beta.append(beta[-1] + alpha / sigma_E)
                start = np.empty((R, N), dtype=np.int8)
                perm = np.empty((R, S, N), dtype=np.int32) if numpy_mode else None
                u = np.empty((R, S, N)) if numpy_mode else None
                for j in range(R):                                        # program order of the in-order pool (:160-171)
                    if it == 1:
                        r = np.random.rand(N, 1) if numpy_mode else host_rng.random((N, 1))
                        start[j] = np.sign(2. * r - 1).reshape(-1)
                    else:
                        start[j] = saved_state[j, :]
                    if numpy_mode:
                        perm[j], u[j] = hostlogic.draw_legacy_stream(S, N)
                eng.set_spins(start)
                if numpy_mode:
                    o = eng.sweep_stream(perm, u, float(beta[-1]), want_energy=True)
                else:
                    # fused windows with the per-sweep energy trace where the instance qualifies (same bits as sweep by sweep)
                    o = eng.sweep_philox_windows(S, self.seed, sweep0=self._sweep_counter, beta=float(beta[-1]), want_energy=True)
                    self._sweep_counter += S
                Energy = o["energy"][:, S - num_sweeps_read:] if num_sweeps_read else np.zeros((R, 0))
                saved_state[:, :] = eng.get_spins()
                sigma_E = np.mean(np.std(Energy, axis=1))
                print(f'\ncurrent iteration = {it}, β = {beta[-1]:.3f}, and average σ = {sigma_E:.3f}\n')
                if beta[-1] > beta_max:
                    print('Did not converge but hit the max beta limit\n')
                    break
                sigma.append(sigma_E)
                np.save(os.path.join('Results', foldername, f'Energy_iter_{it}.npy'), Energy)
                np.save(os.path.join('Results', foldername, f'sigma_iter_{it}.npy'), sigma_E)
                it += 1
        finally:
            eng.close()
        np.save('beta_list_python.npy', beta)
        np.save('sigma_list_python.npy', sigma)
        if plot:
            self.plot_results(beta, sigma)
        return beta, sigma

    def plot_results(self, beta, sigma):
        """NPT/apt_preprocessor.py:206-231 (presentation only)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax1 = plt.subplots()
        ax1.plot(beta[:len(sigma)], sigma, 'o-')
        ax1.set_xlabel('beta')
        ax1.set_ylabel('sigma_E')
        fig.savefig('beta_sigma.png')
        plt.close(fig)
