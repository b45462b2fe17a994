"""FP64 numpy oracle for the sparse GP regression (SGPR) of the region model -- TEST INFRASTRUCTURE ONLY.

The reference builds its calibration GP from gpytorch (ExactGP + InducingPointKernel(ScaleKernel(RBFKernel)) +
GaussianLikelihood, DIGDriver/region_model/trainers/gp_trainer.py:28-45), a dependency that is neither in the
reference tree nor version-pinned nor installable here: **parity with gpytorch is unpinned**.  What CAN be pinned is
that digdriver_amd/region_model/trainers/gp_trainer.py evaluates the published model correctly.  This module restates
Titsias' collapsed bound ("Variational learning of inducing variables in sparse Gaussian processes", AISTATS 2009,
eq. 9) and the SGPR predictive equations with plain numpy linear algebra in two INDEPENDENT forms:

  dense      log N(y | c, Qnn + s2 I) - tr(Knn - Qnn) / (2 s2) with the n x n matrices written out (small n only);
  woodbury   the same quantity through m x m matrices (np.linalg.solve / slogdet on Kmm + Kmn Knm / s2), usable at the
             reference's sizes (n = 150 000, m = 400).

The product uses neither: it works with Cholesky factors of Kmm and of I + A A^T on the GPU.  Model definition shared
with the product: k(a, b) = outputscale * exp(-|a - b|^2 / (2 lengthscale^2)), constant mean c, noise s2, and
Kmm + jitter * outputscale * I (jitter = 1e-6) wherever Kmm is inverted.
"""
import numpy as np


def rbf(a, b, lengthscale, outputscale):
    d2 = (a * a).sum(1)[:, None] - 2.0 * a @ b.T + (b * b).sum(1)[None, :]
    return outputscale * np.exp(-0.5 * np.maximum(d2, 0.0) / lengthscale ** 2)


def bound_dense(X, y, Z, lengthscale, outputscale, noise, mean, jitter=1e-6):
    """Titsias' bound with n x n matrices (O(n^3): a few hundred rows at most)."""
    n = len(y)
    Kmm = rbf(Z, Z, lengthscale, outputscale) + jitter * outputscale * np.eye(len(Z))
    Kmn = rbf(Z, X, lengthscale, outputscale)
    Qnn = Kmn.T @ np.linalg.solve(Kmm, Kmn)
    cov = Qnn + noise * np.eye(n)
    r = y - mean
    sign, logdet = np.linalg.slogdet(cov)
    assert sign > 0
    loglik = -0.5 * (n * np.log(2 * np.pi) + logdet + r @ np.linalg.solve(cov, r))
    return loglik - 0.5 * (n * outputscale - np.trace(Qnn)) / noise


def bound_woodbury(X, y, Z, lengthscale, outputscale, noise, mean, jitter=1e-6, jitter_abs=None):
    """The same bound through m x m matrices only.  jitter_abs: the diagonal term as an absolute value (the product treats
    jitter * outputscale as a constant when differentiating; finite differences in outputscale must hold it fixed too)."""
    n, m = len(y), len(Z)
    Kmm = rbf(Z, Z, lengthscale, outputscale) + (jitter * outputscale if jitter_abs is None else jitter_abs) * np.eye(m)
    Kmn = rbf(Z, X, lengthscale, outputscale)
    G = Kmn @ Kmn.T                                    # [m, m]
    M = Kmm + G / noise
    r = y - mean
    v = Kmn @ r
    logdet = n * np.log(noise) + np.linalg.slogdet(M)[1] - np.linalg.slogdet(Kmm)[1]
    quad = (r @ r - v @ np.linalg.solve(noise * Kmm + G, v)) / noise
    trace = n * outputscale - np.trace(np.linalg.solve(Kmm, G))
    return -0.5 * (n * np.log(2 * np.pi) + logdet + quad) - 0.5 * trace / noise


def predict(X, y, Z, Xs, lengthscale, outputscale, noise, mean, jitter=1e-6):
    """SGPR predictive mean and LATENT standard deviation at Xs (Titsias 2009, eq. 6 with the optimal q(u))."""
    m = len(Z)
    Kmm = rbf(Z, Z, lengthscale, outputscale) + jitter * outputscale * np.eye(m)
    Kmn = rbf(Z, X, lengthscale, outputscale)
    Ksm = rbf(Xs, Z, lengthscale, outputscale)
    M = Kmm + Kmn @ Kmn.T / noise
    mu = mean + Ksm @ np.linalg.solve(M, Kmn @ (y - mean)) / noise
    var = outputscale - np.einsum("ij,ji->i", Ksm, np.linalg.solve(Kmm, Ksm.T)) + np.einsum("ij,ji->i", Ksm, np.linalg.solve(M, Ksm.T))
    return mu, np.sqrt(np.maximum(var, 0.0))
