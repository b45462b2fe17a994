"""Negative-binomial burden tests -- host mirror of DIGDriver/sequence_model/nb_model.py.

Same names, argument meaning and NaN behaviour as the reference functions; the
arithmetic runs in the HIP kernels of libdig_hip.so (dig_nb.hip / dig_math.hpp).
Inputs may be scalars, numpy arrays, pandas Series (host path: staged through
the *_host entry points) or torch CUDA tensors (device path: no copies, enqueued
on torch's current stream).  There is no CPU fallback.
"""
import numpy as np

from .. import _lib


def _is_cuda_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _series_like(*xs):
    for x in xs:
        if type(x).__name__ == "Series":
            return x
    return None


def _vec3(name, k, alpha, p, device=0):
    if any(_is_cuda_tensor(x) for x in (k, alpha, p)):
        import torch
        dev = next(x.device for x in (k, alpha, p) if _is_cuda_tensor(x))
        k, alpha, p = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (k, alpha, p)])
        k, alpha, p = k.contiguous(), alpha.contiguous(), p.contiguous()
        out = torch.empty_like(k)
        with torch.cuda.device(dev):
            _lib.call(name, _lib.dev_ptr(k), _lib.dev_ptr(alpha), _lib.dev_ptr(p), _lib.dev_ptr(out), k.numel(),
                      _lib.stream_ptr())
        return out
    ser = _series_like(k, alpha, p)
    kb, ab, pb = np.broadcast_arrays(np.asarray(k, dtype=np.float64), np.asarray(alpha, dtype=np.float64),
                                     np.asarray(p, dtype=np.float64))
    shape = kb.shape
    kc, ac, pc = (_lib.as_host(v, np.float64).ravel() for v in (kb, ab, pb))
    out = np.empty(kc.shape, np.float64)
    _lib.call(name + "_host", _lib.host_ptr(kc), _lib.host_ptr(ac), _lib.host_ptr(pc), _lib.host_ptr(out), kc.size,
              device)
    out = out.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(out, index=ser.index)
    return out if shape else float(out)


def normal_params_to_gamma(mu, sigma, device=0):
    """nb_model.py:237-241 -- alpha = mu**2 / sigma**2, theta = sigma**2 / mu."""
    if _is_cuda_tensor(mu) or _is_cuda_tensor(sigma):
        import torch
        dev = mu.device if _is_cuda_tensor(mu) else sigma.device
        mu, sigma = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (mu, sigma)])
        mu, sigma = mu.contiguous(), sigma.contiguous()
        alpha, theta = torch.empty_like(mu), torch.empty_like(mu)
        with torch.cuda.device(dev):
            _lib.call("dig_normal_params_to_gamma", _lib.dev_ptr(mu), _lib.dev_ptr(sigma), _lib.dev_ptr(alpha),
                      _lib.dev_ptr(theta), mu.numel(), _lib.stream_ptr())
        return alpha, theta
    ser = _series_like(mu, sigma)
    mb, sb = np.broadcast_arrays(np.asarray(mu, dtype=np.float64), np.asarray(sigma, dtype=np.float64))
    shape = mb.shape
    mc, sc = _lib.as_host(mb, np.float64).ravel(), _lib.as_host(sb, np.float64).ravel()
    alpha, theta = np.empty(mc.shape), np.empty(mc.shape)
    _lib.call("dig_normal_params_to_gamma_host", _lib.host_ptr(mc), _lib.host_ptr(sc), _lib.host_ptr(alpha),
              _lib.host_ptr(theta), mc.size, device)
    alpha, theta = alpha.reshape(shape), theta.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(alpha, index=ser.index), pd.Series(theta, index=ser.index)
    if not shape:
        return float(alpha), float(theta)
    return alpha, theta


def nb_pvalue_greater_midp(k, alpha, p, device=0):
    """UPPER TAIL NB p-value with mid-p correction (nb_model.py:271-278)."""
    return _vec3("dig_nb_midp_upper", k, alpha, p, device)


def nb_pvalue_greater(k, alpha, p, device=0):
    """UPPER TAIL NB p-value (nb_model.py:243-256); vectorised over the reference's scalar form."""
    return _vec3("dig_nb_greater", k, alpha, p, device)


def nb_pvalue_exact(k, alpha, p, mu=None, device=0):
    """Upper or lower tail depending on k vs the expectation (nb_model.py:298-314).
    `mu` other than None/0 is not supported (no live caller passes it)."""
    if mu:
        raise NotImplementedError("explicit mu is not supported; the reference's callers never pass it")
    return _vec3("dig_nb_exact", k, alpha, p, device)


def nb_pvalue_midp(k, alpha, p, mu=None, device=0):
    """Two-sided-by-side mid-p variant (nb_model.py:316-337)."""
    if mu:
        raise NotImplementedError("explicit mu is not supported; the reference's callers never pass it")
    return _vec3("dig_nb_midp_twosided", k, alpha, p, device)


def fisher_combine(p1, p2, device=0):
    """chi2.sf(-2 (ln p1 + ln p2), df=4) (transfer_tools.py:860-861,1086-1087)."""
    if _is_cuda_tensor(p1) or _is_cuda_tensor(p2):
        import torch
        dev = p1.device if _is_cuda_tensor(p1) else p2.device
        p1, p2 = torch.broadcast_tensors(*[torch.as_tensor(x, dtype=torch.float64, device=dev) for x in (p1, p2)])
        p1, p2 = p1.contiguous(), p2.contiguous()
        out = torch.empty_like(p1)
        with torch.cuda.device(dev):
            _lib.call("dig_fisher", _lib.dev_ptr(p1), _lib.dev_ptr(p2), _lib.dev_ptr(out), p1.numel(), _lib.stream_ptr())
        return out
    ser = _series_like(p1, p2)
    a, b = np.broadcast_arrays(np.asarray(p1, dtype=np.float64), np.asarray(p2, dtype=np.float64))
    shape = a.shape
    ac, bc = _lib.as_host(a, np.float64).ravel(), _lib.as_host(b, np.float64).ravel()
    out = np.empty(ac.shape)
    _lib.call("dig_fisher_host", _lib.host_ptr(ac), _lib.host_ptr(bc), _lib.host_ptr(out), ac.size, device)
    out = out.reshape(shape)
    if ser is not None:
        import pandas as pd
        return pd.Series(out, index=ser.index)
    return out if shape else float(out)
