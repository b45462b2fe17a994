"""Batched, device-resident burden-test engine: all elements x all cohorts per launch.

This is the MI355X-first form of the path the reference walks one cohort and one
element at a time:

    accumulate_elements  <- genic_driver_tools.nonc_model / genic_model / tiled_nonc_model
                            (genic_driver_tools.py:300-431, 31-203, 599-690)
    element_stats        <- transfer_tools.transfer_element_model_with_indels, element_expected_muts_nb,
                            element_pvalue_burden_nb(_by_sample), element_pvalue_indel, Fisher
                            (transfer_tools.py:272-302,343-344,473-482,594-615,731-747,1086-1087)
    gather_bins          <- LazyLoadDatasetFromH5.__getitem__ (mut_dataset.py:76-81)
    tiled_nb_test        <- nb_model.apply_nb_to_region arithmetic (nb_model.py:141-178)

Every function takes either torch CUDA tensors (device entry points, enqueued on
torch's current stream, zero copies) or numpy arrays (``*_host`` twins).  PyTorch is
used only for device memory and streams.
"""
import numpy as np

from . import _lib

ES_PLANES = _lib.ES_PLANES
GS_PLANES = _lib.GS_PLANES


def _is_cuda(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


def _t(x, dtype, device):
    _lib._need_torch()
    import torch
    if x is None:
        return None
    t = torch.as_tensor(x, device=device)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


_WS_CACHE = {}


def _workspace(kind, E, C, dev, private=False):
    """Device scratch for one launch sequence (a torch uint8 tensor, 256-byte aligned by the caching allocator).

    The one-shot entry points share a cached buffer per (kind, device, STREAM): launches on one stream are ordered, so they
    may reuse it; two callers overlapping the same operation on two streams get two buffers.  Plan objects
    (`private=True`) own theirs for their lifetime -- their launches may be enqueued on any stream, any number of steps
    ahead."""
    import torch
    need = _lib.workspace_bytes(kind, E, C)
    if need <= 0:
        return None, 0
    if private:
        return torch.empty(need, dtype=torch.uint8, device=dev), need
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (kind, idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _WS_CACHE[key] = ws
    return ws, need


# ---------------------------------------------------------------------------
def element_stats(mu, sigma, pi_sum, pi_indel, obs_snv, obs_samples, obs_indel, cj, cj_indel,
                  mu_indel=None, sigma_indel=None, device=0, out=None, use_workspace=True):
    """Seven result planes for a dense [E, C] problem.

    mu, sigma, pi_sum : f64 [E, C];  pi_indel : f64 [E] or [E, C];  obs_* : i32 [E, C];
    cj, cj_indel : f64 [C].  Returns a dict plane-name -> [E, C] array/tensor (views of one
    [7, E, C] buffer).  CUDA tensors in -> CUDA tensors out.
    """
    if _is_cuda(mu):
        import torch
        dev = mu.device
        f64, i32 = torch.float64, torch.int32
        mu, sigma, pi_sum = (_t(v, f64, dev) for v in (mu, sigma, pi_sum))
        E, C = mu.shape
        pi_indel = _t(pi_indel, f64, dev)
        per_cohort = int(pi_indel.dim() == 2)
        obs_snv, obs_samples, obs_indel = (_t(v, i32, dev) for v in (obs_snv, obs_samples, obs_indel))
        cj, cj_indel = _t(cj, f64, dev).reshape(-1), _t(cj_indel, f64, dev).reshape(-1)
        assert cj.numel() == C and cj_indel.numel() == C
        mu_indel, sigma_indel = _t(mu_indel, f64, dev), _t(sigma_indel, f64, dev)
        if out is None:
            out = torch.empty((len(ES_PLANES), E, C), dtype=f64, device=dev)
        ws, wsb = _workspace("element_stats", E, C, dev) if use_workspace else (None, 0)
        with torch.cuda.device(dev):
            _lib.call("dig_element_stats", _lib.dev_ptr(mu), _lib.dev_ptr(sigma), _lib.dev_ptr(mu_indel),
                      _lib.dev_ptr(sigma_indel), _lib.dev_ptr(pi_sum), _lib.dev_ptr(pi_indel), per_cohort,
                      _lib.dev_ptr(obs_snv), _lib.dev_ptr(obs_samples), _lib.dev_ptr(obs_indel), _lib.dev_ptr(cj),
                      _lib.dev_ptr(cj_indel), _lib.dev_ptr(out), E, C, _lib.dev_ptr(ws), wsb, _lib.stream_ptr())
        return {name: out[i] for i, name in enumerate(ES_PLANES)}
    mu = _lib.as_host(mu, np.float64)
    if mu.ndim == 1:
        mu = mu[:, None]
    E, C = mu.shape
    sigma = _lib.as_host(sigma, np.float64).reshape(E, C)
    pi_sum = _lib.as_host(pi_sum, np.float64).reshape(E, C)
    pi_indel = _lib.as_host(pi_indel, np.float64)
    per_cohort = int(pi_indel.ndim == 2)
    assert pi_indel.shape == ((E, C) if per_cohort else (E,))
    obs = [_lib.as_host(v, np.int32).reshape(E, C) for v in (obs_snv, obs_samples, obs_indel)]
    cj = _lib.as_host(np.broadcast_to(np.asarray(cj, np.float64).reshape(-1), (C,)), np.float64)
    cj_indel = _lib.as_host(np.broadcast_to(np.asarray(cj_indel, np.float64).reshape(-1), (C,)), np.float64)
    mi = None if mu_indel is None else _lib.as_host(mu_indel, np.float64).reshape(E, C)
    si = None if sigma_indel is None else _lib.as_host(sigma_indel, np.float64).reshape(E, C)
    res = np.empty((len(ES_PLANES), E, C), np.float64)
    _lib.call("dig_element_stats_host", _lib.host_ptr(mu), _lib.host_ptr(sigma), _lib.host_ptr(mi), _lib.host_ptr(si),
              _lib.host_ptr(pi_sum), _lib.host_ptr(pi_indel), per_cohort, _lib.host_ptr(obs[0]), _lib.host_ptr(obs[1]),
              _lib.host_ptr(obs[2]), _lib.host_ptr(cj), _lib.host_ptr(cj_indel), _lib.host_ptr(res), E, C, device)
    return {name: res[i] for i, name in enumerate(ES_PLANES)}


# ---------------------------------------------------------------------------
def gene_stats(mu, sigma, pi, pi_indel, obs, n_samp, cj, t_indel=None, mu_indel=None, sigma_indel=None, device=0, out=None):
    """The gene route's statistics block (transfer_tools.py:331-340,394-456,554-583,709-729,860-861) for G genes x C cohorts
    in ONE launch (dig_gene_stats).  mu, sigma f64 [G, C]; pi f64 [G, 6 or 4, C]; pi_indel [G] or [G, C]; obs i32 [G, 5, C]
    (SYN, MIS, NONS, SPL, INDEL); n_samp i32 [G, 6, C]; cj [C]; t_indel [C] or None (no indel block: those planes are NaN).
    Returns dict plane name -> [G, C] (views of one [22, G, C] buffer); CUDA tensors in -> CUDA tensors out."""
    with_indel = int(t_indel is not None)
    if _is_cuda(mu):
        import torch
        dev = mu.device
        f64, i32 = torch.float64, torch.int32
        mu, sigma, pi = _t(mu, f64, dev), _t(sigma, f64, dev), _t(pi, f64, dev)
        G, C = mu.shape
        n_pi = pi.shape[1]
        assert pi.shape == (G, n_pi, C) and n_pi in (4, 6)
        pi_indel = _t(pi_indel, f64, dev)
        per_cohort = int(pi_indel is not None and pi_indel.dim() == 2)
        obs, n_samp = _t(obs, i32, dev), _t(n_samp, i32, dev)
        assert obs.shape == (G, 5, C) and n_samp.shape == (G, 6, C)
        cj, t_indel = _t(cj, f64, dev).reshape(-1), (None if t_indel is None else _t(t_indel, f64, dev).reshape(-1))
        mu_indel, sigma_indel = _t(mu_indel, f64, dev), _t(sigma_indel, f64, dev)
        if out is None:
            out = torch.empty((len(GS_PLANES), G, C), dtype=f64, device=dev)
        p = _lib.dev_ptr
        with torch.cuda.device(dev):
            _lib.call("dig_gene_stats", p(mu), p(sigma), p(mu_indel), p(sigma_indel), p(pi), n_pi, p(pi_indel), per_cohort, p(obs),
                      p(n_samp), p(cj), p(t_indel), with_indel, p(out), G, C, _lib.stream_ptr())
        return {name: out[i] for i, name in enumerate(GS_PLANES)}
    mu = _lib.as_host(mu, np.float64)
    if mu.ndim == 1:
        mu = mu[:, None]
    G, C = mu.shape
    sigma = _lib.as_host(sigma, np.float64).reshape(G, C)
    pi = _lib.as_host(pi, np.float64)
    if pi.ndim == 2:
        pi = pi[:, :, None]
    pi = np.ascontiguousarray(pi)
    n_pi = pi.shape[1]
    assert pi.shape == (G, n_pi, C) and n_pi in (4, 6)
    pi_indel = None if pi_indel is None else _lib.as_host(pi_indel, np.float64)
    per_cohort = int(pi_indel is not None and pi_indel.ndim == 2)
    obs = np.ascontiguousarray(_lib.as_host(obs, np.int32).reshape(G, 5, C))
    n_samp = np.ascontiguousarray(_lib.as_host(n_samp, np.int32).reshape(G, 6, C))
    cj = _lib.as_host(np.broadcast_to(np.asarray(cj, np.float64).reshape(-1), (C,)), np.float64)
    ti = None if t_indel is None else _lib.as_host(np.broadcast_to(np.asarray(t_indel, np.float64).reshape(-1), (C,)), np.float64)
    mi = None if mu_indel is None else _lib.as_host(mu_indel, np.float64).reshape(G, C)
    si = None if sigma_indel is None else _lib.as_host(sigma_indel, np.float64).reshape(G, C)
    res = np.empty((len(GS_PLANES), G, C), np.float64)
    h = _lib.host_ptr
    _lib.call("dig_gene_stats_host", h(mu), h(sigma), h(mi), h(si), h(pi), n_pi, h(pi_indel), per_cohort, h(obs), h(n_samp), h(cj),
              h(ti), with_indel, h(res), G, C, device)
    return {name: res[i] for i, name in enumerate(GS_PLANES)}


def gene_pipeline(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, gene_length, d_pr, obs, n_samp, cj,
                  t_indel=None):
    """genic_model (genic_driver_tools.py:31-203) + the gene statistics block as one call on device tensors
    (dig_gene_pipeline): L i32 [G, 4, 192] (silent, missense, nonsense, splice columns), gene_length i32 [G].
    Returns (accumulate dict with P [G, 4, C], dict of the 22 statistics planes)."""
    import torch
    dev = bin_mu.device
    f64, i32, i64, u8 = torch.float64, torch.int32, torch.int64, torch.uint8
    bin_mu, bin_std = _t(bin_mu, f64, dev), _t(bin_std, f64, dev)
    N, C = bin_mu.shape
    bin_y, bin_flag, bin_ctx = _t(bin_y, i32, dev), _t(bin_flag, u8, dev), _t(bin_ctx, i32, dev)
    ov_ptr, ov_idx, L = _t(ov_ptr, i64, dev), _t(ov_idx, i32, dev), _t(L, i32, dev)
    G = L.shape[0]
    assert L.shape == (G, 4, 192) and ov_ptr.numel() == G + 1
    strand_minus, gene_length, d_pr = _t(strand_minus, u8, dev), _t(gene_length, i32, dev), _t(d_pr, f64, dev)
    obs, n_samp = _t(obs, i32, dev), _t(n_samp, i32, dev)
    assert obs.shape == (G, 5, C) and n_samp.shape == (G, 6, C)
    cj = _t(cj, f64, dev).reshape(-1)
    ti = None if t_indel is None else _t(t_indel, f64, dev).reshape(-1)
    o = alloc_accumulate_outputs(G, C, 4, dev)
    out = torch.empty((len(GS_PLANES), G, C), dtype=f64, device=dev)
    ws, wsb = _workspace("accumulate", G, C, dev)
    p = _lib.dev_ptr
    with torch.cuda.device(dev):
        _lib.call("dig_gene_pipeline", p(bin_mu), p(bin_std), p(bin_y), p(bin_flag), p(bin_ctx), p(ov_ptr), p(ov_idx), p(L),
                  p(strand_minus), p(gene_length), p(d_pr), p(obs), p(n_samp), p(cj), p(ti), int(ti is not None), p(o["MU"]),
                  p(o["SIGMA"]), p(o["R_OBS"]), p(o["FLAG"]), p(o["P"]), p(o["R_SIZE"]), p(o["ELT_SIZE"]), p(o["P_INDEL"]), p(out),
                  N, G, C, p(ws), wsb, _lib.stream_ptr())
    return o, {name: out[i] for i, name in enumerate(GS_PLANES)}


# ---------------------------------------------------------------------------
def scale_suffstats(bin_mu, bin_flag, device=0, out=None):
    """Per-cohort sum of Y_PRED over unflagged bins (transfer_tools.py:148-156): [N, C] -> [C]."""
    if _is_cuda(bin_mu):
        import torch
        dev = bin_mu.device
        bin_mu, bin_flag = _t(bin_mu, torch.float64, dev), _t(bin_flag, torch.uint8, dev)
        N, C = bin_mu.shape
        if out is None:
            out = torch.empty(C, dtype=torch.float64, device=dev)
        ws, wsb = _workspace("suffstats", N, C, dev)
        with torch.cuda.device(dev):
            _lib.call("dig_scale_suffstats", _lib.dev_ptr(bin_mu), _lib.dev_ptr(bin_flag), N, C, _lib.dev_ptr(out),
                      _lib.dev_ptr(ws), wsb, _lib.stream_ptr())
        return out
    bin_mu = _lib.as_host(bin_mu, np.float64)
    if bin_mu.ndim == 1:
        bin_mu = bin_mu[:, None]
    N, C = bin_mu.shape
    bin_flag = _lib.as_host(bin_flag, np.uint8).reshape(N, C)
    res = np.empty(C)
    _lib.call("dig_scale_suffstats_host", _lib.host_ptr(bin_mu), _lib.host_ptr(bin_flag), N, C, _lib.host_ptr(res), device)
    return res


def scale_factors_local(bin_mu, bin_flag, n_snv_obs, n_ind_obs, out=None):
    """Single-shard scale factors (transfer_tools.py:148-156): masked column sums and the two divisions in one pair of
    kernels (dig_scale_factors_local).  Returns (cj, cj_indel, exp_sum) device tensors."""
    import torch
    dev = bin_mu.device
    bin_mu, bin_flag = _t(bin_mu, torch.float64, dev), _t(bin_flag, torch.uint8, dev)
    N, C = bin_mu.shape
    n_snv_obs, n_ind_obs = _t(n_snv_obs, torch.float64, dev), _t(n_ind_obs, torch.float64, dev)
    if out is None:
        out = tuple(torch.empty(C, dtype=torch.float64, device=dev) for _ in range(3))
    ws, wsb = _workspace("suffstats", N, C, dev)
    with torch.cuda.device(dev):
        _lib.call("dig_scale_factors_local", _lib.dev_ptr(bin_mu), _lib.dev_ptr(bin_flag), N, C, _lib.dev_ptr(n_snv_obs),
                  _lib.dev_ptr(n_ind_obs), _lib.dev_ptr(out[2]), _lib.dev_ptr(out[0]), _lib.dev_ptr(out[1]), _lib.dev_ptr(ws),
                  wsb, _lib.stream_ptr())
    return out


def scale_factors_from_parts(parts, out=None):
    """cj, cj_indel from the all-gathered per-shard statistics `parts` f64 [world, 3, C] on the device
    (rank-ordered sums + division in one kernel; transfer_tools.py:153-154)."""
    import torch
    dev = parts.device
    parts = _t(parts, torch.float64, dev)
    world, three, C = parts.shape
    assert three == 3
    if out is None:
        out = (torch.empty(C, dtype=torch.float64, device=dev), torch.empty(C, dtype=torch.float64, device=dev))
    with torch.cuda.device(dev):
        _lib.call("dig_scale_factors", _lib.dev_ptr(parts), world, C, _lib.dev_ptr(out[0]), _lib.dev_ptr(out[1]),
                  _lib.stream_ptr())
    return out


# ---------------------------------------------------------------------------
def ideal_overlaps(elt_chrom, blk_ptr, blk_start, blk_end, window, bin_chrom, bin_start):
    """CSR of overlapped bin rows per element (genic_driver_tools.py:275-283), ascending rows.
    Host-side integer index construction in the C++ runtime."""
    elt_chrom = _lib.as_host(elt_chrom, np.int32)
    blk_ptr = _lib.as_host(blk_ptr, np.int64)
    blk_start = _lib.as_host(blk_start, np.int64)
    blk_end = _lib.as_host(blk_end, np.int64)
    bin_chrom = _lib.as_host(bin_chrom, np.int32)
    bin_start = _lib.as_host(bin_start, np.int64)
    E, N = len(elt_chrom), len(bin_chrom)
    ov_ptr = np.zeros(E + 1, np.int64)
    args = [_lib.host_ptr(elt_chrom), _lib.host_ptr(blk_ptr), _lib.host_ptr(blk_start), _lib.host_ptr(blk_end), E,
            int(window), _lib.host_ptr(bin_chrom), _lib.host_ptr(bin_start), N, _lib.host_ptr(ov_ptr)]
    _lib.call("dig_ideal_overlaps_host", *args, None)
    ov_idx = np.zeros(max(int(ov_ptr[E]), 1), np.int32)
    _lib.call("dig_ideal_overlaps_host", *args, _lib.host_ptr(ov_idx))
    return ov_ptr, ov_idx[:int(ov_ptr[E])]


def alloc_accumulate_outputs(E, C, n_class, dev):
    """Output tensors of accumulate_elements (reusable across calls through `out=`)."""
    import torch
    f64, i32 = torch.float64, torch.int32
    return dict(MU=torch.empty((E, C), dtype=f64, device=dev), SIGMA=torch.empty((E, C), dtype=f64, device=dev),
                R_OBS=torch.empty((E, C), dtype=i32, device=dev), FLAG=torch.empty((E, C), dtype=i32, device=dev),
                P=torch.empty((E, n_class, C), dtype=f64, device=dev), R_SIZE=torch.empty(E, dtype=i32, device=dev),
                ELT_SIZE=torch.empty(E, dtype=i32, device=dev), P_INDEL=torch.empty(E, dtype=f64, device=dev))


def accumulate_elements(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr,
                        gene_length=None, device=0, out=None, use_workspace=True):
    """Per-element accumulation for all cohorts (see include/dig_hip.h: dig_accumulate_elements).

    Returns dict(MU, SIGMA [E,C] f64; R_OBS, FLAG [E,C] i32; P [E,n_class,C] f64;
                 R_SIZE, ELT_SIZE [E] i32; P_INDEL [E] f64)."""
    if _is_cuda(bin_mu):
        import torch
        dev = bin_mu.device
        f64, i32, i64, u8 = torch.float64, torch.int32, torch.int64, torch.uint8
        bin_mu, bin_std = _t(bin_mu, f64, dev), _t(bin_std, f64, dev)
        N, C = bin_mu.shape
        bin_y, bin_flag, bin_ctx = _t(bin_y, i32, dev), _t(bin_flag, u8, dev), _t(bin_ctx, i32, dev)
        ov_ptr, ov_idx = _t(ov_ptr, i64, dev), _t(ov_idx, i32, dev)
        L = _t(L, i32, dev)
        if L.dim() == 2:
            L = L[:, None, :]
        E, n_class, K = L.shape
        assert K == 192 and bin_ctx.shape == (N, 64) and ov_ptr.numel() == E + 1
        strand_minus = _t(strand_minus, u8, dev)
        gene_length = _t(gene_length, i32, dev)
        d_pr = _t(d_pr, f64, dev)
        assert d_pr.shape == (C, 192)
        o = out if out is not None else alloc_accumulate_outputs(E, C, n_class, dev)
        ws, wsb = _workspace("accumulate", E, C, dev) if use_workspace else (None, 0)
        with torch.cuda.device(dev):
            _lib.call("dig_accumulate_elements", _lib.dev_ptr(bin_mu), _lib.dev_ptr(bin_std), _lib.dev_ptr(bin_y),
                      _lib.dev_ptr(bin_flag), _lib.dev_ptr(bin_ctx), _lib.dev_ptr(ov_ptr), _lib.dev_ptr(ov_idx),
                      _lib.dev_ptr(L), n_class, _lib.dev_ptr(strand_minus), _lib.dev_ptr(gene_length), _lib.dev_ptr(d_pr),
                      _lib.dev_ptr(o["MU"]), _lib.dev_ptr(o["SIGMA"]), _lib.dev_ptr(o["R_OBS"]), _lib.dev_ptr(o["FLAG"]),
                      _lib.dev_ptr(o["P"]), _lib.dev_ptr(o["R_SIZE"]), _lib.dev_ptr(o["ELT_SIZE"]),
                      _lib.dev_ptr(o["P_INDEL"]), N, E, C, _lib.dev_ptr(ws), wsb, _lib.stream_ptr())
        return o
    bin_mu = _lib.as_host(bin_mu, np.float64)
    if bin_mu.ndim == 1:
        bin_mu = bin_mu[:, None]
    N, C = bin_mu.shape
    bin_std = _lib.as_host(bin_std, np.float64).reshape(N, C)
    bin_y = _lib.as_host(bin_y, np.int32).reshape(N, C)
    bin_flag = _lib.as_host(bin_flag, np.uint8).reshape(N, C)
    bin_ctx = _lib.as_host(bin_ctx, np.int32).reshape(N, 64)
    ov_ptr, ov_idx = _lib.as_host(ov_ptr, np.int64), _lib.as_host(ov_idx, np.int32)
    L = _lib.as_host(L, np.int32)
    if L.ndim == 2:
        L = L[:, None, :]
    L = np.ascontiguousarray(L)
    E, n_class, K = L.shape
    assert K == 192 and len(ov_ptr) == E + 1
    strand_minus = _lib.as_host(strand_minus, np.uint8)
    gl = None if gene_length is None else _lib.as_host(gene_length, np.int32)
    d_pr = _lib.as_host(d_pr, np.float64).reshape(C, 192)
    o = dict(MU=np.empty((E, C)), SIGMA=np.empty((E, C)), R_OBS=np.empty((E, C), np.int32),
             FLAG=np.empty((E, C), np.int32), P=np.empty((E, n_class, C)), R_SIZE=np.empty(E, np.int32),
             ELT_SIZE=np.empty(E, np.int32), P_INDEL=np.empty(E))
    if len(ov_idx) == 0:
        ov_idx = np.zeros(1, np.int32)
    _lib.call("dig_accumulate_elements_host", _lib.host_ptr(bin_mu), _lib.host_ptr(bin_std), _lib.host_ptr(bin_y),
              _lib.host_ptr(bin_flag), _lib.host_ptr(bin_ctx), _lib.host_ptr(ov_ptr), _lib.host_ptr(ov_idx),
              _lib.host_ptr(L), n_class, _lib.host_ptr(strand_minus), _lib.host_ptr(gl), _lib.host_ptr(d_pr),
              _lib.host_ptr(o["MU"]), _lib.host_ptr(o["SIGMA"]), _lib.host_ptr(o["R_OBS"]), _lib.host_ptr(o["FLAG"]),
              _lib.host_ptr(o["P"]), _lib.host_ptr(o["R_SIZE"]), _lib.host_ptr(o["ELT_SIZE"]), _lib.host_ptr(o["P_INDEL"]),
              N, E, C, device)
    return o


# ---------------------------------------------------------------------------
_NP_DT = {np.dtype(np.float32): _lib.DIG_F32, np.dtype(np.float64): _lib.DIG_F64, np.dtype(np.int16): _lib.DIG_I16}


def element_pipeline(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr, obs_snv,
                     obs_samples, obs_indel, cj, cj_indel, gene_length=None, out_acc=None, out_stats=None, stages=7, compact=False):
    """accumulate_elements (n_class = 1) + element_stats as one operation on device tensors (dig_element_pipeline):
    the rate sums are formed inside the statistics kernel.  Returns (accumulate dict, statistics tensor [7, E, C]).
    stages: bit mask 1 = context kernel, 2 = dot kernel, 4 = statistics (7 = all); separate calls must keep that order.
    compact="auto": check L for the three-fold context repetition first (one more pass over L and a stream synchronisation,
    see PipelinePlan) and run the 64-context form when it holds; a loop over the same element set should keep a PipelinePlan."""
    import torch
    if compact:
        plan = PipelinePlan(bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr, obs_snv, obs_samples,
                            obs_indel, out_acc=out_acc, out_stats=out_stats, gene_length=gene_length, compact=compact,
                            pack_bins=False)               # (one call: packing the tables would cost more than it saves)
        return plan.run(_t(cj, torch.float64, plan.dev), _t(cj_indel, torch.float64, plan.dev), stages=stages)
    dev = bin_mu.device
    f64, i32, i64, u8 = torch.float64, torch.int32, torch.int64, torch.uint8
    bin_mu, bin_std = _t(bin_mu, f64, dev), _t(bin_std, f64, dev)
    N, C = bin_mu.shape
    bin_y, bin_flag, bin_ctx = _t(bin_y, i32, dev), _t(bin_flag, u8, dev), _t(bin_ctx, i32, dev)
    ov_ptr, ov_idx = _t(ov_ptr, i64, dev), _t(ov_idx, i32, dev)
    L = _t(L, i32, dev)
    if L.dim() == 2:
        L = L[:, None, :]
    E, n_class, K = L.shape
    assert n_class == 1 and K == 192 and bin_ctx.shape == (N, 64) and ov_ptr.numel() == E + 1
    strand_minus, gene_length, d_pr = _t(strand_minus, u8, dev), _t(gene_length, i32, dev), _t(d_pr, f64, dev)
    obs = [_t(x, i32, dev) for x in (obs_snv, obs_samples, obs_indel)]
    assert all(x.shape == (E, C) for x in obs)
    cj, cj_indel = _t(cj, f64, dev), _t(cj_indel, f64, dev)
    o = out_acc if out_acc is not None else alloc_accumulate_outputs(E, C, 1, dev)
    st = out_stats if out_stats is not None else torch.empty((len(ES_PLANES), E, C), dtype=f64, device=dev)
    ws, wsb = _workspace("pipeline", E, C, dev)
    if ws is None:
        raise _lib.DigHipError("dig_element_pipeline: problem too large for the fused path (E * C >= 2^32 - 1)")
    p = _lib.dev_ptr
    with torch.cuda.device(dev):
        _lib.call("dig_element_pipeline", p(bin_mu), p(bin_std), p(bin_y), p(bin_flag), p(bin_ctx), p(ov_ptr), p(ov_idx),
                  p(L), p(strand_minus), p(gene_length), p(d_pr), p(obs[0]), p(obs[1]), p(obs[2]), p(cj), p(cj_indel),
                  p(o["MU"]), p(o["SIGMA"]), p(o["R_OBS"]), p(o["FLAG"]), p(o["P"]), p(o["R_SIZE"]), p(o["ELT_SIZE"]),
                  p(o["P_INDEL"]), p(st), N, E, C, None, int(stages), p(ws), wsb, _lib.stream_ptr())
    return o, st


_RECORDS_FASTER = {}          # device index -> whether the record form of the statistics outputs is the faster one there


def records_form_is_faster(plane_plan, record_plan, cj, cj_indel, rounds=2, warm=8, launches=25):
    """Which layout of the statistics stage's outputs is the faster one on THIS card: eleven planes, or tile-blocked records
    (PipelinePlan(records_out=True))?  The pool's MI355X fall into two kinds -- on most the eleven store streams of the plane
    form cost ~ 10 us of a 180-us pass in a loop, on the others the record form does.  Developer A/B
    (`bench.py --outputs auto`): whole passes of the two plans (same inputs), best of `rounds` x `launches` behind `warm` untimed
    ones (bursts of five measure the clocks coming up, not the forms), HIP events on torch's current stream; a tie keeps the
    planes; once per process and device.  Returns (bool, {"planes": us, "records": us})."""
    import torch
    key = plane_plan.dev.index
    if key in _RECORDS_FASTER:
        return _RECORDS_FASTER[key]
    us = {}
    with torch.cuda.device(plane_plan.dev):
        for name, pl in (("planes", plane_plan), ("records", record_plan)):
            best = float("inf")
            for _ in range(rounds):
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _w in range(warm):
                    pl.run(cj, cj_indel)
                ea.record()
                for _r in range(launches):
                    pl.run(cj, cj_indel)
                eb.record()
                eb.synchronize()
                best = min(best, ea.elapsed_time(eb) / launches * 1e3)
            us[name] = best
    _RECORDS_FASTER[key] = (us["records"] < us["planes"] - 1.5, us)
    return _RECORDS_FASTER[key]


class PipelinePlan:
    """dig_element_pipeline / dig_scale_factors_local with the argument marshalling done ONCE, for loops that run
    the same tensors many times (the driver's cohort loop, bench.py): a call is then one ctypes invocation (~10 us of
    host time instead of ~60), which keeps the host ahead of a 0.3 ms GPU step.  The tensors are held by the plan."""

    _ws_owner = {}         # data_ptr of a caller-owned workspace -> weak reference to the plan that owns it

    def __init__(self, bin_mu, bin_std, bin_y, bin_flag, bin_ctx, ov_ptr, ov_idx, L, strand_minus, d_pr, obs_snv,
                 obs_samples, obs_indel, out_acc=None, out_stats=None, gene_length=None, compact="auto", workspace=None,
                 pack_bins=True, records_out=False):
        """compact: "auto" (default) checks ONCE, here, on the device whether L repeats every context count three times
        (sequence_tools.py:560-564: true for every elementModel / tiledModel / quickDriver set) and, if so, runs the
        64-context form of the accumulation (contexts + dot in one kernel, half the matrix work); False forces the
        general 192-substitution form.  `self.compact` tells which one runs.
        pack_bins: rewrite the four bin tables ONCE, here, as the packed records the statistics stage gathers from
        (dig_bin_records_pack: 20 bytes per (bin, cohort) of plan-owned memory, two gathers per bin instead of four, same
        bits).  The plan then reads the records, not the tables: call `repack_bins()` after changing a table in place.
        Another PipelinePlan over the same tables may be given instead of True: its records are shared.
        records_out: the statistics stage writes tile-blocked records (DIG_PIPE_RECORDS: the ten outputs of a pair -- seven
        statistics, MU, SIGMA, R_OBS | FLAG -- as field f of block i / 64 at [f / 2, i % 64, f % 2]: one aligned 5 120-byte run
        per 64-pair tile instead of eleven store streams) into `self.out_records` [ceil(E C / 64), 5, 64, 2]; `self.stats` and
        MU / SIGMA / R_OBS / FLAG of `self.acc` are filled by `unpack()` (same bits).  Needs the packed bin records (pack_bins)."""
        import ctypes
        import torch
        dev = bin_mu.device
        f64, i32, i64, u8 = torch.float64, torch.int32, torch.int64, torch.uint8
        L = _t(L, i32, dev)
        if L.dim() == 2:
            L = L[:, None, :]
        self.keep = [_t(bin_mu, f64, dev), _t(bin_std, f64, dev), _t(bin_y, i32, dev), _t(bin_flag, u8, dev),
                     _t(bin_ctx, i32, dev), _t(ov_ptr, i64, dev), _t(ov_idx, i32, dev), L, _t(strand_minus, u8, dev),
                     _t(gene_length, i32, dev), _t(d_pr, f64, dev), _t(obs_snv, i32, dev), _t(obs_samples, i32, dev),
                     _t(obs_indel, i32, dev)]
        self.N, self.C = self.keep[0].shape
        self.E, n_class, K = L.shape
        assert n_class == 1 and K == 192 and self.keep[4].shape == (self.N, 64) and self.keep[5].numel() == self.E + 1
        assert all(x.shape == (self.E, self.C) for x in self.keep[11:14])
        self.dev = dev
        self.acc = out_acc if out_acc is not None else alloc_accumulate_outputs(self.E, self.C, 1, dev)
        self.stats = out_stats if out_stats is not None else torch.empty((len(ES_PLANES), self.E, self.C), dtype=f64, device=dev)
        if workspace is not None:
            # A caller-owned buffer (uint8, at least dig_element_pipeline_workspace bytes).  NOT plain scratch: a compact plan
            # keeps its [E, 64] copy of L and the repetition flag in it from construction on (dig_element_pipeline_prepare), so
            # the buffer belongs to THIS plan (or to plans over the very same L tensor) for the plan's lifetime -- two live plans
            # over different element sets on one buffer would compute P from each other's L: refused below.
            self.ws, self.wsb = workspace, _lib.workspace_bytes("pipeline", self.E, self.C)
            assert workspace.numel() >= self.wsb and workspace.data_ptr() % 256 == 0
            import weakref
            owner = PipelinePlan._ws_owner.get(workspace.data_ptr())
            if owner is not None and owner[0]() is not None and owner[0]() is not self and owner[1] != (L.data_ptr(), tuple(L.shape)):
                raise ValueError("this workspace buffer already belongs to a live PipelinePlan over ANOTHER element set (a compact "
                                 "plan keeps its copy of L in it): give every plan its own workspace")
            PipelinePlan._ws_owner[workspace.data_ptr()] = (weakref.ref(self), (L.data_ptr(), tuple(L.shape)))
        else:
            self.ws, self.wsb = _workspace("pipeline", self.E, self.C, dev, private=True)
        if self.ws is None:
            raise _lib.DigHipError("dig_element_pipeline: problem too large for the fused path (E * C >= 2^32 - 1)")
        p = _lib.dev_ptr
        o = self.acc
        self._head = [p(x) for x in self.keep]
        self.records = None
        if isinstance(pack_bins, PipelinePlan):              # several plans over the SAME bin tables share one set of records
            assert pack_bins.keep[0].data_ptr() == self.keep[0].data_ptr() and (pack_bins.N, pack_bins.C) == (self.N, self.C)
            self.records = pack_bins.records
        elif pack_bins and self.N >= 1:
            nb = int(_lib.load().dig_bin_records_bytes(self.N, self.C))
            self.records = torch.empty(nb, dtype=torch.uint8, device=dev)
            self.repack_bins()
        self.records_out = bool(records_out)
        self.out_records = None
        if self.records_out:
            if self.records is None:
                raise ValueError("records_out needs the packed bin records (pack_bins)")
            nrec = (self.E * self.C + 63) // 64 * 64
            self.out_records = torch.empty((nrec // 64, _lib.DIG_REC_DOUBLES // 2, 64, 2), dtype=f64, device=dev)
            assert self.out_records.data_ptr() % 256 == 0
        self._tail = [p(o["MU"]), p(o["SIGMA"]), p(o["R_OBS"]), p(o["FLAG"]), p(o["P"]), p(o["R_SIZE"]), p(o["ELT_SIZE"]),
                      p(o["P_INDEL"]), p(self.out_records if self.records_out else self.stats), self.N, self.E, self.C,
                      p(self.records)]
        self._ws = p(self.ws)
        self._fn = getattr(_lib.load(), "dig_element_pipeline")
        self.compact = False
        if compact and self.N >= 1:
            ok = ctypes.c_int(0)
            with torch.cuda.device(dev):
                _lib.call("dig_element_pipeline_prepare", p(L), self.E, self.C, self._ws, self.wsb, ctypes.byref(ok),
                          _lib.stream_ptr())
            self.compact = bool(ok.value)
        self._flags = (_lib.DIG_PIPE_COMPACT_L if self.compact else 0) | (_lib.DIG_PIPE_RECORDS if self.records_out else 0)

    def unpack(self, cohort_major=False, stream=None, stats=None, rates=None):
        """records_out plans: out_records -> the plane form (self.stats [7, E, C] and MU, SIGMA, R_OBS, FLAG of self.acc) on
        `stream`; cohort_major=True writes every plane as [C, E] instead (a result frame's column is then one contiguous row)
        into `stats` [7, C, E] (required then) and, when given, `rates` = dict(MU, SIGMA: float64 [C, E]; R_OBS, FLAG: int32
        [C, E]); self.acc is left alone."""
        import torch
        assert self.records_out
        p = _lib.dev_ptr
        o = self.acc
        with torch.cuda.device(self.dev):
            if cohort_major:
                assert stats is not None and tuple(stats.shape) == (len(ES_PLANES), self.C, self.E)
                r = rates or {}
                assert all(tuple(v.shape) == (self.C, self.E) and v.is_contiguous() for v in r.values())
                assert all(r[k].dtype == (torch.float64 if k in ("MU", "SIGMA") else torch.int32) for k in r)
                _lib.call("dig_element_records_unpack", p(self.out_records), self.E, self.C, p(stats), p(r.get("MU")), p(r.get("SIGMA")),
                          p(r.get("R_OBS")), p(r.get("FLAG")), 1, _lib.stream_ptr(stream))
                return stats
            st = self.stats if stats is None else stats
            _lib.call("dig_element_records_unpack", p(self.out_records), self.E, self.C, p(st), p(o["MU"]), p(o["SIGMA"]),
                      p(o["R_OBS"]), p(o["FLAG"]), 0, _lib.stream_ptr(stream))
        return o, st

    def repack_bins(self):
        """(Re)build the packed bin records from the plan's bin tables on torch's current stream (waits for it)."""
        import torch
        if self.records is None:
            return
        k = self.keep
        with torch.cuda.device(self.dev):
            _lib.call("dig_bin_records_pack", _lib.dev_ptr(k[0]), _lib.dev_ptr(k[1]), _lib.dev_ptr(k[2]), _lib.dev_ptr(k[3]),
                      self.N, self.C, _lib.dev_ptr(self.records), self.records.numel(), _lib.stream_ptr())

    def run(self, cj, cj_indel, stages=7, stream=None):
        """Enqueue the pipeline (or one of its stages) on `stream` (default: torch's current stream)."""
        rc = self._fn(*self._head, _lib.dev_ptr(cj), _lib.dev_ptr(cj_indel), *self._tail, int(stages) | self._flags, self._ws,
                      self.wsb, _lib.stream_ptr(stream))
        if rc != 0:
            raise _lib.DigHipError("dig_element_pipeline failed (%d): %s" % (rc, _lib.last_error()))
        return self.acc, self.stats


class PlanRing:
    """Several PipelinePlans -- one per batch in flight, each with its own outputs and workspace -- taking turns on as
    many streams.  A pass is a chain of dependent kernels (contexts -> dot -> statistics) and each of them leaves part of
    the chip idle while it ramps up and while its last waves finish; with two passes over different batches in flight
    the kernels of one fill those gaps of the other: 188 us per pass against 193 us one after the other on an MI355X at
    the whole-genome x 37-cohort size (tools/overlap_probe.py; 186 against 207 while the pass still had a fourth kernel).  The caller keeps the
    order of the passes on ONE plan (they share a stream); passes on different plans are independent."""

    def __init__(self, plans, streams=None):
        import torch
        self.plans = list(plans)
        assert self.plans, "at least one plan"
        dev = self.plans[0].dev
        self.streams = list(streams) if streams is not None else [torch.cuda.Stream(device=dev) for _ in self.plans]
        assert len(self.streams) == len(self.plans)

    def run(self, k, cj, cj_indel):
        """Enqueue pass k on plan k mod len(plans); returns that plan's (accumulation outputs, statistics planes), valid
        once the plan's stream has been synchronised."""
        j = k % len(self.plans)
        return self.plans[j].run(cj, cj_indel, stream=self.streams[j])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()


class ScaleFactorPlan:
    """dig_scale_factors_local with cached arguments (see PipelinePlan)."""

    def __init__(self, bin_mu, bin_flag, n_snv_obs, n_ind_obs):
        import torch
        dev = bin_mu.device
        self.keep = [_t(bin_mu, torch.float64, dev), _t(bin_flag, torch.uint8, dev), _t(n_snv_obs, torch.float64, dev),
                     _t(n_ind_obs, torch.float64, dev)]
        self.N, self.C = self.keep[0].shape
        self.ws, self.wsb = _workspace("suffstats", self.N, self.C, dev, private=True)
        p = _lib.dev_ptr
        self._args = [p(self.keep[0]), p(self.keep[1]), self.N, self.C, p(self.keep[2]), p(self.keep[3])]
        self._ws = p(self.ws)
        self._fn = getattr(_lib.load(), "dig_scale_factors_local")

    def run(self, out_sum, cj, cj_indel, stream=None):
        rc = self._fn(*self._args, _lib.dev_ptr(out_sum), _lib.dev_ptr(cj), _lib.dev_ptr(cj_indel), self._ws, self.wsb,
                      _lib.stream_ptr(stream))
        if rc != 0:
            raise _lib.DigHipError("dig_scale_factors_local failed (%d): %s" % (rc, _lib.last_error()))

    def run_sharded(self, part, cj, cj_indel, group=None):
        """Bins sharded over ranks: this shard's sums into part[0] (dig_scale_suffstats), all-gather of the [3, C]
        parts over the process group, rank-ordered sum + division (dig_scale_factors).  Runs on torch's CURRENT stream
        (the collective follows it)."""
        import torch
        import torch.distributed as dist
        lib = _lib.load()
        sp = _lib.stream_ptr()
        rc = lib.dig_scale_suffstats(self._args[0], self._args[1], self.N, self.C, _lib.dev_ptr(part[0]), self._ws, self.wsb, sp)
        if rc != 0:
            raise _lib.DigHipError("dig_scale_suffstats failed (%d): %s" % (rc, _lib.last_error()))
        from . import parallel
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        parts = part
        if parallel.collectives_on(group):
            if getattr(self, "_parts", None) is None or self._parts.shape[0] != world:
                self._parts = torch.empty((world, 3, self.C), dtype=torch.float64, device=part.device)
            dist.all_gather_into_tensor(self._parts, part, group=group)
            parts = self._parts
        rc = lib.dig_scale_factors(_lib.dev_ptr(parts), world, self.C, _lib.dev_ptr(cj), _lib.dev_ptr(cj_indel), sp)
        if rc != 0:
            raise _lib.DigHipError("dig_scale_factors failed (%d): %s" % (rc, _lib.last_error()))


def gather_bins(x_data, bin_rows, tracks=None, out_dtype="f32", transpose=False, device=0):
    """x_data[bin_rows, :, tracks] as float32 (or bf16) -- mut_dataset.py:76-81 for a batch.
    transpose=True returns channels-first [B, T_sel, L] (cnn_predictors.py:131)."""
    if _is_cuda(x_data):
        import torch
        dev = x_data.device
        tmap = {torch.float32: _lib.DIG_F32, torch.float64: _lib.DIG_F64, torch.int16: _lib.DIG_I16}
        assert x_data.dtype in tmap and x_data.is_contiguous() and x_data.dim() == 3
        N, L, T = x_data.shape
        rows = _t(bin_rows, torch.int64, dev)
        tr = None if tracks is None else _t(tracks, torch.int32, dev)      # NULL = all tracks (contiguous block copy)
        B, Ts = rows.numel(), T if tr is None else tr.numel()
        odt = torch.float32 if out_dtype == "f32" else torch.bfloat16
        out = torch.empty((B, Ts, L) if transpose else (B, L, Ts), dtype=odt, device=dev)
        with torch.cuda.device(dev):
            _lib.call("dig_gather_bins", _lib.dev_ptr(x_data), tmap[x_data.dtype], N, L, T, _lib.dev_ptr(rows), B,
                      _lib.dev_ptr(tr), Ts, _lib.dev_ptr(out), _lib.DIG_F32 if out_dtype == "f32" else _lib.DIG_BF16,
                      int(transpose), _lib.stream_ptr())
        return out
    x = np.ascontiguousarray(x_data)
    assert x.dtype in _NP_DT and x.ndim == 3
    N, L, T = x.shape
    rows = _lib.as_host(bin_rows, np.int64).ravel()
    tr = None if tracks is None else _lib.as_host(tracks, np.int32).ravel()
    B, Ts = len(rows), T if tr is None else len(tr)
    if out_dtype != "f32":
        raise ValueError("host path returns float32 only")
    out = np.empty((B, Ts, L) if transpose else (B, L, Ts), np.float32)
    _lib.call("dig_gather_bins_host", _lib.host_ptr(x), _NP_DT[x.dtype], N, L, T, _lib.host_ptr(rows), B,
              _lib.host_ptr(tr), Ts, _lib.host_ptr(out), _lib.DIG_F32, int(transpose), device)
    return out


def count_contexts(genome, chroms, starts, ends, minus=None, device=0, on_device=True, form="2bit"):
    """Trinucleotide context counts [R, 64] of regions of a PackedGenome (sequence_tools.py:65-99,527-566).
    on_device=True keeps the genome resident in HBM (uploaded on first use) and returns a device tensor; False goes
    through the host twin (uploads the genome for this call; small genomes / tests).
    form: "2bit" (dig_count_contexts2: the genome at 2 bits per base + the list of non-ACGT runs; what everything uses) or
    "4bit" (dig_count_contexts, the first form; kept as a cross-check) -- the same counts."""
    ci = genome.chrom_index(chroms)
    R = len(ci)
    st, en = _lib.as_host(starts, np.int64).ravel(), _lib.as_host(ends, np.int64).ravel()
    mi = np.zeros(R, np.uint8) if minus is None else _lib.as_host(np.asarray(minus).astype(np.uint8), np.uint8).ravel()
    assert len(st) == len(en) == len(mi) == R and form in ("2bit", "4bit")
    if (st < 0).any() or (en < 0).any():
        raise ValueError("negative region coordinates")
    if on_device:
        import torch
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        t = lambda a: torch.as_tensor(a, device=dev)
        out = torch.empty((R, 64), dtype=torch.int32, device=dev)
        rc, rs, re_, rm = t(ci), t(st), t(en), t(mi)
        p = _lib.dev_ptr
        with torch.cuda.device(dev):
            if form == "2bit":
                w2, ns, ne, bk, off, ln = genome.on_device2(dev)
                _lib.call("dig_count_contexts2", p(w2), w2.numel(), p(ns) if ns.numel() else None, p(ne) if ns.numel() else None,
                          ns.numel(), p(bk) if ns.numel() else None, bk.numel(), p(off), p(ln), len(genome.names), p(rc), p(rs),
                          p(re_), p(rm), R, p(out), _lib.stream_ptr())
            else:
                words, off, ln = genome.on_device(dev)
                _lib.call("dig_count_contexts", p(words), words.numel(), p(off), p(ln), len(genome.names), p(rc), p(rs), p(re_),
                          p(rm), R, p(out), _lib.stream_ptr())
        return out
    out = np.empty((R, 64), np.int32)
    h = _lib.host_ptr
    dv = device if isinstance(device, int) else 0
    if form == "2bit":
        w2, ns, ne, bk = genome.two_bit()
        _lib.call("dig_count_contexts2_host", h(w2), w2.size, h(ns) if ns.size else None, h(ne) if ns.size else None, ns.size,
                  h(bk) if ns.size else None, bk.size, h(genome.offsets), h(genome.lengths), len(genome.names), h(ci), h(st), h(en),
                  h(mi), R, h(out), dv)
    else:
        _lib.call("dig_count_contexts_host", h(genome.words), genome.words.size, h(genome.offsets), h(genome.lengths),
                  len(genome.names), h(ci), h(st), h(en), h(mi), R, h(out), dv)
    return out


def tiled_nb_test(pt, k, mu, sigma, device=0):
    """Per-tile exact NB test (nb_model.py:141-178).  pt f64 [n_bins, n_tiles] or [C, n_bins, n_tiles];
    k i32 [C, n_bins, n_tiles]; mu, sigma f64 [C, n_bins].  Returns (pval, exp) [C, n_bins, n_tiles]."""
    if _is_cuda(k):
        import torch
        dev = k.device
        k = _t(k, torch.int32, dev)
        C, nb, nt = k.shape
        pt = _t(pt, torch.float64, dev)
        mu, sigma = _t(mu, torch.float64, dev).reshape(C, nb), _t(sigma, torch.float64, dev).reshape(C, nb)
        pval = torch.empty((C, nb, nt), dtype=torch.float64, device=dev)
        ex = torch.empty_like(pval)
        with torch.cuda.device(dev):
            _lib.call("dig_tiled_nb_test", _lib.dev_ptr(pt), int(pt.dim() == 3), _lib.dev_ptr(k), _lib.dev_ptr(mu),
                      _lib.dev_ptr(sigma), _lib.dev_ptr(pval), _lib.dev_ptr(ex), C, nb, nt, _lib.stream_ptr())
        return pval, ex
    k = _lib.as_host(k, np.int32)
    C, nb, nt = k.shape
    pt = _lib.as_host(pt, np.float64)
    mu, sigma = _lib.as_host(mu, np.float64).reshape(C, nb), _lib.as_host(sigma, np.float64).reshape(C, nb)
    pval, ex = np.empty((C, nb, nt)), np.empty((C, nb, nt))
    _lib.call("dig_tiled_nb_test_host", _lib.host_ptr(pt), int(pt.ndim == 3), _lib.host_ptr(k), _lib.host_ptr(mu),
              _lib.host_ptr(sigma), _lib.host_ptr(pval), _lib.host_ptr(ex), C, nb, nt, device)
    return pval, ex


# ---------------------------------------------------------------------------
# per-base / tiled route: front half (dig_base_tile_probs + dig_tile_mut_counts) and the whole chain
# ---------------------------------------------------------------------------
TILE_CTX_MAX_POSITIONS = 16384      # positions of a region base_tile_probs_ctx_kernel evaluates (kCtxMaxPos, dig_tiles.hip)


def check_tile_regions(starts, ends, n_up):
    """What every caller of dig_base_tile_probs_ctx with host-side coordinates must check (the device entry point sees
    device pointers only): a region that starts inside (0, n_up) would fetch from a negative position -- the reference's
    pysam fetch fails there too (sequence_tools.py:21-29)."""
    st, en = np.asarray(starts, np.int64), np.asarray(ends, np.int64)
    if st.size and n_up > 1 and int((en - st).max()) > TILE_CTX_MAX_POSITIONS:
        raise ValueError("penta-nucleotide regions may hold at most %d positions" % TILE_CTX_MAX_POSITIONS)
    if st.size and n_up > 1 and ((st > 0) & (st < n_up)).any():
        raise ValueError("a region that starts at 1 would fetch from a negative position (the reference's pysam fetch fails too)")


def base_tile_probs(genome, chroms, starts, ends, s_prob, binsize, n_tiles=None, device=0):
    """Tile probabilities of regions of a PackedGenome for C cohorts at once (sequence_tools.py:292-317 + the tiling of
    nb_model.py:126-186).  s_prob: f64 [C, 64] (trinucleotide contexts, index 16 b0 + 4 b1 + b2) or [C, 1024]
    (penta-nucleotide: the reference's default n_up = n_down = 2; index in itertools.product('ACGT', repeat=5) order).
    Returns device tensors (pt [C, R, n_tiles], first_pos [R], n_valid [R]); n_tiles defaults to what the longest region
    needs."""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    ci = genome.chrom_index(chroms)
    R = len(ci)
    st, en = _lib.as_host(starts, np.int64).ravel(), _lib.as_host(ends, np.int64).ravel()
    if (st < 0).any() or (en < 0).any():
        raise ValueError("negative region coordinates")
    binsize = int(binsize)
    if n_tiles is None:
        n_tiles = int(max(1, -(-int((en - st).max() if R else 1) // binsize)))
    s_prob = _t(s_prob, torch.float64, dev)
    assert s_prob.dim() == 2 and s_prob.shape[1] in (64, 1024), "s_prob must be [C, 64] (trinucleotide) or [C, 1024] (penta-nucleotide)"
    C = s_prob.shape[0]
    n_up = 1 if s_prob.shape[1] == 64 else 2
    check_tile_regions(st, en, n_up)
    words, off, ln = genome.on_device(dev)
    t = lambda a: torch.as_tensor(a, device=dev)
    rc, rs, re_ = t(ci), t(st), t(en)
    pt = torch.empty((C, R, n_tiles), dtype=torch.float64, device=dev)
    first = torch.empty(R, dtype=torch.int64, device=dev)
    nval = torch.empty(R, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("dig_base_tile_probs_ctx", _lib.dev_ptr(words), words.numel(), _lib.dev_ptr(off), _lib.dev_ptr(ln),
                  len(genome.names), _lib.dev_ptr(rc), _lib.dev_ptr(rs), _lib.dev_ptr(re_), R, _lib.dev_ptr(s_prob), C, n_up,
                  binsize, n_tiles, _lib.dev_ptr(pt), _lib.dev_ptr(first), _lib.dev_ptr(nval), _lib.stream_ptr())
    return pt, first, nval


def tile_mut_counts(genome, chroms, starts, ends, first_pos, n_valid, mut_chrom, mut_start, mut_end, mut_cohort, C, binsize,
                    n_tiles):
    """k i32 [C, R, n_tiles]: mutation rows per tile and cohort (nb_model.py:133-136,160-163).  Regions and mutations
    are joined with the interval-join kernels (a tabix fetch returns the rows overlapping the region); a row counts in
    the tile that holds its START.  mut_* are device tensors or host arrays (chromosome labels as in `chroms`)."""
    import torch
    from .data_tools import tabulate_gpu
    dev = first_pos.device
    R = first_pos.numel()
    ci = genome.chrom_index(chroms)
    blocks = tabulate_gpu.ElementBlocks(ci, _lib.as_host(starts, np.int64).ravel(), _lib.as_host(ends, np.int64).ravel(),
                                        np.arange(R), R, dev)
    mc = torch.as_tensor(genome.chrom_index(list(np.asarray(mut_chrom).astype(str))) if not _is_cuda(mut_chrom) else mut_chrom,
                         device=dev).to(torch.int64)
    ms, me = _t(mut_start, torch.int64, dev), _t(mut_end, torch.int64, dev)
    co = _t(mut_cohort, torch.int32, dev)
    pm, pb = tabulate_gpu.overlap_pairs(blocks, mc, ms, me)
    pr = blocks.elt[pb.long()].to(torch.int32).contiguous()          # block row -> region index
    k = torch.empty((int(C), R, int(n_tiles)), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("dig_tile_mut_counts", _lib.dev_ptr(pm), _lib.dev_ptr(pr), pm.numel(), _lib.dev_ptr(ms), _lib.dev_ptr(co),
                  _lib.dev_ptr(first_pos), _lib.dev_ptr(n_valid), int(binsize), int(n_tiles), R, int(C), _lib.dev_ptr(k),
                  _lib.stream_ptr())
    return k


def tiled_nb_model(genome, chroms, starts, ends, s_prob, mu, sigma, mut_chrom, mut_start, mut_end, mut_cohort, binsize=50,
                   device=0):
    """nb_model (nb_model.py:188-234) for C cohorts x R regions in three launches + the interval join:
    base_tile_probs -> tile_mut_counts -> tiled_nb_test.  mu, sigma: [C, R].  Returns a dict of device tensors
    pval, exp, pt [C, R, n_tiles], k i32 [C, R, n_tiles], first_pos [R], n_valid [R]."""
    pt, first, nval = base_tile_probs(genome, chroms, starts, ends, s_prob, binsize, device=device)
    C, R, n_tiles = pt.shape
    k = tile_mut_counts(genome, chroms, starts, ends, first, nval, mut_chrom, mut_start, mut_end, mut_cohort, C, binsize, n_tiles)
    pval, ex = tiled_nb_test(pt, k, mu, sigma)
    return dict(pval=pval, exp=ex, pt=pt, k=k, first_pos=first, n_valid=nval)


# ---------------------------------------------------------------------------
# scale factors in canonical chunks: identical bits for any sharding of the bins (dig_scale_suffstats_chunked)
# ---------------------------------------------------------------------------
class ChunkedScaleFactorPlan:
    """Scale factors of one shard of the bin grid (transfer_tools.py:148-156) with cached arguments.

    bin_mu / bin_flag: this rank's rows (device tensors, halo included); chunk_rows: first row, in THIS table, of every
    canonical chunk the rank owns (+ end) -- parallel.canonical_chunks / parallel.plan_shards give them; n_snv_obs,
    n_ind_obs: this rank's observed counts per cohort (integers).  run():
        own chunk sums (dig_scale_suffstats_chunked) -> all-gather of [chunk sums ; observed counts] over the process group
        -> every rank adds the K chunk sums first to last and divides (dig_scale_factors_chunked).
    Without a process group the all-gather is skipped: same kernels, same bits."""

    def __init__(self, bin_mu, bin_flag, n_snv_obs, n_ind_obs, chunk_rows, n_chunks_total, group=None, world=None, premask=True):
        """premask: keep a plan-time copy of bin_mu with +0.0 in the flagged entries and sum THAT every step (8 instead of 9
        bytes per (bin, cohort) and step, the same additions and bits); call `remask()` after changing a table in place."""
        import torch
        import torch.distributed as dist
        dev = bin_mu.device
        self.mu, self.flag = _t(bin_mu, torch.float64, dev), _t(bin_flag, torch.uint8, dev)
        self.masked = None
        if premask:
            self.masked = torch.empty_like(self.mu)
            self.remask()
        self.C = int(self.mu.shape[1])
        self.chunk_rows = np.ascontiguousarray(chunk_rows, np.int64)
        self.n_own, self.n_total, self.group = len(self.chunk_rows) - 1, int(n_chunks_total), group
        self.dist_world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # (world may be given explicitly to walk the ranks of a plan one after the other on one device:
        #  enqueue_part() per rank, then finish() on the stacked parts -- tests/test_gpu_sharded.py)
        self.world = int(world) if world is not None else self.dist_world
        assert self.n_own * self.world == self.n_total, "every rank owns n_chunks_total / world canonical chunks"
        lib = _lib.load()
        self.wsb = int(lib.dig_scale_suffstats_chunked_workspace(_lib.host_ptr(self.chunk_rows), self.n_own, self.C))
        self.ws = torch.empty(max(self.wsb, 8), dtype=torch.uint8, device=dev)
        # what a rank contributes: its chunk sums [n_own, C] followed by its observed counts [2, C]
        self.part = torch.zeros((self.n_own + 2, self.C), dtype=torch.float64, device=dev)
        self.part[self.n_own] = _t(n_snv_obs, torch.float64, dev)
        self.part[self.n_own + 1] = _t(n_ind_obs, torch.float64, dev)
        # the exchange step: with more than one rank, or when parallel.FORCE_COLLECTIVES sends a world of one through RCCL too
        from . import parallel
        self.exchange = self.world == self.dist_world and parallel.collectives_on(group)
        self.all = torch.empty((self.world, self.n_own + 2, self.C), dtype=torch.float64, device=dev) \
            if (self.world > 1 or self.exchange) else None
        self.sums = torch.empty((self.n_total, self.C), dtype=torch.float64, device=dev)
        self.obs = torch.empty((self.world, 2, self.C), dtype=torch.float64, device=dev)
        self._lib = lib

    def remask(self):
        """(Re)build the masked copy of the rate table on torch's current stream."""
        import torch
        if self.masked is not None:
            torch.where(self.flag != 0, torch.zeros((), dtype=torch.float64, device=self.mu.device), self.mu, out=self.masked)

    def enqueue_part(self, stream=None):
        """This rank's contribution [n_own + 2, C]: chunk sums of its own chunks, then its observed counts."""
        p = _lib.dev_ptr
        mu, flag = (self.masked, None) if self.masked is not None else (self.mu, self.flag)
        rc = self._lib.dig_scale_suffstats_chunked(p(mu), p(flag), self.C, _lib.host_ptr(self.chunk_rows), self.n_own,
                                                   p(self.part), p(self.ws), self.wsb, _lib.stream_ptr(stream))
        if rc != 0:
            raise _lib.DigHipError("dig_scale_suffstats_chunked failed (%d): %s" % (rc, _lib.last_error()))
        return self.part

    def _on(self, stream):
        """Context in which torch's own work (copies, the collective) lands on `stream` -- the kernels of this plan take the
        stream as an argument, torch's operations take the CURRENT stream, and the two must be the same queue."""
        import contextlib
        import torch
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def finish(self, all_parts, cj, cj_indel, out_sum=None, stream=None):
        """all_parts [world, n_own + 2, C] (rank order = chunk order) -> scale factors on this rank."""
        p = _lib.dev_ptr
        if self.world > 1:
            with self._on(stream):
                self.sums.view(self.world, self.n_own, self.C).copy_(all_parts[:, :self.n_own])
                self.obs.copy_(all_parts[:, self.n_own:])
            sums, obs = self.sums, self.obs
        else:
            sums, obs = all_parts[0, :self.n_own], all_parts[0, self.n_own:]
        rc = self._lib.dig_scale_factors_chunked(p(sums), self.n_total, p(obs), self.world, self.C,
                                                 p(out_sum) if out_sum is not None else None, p(cj), p(cj_indel),
                                                 _lib.stream_ptr(stream))
        if rc != 0:
            raise _lib.DigHipError("dig_scale_factors_chunked failed (%d): %s" % (rc, _lib.last_error()))

    def run(self, cj, cj_indel, out_sum=None, stream=None):
        """Enqueue on `stream` (default: torch's current stream).  With a process group the all-gather and the copies that
        unpack it are issued with `stream` made current, so they are ordered with the two kernels around them."""
        import torch.distributed as dist
        assert self.world in (1, self.dist_world), "run() needs the process group the plan was built for"
        part = self.enqueue_part(stream)
        if self.exchange:
            with self._on(stream):
                dist.all_gather_into_tensor(self.all, part, group=self.group)
            self.finish(self.all, cj, cj_indel, out_sum, stream)
        else:
            self.finish(part.unsqueeze(0), cj, cj_indel, out_sum, stream)


class StageTimer:
    """How long did the dot kernel / the statistics kernel of a dig_element_pipeline call run?  (dig_stage_timer_*, include/dig_hip.h:
    the kernel's own begin and end, taken from its dispatch -- nothing is added to the stream.)

        tm = StageTimer(); tm.arm(_lib.DIG_PIPE_STATISTICS); plan.run(...); ms = tm.read_ms(); tm.close()
    """

    def __init__(self):
        import ctypes
        self._h = ctypes.c_void_p()
        _lib.call("dig_stage_timer_create", ctypes.byref(self._h))

    def arm(self, stage):
        _lib.call("dig_stage_timer_arm", self._h, int(stage))

    def read_ms(self):
        import ctypes
        ms = ctypes.c_double()
        _lib.call("dig_stage_timer_read", self._h, ctypes.byref(ms))
        return float(ms.value)

    def selftest(self, stream=None):
        """Time a kernel that does nothing (then read_ms()): what of a reading is dispatch, not kernel."""
        _lib.call("dig_stage_timer_selftest", self._h, _lib.stream_ptr(stream))

    def close(self):
        if self._h:
            _lib.call("dig_stage_timer_destroy", self._h)
            import ctypes
            self._h = ctypes.c_void_p()
