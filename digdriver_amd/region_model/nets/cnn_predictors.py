"""1-D ResNet regressor over (positions x epigenomic tracks) bins -- the region model's CNN.

Mirror of SimpleMultiTaskResNet (DIGDriver/region_model/nets/cnn_predictors.py:77-171): same layer
names, shapes and registration order (so a reference ``state_dict`` loads unchanged and seeded
initialisation reproduces the reference's weights), same ``forward`` contract

    model(x[B, L, T]) -> (outputs: list of C tensors [B], feature_vecs: list of C tensors [B, 16], None)

The dense conv1d / linear layers stay in PyTorch-ROCm (MIOpen / hipBLASLt on the MFMA units), per the
north star.  MI355X-first additions:
  * ``forward_channels_first`` takes the [B, T, L] layout produced directly by dig_gather_bins
    (transpose fused into the gather, no extra pass over the batch);
  * ``fold_batchnorm()`` returns an inference copy with every BatchNorm folded into the preceding
    convolution (eval-mode BN is an affine map), halving the elementwise traffic between convs;
  * the C task heads are evaluated as three batched matmuls instead of 3*C small linears.
"""
import copy

import torch
from torch import nn
from torch.nn import functional as F

# (name, in_channels or None = input tracks, out_channels, kernel, padding, stride)
_CONV_SPEC = [
    ("11", None, 128, 5, 1, 1), ("12", 128, 256, 3, 1, 2),
    ("21", 256, 256, 3, 1, 1), ("22", 256, 256, 3, 1, 1),
    ("3", 256, 512, 3, 1, 2),
    ("41", 512, 512, 3, 1, 1), ("42", 512, 512, 3, 1, 1),
    ("5", 512, 1024, 3, 1, 2),
    ("61", 1024, 1024, 3, 1, 1), ("62", 1024, 1024, 3, 1, 1),
]
_FLAT = 1024 * 13          # cnn_predictors.py:126,160: L = 100 -> 98 -> 49 -> 25 -> 13 positions


class _TapConv(torch.autograd.Function):
    """conv1d on channels-last activations x [B, L, Cin] -> [B, Lout, Cout] as k accumulated GEMMs (one per filter tap, no im2col
    copy: SimpleMultiTaskResNet._conv_gemm describes the row arithmetic), with its OWN backward.  Left to autograd, the same
    forward costs one zero-filled gradient buffer + one strided copy + one add per TAP for the input gradient, and a copy of every
    tap's weight slice per GEMM (a [Cin, Cout] view of the [Cout, Cin, k] parameter has no unit stride): 194 strided copies, 96
    fills and 46 adds per step at batch 128, a quarter of the step's kernel time (tools/nntrainer_prof.py).  Here the weight is
    brought to [k, Cin, Cout] once per call, the input gradient is accumulated by k in-place GEMMs into ONE buffer, the weight
    gradient is k GEMMs into one [k, Cin, Cout] buffer.  Same sums as the autograd form up to the order of the taps' additions."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, pad, stride):
        B, L, C = x.shape
        Lout = (L + 2 * pad - k) // stride + 1
        extra = (-(L + 2 * pad)) % stride                      # make Lp a multiple of the stride
        Lp = L + 2 * pad + extra
        flat = F.pad(x, (0, 0, pad, pad + extra)).view(B * Lp, C)
        w = weight.permute(2, 1, 0).contiguous()               # [k, Cin, Cout]
        rows_out = B * Lp // stride
        m = (B * Lp - (k - 1) + stride - 1) // stride          # output rows that have all k taps inside `flat`
        full = torch.empty((rows_out, w.shape[2]), dtype=x.dtype, device=x.device)
        head = full[:m]
        torch.addmm(bias, flat[0:(m - 1) * stride + 1:stride], w[0], out=head)
        for tap in range(1, k):
            head.addmm_(flat[tap:tap + (m - 1) * stride + 1:stride], w[tap])
        ctx.save_for_backward(flat, w)
        ctx.geom = (B, L, C, Lout, Lp, k, pad, stride, m, rows_out)
        return full.view(B, Lp // stride, -1)[:, :Lout].contiguous()      # (rows m .. of `full` lie behind Lout: never read)

    @staticmethod
    def backward(ctx, dy):
        flat, w = ctx.saved_tensors
        B, L, C, Lout, Lp, k, pad, stride, m, rows_out = ctx.geom
        Cout = w.shape[2]
        dfull = torch.empty((rows_out, Cout), dtype=dy.dtype, device=dy.device)     # (only the rows behind Lout are zero-filled)
        dview = dfull.view(B, Lp // stride, Cout)
        dview[:, :Lout].copy_(dy)
        if Lp // stride > Lout:
            dview[:, Lout:].zero_()
        dhead = dfull[:m]
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dwk = torch.empty_like(w)
            for tap in range(k):
                torch.mm(flat[tap:tap + (m - 1) * stride + 1:stride].t(), dhead, out=dwk[tap])
            dw = dwk.permute(2, 1, 0)
        if ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 1))
        if ctx.needs_input_grad[0]:
            if stride == 1:                                    # tap 0 WRITES its rows (0 .. m - 1); only the k - 1 rows behind them are filled
                dflat = torch.empty_like(flat)
                torch.mm(dhead, w[0].t(), out=dflat[:m])
                dflat[m:].zero_()
                first = 1
            else:
                dflat = torch.zeros_like(flat)
                first = 0
            for tap in range(first, k):
                dflat[tap:tap + (m - 1) * stride + 1:stride].addmm_(dhead, w[tap].t())
            dx = dflat.view(B, Lp, C)[:, pad:pad + L]
        return dx, dw, db, None, None, None


class _HeadRows(torch.autograd.Function):
    """big [C, ...] whose rows are the storage of the C parameters `rows`: forward = big itself; backward hands every parameter ITS
    row of the gradient as a view (set or added to `.grad` here: returned to the engine, each row would be cloned by the
    parameter's gradient accumulator -- the 252 MB the stack's backward used to copy).  For `loss.backward()` -- what the trainers
    do; `torch.autograd.grad` with the head parameters as inputs sees no gradient for them (the engine is handed None)."""

    @staticmethod
    def forward(ctx, big, *rows):
        ctx.rows = rows
        return big.view_as(big)

    @staticmethod
    def backward(ctx, g):
        for p, gi in zip(ctx.rows, g.contiguous().unbind(0)):    # (rows in the parameters' layout: the fused Adam's fast path)
            if p.requires_grad:
                if p.grad is None:
                    p.grad = gi
                else:
                    p.grad += gi
        return (None,) * (1 + len(ctx.rows))


class SimpleMultiTaskResNet(nn.Module):
    def __init__(self, shape, task_num, get_attention_maps=False):
        super().__init__()
        self.get_attention_maps = get_attention_maps
        self.inp_len, self.inp_size, self.task_num = shape[1], shape[2], task_num
        self.hidden_dim, self.fc2_dim, self.fc3_dim = 128, 128, 16
        if get_attention_maps:   # cnn_predictors.py:90-94
            self.att_conv1 = nn.Conv1d(self.inp_size, self.inp_size, kernel_size=5, padding=2, stride=1)
            self.att_conv2 = nn.Conv1d(self.inp_size, self.inp_size, kernel_size=3, padding=1, stride=1)
        for name, cin, cout, k, pad, stride in _CONV_SPEC:
            setattr(self, "conv" + name, nn.Conv1d(cin or self.inp_size, cout, kernel_size=k, padding=pad, stride=stride))
            setattr(self, "bn" + name, nn.BatchNorm1d(cout))
        self.fc1_lst, self.fc2_lst, self.fc3_lst = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in range(task_num):
            self.fc1_lst.append(nn.Linear(_FLAT, self.fc2_dim))
            self.fc2_lst.append(nn.Linear(self.fc2_dim, self.fc3_dim))
            self.fc3_lst.append(nn.Linear(self.fc3_dim, 1))
        self._folded = False
        self._gw = None

    # ---- trunk ------------------------------------------------------------------------------
    def _block(self, x, name):
        x = getattr(self, "conv" + name)(x)
        if not self._folded:
            x = getattr(self, "bn" + name)(x)
        return F.relu(x)

    def trunk(self, x):
        """x: [B, T, L] channels-first -> [B, 13312] (cnn_predictors.py:143-160)."""
        att = None
        if self.get_attention_maps:
            att = F.softmax(F.relu(self.att_conv2(F.relu(self.att_conv1(x)))), dim=2)
            x = x * att
        x = self._block(self._block(x, "11"), "12")
        x = self._block(self._block(x, "21"), "22") + x
        x = self._block(x, "3")
        x = self._block(self._block(x, "41"), "42") + x
        x = self._block(x, "5")
        x = self._block(self._block(x, "61"), "62") + x
        return x.reshape(-1, _FLAT), att

    # ---- heads ------------------------------------------------------------------------------
    _HEAD_PARTS = (("fc1_lst", "weight"), ("fc1_lst", "bias"), ("fc2_lst", "weight"), ("fc2_lst", "bias"), ("fc3_lst", "weight"),
                   ("fc3_lst", "bias"))

    def _stacked_heads(self):
        """The heads' six parameter kinds as six tensors [C, ...] WITHOUT a copy per step: the C parameters of a kind live in one
        buffer (their `.data` are rows of it), `_HeadRows` hands that buffer to autograd and deals the gradient's rows back to the
        parameters as views.  torch.stack copied 252 MB forward (37 heads x [128, 13312]) and its backward + the gradient
        accumulators copied them back; state_dict, load_state_dict and the optimizer see the same C parameters as before.  The
        buffers are rebuilt when something replaced the parameters' storage (.to(), .double(), a fresh copy of the module)."""
        st = self.__dict__.get("_stacked")
        out = []
        for j, (lst, attr) in enumerate(self._HEAD_PARTS):
            ps = [getattr(m, attr) for m in getattr(self, lst)]
            big = st[j] if st is not None else None
            step = ps[0].numel() * ps[0].element_size()
            if (big is None or big.dtype != ps[0].dtype or big.device != ps[0].device
                    or any(p.data_ptr() != big.data_ptr() + i * step for i, p in enumerate(ps))):
                big = torch.stack([p.data for p in ps])
                for i, p in enumerate(ps):
                    p.data = big[i]
                if st is None:
                    st = self.__dict__["_stacked"] = [None] * len(self._HEAD_PARTS)
                st[j] = big
            out.append(_HeadRows.apply(big, *ps))
        return out

    def heads(self, flat):
        """All task heads as batched matmuls: [B, 13312] -> outputs [C, B], features [C, B, 16]."""
        W1, b1, W2, b2, W3, b3 = self._stacked_heads()               # [C, 128, 13312], [C, 128], [C, 16, 128], ...
        C, B = W1.shape[0], flat.shape[0]
        # the first layer of all heads as ONE GEMM [B, 13312] x [13312, C * 128]: its weight gradient comes out as [C * 128, 13312],
        # i.e. in the parameters' own layout (an einsum over (c, o) left it k-major: a 252 MB re-layout per step in front of Adam)
        h1 = F.relu(F.linear(flat, W1.view(C * self.fc2_dim, -1), b1.view(-1))).view(B, C, self.fc2_dim).transpose(0, 1)
        h2 = F.relu(torch.baddbmm(b2[:, None, :], h1, W2.transpose(1, 2)))
        out = torch.baddbmm(b3[:, None, :], h2, W3.transpose(1, 2)).squeeze(-1)
        return out, h2

    def forward_channels_first(self, x):
        flat, att = self.trunk(x)
        out, feats = self.heads(flat)
        outputs = [out[i] for i in range(self.task_num)]
        feature_vecs = [feats[i] for i in range(self.task_num)]
        return outputs, feature_vecs, att

    def forward(self, x):
        """x: [B, L, T] as stored (cnn_predictors.py:130-131 transposes first)."""
        return self.forward_channels_first(x.transpose(1, 2))

    # ---- GEMM formulation, differentiable (training and evaluation epochs) --------------------
    def _block_rows(self, x, name):
        """conv + BatchNorm + ReLU on channels-last activations x [B, L, C] as k accumulated GEMMs (one per filter tap, no
        im2col copy: see _conv_gemm), out of place so that autograd differentiates it: the backward pass is GEMMs too
        (input gradients w.r.t. the strided row slices, weight gradients as flat[tap::s]^T @ dY) instead of MIOpen's fp32
        weight-gradient convolutions, which fall back to slow solvers at these shapes (54.8 ms per batch of 128 bins at
        T = 735, measured; profiles/r05_aux.json).  BatchNorm sees the [B * Lout, C] matrix: per-channel statistics over batch
        and positions, exactly BatchNorm1d on [B, C, Lout]."""
        conv, bn = getattr(self, "conv" + name), getattr(self, "bn" + name)
        y = _TapConv.apply(x, conv.weight, conv.bias, conv.kernel_size[0], conv.padding[0], conv.stride[0])   # [B, Lout, Cout]
        B, Lout, Cout = y.shape
        y = y.view(B * Lout, Cout)
        if not self._folded:
            y = bn(y)
        return F.relu(y).view(B, Lout, Cout)

    def forward_rows_stacked(self, x):
        """x [B, L, T] row-major (what dig_gather_bins produces without a transpose) -> (outputs [C, B], features [C, B, 16]),
        in train or eval mode, differentiable: the trunk in channels-last GEMM form (_block_rows), the heads as batched
        matmuls on the reference's flatten order (channel-major).  The task dimension is kept: a trainer scores all tasks
        with one chain of kernels instead of one per task."""
        assert not self.get_attention_maps, "the attention branch runs through forward() / forward_channels_first()"
        blk = self._block_rows
        x = blk(blk(x, "11"), "12")
        x = blk(blk(x, "21"), "22") + x
        x = blk(x, "3")
        x = blk(blk(x, "41"), "42") + x
        x = blk(x, "5")
        x = blk(blk(x, "61"), "62") + x
        flat = x.transpose(1, 2).reshape(x.shape[0], _FLAT)    # reference flatten index = c * 13 + l
        return self.heads(flat)

    def forward_rows(self, x):
        """forward_rows_stacked with the (outputs, feature_vecs, None) contract of forward()."""
        out, feats = self.forward_rows_stacked(x)
        return [out[i] for i in range(self.task_num)], [feats[i] for i in range(self.task_num)], None

    # ---- GEMM formulation (inference) -------------------------------------------------------
    def _gemm_weights(self):
        """Conv weights as [k*Cin, Cout] matrices for channels-last windows, head weights with the flatten order
        of a channels-last trunk; cached on the (folded) module."""
        ref = self.conv11.weight
        if getattr(self, "_gw", None) is None or self._gw["W1"].dtype != ref.dtype or self._gw["W1"].device != ref.device:
            gw = {}
            for name, cin, cout, k, pad, stride in _CONV_SPEC:
                conv = getattr(self, "conv" + name)
                gw[name] = (conv.weight.permute(2, 1, 0).contiguous(), conv.bias, k, pad, stride)   # [k, Cin, Cout]
            # reference flatten index = c * 13 + l  ->  channels-last flatten index = l * 1024 + c
            W1 = torch.stack([m.weight.view(self.fc2_dim, 1024, 13).permute(0, 2, 1).reshape(self.fc2_dim, _FLAT)
                              for m in self.fc1_lst])                                  # [C, 128, 13312]
            gw["W1"] = W1.reshape(-1, _FLAT).t().contiguous()                            # [13312, C*128]
            gw["b1"] = torch.cat([m.bias for m in self.fc1_lst])
            gw["W2"] = torch.stack([m.weight for m in self.fc2_lst]).transpose(1, 2).contiguous()   # [C, 128, 16]
            gw["b2"] = torch.stack([m.bias for m in self.fc2_lst])
            gw["W3"] = torch.stack([m.weight for m in self.fc3_lst]).transpose(1, 2).contiguous()   # [C, 16, 1]
            gw["b3"] = torch.stack([m.bias for m in self.fc3_lst])
            self._gw = gw
        return self._gw

    @staticmethod
    def _conv_gemm(x, w, b, k, pad, stride):
        """conv1d on channels-last activations x [B, L, C] as k accumulated GEMMs, one per filter tap, WITHOUT an
        im2col copy: the zero-padded batch is viewed as one [B*Lp, C] matrix; the rows `tap, tap+s, tap+2s, ...` of
        that matrix are exactly the tap-th inputs of all output positions of all batch elements (plus a few rows
        that straddle two batch elements, which land in discarded output rows), so

            out_full = bias + sum_tap  flat[tap::s] @ W_tap          (hipBLASLt, MFMA)

        and the valid outputs are out_full.view(B, Lp/s, Cout)[:, :Lout].  w: [k, C, Cout]."""
        B, L, C = x.shape
        Lout = (L + 2 * pad - k) // stride + 1
        extra = (-(L + 2 * pad)) % stride                     # make Lp a multiple of the stride
        xp = F.pad(x, (0, 0, pad, pad + extra))
        Lp = L + 2 * pad + extra
        flat = xp.view(B * Lp, C)
        rows_out = B * Lp // stride
        out_full = torch.empty((rows_out, w.shape[2]), dtype=x.dtype, device=x.device)
        m = (B * Lp - (k - 1) + stride - 1) // stride         # output rows that have all k taps inside `flat`
        head = out_full[:m]
        torch.addmm(b, flat[0:(m - 1) * stride + 1:stride], w[0], out=head)
        for tap in range(1, k):
            head.addmm_(flat[tap:tap + (m - 1) * stride + 1:stride], w[tap])
        if m < rows_out:
            out_full[m:].zero_()
        return torch.relu_(out_full).view(B, Lp // stride, -1)[:, :Lout]

    def forward_gemm(self, x):
        """Inference on x [B, L, T] (the row-major batch dig_gather_bins produces): BN must be folded.  Same
        (outputs, feature_vecs, None) contract as forward()."""
        assert self._folded and not self.get_attention_maps, "forward_gemm needs fold_batchnorm() and no attention branch"
        gw = self._gemm_weights()
        cv = lambda t, n: self._conv_gemm(t, *gw[n])
        x = cv(cv(x, "11"), "12")
        x = cv(cv(x, "21"), "22") + x
        x = cv(x, "3")
        x = cv(cv(x, "41"), "42") + x
        x = cv(x, "5")
        x = cv(cv(x, "61"), "62") + x
        flat = x.reshape(x.shape[0], _FLAT)
        C = self.task_num
        h1 = F.relu(torch.addmm(gw["b1"], flat, gw["W1"])).view(-1, C, self.fc2_dim).transpose(0, 1)   # [C, B, 128]
        h2 = F.relu(torch.baddbmm(gw["b2"][:, None, :], h1, gw["W2"]))                                  # [C, B, 16]
        out = torch.baddbmm(gw["b3"][:, None, :], h2, gw["W3"]).squeeze(-1)                             # [C, B]
        return [out[i] for i in range(C)], [h2[i] for i in range(C)], None

    # ---- inference copy ---------------------------------------------------------------------
    @torch.no_grad()
    def fold_batchnorm(self):
        """Eval-mode copy with BN folded: w' = w * g / sqrt(var + eps), b' = (b - mean) * g / sqrt(var + eps) + beta."""
        m = copy.deepcopy(self).eval()
        for name, *_ in _CONV_SPEC:
            conv, bn = getattr(m, "conv" + name), getattr(m, "bn" + name)
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight.mul_(scale[:, None, None])
            conv.bias.copy_((conv.bias - bn.running_mean) * scale + bn.bias)
            setattr(m, "bn" + name, nn.Identity())
        m._folded = True
        m._gw = None
        return m


def flops_per_bin(n_tracks, task_num, length=100):
    """Multiply-add count of one forward pass (x2 for FLOP): used for the MFMA roofline of the CNN."""
    pos = [98, 49, 49, 49, 25, 25, 25, 13, 13, 13]
    macs = 0
    for (name, cin, cout, k, pad, stride), p in zip(_CONV_SPEC, pos):
        macs += p * (cin or n_tracks) * cout * k
    macs += task_num * (_FLAT * 128 + 128 * 16 + 16)
    return 2 * macs
