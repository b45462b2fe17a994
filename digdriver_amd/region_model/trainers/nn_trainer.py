"""CNN training / evaluation epochs over an HBM-resident track matrix.

Mirror of NNTrainer (DIGDriver/region_model/trainers/nn_trainer.py:17-141): Adam + summed per-task MSE, per-batch
squared-Pearson "accuracy", and -- as in the reference -- the features handed to the GP are the activations captured
DURING the training epoch (train-mode BatchNorm, :64), not a clean re-forward.  Differences that are MI355X-first
rather than behavioural: batches come from dig_gather_bins (channels-first, no DataLoader workers, no host copies);
the shuffle is seeded; with a process group every rank trains on its slice of each batch and the gradients are
averaged with one flat all-reduce (parallel.average_gradients) instead of nn.DataParallel.
"""
import numpy as np
import torch

from .. import predict as _predict
from ... import parallel


class NNTrainer:
    def __init__(self, model, optimizer, loss_fn, bs, label_ids, store, train_rows, test_rows, labels, device, seed=0,
                 group=None):
        self.device = device
        self.model = model.to(device)
        self.optimizer, self.loss_fn, self.bs, self.label_ids = optimizer, loss_fn, int(bs), list(label_ids)
        self.store = store
        self.train_rows, self.test_rows = np.asarray(train_rows), np.asarray(test_rows)
        self.labels = [torch.as_tensor(np.asarray(l), dtype=torch.float32, device=device) for l in labels]   # [C][N]
        self.rng = np.random.default_rng(seed)
        self.group = group
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        tune_gemms(self.device)

    def _rows_form(self):
        """The trunk as GEMMs on the row-major batch (forward_rows: forward AND backward on hipBLASLt, no transpose in the
        gather); a model with the attention branch keeps the convolution path.  On the GPU only: on the CPU -- the golden-epoch
        test -- the convolution path is the reference's own arithmetic; Adam turns last-bit differences of near-zero gradients
        into whole learning-rate steps, so the two paths' weights drift apart by a few 1e-3 over an epoch although every
        forward and backward pass agrees to 1e-6."""
        return (hasattr(self.model, "forward_rows") and not getattr(self.model, "get_attention_maps", False)
                and torch.device(self.device).type == "cuda")

    # ---- the training step as a hipGraph -------------------------------------------------------------------------------------
    # A step at batch 128 is ~ 600 kernels of 5.7 ms together, and the interpreter needs 8 - 9 ms to enqueue them (autograd nodes,
    # the tap loops, 75 M parameters in 100 tensors): the GPU idles a third of the step.  Whole batches of the stacked route are
    # therefore captured ONCE -- gather, forward, the 37 losses, backward, fused Adam, the scores -- and replayed with the batch's
    # rows copied into a static index buffer; the first two batches of a trainer (and a last, smaller batch) run eagerly.  One
    # device, no process group (a collective inside a captured step is not attempted), fused + capturable Adam (adam_for).
    # DIG_NN_GRAPH=0 switches it off; a capture that raises leaves the trainer on the eager path.
    def _graph_ready(self, n_rows, j):
        import os
        if getattr(self, "_graph_off", False) or os.environ.get("DIG_NN_GRAPH", "1") == "0":
            return False
        if (self.world > 1 or parallel.collectives_on(self.group) or n_rows != self.bs or torch.device(self.device).type != "cuda"
                or not hasattr(torch.cuda, "CUDAGraph") or not self.optimizer.defaults.get("capturable", False)
                or not self.optimizer.defaults.get("fused", False) or self.store.x.device != torch.device(self.device)
                or self.store.row_offset):
            return False
        self._eager_steps = getattr(self, "_eager_steps", 0)
        if getattr(self, "_graph", None) is None and self._eager_steps < 2:
            self._eager_steps += 1                             # (optimizer state, hipBLASLt's choices and the allocator settle first)
            return False
        return True

    def _step_body(self, r, lab_all):
        """One batch of the stacked route (all tasks with one chain of kernels: the labels as one [C, N] tensor, nn.MSELoss with the
        default mean reduction as one ((Y - T)^2).mean(1)): rows r (device tensor) -> (Y [C, B], features [C, B, 16], T, losses [C]
        float64, scores [C]), all detached."""
        x = self.store.batch(r, channels_first=False)
        if x.device != self.device:                            # (a store on the host or on another device, as in _forward)
            x = x.to(self.device)
        Y, FV = self.model.forward_rows_stacked(x.float())
        Tm = lab_all[:, r]
        losses = ((Y - Tm) ** 2).mean(dim=1)
        loss = losses.sum()
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        parallel.average_gradients(list(self.model.parameters()), self.group)
        self.optimizer.step()
        return Y.detach(), FV.detach(), Tm, losses.detach().double(), _predict.r2_rows(Tm, Y.detach())

    def _graph_step(self, r, lab_all):
        if getattr(self, "_graph", None) is None:
            try:
                self._g_rows = r.clone()
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._g_out = self._step_body(self._g_rows, lab_all)
                self._graph = g
            except Exception as e:                             # the capture recorded nothing that ran: the batch is done eagerly
                self._graph_off, self._graph = True, None
                print("NNTrainer: training step not captured (%s: %s); eager steps" % (type(e).__name__, e))
                torch.cuda.synchronize(self.device)
                return self._step_body(r, lab_all)
        self._g_rows.copy_(r)
        self._graph.replay()
        Y, FV, Tm, losses, acc = self._g_out
        return Y.clone(), FV.clone(), Tm.clone(), losses, acc     # (losses / acc are added to the epoch's sums before the next replay)

    def _forward(self, rows):
        rows_form = self._rows_form()
        x = self.store.batch(rows, channels_first=not rows_form)
        if x.device != self.device:
            x = x.to(self.device)
        return self.model.forward_rows(x.float()) if rows_form else self.model.forward_channels_first(x.float())

    def train(self, epoch, run=0, print_interval=10):
        """nn_trainer.py:40-91 -> (losses [C], accs [C], features [C][n,16], preds [C][n], true [C][n]) over the
        rows this rank saw, in visiting order."""
        self.model.train()
        C = len(self.label_ids)
        order = self.rng.permutation(self.train_rows)
        n_batches = (len(order) + self.bs - 1) // self.bs
        # per-batch losses and squared-Pearson scores stay ON THE DEVICE (float64 sums; the reference's float(loss) and
        # pearsonr per task and batch are 74 host round trips per step at 37 tasks): one copy at the end of the epoch, and
        # one every print interval
        loss_dev = torch.zeros(C, dtype=torch.float64, device=self.device)
        acc_dev = torch.zeros(C, dtype=torch.float64, device=self.device)
        feats, preds, true = [[] for _ in range(C)], [[] for _ in range(C)], [[] for _ in range(C)]
        seen = []
        print('Training epoch {}'.format(epoch))
        # all tasks at once where that changes nothing but the number of launches: the labels as one [C, N] tensor, nn.MSELoss with
        # the default mean reduction as one ((Y - T)^2).mean(1) over the stacked outputs (37 tasks: 37 gathers + 37 losses +
        # their backward nodes per batch otherwise -- most of a 13 ms step at batch 128 was launches)
        stacked = self._rows_form() and isinstance(self.loss_fn, torch.nn.MSELoss) and self.loss_fn.reduction == "mean"
        if stacked and getattr(self, "_lab_all", None) is None:
            self._lab_all = torch.stack(self.labels)
        lab_all = self._lab_all if stacked else None
        for j in range(n_batches):
            rows = order[j * self.bs:(j + 1) * self.bs][self.rank::self.world]
            if len(rows) == 0:
                continue
            seen.append(rows)
            r = torch.as_tensor(rows, device=self.device)
            if stacked:
                # (the step in a function of its own: nothing of its autograd graph survives it -- a captured step must not meet
                # gradient accumulators of an earlier, eager step that live on another stream)
                step = self._graph_step if self._graph_ready(len(rows), j) else self._step_body
                Y, FV, Tm, losses, acc = step(r, lab_all)
                feats[0].append(FV)
                preds[0].append(Y)
                true[0].append(Tm)
                loss_dev += losses
                acc_dev += acc
                if n_batches >= 10 and j % max(1, int(n_batches * print_interval / 100)) == 0 and j > 0:
                    print('Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {}'.format(epoch, j, n_batches, 100. * j / n_batches,
                                                                              loss_dev.cpu().numpy() / (j + 1)))
                continue
            y_lst, fv_lst, _ = self._forward(rows)
            losses = []
            for i in range(C):
                t = self.labels[i][r]
                losses.append(self.loss_fn(y_lst[i], t))
                feats[i].append(fv_lst[i].detach())
                preds[i].append(y_lst[i].detach())
                true[i].append(t)
            loss = torch.sum(torch.stack(losses))
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            parallel.average_gradients(list(self.model.parameters()), self.group)
            self.optimizer.step()
            loss_dev += torch.stack(losses).detach().double()
            acc_dev += _predict.r2_rows(torch.stack([true[i][-1] for i in range(C)]), torch.stack([preds[i][-1] for i in range(C)]))
            if n_batches >= 10 and j % max(1, int(n_batches * print_interval / 100)) == 0 and j > 0:
                print('Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {}'.format(epoch, j, n_batches, 100. * j / n_batches,
                                                                          loss_dev.cpu().numpy() / (j + 1)))
        loss_sums, acc_sums = loss_dev.cpu().numpy(), acc_dev.cpu().numpy()
        if stacked and feats[0]:
            fv_all, y_all, t_all = torch.cat(feats[0], dim=1), torch.cat(preds[0], dim=1), torch.cat(true[0], dim=1)
            feats, preds, true = [[fv_all[i]] for i in range(C)], [[y_all[i]] for i in range(C)], [[t_all[i]] for i in range(C)]
        # With a process group every rank saw batch[rank::world]: put the per-row outputs of all ranks back into the global
        # visiting order (the GP is fitted on the activations of ALL training rows, kfold_mutations_main.py:177), take
        # rank 0's BatchNorm running statistics (what nn.DataParallel keeps) and average the per-batch scores.
        n_rows = len(order)
        if self.world > 1 or parallel.collectives_on(self.group):
            whole = lambda v, width: parallel.gather_visiting_order(
                torch.cat(v) if v else torch.zeros((0,) + width, dtype=torch.float32, device=self.device), n_rows, self.bs, self.group)
            feats = [[whole(f, (16,))] for f in feats]
            preds = [[whole(q, ())] for q in preds]
            true = [[whole(t, ())] for t in true]
            parallel.broadcast_module_buffers(self.model, 0, self.group)
            sc = torch.as_tensor(np.stack([loss_sums, acc_sums]), device=self.device)
            sc = parallel.rank_ordered_sum(sc, self.group) / self.world
            loss_sums, acc_sums = sc[0].cpu().numpy(), sc[1].cpu().numpy()
            self.last_train_rows = order
        else:
            self.last_train_rows = np.concatenate(seen) if seen else np.zeros(0, np.int64)   # visiting order of the features
        cat = lambda lst: [(torch.cat(v) if v else torch.zeros(0)).cpu().numpy() for v in lst]
        losses, accs = loss_sums / n_batches, acc_sums / n_batches
        print('====> Epoch: {}, Average loss: {}, Average accuracy: {}'.format(epoch, losses, accs))
        return losses, accs, cat(feats), cat(preds), cat(true)

    @torch.no_grad()
    def test(self, epoch, run=0):
        """nn_trainer.py:93-141 on the validation rows: eval mode, (losses, accs, features, preds, true, None)."""
        self.model.eval()
        C = len(self.label_ids)
        n_batches = max(1, (len(self.test_rows) + self.bs - 1) // self.bs)
        loss_sums, acc_sums = np.zeros(C), np.zeros(C)
        feats, preds, true = [[] for _ in range(C)], [[] for _ in range(C)], [[] for _ in range(C)]
        stacked = self._rows_form() and isinstance(self.loss_fn, torch.nn.MSELoss) and self.loss_fn.reduction == "mean"
        if stacked:
            # all tasks with one chain of kernels and the scores on the device, as in train(): the reference's float(loss) and r2_score
            # per task and batch are 74 host round trips + 74 small copies per batch at 37 tasks -- three times the batch's forward pass
            if getattr(self, "_lab_all", None) is None:
                self._lab_all = torch.stack(self.labels)
            loss_dev = torch.zeros(C, dtype=torch.float64, device=self.device)
            acc_dev = torch.zeros(C, dtype=torch.float64, device=self.device)
            for j in range(n_batches):
                rows = self.test_rows[j * self.bs:(j + 1) * self.bs]
                if len(rows) == 0:
                    continue
                r = torch.as_tensor(rows, device=self.device)
                x = self.store.batch(r if self.store.x.device == torch.device(self.device) and not self.store.row_offset else rows,
                                     channels_first=False)
                if x.device != self.device:
                    x = x.to(self.device)
                Y, FV = self.model.forward_rows_stacked(x.float())
                Tm = self._lab_all[:, r]
                loss_dev += ((Y - Tm) ** 2).mean(dim=1).double()
                acc_dev += _predict.r2_rows(Tm, Y)
                feats[0].append(FV)
                preds[0].append(Y)
                true[0].append(Tm)
            losses, accs = loss_dev.cpu().numpy() / n_batches, acc_dev.cpu().numpy() / n_batches
            if feats[0]:
                fv_all, y_all, t_all = torch.cat(feats[0], dim=1).cpu().numpy(), torch.cat(preds[0], dim=1).cpu().numpy(), torch.cat(true[0], dim=1).cpu().numpy()
                out_f, out_p, out_t = [fv_all[i] for i in range(C)], [y_all[i] for i in range(C)], [t_all[i] for i in range(C)]
            else:
                out_f, out_p, out_t = [np.zeros((0, 16), np.float32)] * C, [np.zeros(0, np.float32)] * C, [np.zeros(0, np.float32)] * C
            print('====> Test set loss: {}, accuracy: {}'.format(losses, accs))
            return losses, accs, out_f, out_p, out_t, None
        for j in range(n_batches):
            rows = self.test_rows[j * self.bs:(j + 1) * self.bs]
            if len(rows) == 0:
                continue
            y_lst, fv_lst, _ = self._forward(rows)
            r = torch.as_tensor(rows, device=self.device)
            for i in range(C):
                t = self.labels[i][r]
                loss_sums[i] += float(self.loss_fn(y_lst[i], t))
                acc_sums[i] += _predict.r2_score(t.cpu().numpy(), y_lst[i].cpu().numpy())
                feats[i].append(fv_lst[i])
                preds[i].append(y_lst[i])
                true[i].append(t)
        cat = lambda lst: [torch.cat(v).cpu().numpy() for v in lst]
        losses, accs = loss_sums / n_batches, acc_sums / n_batches
        print('====> Test set loss: {}, accuracy: {}'.format(losses, accs))
        return losses, accs, cat(feats), cat(preds), cat(true), None


tune_gemms = _predict.tune_gemms          # (one switch for the trainer's and the predictor's GEMMs)


def adam_for(model, device):
    """Adam(lr = 1e-3) as the reference builds it (mutations_main.py:256 / kfold_mutations_main.py:160); on the GPU the fused
    multi-tensor form (one kernel for the 75 M parameters of a 37-task model instead of a chain of foreach kernels: 1.8 ->
    0.5 ms of a 9.8 ms step) -- the model is moved to the device first."""
    from torch import optim
    dev = torch.device(device)
    if dev.type == "cuda":
        model.to(dev)
        return optim.Adam(model.parameters(), lr=1e-3, amsgrad=False, fused=True, capturable=True)   # (capturable: the step count lives on the device)
    return optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
