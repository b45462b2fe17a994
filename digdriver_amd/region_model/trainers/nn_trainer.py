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

    def _rows_form(self):
        """The trunk as GEMMs on the row-major batch (forward_rows: forward AND backward on hipBLASLt, no transpose in the
        gather); a model with the attention branch keeps the convolution path.  On the GPU only: on the CPU -- the golden-epoch
        test -- the convolution path is the reference's own arithmetic; Adam turns last-bit differences of near-zero gradients
        into whole learning-rate steps, so the two paths' weights drift apart by a few 1e-3 over an epoch although every
        forward and backward pass agrees to 1e-6."""
        return (hasattr(self.model, "forward_rows") and not getattr(self.model, "get_attention_maps", False)
                and torch.device(self.device).type == "cuda")

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
        lab_all = torch.stack(self.labels) if stacked else None
        for j in range(n_batches):
            rows = order[j * self.bs:(j + 1) * self.bs][self.rank::self.world]
            if len(rows) == 0:
                continue
            seen.append(rows)
            r = torch.as_tensor(rows, device=self.device)
            if stacked:
                x = self.store.batch(rows, channels_first=False)
                if x.device != self.device:                  # (a store on the host or on another device, as in _forward)
                    x = x.to(self.device)
                Y, FV = self.model.forward_rows_stacked(x.float())
                Tm = lab_all[:, r]
                losses = ((Y - Tm) ** 2).mean(dim=1)
                feats[0].append(FV.detach())
                preds[0].append(Y.detach())
                true[0].append(Tm)
                loss = losses.sum()
                self.optimizer.zero_grad(set_to_none=True)
                loss.backward()
                parallel.average_gradients(list(self.model.parameters()), self.group)
                self.optimizer.step()
                loss_dev += losses.detach().double()
                acc_dev += _predict.r2_rows(Tm, Y.detach())
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


def adam_for(model, device):
    """Adam(lr = 1e-3) as the reference builds it (mutations_main.py:256 / kfold_mutations_main.py:160); on the GPU the fused
    multi-tensor form (one kernel for the 75 M parameters of a 37-task model instead of a chain of foreach kernels: 1.8 ->
    0.5 ms of a 9.8 ms step) -- the model is moved to the device first."""
    from torch import optim
    dev = torch.device(device)
    if dev.type == "cuda":
        model.to(dev)
        return optim.Adam(model.parameters(), lr=1e-3, amsgrad=False, fused=True)
    return optim.Adam(model.parameters(), lr=1e-3, amsgrad=False)
