"""Agent counterpart: the training loop that drives dpLGAR the way the reference's DifferentiableLGAR agent does
(/root/reference/dpLGAR/agents/DifferentiableLGAR.py:20-172): Adam over alpha/n/Ksat, MSE on runoff after a warm-up
slice plus a range-bound penalty (models/functions/loss.py:10-36), NSE log, per-epoch state and mass-balance reset.

MI355X-first difference: an epoch is ONE forward launch over the whole forcing series (model(x[T, N, 2])) and one batch
of tangent launches for the backward pass, instead of T Python-level forward() calls with a 10 ms sleep each
(DifferentiableLGAR.py:117-125).  `stepwise=True` reproduces the reference's row-by-row call sequence exactly.
"""
import numpy as np
import torch
import torch.nn as nn

from . import data as D
from . import distributed as dist_
from .model import MassBalance, dpLGAR
from .workloads import shard_bounds


class RangeBoundLoss(nn.Module):
    """Penalty that keeps parameters inside [lb, ub] (models/functions/loss.py:10-36): for each ParameterList,
    sum(relu(p - ub)) + mean(relu(lb - p)); the last entry (ponded_depth_max) is a plain tensor."""

    def __init__(self, lb, ub, factor=1.0):
        super().__init__()
        self.lb = torch.tensor([float(v) for v in lb], dtype=torch.float64)
        self.ub = torch.tensor([float(v) for v in ub], dtype=torch.float64)
        self.factor = torch.tensor(float(factor), dtype=torch.float64)

    def forward(self, params):
        loss = torch.tensor(0.0, dtype=torch.float64)
        for i in range(len(params) - 1):
            # every element is bounded (per-column ensembles: [N] per layer), as the reference bounds each of its scalars
            t = torch.stack([p.reshape(-1) for p in params[i]]).cpu()  # [L, N] (N = 1 for the single column)
            loss = loss + torch.sum(self.factor * torch.relu(t - self.ub[i]).mean(1)) + torch.mean(self.factor * torch.relu(self.lb[i] - t))
        t = params[-1].cpu()
        return loss + self.factor * torch.relu(t - self.ub[-1]) + self.factor * torch.relu(self.lb[-1] - t)


class DifferentiableLGAR:
    """The training loop.  With `torch.distributed` initialised (one process per GPU, RCCL; gloo in the CPU tests) the basin's
    columns are SHARDED: rank r owns the contiguous columns workloads.shard_bounds(N, world, r) -- parameters shared, forcing
    and state sharded, like every other job of this engine (distributed.py) -- and an epoch exchanges two things: the
    per-timestep runoff sum [T] (all-reduce, so that every rank holds the basin-mean series the loss is taken on) and, before
    optimizer.step(), the L x 3 parameter gradients (distributed.reduce_parameter_gradients: SURVEY.md section 8e).  Every
    rank then takes the same Adam step on the same numbers: the parameters stay bit-equal across ranks.

    forcing_scale: optional [N] per-column multiplier of the precipitation (a basin with uneven rainfall; what makes the
    columns of a shared-parameter basin differ from one another)."""

    def __init__(self, cfg, observations=None, stepwise=False, log=print, forcing_scale=None, group=None):
        self.cfg = cfg
        self.group = group
        self.rank, self.world = dist_.world_info()
        self.sharded = dist_._collective_on(group)
        self.log = log if self.rank == 0 else (lambda s: None)
        self.stepwise = stepwise
        torch.manual_seed(0)
        self.data = D.Data(cfg)
        if observations is not None:
            self.data.y = torch.as_tensor(observations, dtype=torch.float64)
        self.n_total = int(cfg.get("n_columns", 1) or 1)
        self.lo, self.hi = shard_bounds(self.n_total, self.world, self.rank) if self.sharded else (0, self.n_total)
        if self.hi - self.lo < 1:
            raise ValueError("rank %d of %d has no column of the %d-column basin" % (self.rank, self.world, self.n_total))
        self.forcing_scale = None
        if forcing_scale is not None:
            fs = torch.as_tensor(np.asarray(forcing_scale), dtype=torch.float64)
            if tuple(fs.shape) != (self.n_total,):
                raise ValueError("forcing_scale must be [n_columns]")
            self.forcing_scale = fs[self.lo:self.hi]
        self.model = dpLGAR(cfg, n_columns=self.hi - self.lo)
        self.mass_balance = MassBalance(cfg, self.model)
        self.criterion = nn.MSELoss()
        hp = cfg.models.hyperparameters
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=hp.learning_rate)
        self.range_bound_loss = RangeBoundLoss(hp.lb, hp.ub, factor=1.0)
        self.y_hat = self.y_t = None
        self.current_epoch = 0
        self.history = []

    def run(self):
        try:
            self.train()
        except KeyboardInterrupt:
            self.log("You have entered CTRL+C.. Wait to finalize")

    def train(self):
        self.model.train()
        for _ in range(int(self.cfg.models.hyperparameters.epochs)):
            self.train_one_epoch()
            self.current_epoch += 1
            self.model.set_internal_states()           # DifferentiableLGAR.py:105
            self.mass_balance.reset_mass(self.model)   # :107

    def _forcing(self, x):
        """x [T, 2] or [2] -> this rank's [T, n_local, 2] / [n_local, 2] block (precipitation scaled per column)."""
        n = self.model.n_columns
        xb = x[..., None, :].expand(*x.shape[:-1], n, 2)
        if self.forcing_scale is not None:
            xb = torch.stack([xb[..., 0] * self.forcing_scale, xb[..., 1]], dim=-1)
        return xb

    def _basin_mean(self, local_sum):
        """[T] sum of this rank's columns (graph-connected) -> the basin-mean series every rank takes the loss on.  The other
        ranks' part enters as a constant: a rank's backward pass yields ITS columns' share of the gradient, and the shares are
        summed by reduce_parameter_gradients."""
        others = dist_.all_reduce_sum(local_sum.detach().clone(), self.group) - local_sum.detach()
        return (local_sum + others) / float(self.n_total)

    def train_one_epoch(self):
        self.optimizer.zero_grad()
        x, y = self.data.x, self.data.y
        N = self.model.n_columns
        if self.stepwise:
            rows = []
            for i in range(len(self.data)):
                runoff, _ = self.model(self._forcing(x[i]) if (N > 1 or self.forcing_scale is not None) else x[i])
                if self.sharded:
                    rows.append(runoff.reshape(-1).sum())
                else:
                    rows.append(runoff.reshape(-1)[0] if N > 1 else runoff)
                self.mass_balance.change_mass(self.model)
            y_hat = torch.stack(rows)
            if self.sharded:
                y_hat = self._basin_mean(y_hat)
        else:
            runoff, _ = self.model(self._forcing(x))
            y_hat = self._basin_mean(runoff.sum(dim=1)) if self.sharded else runoff.mean(dim=1)  # basin mean over the columns
            self.mass_balance.change_mass(self.model)
        self.mass_balance.report_mass(self.model, log=lambda s: None)
        warmup = int(self.cfg.models.hyperparameters.warmup)
        self.y_hat = y_hat[warmup:]
        self.y_t = y[warmup:].to(self.y_hat.device)
        return self.validate()

    def validate(self):
        nse = D.calculate_nse(self.y_hat.detach().cpu().numpy(), self.y_t.detach().cpu().numpy())
        loss_mse = self.criterion(self.y_hat, self.y_t)
        bound = self.range_bound_loss([self.model.alpha, self.model.n, self.model.ksat, self.model.ponded_depth_max])
        loss = loss_mse + bound.to(loss_mse.device)
        if self.sharded:
            # the range-bound penalty is a function of the (shared) parameters alone: every rank computes the same one, so each
            # contributes 1/world of its gradient to the sum
            (loss_mse + bound.to(loss_mse.device) / float(self.world)).backward()
            dist_.reduce_parameter_gradients([p.grad for p in self.model.parameters() if p.grad is not None], self.group)
        else:
            loss.backward()
        self.log("epoch %d: NSE %.4f loss %.6e" % (self.current_epoch + 1, nse, float(loss)))
        self.optimizer.step()
        self.history.append(dict(epoch=self.current_epoch + 1, nse=float(nse), loss=float(loss)))
        return float(loss)

    def parameters_in_sync(self):
        """Shared-parameter training keeps the parameters BIT-equal across ranks (same reduced gradients, same Adam state): True
        when this rank's parameters equal rank 0's bit for bit (always True without a process group)."""
        if not self.sharded:
            return True
        import torch.distributed as dist
        mine = torch.cat([p.detach().reshape(-1).to(torch.float64).cpu() for p in self.model.parameters()])
        ref = mine.clone()
        if dist.get_backend(self.group) == "gloo":
            dist.broadcast(ref, src=0, group=self.group)
        else:
            dev = ref.to(self.model.device)
            dist.broadcast(dev, src=0, group=self.group)
            ref = dev.cpu()
        return bool(torch.equal(mine.view(torch.int64), ref.view(torch.int64)))
