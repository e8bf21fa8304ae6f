"""Agent counterpart: the training loop that drives dpLGAR the way the reference's DifferentiableLGAR agent does
(/root/reference/dpLGAR/agents/DifferentiableLGAR.py:20-172): Adam over alpha/n/Ksat, MSE on runoff after a warm-up
slice plus a range-bound penalty (models/functions/loss.py:10-36), NSE log, per-epoch state and mass-balance reset.

MI355X-first difference: an epoch is ONE forward launch over the whole forcing series (model(x[T, N, 2])) and one batch
of tangent launches for the backward pass, instead of T Python-level forward() calls with a 10 ms sleep each
(DifferentiableLGAR.py:117-125).  `stepwise=True` reproduces the reference's row-by-row call sequence exactly.
"""
import torch
import torch.nn as nn

from . import data as D
from .model import MassBalance, dpLGAR


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
    def __init__(self, cfg, observations=None, stepwise=False, log=print):
        self.cfg = cfg
        self.log = log
        self.stepwise = stepwise
        torch.manual_seed(0)
        self.data = D.Data(cfg)
        if observations is not None:
            self.data.y = torch.as_tensor(observations, dtype=torch.float64)
        self.model = dpLGAR(cfg)
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

    def train_one_epoch(self):
        self.optimizer.zero_grad()
        x, y = self.data.x, self.data.y
        N = self.model.n_columns
        if self.stepwise:
            rows = []
            for i in range(len(self.data)):
                runoff, _ = self.model(x[i])
                rows.append(runoff.reshape(-1)[0] if N > 1 else runoff)
                self.mass_balance.change_mass(self.model)
            y_hat = torch.stack(rows)
        else:
            runoff, _ = self.model(x[:, None, :].expand(x.shape[0], N, 2))
            y_hat = runoff.mean(dim=1)  # basin mean over the columns
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
        loss.backward()
        self.log("epoch %d: NSE %.4f loss %.6e" % (self.current_epoch + 1, nse, float(loss)))
        self.optimizer.step()
        self.history.append(dict(epoch=self.current_epoch + 1, nse=float(nse), loss=float(loss)))
        return float(loss)
