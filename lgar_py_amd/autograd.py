"""Differentiable LGAR: torch autograd through the HIP kernels.

The reference differentiates forward() by recording every scalar torch op
(/root/reference/dpLGAR/agents/DifferentiableLGAR.py:119,163).  Here the vector-Jacobian product is assembled from
forward-mode tangents computed on the GPU (csrc/lgar_tangent.hip): columns are independent, so one tangent launch
with a one-hot direction over (alpha | n | Ksat, layer) yields d runoff_t / d p for every column's own parameter,
contracted on the fly with the incoming gradient.  3 x L launches per backward, nothing stored per step.
Line-search offsets are constants w.r.t. the parameters, exactly as in the reference (Layer.py:277-288, 683-696).
"""
import torch

from .engine import LgarEngine

KINDS = ("alpha", "n", "ksat")
# below this many (column, direction) pairs all directions of a backward pass ride as extra columns of ONE tangent
# launch (latency-bound small jobs: the reference's single-column training loop); above it one launch per direction
BATCH_DIRECTIONS_MAX_COLUMNS = 1 << 16


def parameter_vjp(eng, precip, pet, w_runoff, w_perc, wanted):
    """Vector-Jacobian product for every (kind, layer) in `wanted` (list of (kind, l)).
    Returns {(kind, l): grad[N]}.  Columns are independent, so for small jobs the directions are laid side by side as
    len(wanted) * N columns of a single launch; large jobs loop over directions (same total work, N-column memory)."""
    L, N, D = eng.L, eng.N, len(wanted)
    out = {}
    if D == 0:
        return out
    if D * N <= BATCH_DIRECTIONS_MAX_COLUMNS and D > 1:
        rep = lambda t: t.repeat(1, D)
        d = eng.dims
        big = LgarEngine(rep(eng.alpha), rep(eng.n), rep(eng.ksat), rep(eng.theta_e), rep(eng.theta_r), rep(eng.thickness),
                         dt_h=d.dt_h, num_subcycles=d.num_subcycles, initial_psi=d.initial_psi,
                         ponded_depth_max=d.ponded_depth_max, wilting_point_psi=d.wilting_point_psi,
                         frozen_factor=d.frozen_factor, nint=d.nint, giuh_ordinates=tuple(d.giuh[i] for i in range(d.n_giuh)),
                         dtype=eng.dtype, device=eng.device, iter_cap=d.iter_cap, search_mode=d.search_mode,
                         bottom_mode=d.bottom_mode, use_closed_form_G=bool(d.use_closed_form_G))
        dirs = {k: torch.zeros(L, D * N, dtype=eng.dtype, device=eng.device) for k in KINDS}
        for b, (kind, l) in enumerate(wanted):
            dirs[kind][l, b * N:(b + 1) * N] = 1.0
        tile = lambda t: None if t is None else t.repeat(1, D)
        g, _, _ = big.tangent(dirs, tile(precip), tile(pet), w_runoff=tile(w_runoff), w_perc=tile(w_perc))
        for b, key in enumerate(wanted):
            out[key] = g[b * N:(b + 1) * N]
        return out
    for kind, l in wanted:
        dmat = torch.zeros(L, N, dtype=eng.dtype, device=eng.device)
        dmat[l] = 1.0
        out[(kind, l)], _, _ = eng.tangent({kind: dmat}, precip, pet, w_runoff=w_runoff, w_perc=w_perc)
    return out


class LgarSeriesFunction(torch.autograd.Function):
    """(alpha, n, ksat)[L, N] -> (runoff, percolation)[T, N] from a fresh state; differentiable in alpha/n/ksat."""

    @staticmethod
    def forward(ctx, alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, engine_kw):
        engine_kw = dict(engine_kw)
        check = engine_kw.pop("check", True)
        status_out = engine_kw.pop("status_out", None)
        eng = LgarEngine(alpha.detach(), n.detach(), ksat.detach(), theta_e, theta_r, thickness, **engine_kw)
        out = eng.forward(precip, pet, series=("runoff", "percolation"), check=check)
        if status_out is not None:
            status_out.append(eng.status)
        ctx.engine = eng
        ctx.forcing = (precip, pet)
        ctx.in_dtypes = (alpha.dtype, n.dtype, ksat.dtype)
        return out["runoff"], out["percolation"]

    @staticmethod
    def backward(ctx, g_runoff, g_perc):
        eng = ctx.engine
        precip, pet = ctx.forcing
        wanted = [(kind, l) for ki, kind in enumerate(KINDS) if ctx.needs_input_grad[ki] for l in range(eng.L)]
        vj = parameter_vjp(eng, precip, pet, g_runoff, g_perc, wanted)
        grads = []
        for ki, kind in enumerate(KINDS):
            if not ctx.needs_input_grad[ki]:
                grads.append(None)
                continue
            grads.append(torch.stack([vj[(kind, l)] for l in range(eng.L)]).to(ctx.in_dtypes[ki]))
        return (*grads, None, None, None, None, None, None)


def lgar_series(alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, **engine_kw):
    """Differentiable run: parameters [L, N] (torch tensors, may require grad), forcing [T, N] -> runoff, percolation [T, N].
    engine_kw: LgarEngine keywords, plus check=False to keep going when columns fault (mask them with the status
    tensor appended to the list passed as status_out)."""
    return LgarSeriesFunction.apply(alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, engine_kw)


class StepTape:
    """Autograd for the reference's calling convention (`model(x[i])` once per forcing row -- or a whole [T, N, 2] block --
    with the loss taken at the end of the epoch).  Every returned runoff/percolation block is a leaf that records the
    gradient it receives; when the backward pass finishes, ONE batch of tangent launches over the recorded forcing
    series turns the recorded weights into parameter gradients (accumulated into .grad like autograd would) --
    O(T) work per epoch, nothing stored per step but the forcing rows."""

    def __init__(self, model):
        self.model = model
        self.reset()

    def reset(self):
        for h in getattr(self, "handles", []):
            h.remove()
        self.handles = []    # hook handles of the leaves handed out
        self.x = []          # forcing chunks [Tc, N, 2] since the last set_internal_states()
        self.w = {}          # (chunk, 0|1) -> gradient received, [Tc, N]
        self.queued = False

    def record(self, x_chunk, runoff_chunk, perc_chunk):
        """x_chunk [Tc, N, 2]; runoff/perc [Tc, N].  Returns leaf tensors standing for this chunk's per-step values."""
        ci = len(self.x)
        self.x.append(x_chunk.detach())
        outs = []
        for which, v in enumerate((runoff_chunk, perc_chunk)):
            leaf = v.detach().clone().requires_grad_(True)
            self.handles.append(leaf.register_hook(lambda g, ci=ci, which=which: self._on_grad(ci, which, g)))
            outs.append(leaf)
        return outs

    def _on_grad(self, ci, which, g):
        key = (ci, which)
        self.w[key] = self.w[key] + g.detach() if key in self.w else g.detach().clone()
        if not self.queued:
            self.queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        return None

    def _finalize(self):
        self.queued = False
        m = self.model
        eng = m.engine
        X = torch.cat(self.x).to(eng.device, eng.dtype)  # [T, N, 2]
        T = X.shape[0]
        W = torch.zeros(2, T, eng.N, dtype=eng.dtype, device=eng.device)
        t0 = 0
        for ci, xc in enumerate(self.x):
            for which in (0, 1):
                g = self.w.get((ci, which))
                if g is not None:
                    W[which, t0:t0 + xc.shape[0]] = g.to(eng.device, eng.dtype).reshape(xc.shape[0], eng.N)
            t0 += xc.shape[0]
        self.w = {}
        precip, pet = X[:, :, 0].contiguous(), X[:, :, 1].contiguous()
        ff = float(m.cfg.constants.frozen_factor)
        plists = (("alpha", m.alpha), ("n", m.n), ("ksat", m.ksat))
        wanted = [(kind, l) for kind, plist in plists for l, p in enumerate(plist) if p.requires_grad]
        vj = parameter_vjp(eng, precip, pet, W[0], W[1], wanted)
        for kind, plist in plists:
            for l, p in enumerate(plist):
                if not p.requires_grad:
                    continue
                g = vj[(kind, l)]
                if kind == "ksat":
                    g = g / ff  # the Parameter already carries frozen_factor (models/dpLGAR.py:57)
                g = g.to(torch.float64).to(p.device)
                g = g.sum() if p.dim() == 0 else g
                p.grad = g.to(p.dtype) if p.grad is None else p.grad + g.to(p.dtype)
        # the recorded series is consumed: drop the hooks (they hold this tape, the tape holds the model) so nothing
        # autograd-related is left for interpreter shutdown to untangle
        for h in self.handles:
            h.remove()
        self.handles = []
