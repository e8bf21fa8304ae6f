"""Differentiable LGAR: torch autograd through the HIP kernels.

The reference differentiates forward() by recording every scalar torch op
(/root/reference/dpLGAR/agents/DifferentiableLGAR.py:119,163).  Here the vector-Jacobian product is assembled from
forward-mode tangents computed on the GPU (csrc/lgar_tangent_nl.hip): columns are independent, so one tangent launch
with a one-hot direction over (alpha | n | Ksat, layer) yields d runoff_t / d p for every column's own parameter,
contracted on the fly with the incoming gradient.  3 x L launches per backward, nothing stored per step.
Line-search offsets are constants w.r.t. the parameters, exactly as in the reference (Layer.py:277-288, 683-696).

Both entry points are ordinary torch.autograd.Function nodes whose inputs are the parameters, so the outputs carry a
grad_fn that reaches them (models/dpLGAR.py:299) and torch.autograd.grad / loss.backward() both work.
"""
import weakref

import torch

from ._capi import ST_FAULT_MASK
from .engine import LgarEngine, LgarError, LgarStatusError

KINDS = ("alpha", "n", "ksat")
# the directions of a backward pass ride side by side as extra columns of a tangent launch (column-major: column
# c * D + b integrates column c's parameters perturbed in direction b, so that a column's directions sit in adjacent lanes);
# forcing and weights are NOT replicated -- the kernels broadcast them (LgarDims.forcing_columns / forcing_group).  Above this
# many (column, direction) pairs the directions go in groups; in fp64 fast modes the lanes of a column share the Geff trapezoid.
BATCH_DIRECTIONS_MAX_COLUMNS = 1 << 23
SHARE_MIN_LANES = 65536  # columns x directions: one wave on every SIMD of the chip


def parameter_vjp(eng, precip, pet, w_runoff, w_perc, wanted, check=True):
    """Vector-Jacobian product for every (kind, layer) in `wanted` (list of (kind, l)).
    Returns ({(kind, l): grad[N]}, tangent_status[N]).  Columns are independent, so the directions are laid side by side as
    len(wanted) * N columns of a single launch (fills the chip even for ensembles of 10^5 columns; very large jobs go in
    groups of directions).  A column whose tangent integration faults (status != 0: NaN, iteration cap, front overflow
    ...) has no valid gradient: with check=True that raises LgarStatusError (a ValueError, like the reference's physics
    faults), with check=False its gradient entries are zeroed and the status tensor says which columns those are."""
    L, N, D = eng.L, eng.N, len(wanted)
    out = {}
    status = torch.zeros(N, dtype=torch.int32, device=eng.device)
    if D == 0:
        return out, status
    group = max(1, min(D, BATCH_DIRECTIONS_MAX_COLUMNS // max(N, 1)))
    d = eng.dims
    # fp64 fast modes: the directions of a column sit in adjacent lanes that SHARE the Geff trapezoid (LgarDims.tangent_share =
    # group width; the tangent of the trapezoid needs only five direction-independent sums, so all 3 x L directions of a column
    # fit one group: 9 lanes for three layers, seven groups per wavefront)
    # (only when the launch fills the chip: a small job is bound by the latency of ONE wave)
    share = (eng.dtype == torch.float64 and d.search_mode != 0 and not d.use_closed_form_G and 2 <= group <= 32
             and N * group >= SHARE_MIN_LANES)
    for g0 in range(0, D, group):
        part = wanted[g0:g0 + group]
        Dg = len(part)
        if Dg == 1:
            kind, l = part[0]
            dmat = torch.zeros(L, N, dtype=eng.dtype, device=eng.device)
            dmat[l] = 1.0
            out[(kind, l)], _, st = eng.tangent({kind: dmat}, precip, pet, w_runoff=w_runoff, w_perc=w_perc)
            status |= st
            continue
        # column n's Dg directions sit in ADJACENT lanes (column n * Dg + b): they follow the same branches, so a wavefront
        # diverges over 64 / Dg columns instead of 64; the kernel reads forcing / weight column c // Dg (forcing_group)
        rep = lambda t: t.repeat_interleave(Dg, dim=1)
        big = LgarEngine(rep(eng.alpha), rep(eng.n), rep(eng.ksat), rep(eng.theta_e), rep(eng.theta_r), rep(eng.thickness),
                         dt_h=d.dt_h, num_subcycles=d.num_subcycles, initial_psi=d.initial_psi,
                         ponded_depth_max=d.ponded_depth_max, wilting_point_psi=d.wilting_point_psi,
                         frozen_factor=d.frozen_factor, nint=d.nint, giuh_ordinates=tuple(d.giuh[i] for i in range(d.n_giuh)),
                         dtype=eng.dtype, device=eng.device, iter_cap=d.iter_cap, search_mode=d.search_mode,
                         bottom_mode=d.bottom_mode, use_closed_form_G=bool(d.use_closed_form_G), front_slots=eng.front_slots,
                         with_state=False)
        dirs = {k: torch.zeros(L, Dg * N, dtype=eng.dtype, device=eng.device) for k in KINDS}
        for b, (kind, l) in enumerate(part):
            dirs[kind][l, b::Dg] = 1.0
        g, _, st = big.tangent(dirs, precip, pet, w_runoff=w_runoff, w_perc=w_perc, forcing_group=Dg,  # forcing / weights broadcast by the kernel
                               share=Dg if share else 0)
        for b, key in enumerate(part):
            out[key] = g[b::Dg].contiguous()
            status |= st[b::Dg]
    status &= ST_FAULT_MASK
    bad = status != 0
    if bool(bad.any()):
        if check:
            first = int(torch.nonzero(bad)[0].item())
            raise LgarStatusError("%d of %d columns faulted in the tangent (gradient) integration; first column %d, status %d"
                                  % (int(bad.sum().item()), N, first, int(status[first].item())))
        for key in out:
            out[key] = torch.where(bad, torch.zeros_like(out[key]), out[key])
    return out, status


class LgarSeriesFunction(torch.autograd.Function):
    """(alpha, n, ksat)[L, N] -> (runoff, percolation)[T, N] from a fresh state; differentiable in alpha/n/ksat."""

    @staticmethod
    def forward(ctx, alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, engine_kw):
        engine_kw = dict(engine_kw)
        check = engine_kw.pop("check", True)
        status_out = engine_kw.pop("status_out", None)
        eng = LgarEngine(alpha.detach(), n.detach(), ksat.detach(), theta_e, theta_r, thickness, **engine_kw)
        precip, pet = torch.as_tensor(precip), torch.as_tensor(pet)
        if precip.dim() == 2 and precip.shape[1] != eng.N:
            # broadcast forcing [T, Nf]: the backward pass pairs the forcing with the incoming [T, N] gradient column by
            # column (the tangent kernels take one layout for both), so the series is expanded here, once
            if eng.N % max(int(precip.shape[1]), 1) != 0 or precip.shape != pet.shape:
                raise LgarError("forcing must be [T, N] or [T, Nf] with Nf dividing N = %d; got %s / %s"
                                % (eng.N, tuple(precip.shape), tuple(pet.shape)))
            rep = eng.N // int(precip.shape[1])
            precip, pet = precip.repeat(1, rep).contiguous(), pet.repeat(1, rep).contiguous()  # column c reads c % Nf
        out = eng.forward(precip, pet, series=("runoff", "percolation"), check=check)
        if status_out is not None:
            status_out.append(eng.status)
        ctx.engine = eng
        ctx.forcing = (precip, pet)
        ctx.in_dtypes = (alpha.dtype, n.dtype, ksat.dtype)
        ctx.check = check
        ctx.status_out = status_out
        return out["runoff"], out["percolation"]

    @staticmethod
    def backward(ctx, g_runoff, g_perc):
        eng = ctx.engine
        precip, pet = ctx.forcing
        wanted = [(kind, l) for ki, kind in enumerate(KINDS) if ctx.needs_input_grad[ki] for l in range(eng.L)]
        # columns that already faulted in the forward run are the caller's to mask (status_out); they carry no gradient
        fwd_bad = (eng.status & ST_FAULT_MASK) != 0
        keep = (~fwd_bad).to(g_runoff.dtype)[None, :] if g_runoff is not None else None
        gr = None if g_runoff is None else g_runoff * keep
        gp = None if g_perc is None else g_perc * ((~fwd_bad).to(g_perc.dtype)[None, :])
        vj, st = parameter_vjp(eng, precip, pet, gr, gp, wanted, check=False)
        st = torch.where(fwd_bad, torch.zeros_like(st), st)
        if ctx.status_out is not None:
            ctx.status_out.append(st)  # [forward status, tangent status]
        if ctx.check and bool((st != 0).any()):
            raise LgarStatusError("%d columns faulted in the tangent (gradient) integration" % int((st != 0).sum().item()))
        grads = []
        for ki, kind in enumerate(KINDS):
            if not ctx.needs_input_grad[ki]:
                grads.append(None)
                continue
            g = torch.stack([vj[(kind, l)] for l in range(eng.L)])
            g = torch.where(fwd_bad[None, :], torch.zeros_like(g), g)
            grads.append(g.to(ctx.in_dtypes[ki]))
        return (*grads, None, None, None, None, None, None)


def lgar_series(alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, **engine_kw):
    """Differentiable run: parameters [L, N] (torch tensors, may require grad), forcing [T, N] -> runoff, percolation [T, N].
    engine_kw: LgarEngine keywords, plus check=False to keep going when columns fault: pass status_out=[] to receive the
    forward status tensor (appended by the forward pass) and the tangent status tensor (appended by the backward pass);
    faulted columns get zero gradient."""
    return LgarSeriesFunction.apply(alpha, n, ksat, theta_e, theta_r, thickness, precip, pet, engine_kw)


class _ChunkFunction(torch.autograd.Function):
    """One model.forward() call's (runoff, percolation) block as an autograd node with the parameters as inputs.
    `token` chains the nodes of an epoch in call order, so autograd necessarily reaches chunk 0 last: there the weights
    recorded by all chunks are turned into parameter gradients by ONE batch of tangent launches over the whole series."""

    @staticmethod
    def forward(ctx, tape, ci, token, runoff, perc, *params):
        ctx.tape_ref, ctx.ci, ctx.n_params = weakref.ref(tape), ci, len(params)  # the model owns the tape; the graph does not
        ctx.set_materialize_grads(False)
        return runoff.clone(), perc.clone(), token.new_zeros(())

    @staticmethod
    def backward(ctx, g_runoff, g_perc, g_token):
        tape = ctx.tape_ref()
        if tape is None:
            raise LgarError("backward through a model that no longer exists")
        tape.add_weights(ctx.ci, g_runoff, g_perc)
        if ctx.ci == 0:
            return (None, None, None, None, None, *tape.finalize())
        # a defined (zero) gradient for the token keeps the chain down to chunk 0 alive
        return (None, None, torch.zeros((), dtype=torch.float64), None, None, *((None,) * ctx.n_params))


class StepTape:
    """Autograd for the reference's calling convention (`model(x[i])` once per forcing row -- or a whole [T, N, 2] block --
    with the loss taken at the end of the epoch).  Every returned runoff/percolation block is the output of an autograd
    node whose inputs are the model's parameters; the backward pass records the gradient each block receives and, at the
    node of the epoch's first block, ONE batch of tangent launches over the recorded forcing series turns the recorded
    weights into parameter gradients -- O(T) work per epoch, nothing stored per step but the forcing rows."""

    def __init__(self, model):
        self._model = weakref.ref(model)  # the model owns the tape, not the other way round
        self.reset()

    def reset(self):
        self.x = []          # forcing chunks [Tc, N, 2] since the last set_internal_states()
        self.n_rows = 0      # ... and how many forcing rows they hold
        self.w = {}          # (chunk, 0|1) -> gradient received, [Tc, N]
        self.token = None
        self.versions = None
        self._params = None  # the model's parameters in (alpha, n, ksat) x layer order, looked up once per recorded series

    def _param_lists(self):
        m = self._model()
        if m is None:
            raise LgarError("the model this tape belongs to is gone")
        return m, (("alpha", m.alpha), ("n", m.n), ("ksat", m.ksat))

    def record(self, x_chunk, runoff_chunk, perc_chunk, steps_before=None):
        """x_chunk [Tc, N, 2]; runoff/perc [Tc, N] (engine outputs); steps_before: forcing rows the model had integrated
        since set_internal_states() before this chunk.  Returns the graph-connected blocks."""
        if self._params is None:
            _, plists = self._param_lists()
            self._params = [p for _, pl in plists for p in pl]
        params = self._params
        recorded = self.n_rows
        if steps_before is not None and steps_before != recorded:
            # finalize() re-integrates the tangent from a FRESH state over the recorded rows only: rows the model advanced
            # off the tape (under torch.no_grad(), or while no parameter required grad -- a spin-up, say) would make the
            # backward pass replay another forcing series and another state than the forward run
            raise LgarError("the model advanced %d forcing rows since set_internal_states() that are not on the autograd tape "
                            "(%d recorded): forward() calls under torch.no_grad() cannot be mixed with differentiated ones "
                            "within one series; call set_internal_states() first" % (steps_before - recorded, recorded))
        versions = tuple(p._version for p in params)
        if self.versions is None:
            self.versions = versions
        elif versions != self.versions:
            raise LgarError("parameters changed in the middle of a recorded series (update_soil_parameters / optimizer step "
                            "before backward): call set_internal_states() first, the gradient would belong to another run")
        ci = len(self.x)
        self.x.append(x_chunk.detach())
        self.n_rows += int(x_chunk.shape[0])
        token = self.token if self.token is not None else torch.zeros((), dtype=torch.float64)
        r, p, self.token = _ChunkFunction.apply(self, ci, token, runoff_chunk, perc_chunk, *params)
        return r, p

    def add_weights(self, ci, g_runoff, g_perc):
        for which, g in enumerate((g_runoff, g_perc)):
            if g is None:
                continue
            key = (ci, which)
            self.w[key] = self.w[key] + g.detach() if key in self.w else g.detach().clone()

    def finalize(self):
        """Parameter gradients (in parameter order) from the weights recorded so far; called by chunk 0's backward."""
        m, plists = self._param_lists()
        eng = m.engine
        X = torch.cat(self.x).to(eng.device, eng.dtype)  # [T, N, 2]
        T = X.shape[0]
        if getattr(m, "steps_advanced", T) != T:
            raise LgarError("the tape holds %d forcing rows but the model advanced %d since set_internal_states(): the gradient "
                            "would belong to another run" % (T, m.steps_advanced))
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
        wanted = [(kind, l) for kind, plist in plists for l, p in enumerate(plist) if p.requires_grad]
        vj, _ = parameter_vjp(eng, precip, pet, W[0], W[1], wanted, check=True)
        grads = []
        for kind, plist in plists:
            for l, p in enumerate(plist):
                if not p.requires_grad:
                    grads.append(None)
                    continue
                g = vj[(kind, l)]
                if kind == "ksat":
                    g = g / ff  # the Parameter already carries frozen_factor (models/dpLGAR.py:57)
                g = g.to(torch.float64).to(p.device)
                g = g.sum() if p.dim() == 0 else g
                grads.append(g.to(p.dtype))
        return tuple(grads)
