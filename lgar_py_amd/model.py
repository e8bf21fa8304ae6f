"""dpLGAR(nn.Module): the reference's model surface (/root/reference/dpLGAR/models/dpLGAR.py:30-299) backed by
the MI355X engine, for one column (drop-in) or N columns (batched).

Same constructor (`dpLGAR(cfg)`), same learnable attributes (.alpha/.n/.ksat, .ponded_depth_max), same
`.forward(x) -> (runoff, percolation)`, `.set_internal_states()`, `.update_soil_parameters()`, and the same
accumulator attributes that MassBalance reads and zeroes (physics/MassBalance.py:31-53).  The per-timestep
physics runs in csrc/*.hip through the C-ABI; nothing here computes LGAR on the host.
"""
import numpy as np
import torch
import torch.nn as nn

from . import data as D
from ._capi import ACC_NAMES, LgarError
from .engine import LgarEngine

SOIL_INDEX = {"theta_r": 0, "theta_e": 1, "theta_wp": 2, "theta_init": 3, "m": 4, "bc_lambda": 5, "bc_psib_cm": 6,
              "h_min_cm": 7}


class GlobalParams:
    """The handful of GlobalParams fields callers read (physics/GlobalParams.py:79-138)."""

    def __init__(self, cfg, ponded_depth_max):
        self.device = cfg.device
        self.layer_thickness_cm = torch.tensor(list(cfg.data.layer_thickness), dtype=torch.float64)
        self.cum_layer_thickness = torch.cumsum(self.layer_thickness_cm, 0)
        self.num_layers = len(cfg.data.layer_thickness)
        self.soil_depth_cm = self.cum_layer_thickness[-1]
        self.initial_psi = torch.tensor(float(cfg.data.initial_psi), dtype=torch.float64)
        self.ponded_depth_max = ponded_depth_max.clone()
        self.use_closed_form_G = cfg.data.use_closed_form_G
        self.layer_soil_type = np.array(cfg.data.layer_soil_type) - 1
        self.wilting_point_psi_cm = torch.tensor(float(cfg.data.wilting_point_psi), dtype=torch.float64)
        self.frozen_factor = torch.tensor(float(cfg.constants.frozen_factor), dtype=torch.float64)
        self.nint = torch.tensor(int(cfg.constants.nint))
        self.giuh_ordinates = torch.tensor(list(cfg.data.giuh_ordinates), dtype=torch.float64)
        self.num_giuh_ordinates = len(self.giuh_ordinates)
        self.giuh_runoff = torch.zeros(self.num_giuh_ordinates, dtype=torch.float64)
        self.sft_coupled = False
        self.soil_index = cfg.data.soil_index


class dpLGAR(nn.Module):
    # plain-tensor attributes the agent loop rewrites ~20 times per forcing row (forward() and MassBalance.change_mass):
    # stored without nn.Module's parameter / buffer / submodule bookkeeping (2.5 us per assignment, a third of a row's host time)
    _PLAIN = frozenset(("precip", "PET", "AET", "infiltration", "runoff", "giuh_runoff", "discharge", "groundwater_discharge",
                        "percolation", "previous_precip", "_latest", "steps_advanced"))

    def __setattr__(self, name, value):
        if name in dpLGAR._PLAIN:
            object.__setattr__(self, name, value)
        else:
            super().__setattr__(name, value)

    def __init__(self, cfg, n_columns=None, theta_e=None, theta_r=None, alpha=None, n=None, ksat=None):
        """cfg: the reference's config keys (see config.py).  Optional per-column overrides ([N, L] tensors) turn the
        single column into an ensemble; otherwise every column gets the table values of cfg.data.layer_soil_type."""
        super().__init__()
        self.cfg = cfg
        self.n_columns = int(n_columns if n_columns is not None else cfg.get("n_columns", 1) or 1)
        if str(cfg.device) == "cpu":
            # the reference only ever runs with `device: cpu` (its configs say so); this engine has no CPU path, and
            # silently moving the model to a GPU would hide that: ask for the device explicitly
            raise LgarError("cfg.device is 'cpu': the MI355X LGAR engine has no CPU fallback; set cfg.device to 'cuda:0' "
                            "(config.load_config does this by default)")
        self.device = torch.device(cfg.device)
        self.dtype = torch.float32 if str(cfg.get("dtype", "float64")) in ("float32", "f32") else torch.float64
        N = self.n_columns
        alpha_, n_, ksat_ = D.read_test_params(cfg)
        st = list(cfg.data.layer_soil_type)
        self.ponded_depth_max = torch.tensor(float(cfg.data.ponded_depth_max), dtype=torch.float64)
        ff = float(cfg.constants.frozen_factor)

        def plist(table, override, scale=1.0):
            if override is not None:  # ensemble: one [N] Parameter per layer
                o = torch.as_tensor(override, dtype=torch.float64)
                return nn.ParameterList([nn.Parameter(o[:, i].clone() * scale) for i in range(o.shape[1])])
            return nn.ParameterList([nn.Parameter(table[i].clone() * scale) for i in st])

        self.alpha = plist(alpha_, alpha)
        self.n = plist(n_, n)
        self.ksat = plist(ksat_, ksat, ff)  # frozen factor folded in, models/dpLGAR.py:57
        self._theta_e_override, self._theta_r_override = theta_e, theta_r
        self.cfg.data.soil_index = dict(SOIL_INDEX)
        self.soils_df = self.texture_map = self.c = None
        self.global_params = None
        self.engine = None
        self.num_wetting_fronts = None
        self.tape = None
        self.set_internal_states()

    # ----------------------------------------------------------------------------------------------
    def _param_matrix(self, plist):
        """ParameterList of L entries (0-d or [N]) -> [L, N] fp64 tensor."""
        rows = []
        for p in plist:
            v = p.detach().to(torch.float64).cpu()
            rows.append(v.expand(self.n_columns) if v.dim() == 0 else v)
        return torch.stack(rows)

    def _soil_thetas(self):
        L, N = len(self.alpha), self.n_columns
        st = list(self.cfg.data.layer_soil_type)
        te = torch.tensor(self.soils_df["theta_e"][st], dtype=torch.float64)[:, None].expand(L, N)
        tr = torch.tensor(self.soils_df["theta_r"][st], dtype=torch.float64)[:, None].expand(L, N)
        if self._theta_e_override is not None:
            te = torch.as_tensor(self._theta_e_override, dtype=torch.float64).T
        if self._theta_r_override is not None:
            tr = torch.as_tensor(self._theta_r_override, dtype=torch.float64).T
        return te.contiguous(), tr.contiguous()

    def set_internal_states(self):
        """models/dpLGAR.py:97-147: rebuild the soil stack from the current parameters and zero the accumulators."""
        cfg = self.cfg
        self.soils_df = D.read_soil_table(cfg.data.soil_params_file)
        self.texture_map = dict(enumerate(self.soils_df["Texture"]))
        self.global_params = GlobalParams(cfg, self.ponded_depth_max)
        te, tr = self._soil_thetas()
        ff = float(cfg.constants.frozen_factor)
        L, N = len(self.alpha), self.n_columns
        thick = torch.tensor(list(cfg.data.layer_thickness), dtype=torch.float64)[:, None].expand(L, N)
        # the engine applies frozen_factor itself; self.ksat already carries it
        self.engine = LgarEngine(self._param_matrix(self.alpha), self._param_matrix(self.n),
                                 self._param_matrix(self.ksat) / ff, te, tr, thick,
                                 dt_h=float(cfg.models.subcycle_length_h), num_subcycles=int(cfg.models.num_subcycles),
                                 initial_psi=float(cfg.data.initial_psi),
                                 ponded_depth_max=float(self.ponded_depth_max),
                                 wilting_point_psi=float(cfg.data.wilting_point_psi), frozen_factor=ff,
                                 nint=int(cfg.constants.nint), giuh_ordinates=tuple(cfg.data.giuh_ordinates),
                                 use_closed_form_G=bool(cfg.data.use_closed_form_G),
                                 geff_precision=str(cfg.get("geff_precision", "native") or "native"),
                                 dtype=self.dtype, device=self.device)
        self.c = self._soil_metrics(te, tr)
        from .autograd import StepTape
        self.tape = StepTape(self)
        self._latest = None  # ponded_water / ending_volume as of the last forward() (rows 8, 9 of its call_sums)
        self.steps_advanced = 0  # forcing rows integrated since this reset (the tape checks its record against it)
        self.num_wetting_fronts = self.calc_num_wetting_fronts()
        # Small models keep the accumulator attributes on the HOST, where the reference keeps them: the agent's loop touches
        # ~20 of them per forcing row (MassBalance.change_mass), and each touch of a GPU tensor is a kernel launch.
        self.attr_device = torch.device("cpu") if N <= 64 else self.device
        z = lambda: self._shape(torch.zeros(N, dtype=torch.float64, device=self.attr_device))
        self.precip, self.PET, self.AET = z(), z(), z()
        self.infiltration, self.runoff, self.giuh_runoff = z(), z(), z()
        self.discharge, self.groundwater_discharge, self.percolation = z(), z(), z()
        self.previous_precip = z()

    def _soil_metrics(self, te, tr):
        """Per-layer derived table [theta_r, theta_e, theta_wp, theta_init, m, bc_lambda, bc_psib, h_min] of column 0
        (data/utils.py:40-105); informational, the kernels derive what they need themselves."""
        a = self._param_matrix(self.alpha)[:, 0]
        n = self._param_matrix(self.n)[:, 0]
        m = 1.0 - 1.0 / n
        th = lambda h: tr[:, 0] + (te[:, 0] - tr[:, 0]) / (1.0 + (a * h) ** n) ** m
        p = 1.0 + 2.0 / m
        lam = 2.0 / (p - 3.0)
        psib = (p + 3.0) * (147.8 + 8.1 * p + 0.092 * p * p) / (2.0 * a * p * (p - 1.0) * (55.6 + 7.4 * p + p * p))
        hmin = psib * (2.0 + 3.0 / lam) / (1.0 + 3.0 / lam)
        return torch.stack([tr[:, 0], te[:, 0], th(float(self.cfg.data.wilting_point_psi)),
                            th(float(self.cfg.data.initial_psi)), m, lam, psib, hmin], 1)

    def update_soil_parameters(self):
        """models/dpLGAR.py:149-152: push the current alpha/n/Ksat/ponded_depth_max to the engine, keeping the state."""
        self.global_params.ponded_depth_max = self.ponded_depth_max.clone()
        e = self.engine
        ff = float(self.cfg.constants.frozen_factor)
        e.alpha.copy_(self._param_matrix(self.alpha).to(e.device, e.dtype))
        e.n.copy_(self._param_matrix(self.n).to(e.device, e.dtype))
        e.ksat.copy_((self._param_matrix(self.ksat) / ff).to(e.device, e.dtype))
        e.dims.ponded_depth_max = float(self.ponded_depth_max)

    # ----------------------------------------------------------------------------------------------
    def _shape(self, t):
        return t[0] if self.n_columns == 1 else t

    def forward(self, x):
        """x: [2] (precip, PET in cm/h; one column or broadcast), [N, 2], or [T, N, 2] for T steps in one launch.
        Returns (runoff, percolation): the accumulators, like the reference (models/dpLGAR.py:299); for [T, N, 2]
        input the per-step series [T, N]."""
        x = torch.as_tensor(x, dtype=torch.float64)
        N = self.n_columns
        series_mode = x.dim() == 3
        if x.dim() == 1:
            x = x[None, None, :].expand(1, N, 2)
        elif x.dim() == 2:
            x = x[None]
        if x.shape[1] != N or x.shape[2] != 2:
            raise LgarError("forcing must be [2], [N, 2] or [T, N, 2] with N = %d" % N)
        grad_mode = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        want = ("runoff", "percolation") if (series_mode or grad_mode) else ()
        status_host = None
        if x.shape[0] == 1 and self.attr_device.type == "cpu" and not x.is_cuda:
            # the reference's convention, one row per call with the accumulators on the host: the lean path
            sums_h, r_h, p_h, status_host = self.engine.step_rows_host(x[0, :, 0], x[0, :, 1])
            out = {"call_sums": sums_h.to(torch.float64).clone()}
            if want:
                out["runoff"], out["percolation"] = r_h[None, :].to(torch.float64).clone(), p_h[None, :].to(torch.float64).clone()
        else:
            out = self.engine.forward(x[:, :, 0], x[:, :, 1], series=want, check=False, call_sums=True)
        steps_before = self.steps_advanced
        self.steps_advanced += int(x.shape[0])
        # the 8 accumulators summed over this call's steps (+ latest ponded_water / ending_volume), [8+2, N]
        sums = out["call_sums"].to(self.attr_device, torch.float64)
        self._latest = sums[8:10]
        for j, nm in enumerate(ACC_NAMES[:8]):
            if grad_mode and nm in ("runoff", "percolation"):
                continue
            setattr(self, nm, getattr(self, nm) + self._shape(sums[j]))
        r_series, p_series = out.get("runoff"), out.get("percolation")
        if grad_mode:
            # graph-connected outputs (models/dpLGAR.py:299): this block's per-step runoff / percolation become the
            # outputs of an autograd node whose inputs are the parameters (autograd.StepTape)
            r_series, p_series = self.tape.record(x, r_series.to(self.attr_device, torch.float64),
                                                  p_series.to(self.attr_device, torch.float64), steps_before)
            self.runoff = self.runoff + self._shape(r_series.sum(0))
            self.percolation = self.percolation + self._shape(p_series.sum(0))
        self.previous_precip = self._shape((x[-1, :, 0] * float(self.cfg.models.subcycle_length_h)).to(self.attr_device))
        self.groundwater_discharge = self.groundwater_discharge * 0.0
        # raises ValueError like the reference, after the attributes are up to date
        if status_host is not None:
            self.engine.raise_for_status(status_host)
        else:
            self.engine.check_status()
        if series_mode:
            return r_series, p_series
        return self.runoff, self.percolation

    # state the agent / MassBalance read -----------------------------------------------------------
    @property
    def ponded_water(self):
        if self._latest is not None:
            return self._shape(self._latest[0])
        return self._shape(self.engine.ponded_water.to(self.attr_device, torch.float64))

    @property
    def ending_volume(self):
        if self._latest is not None:
            return self._shape(self._latest[1])
        return self._shape(self.engine.ending_volume.to(self.attr_device, torch.float64))

    @property
    def giuh_runoff_queue(self):
        q = self.engine.giuh_runoff_queue.to(torch.float64)
        return q[:, 0] if self.n_columns == 1 else q

    def calc_mass_balance(self):
        return self.ending_volume

    def calc_num_wetting_fronts(self):
        nf = self.engine.n_fronts
        return int(nf[0]) if self.n_columns == 1 else nf.clone()

    def wetting_fronts(self, column=0):
        """Front table of one column, top -> bottom: list of dicts with the WettingFront fields
        (layers/WettingFront.py:38-49)."""
        fr = self.engine.fronts()
        nf = int(fr["n_fronts"][column])
        return [dict(depth=float(fr["depth"][i, column]), theta=float(fr["theta"][i, column]),
                     psi_cm=float(fr["psi"][i, column]), k_cm_per_h=float(fr["k"][i, column]),
                     dzdt=float(fr["dzdt"][i, column]), layer_num=int(fr["layer"][i, column]),
                     to_bottom=bool(fr["to_bottom"][i, column])) for i in range(nf)]

    def print_params(self):
        for nm, pl in (("Alpha", self.alpha), ("n", self.n), ("Ksat", self.ksat)):
            for i, p in enumerate(pl):
                print("%s for soil %d: %.4f" % (nm, i + 1, float(p.detach().reshape(-1)[0])))
        print("Max Ponded Depth: %.4f" % float(self.ponded_depth_max))


class MassBalance:
    """Counterpart of physics/MassBalance.py: accumulates the model's per-step accumulators and zeroes them."""

    NAMES = ("precip", "infiltration", "AET", "percolation", "runoff", "giuh_runoff", "discharge", "PET",
             "groundwater_discharge")

    def __init__(self, cfg, model):
        self.device = cfg.device
        self.set_internal_states(model)

    def set_internal_states(self, model):
        z = lambda: torch.zeros_like(model.precip)
        for nm in self.NAMES:
            setattr(self, nm, z())
        self.starting_volume = model.ending_volume.clone()
        self.ending_volume = z()
        self.ponded_depth = z()
        self.ponded_water = z()

    def reset_mass(self, model):
        self.set_internal_states(model)

    def change_mass(self, model):
        for nm in self.NAMES:
            setattr(self, nm, getattr(self, nm) + getattr(model, nm))
            setattr(model, nm, torch.zeros_like(getattr(model, nm)))
        self.ponded_water = model.ponded_water

    def report_mass(self, model, log=print):
        self.ending_volume = model.ending_volume
        err = (self.starting_volume + self.precip - self.runoff - self.AET - self.ponded_water - self.percolation
               - self.ending_volume)
        m = lambda t: float(t.double().mean())
        log("-------------------- Simulation Summary (column mean) ----------------- ")
        for label, v in (("Initial water in soil", self.starting_volume), ("Total precipitation", self.precip),
                         ("Total infiltration", self.infiltration), ("Final water in soil", self.ending_volume),
                         ("Surface ponded water", self.ponded_water), ("Surface runoff", self.runoff),
                         ("GIUH runoff", self.giuh_runoff), ("Total percolation", self.percolation),
                         ("Total AET", self.AET), ("Total PET", self.PET), ("Total discharge (Q)", self.discharge)):
            log("%-24s = %14f cm" % (label, m(v)))
        log("Global balance           =   %.6e cm" % m(err))
        return err
