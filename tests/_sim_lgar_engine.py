"""TEST INFRASTRUCTURE ONLY: the device-code simulator (tests/devsim) behind the slice of lgar_py_amd.engine.LgarEngine's
interface that model.dpLGAR, autograd.StepTape / parameter_vjp and the agent use -- torch CPU tensors in and out.  CPU tests
inject it (monkeypatching lgar_py_amd.model.LgarEngine and lgar_py_amd.autograd.LgarEngine) to run the host-side logic --
the model surface, the autograd tape, the sharded training loop under gloo -- without a GPU.  Never imported by the product."""
import numpy as np
import torch

import devsim
from lgar_py_amd._capi import ACC_NAMES, NACC
from lgar_py_amd.engine import LgarError, LgarStatusError


class SimLgarEngine:
    def __init__(self, alpha, n, ksat, theta_e, theta_r, thickness, *, n_columns=None, dt_h=1.0, num_subcycles=1,
                 initial_psi=2000.0, ponded_depth_max=0.0, wilting_point_psi=15495.0, frozen_factor=1.0, nint=120,
                 giuh_ordinates=(0.06, 0.51, 0.28, 0.12, 0.03), dtype=torch.float64, device="cpu", iter_cap=0, search_mode=1,
                 bottom_mode=0, use_closed_form_G=False, front_slots=None, with_state=True, geff_precision="native",
                 forward_lanes=0, basin_scratch_bytes=0):
        npdt = np.float64 if dtype == torch.float64 else np.float32
        c = lambda t: torch.as_tensor(t, dtype=torch.float64).cpu().numpy()
        self.sim = devsim.SimEngine(c(alpha), c(n), c(ksat), c(theta_e), c(theta_r), c(thickness), n_columns=n_columns,
                                    dt_h=dt_h, num_subcycles=num_subcycles, initial_psi=initial_psi,
                                    ponded_depth_max=ponded_depth_max, wilting_point_psi=wilting_point_psi,
                                    frozen_factor=frozen_factor, nint=nint, giuh_ordinates=tuple(giuh_ordinates), dtype=npdt,
                                    iter_cap=iter_cap, search_mode=search_mode, bottom_mode=bottom_mode,
                                    use_closed_form_G=use_closed_form_G, front_slots=front_slots,
                                    geff_mode=1 if geff_precision == "f32" else 0)
        s = self.sim
        self.device, self.dtype = torch.device("cpu"), dtype
        self.L, self.N, self.dims, self.front_slots = s.L, s.N, s.dims, int(s.dims.front_slots)
        v = torch.from_numpy  # views: update_soil_parameters() writes through them
        self.alpha, self.n, self.ksat = v(s.alpha), v(s.n), v(s.ksat)
        self.theta_e, self.theta_r, self.thickness = v(s.theta_e), v(s.theta_r), v(s.thickness)
        self.status, self.n_fronts, self.scalars, self.totals = v(s.status), v(s.n_fronts), v(s.scalars), v(s.totals)
        self.depth, self.theta, self.psi, self.flags = v(s.depth), v(s.theta), v(s.psi), v(s.flags)

    def reset(self):
        self.sim.reset()

    def forward(self, precip, pet, series=("runoff", "percolation"), out=None, check=True, basin=(), weights=None,
                call_sums=False, forcing_group=1):
        a = lambda t: torch.as_tensor(t, dtype=torch.float64).cpu().numpy()
        precip, pet = a(precip), a(pet)
        if precip.ndim == 1:
            precip, pet = precip[None, :], pet[None, :]
        self.dims.ponded_depth_max = float(self.dims.ponded_depth_max)
        res = self.sim.forward(precip, pet, series=tuple(series), call_sums=call_sums, forcing_group=forcing_group)
        res = {k: torch.from_numpy(np.ascontiguousarray(r)) for k, r in res.items()}
        if check:
            self.check_status()
        return res

    def step_rows_host(self, precip_row, pet_row):
        a = lambda t: torch.as_tensor(t, dtype=torch.float64).reshape(1, -1)
        res = self.forward(a(precip_row), a(pet_row), series=("runoff", "percolation"), check=False, call_sums=True)
        return res["call_sums"], res["runoff"][0], res["percolation"][0], self.status.clone()

    def tangent(self, direction, precip, pet, w_runoff=None, w_perc=None, want_series=False, forcing_group=1, share=0):
        a = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float64).cpu().numpy()
        g, ser, st = self.sim.tangent({k: a(t) for k, t in direction.items()}, a(precip), a(pet), a(w_runoff), a(w_perc),
                                      want_series=want_series, forcing_group=forcing_group)
        return torch.from_numpy(g), None if ser is None else torch.from_numpy(ser), torch.from_numpy(st)

    def raise_for_status(self, status_host):
        if bool((status_host != 0).any()):
            self.check_status()

    def check_status(self):
        bad = int((self.status != 0).sum())
        if bad:
            raise LgarStatusError("%d of %d columns faulted" % (bad, self.N))

    def fronts(self):
        return self.sim.fronts()

    @property
    def ponded_water(self):
        return self.scalars[0]

    @property
    def previous_precip(self):
        return self.scalars[1]

    @property
    def ending_volume(self):
        return self.scalars[2]

    @property
    def giuh_runoff_queue(self):
        return self.scalars[3:3 + self.dims.n_giuh]


def install():
    """Put the simulator behind the model and the autograd tape (in THIS process)."""
    import lgar_py_amd.autograd as A
    import lgar_py_amd.model as M
    M.LgarEngine = SimLgarEngine
    A.LgarEngine = SimLgarEngine
