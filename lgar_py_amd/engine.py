"""Many-column LGAR engine: owns the struct-of-arrays device state and drives the HIP kernels
through the C-ABI (include/lgar.h).  Host-side mirror of what dpLGAR(nn.Module) keeps per column
(/root/reference/dpLGAR/models/dpLGAR.py:97-147), laid out column-fastest in HBM.

PyTorch is plumbing here (device memory, streams); the computation is in csrc/*.hip.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import ACC_NAMES, GMAX, LMAX, LMIN, NACC, NSCAL, LgarError

BASIN_SCRATCH_BYTES = 8 << 30  # most series memory, over ALL basin names, an engine allocates on its own behind basin sums


def _require_gpu(device):
    if not torch.cuda.is_available():
        raise LgarError("no ROCm GPU visible: the LGAR engine has no CPU fallback (device=%s)" % (device,))


class LgarStatusError(ValueError):
    """Physics fault in one or more columns (the reference raises ValueError / IndexError / AttributeError)."""


class LgarEngine:
    """N independent soil columns advanced by the gfx950 kernels.

    Parameters are [L] (shared by all columns) or [L, N] tensors/sequences; forcing is [T, N] (cm/h).
    search_mode: 1 (default) = bracketed-Newton psi search + closed-form jumps in the depth search (same roots, same
    tolerances); 0 = verification mode: the reference's literal fixed-step line searches (Layer.py:275-317, 681-701),
    its update_psi pass and, in fp64, its trapezoid operation by operation; 2 = like 1 with the front-capacity chain
    (8 -> 16 -> 32 slots, include/lgar.h) forced even for small jobs (tests).
    front_slots: rows of the per-front state arrays = the most fronts a column can hold (<= 32; the reference's lists are
    unbounded, Layer.py:1336-1416).
    geff_precision: "native" (default) = the Geff trapezoid (lgar/green_ampt.py:45-84) in `dtype`; "f32" (fp64 fast modes
    only) = mixed precision: fp64 column state, branches, mass bookkeeping, trapezoid heads and end nodes; the 119 interior
    nodes with the fp32 hardware transcendentals, summed in fp64 (LgarDims.geff_mode = 1; DESIGN.md section 4 states the
    tolerance this reaches against the reference).
    forward_lanes: lanes per column in lgar_forward.  0 (default) = the library gives fp64 trapezoid jobs under one wave per
    SIMD 4..64 cooperating lanes per column (same results bit for bit); 1 = never; 4..64 = exactly that many (a group needs
    four lanes for the trapezoid's end points, so 2 and 3 do not exist).  Honoured by the fp64 fast modes (native or
    mixed-precision trapezoid) with nint <= 128 only; an explicit request that cannot be honoured raises.
    basin_scratch_bytes: most series memory the engine may allocate on its own behind basin sums whose series the caller
    did not ask for (see forward); 0 = never, such sums are taken by the in-kernel atomics instead.
    bottom_mode: 0 (default) = like the reference, a front reaching the domain bottom faults the column; 1 = it leaves
    the column as percolation (LGAR-C intent; parity unpinned, the reference crashes there).
    """

    def __init__(self, alpha, n, ksat, theta_e, theta_r, thickness, *, n_columns=None, dt_h=1.0, num_subcycles=1,
                 initial_psi=2000.0, ponded_depth_max=0.0, wilting_point_psi=15495.0, frozen_factor=1.0, nint=120,
                 giuh_ordinates=(0.06, 0.51, 0.28, 0.12, 0.03), dtype=torch.float64, device="cuda:0",
                 iter_cap=0, search_mode=1, bottom_mode=0, use_closed_form_G=False, front_slots=None, with_state=True,
                 geff_precision="native", forward_lanes=0, basin_scratch_bytes=BASIN_SCRATCH_BYTES):
        self.device = torch.device(device)
        _require_gpu(self.device)
        self.lib = _capi.load()
        if dtype not in (torch.float32, torch.float64):
            raise LgarError("dtype must be torch.float32 or torch.float64")
        self.dtype = dtype
        self._dt = _capi.F64 if dtype == torch.float64 else _capi.F32

        def prep(x):
            t = torch.as_tensor(x, dtype=torch.float64)
            if t.dim() == 1:
                if n_columns is None:
                    raise LgarError("n_columns is required when parameters are per-layer vectors")
                t = t[:, None].expand(-1, n_columns)
            return t.to(self.device, dtype).contiguous()

        self.alpha, self.n, self.ksat = prep(alpha), prep(n), prep(ksat)
        self.theta_e, self.theta_r, self.thickness = prep(theta_e), prep(theta_r), prep(thickness)
        L, N = self.alpha.shape
        if not LMIN <= L <= LMAX:
            raise LgarError("this build supports %d..%d soil layers (got %d)" % (LMIN, LMAX, L))
        for t in (self.n, self.ksat, self.theta_e, self.theta_r, self.thickness):
            if tuple(t.shape) != (L, N):
                raise LgarError("parameter shapes differ: expected %s, got %s" % ((L, N), tuple(t.shape)))
        if len(giuh_ordinates) > GMAX:
            raise LgarError("at most %d GIUH ordinates" % GMAX)
        # physical sanity of the soil table (the reference would run into NaNs / negative pow bases much later)
        bad = [nm for nm, ok in (("alpha > 0", self.alpha > 0), ("n > 1", self.n > 1), ("ksat > 0", self.ksat > 0),
                                 ("theta_e > theta_r", self.theta_e > self.theta_r), ("theta_r >= 0", self.theta_r >= 0),
                                 ("thickness > 0", self.thickness > 0)) if not bool(ok.all())]
        if bad:
            raise LgarError("invalid soil parameters: need " + ", ".join(bad))
        if not (float(dt_h) > 0 and int(num_subcycles) >= 1 and int(nint) >= 1 and float(initial_psi) > 0):
            raise LgarError("need dt_h > 0, num_subcycles >= 1, nint >= 1, initial_psi > 0")
        self.L, self.N = L, N
        self.dims = _capi.LgarDims()
        d = self.dims
        d.n_columns, d.n_layers, d.n_steps, d.num_subcycles = N, L, 0, int(num_subcycles)
        d.nint, d.n_giuh, d.search_mode = int(nint), len(giuh_ordinates), int(search_mode)
        d.dt_h, d.initial_psi, d.ponded_depth_max = float(dt_h), float(initial_psi), float(ponded_depth_max)
        d.wilting_point_psi, d.frozen_factor = float(wilting_point_psi), float(frozen_factor)
        for i, g in enumerate(giuh_ordinates):
            d.giuh[i] = float(g)
        d.iter_cap = int(iter_cap)
        d.bottom_mode = int(bottom_mode)
        d.use_closed_form_G = int(bool(use_closed_form_G))
        if geff_precision not in ("native", "f32"):
            raise LgarError("geff_precision must be 'native' or 'f32'")
        if geff_precision == "f32" and (dtype != torch.float64 or int(search_mode) == 0):
            raise LgarError("geff_precision='f32' is the mixed mode of the fp64 fast searches (dtype float64, search_mode 1 or 2)")
        d.geff_mode = 1 if geff_precision == "f32" else 0
        self.geff_precision = geff_precision
        forward_lanes = int(forward_lanes)
        if forward_lanes not in (0, 1) and not 4 <= forward_lanes <= 64:
            raise LgarError("forward_lanes must be 0 (library's choice), 1, or 4..64 (got %d)" % forward_lanes)
        if forward_lanes > 1 and (dtype != torch.float64 or int(search_mode) == 0 or use_closed_form_G or int(nint) > 128):
            raise LgarError("forward_lanes=%d cannot be honoured: cooperating lanes exist for the fp64 fast modes (native or "
                            "mixed-precision trapezoid; no closed-form G, nint <= 128) only" % forward_lanes)
        d.forward_lanes = forward_lanes
        self.basin_scratch_bytes = int(basin_scratch_bytes)
        FMAX = int(front_slots) if front_slots else _capi.FMAX
        if not L + 1 <= FMAX <= _capi.FMAX:
            raise LgarError("front_slots must be in %d..%d" % (L + 1, _capi.FMAX))
        d.front_slots = FMAX
        self.front_slots = FMAX

        self.status = torch.zeros(N, dtype=torch.int32, device=self.device)
        if not with_state:  # tangent-only engine (autograd.parameter_vjp): the tangent kernels keep no state in HBM
            self._params = _capi.LgarParams(*[t.data_ptr() for t in (self.alpha, self.n, self.ksat, self.theta_e,
                                                                     self.theta_r, self.thickness)])
            self._state = None
            return
        z = lambda *shape, dt=dtype: torch.zeros(*shape, dtype=dt, device=self.device)
        self.depth, self.theta, self.psi = z(FMAX, N), z(FMAX, N), z(FMAX, N)
        self.k, self.dzdt = z(FMAX, N), z(FMAX, N)
        self.flags = z(FMAX, N, dt=torch.uint8)
        self.n_fronts = z(N, dt=torch.int32)
        self.scalars = z(NSCAL, N)
        self.totals = z(NACC, N)
        self.counters = z(_capi.NCOUNTERS, dt=torch.int64)
        self._basin_scratch = (0, {})  # series buffers behind basin sums whose series the caller did not ask for
        self.tickets = z(_capi.NTICKETS, dt=torch.int32)  # work counters of the persistent-wave schedule

        self._params = _capi.LgarParams(*[t.data_ptr() for t in (self.alpha, self.n, self.ksat, self.theta_e,
                                                                 self.theta_r, self.thickness)])
        self._state = _capi.LgarState(*[t.data_ptr() for t in (self.depth, self.theta, self.psi, self.k, self.dzdt,
                                                               self.flags, self.n_fronts, self.scalars, self.totals,
                                                               self.tickets)])
        self.reset()

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self):
        """dpLGAR.set_internal_states() for every column."""
        with torch.cuda.device(self.device):
            rc = self.lib.lgar_state_init(C.byref(self.dims), C.byref(self._params), C.byref(self._state),
                                          self.status.data_ptr(), self._dt, self._stream())
        _capi.check(rc, "lgar_state_init")

    def release_scratch(self):
        """Free the series buffers forward() keeps behind basin sums (they come back on the next such call)."""
        self._basin_scratch = (0, {})

    def _set_forcing_layout(self, precip, pet, forcing_group):
        g = max(1, int(forcing_group))
        if (precip.shape != pet.shape or precip.dim() != 2 or precip.shape[1] < 1 or self.N % g != 0
                or (self.N // g) % precip.shape[1] != 0):
            raise LgarError("forcing must be [T, %d] (or [T, Nf] with forcing_group * Nf dividing it: column c reads forcing "
                            "column (c // forcing_group) %% Nf); got %s / %s, forcing_group %d"
                            % (self.N, tuple(precip.shape), tuple(pet.shape), g))
        self.dims.forcing_columns = precip.shape[1]
        self.dims.forcing_group = g

    def forward(self, precip, pet, series=("runoff", "percolation"), out=None, check=True, basin=(), weights=None,
                call_sums=False, forcing_group=1):
        """Advance every column by T forcing steps.  precip/pet: [T, N] cm/h on self.device, or [T, Nf] with Nf dividing N
        (broadcast: column c reads forcing column c % Nf; Nf = 1 is one basin series for every column); with forcing_group
        G > 1, G consecutive columns share a forcing column: column c reads (c // G) % Nf.

        Returns {name: tensor[T, N]} for the requested per-step series (the model accumulators as they
        stand after each forward(), before MassBalance.change_mass zeroes them).  basin: names whose per-step sum over
        this engine's columns (optionally weighted by weights[N]) is reduced on the device; returned under
        "basin:<name>" as fp64 [T] tensors.  MEMORY: a basin sum is taken from the stored series (one deterministic pass, ~10x
        cheaper than in-kernel atomics), so a basin name that is not also in `series` gets a [T, N] scratch series of its own,
        kept for the next call of the same T -- as long as all such buffers together stay within the engine's
        basin_scratch_bytes (default 8 GiB; 0 = never); past that the sum falls back to the in-kernel atomics and nothing is
        allocated.  release_scratch() frees them.  call_sums=True adds "call_sums": [NACC, N], the accumulators summed over this
        call's steps (rows 8, 9: latest ponded_water / ending_volume)."""
        if self._state is None:
            raise LgarError("this engine was created with with_state=False (tangent launches only)")
        precip = torch.as_tensor(precip).to(self.device, self.dtype).contiguous()
        pet = torch.as_tensor(pet).to(self.device, self.dtype).contiguous()
        if precip.dim() == 1:
            precip, pet = precip[None, :], pet[None, :]
        self._set_forcing_layout(precip, pet, forcing_group)
        T = precip.shape[0]
        res = {}
        so = _capi.LgarStepOut()
        for nm in series:
            j = ACC_NAMES.index(nm)
            buf = out[nm] if out is not None and nm in out else torch.empty(T, self.N, dtype=self.dtype, device=self.device)
            if tuple(buf.shape) != (T, self.N) or buf.dtype != self.dtype or not buf.is_contiguous():
                raise LgarError("bad output buffer for series %r" % nm)
            res[nm] = buf
            so.series[j] = buf.data_ptr()
        w = None
        if basin:
            block = torch.zeros(NACC, T, dtype=torch.float64, device=self.device)
            so.basin = block.data_ptr()
            for nm in basin:
                j = ACC_NAMES.index(nm)
                so.basin_mask |= 1 << j
                res["basin:" + nm] = block[j]
                # a basin sum is taken from the stored series in one deterministic pass after the launch (include/lgar.h:
                # LgarStepOut.basin); the in-kernel atomics are ~10x as expensive, so a name whose series the caller does
                # not want still gets a scratch series (kept for the next call of the same shape) unless it would be huge
                if not so.series[j]:
                    if self._basin_scratch[0] != T:  # buffers of one call shape at a time
                        self._basin_scratch = (T, {})
                    scratch = self._basin_scratch[1].get(nm)
                    one = T * self.N * self.totals.element_size()
                    if scratch is None and (len(self._basin_scratch[1]) + 1) * one <= self.basin_scratch_bytes:
                        scratch = self._basin_scratch[1][nm] = torch.empty(T, self.N, dtype=self.dtype, device=self.device)
                    if scratch is not None:
                        so.series[j] = scratch.data_ptr()
            if weights is not None:
                w = torch.as_tensor(weights).to(self.device, self.dtype).contiguous()
                if tuple(w.shape) != (self.N,):
                    raise LgarError("weights must be [N]")
                so.weights = w.data_ptr()
        so.counters = self.counters.data_ptr()
        if call_sums:
            res["call_sums"] = torch.zeros(NACC, self.N, dtype=self.dtype, device=self.device)
            so.call_sums = res["call_sums"].data_ptr()
        self.dims.n_steps = T
        fo = _capi.LgarForcing(precip.data_ptr(), pet.data_ptr())
        with torch.cuda.device(self.device):
            rc = self.lib.lgar_forward(C.byref(self.dims), C.byref(self._params), C.byref(self._state), C.byref(fo),
                                       C.byref(so), self.status.data_ptr(), self._dt, self._stream())
        _capi.check(rc, "lgar_forward")
        if check:
            self.check_status()
        return res

    def step_rows_host(self, precip_row, pet_row):
        """The drop-in's calling convention -- ONE forcing row per call, results wanted on the host (the reference keeps its
        accumulators in host tensors and reads them after every forward(), physics/MassBalance.py:31-53) -- with the host-side
        cost of a call cut to the bone: persistent pinned staging buffers, argument structs built once, one upload, the kernel
        launches of lgar_forward, two downloads and ONE stream synchronisation; no allocation, no other torch op.

        precip_row / pet_row: length-N sequences or tensors (cm/h).  Returns (call_sums [NACC, N], runoff [N], percolation [N],
        status [N]) as views of the pinned host buffers, valid until the next call."""
        if self._state is None:
            raise LgarError("this engine was created with with_state=False (tangent launches only)")
        st = getattr(self, "_row_stepper", None)
        if st is None:
            N, dev = self.N, self.device
            st = self._row_stepper = {}
            st["h_in"] = torch.zeros(2, 1, N, dtype=self.dtype).pin_memory()
            st["d_in"] = torch.zeros(2, 1, N, dtype=self.dtype, device=dev)
            st["d_out"] = torch.zeros(NACC + 2, N, dtype=self.dtype, device=dev)  # call_sums rows, then runoff, percolation
            st["h_out"] = torch.zeros(NACC + 2, N, dtype=self.dtype).pin_memory()
            st["h_status"] = torch.zeros(N, dtype=torch.int32).pin_memory()
            so = _capi.LgarStepOut()
            so.series[ACC_NAMES.index("runoff")] = st["d_out"][NACC].data_ptr()
            so.series[ACC_NAMES.index("percolation")] = st["d_out"][NACC + 1].data_ptr()
            so.call_sums = st["d_out"].data_ptr()
            so.counters = self.counters.data_ptr()
            st["so"] = so
            st["fo"] = _capi.LgarForcing(st["d_in"][0].data_ptr(), st["d_in"][1].data_ptr())
        h_in = st["h_in"]
        h_in[0, 0] = torch.as_tensor(precip_row, dtype=self.dtype)
        h_in[1, 0] = torch.as_tensor(pet_row, dtype=self.dtype)
        d = self.dims
        d.n_steps, d.forcing_columns, d.forcing_group = 1, self.N, 1
        with torch.cuda.device(self.device):
            st["d_in"].copy_(h_in, non_blocking=True)
            rc = self.lib.lgar_forward(C.byref(d), C.byref(self._params), C.byref(self._state), C.byref(st["fo"]),
                                       C.byref(st["so"]), self.status.data_ptr(), self._dt, self._stream())
            _capi.check(rc, "lgar_forward")
            st["h_out"].copy_(st["d_out"], non_blocking=True)
            st["h_status"].copy_(self.status, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        return st["h_out"][:NACC], st["h_out"][NACC], st["h_out"][NACC + 1], st["h_status"]

    def raise_for_status(self, status_host):
        """check_status() on a host copy of the status words (no device round trip)."""
        if bool((status_host != 0).any()):
            self.check_status()

    def tangent(self, direction, precip, pet, w_runoff=None, w_perc=None, want_series=False, forcing_group=1, share=0):
        """Forward-mode tangent from a FRESH state (set_internal_states) over the whole forcing series.

        direction: {"alpha" | "n" | "ksat": [L, N] tensor} -- the parameter perturbation (missing = 0).
        precip / pet / w_runoff / w_perc: [T, N], or all [T, Nf] with forcing_group * Nf dividing N (column c uses column
        (c // forcing_group) % Nf of each).
        share = W (2..32): each group of W consecutive columns is ONE soil column along W directions; the W lanes share the
        Geff trapezoid (LgarDims.tangent_share).
        Returns (grad[N], tangent_runoff[T, N] or None, status[N]) with
        grad[c] = sum_t w_runoff[t, c] * d runoff_t[c] + w_perc[t, c] * d percolation_t[c].  status != 0 marks columns whose
        tangent integration faulted (their grad entry is not a gradient): callers must check it (autograd.parameter_vjp does)."""
        prep = lambda t: None if t is None else torch.as_tensor(t).to(self.device, self.dtype).contiguous()
        precip, pet, w_runoff, w_perc = prep(precip), prep(pet), prep(w_runoff), prep(w_perc)
        self._set_forcing_layout(precip, pet, forcing_group)
        share = int(share)
        if share != 0 and (not 2 <= share <= 32 or self.N % share != 0):
            raise LgarError("share must be 0 or 2..32 (with n_columns a multiple of it)")
        if share:
            # the kernel takes the caller's word that each group of `share` columns is one soil column; a violation would
            # give silently wrong gradients, so the wrapper checks (six small reductions)
            for t in (self.alpha, self.n, self.ksat, self.theta_e, self.theta_r, self.thickness):
                gw = t.reshape(t.shape[0], -1, share)
                if not bool((gw == gw[:, :, :1]).all()):
                    raise LgarError("share=%d needs identical soil parameters within each group of %d columns" % (share, share))
            if forcing_group % share != 0 and precip.shape[1] != 1:
                raise LgarError("share=%d needs the columns of a group to read the same forcing column (forcing_group a "
                                "multiple of it, or one forcing column for all)" % share)
        self.dims.tangent_share = int(share)
        for nm, w in (("w_runoff", w_runoff), ("w_perc", w_perc)):
            if w is not None and w.shape != precip.shape:
                raise LgarError("%s must be [T, N] like the forcing; got %s" % (nm, tuple(w.shape)))
        T = precip.shape[0]
        dirs = {k: prep(direction.get(k)) for k in ("alpha", "n", "ksat")}
        for k, v in dirs.items():
            if v is not None and tuple(v.shape) != (self.L, self.N):
                raise LgarError("direction[%r] must be [L, N]" % k)
        ptr = lambda t: None if t is None else t.data_ptr()
        dstruct = _capi.LgarParams(ptr(dirs["alpha"]), ptr(dirs["n"]), ptr(dirs["ksat"]), None, None, None)
        grad = torch.zeros(self.N, dtype=self.dtype, device=self.device)
        ser = torch.empty(T, self.N, dtype=self.dtype, device=self.device) if want_series else None
        st = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        tickets = torch.zeros(_capi.NTICKETS, dtype=torch.int32, device=self.device)  # persistent-wave work counters
        self.dims.n_steps = T
        fo = _capi.LgarForcing(precip.data_ptr(), pet.data_ptr())
        with torch.cuda.device(self.device):
            rc = self.lib.lgar_forward_tangent(C.byref(self.dims), C.byref(self._params), C.byref(dstruct), C.byref(fo),
                                               ptr(w_runoff), ptr(w_perc), grad.data_ptr(), ptr(ser), st.data_ptr(),
                                               self._dt, self._stream(), tickets.data_ptr())
        _capi.check(rc, "lgar_forward_tangent")
        return grad, ser, st

    def cooperating_lanes(self):
        """Lanes per column lgar_forward uses for this engine's job (1, or 4..64 for fp64 jobs under one wave per SIMD)."""
        return int(self.lib.lgar_cooperating_lanes(C.byref(self.dims), self._dt))

    def geff_wave_calls(self, reset=True):
        """Wave-level Geff evaluations since the last reset (measurement: LgarStepOut.counters[0])."""
        n = int(self.counters[0].item())
        if reset:
            self.counters.zero_()
        return n

    def check_status(self):
        """Raise like the reference does (ValueError) if any column hit a physics fault."""
        if not bool((self.status != 0).any()):
            return
        bad = int((self.status != 0).sum().item())
        if bad:
            bits = int(torch.bitwise_or(self.status, torch.zeros_like(self.status)).max().item())
            allbits = 0
            for b in _capi.STATUS_NAMES:
                if int(((self.status & b) != 0).sum().item()):
                    allbits |= b
            names = [v for b, v in _capi.STATUS_NAMES.items() if allbits & b]
            first = int(torch.nonzero(self.status)[0].item())
            raise LgarStatusError("%d of %d columns faulted (%s); first column %d, max status %d"
                                  % (bad, self.N, ", ".join(names), first, bits))

    # ------------------------------------------------------------------------------------------
    def fronts(self):
        """Front tables as host numpy arrays: depth/theta/psi/k/dzdt [front_slots, N], layer, to_bottom, n_fronts (rows at and
        beyond a column's n_fronts are not meaningful)."""
        fl = self.flags.cpu().numpy()
        return dict(depth=self.depth.cpu().numpy(), theta=self.theta.cpu().numpy(), psi=self.psi.cpu().numpy(),
                    k=self.k.cpu().numpy(), dzdt=self.dzdt.cpu().numpy(), layer=(fl & 0x7F).astype("int8"),
                    to_bottom=(fl >> 7).astype("int8"), n_fronts=self.n_fronts.cpu().numpy())

    def total(self, name):
        return self.totals[ACC_NAMES.index(name)]

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


def leaf_batch(op, x, y=None, z=0.0, *, alpha, n, ksat, theta_e, theta_r, nint=120, wilting_point_psi=15495.0,
               dtype=torch.float64, device="cuda:0"):
    """Element-wise leaf kernels (known-answer tests): see lgar_leaf_batch in include/lgar.h."""
    dev = torch.device(device)
    _require_gpu(dev)
    lib = _capi.load()
    ops = {"theta_from_h": 0, "se_from_h": 1, "k_from_se": 2, "h_from_se": 3, "geff": 4, "aet": 5, "geff_literal": 6,
           "log2": 7, "exp2": 8, "pow": 9, "geff_mixed": 10, "div": 11, "pow_pairwise": 12, "log2_pairwise": 13, "exp2_pairwise": 14}
    prep = lambda t: None if t is None else torch.as_tensor(t, dtype=torch.float64).to(dev, dtype).contiguous()
    x, y, alpha, n, ksat, theta_e, theta_r = map(prep, (x, y, alpha, n, ksat, theta_e, theta_r))
    out = torch.empty_like(x)
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.lgar_leaf_batch(ops[op], x.numel(), ptr(x), ptr(y), float(z), ptr(alpha), ptr(n), ptr(ksat),
                                 ptr(theta_e), ptr(theta_r), int(nint), float(wilting_point_psi), ptr(out),
                                 _capi.F64 if dtype == torch.float64 else _capi.F32,
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _capi.check(rc, "lgar_leaf_batch")
    return out
