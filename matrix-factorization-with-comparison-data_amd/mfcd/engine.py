"""Device-side driver of the hot path: owns nothing but a workspace; reads/writes the caller's
model parameters and the caller's torch.optim.Adam state in place (the optimizer
object stays valid after a fused run).

All compute goes through the C-ABI in libmfcd_hip.so; there is no eager/torch fallback.
"""
import numpy as np
import torch

from . import _lib
from ._lib import _raw_stream
from .batching import dataset_records, epoch_order, n_batches, pack_records


def _require_cuda_param(t, name, dtypes=(torch.float32,)):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MfcdError(f"{name} must live on a GPU device: the triplet hot path is MI355X-only "
                             "(pass device='cuda'; there is no CPU fallback)")
    if t.dtype not in dtypes or not t.is_contiguous():
        raise _lib.MfcdError(f"{name} must be a contiguous tensor of dtype {' or '.join(str(x) for x in dtypes)}")


_FACTOR_DTYPES = (torch.float32, torch.bfloat16)


class AdamBinding:
    """View of a caller-owned torch.optim.Adam over exactly (model.U, model.V)."""

    def __init__(self, model, optimizer):
        if type(optimizer) is not torch.optim.Adam:
            raise NotImplementedError("the fused step implements torch.optim.Adam only "
                                      f"(got {type(optimizer).__name__})")
        if len(optimizer.param_groups) != 1:
            raise NotImplementedError("the fused step needs a single Adam param group")
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("decoupled_weight_decay"):
            raise NotImplementedError("amsgrad / maximize / differentiable / decoupled_weight_decay are not fused")
        if isinstance(g["lr"], torch.Tensor) or any(isinstance(b, torch.Tensor) for b in g["betas"]):
            raise NotImplementedError("tensor-valued lr / betas are not fused")
        params = [p for p in g["params"]]
        if len(params) != 2 or not any(p is model.U for p in params) or not any(p is model.V for p in params):
            raise NotImplementedError("the optimizer must hold exactly model.U and model.V")
        self.model, self.opt, self.group = model, optimizer, g
        _require_cuda_param(model.U.data, "model.U", _FACTOR_DTYPES)
        _require_cuda_param(model.V.data, "model.V", _FACTOR_DTYPES)
        if model.U.shape[1] != model.V.shape[1]:
            raise ValueError("U and V must share the latent dimension")
        if model.U.dtype != model.V.dtype:
            raise ValueError("U and V must share the storage dtype")
        for p in (model.U, model.V):  # lazily created as Adam._init_group does — except that the moments are
            st = optimizer.state[p]   # always fp32 (bf16 factors keep fp32 exp_avg / exp_avg_sq: SURVEY section 7(8))
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.get_default_dtype())
                st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.preserve_format)
            _require_cuda_param(st["exp_avg"], "exp_avg")
            _require_cuda_param(st["exp_avg_sq"], "exp_avg_sq")
        su, sv = optimizer.state[model.U]["step"], optimizer.state[model.V]["step"]
        if float(su) != float(sv):
            raise NotImplementedError("U and V must have taken the same number of Adam steps")
        self._pending, self._base = 0, None
        self._fast, self._fast_stream = None, None     # train_steps' per-binding fast path (engine._FastCall)

    @property
    def step(self):
        """Adam steps taken so far: the optimizer's `step` tensor plus the steps counted but not yet written back."""
        if self._base is None:
            self._base = int(float(self.opt.state[self.model.U]["step"]))
        return self._base + self._pending

    def advance(self, k, defer=False):
        """Count k more optimiser steps.  defer=True (epoch loops): only a Python counter moves and `flush()` writes
        the optimizer's `step` tensors once at the end (two CPU tensor updates per call are ~4 us of host time)."""
        self._pending += k
        if not defer:
            self.flush()

    def flush(self):
        if self._pending:
            for p in (self.model.U, self.model.V):
                self.opt.state[p]["step"] += self._pending
        self._pending, self._base = 0, None

    def hyper(self):
        g = self.group
        return float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"])

    def tensors(self):
        U, V = self.model.U.data, self.model.V.data
        su, sv = self.opt.state[self.model.U], self.opt.state[self.model.V]
        return U, V, su["exp_avg"], su["exp_avg_sq"], sv["exp_avg"], sv["exp_avg_sq"]

    def call_context(self):
        """(six raw table pointers, n, m, d, device, dtype) for the C-ABI, computed once per binding: the binding is
        a view of tensors that stay where they are for its lifetime (`refresh()` after replacing any of them)."""
        ctx = getattr(self, "_ctx", None)
        if ctx is None:
            t = self.tensors()
            U, V = t[0], t[1]
            ctx = self._ctx = (tuple(_lib.ptr(x) for x in t), U.shape[0], V.shape[0], U.shape[1], U.device, U.dtype)
        return ctx

    def refresh(self):
        self._ctx = None
        self._fast = None
        self._big = None
        self.drop_prepared()

    # ---- prepared calls (include/mfcd.h: mfcd_train_call_*) ----
    def prepared(self, ws, batch_size):
        """Handle of a prepared call for this binding on planned workspace `ws` (a Workspace) and batch size, made on
        first use and re-made when anything it binds has changed: the workspace buffer (re-plan), a hyper-parameter
        (learning-rate schedules write param_groups between calls) or the tables (refresh())."""
        g = self.group
        key = (ws.buf.data_ptr(), batch_size, g["lr"], g["betas"], g["eps"], g["weight_decay"])
        hit = getattr(self, "_prep", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        import ctypes
        self.drop_prepared()
        L = _lib.load()
        ptrs, n, m, d, dev, dtype = self.call_context()
        lr, b1, b2, eps, wd = self.hyper()
        h = ctypes.c_void_p()
        _lib.check(L.mfcd_train_call_prepare(*ptrs, int(dtype == torch.bfloat16), batch_size, n, m, d, lr, b1, b2, eps,
                                             wd, ws.buf.data_ptr(), ws.buf.numel(), ctypes.byref(h)))
        self._prep = (key, h, ws.buf)          # (the buffer is kept alive as long as the handle names it)
        return h

    def drop_prepared(self):
        hit = getattr(self, "_prep", None)
        self._fast = None
        if hit is not None:
            self._prep = None
            _lib.load().mfcd_train_call_release(hit[1])

    def __del__(self):
        try:
            self.drop_prepared()
        except Exception:
            pass


import weakref

_big_forms = weakref.WeakSet()          # live BigResident objects: check_status() reads their status words too
_BIG_MODE = "auto"                      # "auto": train_steps takes the big resident form where it applies; "off"; "force"
BIG_MIN_STEPS = 256                     # shorter calls stay with the streaming form (state load / write-back, mailbox clear)


def set_big_resident(mode):
    """"auto" (default): `train_steps` uses the big resident form (BigResident) for fp32 models with d == 64 whose state
    is too large for the regular resident form but fits one GPU's registers (n + m <= 131072: BASELINE configs[3]),
    for calls of at least BIG_MIN_STEPS steps of at most 64 samples whose stream passes the form's pre-check;
    "off": never; "force": whenever the shape allows (tests)."""
    global _BIG_MODE
    if mode not in ("auto", "off", "force"):
        raise ValueError(mode)
    _BIG_MODE = mode


def _big_for(binding, N, batch_size, samples_dev):
    """The binding's BigResident if this call should take it, else None."""
    if _BIG_MODE == "off" or batch_size > 64:
        return None
    big = getattr(binding, "_big", None)
    if big is None:
        U, V = binding.model.U.data, binding.model.V.data
        ok = U.dtype == torch.float32 and U.shape[1] == 64 and U.shape[0] + V.shape[0] <= 131072
        if ok and _BIG_MODE == "auto":      # only where the library itself would stream (state beyond the resident form)
            ok = train_plan(max(N, BIG_MIN_STEPS * batch_size), batch_size, U.shape[0], V.shape[0], 64)["form_name"] == "streaming"
        big = binding._big = BigResident(binding) if ok else False
    if big is False:
        return None
    nsteps = (N + batch_size - 1) // batch_size
    if _BIG_MODE == "auto" and (nsteps < BIG_MIN_STEPS or TRAIN_PATH_NOW[0] != "auto"):
        return None
    return big if big.takes(samples_dev, batch_size) else None


class BigResident:
    """Opt-in step form for states of up to 8 388 608 elements at d = 64 (include/mfcd.h: mfcd_train_steps_big; BASELINE
    configs[3] on ONE GPU): the Adam moments of the whole model in registers, the parameters in LDS, one persistent launch
    per call.  Same results as the streaming form, bit for bit."""

    def __init__(self, binding):
        self.b, self.L = binding, _lib.load()
        U, V = binding.model.U.data, binding.model.V.data
        if U.dtype != torch.float32 or U.shape[1] != 64 or U.shape[0] + V.shape[0] > 131072:
            raise NotImplementedError("the big resident form takes fp32 tables with d == 64 and n + m <= 131072")
        self.dev, self.ws = U.device, None
        _big_forms.add(self)

    def train_steps(self, stream, B, loss_out=None, defer_step=False):
        U, V, mU, vU, mV, vV = self.b.tensors()
        N = stream.shape[0]
        nsteps = (N + B - 1) // B
        if loss_out is None:
            loss_out = torch.empty(max(nsteps, 1), dtype=torch.float32, device=self.dev)
        need = self.L.mfcd_train_big_workspace_bytes(N, B)
        if need == 0:
            raise _lib.MfcdError("the big resident form takes batches of at most 64 samples")
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need), dtype=torch.uint8, device=self.dev)
        lr, b1, b2, eps, wd = self.b.hyper()
        _lib.check(self.L.mfcd_train_steps_big(_lib.ptr(U), _lib.ptr(V), _lib.ptr(mU), _lib.ptr(vU), _lib.ptr(mV),
                                               _lib.ptr(vV), _lib.ptr(stream), N, B, self.b.step, U.shape[0], V.shape[0],
                                               64, lr, b1, b2, eps, wd, _lib.ptr(loss_out), _lib.ptr(self.ws),
                                               self.ws.numel(), _lib.stream_ptr(self.dev)))
        self.b.advance(nsteps, defer_step)
        return loss_out[:nsteps]

    def takes(self, stream, B):
        """Pre-check of a record stream (mfcd_train_big_check; one small kernel and a host read): no batch sends more
        row references into one wave's slice than the kernel has gradient slots for."""
        N = stream.shape[0]
        if N == 0 or B > 64:
            return False
        worst = torch.zeros(1, dtype=torch.int32, device=self.dev)
        U, V = self.b.model.U.data, self.b.model.V.data
        _lib.check(self.L.mfcd_train_big_check(_lib.ptr(stream), N, B, U.shape[0], V.shape[0], _lib.ptr(worst),
                                               _lib.stream_ptr(self.dev)))
        return int(worst.item()) <= self.L.mfcd_train_big_slots()

    def status(self):
        import ctypes
        if self.ws is None:
            return
        out = ctypes.c_int(0)
        _lib.check(self.L.mfcd_train_big_status(_lib.ptr(self.ws), ctypes.byref(out), _lib.stream_ptr(self.dev)))
        if out.value:
            raise _lib.MfcdError({1: "big resident form: a bounded wait expired",
                                  2: "big resident form: a batch named more rows of one wave's slice than it has gradient slots for"}.get(
                                      out.value, f"big resident form: status {out.value}"))


def generate_labels(triplets, X, scale=1.0, K=1, soft=False, seed=0, device=None):
    """BTL labels drawn ON the device (include/mfcd.h: mfcd_generate_labels) → int32 [N, 4] device tensor
    of mfcd_sample records (N = T*K hard-label rows, or T soft-label rows), ready for SampleStore / train_steps.
    `X` is a dense [n, m] fp32 tensor (moved to the device if needed) or a generation_data.FactoredMatrix, whose n x m
    product is never formed.  Reproducible for a seed; not the reference's CPU generator stream."""
    L = _lib.load()
    n, m = X.shape
    if device is None:
        device = X.device if isinstance(X, torch.Tensor) and X.is_cuda else torch.device("cuda")
    device = torch.device(device)
    if isinstance(triplets, torch.Tensor) and triplets.is_cuda:
        # triplets made on the device (mfcd.sampling): they never visit the host
        trip = triplets.to(device=device, dtype=torch.int32).reshape(-1, 3).contiguous()
        T = trip.shape[0]
        if T:
            lo, hi_u, hi_i = (int(v) for v in torch.stack((trip.min(), trip[:, 0].max(), trip[:, 1:].max())).tolist())
            if lo < 0 or hi_u >= n or hi_i >= m:
                raise IndexError("triplet index out of range for X")
    else:
        idx = np.ascontiguousarray(np.asarray(triplets, dtype=np.int64).reshape(-1, 3))
        T = idx.shape[0]
        if T and (idx.min() < 0 or idx[:, 0].max() >= n or idx[:, 1:].max() >= m):
            raise IndexError("triplet index out of range for X")          # as X[u, i] would (structure.py:509)
        trip = torch.from_numpy(idx.astype(np.int32)).to(device)
    out = torch.empty((T if soft else T * K, 4), dtype=torch.int32, device=device)
    if isinstance(X, torch.Tensor):
        Xd = X.detach().to(device=device, dtype=torch.float32).contiguous()
        args = (_lib.ptr(Xd), n, m, None, None, 0)
    else:                                                              # FactoredMatrix
        A = X.A.to(device).contiguous()
        B = X.B.to(device).contiguous()
        args = (None, n, m, _lib.ptr(A), _lib.ptr(B), A.shape[1])
    _lib.check(L.mfcd_generate_labels(_lib.ptr(trip) if T else None, T, *args, float(scale), int(K), int(bool(soft)),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(out) if out.numel() else None,
                                      _lib.stream_ptr(device)))
    return out


class SampleStore:
    """A dataset's (u,i,j,z) records resident in HBM as 16-byte mfcd_sample structs."""

    def __init__(self, rows, n, m, device):
        if isinstance(rows, torch.Tensor) and rows.is_cuda and rows.dtype == torch.int32:
            self.dev = rows.to(device).contiguous()         # records made on the device (generate_labels): no host copy
            self.N, self.host = self.dev.shape[0], None
            self.device = self.dev.device
            return
        rec = pack_records(rows, n, m)
        self.N = rec.shape[0]
        self.host = rec
        self.dev = torch.from_numpy(rec).to(device)  # int32 [N,4]
        self.device = self.dev.device

    @classmethod
    def from_loader(cls, loader, n, m, device):
        ds = loader.dataset
        fn = getattr(ds, "_mfcd_records", None)
        rows = fn() if callable(fn) else None
        key = (id(rows if rows is not None else getattr(ds, "data", ds)), len(ds), n, m, str(device))
        cached = getattr(ds, "_mfcd_store", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        dev_fn = getattr(ds, "_mfcd_device_records", None)
        dev_rec = dev_fn() if callable(dev_fn) else None
        store = cls(dev_rec if dev_rec is not None else dataset_records(ds), n, m, device)
        try:
            ds._mfcd_store = (key, store)
        except AttributeError:
            pass
        return store

    def ordered(self, order):
        """Records gathered in `order` (an int64 CPU tensor) — the per-epoch stream, built on device."""
        if order.numel() == self.N and bool((order[:1] == 0).all()) and order.numel() > 1 and \
                bool((order[1:] - order[:-1] == 1).all()):
            return self.dev
        # pinned staging buffer -> the upload is asynchronous and does not drain the stream (a pageable
        # H2D copy would block the host until every queued step kernel has finished)
        return self.dev.index_select(0, order.pin_memory().to(self.device, non_blocking=True))


class StreamPrefetch:
    """Builds an epoch's record stream (order upload + device gather) on a side stream, so that it runs UNDER the
    previous epoch's persistent step kernel instead of between two epochs (the kernel leaves wave slots free; the
    gather is ~25 us of the ~0.6 ms epoch at C2) — and, round 3, STAGES the next call's prologue there too
    (include/mfcd.h: mfcd_train_call_stage: sample translation and the per-wave event lists, ~14 us at C2).
    `take()` makes the consumer stream wait for it."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        # everything created so far (the sample stores) is visible to the side stream; it must NOT wait for the main
        # stream again later, or it would queue behind the very kernel it is meant to run under
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        self.pending = None

    def start(self, store, order, after=None, stage=None):
        """after: an event the side stream waits for first (the step-kernel launch TWO calls back: a staged prologue
        writes the set of workspace regions that launch read); stage(rec, side_stream): stages the call that will
        consume `rec` (stage_next_call below)."""
        with torch.cuda.stream(self.side):
            if after is not None:
                self.side.wait_event(after)
            rec = store.ordered(order)
            if stage is not None:
                stage(rec, self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.pending = (rec, ev)

    def take(self):
        rec, ev = self.pending
        self.pending = None
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        rec.record_stream(main)               # allocated on the side stream, consumed on the main one
        return rec


def stage_next_call(binding, samples_dev, batch_size, loss_out, side_stream):
    """Run the prologue of the NEXT train_steps(binding, samples_dev, batch_size, loss_out=loss_out) call now, on
    `side_stream` (include/mfcd.h: mfcd_train_call_stage).  Ordering is the caller's (StreamPrefetch does it): the side
    stream must already wait for every launch on this workspace OLDER than the most recent one (the staged prologue
    writes the set of regions the most recent launch is not using), and the stream of the consuming call must wait
    for the side stream.  A staged prologue whose call never comes costs nothing but its own time: the library clears
    what it left before that set is written again.  Returns False when there is nothing to stage (no prepared call
    yet, another form, another workspace plan)."""
    fc = binding._fast
    N = samples_dev.shape[0]
    if fc is None or not fc.still_valid(N, batch_size) or not samples_dev.is_contiguous():
        return False
    code = _lib.load().mfcd_train_call_stage(fc.handle, samples_dev.data_ptr(), N, binding.step, loss_out.data_ptr(),
                                             side_stream.cuda_stream)
    if code:
        _lib.check(code)
    return True


class Workspace:
    """One PLANNED training workspace (include/mfcd.h: mfcd_train_workspace_init): a device buffer sized for a
    capacity (most samples per call, batch size, table shape) plus the library-side state registered for it.
    It is re-planned only when a call does not fit; calls that fit re-initialise nothing."""

    def __init__(self):
        self.buf = None
        self.plan = None       # (N_cap, B, n, m, d)
        self.k_cap = 0
        self.retired = []      # replaced buffers whose status word has not been looked at yet

    def fits(self, N, B, n, m, d):
        if self.buf is None or self.plan[2:] != (n, m, d):
            return False
        return N <= self.plan[0] and n_batches(N, B) <= self.k_cap

    def ensure(self, N, B, n, m, d, device):
        if self.fits(N, B, n, m, d) and self.buf.device == device:
            return self.buf
        L = _lib.load()
        n_cap = max(int(N), 1)
        if self.plan is not None and self.plan[2:] == (n, m, d):
            n_cap = max(n_cap, self.plan[0])      # never shrink a plan for the same tables
        self.drop(keep_for_status=True)
        nbytes = int(L.mfcd_train_workspace_bytes(n_cap, B, n, m, d))
        self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _lib.check(L.mfcd_train_workspace_init(_lib.ptr(self.buf), self.buf.numel(), n_cap, B, n, m, d,
                                               _lib.stream_ptr(device)))
        self.plan, self.k_cap = (n_cap, B, n, m, d), n_batches(n_cap, B)
        return self.buf

    def drop(self, keep_for_status=False):
        if self.buf is not None:
            _lib.check(_lib.load().mfcd_train_workspace_release(_lib.ptr(self.buf)))
            if keep_for_status:
                self.retired.append(self.buf[:4].clone())   # stream-ordered copy of the word: the buffer itself is freed
        self.buf, self.plan, self.k_cap = None, None, 0

    def status(self):
        """Status words of this workspace and of the buffers it replaced since the last look.  Synchronises."""
        words = list(self.retired) + ([self.buf[:4]] if self.buf is not None else [])
        self.retired = []
        return [int(w.view(torch.int32).item()) for w in words]


_workspaces = {}   # (device index, stream handle, (n, m, d)) -> Workspace: one per device, stream and table shape, so
                   # that two models of different shapes trained alternately do not re-plan each other's workspace
_last_workspace = {}   # (device index, stream handle) -> the Workspace used last (diagnostics)
_MAX_WORKSPACES_PER_STREAM = 6


def workspace_for(device, shape=None):
    """The planned workspace of the current stream of `device` for tables of `shape` = (n, m, d); without a shape, the
    one used last on that stream."""
    dev = device if isinstance(device, torch.device) else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    base = (idx, _lib.stream_ptr(torch.device("cuda", idx)))
    if shape is None:
        ws = _last_workspace.get(base)
        if ws is not None:
            return ws
        shape = ()
    key = base + (tuple(shape),)
    ws = _workspaces.pop(key, None)
    if ws is None:
        ws = Workspace()
        mine = [k for k in _workspaces if k[:2] == base]
        for k in mine[:max(0, len(mine) - (_MAX_WORKSPACES_PER_STREAM - 1))]:   # least recently used first
            old = _workspaces.pop(k)
            old.drop(keep_for_status=True)
            ws.retired += old.retired          # their status words are still looked at by check_status()
    _workspaces[key] = ws                      # (re)inserted last = most recently used
    _last_workspace[base] = ws
    return ws


def reserve_workspace(n_samples, batch_size, n, m, d, device):
    """Plan the current stream's workspace for calls of up to `n_samples` samples (an epoch) BEFORE the first call, so
    that a short first call (warm-up, a partial epoch) does not size it too small and force a re-plan later."""
    return workspace_for(device, (n, m, d)).ensure(n_samples, batch_size, n, m, d, torch.device(device))


TRAIN_PATHS = {"auto": 0, "streaming": 1, "resident": 2, "local": 3}
FORM_NAMES = {1: "streaming", 2: "resident", 3: "local"}


TRAIN_PATH_NOW = ["auto"]


def set_train_path(mode):
    """Select the form of the fused step: "auto" (default), "streaming", "resident" or "local" (include/mfcd.h)."""
    _lib.check(_lib.load().mfcd_set_train_path(TRAIN_PATHS[mode]))
    TRAIN_PATH_NOW[0] = mode


def set_resident_math(flavour):
    """Adam arithmetic inside the resident form: "fast" (default) or "ieee" (include/mfcd.h)."""
    _lib.check(_lib.load().mfcd_set_resident_math({"ieee": 0, "fast": 1}[flavour]))


def set_tuning(**knobs):
    """Experiment / test knobs of include/mfcd.h (mfcd_set_tuning), e.g. set_tuning(resident_lookahead=0)."""
    L = _lib.load()
    for name, value in knobs.items():
        _lib.check(L.mfcd_set_tuning(_lib.TUNE_KEYS[name], int(value)))


def train_plan(n_samples, batch_size, n, m, d, bf16=False):
    """What the library would do for a call of these sizes under the current settings → dict (mfcd_train_plan)."""
    import ctypes
    out = _lib.TrainPlan()
    _lib.check(_lib.load().mfcd_train_plan_query(n_samples, batch_size, n, m, d, int(bool(bf16)), ctypes.byref(out)))
    plan = {k: int(getattr(out, k)) for k, _ in _lib.TrainPlan._fields_ if k != "reserved"}
    plan["form_name"] = FORM_NAMES[plan["form"]]
    return plan


def check_status():
    """Raise if ANY resident launch since the last check gave up on a bounded wait (the workspace status word is
    sticky: an abort in an early epoch is still there at the end of the run).  Synchronises.  A workspace that
    reported an abort is dropped, so the next call plans and zero-fills a fresh one."""
    bad = None
    for ws in list(_workspaces.values()):
        for code in ws.status():
            if code != 0:
                bad = code
                ws.drop()
    if bad is not None:
        raise _lib.MfcdError(f"resident training kernel aborted (status {bad}): a bounded in-kernel wait expired; "
                             "parameters are undefined")
    for big in list(_big_forms):
        big.status()


class _FastCall:
    """Everything train_steps() needs for its next call on the same binding / workspace / batch size / stream, looked up
    once: a prepared-call handle (include/mfcd.h: mfcd_train_call_*), the bound C entry, the workspace it names.  The
    per-call Python is then one validity check and one ctypes call with six scalars."""
    __slots__ = ("handle", "fn", "wso", "buf", "batch", "n_cap", "k_cap", "hyper", "group", "dev_index", "dev")

    def still_valid(self, N, batch_size):
        g = self.group
        return (batch_size == self.batch and 0 < N <= self.n_cap and self.wso.buf is self.buf and
                (g["lr"], g["betas"], g["eps"], g["weight_decay"]) == self.hyper)


def _make_fast_call(binding, wso, batch_size, dev):
    fc = _FastCall()
    g = binding.group
    fc.handle = binding.prepared(wso, batch_size)
    fc.fn = _lib.load().mfcd_train_call_run
    fc.wso, fc.buf, fc.batch = wso, wso.buf, batch_size
    fc.n_cap, fc.k_cap = wso.plan[0], wso.k_cap
    fc.hyper, fc.group = (g["lr"], g["betas"], g["eps"], g["weight_decay"]), g
    fc.dev, fc.dev_index = dev, dev.index
    return fc


def train_steps(binding, samples_dev, batch_size, loss_out=None, kernel_us=None, defer_step=False):
    """Run ceil(N/B) optimiser steps over `samples_dev` (int32 [N,4] device records, in order).
    Returns the fp32 device tensor of per-step batch-mean losses.  No host sync — unless `kernel_us`
    (a 3-element list) is given: then the diagnostic twin is used, which brackets every step launch
    with HIP events, waits, and fills kernel_us with [avg, min, max] microseconds (bench.py only)."""
    N = samples_dev.shape[0]
    if kernel_us is None:
        fc = binding._fast
        if fc is not None and fc.still_valid(N, batch_size) and samples_dev.is_contiguous():
            nsteps = (N + batch_size - 1) // batch_size
            # (a binding whose shape the big resident form takes leaves this fast path for calls long enough for it)
            if nsteps <= fc.k_cap and not (getattr(binding, "_big", False) and nsteps >= BIG_MIN_STEPS and _BIG_MODE != "off"):
                if loss_out is None:
                    loss_out = torch.empty(nsteps, dtype=torch.float32, device=fc.dev)
                stream = _raw_stream(fc.dev_index) if _raw_stream is not None else _lib.stream_ptr(fc.dev)
                if stream == binding._fast_stream:          # (a workspace belongs to one stream)
                    code = fc.fn(fc.handle, samples_dev.data_ptr(), N, binding.step, loss_out.data_ptr(), stream)
                    if code:
                        _lib.check(code)
                    binding._pending += nsteps
                    if not defer_step:
                        binding.flush()
                    return loss_out if loss_out.shape[0] == nsteps else loss_out[:nsteps]
    L = _lib.load()
    ptrs, n, m, d, dev, dtype = binding.call_context()
    nsteps = (N + batch_size - 1) // batch_size
    if N == 0:
        return torch.empty(0, dtype=torch.float32, device=dev) if loss_out is None else loss_out[:0]
    if getattr(binding, "_big", None) is None and d != 64:
        binding._big = False                                      # decided once per binding (refresh() clears it)
    if kernel_us is None and getattr(binding, "_big", None) is not False:
        big = _big_for(binding, N, batch_size, samples_dev)      # C4-sized states: the big resident form (csrc/big.hip)
        if big is not None:
            return big.train_steps(samples_dev, batch_size, loss_out, defer_step)
    if loss_out is None:
        loss_out = torch.empty(nsteps, dtype=torch.float32, device=dev)
    wso = workspace_for(dev, (n, m, d))
    ws = wso.ensure(N, batch_size, n, m, d, dev)
    if kernel_us is None:
        # the prepared call: tables, shape, hyper-parameters and workspace were bound once; six scalars cross the boundary
        if not samples_dev.is_contiguous():
            raise _lib.MfcdError("the HIP hot path needs contiguous tensors")
        binding._fast = _make_fast_call(binding, wso, batch_size, dev)
        binding._fast_stream = _lib.stream_ptr(dev)
        code = L.mfcd_train_call_run(binding._fast.handle, samples_dev.data_ptr(), N, binding.step,
                                     loss_out.data_ptr(), binding._fast_stream)
        if code:
            _lib.check(code)
        binding.advance(nsteps, defer_step)
        return loss_out if loss_out.shape[0] == nsteps else loss_out[:nsteps]
    lr, b1, b2, eps, wd = binding.hyper()
    args = ptrs + (_lib.ptr(samples_dev), N, batch_size, binding.step, n, m, d, lr, b1, b2, eps, wd,
                   loss_out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev))
    if dtype == torch.bfloat16:
        raise NotImplementedError("the timed diagnostic twin exists for fp32 factors only")
    import ctypes
    out = (ctypes.c_float * 3)()
    _lib.check(L.mfcd_train_steps_timed(*args, ctypes.cast(out, ctypes.c_void_p)))
    kernel_us[:] = [float(out[0]), float(out[1]), float(out[2])]
    binding.advance(nsteps, defer_step)
    return loss_out if loss_out.shape[0] == nsteps else loss_out[:nsteps]   # (a slice is ~1.6 us of host time)


def eval_batches(U, V, samples_dev, batch_size, want_p=False):
    """Forward + BCE per batch.  Returns (loss_per_batch fp32, correct_per_batch int32, p or None), on device."""
    L = _lib.load()
    _require_cuda_param(U, "U", _FACTOR_DTYPES)
    _require_cuda_param(V, "V", _FACTOR_DTYPES)
    (n, d), m = U.shape, V.shape[0]
    N = samples_dev.shape[0]
    nb = n_batches(N, batch_size)
    fn = L.mfcd_eval_batches_bf16 if U.dtype == torch.bfloat16 else L.mfcd_eval_batches
    loss = torch.empty(max(nb, 1), dtype=torch.float32, device=U.device)
    corr = torch.empty(max(nb, 1), dtype=torch.int32, device=U.device)
    p = torch.empty(max(N, 1), dtype=torch.float32, device=U.device) if want_p else None
    _lib.check(fn(_lib.ptr(U), _lib.ptr(V), _lib.ptr(samples_dev), N, batch_size, n, m, d,
                                   _lib.ptr(loss), _lib.ptr(corr), _lib.ptr(p), _lib.stream_ptr(U.device)))
    return loss[:nb], corr[:nb], (p[:N] if want_p else None)


def dense_grad_from_coefficients(U, V, samples_dev, g):
    """Dense [n,d], [m,d] fp32 gradients of a scalar loss w.r.t. U, V from g[t] = dLoss/dx_t (include/mfcd.h:
    mfcd_dense_grad_from_coefficients) — the backward of the forward kernel for ANY loss on its output."""
    L = _lib.load()
    _require_cuda_param(U, "U", _FACTOR_DTYPES)
    _require_cuda_param(V, "V", _FACTOR_DTYPES)
    if U.dtype != V.dtype:
        raise ValueError("U and V must share the storage dtype")
    if U.dtype != torch.float32:
        # the scatter kernel is fp32-only: widen exactly, accumulate in fp32, hand back gradients of the tables' dtype
        # (running the fp32 kernel over bf16 storage would read and write twice the tables' size)
        gU, gV = dense_grad_from_coefficients(U.float(), V.float(), samples_dev, g)
        return gU.to(U.dtype), gV.to(V.dtype)
    g = g.to(dtype=torch.float32).contiguous()
    if g.numel() != samples_dev.shape[0]:
        raise ValueError("one coefficient per sample record is needed")
    (n, d), m = U.shape, V.shape[0]
    gU, gV = torch.empty_like(U), torch.empty_like(V)
    _lib.check(L.mfcd_dense_grad_from_coefficients(_lib.ptr(U), _lib.ptr(V), _lib.ptr(samples_dev), _lib.ptr(g),
                                                   samples_dev.shape[0], n, m, d, _lib.ptr(gU), _lib.ptr(gV),
                                                   _lib.stream_ptr(U.device)))
    return gU, gV


class TripletForward(torch.autograd.Function):
    """p_t = sigmoid(U[u_t] . (V[i_t] - V[j_t])) (structure.py:773-795) with an autograd graph: forward is the fused
    gather + dot + sigmoid kernel, backward is sigmoid_backward (grad * (1 - p) * p, as ATen) followed by the
    scatter-accumulate kernel, so `loss.backward()` on `model(u, i, j)` works for any loss and any optimiser."""

    @staticmethod
    def forward(ctx, U, V, rec):
        N = rec.shape[0]
        _, _, p = eval_batches(U.detach(), V.detach(), rec, min(max(N, 1), 4096), want_p=True)
        ctx.save_for_backward(U, V, rec, p)
        return p

    @staticmethod
    def backward(ctx, grad_p):
        U, V, rec, p = ctx.saved_tensors
        g = (grad_p.float() * (1.0 - p) * p).contiguous()
        gU, gV = dense_grad_from_coefficients(U.detach(), V.detach(), rec, g)
        return gU, gV, None


def records_from_indices(u, i, j, n, m, device, z=None):
    """int32 [N, 4] device records from index tensors (any device, any integer dtype); raises IndexError for an index
    outside the tables, as U[u] / V[i] would (structure.py:787-789).  Python-style negative indices are accepted."""
    cols = []
    for t, size, what in ((u, n, "U"), (i, m, "V"), (j, m, "V")):
        t = torch.as_tensor(t).reshape(-1).to(device=device, dtype=torch.int64)
        if t.numel() and (int(t.min()) < -size or int(t.max()) >= size):      # one host sync, as the reference's gather
            raise IndexError(f"index out of range for {what} with {size} rows")
        cols.append(torch.where(t < 0, t + size, t).to(torch.int32))
    N = cols[0].numel()
    zf = torch.zeros(N, dtype=torch.float32, device=device) if z is None else \
        torch.as_tensor(z).reshape(-1).to(device=device, dtype=torch.float32)
    return torch.stack(cols + [zf.view(torch.int32)], dim=1).contiguous()


def fit_generic(model, train_loader, val_loader, optimizer, num_epochs, progress=None):
    """The reference's own loop (structure.py:840-868) for optimisers the fused step does not implement (anything but
    plain torch.optim.Adam): per batch zero_grad, forward kernel, F.binary_cross_entropy, backward (scatter kernel),
    optimizer.step(), loss.item() — every step a host round trip, as there; batch order and RNG use identical."""
    import torch.nn.functional as F
    U, V = model.U, model.V
    _require_cuda_param(U.data, "model.U")
    _require_cuda_param(V.data, "model.V")
    dev, n, m = U.device, U.shape[0], V.shape[0]
    train = SampleStore.from_loader(train_loader, n, m, dev)
    val = SampleStore.from_loader(val_loader, n, m, dev)
    train_losses, val_losses = [], []
    it = range(num_epochs) if progress is None else progress(range(num_epochs))
    for _ in it:
        order, bs = epoch_order(train_loader)
        rec = train.ordered(order)
        total, nb = 0.0, 0
        for off in range(0, rec.shape[0], bs):
            batch = rec[off:off + bs]
            optimizer.zero_grad()
            pred = TripletForward.apply(U, V, batch)
            loss = F.binary_cross_entropy(pred, batch[:, 3].view(torch.float32))
            loss.backward()
            optimizer.step()
            total += loss.item()
            nb += 1
        train_losses.append(total / max(nb, 1))
        vorder, vbs = epoch_order(val_loader)
        vl, _, _ = eval_batches(U.data, V.data, val.ordered(vorder), vbs)
        val_losses.append(python_float_sum(vl.cpu().numpy()) / max(len(vl), 1))
    return train_losses, val_losses


def fused_step_applies(model, optimizer):
    """True when `optimizer` is what the fused step implements (AdamBinding would accept it)."""
    try:
        AdamBinding(model, optimizer)
        return True
    except NotImplementedError:
        return False


def python_float_sum(x):
    """Sequential f64 accumulation of fp32 values, as `total += loss.item()` does (structure.py:852)."""
    a = np.asarray(x, dtype=np.float64)
    return float(np.cumsum(a)[-1]) if a.size else 0.0


def fit(model, train_loader, val_loader, optimizer, num_epochs, progress=None):
    """The epoch loop of train_model (structure.py:840-868) on the device path.
    Returns (train_losses, val_losses): per-epoch mean of batch means, Python floats."""
    binding = AdamBinding(model, optimizer)
    U, V = model.U.data, model.V.data
    dev = U.device
    n, m = U.shape[0], V.shape[0]
    train = SampleStore.from_loader(train_loader, n, m, dev)
    val = SampleStore.from_loader(val_loader, n, m, dev)
    if train.N and train_loader.batch_size:
        reserve_workspace(train.N, train_loader.batch_size, n, m, U.shape[1], dev)
    per_epoch_train, per_epoch_val = [], []
    it = range(num_epochs) if progress is None else progress(range(num_epochs))
    # The host draws the epoch orders in the reference's sequence (train_0, val_0, train_1, val_1, ...: every
    # iter(loader) consumes the global generator), but the NEXT epoch's stream is built on a side stream as soon as
    # this epoch's step kernel is enqueued, i.e. underneath it — and the next call's prologue is staged there too.
    pre = StreamPrefetch(dev)
    bs = None
    launched = []            # one event per enqueued train call: a staged prologue reuses the regions of the call two back
    try:
        if num_epochs > 0:
            order, bs = epoch_order(train_loader)           # structure.py:845 iter(train_loader)
            pre.start(train, order)
        nsteps = n_batches(train.N, bs) if bs else 0
        loss_bufs = torch.empty((max(num_epochs, 1), max(nsteps, 1)), dtype=torch.float32, device=dev)
        for e in it:
            stream = pre.take()
            per_epoch_train.append(train_steps(binding, stream, bs, loss_out=loss_bufs[e], defer_step=True))
            ev = torch.cuda.Event()
            ev.record()
            launched.append(ev)
            vorder, vbs = epoch_order(val_loader)           # structure.py:861 iter(val_loader)
            if e + 1 < num_epochs:
                order, bs = epoch_order(train_loader)       # next epoch's structure.py:845
                nxt = loss_bufs[e + 1]
                pre.start(train, order, after=launched[e - 1] if e >= 1 else None,
                          stage=lambda rec, side, nxt=nxt: stage_next_call(binding, rec, bs, nxt, side))
            vl, _, _ = eval_batches(U, V, val.ordered(vorder), vbs)
            per_epoch_val.append(vl)
    finally:
        binding.flush()     # an interrupt must not leave the moments ahead of the optimizer's `step` tensors
    # one device->host transfer for the whole run (the reference syncs every step at 852); the status word is
    # sticky, so an abort in ANY epoch surfaces here
    check_status()
    tl = [python_float_sum(t.cpu().numpy()) / max(len(t), 1) for t in per_epoch_train]
    vl = [python_float_sum(t.cpu().numpy()) / max(len(t), 1) for t in per_epoch_val]
    return tl, vl


def evaluate(model, test_loader):
    """evaluate_model (structure.py:881-921) on the device path → (mean batch BCE, accuracy)."""
    U, V = model.U.data, model.V.data
    store = SampleStore.from_loader(test_loader, U.shape[0], V.shape[0], U.device)
    order, bs = epoch_order(test_loader)
    loss, corr, _ = eval_batches(U, V, store.ordered(order), bs)
    total = int(order.numel())
    lsum = python_float_sum(loss.cpu().numpy())
    correct = int(corr.sum().item()) if total else 0
    return lsum / max(len(loss), 1), (correct / total if total > 0 else 0.0)
